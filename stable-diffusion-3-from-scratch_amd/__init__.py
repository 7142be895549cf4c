"""MI355X-native MMDiT flow-matching training path (drop-in for the reference's src/models + src/blocks).

Layout:  csrc/ (HIP kernels + C ABI, built into libmmdit_hip.so) · _lib.py/ops.py (ctypes binding) ·
engine.py (explicit forward/backward kernel schedules) · blocks/, models/, model_trainer.py (host-side
mirror of the reference's module interface).
"""
__version__ = "0.1.0"
