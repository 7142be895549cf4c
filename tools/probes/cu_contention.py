"""What a long-running kernel that holds C compute units (a collective's channels) does to the training step, whose persistent GEMM launches assume all 256.
An `occupy` kernel (mmdit_debug_occupy, formerly tools/probes/occupy.hip: C one-wave workgroups with a little LDS, spinning) runs on a side stream for the whole step; the step is timed
(hipGraph replay) for C = 0 / 8 / 16 / 32, and -- with MMDIT_CU_BUDGET support -- again with the GEMM grids capped at 256 - C.
python tools/probes/cu_contention.py"""
import ctypes
import contextlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sd3_amd  # noqa: E402,F401
from sd3_amd import _lib  # noqa: E402
from sd3_amd.model_trainer import model_trainer  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402

if hasattr(_lib.lib(), "mmdit_debug_occupy"):      # (round 6: the occupant kernel is a library entry point)
    def occupy(wgs, cycles, stream):
        return _lib.lib().mmdit_debug_occupy(wgs, cycles, stream)
else:
    occ = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "liboccupy.so"))
    occ.occupy.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]

    def occupy(wgs, cycles, stream):
        return occ.occupy(wgs, cycles, stream)
dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev, positional_encoding="RoPE2d",
                 dim=768, num_heads=12, num_blocks=12)
with contextlib.redirect_stdout(sys.stderr):
    tr = model_trainer(net, batchSize=64, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999, warmup_steps=1000, use_lr_scheduler=False,
                       device=dev, saveDir="/tmp/_cc", numSaveSteps=10 ** 9, max_res=256, device_rng=True, use_ema=False, hip_optimizer=True)
net.train()
if hasattr(_lib.lib(), "mmdit_gemm_set_claiming"):
    _lib.lib().mmdit_gemm_set_claiming(0 if "--static" in sys.argv else 1)      # (round 6) tile claiming: what model_trainer turns on when gradients are reduced
side = torch.cuda.Stream()
BWD_ONLY = "--bwd-only" in sys.argv
has_budget = hasattr(_lib.lib(), "mmdit_set_cu_budget")


def run(C, budget, steps=6):
    global tr
    if has_budget:
        _lib.lib().mmdit_set_cu_budget(budget)
    tr._graph = None
    step = [0]
    for _ in range(3):
        step[0] += 1
        tr.train_step(step[0])
    tr.capture_graph(step[0] + 1)
    for _ in range(2):
        step[0] += 1
        tr.train_step(step[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if C and not BWD_ONLY:
            occupy(C, int(26e-3 * 2.0e9), ctypes.c_void_p(side.cuda_stream))      # ~26 ms at ~2 GHz: the whole step
        step[0] += 1
        tr.train_step(step[0])
        if C and BWD_ONLY:
            # (--bwd-only) what a reducer does: nothing beside the forward (~8.5 ms), collectives beside the backward.  The replay is one asynchronous host
            # call; the occupant is launched from the host 8.5 ms later and holds its CUs for ~19 ms (to the end of the step)
            time.sleep(8.5e-3)
            occupy(C, int(19e-3 * 2.0e9), ctypes.c_void_p(side.cuda_stream))
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


BWD_ONLY = "--bwd-only" in sys.argv
only = [int(a) for a in sys.argv[1:] if a.isdigit()]      # e.g. `cu_contention.py 8` under rocprofv3: that occupancy only, no budget leg
robust = "--robust" in sys.argv or not only                # (round 6) + the data-parallel trainer's setting: the BACKWARD planned for 224 CUs, whatever C is
for C in (only or (0, 8, 16, 32)):
    net.bwd_cu_budget = None
    line = f"occupied CUs {C:2d}: step {run(C, 256):7.2f} ms with the whole-chip plan"
    if robust and hasattr(_lib.lib(), "mmdit_debug_occupy"):
        net.bwd_cu_budget = 224
        line += f", {run(C, 256):7.2f} ms with the weight gradients planned for 224 CUs (model_trainer reserved_cus = 32: the data-parallel setting, no knowledge of C)"
        net.bwd_cu_budget = None
    if has_budget and C and not only:
        line += f", {run(C, 256 - C):7.2f} ms with every launch planned AND capped at {256 - C}"
    print(line, flush=True)
if has_budget:
    _lib.lib().mmdit_set_cu_budget(256)
