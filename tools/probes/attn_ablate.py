"""(needs a probes build of the library: `bash tools/build_variant.sh probes -DMMDIT_PROBES` and MMDIT_LIB=tools/scratch/probes/libmmdit_hip.so)
Ablations of the attention forward kernel (mmdit_probe_attn_fwd_dbg): which part of the loop carries the time?"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd import _lib
B, H, N, M = 64, 12, 256, 154
S = N + M
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
Q, K, V = rnd(B, H, S, 64), rnd(B, H, S, 64), rnd(B, H, S, 64)
Ox = torch.empty((B, N, H * 64), dtype=torch.bfloat16, device="cuda")
Oc = torch.empty((B, M, H * 64), dtype=torch.bfloat16, device="cuda")
lse = torch.empty((B, H, S), dtype=torch.float32, device="cuda")
L = ctypes.CDLL(_lib.LIB_PATH)
fn = L.mmdit_probe_attn_fwd_dbg
vp = ctypes.c_void_p
fn.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, vp, vp, ctypes.c_int, vp]
names = {0: "full kernel", 1: "no exp", 32: "no rescale / row sums", 33: "no exp, no rescale / sums", 2: "no P V MFMAs", 4: "no K Q^T MFMAs", 6: "no MFMAs at all",
         39: "no MFMAs, no softmax arithmetic", 8: "no LDS fragment reads", 16: "no per-tile wait / barrier / DMA", 24: "no LDS reads, no barriers / DMA",
         57: "MFMAs only (no reads, barriers, softmax)", 63: "empty loop", 191: "empty loop, no epilogue", 64: "launch floor (return at entry)", 128: "full loop, no epilogue"}
for dbg, name in names.items():
    call = lambda: fn(Q.data_ptr(), K.data_ptr(), V.data_ptr(), B, H, S, N, 0.125, Ox.data_ptr(), Oc.data_ptr(), lse.data_ptr(), dbg, torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(f"dbg {dbg:2d}  {name:<44} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
