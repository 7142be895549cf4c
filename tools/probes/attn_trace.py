"""(needs a probes build of the library: `bash tools/build_variant.sh probes -DMMDIT_PROBES` and MMDIT_LIB=tools/scratch/probes/libmmdit_hip.so)
Per-phase cycle trace of the 64-queries-per-wave attention forward (mmdit_probe_attn_fwd_trace): where does a wave's lifetime go?
Stamps per wave: 0 start, 1 Q loaded, then per KV tile [wait done, barrier done, QK done, softmax done], last-1 loop end, last stores done."""
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
trace = torch.zeros((2048, 4, 40), dtype=torch.int64, device="cuda")
L = ctypes.CDLL(_lib.LIB_PATH)
fn = L.mmdit_probe_attn_fwd_trace
vp = ctypes.c_void_p
fn.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, vp, vp, vp, vp]
for _ in range(3):
    fn(Q.data_ptr(), K.data_ptr(), V.data_ptr(), B, H, S, N, 0.125, Ox.data_ptr(), Oc.data_ptr(), lse.data_ptr(), trace.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
t = trace.cpu().numpy().astype("float64")
nkv = (S + 63) // 64
t0 = t[:, :, 0].min()
names = ["Q load"] + sum([[f"t{j} wait", f"t{j} barrier", f"t{j} QK", f"t{j} softmax", ] for j in range(nkv)], [])
# stamps: 0 start, 1 after Q, then per tile: wait, barrier, QK, softmax (PV ends at the next tile's 'wait' stamp), then loop end, stores
import numpy as np
full = t[::2, 0, :]          # wave 0 of the first workgroup of every (b, h): all 64 queries real
d = np.diff(full[:, :2 + 4 * nkv + 2], axis=1)
lab = ["Q load"]
for j in range(nkv):
    lab += [f"tile {j}: PV(prev)+vmcnt wait" if j else "tile 0: prologue DMA wait", f"tile {j}: barrier", f"tile {j}: QK^T", f"tile {j}: softmax"]
lab += ["PV(last)", "O / lse stores"]
print(f"workgroups traced: {t.shape[0]}, kernel span {(t[:, :, :2 + 4 * nkv + 2].max() - t0):.0f} ticks")
print(f"wave lifetime (median): {np.median(full[:, 1 + 4 * nkv + 2] - full[:, 0]):.0f} ticks")
for i, name in enumerate(lab):
    print(f"  {name:<34} median {np.median(d[:, i]):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f}")
starts = np.sort(t[:, 0, 0] - t0)
print("workgroup start times (ticks after the first), deciles:", [int(x) for x in np.percentile(starts, [0, 10, 25, 50, 75, 90, 100])])
ends = np.sort(t[:, 0, 1 + 4 * nkv + 2] - t0)
print("workgroup end times, deciles:", [int(x) for x in np.percentile(ends, [0, 10, 25, 50, 75, 90, 100])])
