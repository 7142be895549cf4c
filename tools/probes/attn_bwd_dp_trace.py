"""(probes build) Per-phase cycle stamps of the DE-PHASED dK/dV kernel (mmdit_probe_attn_bwd_dkv_dp_trace): per block of 32 queries
V-phase arithmetic | barrier | row reads + DMA issue | dV / dK MFMAs | S / dP MFMAs + barrier, for the waves of both groups.
MMDIT_LIB=tools/scratch/probes/libmmdit_hip.so python tools/probes/attn_bwd_dp_trace.py [B H N M]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd import _lib, ops

B, H, N, M = 64, 12, 256, 154
if len(sys.argv) > 4:
    B, H, N, M = (int(v) for v in sys.argv[1:5])
S = N + M
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
Q, K, V = rnd(B, H, S, 64), rnd(B, H, S, 64), rnd(B, H, S, 64)
dOx, dOc = rnd(B, N, H * 64), rnd(B, M, H * 64)
Ox, Oc, lse = ops.attn_fwd(Q, K, V, N, 0.125, 0)
dQ, dK, dV = ops.attn_bwd(Q, K, V, Ox, Oc, dOx, dOc, lse, N, 0.125, torch.bfloat16)     # (fills delta through the dQ kernel)
delta = torch.empty((B, H, S), dtype=torch.float32, device="cuda")
L = _lib.lib()
vp, ci = ctypes.c_void_p, ctypes.c_int
st = torch.cuda.current_stream().cuda_stream
L.mmdit_attn_bwd.argtypes = [vp] * 9 + [ci] * 4 + [ctypes.c_float] + [vp] * 3 + [ci, vp]
assert L.mmdit_attn_bwd(Q.data_ptr(), K.data_ptr(), V.data_ptr(), Ox.data_ptr(), Oc.data_ptr(), dOx.data_ptr(), dOc.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                        B, H, S, N, 0.125, dQ.data_ptr(), dK.data_ptr(), dV.data_ptr(), 1, st) == 0
NS, PT = 72, 5
trace = torch.zeros((2048, 8, NS), dtype=torch.int64, device="cuda")
fn = ctypes.CDLL(_lib.LIB_PATH).mmdit_probe_attn_bwd_dkv_dp_trace
fn.argtypes = [vp] * 7 + [ci] * 4 + [ctypes.c_float] + [vp] * 4
for _ in range(3):
    assert fn(Q.data_ptr(), K.data_ptr(), V.data_ptr(), dOx.data_ptr(), dOc.data_ptr(), lse.data_ptr(), delta.data_ptr(), B, H, S, N, 0.125,
              dK.data_ptr(), dV.data_ptr(), trace.data_ptr(), st) == 0
torch.cuda.synchronize()
t = trace.cpu().numpy().astype("float64")
nblk_all = (S + 31) // 32
nb = min(nblk_all, (NS - 2) // PT)
ntile = (S + 255) // 256
nwg = ntile * B * H
t = t[:min(nwg, 2048)]
ids = np.arange(t.shape[0])
sel = t[((ids >> 3) % ntile) == 0] if (B * H) % 8 == 0 else t[(ids % ntile) == 0]
names = ["V phase: softmax-backward arithmetic", "wait for tile + barrier A", "row reads + DMA issue", "dV / dK MFMAs issued", "tr reads + S / dP MFMAs + barrier B"]
print(f"B {B} H {H} S {S}: {nblk_all} blocks of 32 queries, {nb} traced; workgroups {t.shape[0]}")
for w in (0, 4, 1, 5):
    d = np.diff(sel[:, w, :2 + PT * nb], axis=1)
    mid = np.stack([d[:, 1 + PT * j:1 + PT * (j + 1)] for j in range(2, nb - 1)], 0).mean(0)
    print(f"--- wave {w} (group {w >> 2}): prologue {np.median(d[:, 0]):.0f}; median cycles per phase, blocks 2..{nb - 2} averaged")
    for k, nm in enumerate(names):
        print(f"    {nm:<40} {np.median(mid[:, k]):8.0f}")
    print(f"    {'block total':<40} {np.median(mid.sum(1)):8.0f}   (two blocks = one 64-query tile: {2 * np.median(mid.sum(1)):.0f})")
wg = sel[5]
base = wg[0, 1 + PT * 4]
print("blocks 4-5 of one workgroup, stamp times relative to wave 0's block-4 start (rows: waves 0..7; columns: 5 stamps per block x 2 blocks):")
for w in range(8):
    print("   wave", w, [int(x - base) for x in wg[w, 1 + PT * 4:1 + PT * 6 + 1]])
