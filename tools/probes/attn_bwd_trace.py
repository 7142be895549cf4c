"""(needs a probes build of the library: `bash tools/build_variant.sh probes -DMMDIT_PROBES` and MMDIT_LIB=tools/scratch/probes/libmmdit_hip.so)
Per-phase cycle trace of the attention backward dK/dV kernel (mmdit_probe_attn_bwd_dkv_trace): where does a wave's lifetime go?
Stamps per wave: 0 start, 1 K/V fragments + first tile requested; per Q/dO tile: +0 barrier (everyone left the previous tile), +1 chunks landed
and written to LDS, +2 barrier, +3 first 32 queries' S/dP MFMAs + softmax arithmetic issued, +4 tile arithmetic issued; then loop end, stores done."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd import _lib, ops

B, H, N, M = 64, 12, 256, 154
if len(sys.argv) > 4:      # python tools/probes/attn_bwd_trace.py B H N M   (e.g. 4 19 4096 154: the 1024^2 stage; the first tiles are traced)
    B, H, N, M = (int(v) for v in sys.argv[1:5])
S = N + M
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
Q, K, V = rnd(B, H, S, 64), rnd(B, H, S, 64), rnd(B, H, S, 64)
dOx, dOc = rnd(B, N, H * 64), rnd(B, M, H * 64)
Ox, Oc, lse = ops.attn_fwd(Q, K, V, N, 0.125, 0)
delta = torch.empty((B, H, S), dtype=torch.float32, device="cuda")
dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
L = _lib.lib()
vp, ci = ctypes.c_void_p, ctypes.c_int
L.mmdit_attn_bwd.argtypes = [vp] * 9 + [ci] * 4 + [ctypes.c_float] + [vp] * 3 + [ci, vp]
st = torch.cuda.current_stream().cuda_stream
rc = L.mmdit_attn_bwd(Q.data_ptr(), K.data_ptr(), V.data_ptr(), Ox.data_ptr(), Oc.data_ptr(), dOx.data_ptr(), dOc.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                      B, H, S, N, 0.125, dQ.data_ptr(), dK.data_ptr(), dV.data_ptr(), 1, st)
assert rc == 0, rc
NS = 72
trace = torch.zeros((2048, 8, NS), dtype=torch.int64, device="cuda")
fn = ctypes.CDLL(_lib.LIB_PATH).mmdit_probe_attn_bwd_dkv_trace
fn.argtypes = [vp] * 7 + [ci] * 4 + [ctypes.c_float] + [vp] * 4
for _ in range(3):
    rc = fn(Q.data_ptr(), K.data_ptr(), V.data_ptr(), dOx.data_ptr(), dOc.data_ptr(), lse.data_ptr(), delta.data_ptr(), B, H, S, N, 0.125,
            dK.data_ptr(), dV.data_ptr(), trace.data_ptr(), st)
    assert rc == 0, rc
torch.cuda.synchronize()
t = trace.cpu().numpy().astype("float64")
nq_all = (S + 63) // 64
PT = 9                      # stamps per tile
nq = min(nq_all, (NS - 3) // PT)      # tiles that fit the 72 stamp slots of a wave (long sequences: the first 7)
last = 2 + PT * nq + 1 if nq == nq_all else 1 + PT * nq
nwg = ((S + 255) // 256) * B * H
t = t[:min(nwg, 2048)]
ids = np.arange(t.shape[0])
ntile = (S + 255) // 256
sel = t[((ids >> 3) % ntile) == 0] if (B * H) % 8 == 0 else t[(ids % ntile) == 0]      # workgroups of key tile 0 (all 256 keys real)
names = ["barrier 1 (leave previous tile)", "wait for chunks + LDS write", "barrier 2", "q 0-31: S/dP MFMAs issued", "q 0-31: softmax arithmetic", "q 0-31: dV/dK MFMAs issued",
         "q 32-63: S/dP MFMAs issued", "q 32-63: softmax arithmetic", "q 32-63: dV/dK MFMAs issued"]
print(f"B {B} H {H} S {S} ({nq_all} tiles, {nq} traced); workgroups traced: {t.shape[0]}; start -> last stamp (median over waves 0..7 of key-tile-0 workgroups): "
      f"{[int(np.median(sel[:, w, last] - sel[:, w, 0])) for w in range(8)]} ticks")
for w in (0, 4, 1, 7):
    full = sel[:, w, :]
    d = np.diff(full[:, :last + 1], axis=1)
    print(f"--- wave {w}: prologue {np.median(d[:, 0]):.0f}; median ticks per phase, tiles 1..{nq - 2} averaged (tile 0 and the ragged last tile apart)")
    mid = np.stack([d[:, 1 + PT * j:1 + PT * (j + 1)] for j in range(1, nq - 1)], 0).mean(0)
    for k, nm in enumerate(names):
        print(f"    {nm:<34} {np.median(mid[:, k]):8.0f}   (tile 0: {np.median(d[:, 1 + k]):6.0f}, last traced tile: {np.median(d[:, 1 + PT * (nq - 1) + k]):6.0f})")
    print(f"    {'tile total':<34} {np.median(mid.sum(1)):8.0f}" + (f";   loop end + stores {np.median(d[:, 1 + PT * nq] + d[:, 2 + PT * nq]):.0f}" if nq == nq_all else ""))
# one workgroup's two waves of a SIMD side by side: absolute times inside tile 3
wg = sel[5]
t3 = wg[:, 2 + PT * 3:2 + PT * 4 + 1] - wg[0, 2 + PT * 3]
print("tile 3 of one workgroup, stamp times relative to wave 0's tile start (rows: waves 0..7; columns: the 9 stamps + next tile's first):")
for w in range(8):
    print("   wave", w, [int(x) for x in t3[w]])
