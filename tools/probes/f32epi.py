import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sd3_amd
from sd3_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s, dt=torch.bfloat16: torch.randn(s, generator=g, device="cuda").to(dt)
def bench(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
Mx, Mc = 16384, 9856
for name, N, K in (("out", 768, 768), ("w3", 768, 3072)):
    Ax, Ac, W = rnd(Mx, K), rnd(Mc, K), rnd(N, K)
    rx, rc = rnd(Mx, N, dt=torch.float32), rnd(Mc, N, dt=torch.float32)
    gx, gc = rnd(64, N, dt=torch.float32), rnd(64, N, dt=torch.float32)
    ax, ac = torch.empty(Mx, N, dtype=torch.bfloat16, device="cuda"), torch.empty(Mc, N, dtype=torch.bfloat16, device="cuda")
    fn = lambda: ops.gemm_grouped([dict(A=Ax, B=W, gate=gx, rows_per_batch=256, residual=rx, aux=ax, out_dtype=torch.float32),
                                   dict(A=Ac, B=W, gate=gc, rows_per_batch=154, residual=rc, aux=ac, out_dtype=torch.float32)])
    t = bench(fn)
    print(f"{name} fwd gate+res+aux grouped cfg={os.environ.get('MMDIT_GEMM_CFG','auto')}: {t*1e6:.1f} us {2.0*(Mx+Mc)*N*K/t/1e12:.1f} TF")
