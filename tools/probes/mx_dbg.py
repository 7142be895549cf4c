import sys, torch
sys.path.insert(0, "/root/repo")
import sd3_amd
from sd3_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
def rel(a, b): return float((a - b).norm() / b.norm())
for M, N, K in ((256, 256, 128), (256, 256, 64 * 6), (1024, 768, 768)):
    for mode in ("const", "rowvar", "kvar", "all"):
        A = torch.randn((M, K), generator=g, device="cuda").sign() * (300 + 100 * torch.rand((M, K), generator=g, device="cuda"))
        W = torch.randn((N, K), generator=g, device="cuda").sign() * (300 + 100 * torch.rand((N, K), generator=g, device="cuda"))
        if mode in ("rowvar", "all"):
            A = A * torch.exp2(torch.randint(-3, 4, (M, 1), generator=g, device="cuda").float())
        if mode in ("kvar", "all"):
            A = (A.reshape(M, K // 32, 32) * torch.exp2(torch.randint(-3, 4, (1, K // 32, 1), generator=g, device="cuda").float())).reshape(M, K)
        A, W = A.to(torch.bfloat16), W.to(torch.bfloat16)
        qa, sa = ops.quant_mxfp8(A)
        qw, sw = ops.quant_mxfp8(W)
        y = ops.gemm(qa, qw, out_dtype=torch.float32, scale_a=sa, scale_b=sw, scale_mode=1)
        ref = A.float() @ W.float().t()
        one = torch.ones(1, device="cuda")
        y0 = ops.gemm(qa, qw, out_dtype=torch.float32, scale_a=one, scale_b=one)
        print("   per-tensor path on the same codes (valid when all scales are 127): rel", rel(y0, ref), bool(torch.isfinite(y0).all()))
        print(M, N, K, mode, "rel", rel(y, ref), "finite", bool(torch.isfinite(y).all()), "sa uniq", sa[: (K // 32) * M].unique().tolist()[:8], "sw uniq", sw[: (K // 32) * N].unique().tolist()[:4])
