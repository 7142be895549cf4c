"""Are the epilogue phases of the SwiGLU-backward data gradient (csrc/gemm8p.hip epi8_swiglu_bwd: 256 KB of saved pre-activations read + 256 KB of d[g | u]
written per 256 x 256 tile) HBM-bound because every workgroup is in its epilogue at the same time?  Times the MMDiT-B launch (26240 x 3072 x 768) with the
workgroups' start times spread over `span` x 256 shader cycles (probes build, MMDIT_GEMM_DEBUG = span << 16; spread inside each XCD), one process per span.
    MMDIT_LIB=tools/scratch/probes/libmmdit_hip.so python tools/probes/swiglu_bwd_phase.py [span ...]      (bash tools/build_variant.sh probes -DMMDIT_PROBES)"""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import sd3_amd  # noqa: F401
    from sd3_amd import ops
    M, d, h = 26240, 768, 3072
    g = torch.Generator(device="cuda").manual_seed(0)
    dY = torch.randn((M, d), generator=g, device="cuda").to(torch.bfloat16)
    W3 = (torch.randn((d, h), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    GU = torch.randn((M, 2 * h), generator=g, device="cuda").to(torch.bfloat16)
    for _ in range(5):
        ops.gemm_swiglu_bwd([dict(A=dY, B=W3, aux=GU)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(20):
            ops.gemm_swiglu_bwd([dict(A=dY, B=W3, aux=GU)])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(f"span {int(os.environ.get('MMDIT_GEMM_DEBUG', '0')) >> 16:5d} x 256 cycles: {best:7.1f} us per launch")
else:
    spans = [int(x) for x in sys.argv[1:]] or [0, 45, 90, 180, 360, 720]
    for s in spans:
        env = dict(os.environ, MMDIT_EXPERIMENTS="1", MMDIT_GEMM_DEBUG=str(s << 16))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env)
