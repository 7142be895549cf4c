"""Are hipGraph MEMSET / D2D MEMCPY nodes kept in stream order behind the kernel nodes that precede them on the captured stream?
(Root cause of round 2's `final_loss: 0.0` under hipGraph replay: DESIGN.md 5.)

Captured chain (ONE stream, linear; `heavy` = a 4096^3 bf16 matmul in front):
   P = A + 0  ->  [heavy]  ->  A.fill_(1)  ->  hipMemsetAsync(A, 0)  ->  B = A + 0  ->  [heavy]  ->  Q = A + 0
Correct execution: B = 0, Q = 0 in every replay and P = 0 from the second replay on.
  memset dropped:                       B = 1, Q = 1, P = 1
  memset early (in front of the fill):  B = 1, Q = 1, P = 0 or 1
  memset late (behind the read of B):   B = 1, Q = 0, P = 0
Second chain: A2 = src + 0 [kernel] -> hipMemcpyAsync(M, A2, D2D) [memcpy node] -> C = M + 0 [kernel]; src changes between replays.
Host patterns between replays: burst (back to back) / torch.cuda.synchronize() / .item()."""
import ctypes

import torch

hip = ctypes.CDLL("libamdhip64.so")     # (already loaded by torch: the same runtime instance)
dev = torch.device("cuda:0")


def run(nbytes, pattern, heavy, via="hipMemsetAsync", rewrite=False):
    n = max(nbytes // 4, 1)
    A, A2, M = (torch.zeros(n, device=dev) for _ in range(3))
    P, B, Q, C = (torch.full((n,), -1.0, device=dev) for _ in range(4))
    src = torch.zeros(n, device=dev)
    X = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    Y = X @ X          # (library initialisation outside the capture)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        torch.add(A, 0.0, out=P)
        if heavy:
            Y = X @ X
        A.fill_(1.0)
        if via == "hipMemsetAsync":
            assert hip.hipMemsetAsync(ctypes.c_void_p(A.data_ptr()), 0, ctypes.c_size_t(nbytes), s) == 0
        else:
            A.zero_()          # torch's own zero fill (a kernel in this build)
        torch.add(A, 0.0, out=B)
        if heavy:
            Y = X @ X
        torch.add(A, 0.0, out=Q)
        A2 = A if rewrite else A2      # rewrite: the memset's target is written AGAIN later in the same graph (A = src + 0)
        torch.add(src, 0.0, out=A2)
        assert hip.hipMemcpyAsync(ctypes.c_void_p(M.data_ptr()), ctypes.c_void_p(A2.data_ptr()), ctypes.c_size_t(nbytes), 3, s) == 0   # hipMemcpyDeviceToDevice
        torch.add(M, 0.0, out=C)
    seen, bad_cpy = [], 0
    for i in range(8):
        src.fill_(float(i + 2))
        if pattern == "sync":
            torch.cuda.synchronize()
        elif pattern == "item":
            float(B[0])
        g.replay()
        torch.cuda.synchronize()
        seen.append("".join(str(int(float(t[0]))) for t in (P, B, Q)))
        bad_cpy += int((C != float(i + 2)).any())
    return seen, bad_cpy


print("per replay: P B Q (first element); correct = B and Q always 0 (P: 0, or with rewrite the src value of the previous replay)")
for rewrite in (False, True):
    for via in ("hipMemsetAsync", "tensor.zero_()"):
        for nbytes in (4, 4096, 4 << 20):
            for pattern in ("burst", "sync", "item"):
                for heavy in (True, False):
                    seen, bad = run(nbytes, pattern, heavy, via, rewrite)
                    ok = all(x[1] == "0" and x[2] == "0" for x in seen)
                    print(f"rewrite={rewrite!s:<5} {via:<15} {nbytes:>8} B  {pattern:<5} heavy={heavy!s:<5}: {'ok      ' if ok else 'MISORDER'} {' '.join(seen)}   wrong D2D memcpy results {bad}/8")
