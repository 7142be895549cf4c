"""Are hipGraph MEMSET / D2D MEMCPY nodes ordered behind the kernel nodes that precede them on the captured stream?

Captured chain (one stream, linear):   heavy matmul -> A.fill_(1)  [kernel]  ->  hipMemsetAsync(A, 0)  [memset node]  ->  B = A + 0  [kernel]
B must be 0 after every replay.  If the memset node runs early (before the fill kernel), B reads 1.
Second chain:  heavy matmul -> A = src + 0 [kernel] -> hipMemcpyAsync(M, A) [D2D memcpy node] -> C = M + 0 [kernel]; src changes between replays.
Replay patterns: back to back / torch.cuda.synchronize() before every replay / .item() before every replay."""
import ctypes
import sys

import torch

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")


def run(nbytes, pattern, heavy=True):
    n = max(nbytes // 4, 1)
    A = torch.zeros(n, device=dev)
    M = torch.zeros(n, device=dev)
    B = torch.full((n,), -1.0, device=dev)
    C = torch.full((n,), -1.0, device=dev)
    src = torch.zeros(n, device=dev)
    X = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    Y = X @ X          # (library initialisation outside the capture)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if heavy:
            Y = X @ X
        A.fill_(1.0)
        assert hip.hipMemsetAsync(ctypes.c_void_p(A.data_ptr()), 0, ctypes.c_size_t(nbytes), s) == 0
        torch.add(A, 0.0, out=B)
        torch.add(src, 0.0, out=A)
        assert hip.hipMemcpyAsync(ctypes.c_void_p(M.data_ptr()), ctypes.c_void_p(A.data_ptr()), ctypes.c_size_t(nbytes), 3, s) == 0   # hipMemcpyDeviceToDevice
        torch.add(M, 0.0, out=C)
    bad_set = bad_cpy = 0
    for i in range(12):
        src.fill_(float(i + 2))
        if pattern == "sync":
            torch.cuda.synchronize()
        elif pattern == "item":
            float(B[0])
        g.replay()
        torch.cuda.synchronize()
        bad_set += int((B != 0).any())
        bad_cpy += int((C != float(i + 2)).any())
    return bad_set, bad_cpy


for nbytes in (4, 4096, 4 << 20):
    for pattern in ("burst", "sync", "item"):
        for heavy in (True, False):
            print(f"memset/memcpy of {nbytes:>8} B, pattern {pattern:<5}, heavy kernel in front {heavy!s:<5}: wrong memset results {run(nbytes, pattern, heavy)[0]}/12, wrong memcpy results {run(nbytes, pattern, heavy)[1]}/12")
