"""Algorithmic FLOPs of the MMDiT: GEMM + attention matmuls only, multiply-add = 2 FLOP; training = 3 x forward
(forward + data gradient + weight gradient; recompute is never counted).  Shared by the measurement scripts."""


def fwd_flops(d, blocks, N, M=154):
    """One forward per image: d = model width, N = image tokens = (res / 16)^2, M = text tokens."""
    S, h = N + M, 4 * d
    f = 2 * d * d + 2 * 768 * d + 2 * M * 2304 * d + 2 * N * 64 * d + 2 * N * d * d + 4 * d * d + 2 * N * d * 64
    for i in range(blocks):
        last = i == blocks - 1
        f += 2 * d * d + 4 * d * d * (3 if last else 4) + 2 * d * d * (2 if last else 4) + 8 * N * d * d + 2 * M * d * d * (3 if last else 4) + 4 * S * S * d
        f += 6 * d * h * N + (0 if last else 6 * d * h * M)
    return f


def train_flops(d, blocks, N, M=154):
    return 3 * fwd_flops(d, blocks, N, M)


if __name__ == "__main__":
    for name, d, b, res in (("B 256", 768, 12, 256), ("L 512", 1024, 24, 512), ("trained 256", 1216, 19, 256), ("trained 512", 1216, 19, 512), ("trained 1024", 1216, 19, 1024)):
        print(f"{name}: forward {fwd_flops(d, b, (res // 16) ** 2) / 1e9:.2f} GFLOP / image, training {train_flops(d, b, (res // 16) ** 2) / 1e9:.1f}")
