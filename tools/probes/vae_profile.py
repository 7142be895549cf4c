"""FLUX-VAE encode / decode at config 4's shape (512^2, batch 16): wall time and, under rocprofv3, the kernel table."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa: E402,F401
from oracle.vae_oracle import make_state_dict  # noqa: E402
from sd3_amd.vae import AutoencoderKL  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R = int(sys.argv[2]) if len(sys.argv) > 2 else 512
net = AutoencoderKL(device="cuda")
net.load_state_dict(make_state_dict(0))
x = torch.rand(B, 3, R, R, device="cuda") * 2 - 1
z = torch.randn(B, 16, R // 8, R // 8, device="cuda")
for name, fn in (("encode", lambda: net.encode(x).latent_dist.sample()), ("decode", lambda: net.decode(z).sample)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 3 * 1e-3
    print(f"VAE {R}x{R} batch {B}: {name} {t * 1e3:.1f} ms ({B / t:.1f} img/s)")
