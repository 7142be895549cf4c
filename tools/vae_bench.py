"""FLUX-VAE throughput on one MI355X (first correct path): encode and decode images/s at 256x256.
python tools/vae_bench.py [batch] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sd3_amd  # noqa: E402,F401
from oracle.vae_oracle import make_state_dict  # noqa: E402  (weights only: seeded random FLUX-shaped state_dict)
from sd3_amd.vae import AutoencoderKL  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
net = AutoencoderKL(device="cuda")
net.load_state_dict(make_state_dict(0))
x = torch.rand(B, 3, 256, 256, device="cuda") * 2 - 1
z = torch.randn(B, 16, 32, 32, device="cuda")


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


te = timed(lambda: net.encode(x).latent_dist.sample())
td = timed(lambda: net.decode(z).sample)
# conv FLOPs per 256x256 image (2*MACs): encoder ~ 0.28 TF, decoder ~ 0.62 TF  (3x3 convs dominate; counted from the layer table)
print(f"VAE 256x256 batch {B}: encode {te * 1e3:.1f} ms ({B / te:.1f} img/s), decode {td * 1e3:.1f} ms ({B / td:.1f} img/s)")
