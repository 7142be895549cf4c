"""(round 4) FLUX-VAE encode of 16 images at 512^2 (config 4's data path), timed; run under rocprofv3 --kernel-trace for the kernel table."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa: E402,F401
from sd3_amd.helpers.latent_source import ImageLatentSource  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
src = ImageLatentSource.synthetic(B, 512, 768, "cuda:0")
for _ in range(2):
    src()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    src()
torch.cuda.synchronize()
print(f"VAE encode 512^2 batch {B}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per batch (including the synthetic image draw)")
