"""(round 6) Premise check for the rotated step (AdamW of step k under the forward of step k + 1): the MMDiT-B batch-64 forward on one stream and
the HIP optimizer step (unscale + clip + AdamW: HBM-bound, 10.1 GB) on another, against the two run one after the other.  Timing only -- the forward reads the bf16 weight
copies while the update rewrites them.  With and without dynamic tile claiming (a GEMM workgroup needs a whole CU: beside the update's small workgroups it starts late).
    python tools/probes/adamw_overlap.py"""
import contextlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sd3_amd  # noqa: E402,F401
from sd3_amd import _lib  # noqa: E402
from sd3_amd.model_trainer import model_trainer  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev, positional_encoding="RoPE2d",
                 dim=768, num_heads=12, num_blocks=12)
with contextlib.redirect_stdout(sys.stderr):
    tr = model_trainer(net, batchSize=64, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999, warmup_steps=1000, use_lr_scheduler=False,
                       device=dev, saveDir="/tmp/_ao", numSaveSteps=10 ** 9, max_res=256, device_rng=True, use_ema=False, hip_optimizer=True)
net.train()
for s in range(3):
    tr.train_step(s + 1)
# gradients for the optimizer: one forward + backward without the step
tr.micro_step(final=True)
x0, c, cp = tr.data_source()
t = torch.rand(64, device=dev)
xt = torch.randn_like(x0.float())
sideB = torch.cuda.Stream()
sc = tr.grad_scaler


def fwd():
    with torch.no_grad():
        return net(xt, t, c.float().clone(), cp.float().clone(), None, None, None)


def opt():
    tr.optim.step_clipped(sc._scale if sc is not None and sc.is_enabled() else None, 1.0)


def timed(f, reps=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def serial():
    fwd()
    opt()


def concurrent():
    sideB.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sideB):
        opt()
    fwd()
    torch.cuda.current_stream().wait_stream(sideB)


for claiming in (0, 1):
    _lib.lib().mmdit_gemm_set_claiming(claiming)
    tf, to = timed(fwd), timed(opt)
    ts, tc = timed(serial), timed(concurrent)
    print(f"claiming {claiming}: forward {tf:6.2f} ms   optimizer {to:5.2f} ms   one after the other {ts:6.2f} ms   on two streams {tc:6.2f} ms", flush=True)
_lib.lib().mmdit_gemm_set_claiming(0)
