"""Fourth probe: repeatability of the lost loss + contents of the loss block / the small segment at the end.  python graph_loss_probe4.py MODE n"""
import ctypes
import os
import struct
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
import sd3_amd  # noqa: E402,F401
from sd3_amd.model_trainer import model_trainer  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def peek(ptr, nbytes):
    buf = (ctypes.c_ubyte * nbytes)()
    rc = hip.hipMemcpy(buf, ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2)
    assert rc == 0, rc
    return bytes(buf)


mode, n = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                 positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, dim=768, num_heads=12, num_blocks=12)
tr = model_trainer(net, batchSize=64, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999,
                   warmup_steps=1000, use_lr_scheduler=False, device=dev, saveDir="/tmp/bench_ckpt", numSaveSteps=10 ** 9,
                   null_prob_pooled=0.1, null_prob_gemma=0.316, null_prob_bert=0.316, use_amp=True, max_res=256,
                   device_rng=True, use_ema=False)
net.train()
step = 0
for _ in range(5):
    step += 1
    tr.train_step(step)
ptrs = {}
real = tr._sample_conditioning


def wrapped(k):
    out = real(k)
    ptrs["cond"] = [x.data_ptr() for x in out]       # (addresses only: no extra references)
    return out


tr._sample_conditioning = wrapped
tr.capture_graph(step + 1)
gl = tr._graph_loss
print("ptrs: loss %#x  t/masks %s" % (gl.data_ptr(), [hex(p) for p in ptrs["cond"]]))
if mode == "sync":
    for _ in range(n):
        step += 1
        tr.train_step(step)
        torch.cuda.synchronize()
elif mode == "bench":
    for _ in range(2):
        step += 1
        tr.train_step(step)
    torch.cuda.synchronize()
    for _ in range(n):
        step += 1
        tr.train_step(step)
    torch.cuda.synchronize()
a = peek(gl.data_ptr() & ~0xfff, 4096)
off = gl.data_ptr() & 0xfff
print(f"[{mode} n={n}] loss dword {a[off:off+4].hex()} = {struct.unpack('<f', a[off:off+4])[0]:.6g}   float(loss) {float(gl)!r}")
for o in range(0, 4096, 512):
    print(f"   +{o:#06x}: {a[o:o+48].hex()}")
