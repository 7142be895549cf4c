"""Third probe: in the failing pattern (synchronize after every replay), what is in the loss block, and does it change after the synchronize returned?"""
import ctypes
import os
import sys
import time

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
stash = {}
real = tr._sample_conditioning


def wrapped(n):
    out = real(n)
    stash["cond"] = out
    return out


tr._sample_conditioning = wrapped
tr.capture_graph(step + 1)
gl = tr._graph_loss
t, m0, m1, m2 = stash["cond"]
print("ptrs: loss %#x  t %#x  masks %#x %#x %#x" % (gl.data_ptr(), t.data_ptr(), m0.data_ptr(), m1.data_ptr(), m2.data_ptr()))
for k in range(12):
    step += 1
    tr.train_step(step)
    torch.cuda.synchronize()
    a = peek(gl.data_ptr(), 64)
    time.sleep(0.3)
    b = peek(gl.data_ptr(), 64)
    import struct
    print(f"replay {k}: loss bytes right after synchronize {a[:8].hex()} = {struct.unpack('<f', a[:4])[0]:.5g}; 0.3 s later {b[:8].hex()} = {struct.unpack('<f', b[:4])[0]:.5g}; "
          f"block[4:64] {a[4:64].hex()[:48]}..  mask0 {bytes(m0.cpu().view(torch.uint8).tolist())[:8].hex()} mask1 {bytes(m1.cpu().view(torch.uint8).tolist())[:8].hex()}")
print("float(loss) at the end:", float(gl))
