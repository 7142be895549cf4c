"""Root-cause probe for `final_loss: 0.0` under hipGraph replay at MMDiT-B size (task 1).

  python tools/probes/graph_loss_probe.py [batch]

1. builds the bench's trainer (MMDiT-B, batch 64), runs eager steps, captures the step;
2. replays and prints: the capture-time loss tensor (`_graph_loss`), a copy made INSIDE the graph into a persistent buffer,
   the raw bytes of the 512-byte allocator block around the loss, and a parameter checksum (did the replays train?);
3. prints who lives next to the loss tensor in the graph's private pool (torch.cuda.memory snapshot with allocation stacks).
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
import sd3_amd  # noqa: E402,F401
from sd3_amd.model_trainer import model_trainer  # noqa: E402
from sd3_amd.models.diff_model import diff_model  # noqa: E402

B_CFG = dict(dim=768, num_heads=12, num_blocks=12)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")


def peek(ptr, nbytes):
    buf = (ctypes.c_ubyte * nbytes)()
    torch.cuda.synchronize()
    rc = hip.hipMemcpy(buf, ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2)   # hipMemcpyDeviceToHost
    assert rc == 0, rc
    return bytes(buf)


torch.manual_seed(1234)
net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                 positional_encoding="RoPE2d", checkpoint_MLP=False, checkpoint_attn=False, **B_CFG)
tr = model_trainer(net, batchSize=batch, accumulation_steps=1, totalSteps=10 ** 9, lr=1e-4, ema_update_freq=10 ** 9, ema_decay=0.999,
                   warmup_steps=1000, use_lr_scheduler=False, device=dev, saveDir="/tmp/bench_ckpt", numSaveSteps=10 ** 9,
                   null_prob_pooled=0.1, null_prob_gemma=0.316, null_prob_bert=0.316, use_amp=True, max_res=256,
                   device_rng=True, use_ema=False)
net.train()
step = 0
for _ in range(5):
    step += 1
    l = tr.train_step(step)
    print(f"eager step {step}: loss {float(l):.5f} dtype {l.dtype}")


def checksum():
    return float(sum(p.detach().double().abs().sum() for p in net.parameters()))


torch.cuda.memory._record_memory_history(max_entries=200000)
tr.capture_graph(step + 1)
gl = tr._graph_loss
print(f"_graph_loss: ptr {gl.data_ptr():#x} dtype {gl.dtype} shape {tuple(gl.shape)} storage_offset {gl.storage_offset()} "
      f"storage nbytes {gl.untyped_storage().nbytes()}")
c0 = checksum()
for k in range(3):
    step += 1
    l = tr.train_step(step)
    raw = peek(gl.data_ptr() & ~511, 512)
    off = gl.data_ptr() & 511
    nz = [i for i in range(0, 512, 4) if raw[i:i + 4] != b"\0\0\0\0"]
    print(f"replay {k}: float(_graph_loss) {float(l)!r}  loss_out {getattr(tr, 'last_loss_value', None)}  block offset {off}, nonzero dwords at {nz[:16]}")
c1 = checksum()
print(f"parameter checksum before/after 3 replays: {c0:.6f} -> {c1:.6f}  (moved: {c0 != c1})")

# neighbours of the loss in the graph's private pool
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
p = gl.data_ptr()
for seg in snap["segments"]:
    a0 = seg["address"]
    if not (a0 <= p < a0 + seg["total_size"]):
        continue
    print(f"segment {a0:#x} size {seg['total_size']} pool {seg.get('segment_pool_id')} type {seg.get('segment_type')}")
    addr = a0
    rows = []
    for blk in seg["blocks"]:
        rows.append((addr, blk))
        addr += blk["size"]
    idx = next(i for i, (a, b) in enumerate(rows) if a <= p < a + b["size"])
    for a, b in rows[max(0, idx - 4): idx + 5]:
        fr = [f for f in b.get("frames", []) if "site-packages/torch" not in f["filename"]][:4]
        where = " <- ".join(f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in fr)
        print(f"  {'>>' if a <= p < a + b['size'] else '  '} {a:#x} size {b['size']:>9} req {b.get('requested_size', 0):>9} {b['state']:<18} {where}")
