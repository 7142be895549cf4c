"""Is the training step reproducible run-to-run, and does the (one-rank) reducer path change it?  Prints, per variant pair, the largest
parameter difference after two optimizer steps of the micro model with gradient accumulation 1 and 2."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
import torch.distributed as dist
from oracle.weights import make_state_dict
from sd3_amd.model_trainer import model_trainer
from sd3_amd.models.diff_model import diff_model

CFG = dict(dim=128, num_heads=2, num_blocks=3)


def run(force, accum):
    torch.manual_seed(0)
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu",
                     device=torch.device("cuda:0"), positional_encoding="RoPE2d", **CFG)
    net.load_state_dict(make_state_dict(0, **CFG))
    tr = model_trainer(net, batchSize=4, accumulation_steps=accum, totalSteps=10, lr=1e-3, ema_update_freq=1, ema_decay=0.9, warmup_steps=2,
                       use_lr_scheduler=False, device=torch.device("cuda:0"), saveDir="/tmp/_t", numSaveSteps=100, null_prob_pooled=0.1,
                       null_prob_gemma=0.316, null_prob_bert=0.316, max_res=128, device_rng=True, use_ema=False, force_reducer=force)
    grads = []
    for s in (1, 2):
        for k in range(accum):
            tr.micro_step(final=(k == accum - 1))
        if s == 1:
            grads = [p.grad.detach().clone() if p.grad is not None else None for p in net.parameters()]
        tr.optimizer_step(s)
    torch.cuda.synchronize()
    return [p.detach().clone() for p in net.parameters()], grads, [n for n, _ in net.named_parameters()]


def diff(a, b, names, what):
    worst = (0.0, "")
    for x, y, n in zip(a, b, names):
        if x is None:
            continue
        d = float((x.double() - y.double()).abs().max()) / (float(y.double().abs().max()) + 1e-30)
        if d > worst[0]:
            worst = (d, n)
    print(f"  {what}: worst max-abs difference relative to the tensor's max = {worst[0]:.3e} ({worst[1]})")


os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
for accum in (1, 2):
    p0, g0, names = run(False, accum)
    p1, g1, _ = run(False, accum)
    p2, g2, _ = run(True, accum)
    print(f"accumulation {accum}:")
    diff(g1, g0, names, "first-step gradients, same code twice     ")
    diff(g2, g0, names, "first-step gradients, reducer forced on   ")
    diff(p1, p0, names, "parameters after 2 steps, same code twice ")
    diff(p2, p0, names, "parameters after 2 steps, reducer forced on")
dist.destroy_process_group()
