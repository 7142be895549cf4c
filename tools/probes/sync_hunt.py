"""Which host call in the optimizer phase waits for the GPU?  Queue ~100 ms of GPU work, then time each call on the host."""
import os, sys, time, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sd3_amd  # noqa
from sd3_amd.optim import ClipAdamW
from sd3_amd import _lib

dev = torch.device("cuda:0")
ps = [torch.nn.Parameter(torch.randn(1000 + i, device=dev)) for i in range(400)]
for p in ps:
    p.grad = torch.randn_like(p)
opt = ClipAdamW(ps, lr=1e-3)
scale = torch.tensor(1024.0, device=dev)
opt.step_clipped(scale, 1.0)
torch.cuda.synchronize()
A = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)

def busy():
    for _ in range(40):
        torch.mm(A, A)

def timed(name, fn):
    torch.cuda.synchronize()
    busy()
    t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:<40} host {1e3 * (t1 - t0):8.3f} ms   (GPU drained after another {1e3 * (t2 - t1):7.3f} ms)")
    return r

steps = [opt.state[p]["step"] for p in ps]
out3 = torch.zeros(3, device=dev)
timed("busy only", lambda: None)
timed("step_clipped", lambda: opt.step_clipped(scale, 1.0))
timed("torch AdamW.step on the same object", lambda: opt.step())
timed("1.0 - found_inf", lambda: 1.0 - out3[1])
inc = 1.0 - out3[1]
timed("_foreach_add_(steps, tensor)", lambda: torch._foreach_add_(steps, inc))
timed("_foreach_add_(steps, 1.0)", lambda: torch._foreach_add_(steps, 1.0))
timed("out3.clone()", lambda: out3.clone())
flat = torch.zeros(400, device=dev)
timed("flat.add_(tensor)", lambda: flat.add_(inc))
sc = torch.amp.GradScaler("cuda")
sc.scale(torch.ones((), device=dev))
def upd():
    from torch.amp.grad_scaler import OptState
    st = sc._per_optimizer_states[id(opt)]
    st["found_inf_per_device"] = {dev: out3[1].clone()}
    st["stage"] = OptState.STEPPED
    sc.update()
timed("GradScaler.update()", upd)
timed("zero_grad", lambda: opt.zero_grad())
