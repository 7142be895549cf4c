"""Canary-guarded allocations (sd3_amd.debug_guard): the debug mode that looks for out-of-bounds writes of the HIP kernels where GPU
ASan is not available.  A positive control (a launch told to write past its output IS reported, with the launch named) and a clean
micro training step + inference forwards in every precision mode (tools/probes/guard_step.py runs the same at MMDiT-B / -L size;
its output of the final binary is committed under profiles/)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.weights import make_state_dict  # noqa: E402


def test_guard_reports_an_out_of_bounds_write():
    import sd3_amd  # noqa: F401
    from sd3_amd import _lib, debug_guard, ops
    reg = debug_guard.install(per_launch=True)
    try:
        src = torch.randn(1024 + 8, device="cuda")
        out = ops.torch.empty(1024, dtype=torch.bfloat16, device=src.device)        # guarded allocation (the proxy ops allocates through)
        assert not reg.verify("before")
        # a correct launch leaves the guards alone ...
        ops.check(_lib.lib().mmdit_cast(src.data_ptr(), _lib.F32, out.data_ptr(), _lib.BF16, 1024, torch.cuda.current_stream().cuda_stream), "mmdit_cast ok")
        assert not reg.found
        # ... one that is told to convert 8 elements too many writes 16 bytes into the back guard
        ops.check(_lib.lib().mmdit_cast(src.data_ptr(), _lib.F32, out.data_ptr(), _lib.BF16, 1024 + 8, torch.cuda.current_stream().cuda_stream), "mmdit_cast overrun")
        assert len(reg.found) == 1
        v = reg.found[0]
        assert v["side"] == "back" and v["first"] == 0 and v["bytes"] >= 12 and "mmdit_cast overrun" in v["after"]
    finally:
        debug_guard.uninstall()
    assert ops.torch is torch and not ops.NO_POOL


def test_training_step_and_inference_modes_are_clean_under_guards():
    import sd3_amd  # noqa: F401
    from sd3_amd import debug_guard
    from sd3_amd.model_trainer import model_trainer
    from sd3_amd.models.diff_model import diff_model
    cfg = dict(dim=128, num_heads=2, num_blocks=3)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                     positional_encoding="RoPE2d", **cfg)
    net.load_state_dict(make_state_dict(0, **cfg))
    tr = model_trainer(net, batchSize=4, accumulation_steps=1, totalSteps=10, lr=1e-3, ema_update_freq=1, ema_decay=0.9, warmup_steps=2,
                       use_lr_scheduler=False, device=dev, saveDir="/tmp/_t", numSaveSteps=100, max_res=96, device_rng=True, use_ema=False)
    l_plain = float(tr.train_step(1))
    reg = debug_guard.install()
    try:
        l_guard = float(tr.train_step(2))
        torch.cuda.synchronize()
        bad = reg.verify("training step")
        assert reg.seq > 100 and reg.launches > 50 and not bad, bad
        x = torch.randn((3, 16, 12, 20), device=dev)
        c, cp, t = torch.randn((3, 154, 2304), device=dev), torch.randn((3, 768), device=dev), torch.rand((3,), device=dev)
        net.eval()
        for prec in ("fast", "parity", "fp8", "mxfp8"):
            net.set_precision(prec)
            with torch.no_grad():
                v = net(x, t, c.clone(), cp.clone())
            torch.cuda.synchronize()
            bad = reg.verify(prec)
            assert torch.isfinite(v).all() and not bad, (prec, bad)
    finally:
        debug_guard.uninstall()
        net.set_precision("fast")
    assert 1e-3 < l_guard < 10 and 1e-3 < l_plain < 10
