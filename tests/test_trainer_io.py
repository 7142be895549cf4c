"""Trainer I/O: checkpoint streaming (helpers/checkpoint_stream.py) -- same six files as the reference's saveModel
(diff_model.py:489-536), written in the background from a snapshot taken at the call."""
import json
import os

import pytest
import torch

from oracle.weights import make_state_dict

CONFIGS = {"micro": dict(dim=128, num_heads=2, num_blocks=3)}


def test_streamer_snapshot_semantics_cpu(tmp_path):
    """The files hold the values at the time of save(), whatever happens to the tensors afterwards; nested containers and
    non-tensor leaves survive; a writer error surfaces in wait()."""
    import sd3_amd  # noqa: F401
    from sd3_amd.helpers.checkpoint_stream import CheckpointStreamer
    st = CheckpointStreamer("cpu")
    a, b = torch.arange(10.0), torch.ones(3, 4)
    obj = {"state": {0: {"step": torch.tensor(5.0), "exp_avg": a}}, "param_groups": [{"lr": 1e-3, "params": [0]}], "w": b}
    st.save([(obj, str(tmp_path / "sub" / "optim_5s.pkl"))], [({"dim": 256, "start_step": 5}, str(tmp_path / "sub" / "model_params_5s.json"))])
    a.mul_(0.0)
    b.add_(7.0)
    st.wait()
    got = torch.load(tmp_path / "sub" / "optim_5s.pkl", weights_only=False)
    assert torch.equal(got["state"][0]["exp_avg"], torch.arange(10.0)) and torch.equal(got["w"], torch.ones(3, 4))
    assert got["param_groups"] == [{"lr": 1e-3, "params": [0]}] and float(got["state"][0]["step"]) == 5.0
    assert json.load(open(tmp_path / "sub" / "model_params_5s.json")) == {"dim": 256, "start_step": 5}
    blocker = tmp_path / "file"
    blocker.write_text("x")
    st.save([(obj, str(blocker / "cannot" / "x.pkl"))])     # a directory cannot be created under a regular file
    with pytest.raises(RuntimeError):
        st.wait()
    st.wait()                                                # error reported once


@pytest.mark.gpu
def test_trainer_streams_checkpoint_while_training(tmp_path):
    """save_checkpoint(n) returns before anything is on disk, training goes on (parameters, moments and EMA change), and the six
    files hold exactly the state at the call: model, GPU-resident EMA, ClipAdamW state, scheduler, scaler, constructor JSON."""
    import sd3_amd  # noqa: F401
    from sd3_amd.model_trainer import model_trainer
    from sd3_amd.models.diff_model import diff_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                     positional_encoding="RoPE2d", **CONFIGS["micro"])
    net.load_state_dict(make_state_dict(0, **CONFIGS["micro"]))
    tr = model_trainer(net, batchSize=4, accumulation_steps=1, totalSteps=10, lr=1e-3, ema_update_freq=1, ema_decay=0.9, warmup_steps=2,
                       use_lr_scheduler=True, device=dev, saveDir=str(tmp_path), numSaveSteps=100, max_res=128, device_rng=True, use_ema=True)
    assert tr.ckpt_stream is not None and tr._ema_gpu is not None
    for s in (1, 2):
        tr.train_step(s)
        tr.update_ema()
    want_model = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
    want_ema = {k: v.detach().clone() for k, v in tr.ema_state_dict().items()}
    want_opt = {i: {k: v.detach().clone() for k, v in st.items()} for i, st in tr.optim.state_dict()["state"].items()}
    want_scale = tr.grad_scaler.state_dict()
    tr.save_checkpoint(2)
    for s in (3, 4):                       # keeps training while the checkpoint is written
        tr.train_step(s)
        tr.update_ema()
    tr.ckpt_stream.wait()
    names = ["model_2s.pkl", "model_ema_2s.pkl", "optim_2s.pkl", "scheduler_2s.pkl", "scaler_2s.pkl", "model_params_2s.json"]
    assert sorted(os.listdir(tmp_path)) == sorted(names)
    got = torch.load(tmp_path / "model_2s.pkl", map_location=dev, weights_only=False)
    assert list(got) == list(want_model) and all(torch.equal(got[k], want_model[k]) for k in got)
    assert any(not torch.equal(tr.model.state_dict()[k], want_model[k]) for k in got)     # (training did move on)
    ema = torch.load(tmp_path / "model_ema_2s.pkl", map_location=dev, weights_only=False)
    assert list(ema) == list(want_ema) and all(torch.equal(ema[k].to(dev), want_ema[k].to(dev)) for k in ema)
    opt = torch.load(tmp_path / "optim_2s.pkl", map_location=dev, weights_only=False)
    assert set(opt["state"]) == set(want_opt)
    for i, st in opt["state"].items():
        assert all(torch.equal(st[k], want_opt[i][k]) for k in ("step", "exp_avg", "exp_avg_sq")) and float(st["step"]) == 2.0
    assert torch.load(tmp_path / "scaler_2s.pkl", weights_only=False)["scale"] == want_scale["scale"]
    assert json.load(open(tmp_path / "model_params_2s.json"))["start_step"] == 2
    # and the files resume a trainer exactly like the reference's (strict load + optimizer / scheduler / scaler state)
    net2 = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                      positional_encoding="RoPE2d", **CONFIGS["micro"])
    net2.loadModel(str(tmp_path), "model_2s.pkl", "model_params_2s.json")
    assert all(torch.equal(v, want_model[k]) for k, v in net2.state_dict().items())


@pytest.mark.parametrize("h,w", [(32, 32), (24, 40), (16, 16), (40, 24)])
def test_inf_padded_wire_format_roundtrip(h, w):
    """pad_latents / unpad_latents are the reference's loader -> model rank format: +inf padding to (max_res/8)^2
    (VAE_T5_CLIP.py:438) and the receiver's recovery, here checked against the reference's own expression
    x[x != inf].reshape(B, C, H - #inf rows, W - #inf cols) (model_trainer.py:362-370)."""
    import sd3_amd  # noqa: F401
    from sd3_amd.helpers.wire_format import pad_latents, unpad_latents
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn((3, 16, h, w), generator=g).to(torch.bfloat16)
    p = pad_latents(x, 320)
    assert p.shape == (3, 16, 40, 40) and (p[:, :, h:, :] == float("inf")).all() and (p[:, :, :, w:] == float("inf")).all()
    orig_shape = (3, 16, p.shape[2] - (p[0, 0] == torch.inf).sum(-2)[0].item(), p.shape[3] - (p[0, 0] == torch.inf).sum(-1)[0].item())
    ref = p[p != torch.inf].reshape(orig_shape)
    got = unpad_latents(p)
    assert got.shape == x.shape and torch.equal(got, ref) and torch.equal(got, x)
    with pytest.raises(RuntimeError):
        pad_latents(x, 8 * min(h, w) - 8)


@pytest.mark.gpu
def test_trainer_consumes_inf_padded_bucketed_batches():
    """A data source in the reference's wire format -- aspect-ratio buckets that change from batch to batch, each padded with
    +inf to (max_res/8)^2 -- drives the trainer (inf_padded_latents=True): the model sees the unpadded latent of every bucket
    and the loss stays finite."""
    import sd3_amd  # noqa: F401
    from sd3_amd.helpers.wire_format import pad_latents
    from sd3_amd.model_trainer import model_trainer
    from sd3_amd.models.diff_model import diff_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = diff_model(inCh=16, class_dim=768, patch_size=2, hidden_scale=4.0, attn_type="softmax_flash", MLP_type="swiglu", device=dev,
                     positional_encoding="RoPE2d", **CONFIGS["micro"])
    net.load_state_dict(make_state_dict(0, **CONFIGS["micro"]))
    buckets, g, calls = [(16, 16), (12, 20), (20, 12), (8, 16)], torch.Generator(device="cuda").manual_seed(3), []

    def source():
        h, w = buckets[len(calls) % len(buckets)]
        calls.append((h, w))
        x = torch.randn((4, 16, h, w), generator=g, device=dev).to(torch.bfloat16)
        return (pad_latents(x, 160), torch.randn((4, 154, 2304), generator=g, device=dev).to(torch.bfloat16),
                torch.randn((4, 768), generator=g, device=dev).to(torch.bfloat16))

    tr = model_trainer(net, batchSize=4, accumulation_steps=1, totalSteps=10, lr=1e-3, ema_update_freq=1, ema_decay=0.9, warmup_steps=2,
                       use_lr_scheduler=True, device=dev, saveDir="/tmp/_t", numSaveSteps=100, max_res=160, device_rng=True, use_ema=False,
                       data_source=source, inf_padded_latents=True)
    seen, fwd = [], net.forward

    def spy(x_t, *a, **kw):
        seen.append(tuple(x_t.shape[-2:]))
        return fwd(x_t, *a, **kw)

    net.forward = spy
    losses = [float(tr.train_step(s)) for s in range(1, 6)]
    assert seen == [buckets[i % 4] for i in range(5)]
    assert all(l == l and abs(l) < 1e3 for l in losses)
