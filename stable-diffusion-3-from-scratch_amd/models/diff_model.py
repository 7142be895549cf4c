"""MMDiT velocity network: drop-in mirror of the reference's src/models/diff_model.py.

Same constructor kwargs (diff_model.py:83), forward signature (264), noise_batch (229),
sample_imgs (368), saveModel / loadModel checkpoint layout (489-578), state_dict keys, parameter
registration order and behavioural quirks (in-place null masking of the caller's c / c_pooled,
JSON always records device "cpu", attn_type string preserved).  The arithmetic of forward() and of
its backward runs in libmmdit_hip.so through sd3_amd.engine; there is no PyTorch/CPU fallback.
"""
import json
import os
from types import SimpleNamespace as NS

import numpy as np
import torch
from torch import nn

from .. import engine
from ..blocks.ImagePositionalEncoding import PatchEmbed
from ..blocks.Norm import Norm
from ..blocks.PositionalEncoding import PositionalEncoding
from ..blocks.Transformer_Block_Dual import Transformer_Block_Dual
from ..packing import Pack
from .. import ops


class _MMDiTFn(torch.autograd.Function):
    """Whole-network forward/backward as explicit kernel schedules (engine.model_fwd / model_bwd)."""

    @staticmethod
    def forward(ctx, net, x_t, t, c, c_pooled, *params):
        m = net._mode()
        W = net.weights(m)
        rope = net.blocks[0].attn.rotary_emb.tables(x_t.shape[-2] // 2, x_t.shape[-1] // 2, x_t.device)
        v, sv = engine.model_fwd(m, W, x_t, t, c, c_pooled, rope, keep=net._keep_saved)
        if any(ctx.needs_input_grad):
            ctx.net, ctx.m, ctx.sv, ctx.rope = net, m, sv, rope
        return v

    @staticmethod
    def backward(ctx, dv):
        net, m = ctx.net, ctx.m
        W = net.weights(m)
        red = net.grad_reducer
        # Data parallel: the collectives of the reducer's side stream run beside THIS pass and hold compute units the GEMMs cannot share.  The weight-gradient
        # launches (a block's eight, grouped: one round of whole-K tiles + a split tail) are then PLANNED for fewer compute units (net.bwd_cu_budget, set by
        # model_trainer: ops.WGRAD_CU_BUDGET -> mmdit_set_cu_budget around those launches) while their grids still cover the device and claim their tiles
        # (csrc/gemm8p.hip): measured beside an occupant of 8 CUs 7.05 -> 5.61 ms per step, alone 5.60 -> 5.92 (profiles/r06_robust_split_ab.txt).  Every
        # other launch keeps the whole-chip plan: the same treatment of the one-round data gradients measured slower in both cases.  Frozen with a capture.
        budget = getattr(net, "bwd_cu_budget", None)
        before, ops.WGRAD_CU_BUDGET = ops.WGRAD_CU_BUDGET, (int(budget) if budget else None)
        try:
            g = engine.model_bwd(m, W, ctx.sv, dv, ctx.rope, on_grads=red.add_bucket if red is not None else None)
        finally:
            ops.WGRAD_CU_BUDGET = before
        ctx.sv = None
        out = {}
        net.scatter_grads(g, out)
        return (None, None, None, None, None) + tuple(out.get(id(p)) for p in net._param_list())


class diff_model(nn.Module):
    # inCh - number of latent channels; class_dim - pooled text embedding width; patch_size - 2;
    # dim / hidden_scale / num_heads / num_blocks - transformer geometry (dim = 64*num_heads);
    # attn_type - "softmax" | "softmax_flash"; MLP_type - "swiglu" | "gelu"; positional_encoding - "RoPE2d"
    def __init__(self, inCh, class_dim, patch_size, dim, hidden_scale, num_heads, attn_type, MLP_type, num_blocks, device, positional_encoding,
                 max_res_orig=256, max_res=256, update_max_res=False, kv_merge_attn=False, qk_half_dim=False, text_loss=False,
                 checkpoint_MLP=True, checkpoint_attn=True, start_step=0, wandb_id=None):
        super(diff_model, self).__init__()
        self.update_max_res = update_max_res
        self.max_res = max_res
        self.RoPE_Scale = max_res_orig / max_res
        self.inCh, self.class_dim, self.patch_size = inCh, class_dim, patch_size
        self.start_step, self.wandb_id, self.text_loss = start_step, wandb_id, text_loss

        assert positional_encoding in ["absolute", "RoPE", "NoPE", "RoPE2d", "RoPE2dV2"], "positional_encoding must be 'absolute', 'RoPE', or 'NoPE' or 'RoPE2d' or 'RoPE2dV2'"
        assert MLP_type in ["gelu", "swiglu", "swiglu_old"]
        if positional_encoding != "RoPE2d" or text_loss or kv_merge_attn or qk_half_dim or MLP_type == "swiglu_old":
            raise RuntimeError("diff_model (HIP path): only the trained configuration is implemented: positional_encoding='RoPE2d', "
                               "MLP_type in {'swiglu','gelu'}, text_loss=False, kv_merge_attn=False, qk_half_dim=False")
        if patch_size != 2:
            raise RuntimeError("diff_model (HIP path): patch_size must be 2 (Attention.py:178-179 hard-codes it too)")
        self.legacy_MLP = False

        self.defaults = {
            "inCh": inCh, "class_dim": class_dim, "patch_size": patch_size, "dim": dim, "hidden_scale": hidden_scale,
            "num_heads": num_heads, "attn_type": attn_type, "MLP_type": MLP_type, "num_blocks": num_blocks,
            "positional_encoding": positional_encoding, "max_res_orig": max_res_orig, "max_res": max_res,
            "kv_merge_attn": kv_merge_attn, "qk_half_dim": qk_half_dim, "text_loss": text_loss,
            "device": "cpu", "start_step": start_step, "wandb_id": wandb_id,
        }

        if type(device) is str:
            if device.lower() == "gpu":
                if torch.cuda.is_available():
                    dev = device.lower()
                    try:
                        local_rank = int(os.environ["LOCAL_RANK"])
                    except KeyError:
                        local_rank = 0
                    device = torch.device(f"cuda:{local_rank}")
                else:
                    dev = "cpu"
                    print("GPU not available, defaulting to CPU. Please ignore this message if you do not wish to use a GPU\n")
                    device = torch.device("cpu")
            else:
                dev = "cpu"
                device = torch.device("cpu")
            self.device, self.dev = device, dev
        else:
            self.device = device
            self.dev = "cpu" if device.type == "cpu" else "gpu"

        self.dim, self.num_heads, self.num_blocks = dim, num_heads, num_blocks
        self.blocks = nn.ModuleList([
            Transformer_Block_Dual(dim, c_dim=dim, hidden_scale=hidden_scale, num_heads=num_heads, attn_type=attn_type, MLP_type=MLP_type,
                                   positional_encoding=positional_encoding, RoPE_Scale=self.RoPE_Scale, kv_merge_attn=kv_merge_attn,
                                   qk_half_dim=qk_half_dim, checkpoint_MLP=checkpoint_MLP, checkpoint_attn=checkpoint_attn, layer_idx=i,
                                   last=(i == num_blocks - 1 and not self.text_loss)).to(device)
            for i in range(num_blocks)
        ])
        self.t_emb = PositionalEncoding(dim, device=device).to(device)
        self.t_emb2 = nn.Linear(dim, dim, bias=False).to(device)
        self.cond_MLP = nn.Linear(self.class_dim, dim, bias=False).to(device)
        self.text_hidden_shape = 2304
        self.c_proj = nn.Linear(self.text_hidden_shape, dim, bias=False).to(device)
        self.c_proj2 = nn.Linear(self.text_hidden_shape, dim, bias=False).to(device)
        self.pre_c_norm = nn.RMSNorm(self.text_hidden_shape).to(device)
        self.pre_c_norm2 = nn.RMSNorm(self.text_hidden_shape).to(device)
        self.learnable_scalar = nn.Parameter(torch.tensor([0.01], dtype=torch.float, device=device), requires_grad=True)
        self.learnable_scalar2 = nn.Parameter(torch.tensor([0.01], dtype=torch.float, device=device), requires_grad=True)
        self.patch_emb = nn.Linear(dim, dim).to(device)
        self.pos_enc = PatchEmbed(height=256, width=256, patch_size=self.patch_size, in_channels=inCh, embed_dim=dim, layer_norm=False, flatten=True,
                                  bias=False, interpolation_scale=1, pos_embed_type=positional_encoding, pos_embed_max_size=256).to(device)
        self.out_norm = Norm(dim, dim).to(device)
        self.out_proj = nn.Linear(dim, inCh * patch_size * patch_size).to(device)
        self.time_scale = nn.Parameter(torch.tensor([1000.0], dtype=torch.float, device=device), requires_grad=True)

        # "fast" = bf16 MFMA operands (training default, = the reference's bf16 autocast);
        # "parity" = split-bf16 GEMMs + reference-rounding attention for the 1e-3 golden check.
        self.precision = "fast"
        self._keep_saved = True
        self.grad_reducer = None   # set by model_trainer for data-parallel runs (sd3_amd.reducer.GradReducer)
        self.bwd_cu_budget = None      # data parallel: compute units the backward's weight-gradient planner counts on (model_trainer reserved_cus)
        self._packs = NS(Wt=Pack([self.t_emb2.weight]), Wcond=Pack([self.cond_MLP.weight]), Wc1=Pack([self.c_proj.weight]),
                         Wc2=Pack([self.c_proj2.weight]), Wpatch=Pack([self.pos_enc.proj.weight]), Wpe=Pack([self.patch_emb.weight]),
                         Wmod_out=Pack([self.out_norm.c_shift.weight, self.out_norm.c_scale.weight]), Wout=Pack([self.out_proj.weight]))

    # ------------------------------------------------------------------------------------------
    def _mode(self):
        return {"fast": engine.FAST, "parity": engine.PARITY, "fp8": engine.FP8, "mxfp8": engine.MXFP8}[self.precision]

    def set_precision(self, precision: str):
        """"fast" (bf16, training and inference), "parity" (fp32-exact GEMMs, the 1e-3 golden check) or "fp8": inference
        only -- e4m3 operands with per-tensor scales for the QKV / out / MLP GEMMs of every block (BASELINE config 5)."""
        assert precision in ("fast", "parity", "fp8", "mxfp8")
        engine.FP8._q.clear()
        engine.MXFP8._q.clear()
        for mod in self.modules():
            if hasattr(mod, "precision"):
                mod.precision = "fast" if precision in ("fp8", "mxfp8") else precision     # stand-alone sub-modules have no fp8 path
        self.precision = precision
        return self

    def _param_list(self):
        return [p for p in self.parameters() if p.requires_grad]

    def weights(self, m):
        P = self._packs
        dev = self.time_scale.device
        return NS(dim=self.dim, heads=self.num_heads, split=77,
                  time_scale=self.time_scale.detach(), denom=self.t_emb._denom_on(dev),
                  Wt=P.Wt.get(m), Wcond=P.Wcond.get(m), Wc1=P.Wc1.get(m), Wc2=P.Wc2.get(m),
                  wn1=self.pre_c_norm.weight.detach(), wn2=self.pre_c_norm2.weight.detach(),
                  s1=self.learnable_scalar.detach(), s2=self.learnable_scalar2.detach(),
                  Wpatch=P.Wpatch.get(m), Wpe=P.Wpe.get(m), bpe=self.patch_emb.bias.detach(),
                  Wmod_out=P.Wmod_out.get(m), Wout=P.Wout.get(m), bout=self.out_proj.bias.detach(),
                  blocks=[b.weights(m) for b in self.blocks])

    def scatter_grads(self, g, out: dict):
        P = self._packs
        for k in ("Wt", "Wcond", "Wc1", "Wc2", "Wpatch", "Wpe", "Wmod_out", "Wout"):
            getattr(P, k).split_grad(getattr(g, k), out)
        out[id(self.pre_c_norm.weight)], out[id(self.pre_c_norm2.weight)] = g.wn1, g.wn2
        out[id(self.learnable_scalar)], out[id(self.learnable_scalar2)] = g.s1, g.s2
        out[id(self.patch_emb.bias)], out[id(self.out_proj.bias)] = g.bpe, g.bout
        out[id(self.time_scale)] = g.time_scale
        for b, gb in zip(self.blocks, g.blocks):
            b.scatter_grads(gb, out)

    # ------------------------------------------------------------------------------------------
    def noise_batch(self, X, t):
        """Rectified-flow interpolation x_t = (1-t) x0 + t eps (diff_model.py:229-241); stays PyTorch."""
        X = X.to(self.device)
        t = t.to(self.device)[:, None, None, None]
        epsilon = torch.randn_like(X, device=self.device)
        X_t = (1 - t) * X + t * epsilon
        return X_t, epsilon

    def load_text_encoders(self):
        raise RuntimeError("text encoders / FLUX VAE (helpers/VAE_T5_CLIP_inference.py) are out of this build's scope: "
                           "assign an object with .text_to_embedding(text) and .VAE to self.text_encoders")

    def forward(self, x_t, t, c, c_pooled, nullCls_pooled=None, nullCls_gemma=None, nullCls_bert=None):
        if self.precision in ("fp8", "mxfp8") and torch.is_grad_enabled():
            raise RuntimeError(f'precision "{self.precision}" is inference-only (no backward schedule): call under torch.no_grad()')
        x_t = x_t.to(self.device)
        t = t.to(self.device) if torch.is_tensor(t) else t
        c = c.to(self.device)
        c_pooled = c_pooled.to(self.device)
        nullCls_pooled = nullCls_pooled.to(self.device) if nullCls_pooled is not None else None
        nullCls_gemma = nullCls_gemma.to(self.device) if nullCls_gemma is not None else None
        nullCls_bert = nullCls_bert.to(self.device) if nullCls_bert is not None else None

        # in-place null masking of the caller's tensors, exactly as the reference (diff_model.py:278-287)
        # (written as a broadcast multiply by the 0/1 keep-mask: same values, but no boolean-index
        #  nonzero() and therefore no host synchronisation inside the training step)
        with torch.no_grad():
            if nullCls_pooled is not None:
                c_pooled.mul_((~nullCls_pooled.bool()).to(c_pooled.dtype)[:, None])
            if nullCls_gemma is not None:
                c[:, :77].mul_((~nullCls_gemma.bool()).to(c.dtype)[:, None, None])
            if nullCls_bert is not None:
                c[:, 77:].mul_((~nullCls_bert.bool()).to(c.dtype)[:, None, None])

        if type(t) == int or type(t) == float:
            t = torch.tensor(t).repeat(x_t.shape[0]).to(torch.long).to(self.device)
        elif type(t) == list and type(t[0]) == int:
            t = torch.tensor(t).to(torch.long).to(self.device)
        elif type(t) == torch.Tensor:
            if len(t.shape) == 0:
                t = t.repeat(x_t.shape[0]).to(torch.long)
        else:
            print(f"t values must either be a scalar, list of scalars, or a tensor of scalars, not type: {type(t)}")
            return
        if not x_t.is_cuda:
            raise RuntimeError("diff_model.forward: the MMDiT hot path runs on an MI355X only (no CPU fallback); "
                               "construct the model with device='gpu' or a cuda device")
        if x_t.shape[-1] % 2 or x_t.shape[-2] % 2:
            raise RuntimeError(f"shape '{[x_t.shape[0], x_t.shape[-2] // 2, x_t.shape[-1] // 2, self.inCh, 2, 2]}' is invalid: latent height and width must be even")
        if x_t.dtype not in (torch.float32, torch.bfloat16):
            x_t = x_t.float()
        if c.dtype not in (torch.float32, torch.bfloat16):
            c = c.float()
        if c_pooled.dtype not in (torch.float32, torch.bfloat16):
            c_pooled = c_pooled.float()
        self._keep_saved = torch.is_grad_enabled()      # inference: the backward-only side outputs of the forward schedule are skipped
        return _MMDiTFn.apply(self, x_t.contiguous(), t.float().contiguous(), c.contiguous(), c_pooled.contiguous(), *self._param_list())

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_imgs(self, batchSize, num_steps, text_input, cfg_scale=0.0, width=256, height=256, save_intermediate=False, use_tqdm=False,
                    sampler="euler", generator=None):
        """Rectified-flow sampler with classifier-free guidance; public contract of diff_model.py:367-480 (arguments, `generator`
        semantics -- initial and per-step noise are drawn on the CPU from it --, "euler" | "euler_stochastic" | "heun", return
        value, error for an unknown sampler).

        Device-resident loop: everything that does not depend on the step is built ONCE -- the (2B, ...) guidance batch buffer whose
        two halves hold the same latents, the conditional | null text operands (the null half is what the reference's in-place
        null-masking produces: zeros), their projection into the blocks' text stream (engine.text_fwd: RMSNorm + c_proj of
        2B x 154 x 2304 values, timestep-independent), the packed weights and RoPE tables -- and a step is one engine.model_fwd on the
        resident buffers plus three small torch ops; no host synchronisation inside the loop."""
        if sampler not in ("euler", "euler_stochastic", "heun"):
            raise ValueError("Invalid sampler specified. Choose 'euler', 'euler_stochastic', or 'heun'.")
        self.eval()
        dev, B = self.device, batchSize
        VAE = self.text_encoders.VAE
        lat = (B, VAE.config.latent_channels, width // 8, height // 8)
        if lat[2] % 2 or lat[3] % 2:
            raise RuntimeError(f"shape '{[2 * B, lat[2] // 2, lat[3] // 2, self.inCh, 2, 2]}' is invalid: latent height and width must be even")
        m = self._mode()
        W = self.weights(m)
        rope = self.blocks[0].attn.rotary_emb.tables(lat[2] // 2, lat[3] // 2, dev)

        # guidance batch: rows [0, B) conditional, rows [B, 2B) unconditional
        xx = torch.empty((2 * B,) + lat[1:], dtype=torch.float32, device=dev)
        x = xx[:B]
        x.copy_(torch.randn(lat, generator=generator))
        xx[B:].copy_(x)
        tt = torch.empty((2 * B,), dtype=torch.float32, device=dev)
        hidden, pooled = self.text_encoders.text_to_embedding(text_input)
        c = torch.zeros((2 * B,) + tuple(hidden.shape[1:]), dtype=hidden.dtype if hidden.dtype in (torch.float32, torch.bfloat16) else torch.float32, device=dev)
        cp = torch.zeros((2 * B, pooled.shape[-1]), dtype=c.dtype, device=dev)
        c[:B] = hidden.to(dev)
        cp[:B] = pooled.to(dev)
        C_text = engine.text_fwd(m, W, c)[0]

        def velocity(t_now):
            tt.fill_(t_now)
            v = engine.model_fwd(m, W, xx, tt, c, cp, rope, keep=False, C=C_text)[0]
            return (1 + cfg_scale) * v[:B] - cfg_scale * v[B:]

        def decode(z):
            return VAE.decode((z.to(VAE.dtype) - VAE.config.shift_factor) / VAE.config.scaling_factor).sample.clamp(-1, 1)

        def first_image():
            return decode(x)[0].float().cpu().detach()

        dt = 1 / num_steps
        times = torch.linspace(1, 0 + (1.0 / num_steps), num_steps).tolist()     # host scalars (fp32 values of the reference's linspace)
        if use_tqdm:
            from tqdm import tqdm
            times = tqdm(times, total=num_steps)
        imgs = []
        for t_now in times:
            v1 = velocity(t_now)
            if sampler == "euler":
                x.sub_(v1, alpha=dt)
            elif sampler == "euler_stochastic":
                sigma = t_now * (1 - t_now) / (1 - t_now + 0.008)
                noise = torch.randn(v1.shape, generator=generator).to(dev)
                x.sub_(v1, alpha=dt).add_(noise, alpha=sigma * np.sqrt(dt))
            else:   # heun: trapezoidal corrector on the Euler prediction, second evaluation at t - dt
                x_keep = x.clone()
                x.sub_(v1, alpha=dt)
                xx[B:].copy_(x)
                v2 = velocity(float(torch.tensor(t_now, dtype=torch.float32) - dt))     # (fp32 `t - dt` as the reference's tensor arithmetic rounds it)
                torch.sub(x_keep, v1 + v2, alpha=dt / 2, out=x)
            xx[B:].copy_(x)
            if save_intermediate:
                imgs.append(first_image())
        if save_intermediate:
            imgs.append(first_image())
        output = decode(x).float()
        return (output, imgs) if save_intermediate else output

    # ------------------------------------------------------------------------------------------
    def saveModel(self, saveDir, EMA_state_dict=None, optimizer=None, scheduler=None, grad_scalar=None, step=None, streamer=None):
        """Six-file checkpoint layout of the reference (diff_model.py:489-536).  streamer (helpers.checkpoint_stream.CheckpointStreamer):
        snapshot on the device and write the same files from a background thread instead of blocking the training loop."""
        names = {"model": "model", "ema": "model_ema", "optim": "optim", "sched": "scheduler", "scaler": "scaler", "defs": "model_params"}
        if step:
            names = {k: v + f"_{step}s" for k, v in names.items()}
            self.defaults["start_step"] = step
        self.defaults["wandb_id"] = self.wandb_id
        if not os.path.isdir(saveDir):
            os.makedirs(saveDir)
        if streamer is not None:
            files = [(self.state_dict(), saveDir + os.sep + names["model"] + ".pkl")]
            if EMA_state_dict:
                files.append((EMA_state_dict, saveDir + os.sep + names["ema"] + ".pkl"))
            if optimizer:
                files.append((optimizer.state_dict(), saveDir + os.sep + names["optim"] + ".pkl"))
            if scheduler:
                files.append((scheduler.state_dict(), saveDir + os.sep + names["sched"] + ".pkl"))
            if grad_scalar:
                files.append((grad_scalar.state_dict(), saveDir + os.sep + names["scaler"] + ".pkl"))
            streamer.save(files, [(self.defaults, saveDir + os.sep + names["defs"] + ".json")])
            return
        torch.save(self.state_dict(), saveDir + os.sep + names["model"] + ".pkl")
        if EMA_state_dict:
            torch.save(EMA_state_dict, saveDir + os.sep + names["ema"] + ".pkl")
        if optimizer:
            torch.save(optimizer.state_dict(), saveDir + os.sep + names["optim"] + ".pkl")
        if scheduler:
            torch.save(scheduler.state_dict(), saveDir + os.sep + names["sched"] + ".pkl")
        if grad_scalar:
            torch.save(grad_scalar.state_dict(), saveDir + os.sep + names["scaler"] + ".pkl")
        with open(saveDir + os.sep + names["defs"] + ".json", "w") as f:
            json.dump(self.defaults, f)

    def loadModel(self, loadDir, loadFile, loadDefFile=None, wandb_id=None):
        """Re-initialise from the saved constructor kwargs, then strict load (diff_model.py:553-578)."""
        if loadDefFile:
            device_, dev_, precision_ = self.device, self.dev, self.precision
            with open(loadDir + os.sep + loadDefFile, "r") as f:
                self.defaults = json.load(f)
            D = self.defaults
            if "MLP_type" not in D:
                D["MLP_type"] = "swiglu_old"
            if "text_loss" not in D:
                D["text_loss"] = False
            if self.update_max_res:
                D["max_res"] = self.max_res
            self.__init__(**D)
            self.to(device_)
            self.device, self.dev = device_, dev_
            self.set_precision(precision_)
            self.load_state_dict(torch.load(loadDir + os.sep + loadFile, map_location=self.device, weights_only=False), strict=True)
        else:
            self.load_state_dict(torch.load(loadDir + os.sep + loadFile, map_location=self.device, weights_only=False), strict=True)
        if wandb_id is not None:
            self.wandb_id = wandb_id
