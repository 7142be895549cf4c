"""Flow-matching trainer: mirror of the reference's src/model_trainer.py for the hot path.

Kept from the reference (file:line into src/model_trainer.py): constructor kwargs (117-147), AdamW
(lr, eps 1e-8, wd 0.01, betas (0.9, 0.999); 260), constant-with-warmup / cosine schedule (25-41, 263),
GradScaler state (266-269), CPU EMA copy updated every ema_update_freq steps (256, 537-541), the step
(378-503): t ~ sigmoid(N(0,1)); null masks with p = (pooled, gemma, bert); x_t = (1-t) x0 + t eps;
loss = mean((v - (eps - x0))^2) / accumulation_steps; scale -> backward -> unscale -> clip_grad_norm_(1.0)
-> step -> scheduler.step(step) -> scaler.update -> zero_grad; six-file checkpoints through
diff_model.saveModel (545-548).

MI355X-native differences: one process per GPU, every rank is a model rank (no loader GPUs: batches come
from `data_source`, synthetic by default); gradient averaging is sd3_amd.reducer.GradReducer (bucketed
RCCL all-reduce on a side HIP stream, fired block by block from the backward schedule, skipped on
non-final accumulation micro-steps) instead of DistributedDataParallel; no wandb (stdout / JSONL log).
The loss, the AdamW object (state, param groups, checkpoint format) and the scheduler stay in PyTorch-ROCm as
BASELINE.json's north_star prescribes; with hip_optimizer=True (default on a GPU) the arithmetic of unscale + clip + AdamW
runs as three HIP launches (optim.ClipAdamW), hip_optimizer=False runs torch's own kernels.
"""
import copy
import json
import math
import os
import time

import torch
import torch.distributed as dist
from torch import nn

from .helpers.multi_gpu_helpers import is_main_process
from .helpers.TimeSampler import TimeSampler
from .helpers.checkpoint_stream import CheckpointStreamer
from .helpers.wire_format import unpad_latents
from .optim import ClipAdamW
from .reducer import GradReducer, broadcast_parameters


def get_scheduler(optimizer, num_warmup_steps, num_training_steps, use_lr_scheduler):
    """HF get_cosine_schedule_with_warmup (num_cycles 0.5) / get_constant_schedule_with_warmup lambdas."""
    if use_lr_scheduler:
        def lr_lambda(step):
            if step < num_warmup_steps:
                return float(step) / float(max(1, num_warmup_steps))
            progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))
    else:
        def lr_lambda(step):
            if step < num_warmup_steps:
                return float(step) / float(max(1.0, num_warmup_steps))
            return 1.0
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)


def init_distributed():
    """env:// rendezvous, backend nccl (= RCCL on ROCm) with a gloo fallback (model_trainer.py:46-79)."""
    if dist.is_initialized():
        return
    if "RANK" not in os.environ:
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the flight recorder's per-group status (last enqueued / last completed collective) is what capture_graph() polls to know that
    # the watchdog has retired every eager collective (_wait_for_watchdog); it only exists while the recorder is on
    os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend, init_method="env://", world_size=int(os.environ["WORLD_SIZE"]), rank=int(os.environ["RANK"]),
                            **({"pg_options": rccl_options()} if backend == "nccl" and rccl_options() is not None else {}))


def rccl_options():
    """ProcessGroupNCCL options of the gradient process group: MMDIT_RCCL_MAX_CTAS = n limits a collective to n workgroups (ncclConfig maxCTAs).  A persistent GEMM
    workgroup cannot share a compute unit with anything, and the one-round launches of the backward (249 tiles on 256 CUs) stay one round only while a collective holds
    at most 7 of them: on xGMI links that n workgroups saturate, a small n costs no bandwidth and the GEMMs nothing.  None when the variable is not set."""
    n = os.environ.get("MMDIT_RCCL_MAX_CTAS", "")
    if not n:
        return None
    opts = dist.ProcessGroupNCCL.Options()
    opts.config.max_ctas = max(1, int(n))
    opts.config.min_ctas = 1
    return opts


def _watchdog_status():
    """{process group id: (last collective enqueued to the watchdog, last one it retired)} from the flight recorder, or None when
    the recorder is off / the build does not publish it (see model_trainer._wait_for_watchdog)."""
    try:
        from torch._C import _distributed_c10d as c10d
        doc = json.loads(c10d._dump_nccl_trace_json(includeCollectives=False, onlyActive=True))
        st = doc.get("pg_status") or {}
        out = {k: (int(v["last_enqueued_collective"]), int(v["last_completed_collective"])) for k, v in st.items()}
        return out or None
    except Exception:
        return None


LOSS_WINDOW = (1e-3, 1e3)     # a rectified-flow MSE on latents of O(1) variance starts near 2 and stays far inside this window


def loss_is_plausible(loss, window=LOSS_WINDOW):
    """The ONE test "did this step train" of the launch-mode votes (train(), bench.py): a finite loss inside `window`.  bench.py passes
    the tighter (1e-3, 10) its N(0, 1) synthetic latents justify."""
    v = float(loss)
    return math.isfinite(v) and window[0] < v < window[1]


def vote_fastest(local_ms, group=None, device=None):
    """The collective half of a warm-up A/B: every rank brings its own timings of the same candidates (local_ms[i], milliseconds), the ranks take
    the element-wise MAX (a step is as slow as its slowest rank), and every rank returns the same (index of the smallest maximum -- the first one
    on ties --, list of the maxima).  One all-reduce; without a process group the local timings decide."""
    t = torch.tensor([float(x) for x in local_ms], dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "nccl":
            t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    worst = [float(x) for x in t.cpu()]
    return min(range(len(worst)), key=lambda i: (worst[i], i)), worst


class SyntheticData:
    """Synthetic (x0, text, pooled) batches shaped like the loader-GPU wire format (model_trainer.py:353-355):
    bf16 latents (B,16,res/8,res/8), bf16 text (B,154,2304) with Gemma-like large variance on tokens 0..76 and
    the ModernBERT zero padding on features 1024.. of tokens 77..153, bf16 pooled (B,class_dim)."""

    def __init__(self, batch, inCh, class_dim, latent_hw, device, seed=1234, resample=True):
        self.g = torch.Generator(device=device).manual_seed(seed)
        self.shape = (batch, inCh, latent_hw[0], latent_hw[1])
        self.batch, self.class_dim, self.device, self.resample = batch, class_dim, device, resample
        self._cache = None

    def __call__(self):
        if self._cache is not None and not self.resample:
            return self._cache
        x0 = torch.randn(self.shape, generator=self.g, device=self.device).to(torch.bfloat16)
        c = torch.randn((self.batch, 154, 2304), generator=self.g, device=self.device)
        c[:, :77] *= 30.0
        c[:, 77:, 1024:] = 0
        cp = torch.randn((self.batch, self.class_dim), generator=self.g, device=self.device).to(torch.bfloat16)
        self._cache = (x0, c.to(torch.bfloat16), cp)
        return self._cache


class _FlowLoss(torch.autograd.Function):
    """loss = mean((v - (eps - x0))^2) / accumulation_steps (reference model_trainer.py:429-446) through ops.flow_loss: the forward also
    forms d loss / d v, the backward scales it by the incoming gradient (the GradScaler's loss scale)."""

    @staticmethod
    def forward(ctx, v, x0, eps, accumulation_steps):
        from . import ops
        loss, ctx.dv = ops.flow_loss(v, x0, eps, accumulation_steps, need_grad=ctx.needs_input_grad[0])
        return loss

    @staticmethod
    def backward(ctx, g):
        dv, ctx.dv = ctx.dv, None
        return dv.mul_(g), None, None, None


class model_trainer:
    def __init__(self, diff_model, batchSize, accumulation_steps, totalSteps, lr, ema_update_freq, ema_decay, warmup_steps, use_lr_scheduler,
                 device, saveDir, numSaveSteps, null_prob_pooled=0.1, null_prob_gemma=0.1, null_prob_bert=0.1, text_loss_weight=0.0,
                 load_ema_file=None, optimFile=None, schedulerFile=None, scalerFile=None, use_amp=True, wandb_name=None,
                 wandb_log_gradients=False, reset_wandb=False, reset_optim=False, log_steps=10, loader_to_model_gpu=None,
                 bucket_indices_path=None, data_parquet_folder=None, max_res=256,
                 data_source=None, device_rng=False, use_ema=True, fused_optimizer=True, log_file=None, force_reducer=False,
                 fused_unscale_clip=True, ema_on_gpu=True, hip_optimizer=True, async_checkpoint=True, inf_padded_latents=False, loss_scaling=True,
                 graph_after=None, hip_loss=True, reserved_cus=None, autotune=None):
        self.batchSize, self.accumulation_steps, self.totalSteps = batchSize, accumulation_steps, totalSteps
        self.ema_update_freq, self.ema_decay = ema_update_freq, ema_decay
        self.saveDir, self.numSaveSteps, self.log_steps = saveDir, numSaveSteps, log_steps
        self.null_prob_pooled, self.null_prob_gemma, self.null_prob_bert = null_prob_pooled, null_prob_gemma, null_prob_bert
        self.use_amp, self.max_res, self.log_file = use_amp, max_res, log_file
        self.device_rng = device_rng
        # hip_loss: the loss expression as two HIP launches (ops.flow_loss) instead of ~6 torch kernels; False: torch's own ops
        self.hip_loss = bool(hip_loss)
        # graph_after=N: train() captures the optimizer step into a hipGraph after N eager steps and replays it from then on
        # (None: every step is issued from the host, as the reference does)
        self.graph_after = graph_after
        self.keep_graph = False      # True: capture_graph() keeps the hipGraph_t next to the executable graph (graph_node_types())
        if text_loss_weight != 0.0:
            raise RuntimeError("text_loss is out of scope on the HIP path (text_loss_weight must be 0)")
        if loader_to_model_gpu not in (None, {}):
            raise RuntimeError("loader GPUs are not used: every rank is a model rank; pass a data_source instead")

        init_distributed()
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.subgroup = None
        self.model = diff_model
        self.device = self.model.device
        self.dev = self.model.dev
        broadcast_parameters(self.model)
        self.reducer = GradReducer(self.subgroup, force=force_reducer)
        self.reserved_cus, self.autotune_table = 0, None
        self.autotune = (os.environ.get("MMDIT_REDUCE_AUTOTUNE", "1") != "0") if autotune is None else bool(autotune)      # train(): warm-up A/B of the reducer settings (world > 1)
        if self.reducer.enabled:
            # Compute units the backward's GEMM planner leaves to the collectives' kernels (constructor `reserved_cus`, else MMDIT_RESERVED_CUS, default 32
            # when gradients are reduced, 0 = plan for the whole chip).  A persistent GEMM workgroup needs a whole CU, so while RCCL's channels hold C of them
            # C workgroups of a launch start late.  Round 6: the multi-round launches CLAIM their tiles (csrc/gemm8p.hip), so late workgroups cost C / 256 of
            # the rate instead of a second round; what the reserve changes is the DECOMPOSITION of the backward's launches (diff_model._MMDiTFn.backward sets
            # the planner's budget to CUs - reserve around engine.model_bwd): the weight gradients' split tail and the one-round data gradients are cut so that
            # any number of workgroups between the budget and the whole chip stays busy.  The grids still cover every CU.  How many CUs RCCL's all-reduce
            # takes on an 8-GPU xGMI node could not be measured here (tools/probes/cu_contention.py measures a stand-in); frozen with a captured step.
            if self.device.type == "cuda" and os.environ.get("MMDIT_GEMM_CLAIMING", "1") != "0":
                from . import _lib
                with torch.cuda.device(self.device):
                    _lib.check(_lib.lib().mmdit_gemm_set_claiming(1), "mmdit_gemm_set_claiming")      # (per device; off by default: it costs ~3 us per launch and pays beside the collectives)
            if reserved_cus is None:
                reserved_cus = int(os.environ.get("MMDIT_RESERVED_CUS", "32"))
            self.reserved_cus = max(0, int(reserved_cus)) // 8 * 8
            if self.reserved_cus > 0 and self.device.type == "cuda":
                from . import _lib
                cus = _lib.lib().mmdit_get_cu_budget()
                self.model.bwd_cu_budget = max(64, cus - self.reserved_cus)
            if hasattr(self.model, "grad_reducer"):
                self.model.grad_reducer = self.reducer           # overlapped: fired from the backward schedule
            else:
                self.reducer.attach_hooks(self.model.parameters())

        self.ema_model_cpu = None
        if use_ema:
            red, self.model.grad_reducer = getattr(self.model, "grad_reducer", None), None
            self.ema_model_cpu = copy.deepcopy(self.model).cpu()
            self.ema_model_cpu.eval()
            if hasattr(self.model, "grad_reducer"):
                self.model.grad_reducer = red

        self.fused_unscale_clip = bool(fused_unscale_clip)
        fused = bool(fused_optimizer and self.device.type == "cuda")
        # hip_optimizer: unscale + clip + AdamW as three HIP launches (optim.ClipAdamW) -- the same torch.optim.AdamW
        # object otherwise (state, param_groups, checkpoints, scheduler); False keeps torch's own multi-tensor kernels.
        self.hip_optimizer = bool(hip_optimizer and self.device.type == "cuda")
        opt_cls = ClipAdamW if self.hip_optimizer else torch.optim.AdamW
        self.optim = opt_cls(self.model.parameters(), lr=lr, eps=1e-8, weight_decay=0.01, betas=(0.9, 0.999), fused=fused)
        self.scheduler = get_scheduler(self.optim, num_warmup_steps=warmup_steps, num_training_steps=totalSteps, use_lr_scheduler=use_lr_scheduler)
        # loss_scaling=False: the GradScaler-free bf16 step -- bf16 has fp32's exponent range, the reference's dynamic
        # loss scale is not needed; the scaler object stays (disabled) so that the six-file checkpoint layout is unchanged
        self.grad_scaler = torch.amp.GradScaler("cuda", enabled=self.device.type == "cuda" and bool(loss_scaling)) if self.use_amp else None

        if load_ema_file and self.ema_model_cpu is not None:
            self.ema_model_cpu.load_state_dict(torch.load(load_ema_file, map_location="cpu", weights_only=False))
        # GPU-resident EMA: the reference updates the CPU copy with a blocking per-parameter `.cpu()` loop
        # (model_trainer.py:537-541), a visible stall once a step takes ~40 ms.  The running average lives in HBM (1.26 GB
        # of 288 GB for MMDiT-B) and is updated by two fused multi-tensor passes; `ema_model_cpu` (the reference's
        # attribute, what the checkpoint stores) is refreshed by sync_ema_to_cpu() whenever it is saved.
        self._ema_gpu = None
        if self.ema_model_cpu is not None and ema_on_gpu and self.device.type == "cuda":
            self._ema_gpu = [q.detach().to(self.device, copy=True) for q, p in zip(self.ema_model_cpu.parameters(), self.model.parameters())
                             if p.requires_grad]
        # checkpoint streaming: device-side snapshot + background D2H and file writes instead of a blocking saveModel
        self.ckpt_stream = CheckpointStreamer(self.device) if (async_checkpoint and self.device.type == "cuda") else None
        if optimFile and not reset_optim:
            self.optim.load_state_dict(torch.load(optimFile, map_location=self.device, weights_only=False))
        if schedulerFile:
            self.scheduler.load_state_dict(torch.load(schedulerFile, map_location=self.device, weights_only=False))
        if scalerFile and self.grad_scaler is not None:
            self.grad_scaler.load_state_dict(torch.load(scalerFile, map_location=self.device, weights_only=False))

        self.wandb_id = None if reset_wandb else self.model.wandb_id
        self.start_step = self.model.start_step * self.accumulation_steps
        self.time_sampler = TimeSampler(weighted=True)
        self.data_source = data_source or SyntheticData(batchSize, self.model.inCh, self.model.class_dim, (max_res // 8, max_res // 8), self.device,
                                                        seed=1234 + self.rank)
        self._gen = torch.Generator(device=self.device).manual_seed(4321 + self.rank) if device_rng else None
        self.inf_padded_latents = bool(inf_padded_latents)
        self.last_loss = None
        self._graph, self._slots, self._loss_out, self._graph_loss = None, None, None, None
        self._slots_primed = False   # the batches drawn at capture time are the inputs of the first replay (nothing drawn is dropped)
        self.replayed_steps = 0      # optimizer steps served by the hipGraph so far
        self._carry_inputs = None    # micro-batches drawn at a capture whose graph was dropped before a replay trained on them
        self.capture_handoff = None  # how capture_graph() knew RCCL's watchdog was idle: "polled" / "heuristic" (None: no collectives)
        self.keep_losses, self.loss_history = False, []     # keep_losses: train() appends every step's loss (device scalars)
        self.last_grad_norm = None   # hip_optimizer: device scalar, the unscaled gradient norm of the last step
        if is_main_process():
            total_params = sum(p.numel() for p in self.model.parameters()) / 1e6
            print(f"Number of parameters in the model: {total_params:.2f}M")

    # ------------------------------------------------------------------------------------------
    def _sample_conditioning(self, n):
        """t and the three null masks (model_trainer.py:378-387).  The reference draws them from the CPU
        default generator and moves them; device_rng=True draws on the GPU (no host round trip)."""
        if self.device_rng:
            t = self.time_sampler(n, generator=self._gen, device=self.device)
            r = torch.rand((3, n), generator=self._gen, device=self.device)
            return t, r[0] < self.null_prob_pooled, r[1] < self.null_prob_gemma, r[2] < self.null_prob_bert
        t = self.time_sampler(n)
        pp, pg, pb = torch.rand(n), torch.rand(n), torch.rand(n)
        mk = lambda p, thr: torch.where(p < thr, 1, 0).to(torch.bool).to(self.device)
        return t, mk(pp, self.null_prob_pooled), mk(pg, self.null_prob_gemma), mk(pb, self.null_prob_bert)

    def _draw(self):
        """One micro-batch and its conditioning draw: (x0, text, pooled, t, null_pooled, null_gemma, null_bert)."""
        with torch.no_grad():
            batch_x_0, batch_txt, batch_txt_pooled = self.data_source()
            if self.inf_padded_latents:      # latents arrive in the reference's +inf-padded wire format (model_trainer.py:362-370)
                batch_x_0 = unpad_latents(batch_x_0)
            return (batch_x_0, batch_txt, batch_txt_pooled) + tuple(self._sample_conditioning(batch_x_0.shape[0]))

    def micro_step(self, final: bool, inputs=None):
        """One forward/backward micro-step; returns the (already /accumulation_steps) loss tensor.  inputs: a _draw() result (the graph
        path passes its static input slots), None: draw here."""
        batch_x_0, batch_txt, batch_txt_pooled, t_vals, n_pooled, n_gemma, n_bert = inputs if inputs is not None else self._draw()
        with torch.no_grad():
            batch_x_t, epsilon_t = self.model.noise_batch(batch_x_0, t_vals)
        # Gradient accumulation = DDP.no_sync (reference model_trainer.py:463-480): nothing is reduced on non-final micro-steps.
        #  * hook path (post-accumulate hooks see the ACCUMULATED gradient): the hooks reduce on the final micro-step only;
        #  * engine path, accumulation_steps == 1: the backward schedule hands each block's gradients to the reducer as soon as
        #    they are final (overlapped with the rest of the backward);
        #  * engine path, accumulation_steps > 1: the schedule only ever sees the current micro-step's own gradients, so it is
        #    told to skip, and after the final backward the ACCUMULATED .grad tensors are averaged in flat buckets (one
        #    reduction per optimizer step instead of one per micro-step).
        engine_path = getattr(self.model, "grad_reducer", None) is self.reducer and self.reducer.enabled
        late = engine_path and self.accumulation_steps > 1
        self.reducer.skip = (not final) or late
        v_pred = self.model(batch_x_t.detach(), t_vals, batch_txt, batch_txt_pooled, n_pooled, n_gemma, n_bert)
        x0 = batch_x_0.to(epsilon_t.device)
        if self.hip_loss and v_pred.is_cuda and v_pred.dtype == torch.float32 and x0.dtype == epsilon_t.dtype and v_pred.numel() % 8 == 0:
            # same arithmetic as the torch expression below in two HIP launches with a bit-reproducible reduction (ops.flow_loss)
            loss = _FlowLoss.apply(v_pred, x0.contiguous(), epsilon_t.contiguous(), self.accumulation_steps)     # (unpadded bucket latents are strided views)
        else:
            labels = epsilon_t - x0
            loss = nn.MSELoss(reduction="none")(v_pred, labels.detach().to(v_pred.dtype)).flatten(1, -1).mean()
            loss = loss / self.accumulation_steps
        if self.grad_scaler is not None:
            self.grad_scaler.scale(loss).backward()
        else:
            loss.backward()
        if final:
            self.reducer.skip = False
            if late:
                self.reducer.add([p.grad for p in self.model.parameters() if p.grad is not None])
            self.reducer.finish()
        return loss.detach()

    def _unscale_and_clip(self, max_norm=1.0):
        """GradScaler.unscale_(optim) + clip_grad_norm_(params, max_norm) with ONE pass over the gradients instead of two
        (model_trainer.py:463-470 of the reference calls them back to back).  The loss scale is a power of two, so
        g * (inv_scale * clip_coef) is bit-identical to (g * inv_scale) * clip_coef, and ||g * inv_scale|| = inv_scale * ||g||;
        an inf/nan gradient makes the norm non-finite, which is exactly unscale_'s found_inf.  The scaler is told that the
        gradients are unscaled (the same bookkeeping unscale_ does), so step()/update() behave as in the reference."""
        from torch.amp.grad_scaler import OptState
        sc = self.grad_scaler
        st = sc._per_optimizer_states[id(self.optim)]
        if st["stage"] is not OptState.READY:
            raise RuntimeError("unscale_() has already been called on this optimizer since the last update().")
        grads = [p.grad for p in self.model.parameters() if p.grad is not None]
        total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads, 2.0)), 2.0)     # norm of the SCALED gradients
        inv_scale = sc._scale.double().reciprocal().float()
        coef = inv_scale * torch.clamp(max_norm / (total * inv_scale + 1e-6), max=1.0)
        torch._foreach_mul_(grads, coef)
        st["found_inf_per_device"] = {total.device: (~torch.isfinite(total)).to(torch.float32)}
        st["stage"] = OptState.UNSCALED

    def _hip_optimizer_step(self):
        """unscale_ + clip_grad_norm_(1.0, only under AMP as in the reference) + AdamW in ClipAdamW.step_clipped, with the
        GradScaler bookkeeping that unscale_() / step() would have done so that update() adjusts the loss scale as usual."""
        sc = self.grad_scaler
        scaled = sc is not None and sc.is_enabled()
        found_inf, self.last_grad_norm = self.optim.step_clipped(sc._scale if scaled else None, 1.0 if self.use_amp else None)
        if scaled:
            from torch.amp.grad_scaler import OptState
            st = sc._per_optimizer_states[id(self.optim)]
            if st["stage"] is not OptState.READY:
                raise RuntimeError("the optimizer was already unscaled / stepped since the last GradScaler.update()")
            st["found_inf_per_device"] = {found_inf.device: found_inf}
            st["stage"] = OptState.STEPPED

    def optimizer_step(self, step):
        if self.hip_optimizer:
            self._hip_optimizer_step()
            self.scheduler.step(step)
            if self.grad_scaler is not None:
                self.grad_scaler.update()
            self.optim.zero_grad()
            return
        if self.grad_scaler is not None and self.use_amp and self.grad_scaler.is_enabled() and self.fused_unscale_clip:
            self._unscale_and_clip(1.0)
        else:
            if self.grad_scaler is not None:
                self.grad_scaler.unscale_(self.optim)
            if self.use_amp:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1.0)
        if self.grad_scaler is not None:
            self.grad_scaler.step(self.optim)
        else:
            self.optim.step()
        self.scheduler.step(step)
        if self.grad_scaler is not None:
            self.grad_scaler.update()
        self.optim.zero_grad()

    def train_step(self, step):
        """One optimizer step = accumulation_steps micro-steps + clip + AdamW."""
        if self._graph is not None:
            return self._replay(step)
        return self._eager_step(step)

    def _eager_step(self, step, inputs=None):
        if inputs is None and self._carry_inputs is not None:
            inputs, self._carry_inputs = self._carry_inputs, None      # drawn for a capture that was dropped before any replay used them
        loss = None
        for k in range(self.accumulation_steps):
            l = self.micro_step(final=(k == self.accumulation_steps - 1), inputs=inputs[k] if inputs is not None else None)
            loss = l if loss is None else loss + l
        self.optimizer_step(step)
        self.last_loss = loss
        return loss

    # ------------------------------------------------------------------------------------------
    # hipGraph capture of the whole optimizer step.  The reference has nothing comparable; here a step is ~440 kernel launches
    # issued through ctypes (~26 ms of host time per 30 ms step), and every launch already takes its stream explicitly, allocates
    # nothing and never synchronises, so the step can be captured once and replayed with ONE host call.
    def can_capture(self):
        """None when capture_graph() can run, else the reason it cannot."""
        if self.device.type != "cuda" or not self.hip_optimizer:
            return "needs hip_optimizer=True on a GPU"
        if self.ema_model_cpu is not None and self._ema_gpu is None:
            return "the EMA must live on the GPU (ema_on_gpu=True) or be off"
        if self.reducer.enabled and dist.get_backend(self.subgroup) != "nccl":
            return "the gradient collectives are captured with the step: backend nccl (RCCL) only"
        return None

    def capture_graph(self, step):
        """Capture forward + backward + gradient all-reduce + (unscale, clip, AdamW) + GradScaler.update of one optimizer step into a
        hipGraph; later train_step() calls replay it.  Call after a few eager steps (the bf16 weight copies, the zero pool, the
        optimizer state and the scaler must be in their steady state).
        Inputs: with the trainer's own SyntheticData source and device_rng=True the batch and its conditioning draw are generated
        INSIDE the graph (RNG streams registered with it).  Any other data source / the reference's CPU draw of t and the null masks
        (device_rng=False) stay outside: every replay first copies the next micro-batches into static input slots the graph reads;
        a batch whose shapes differ from the captured ones (aspect-ratio buckets) runs as an eager step.
        Data parallel (reducer enabled, RCCL): the per-block all-reduces and their side stream are part of the captured step -- the
        fork / join of the side stream becomes graph edges, every rank replays the same sequence of collectives."""
        why = self.can_capture()
        if why is not None:
            raise RuntimeError("capture_graph: " + why)
        from . import engine
        if engine._WG_OVERLAP:
            engine._WG_OVERLAP = False     # (the weight-gradient side stream measured no gain: DESIGN.md 5)
        in_graph = isinstance(self.data_source, SyntheticData) and self.device_rng
        self._slots = None
        if not in_graph:
            self._slots = [tuple(x.to(self.device).clone() for x in self._draw()) for _ in range(self.accumulation_steps)]
        torch.cuda.synchronize(self.device)
        if self.reducer.enabled and self.device.type == "cuda":
            # "polled" = the watchdog's own counters said it had retired every eager collective; "heuristic" = fixed delay (the process
            # group was created without TORCH_FR_BUFFER_SIZE, i.e. by a caller that did not go through init_distributed())
            self.capture_handoff = "polled" if self._wait_for_watchdog() else "heuristic"
        self.optim.sync_lr(self.device)
        self.optim.prepare_capture()
        # the loss leaves the graph through a persistent buffer written by a kernel of the graph (not through a tensor of the graph's
        # private pool, whose block is shared with earlier temporaries of the step): DESIGN.md 5, "final_loss 0.0"
        self._loss_out = torch.zeros((), dtype=torch.float32, device=self.device)
        g = torch.cuda.CUDAGraph(keep_graph=True) if self.keep_graph else torch.cuda.CUDAGraph()
        for gen in (self._gen, getattr(self.data_source, "g", None)) if in_graph else ():
            if gen is not None:
                g.register_generator_state(gen)
        self.optim.zero_grad()
        try:
            # thread_local: the RCCL watchdog thread polls events of earlier collectives while this thread captures
            with torch.cuda.graph(g, capture_error_mode="thread_local" if self.reducer.enabled else "global"):
                loss = None
                for k in range(self.accumulation_steps):
                    l = self.micro_step(final=(k == self.accumulation_steps - 1), inputs=None if in_graph else self._slots[k])
                    loss = l if loss is None else loss + l
                self._hip_optimizer_step()
                if self.grad_scaler is not None:
                    self.grad_scaler.update()
                self._write_loss(loss)
                self._graph_loss = loss      # (the capture-time tensor in the graph's private pool: diagnostics only, see tools/probes/graph_loss_probe.py)
        except BaseException:
            self._abandon_capture()
            raise
        self.optim.zero_grad()      # (host bookkeeping only: the graph owns the gradient buffers)
        self.optim.after_replay()   # (the capture itself executed nothing: the eager pointer table / copies bookkeeping is void)
        self._graph = g
        self._slots_primed = self._slots is not None     # the capture executed nothing: the slots still hold an untrained-on draw
        return g

    def capture_graph_agreed(self, step, strict=False):
        """capture_graph() as a COLLECTIVE decision: every rank tries, then the ranks agree (all-reduce MIN over "my capture
        succeeded") before anything else is enqueued; if any rank failed, every rank drops its graph and goes on issuing eager steps --
        a rank that replays while another launches from the host would still run the same collectives per step, but mixed modes are
        not a state anybody has tested, and a rank that RAISES while the others wait in a collective would hang the job.  Returns True
        when every rank replays from now on.  strict=True re-raises this rank's capture error after the agreement (so that the
        other ranks are not left waiting).  Each rank logs its own outcome and its reducer to stderr."""
        import sys
        err = None
        try:
            why = self.can_capture()
            if why is not None:
                raise RuntimeError("capture_graph: " + why)
            self.capture_graph(step)
        except Exception as e:       # capture_graph has put the host state back (_abandon_capture): eager steps can continue
            err = e
            self._graph = None
        ok_local = self._graph is not None
        ok_all = ok_local
        if dist.is_initialized() and self.world > 1:
            flag = torch.tensor([1 if ok_local else 0], dtype=torch.int32,
                                device=self.device if dist.get_backend(self.subgroup) == "nccl" else torch.device("cpu"))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.subgroup)
            ok_all = bool(int(flag))
        if not ok_all and ok_local:
            self._drop_graph()
        print(f"[model_trainer rank {self.rank}/{self.world}] step capture: {'ok' if ok_local else 'FAILED (' + type(err).__name__ + ': ' + str(err) + ')'}; "
              f"launch mode for all ranks: {'hipGraph replay' if ok_all else 'eager'}; {self.reducer.describe()}", file=sys.stderr, flush=True)
        if err is not None and strict:
            raise err
        return ok_all

    def autotune_reducer(self, step, steps_each=2, candidates=None, run_step=None):
        """Warm-up A/B of the data-parallel settings nobody could measure in advance (no multi-GPU node was available to the builder): the bucket
        algorithm of the reducer (allreduce / rs_ag / direct; + direct with a bf16 wire only when MMDIT_REDUCE_TUNE_WIRE=1 -- it rounds the averaged
        gradients) and the compute units the weight-gradient planner leaves to the collectives (reserved_cus in {0, 16, 32}).  Every candidate runs
        one untimed and two blocks of `steps_each` EAGER optimizer steps (the faster block counts) -- real training steps on fresh batches, nothing is thrown away -- timed on the host between device
        synchronisations; the ranks vote (vote_fastest: MAX over ranks per candidate, then the minimum) so that all of them continue with the same
        setting.  Three rounds: algorithms at the current reserve, reserves at the winning algorithm, then how many blocks' buckets are handed to the side stream per fork
        (reducer.blocks_per_fork in {1, 2, 3}).  Call before capture_graph (the choice is frozen
        into the captured step).  Returns (next step, table) with table = {"algorithm": {...ms}, "reserved_cus": {...ms}, "chosen": {...}}; a no-op
        (step, None) when gradients are not reduced."""
        import sys
        if not self.reducer.enabled:
            return step, None
        if self._graph is not None:
            raise RuntimeError("autotune_reducer: call it before the step is captured")
        step_fn = run_step or self.train_step
        self.autotune_losses = []      # the losses of the steps run here, in order (train() files them with the others)

        def run(n):
            loss = step_fn(n)
            self.autotune_losses.append(loss)
            return loss

        cuda = self.device.type == "cuda"

        def time_candidate(apply):
            nonlocal step
            apply()
            step += 1
            run(step)                      # (first step with a new setting: buffers of that algorithm are allocated here)
            best = float("inf")
            for _ in range(2):             # two timed blocks, the faster one counts: one host hiccup in a two-step block is a 15 % outlier
                if cuda:
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps_each):
                    step += 1
                    run(step)
                if cuda:
                    torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / steps_each * 1e3)
            return best

        def set_algo(a, w):
            def f():
                self.reducer.algorithm, self.reducer.wire_dtype = a, w
            return f

        def set_reserve(r):
            def f():
                self.reserved_cus = r
                if cuda and hasattr(self.model, "bwd_cu_budget"):
                    from . import _lib
                    self.model.bwd_cu_budget = max(64, _lib.lib().mmdit_get_cu_budget() - r) if r > 0 else None
            return f

        algos = candidates or [("allreduce", None), ("rs_ag", None), ("direct", None)]
        if candidates is None and os.environ.get("MMDIT_REDUCE_TUNE_WIRE", "") == "1":
            algos.append(("direct", torch.bfloat16))
        name = lambda a, w: a + ("+bf16" if w == torch.bfloat16 else "")
        ms = [time_candidate(set_algo(a, w)) for a, w in algos]
        best, worst = vote_fastest(ms, self.subgroup, self.device)
        set_algo(*algos[best])()
        table = {"algorithm": {name(a, w): round(t, 3) for (a, w), t in zip(algos, worst)}}
        reserves = sorted({0, 16, 32, int(self.reserved_cus)})
        ms = [time_candidate(set_reserve(r)) for r in reserves]
        rbest, rworst = vote_fastest(ms, self.subgroup, self.device)
        set_reserve(reserves[rbest])()
        table["reserved_cus"] = {str(r): round(t, 3) for r, t in zip(reserves, rworst)}

        def set_fork(k):
            def f():
                self.reducer.blocks_per_fork = k
            return f

        forks = sorted({1, 2, 3, int(self.reducer.blocks_per_fork)})
        ms = [time_candidate(set_fork(k)) for k in forks]
        fbest, fworst = vote_fastest(ms, self.subgroup, self.device)
        set_fork(forks[fbest])()
        table["blocks_per_fork"] = {str(k): round(t, 3) for k, t in zip(forks, fworst)}
        table["chosen"] = {"algorithm": name(*algos[best]), "reserved_cus": reserves[rbest], "blocks_per_fork": forks[fbest], "steps_each": steps_each}
        # Fourth round (RCCL only): the workgroups a collective may use (ncclConfig maxCTAs).  The one-round GEMM launches of the backward are 249 tiles on 256 compute
        # units: they stay ONE round while the collectives hold at most 7 CUs, so a communicator limited to 7 workgroups costs the GEMMs nothing -- if 7 workgroups
        # still saturate the xGMI links, which only the real node can tell.  A second communicator (dist.new_group with the limit) is created here, on every rank in the
        # same order, and the reducer keeps whichever the vote prefers.  MMDIT_REDUCE_TUNE_CTAS=0 skips the round; a failure to create the group skips it on the rank
        # that saw it (the next collective of the default group would then show a disagreement as a time-out rather than as silence).
        nccl = dist.is_initialized() and dist.get_backend(self.subgroup) == "nccl"
        if nccl and os.environ.get("MMDIT_REDUCE_TUNE_CTAS", "1") != "0" and (self.world > 1 or os.environ.get("MMDIT_REDUCE_TUNE_CTAS") == "force"):
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.config.max_ctas, opts.config.min_ctas = 7, 1
                limited = dist.new_group(ranks=None, pg_options=opts)
                groups = [("default", self.reducer.group), ("7", limited)]

                def set_group(g):
                    def f():
                        self.reducer.group = g
                    return f

                ms = [time_candidate(set_group(g)) for _, g in groups]
                gbest, gworst = vote_fastest(ms, self.subgroup, self.device)
                set_group(groups[gbest][1])()
                table["max_ctas"] = {n: round(t, 3) for (n, _), t in zip(groups, gworst)}
                table["chosen"]["max_ctas"] = groups[gbest][0]
            except Exception as e:      # (an RCCL build without communicator configs: keep the default group)
                print(f"[model_trainer rank {self.rank}/{self.world}] reducer auto-tune: no CTA-limited communicator ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
        print(f"[model_trainer rank {self.rank}/{self.world}] reducer auto-tune (ms per eager step, max over ranks): {json.dumps(table)}", file=sys.stderr, flush=True)
        self.autotune_table = table
        return step, table

    def _drop_graph(self):
        """Back to eager launches.  Batches drawn into the static slots that no replay has trained on yet (a capture that the ranks
        then voted down) are not thrown away: the next eager step trains on them."""
        if self._slots is not None and self._slots_primed:
            self._carry_inputs = [tuple(x.clone() for x in slot) for slot in self._slots]
        self._graph, self._slots, self._slots_primed = None, None, False

    def keep_graph_if_agreed(self, ok_local, what="the first replayed steps"):
        """Second half of the collective launch-mode decision: after the first replays every rank reports whether ITS replayed steps
        trained (finite, plausible losses; parameters moved).  MIN over the ranks; a failure anywhere -> every rank drops its graph and
        goes on with eager launches (the same collectives per step, issued from the host).  A captured step that replays wrongly on some
        rank (never observed; graph capture with RCCL collectives has only run on one rank so far) then costs the replay's ~1 % instead of
        the run.  Returns True when the ranks keep replaying."""
        import sys
        if self._graph is None:
            return False
        ok_all = bool(ok_local)
        if dist.is_initialized() and self.world > 1:
            flag = torch.tensor([1 if ok_local else 0], dtype=torch.int32,
                                device=self.device if dist.get_backend(self.subgroup) == "nccl" else torch.device("cpu"))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.subgroup)
            ok_all = bool(int(flag))
        if not ok_all:
            self._drop_graph()
        print(f"[model_trainer rank {self.rank}/{self.world}] {what}: {'ok' if ok_local else 'NOT ok on this rank'}; "
              f"launch mode for all ranks from here on: {'hipGraph replay' if ok_all else 'eager'}", file=sys.stderr, flush=True)
        return ok_all

    def _wait_for_watchdog(self, timeout_s=None):
        """Block until RCCL's watchdog thread has retired every EAGER collective of this process.

        Why: the watchdog polls the completion events of enqueued collectives (hipEventQuery) every 100 ms.  Those events were
        recorded on the communicator's internal stream, which joins the capture at the first captured collective; HIP then refuses
        the query ("event last recorded in a capturing stream"), the watchdog thread throws and the process aborts.  Collectives
        issued DURING a capture are never handed to the watchdog, so the hazard is only the eager ones still on its list.
        How: after the device synchronize every eager collective is complete, and the watchdog retires a completed one at its next
        sweep; the flight recorder publishes, per process group, the sequence numbers of the last collective ENQUEUED to the watchdog
        and the last one it COMPLETED (retired).  Poll until they are equal for every group (typically one or two sweeps).  If the
        recorder is off (TORCH_FR_BUFFER_SIZE=0 set by the user) or the build does not expose the status, fall back to the
        HEURISTIC of round 3 -- sleep MMDIT_CAPTURE_SETTLE_S seconds (default 0.5 = five sweeps) -- and say so once on stderr."""
        import sys
        import time
        timeout_s = float(os.environ.get("MMDIT_CAPTURE_WATCHDOG_TIMEOUT_S", "10")) if timeout_s is None else timeout_s
        status = _watchdog_status()
        if status is None:
            if not getattr(model_trainer, "_warned_settle", False):
                model_trainer._warned_settle = True
                print("[model_trainer] flight-recorder status unavailable: waiting a fixed MMDIT_CAPTURE_SETTLE_S before the capture (heuristic)", file=sys.stderr)
            time.sleep(float(os.environ.get("MMDIT_CAPTURE_SETTLE_S", "0.5")))
            return False
        t0 = time.time()
        while any(enq != done for enq, done in status.values()):
            if time.time() - t0 > timeout_s:
                # (ten seconds are a hundred sweeps: the counters do not mean what this code assumes on this build -- say so and go on,
                #  the wait so far is already 20 x the fixed delay that was enough in round 3)
                print(f"[model_trainer] watchdog status did not settle in {timeout_s} s ({status}); capturing anyway", file=sys.stderr)
                return False
            time.sleep(0.02)
            status = _watchdog_status()
        return True

    def graph_node_types(self):
        """Histogram of the captured step's node kinds {"kernel": n, "memcpy": n, "memset": n, ...} (needs keep_graph=True before
        capture_graph()).  The step must not contain memset nodes: a hipMemsetAsync captured into a hipGraph does not reliably keep
        its place in the stream order when the graph is replayed on ROCm 7 (tools/probes/graph_memset_order.py)."""
        import ctypes
        if self._graph is None or not self.keep_graph:
            raise RuntimeError("graph_node_types: set keep_graph = True before capture_graph()")
        hip = ctypes.CDLL("libamdhip64.so")
        graph = ctypes.c_void_p(self._graph.raw_cuda_graph())
        n = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(graph, None, ctypes.byref(n)) != 0:
            raise RuntimeError("hipGraphGetNodes failed")
        nodes = (ctypes.c_void_p * n.value)()
        if hip.hipGraphGetNodes(graph, nodes, ctypes.byref(n)) != 0:
            raise RuntimeError("hipGraphGetNodes failed")
        names = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}
        hist = {}
        for node in nodes:
            t = ctypes.c_int(-1)
            if hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t)) != 0:
                raise RuntimeError("hipGraphNodeGetType failed")
            k = names.get(t.value, f"type{t.value}")
            hist[k] = hist.get(k, 0) + 1
        return hist

    def _write_loss(self, loss):
        torch.mul(loss.detach().to(torch.float32).reshape(()), 1.0, out=self._loss_out)

    def _abandon_capture(self):
        """A capture that raised leaves host state behind that no launch backs: put it back so that eager steps can continue."""
        if self._slots is not None:      # (drawn for the capture, trained on by nobody: the next eager step takes them)
            self._carry_inputs = [tuple(x.clone() for x in slot) for slot in self._slots]
        self._graph, self._slots, self._slots_primed = None, None, False
        self.reducer.reset()
        if self.grad_scaler is not None and self.grad_scaler.is_enabled():
            from torch.amp.grad_scaler import _refresh_per_optimizer_state
            self.grad_scaler._per_optimizer_states[id(self.optim)] = _refresh_per_optimizer_state()
        self.optim.after_replay()
        self.optim.zero_grad()
        torch.cuda.synchronize(self.device)

    def _replay(self, step):
        if self._slots is not None and self._slots_primed:
            self._slots_primed = False      # first replay: the slots already hold the batches drawn at capture time
        elif self._slots is not None:
            inputs = [self._draw() for _ in range(self.accumulation_steps)]
            if any(tuple(a.shape) != tuple(b.shape) or a.dtype != b.dtype for slot, inp in zip(self._slots, inputs) for a, b in zip(slot, inp)):
                return self._eager_step(step, inputs)     # another bucket shape: not the captured step
            for slot, inp in zip(self._slots, inputs):
                for dst, src in zip(slot, inp):
                    dst.copy_(src, non_blocking=True)
        self.optim.sync_lr(self.device)     # the scheduler's current rate -> device (a fill only when it changed)
        self._graph.replay()
        self.optim.after_replay()
        self.scheduler.step(step)
        self.replayed_steps += 1
        self.last_loss = self._loss_out.clone()      # (a fresh tensor per step, as the eager step returns: callers may keep it)
        return self.last_loss

    def update_ema(self):
        """ema = ema * decay + param * (1 - decay)  (model_trainer.py:537-541), on the GPU copy when there is one."""
        with torch.no_grad():
            if self._ema_gpu is not None:
                params = [p.detach() for p in self.model.parameters() if p.requires_grad]
                torch._foreach_mul_(self._ema_gpu, self.ema_decay)
                torch._foreach_add_(self._ema_gpu, params, alpha=1.0 - self.ema_decay)
                return
            for ema_param, param in zip(self.ema_model_cpu.parameters(), self.model.parameters()):
                if param.requires_grad:
                    ema_param.data.mul_(self.ema_decay).add_(param.cpu().data, alpha=(1.0 - self.ema_decay))

    def sync_ema_to_cpu(self):
        """Refresh `ema_model_cpu` from the GPU-resident average (no-op without one); returns the module."""
        if self._ema_gpu is not None:
            with torch.no_grad():
                dst = [q for q, p in zip(self.ema_model_cpu.parameters(), self.model.parameters()) if p.requires_grad]
                for q, e in zip(dst, self._ema_gpu):
                    q.data.copy_(e, non_blocking=True)
            torch.cuda.synchronize(self.device)
        return self.ema_model_cpu

    def ema_state_dict(self):
        """state_dict of the EMA model with the GPU-resident averages as DEVICE tensors (no host copy, no synchronisation):
        what the checkpoint streamer snapshots.  Without a GPU-resident average this is ema_model_cpu.state_dict()."""
        sd = self.ema_model_cpu.state_dict()
        if self._ema_gpu is not None:
            names = [n for n, p in self.model.named_parameters() if p.requires_grad]
            for n, e in zip(names, self._ema_gpu):
                sd[n] = e
        return sd

    def save_checkpoint(self, n):
        """The reference's six files for optimizer step n (model_trainer.py:545-548); streamed in the background when
        async_checkpoint is on (call self.ckpt_stream.wait() before reading them back)."""
        self.model.wandb_id = self.wandb_id
        if self.ckpt_stream is not None:
            ema = self.ema_state_dict() if self.ema_model_cpu is not None else None
        else:
            ema = self.sync_ema_to_cpu().state_dict() if self.ema_model_cpu is not None else None
        self.model.saveModel(saveDir=self.saveDir, EMA_state_dict=ema, optimizer=self.optim, scheduler=self.scheduler, grad_scalar=self.grad_scaler,
                             step=n, streamer=self.ckpt_stream)

    def train(self):
        if dist.is_initialized():
            dist.barrier()
        self.model.train()
        batch_loss, t0 = 0.0, time.time()
        replay_checks = 0
        opt_steps = self.start_step // self.accumulation_steps
        tuned = not (self.autotune and self.reducer.enabled and self.world > 1)
        capture_tried = False
        step = opt_steps
        while step < self.totalSteps:
            if not tuned and step - opt_steps >= 1:
                # data parallel, after one step (allocations done): which bucket algorithm / CU reserve is fastest HERE -- a collective decision over real
                # training steps (autotune_reducer: 45 of them, 55 with the RCCL round, counted; logging / checkpoint hooks resume behind them)
                tuned = True
                if self.totalSteps - step > 60:
                    before = step
                    step, _ = self.autotune_reducer(step)
                    if self.keep_losses:
                        self.loss_history.extend(self.autotune_losses)
                    if self.ema_model_cpu is not None and step // self.ema_update_freq > before // self.ema_update_freq:
                        self.update_ema()
                    continue
            if self.graph_after is not None and self._graph is None and not capture_tried and tuned and step - opt_steps >= max(3, self.graph_after):
                capture_tried = True
                if self.capture_graph_agreed(step + 1):      # (every rank reaches this line at the same step: the decision is collective)
                    replay_checks = 2
            loss = self.train_step(step + 1)
            if replay_checks > 0:
                # second vote, after each of the first two replays: a replay that did not train on some rank sends every rank back to eager launches
                replay_checks -= 1
                if not self.keep_graph_if_agreed(loss_is_plausible(loss), what=f"replayed step {step + 1}"):
                    replay_checks = 0
            if self.keep_losses:
                self.loss_history.append(loss)
            n = step + 1
            batch_loss += float(loss) if n % self.log_steps == 0 else 0.0
            if n % self.log_steps == 0 and is_main_process():
                rec = {"step": n, "loss": float(loss), "lr": self.optim.param_groups[0]["lr"],
                       "images_per_sec": self.world * self.batchSize * self.accumulation_steps * self.log_steps / (time.time() - t0)}
                print(json.dumps(rec), flush=True)
                if self.log_file:
                    with open(self.log_file, "a") as f:
                        f.write(json.dumps(rec) + "\n")
                t0 = time.time()
            if self.ema_model_cpu is not None and n % self.ema_update_freq == 0:
                self.update_ema()
            if n % self.numSaveSteps == 0 and is_main_process():
                self.save_checkpoint(n)
                print("Saving model")
            step += 1
        if self.ckpt_stream is not None:
            self.ckpt_stream.wait()
