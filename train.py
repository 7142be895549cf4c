"""Training entry point: the reference's src/train.py:9-143 on the MI355X-native path, with synthetic data.

Every hyper-parameter below is the reference's literal value (src/train.py:33-66, stage 3 of README.md:251-262): 19 blocks,
dim = 64 * 19 = 1216, 19 heads, SwiGLU, softmax_flash, RoPE2d, per-GPU batch 13 -- the file says 13, its run name says 14 --
x 2 accumulation steps, lr 1e-4 constant after 1000 warm-up steps, EMA 0.99 every 100 steps, null probabilities
0.1 / 0.316 / 0.316, checkpoints every 1000 steps.  What differs, and why:
  * data: the reference's loader GPUs (VAE + text encoders feeding the model GPUs, `loader_to_model_gpu`) are out of this build's scope
    (SURVEY.md 2); batches are synthetic latents / text embeddings of the wire format's shapes (model_trainer.SyntheticData), or --
    with --vae-in-rank -- synthetic IMAGES encoded by the HIP FLUX-VAE inside the training rank;
  * one process per GPU, every rank a model rank: `python train.py` on one GPU, or
        python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py
    for data parallel over RCCL (gradient all-reduce per block on a side HIP stream, reducer.GradReducer);
  * --graph-after N: after N eager optimizer steps the whole step (micro-steps, collectives, clip, AdamW) is replayed from a hipGraph.
No wandb; one JSON line per `log_steps` optimizer steps on stdout (rank 0), and with --json one final summary line.

    python train.py --steps 20                        # the reference's stage-3 shape at 1024^2 (S = 4250), 2 x 13 images per step
    python train.py --steps 50 --max-res 256          # stage 1 resolution
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--steps", type=int, default=1_500_000, help="totalSteps (optimizer steps); the reference trains 1.5 M")
    ap.add_argument("--batch", type=int, default=13, help="per-GPU micro-batch (src/train.py:13)")
    ap.add_argument("--accumulation-steps", type=int, default=2)
    ap.add_argument("--max-res", type=int, default=1024, help="pixel resolution of the synthetic batches (256 / 512 / 1024 = the three stages)")
    ap.add_argument("--num-blocks", type=int, default=19)
    ap.add_argument("--graph-after", type=int, default=None, help="replay the step from a hipGraph after this many eager steps (>= 3)")
    ap.add_argument("--save-dir", default="models/synthetic_run")
    ap.add_argument("--save-steps", type=int, default=1000)
    ap.add_argument("--log-steps", type=int, default=10)
    ap.add_argument("--load", nargs=2, metavar=("DIR", "STEP"), help="resume: loadModel(DIR, model_<STEP>s.pkl, model_params_<STEP>s.json) + optimizer / scheduler / scaler / EMA files")
    ap.add_argument("--vae-in-rank", action="store_true", help="synthetic images -> HIP FLUX-VAE encode inside the rank (SURVEY 8f-1)")
    ap.add_argument("--json", action="store_true", help="print one summary JSON line at the end (tests)")
    a = ap.parse_args()

    import sd3_amd  # noqa: F401
    from sd3_amd.model_trainer import model_trainer
    from sd3_amd.models.diff_model import diff_model

    # ---- src/train.py:33-66, literal --------------------------------------------------------------
    inCh = 16
    class_dim = 768
    patch_size = 2
    num_blocks = a.num_blocks
    dim = int(64 * num_blocks)
    hidden_scale = 4.0
    num_heads = num_blocks
    attn_type = "softmax_flash"
    MLP_type = "swiglu"
    device = "gpu"
    max_res = a.max_res
    max_res_orig = 256
    null_prob_pooled = 0.1
    null_prob_gemma = 0.316
    null_prob_bert = 0.316
    lr = 1e-4
    use_lr_scheduler = False
    ema_update_freq = 100
    ema_decay = 0.99
    warmup_steps = 1000
    checkpoint_MLP = True          # accepted and ignored: 288 GB of HBM, nothing is recomputed
    checkpoint_attn = True
    positional_encoding = "RoPE2d"

    json_log = a.json or a.log_steps == 1
    model = diff_model(inCh=inCh, class_dim=class_dim, patch_size=patch_size, dim=dim, hidden_scale=hidden_scale, num_heads=num_heads,
                       attn_type=attn_type, MLP_type=MLP_type, num_blocks=num_blocks, checkpoint_MLP=checkpoint_MLP, checkpoint_attn=checkpoint_attn,
                       positional_encoding=positional_encoding, max_res_orig=max_res_orig, max_res=max_res, update_max_res=True,
                       kv_merge_attn=False, qk_half_dim=False, text_loss=False, device=device)
    files = {}
    if a.load:
        d, s = a.load
        model.loadModel(d, f"model_{s}s.pkl", f"model_params_{s}s.json")
        files = dict(load_ema_file=os.path.join(d, f"model_ema_{s}s.pkl"), optimFile=os.path.join(d, f"optim_{s}s.pkl"),
                     schedulerFile=os.path.join(d, f"scheduler_{s}s.pkl"), scalerFile=os.path.join(d, f"scaler_{s}s.pkl"))

    data_source = None
    if a.vae_in_rank:
        from sd3_amd.helpers.latent_source import ImageLatentSource
        data_source = ImageLatentSource.synthetic(a.batch, max_res, class_dim, model.device)

    p0 = float(sum(p.detach().double().norm() for p in model.parameters()))
    trainer = model_trainer(diff_model=model, batchSize=a.batch, accumulation_steps=a.accumulation_steps, totalSteps=a.steps, lr=lr,
                            ema_update_freq=ema_update_freq, ema_decay=ema_decay, warmup_steps=warmup_steps, use_lr_scheduler=use_lr_scheduler,
                            saveDir=a.save_dir, numSaveSteps=a.save_steps, null_prob_pooled=null_prob_pooled, null_prob_gemma=null_prob_gemma,
                            null_prob_bert=null_prob_bert, text_loss_weight=0.0, use_amp=True, log_steps=1 if json_log else a.log_steps,
                            device=device, max_res=max_res, data_source=data_source, device_rng=True, graph_after=a.graph_after,
                            log_file=None, **files)
    trainer.keep_losses = a.json
    trainer.train()
    torch.cuda.synchronize()
    if a.json and trainer.rank == 0:
        p1 = float(sum(p.detach().double().norm() for p in model.parameters()))
        print(json.dumps({"steps": a.steps, "dim": dim, "num_heads": num_heads, "num_blocks": num_blocks, "max_res": max_res,
                          "batch": a.batch, "accumulation_steps": a.accumulation_steps, "world": trainer.world,
                          "losses": [float(l) for l in trainer.loss_history], "replayed_steps": trainer.replayed_steps,
                          "param_norm_moved": bool(p1 != p0), "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)


if __name__ == "__main__":
    main()
