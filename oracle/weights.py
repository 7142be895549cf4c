"""Deterministic synthetic weights for the parity tests.  TEST INFRASTRUCTURE ONLY.

state_dict_spec() restates the reference's state_dict key order and shapes
(probed from /root/reference/src/models/diff_model.py:126-216 and
blocks/Transformer_Block_Dual.py:15-53, blocks/Attention.py:16-98; pinned by
tests/golden/state_dict_spec_*.json which tools/make_goldens.py writes from the
real reference).  make_state_dict() fills it from a seeded CPU generator so
weights never need to be stored in fixtures.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch


def state_dict_spec(dim: int, num_heads: int, num_blocks: int, hidden_scale: float = 4.0,
                    inCh: int = 16, class_dim: int = 768, patch_size: int = 2,
                    MLP_type: str = "swiglu", text_hidden: int = 2304) -> List[Tuple[str, Tuple[int, ...]]]:
    d, h, hd = dim, int(dim * hidden_scale), dim // num_heads
    spec: List[Tuple[str, Tuple[int, ...]]] = [("learnable_scalar", (1,)), ("learnable_scalar2", (1,)), ("time_scale", (1,))]

    def mlp(p):
        if MLP_type == "swiglu":
            return [(p + "MLP.w12.weight", (2 * h, d)), (p + "MLP.w12.bias", (2 * h,)),
                    (p + "MLP.w3.weight", (d, h)), (p + "MLP.w3.bias", (d,))]
        return [(p + "lin_up.weight", (h, d)), (p + "lin_up.bias", (h,)),
                (p + "lin_down.weight", (d, h)), (p + "lin_down.bias", (d,))]

    for i in range(num_blocks):
        last = i == num_blocks - 1
        p = f"blocks.{i}."
        spec += [(p + "y_proj.0.weight", (d, d)), (p + "y_proj.0.bias", (d,))]
        spec += mlp(p + "MLP_x.")
        if not last:
            spec += mlp(p + "MLP_c.")
        for n in ["query_proj_x", "key_proj_x", "value_proj_x", "out_proj_x", "query_proj_c", "key_proj_c", "value_proj_c"]:
            spec.append((p + f"attn.{n}.weight", (d, d)))
        if not last:
            spec.append((p + "attn.out_proj_c.weight", (d, d)))
        for n in ["q_norm_x", "k_norm_x", "q_norm_c", "k_norm_c"]:
            spec.append((p + f"attn.{n}.weight", (hd,)))
        spec.append((p + "attn.rotary_emb.freqs", (hd // 4,)))
        norms = ["norm1_x", "norm2_x", "norm1_c"] + ([] if last else ["norm2_c"])
        for n in norms:
            spec += [(p + f"{n}.c_shift.weight", (d, d)), (p + f"{n}.c_scale.weight", (d, d))]
        gates = ["scale1_x", "scale2_x"] + ([] if last else ["scale1_c", "scale2_c"])
        for n in gates:
            spec.append((p + f"{n}.weight", (d, d)))
    spec += [("t_emb2.weight", (d, d)), ("cond_MLP.weight", (d, class_dim)),
             ("c_proj.weight", (d, text_hidden)), ("c_proj2.weight", (d, text_hidden)),
             ("pre_c_norm.weight", (text_hidden,)), ("pre_c_norm2.weight", (text_hidden,)),
             ("patch_emb.weight", (d, d)), ("patch_emb.bias", (d,)),
             ("pos_enc.proj.weight", (d, inCh, patch_size, patch_size)),
             ("out_norm.c_shift.weight", (d, d)), ("out_norm.c_scale.weight", (d, d)),
             ("out_proj.weight", (inCh * patch_size * patch_size, d)), ("out_proj.bias", (inCh * patch_size * patch_size,))]
    return spec


def make_state_dict(seed: int, **cfg) -> Dict[str, torch.Tensor]:
    """Seeded weights: randn*std per key in spec order.  Linear-like weights use
    std = 1/sqrt(fan_in) (keeps activations O(1) through depth), norm weights are
    1 + 0.1*randn, biases 0.02*randn, scalars keep the reference's init values
    (0.01, 0.01, 1000; diff_model.py:171-172, 213), rotary freqs keep their
    closed form (rotary_embedding.py:120)."""
    g = torch.Generator().manual_seed(seed)
    num_heads, dim = cfg["num_heads"], cfg["dim"]
    hd = dim // num_heads
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in state_dict_spec(**cfg):
        leaf = name.split(".")[-1]
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        if name in ("learnable_scalar", "learnable_scalar2"):
            v = torch.tensor([0.01])
        elif name == "time_scale":
            v = torch.tensor([1000.0])
        elif leaf == "freqs":
            d2 = hd // 2
            v = 1.0 / (10000 ** (torch.arange(0, d2, 2)[: d2 // 2].float() / d2))
        elif "norm" in name and leaf == "weight" and len(shape) == 1:
            v = 1.0 + 0.1 * r
        elif leaf == "bias":
            v = 0.02 * r
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            std = fan_in ** -0.5
            if ".c_scale." in name or ".c_shift." in name or "scale1_" in name or "scale2_" in name:
                std *= 0.5  # modulation / gate vectors: keep |gate| < 1 so depth stays stable
            v = std * r
        sd[name] = v.contiguous()
    return sd


def make_inputs(seed: int, batch: int, h: int, w: int, inCh: int = 16, class_dim: int = 768,
                text_hidden: int = 2304, text_scale: float = 1.0):
    """Seeded (x_t, c, c_pooled) triple shaped like the reference's batches
    (model_trainer.py:353-355): tokens 0..76 Gemma-like (optionally large
    variance), tokens 77..153 ModernBERT-like with the upper 1280 features
    zero-padded (helpers/VAE_T5_CLIP.py:419-427)."""
    g = torch.Generator().manual_seed(1000 + seed)
    x = torch.randn((batch, inCh, h, w), generator=g)
    c = torch.randn((batch, 154, text_hidden), generator=g)
    c[:, :77] *= text_scale
    c[:, 77:, 1024:] = 0
    cp = torch.randn((batch, class_dim), generator=g)
    return x, c, cp
