"""Constructor kwargs of the released models and seeded synthetic weights (there is no network for checkpoints:
bench.py, the tools and smoke() run on random-init weights of the right architecture).

The policy kwargs are what a `.model` pickle carries (`run_agent.py:11-14`); 2x is `agent.py:16-36`, 1x / 3x change
(hidsize, attention_heads, impala_width) only and reproduce the published parameter counts.  The IDM
kwargs are the released 4x inverse-dynamics model as recalled in SURVEY.md §8a."""
import math
from typing import Dict

import torch

MODEL_SIZES = {
    "1x": dict(hidsize=1024, attention_heads=8, impala_width=4),
    "2x": dict(hidsize=2048, attention_heads=16, impala_width=8),
    "3x": dict(hidsize=3072, attention_heads=24, impala_width=12),
}


def policy_kwargs_for(name: str) -> dict:
    kw = dict(
        attention_heads=16, attention_mask_style="clipped_causal", attention_memory_size=256,
        diff_mlp_embedding=False, hidsize=2048, img_shape=[128, 128, 3], impala_chans=[16, 32, 32],
        impala_kwargs={"post_pool_groups": 1}, impala_width=8,
        init_norm_kwargs={"batch_norm": False, "group_norm_groups": 1}, n_recurrence_layers=4,
        only_img_input=True, pointwise_ratio=4, pointwise_use_activation=False,
        recurrence_is_residual=True, recurrence_type="transformer", timesteps=128,
        use_pointwise_layer=True, use_pre_lstm_ln=False,
    )
    kw.update(MODEL_SIZES[name])
    return kw


def idm_kwargs_for(name: str = "4x") -> dict:
    kw = dict(
        attention_heads=32, attention_mask_style="none", attention_memory_size=128,
        conv3d_params=dict(inchan=3, outchan=128, kernel_size=[5, 1, 1], padding=[2, 0, 0]),
        hidsize=4096, img_shape=[128, 128, 128], impala_chans=[16, 32, 32], impala_kwargs={"post_pool_groups": 1},
        impala_width=16, init_norm_kwargs={"batch_norm": False, "group_norm_groups": 1}, n_recurrence_layers=2,
        only_img_input=True, pointwise_ratio=4, pointwise_use_activation=False, recurrence_is_residual=True,
        recurrence_type="transformer", single_output=True, timesteps=128, use_pointwise_layer=True,
        use_pre_lstm_ln=False,
    )
    if name == "tiny":
        kw.update(hidsize=512, attention_heads=4, impala_width=2)
    return kw


@torch.no_grad()
def randomize_(module: torch.nn.Module, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded weights in place, by parameter shape/name: conv and linear weights at fan-in scale, EVERY gain / bias /
    relative-position table randomised too (a default init of gain 1 / bias 0 would hide affine bugs and make the
    normalisation layers trivially cheap to get right).  Deterministic across machines (CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    for name, p in module.named_parameters():
        shape = tuple(p.shape)
        if "normalizer." in name:                         # EWMA statistics of the value head: fixed, valid (var > 0)
            w = torch.full(shape, {"running_mean": 0.3, "running_mean_sq": 1.5, "debiasing_term": 0.9}[name.rsplit(".", 1)[1]])
        elif p.dim() >= 3:                                  # conv2d / conv3d
            fan = shape[1] * math.prod(shape[2:])
            w = torch.randn(shape, generator=g) * (1.6 / math.sqrt(fan))
        elif p.dim() == 2 and name.endswith("b_nd"):
            w = 0.2 * torch.randn(shape, generator=g)
        elif p.dim() == 2:
            w = torch.randn(shape, generator=g) * ((0.3 if "pi_head" in name else 1.0) * 1.3 / math.sqrt(shape[1]))
        elif name.endswith("weight"):                     # norm gains
            w = 1.0 + 0.2 * torch.randn(shape, generator=g)
        else:                                             # biases
            w = 0.1 * torch.randn(shape, generator=g)
        p.copy_(w.to(p.device, p.dtype))
    return dict(module.named_parameters())
