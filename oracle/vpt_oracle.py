"""CPU oracle for the VPT policy hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A flat, functional fp32 restatement (torch CPU ops) of the reference's policy forward:
IMPALA CNN -> banded-causal transformer with KV memory and relative-position bias ->
hierarchical categorical action heads + value head.  It takes the reference's `state_dict`
(same key names as the `.weights` file) and plain tensors; it owns no nn.Module.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
file.  The product path (`video-pre-training_amd/`) never does.

Pinning: the reference ships no tests or golden vectors for this path, so this
restatement is pinned against the *live, unmodified* reference imported from /root/reference in the
build container (`tests/golden/make_golden.py` writes `tests/golden/*.npz`; `tests/test_oracle_golden.py`
re-checks the oracle against those files everywhere, and against the live reference when present).

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

NEG_MASK = -1e9  # lib/xf.py:46  bias = (~mask).float() * -1e9
NORM_EPS = 1e-5  # torch default eps of nn.GroupNorm / nn.LayerNorm (lib/util.py:58-62)


# ----------------------------------------------------------------------------------------------
# configuration helpers
# ----------------------------------------------------------------------------------------------
def config_from_policy_kwargs(policy_kwargs: dict, pi_head_kwargs: Optional[dict] = None) -> dict:
    """Distil the numbers the math needs out of the reference's ctor kwargs
    (lib/policy.py:99-190, agent.py:16-38)."""
    pk = policy_kwargs
    width = pk.get("impala_width", 1)
    chans = [int(width * c) for c in pk.get("impala_chans", (16, 32, 32))]
    timesteps = pk.get("timesteps")
    memory = pk.get("attention_memory_size", 2048)
    mask_style = pk.get("attention_mask_style", "clipped_causal")
    return dict(
        chans=chans,
        hidsize=pk.get("hidsize", 512),
        heads=pk.get("attention_heads", 8),
        n_layers=pk.get("n_recurrence_layers", 1),
        maxlen=memory - timesteps,  # lib/masked_attention.py:137
        causal=(mask_style == "clipped_causal"),
        pointwise_ratio=pk.get("pointwise_ratio", 4),
        use_pre_lstm_ln=pk.get("use_pre_lstm_ln", True),
        temperature=float((pi_head_kwargs or {}).get("temperature", 1.0)),
        first_conv_norm=pk.get("first_conv_norm", False),
    )


# ----------------------------------------------------------------------------------------------
# norms
# ----------------------------------------------------------------------------------------------
def group_norm_1(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """GroupNorm(1, C) over (C,H,W) of each frame, biased variance, per-channel affine.
    Reference: nn.GroupNorm(group_norm_groups=1, inchan) built at lib/util.py:59-60."""
    n = x.shape[0]
    flat = x.reshape(n, -1)
    mu = flat.mean(dim=1)
    var = flat.var(dim=1, unbiased=False)
    xh = (x - mu.view(n, 1, 1, 1)) * torch.rsqrt(var + NORM_EPS).view(n, 1, 1, 1)
    return xh * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """LayerNorm over the last dim (lib/util.py:61-62, 169; lib/policy.py:188)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - mu) * torch.rsqrt(var + NORM_EPS) * w + b


# ----------------------------------------------------------------------------------------------
# IMPALA CNN  (lib/impala_cnn.py)
# ----------------------------------------------------------------------------------------------
def _norm_conv_relu(sd, pfx, x):
    """FanInInitReLULayer with layer_type=conv: norm -> conv3x3(pad 1) -> ReLU (lib/util.py:75-82).
    A bias exists iff there is no norm (lib/util.py:64-65)."""
    if pfx + "norm.weight" in sd:
        x = group_norm_1(x, sd[pfx + "norm.weight"], sd[pfx + "norm.bias"])
    x = F.conv2d(x, sd[pfx + "layer.weight"], sd.get(pfx + "layer.bias"), padding=1)
    return torch.relu(x)


def cnn_stack(sd, pfx, x, taps=None):
    """CnnDownStack.forward (lib/impala_cnn.py:114-121): firstconv -> maxpool(3,2,1) -> GroupNorm `n`
    -> two residual blocks x + conv1(conv0(x)) (lib/impala_cnn.py:50-52)."""
    x = _norm_conv_relu(sd, pfx + "firstconv.", x)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    if taps is not None:
        taps[pfx + "pool"] = x
    x = group_norm_1(x, sd[pfx + "n.weight"], sd[pfx + "n.bias"])
    if taps is not None:
        taps[pfx + "n"] = x
    for blk in range(2):
        bp = f"{pfx}blocks.{blk}."
        y = _norm_conv_relu(sd, bp + "conv0.", x)
        if taps is not None:
            taps[bp + "conv0"] = y
        y = _norm_conv_relu(sd, bp + "conv1.", y)
        x = x + y
        if taps is not None:
            taps[bp + "out"] = x
    return x


def impala_cnn(sd, pfx, frames_nhwc_f32, taps=None):
    """ImpalaCNN.forward (lib/impala_cnn.py:187-195) on already-scaled fp32 frames [N,H,W,C]:
    NHWC->NCHW, three stacks, flatten in C,H,W order, dense = LN -> Linear(no bias) -> ReLU."""
    x = frames_nhwc_f32.permute(0, 3, 1, 2)
    for s in range(3):
        x = cnn_stack(sd, f"{pfx}stacks.{s}.", x, taps)
    x = x.reshape(x.shape[0], -1)  # CHW flatten (lib/impala_cnn.py:192-193, lib/torch_util.py:107-112)
    x = layer_norm(x, sd[pfx + "dense.norm.weight"], sd[pfx + "dense.norm.bias"])
    x = torch.relu(x @ sd[pfx + "dense.layer.weight"].t())
    if taps is not None:
        taps[pfx + "dense"] = x
    return x


def img_obs_process(sd, pfx, img_u8, taps=None):
    """ImgPreprocessing (/255, lib/policy.py:39-45) + ImgObsProcess (lib/policy.py:79-80):
    CNN then LN -> Linear(256->hid, no bias) -> ReLU.  img_u8: [B,T,H,W,3] uint8 -> [B,T,hid]."""
    b, t = img_u8.shape[:2]
    x = img_u8.reshape(b * t, *img_u8.shape[2:]).to(torch.float32) / 255.0
    x = impala_cnn(sd, pfx + "cnn.", x, taps)
    x = layer_norm(x, sd[pfx + "linear.norm.weight"], sd[pfx + "linear.norm.bias"])
    x = torch.relu(x @ sd[pfx + "linear.layer.weight"].t())
    return x.reshape(b, t, -1)


# ----------------------------------------------------------------------------------------------
# masked attention with KV memory  (lib/masked_attention.py, lib/xf.py, lib/util.py)
# ----------------------------------------------------------------------------------------------
def band_visibility(t: int, maxlen: int, first_b: torch.Tensor, state_mask: Optional[torch.Tensor]):
    """Which of the T = t+maxlen keys (memory first, then this chunk) each of the t queries may see.
    Restates get_band_diagonal_mask + get_mask (lib/masked_attention.py:38-41, 75-83): key j is in the
    band of query i iff i+1 <= j <= i+maxlen (j = i+maxlen is the query itself); memory keys (j < maxlen)
    additionally need state_mask[b,j] and not first[b,0].  Returns (visible bool[B,t,T], new_state_mask
    bool[B,1,maxlen]) -- the update is lib/masked_attention.py:86-92."""
    bsz = first_b.shape[0]
    T = t + maxlen
    if state_mask is None:
        state_mask = torch.zeros(bsz, 1, maxlen, dtype=torch.bool)
    i = torch.arange(t).view(t, 1)
    j = torch.arange(T).view(1, T)
    band = (j >= i + 1) & (j <= i + maxlen)
    vis = band.unsqueeze(0).repeat(bsz, 1, 1)
    not_first = ~first_b.view(bsz, 1, 1)
    vis[:, :, :maxlen] &= not_first
    vis[:, :, :maxlen] &= state_mask
    new_mask = torch.cat(
        [state_mask[:, :, t:] & not_first, torch.ones(bsz, 1, min(t, maxlen), dtype=torch.bool)], dim=-1
    )
    return vis, new_mask


def rel_pos_bias(r_bthn: torch.Tensor, b_nd: torch.Tensor, t: int, maxlen: int) -> torch.Tensor:
    """Relative-position logits (lib/xf.py:265-271 + bandify, lib/util.py:232-267):
    extra[b,h,i,j] = sum_n R[b,i,h,n] * b_nd[n, maxlen + i - j] inside the band, 0 outside.
    r_bthn: [B,t,H,nbasis]."""
    T = t + maxlen
    i = torch.arange(t).view(t, 1)
    j = torch.arange(T).view(1, T)
    off = maxlen + i - j  # 0 = self, maxlen-1 = oldest visible key
    ok = (off >= 0) & (off < maxlen)
    d_ntT = b_nd[:, off.clamp(0, maxlen - 1)] * ok.unsqueeze(0)  # [n,t,T]
    return torch.einsum("bihn,nij->bhij", r_bthn, d_ntT)


def self_attention(sd, pfx, x1, first_b, state, heads: int, maxlen: int, causal: bool = True):
    """MaskedAttention.forward (lib/masked_attention.py:161-178) around SelfAttentionLayer.residual
    (lib/xf.py:334-360).  x1 is the already layer-normed input; returns (x1 + proj(attn), state_out).
    state = (state_mask or None, (K_mem, V_mem)) with K/V fp32 [B,maxlen,hid], un-split."""
    state_mask, (k_mem, v_mem) = state
    bsz, t, hid = x1.shape
    dh = hid // heads
    q = x1 @ sd[pfx + "q_layer.weight"].t() + sd[pfx + "q_layer.bias"]
    k = x1 @ sd[pfx + "k_layer.weight"].t()
    v = x1 @ sd[pfx + "v_layer.weight"].t()
    # update_state (lib/xf.py:366-391): keys = [memory ; new], next memory = last `maxlen` of that
    k_full = torch.cat([k_mem[:, max(k_mem.shape[1] - maxlen, 0):], k], dim=1)
    v_full = torch.cat([v_mem[:, max(v_mem.shape[1] - maxlen, 0):], v], dim=1)
    T = k_full.shape[1]
    k_out = k_full[:, max(T - maxlen, 0):]
    v_out = v_full[:, max(T - maxlen, 0):]

    def split(z):  # lib/xf.py:96-103: e = head*dh + d
        return z.reshape(bsz, z.shape[1], heads, dh).permute(0, 2, 1, 3)

    qh, kh, vh = split(q), split(k_full), split(v_full)
    logits = torch.matmul(qh, kh.transpose(-1, -2)) * (1.0 / dh)  # muP scale 1/d (lib/xf.py:59)
    if causal:
        vis, new_mask = band_visibility(t, maxlen, first_b, state_mask)
        logits = logits + (~vis).float().unsqueeze(1) * NEG_MASK
    else:
        new_mask = state_mask
    if maxlen > 0:
        r = x1 @ sd[pfx + "r_layer.weight"].t() + sd[pfx + "r_layer.bias"]  # [B,t,heads*10], head-major
        nb = sd[pfx + "b_nd"].shape[0]
        logits = logits + rel_pos_bias(r.reshape(bsz, t, heads, nb), sd[pfx + "b_nd"], t, maxlen)
    w = torch.softmax(logits, dim=-1)
    a = torch.matmul(w, vh).permute(0, 2, 1, 3).reshape(bsz, t, hid)
    out = x1 + a @ sd[pfx + "proj_layer.weight"].t() + sd[pfx + "proj_layer.bias"]
    return out, (new_mask, (k_out, v_out))


def recurrent_block(sd, pfx, x, first_b, state, heads, maxlen, causal=True):
    """ResidualRecurrentBlock.forward (lib/util.py:193-211): x1 = LN(x); x2 = x1 + Attn(x1)
    (the attention's own residual adds to the *normed* input); out = x2 + mlp1(relu(mlp0(LN(x2))))."""
    x1 = layer_norm(x, sd[pfx + "pre_r_ln.weight"], sd[pfx + "pre_r_ln.bias"])
    x2, state_out = self_attention(sd, pfx + "r.orc_block.", x1, first_b, state, heads, maxlen, causal)
    h = layer_norm(x2, sd[pfx + "mlp0.norm.weight"], sd[pfx + "mlp0.norm.bias"])
    h = torch.relu(h @ sd[pfx + "mlp0.layer.weight"].t())
    out = x2 + h @ sd[pfx + "mlp1.layer.weight"].t() + sd[pfx + "mlp1.layer.bias"]
    return out, state_out


# ----------------------------------------------------------------------------------------------
# heads
# ----------------------------------------------------------------------------------------------
LOG0 = -100.0   # lib/action_head.py:13


def categorical_head(sd, pfx, latent, temperature: float, mask=None):
    """CategoricalActionHead.forward (lib/action_head.py:163-174): linear, /temperature, [~mask] = LOG0, fp32 log_softmax.
    Returns [B,T,1,n] like the reference (output_shape = shape + (n,), shape=(1,)); mask: bool [B,T,1,n] or None."""
    z = latent @ sd[pfx + "linear_layer.weight"].t() + sd[pfx + "linear_layer.bias"]
    z = (z / temperature).unsqueeze(-2)
    if mask is not None:
        z = torch.where(mask, z, torch.full_like(z, LOG0))
    return torch.log_softmax(z.float(), dim=-1)


def value_head(sd, pfx, latent):
    """ScaledMSEHead.forward (lib/scaled_mse_head.py:34-35)."""
    return latent @ sd[pfx + "linear.weight"].t() + sd[pfx + "linear.bias"]


def denormalize_value(sd, pfx, v):
    """NormalizeEwma.denormalize (lib/normalize_ewma.py:27-31, 57-60)."""
    deb = sd[pfx + "normalizer.debiasing_term"].clamp(min=1e-5)
    mean = sd[pfx + "normalizer.running_mean"] / deb
    mean_sq = sd[pfx + "normalizer.running_mean_sq"] / deb
    var = (mean_sq - mean ** 2).clamp(min=1e-2)
    return v * torch.sqrt(var) + mean


# ----------------------------------------------------------------------------------------------
# whole policy
# ----------------------------------------------------------------------------------------------
def initial_state(cfg: dict, batch: int):
    """MinecraftAgentPolicy.initial_state (lib/policy.py:243-244 -> lib/masked_attention.py:153-159)."""
    z = lambda: torch.zeros(batch, cfg["maxlen"], cfg["hidsize"])
    return [(None, (z(), z())) for _ in range(cfg["n_layers"])]


def policy_forward(sd: Dict[str, torch.Tensor], cfg: dict, img_u8: torch.Tensor, first: torch.Tensor,
                   state_in: List[Tuple], taps: Optional[dict] = None, grad: bool = False, mask: Optional[dict] = None):
    """MinecraftAgentPolicy.forward (lib/policy.py:252-269) -> MinecraftPolicy.forward (lib/policy.py:193-218).
    img_u8 [B,T,128,128,3] uint8, first [B,T] bool (only first[:,0] is honoured, lib/masked_attention.py:167).
    Returns dict(buttons, camera log-probs [B,T,1,n]; vpred [B,T,1]; latent [B,T,hid]; state_out).
    grad=True keeps the autograd graph (used by bc_loss_and_grads)."""
    with torch.set_grad_enabled(grad):
        x = img_obs_process(sd, "net.img_process.", img_u8, taps)
        if taps is not None:
            taps["img_process"] = x
        if cfg.get("use_pre_lstm_ln", False):
            x = layer_norm(x, sd["net.pre_lstm_ln.weight"], sd["net.pre_lstm_ln.bias"])
        first_b = first[:, 0]
        state_out = []
        for l in range(cfg["n_layers"]):
            x, s = recurrent_block(sd, f"net.recurrent_layer.blocks.{l}.", x, first_b, state_in[l],
                                   cfg["heads"], cfg["maxlen"], cfg["causal"])
            state_out.append(s)
            if taps is not None:
                taps[f"block{l}"] = x
        x = torch.relu(x)
        x = layer_norm(x, sd["net.lastlayer.norm.weight"], sd["net.lastlayer.norm.bias"])
        x = torch.relu(x @ sd["net.lastlayer.layer.weight"].t())
        if taps is not None:
            taps["y"] = x
        x = layer_norm(x, sd["net.final_ln.weight"], sd["net.final_ln.bias"])
        out = dict(
            # obs["mask"] -> DictActionHead.forward(mask=...) (lib/policy.py:257-266, lib/action_head.py:240-248)
            buttons=categorical_head(sd, "pi_head.buttons.", x, cfg["temperature"], (mask or {}).get("buttons")),
            camera=categorical_head(sd, "pi_head.camera.", x, cfg["temperature"], (mask or {}).get("camera")),
            vpred=value_head(sd, "value_head.", x),
            latent=x,
            state_out=state_out,
        )
        return out


# ----------------------------------------------------------------------------------------------
# deterministic synthetic weights (shared by golden generation, tests, smoke and bench)
# ----------------------------------------------------------------------------------------------
def state_dict_spec(cfg: dict) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) for every tensor of MinecraftAgentPolicy.state_dict() in module order
    (SURVEY.md §8b).  kind in {conv, lin, gain, bias, bnd, ewma}."""
    spec = []
    cin = 3
    for s, c in enumerate(cfg["chans"]):
        p = f"net.img_process.cnn.stacks.{s}."
        if s > 0 or cfg.get("first_conv_norm", False):
            spec += [(p + "firstconv.norm.weight", (cin,), "gain"), (p + "firstconv.norm.bias", (cin,), "bias")]
            spec += [(p + "firstconv.layer.weight", (c, cin, 3, 3), "conv")]
        else:
            spec += [(p + "firstconv.layer.weight", (c, cin, 3, 3), "conv"), (p + "firstconv.layer.bias", (c,), "bias")]
        spec += [(p + "n.weight", (c,), "gain"), (p + "n.bias", (c,), "bias")]
        for b in range(2):
            for cv in range(2):
                q = f"{p}blocks.{b}.conv{cv}."
                spec += [(q + "norm.weight", (c,), "gain"), (q + "norm.bias", (c,), "bias"),
                         (q + "layer.weight", (c, c, 3, 3), "conv")]
        cin = c
    flat = cfg["chans"][-1] * 16 * 16
    hid = cfg["hidsize"]
    p = "net.img_process.cnn.dense."
    spec += [(p + "norm.weight", (flat,), "gain"), (p + "norm.bias", (flat,), "bias"), (p + "layer.weight", (256, flat), "lin")]
    p = "net.img_process.linear."
    spec += [(p + "norm.weight", (256,), "gain"), (p + "norm.bias", (256,), "bias"), (p + "layer.weight", (hid, 256), "lin")]
    if cfg.get("use_pre_lstm_ln", False):
        spec += [("net.pre_lstm_ln.weight", (hid,), "gain"), ("net.pre_lstm_ln.bias", (hid,), "bias")]
    r = cfg["pointwise_ratio"]
    for l in range(cfg["n_layers"]):
        p = f"net.recurrent_layer.blocks.{l}."
        spec += [(p + "mlp0.norm.weight", (hid,), "gain"), (p + "mlp0.norm.bias", (hid,), "bias"),
                 (p + "mlp0.layer.weight", (hid * r, hid), "lin"),
                 (p + "mlp1.layer.weight", (hid, hid * r), "lin"), (p + "mlp1.layer.bias", (hid,), "bias"),
                 (p + "pre_r_ln.weight", (hid,), "gain"), (p + "pre_r_ln.bias", (hid,), "bias")]
        o = p + "r.orc_block."
        spec += [(o + "b_nd", (10, cfg["maxlen"]), "bnd"),
                 (o + "q_layer.weight", (hid, hid), "lin"), (o + "q_layer.bias", (hid,), "bias"),
                 (o + "k_layer.weight", (hid, hid), "lin"), (o + "v_layer.weight", (hid, hid), "lin"),
                 (o + "proj_layer.weight", (hid, hid), "lin"), (o + "proj_layer.bias", (hid,), "bias"),
                 (o + "r_layer.weight", (10 * cfg["heads"], hid), "lin"), (o + "r_layer.bias", (10 * cfg["heads"],), "bias")]
    spec += [("net.lastlayer.norm.weight", (hid,), "gain"), ("net.lastlayer.norm.bias", (hid,), "bias"),
             ("net.lastlayer.layer.weight", (hid, hid), "lin"),
             ("net.final_ln.weight", (hid,), "gain"), ("net.final_ln.bias", (hid,), "bias")]
    spec += [("value_head.linear.weight", (1, hid), "lin"), ("value_head.linear.bias", (1,), "bias"),
             ("value_head.normalizer.running_mean", (1,), "ewma"), ("value_head.normalizer.running_mean_sq", (1,), "ewma"),
             ("value_head.normalizer.debiasing_term", (), "ewma")]
    spec += [("pi_head.camera.linear_layer.weight", (cfg.get("n_camera", 121), hid), "lin"),
             ("pi_head.camera.linear_layer.bias", (cfg.get("n_camera", 121),), "bias"),
             ("pi_head.buttons.linear_layer.weight", (cfg.get("n_buttons", 8641), hid), "lin"),
             ("pi_head.buttons.linear_layer.bias", (cfg.get("n_buttons", 8641),), "bias")]
    return spec


def peak_heads(sd: Dict[str, torch.Tensor], seed: int = 0) -> Dict[str, torch.Tensor]:
    """The "peaked" head family: the action heads of a TRAINED policy are far from uniform -- a strong prior over actions
    (large biases, one action clearly ahead) plus an input-dependent part several times the near-uniform init.  Re-draws
    pi_head.*.bias ~ N(0, 4^2), lifts each softmax group's largest bias to 8 above the runner-up, and scales pi_head.*.weight by
    1 / 0.3 (fan-in scale 1.3 / sqrt(hid) instead of 0.39 / sqrt(hid)).  With temperature 2 the prior's top-2 margin is 4 nat;
    the input-dependent part moves logit differences by sigma ~ 0.9 nat from frame to frame around a class-dependent offset of up
    to ~2 nat (the latent's constant component), so the oracle's top-2 margin stays above 1 nat at (nearly) every position
    (near-uniform family: median 0.02-0.05 nat, where an arg-max comparison is a coin flip for ANY
    finite-precision implementation) and the exact-action assertions of the GPU tests cover (nearly) every position in both
    operand formats.  Softmax groups: one per policy head; 20 x 2 / 2 x 11 for the IDM heads.
    Returns a new dict; everything but the pi_head tensors is shared with `sd`."""
    g = torch.Generator().manual_seed(1000 + seed)
    out = dict(sd)
    for key in sorted(k for k in sd if k.startswith("pi_head.")):
        if key.endswith(".weight"):
            out[key] = (sd[key] / 0.3).contiguous()
        else:
            n = sd[key].numel()
            groups = {40: (20, 2), 22: (2, 11)}.get(n, (1, n))
            b = (4.0 * torch.randn(n, generator=g)).view(groups)
            top = b.topk(2, dim=-1)
            b.scatter_(-1, top.indices[:, :1], top.values[:, 1:2] + 8.0)
            out[key] = b.reshape(sd[key].shape).contiguous()
    return out


def scene_frames(b: int, t: int, n_scenes: int, generator, noise: int = 10):
    """uint8 [b, t, 128, 128, 3] frames drawn from `n_scenes` low-frequency base images (+-noise per pixel), and the scene index of every
    frame [b, t]: WHICH scene a frame shows is a property of the input that the network's latent carries linearly (fit_scene_heads)."""
    low = torch.randint(0, 256, (n_scenes, 3, 4, 4), generator=generator).float()
    base = torch.nn.functional.interpolate(low, size=(128, 128), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    ids = torch.randint(0, n_scenes, (b, t), generator=generator)
    nz = torch.randint(-noise, noise + 1, (b, t, 128, 128, 3), generator=generator).float()
    return (base[ids] + nz).clamp(0, 255).to(torch.uint8).contiguous(), ids


def fit_scene_heads(sd: Dict[str, torch.Tensor], latent: torch.Tensor, scene: torch.Tensor, temperature: float, n_live: int = 16,
                    margin_nat: float = 4.0, ridge: float = 1e-2, seed: int = 0) -> Dict[str, torch.Tensor]:
    """The "competitive" head family: INPUT-DRIVEN decisions with margins above 16-bit noise.  Per policy head, `n_live` classes stay in
    play (every other class: zero weights, scaled logit -12 nat, i.e. never chosen and free of input-dependent error); the first
    n_scenes of them are a linear read-out of the scene the current frame shows, fitted by ridge regression on the ORACLE's latents of
    the very frames the test runs (target: margin_nat above the other live classes after the temperature), the remaining live classes
    are random directions of the same norm (they compete, rarely win).  The arg-max therefore changes with the input -- one action per
    scene -- and nothing in the bias decides it: all live classes share bias terms that only centre the read-out.
    latent [N, hid] (oracle, fp32), scene [N] int64.  Returns a new dict; only pi_head tensors differ from `sd`."""
    g = torch.Generator().manual_seed(2000 + seed)
    z = latent.double()
    mu = z.mean(0)
    zc = z - mu
    n_sc = int(scene.max()) + 1
    alpha = margin_nat * temperature
    y = torch.nn.functional.one_hot(scene, n_sc).double() * alpha - alpha / n_sc
    gram = zc @ zc.t()
    coef = torch.linalg.solve(gram + ridge * gram.diagonal().mean() * torch.eye(gram.shape[0], dtype=torch.float64), y)
    w_sc = (zc.t() @ coef).t()                                           # [n_sc, hid]
    out = dict(sd)
    for head in ("buttons", "camera"):
        wk, bk = f"pi_head.{head}.linear_layer.weight", f"pi_head.{head}.linear_layer.bias"
        n = sd[bk].numel()
        live = torch.randperm(n, generator=g)[:max(n_live, n_sc)]
        w = torch.zeros(n, z.shape[1], dtype=torch.float64)
        b = torch.full((n,), -12.0 * temperature, dtype=torch.float64)
        perm = torch.randperm(n_sc, generator=g)                         # the two heads map scenes to classes differently
        w[live[:n_sc]] = w_sc[perm]
        b[live[:n_sc]] = -(w_sc[perm] @ mu)
        extra = live[n_sc:]
        if extra.numel():
            r = torch.randn(extra.numel(), z.shape[1], generator=g, dtype=torch.float64)
            r = r / r.norm(dim=1, keepdim=True) * w_sc.norm(dim=1).mean()
            w[extra] = r
            b[extra] = -(r @ mu) - 0.25 * alpha
        out[wk], out[bk] = w.float().contiguous(), b.float().contiguous()
    return out


def synthetic_state_dict(cfg: dict, seed: int = 0, heads: str = "uniform") -> Dict[str, torch.Tensor]:
    """Seeded weights with the reference's shapes and roughly its init scales, but with *every* 1-D
    parameter randomised (default init leaves gains 1 / biases 0, hiding affine bugs -- SURVEY.md §7).
    Deterministic across machines (CPU torch.Generator); no network, no checkpoint needed.
    heads="peaked": the same weights with peak_heads() applied (trained-policy-like action distributions)."""
    if heads == "peaked":
        return peak_heads(synthetic_state_dict(cfg, seed), seed)
    if heads != "uniform":
        raise ValueError(f"heads must be 'uniform' or 'peaked', got {heads!r}")
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape, kind in state_dict_spec(cfg):
        if kind == "conv":
            fan = shape[1] * 9
            w = torch.randn(shape, generator=g) * (1.6 / math.sqrt(fan))
        elif kind == "lin":
            scale = 1.0
            if "pi_head" in key:
                scale = 0.3
            w = torch.randn(shape, generator=g) * (scale * 1.3 / math.sqrt(shape[1]))
        elif kind == "gain":
            w = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif kind == "bias":
            w = 0.1 * torch.randn(shape, generator=g)
        elif kind == "bnd":
            w = 0.2 * torch.randn(shape, generator=g)
        elif kind == "ewma":
            w = {"running_mean": torch.tensor([0.3]), "running_mean_sq": torch.tensor([1.5]),
                 "debiasing_term": torch.tensor(0.9)}[key.rsplit(".", 1)[1]].clone()
        else:
            raise ValueError(kind)
        sd[key] = w.float()
    return sd


MODEL_CONFIGS = {
    # policy_kwargs of the released foundation models.  2x is agent.py:16-36 verbatim; 1x / 3x change
    # (hidsize, heads, impala_width) only (they reproduce 71M / 248M / 0.5B parameters).
    "1x": dict(hidsize=1024, attention_heads=8, impala_width=4),
    "2x": dict(hidsize=2048, attention_heads=16, impala_width=8),
    "3x": dict(hidsize=3072, attention_heads=24, impala_width=12),
}


def policy_kwargs_for(name: str) -> dict:
    base = dict(
        attention_heads=16, attention_mask_style="clipped_causal", attention_memory_size=256,
        diff_mlp_embedding=False, hidsize=2048, img_shape=[128, 128, 3], impala_chans=[16, 32, 32],
        impala_kwargs={"post_pool_groups": 1}, impala_width=8,
        init_norm_kwargs={"batch_norm": False, "group_norm_groups": 1}, n_recurrence_layers=4,
        only_img_input=True, pointwise_ratio=4, pointwise_use_activation=False,
        recurrence_is_residual=True, recurrence_type="transformer", timesteps=128,
        use_pointwise_layer=True, use_pre_lstm_ln=False,
    )
    base.update(MODEL_CONFIGS[name])
    return base


# ----------------------------------------------------------------------------------------------
# inverse dynamics model (IDM)  --  lib/policy.py:342-467
# ----------------------------------------------------------------------------------------------
def idm_config_from_kwargs(idm_net_kwargs: dict, pi_head_kwargs: Optional[dict] = None) -> dict:
    cfg = config_from_policy_kwargs(idm_net_kwargs, pi_head_kwargs)
    cfg["conv3d"] = idm_net_kwargs.get("conv3d_params")
    cfg["first_conv_norm"] = cfg["conv3d"] is not None  # lib/policy.py:360-363
    cfg["use_pre_lstm_ln"] = idm_net_kwargs.get("use_pre_lstm_ln", True)
    return cfg


def idm_kwargs_for(name: str = "4x") -> dict:
    """Constructor kwargs of the released IDM as recalled in SURVEY.md §8a ([unverified]: no .model file is
    available); `tiny` keeps the structure at sizes a CPU can run for golden vectors."""
    base = dict(
        attention_heads=32, attention_mask_style="none", attention_memory_size=128,
        conv3d_params=dict(inchan=3, outchan=128, kernel_size=[5, 1, 1], padding=[2, 0, 0]),
        hidsize=4096, img_shape=[128, 128, 128], impala_chans=[16, 32, 32], impala_kwargs={"post_pool_groups": 1},
        impala_width=16, init_norm_kwargs={"batch_norm": False, "group_norm_groups": 1}, n_recurrence_layers=2,
        only_img_input=True, pointwise_ratio=4, pointwise_use_activation=False, recurrence_is_residual=True,
        recurrence_type="transformer", single_output=True, timesteps=128, use_pointwise_layer=True,
        use_pre_lstm_ln=False,
    )
    if name == "tiny":
        base.update(hidsize=512, attention_heads=4, impala_width=2)
    return base


def idm_state_dict_spec(cfg: dict, n_buttons=20, n_camera_bins=11):
    spec = [("net.conv3d_layer.layer.weight", (cfg["conv3d"]["outchan"], cfg["conv3d"]["inchan"], 5, 1, 1), "conv3d"),
            ("net.conv3d_layer.layer.bias", (cfg["conv3d"]["outchan"],), "bias")]
    cin = cfg["conv3d"]["outchan"]
    for s, c in enumerate(cfg["chans"]):
        p = f"net.img_process.cnn.stacks.{s}."
        spec += [(p + "firstconv.norm.weight", (cin,), "gain"), (p + "firstconv.norm.bias", (cin,), "bias"),
                 (p + "firstconv.layer.weight", (c, cin, 3, 3), "conv")]
        spec += [(p + "n.weight", (c,), "gain"), (p + "n.bias", (c,), "bias")]
        for b in range(2):
            for cv in range(2):
                q = f"{p}blocks.{b}.conv{cv}."
                spec += [(q + "norm.weight", (c,), "gain"), (q + "norm.bias", (c,), "bias"), (q + "layer.weight", (c, c, 3, 3), "conv")]
        cin = c
    policy_like = [e for e in state_dict_spec(dict(cfg, first_conv_norm=True)) if not e[0].startswith("net.img_process.cnn.stacks.")
                   and not e[0].startswith("value_head.") and not e[0].startswith("pi_head.")]
    spec += policy_like
    hid = cfg["hidsize"]
    spec += [("pi_head.buttons.linear_layer.weight", (n_buttons * 2, hid), "lin"), ("pi_head.buttons.linear_layer.bias", (n_buttons * 2,), "bias"),
             ("pi_head.camera.linear_layer.weight", (2 * n_camera_bins, hid), "lin"), ("pi_head.camera.linear_layer.bias", (2 * n_camera_bins,), "bias")]
    return spec


def idm_synthetic_state_dict(cfg: dict, seed: int = 0, heads: str = "uniform"):
    if heads == "peaked":
        return peak_heads(idm_synthetic_state_dict(cfg, seed), seed)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape, kind in idm_state_dict_spec(cfg):
        if kind == "conv":
            w = torch.randn(shape, generator=g) * (1.6 / math.sqrt(shape[1] * 9))
        elif kind == "conv3d":
            w = torch.randn(shape, generator=g) * (1.6 / math.sqrt(shape[1] * 5))
        elif kind == "lin":
            w = torch.randn(shape, generator=g) * ((0.3 if "pi_head" in key else 1.0) * 1.3 / math.sqrt(shape[1]))
        elif kind == "gain":
            w = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif kind == "bias":
            w = 0.1 * torch.randn(shape, generator=g)
        elif kind == "bnd":
            w = 0.2 * torch.randn(shape, generator=g)
        else:
            raise ValueError(kind)
        sd[key] = w.float()
    return sd


def conv3d_temporal(sd, x_bthwc):
    """InverseActionNet._conv3d_forward (lib/policy.py:394-403): Conv3d(C->O, kernel (5,1,1), pad (2,0,0)) over
    time + ReLU (FanInInitReLULayer without norm => bias).  x [B,T,H,W,C] fp32 -> [B,T,H,W,O]."""
    w = sd["net.conv3d_layer.layer.weight"]  # [O,C,5,1,1]
    b = sd["net.conv3d_layer.layer.bias"]
    x = x_bthwc.permute(0, 4, 1, 2, 3)  # b c t h w
    y = torch.relu(F.conv3d(x, w, b, padding=(2, 0, 0)))
    return y.permute(0, 2, 3, 4, 1)


def idm_forward(sd, cfg, img_u8, taps: Optional[dict] = None):
    """InverseActionPolicy.forward (lib/policy.py:432-446) -> InverseActionNet.forward (lib/policy.py:374-392):
    /255 -> temporal Conv3d -> ImpalaCNN (first conv normed) -> linear -> transformer blocks with mask "none"
    and no memory (maxlen 0: rel-pos bias is identically zero, lib/util.py:256-260) -> ReLU -> final_ln
    (`lastlayer` is computed and DISCARDED by the reference, lib/policy.py:390-391) -> heads.
    Returns dict(buttons [B,T,20,2], camera [B,T,2,11]) of log-probabilities."""
    with torch.no_grad():
        b, t = img_u8.shape[:2]
        x = img_u8.to(torch.float32) / 255.0
        x = conv3d_temporal(sd, x)
        if taps is not None:
            taps["conv3d"] = x
        x = impala_cnn(sd, "net.img_process.cnn.", x.reshape(b * t, *x.shape[2:]), taps)
        p = "net.img_process.linear."
        x = torch.relu(layer_norm(x, sd[p + "norm.weight"], sd[p + "norm.bias"]) @ sd[p + "layer.weight"].t())
        x = x.reshape(b, t, -1)
        hid, heads = cfg["hidsize"], cfg["heads"]
        for l in range(cfg["n_layers"]):
            pfx = f"net.recurrent_layer.blocks.{l}."
            o = pfx + "r.orc_block."
            x1 = layer_norm(x, sd[pfx + "pre_r_ln.weight"], sd[pfx + "pre_r_ln.bias"])
            q = x1 @ sd[o + "q_layer.weight"].t() + sd[o + "q_layer.bias"]
            k = x1 @ sd[o + "k_layer.weight"].t()
            v = x1 @ sd[o + "v_layer.weight"].t()
            sp = lambda z: z.reshape(b, t, heads, hid // heads).permute(0, 2, 1, 3)
            w_ = torch.softmax(torch.matmul(sp(q), sp(k).transpose(-1, -2)) * (1.0 / (hid // heads)), dim=-1)
            a = torch.matmul(w_, sp(v)).permute(0, 2, 1, 3).reshape(b, t, hid)
            x2 = x1 + a @ sd[o + "proj_layer.weight"].t() + sd[o + "proj_layer.bias"]
            h = torch.relu(layer_norm(x2, sd[pfx + "mlp0.norm.weight"], sd[pfx + "mlp0.norm.bias"]) @ sd[pfx + "mlp0.layer.weight"].t())
            x = x2 + h @ sd[pfx + "mlp1.layer.weight"].t() + sd[pfx + "mlp1.layer.bias"]
        x = torch.relu(x)
        x = layer_norm(x, sd["net.final_ln.weight"], sd["net.final_ln.bias"])
        temp = cfg["temperature"]
        zb = (x @ sd["pi_head.buttons.linear_layer.weight"].t() + sd["pi_head.buttons.linear_layer.bias"]).reshape(b, t, -1, 2)
        zc = (x @ sd["pi_head.camera.linear_layer.weight"].t() + sd["pi_head.camera.linear_layer.bias"]).reshape(b, t, 2, -1)
        return dict(buttons=torch.log_softmax(zb / temp, -1), camera=torch.log_softmax(zc / temp, -1), latent=x)


# ----------------------------------------------------------------------------------------------
# behavioural-cloning step (behavioural_cloning.py:86-123), generalised to [B, T] chunks
# ----------------------------------------------------------------------------------------------
def bc_loss_and_grads(sd, cfg, img_u8, first, state_in, act_buttons, act_camera):
    """loss = -mean_{b,t}[ log pi(buttons_bt) + log pi(camera_bt) ] and its gradient w.r.t. every tensor of `sd`
    via torch autograd through this restatement.  The reference computes exactly this per sample with T = 1 and
    B = 1 (`-log_prob / BATCH_SIZE`, behavioural_cloning.py:107-119, log_prob = pi_head.logprob = sum of the two
    heads' gathers, lib/action_head.py:176-184,252-253) and the KV memory enters detached
    (behavioural_cloning.py:111); here the mean runs over the B*T frames of a chunk.  act_* : int64 [B,T].
    Returns (loss float, dict name -> grad tensor (zeros for tensors the loss does not reach), state_out)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point}
    state_det = [(m, (k.detach(), v.detach())) for m, (k, v) in state_in]
    out = policy_forward(leaves, cfg, img_u8, first, state_det, grad=True)
    with torch.enable_grad():    # (the caller may sit inside torch.no_grad(): the GPU tests' inference fixture)
        lp = out["buttons"][:, :, 0].gather(-1, act_buttons.unsqueeze(-1)).squeeze(-1) \
            + out["camera"][:, :, 0].gather(-1, act_camera.unsqueeze(-1)).squeeze(-1)
        loss = -lp.mean()
        names = list(leaves)
        grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    gd = {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in zip(names, grads)}
    state_out = [(m, (k.detach(), v.detach())) for m, (k, v) in out["state_out"]]
    return float(loss.detach()), gd, state_out
