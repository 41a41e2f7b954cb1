"""Forward engine of the VPT policy on MI355X: packs the reference's fp32 weights once, then drives the
HIP kernels (ops.py -> libvpt_hip.so) in the order of MinecraftPolicy.forward (lib/policy.py:193-218).

Data flow for N = B*T frames (all device resident; activations bf16 channel-blocked, residual stream and
KV memory fp32):

  uint8 frames --conv_first--> P0 --affine(n)--> x --[conv3x3, conv3x3+res] x2--> stack0
               --conv3x3--> --maxpool--> --affine(n)--> ... stack1, stack2
               --affine(per-element LN)--> --linear(split-K 65536->256)--> --layernorm(relu)--> --linear(->hid, relu)
  4 x [ layernorm -> linear(QKVR) -> masked_attention (+kv_memory_update) -> linear(proj,+res)
        -> layernorm -> linear(mlp0, relu) -> linear(mlp1,+res) ]
  layernorm(relu in) -> linear(lastlayer, relu) -> layernorm(final) -> linear(heads) -> log_softmax x2
"""
import os
from typing import Dict, List, Optional

import torch

from . import ops, packing

# split-K of the K = 65536 dense layer (per-split partial slices summed in a fixed order: deterministic)
DENSE_SPLITK = int(os.environ.get("VPT_DENSE_SPLITK", "32"))   # 32 x 16 tiles = one full round of workgroups per 1024-frame chunk (16: half a round; profiles/r04_experiments.md section 20)


def config_from_policy_kwargs(policy_kwargs: dict, pi_head_kwargs: Optional[dict] = None) -> dict:
    """The numbers the engine needs, from the reference's ctor kwargs (lib/policy.py:99-190, agent.py:16-38)."""
    pk = policy_kwargs
    width = pk.get("impala_width", 1)
    chans = [int(width * c) for c in pk.get("impala_chans", (16, 32, 32))]
    cfg = dict(
        chans=chans,
        hidsize=pk.get("hidsize", 512),
        heads=pk.get("attention_heads", 8),
        n_layers=pk.get("n_recurrence_layers", 1),
        maxlen=pk.get("attention_memory_size", 2048) - (pk.get("timesteps") or 0),
        causal=pk.get("attention_mask_style", "clipped_causal") == "clipped_causal",
        pointwise_ratio=pk.get("pointwise_ratio", 4),
        use_pre_lstm_ln=pk.get("use_pre_lstm_ln", True),
        temperature=float((pi_head_kwargs or {}).get("temperature", 1.0)),
        img_shape=list(pk.get("img_shape") or [128, 128, 3]),
    )
    return cfg


def check_supported(cfg: dict, idm: bool = False):
    """Fail loudly for configurations the HIP path does not implement yet."""
    h, w, c = cfg["img_shape"]
    if (h, w) != (128, 128) or (c != 3 and not idm):
        raise NotImplementedError(f"HIP path supports 128x128x3 frames only, got {cfg['img_shape']}")
    if any(ch % 32 for ch in cfg["chans"]):
        raise NotImplementedError(f"IMPALA widths must be multiples of 32, got {cfg['chans']}")
    if cfg["hidsize"] != cfg["heads"] * 128:
        raise NotImplementedError("attention kernel is built for d_head = 128")
    if idm:
        if cfg["causal"] or cfg["maxlen"] != 0:
            raise NotImplementedError("IDM path implements attention_mask_style='none' without memory (maxlen 0) only")
    else:
        if not cfg["causal"] or not (1 <= cfg["maxlen"] <= 129):
            raise NotImplementedError("only the clipped_causal mask with 1 <= maxlen <= 129 is implemented")
    if cfg["hidsize"] % 256:
        raise NotImplementedError("hidsize must be a multiple of 256")


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


PRECISIONS = {"bf16": torch.bfloat16, "fp16": torch.float16}


def resolve_precision(precision: Optional[str]) -> str:
    """precision=None -> env VPT_PRECISION -> "fp16", the parity mode (log-probs within the north star's 1e-3 of the fp32
    reference, exact actions outside a 10x narrower noise band).  "bf16" -- the north star's "MFMA bf16 tiles", what bench.py
    reports as its headline -- is the explicit opt-in: same kernels, same speed, 8x coarser operand rounding."""
    p = precision or os.environ.get("VPT_PRECISION", "fp16")
    if p not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {p!r}")
    return p


RNG_STREAM = {"buttons": 0, "camera": 1}     # which draw of a step a head takes from the in-kernel generator (ops.log_softmax_cols: rng)


def action_heads(logits, heads, bsz, t, temperature, mask=None, sample=None, keep_head_logps=False, rng_state=None):
    """DictActionHead.forward (+ sample / logprob) over the fused head logits.  heads: (name, first column, groups, classes);
    a head's logits are [M, groups * classes] -> log-probs [B, T, groups, classes].  Returns the dict entries to merge.
    keep_head_logps: leave the heads' action log-probs un-summed in out["_head_logps"] (the acting step's epilogue kernel adds them).
    sample="stochastic" draws the uniforms of CategoricalActionHead.sample (th.rand_like, lib/action_head.py:200) INSIDE the head
    kernel from `rng_state` (ops.new_rng_state: device {seed, step}); the caller advances the step once all heads have drawn."""
    if sample not in (None, "deterministic", "stochastic"):
        raise ValueError(f"sample must be None, 'deterministic' or 'stochastic', got {sample!r}")
    if sample == "stochastic" and rng_state is None:
        raise ValueError("action_heads: stochastic sampling needs the generator state (ops.new_rng_state)")
    out, actions, logp, head_lps = {}, {}, None, {}
    for name, col0, groups, n in heads:
        z = logits if groups == 1 else logits[:, col0:col0 + groups * n].reshape(-1, n)
        c0 = col0 if groups == 1 else 0
        rows = z.shape[0]
        mk = None
        if mask is not None and mask.get(name) is not None:
            mk = mask[name].reshape(rows, n).to(torch.uint8).contiguous()
        if sample is None:
            lp = ops.log_softmax_cols(z, c0, n, temperature, mask=mk)
        else:
            rng = (rng_state, RNG_STREAM[name]) if sample == "stochastic" else None
            lp, ac, alp = ops.log_softmax_cols(z, c0, n, temperature, mask=mk, want_action=True, rng=rng)
            actions[name] = ac.view(bsz, t, groups)
            alp = alp.view(bsz, t) if groups == 1 else alp.view(bsz, t, groups).sum(-1)     # (no reduction kernel for the usual single group)
            head_lps[name] = alp
            if not keep_head_logps:
                logp = alp if logp is None else logp + alp
        out[name] = lp.view(bsz, t, groups, n)
    if sample is not None:
        out["action"] = actions
        if keep_head_logps:
            out["_head_logps"] = head_lps
        else:
            out["action_log_prob"] = logp
    return out


def unpack_act_tail(keep, bsz):
    """The caller-visible entries of the acting step as views of the packed record (T = 1)."""
    ab, ac, lp, vd, v = ops.unpack_act_keep(keep)
    return dict(action={"buttons": ab.reshape(bsz, 1, 1), "camera": ac.reshape(bsz, 1, 1)}, action_log_prob=lp.reshape(bsz, 1),
                vpred_denorm=vd.reshape(bsz, 1), vpred=v.reshape(bsz, 1, 1))


class PolicyEngine:
    """precision: format of every 16-bit MFMA operand / stored CNN activation.  "fp16" (default: the parity mode) or "bf16"
    (what bench.py's headline measures): the same kernels built with IEEE-half / bfloat16 operands (libvpt_hip_f16.so /
    libvpt_hip.so) -- same MFMA rate and bytes; fp16 rounds 8x finer: log-probs within 2.5e-4 of the fp32 reference instead of
    1.8e-3 (profiles/r02_precision_sweep_1x.md).  Accumulation, statistics, softmax, residual stream and KV memory are fp32
    in both.  The BC step (training.py) runs in either; fp16 with dynamic loss scaling."""

    def __init__(self, cfg: dict, n_buttons: int, n_camera: int, cnn_chunk: int = 1024, cnn_streams: int = 3,
                 precision: Optional[str] = None):
        check_supported(cfg)
        self.precision = resolve_precision(precision)
        self.dtype = PRECISIONS[self.precision]
        self.cfg = cfg
        self.n_buttons, self.n_camera = n_buttons, n_camera
        self.cnn_chunk = int(os.environ.get("VPT_CNN_CHUNK", cnn_chunk))
        self.cnn_streams = int(os.environ.get("VPT_CNN_STREAMS", cnn_streams))
        # frames per conv + pool sub-chunk of stacks 1.. (0: whole chunk).  Measured (profiles/r03_experiments.md section 12): 64 / 128 / 256 frames
        # all give +1.2 % on the forward step (the conv launches themselves run 1.8 % faster at 128 / 256); the pool kernel's own time does not move
        self.pool_subchunk = int(os.environ.get("VPT_POOL_SUBCHUNK", 256))
        # stacks 1..: firstconv and the max-pool behind it as one pass (ops.conv3x3_pool); 0 = the two-kernel path above (A/B, and what the
        # latency tiling of the acting step always takes)
        self.fuse_pool = os.environ.get("VPT_FUSE_POOL", "1") != "0"
        self.fuse_pool_sub = int(os.environ.get("VPT_FUSE_POOL_SUB", 0))      # frames per pool-fused launch (0: the whole chunk)
        # each stack's GroupNorm `n` folded into its first block (no affine pass; needs fuse_pool): 0 = the vpt_affine_kernel pass
        self.fold_n = os.environ.get("VPT_FOLD_N", "1") != "0"
        self.fold_stats_in_producer = os.environ.get("VPT_FOLD_STATS", "1") != "0"     # 0: per-channel sums by a pass of their own (A/B)
        self.fold_dense = os.environ.get("VPT_FOLD_DENSE", "1") != "0"                 # dense.norm (LayerNorm 65536) folded into the dense GEMM
        self._streams = []
        # Cross-step overlap (opt-in, overlap_steps()): the CNN of call i + 1 is ordered behind the point where call i handed its CNN output to the
        # transformer -- not behind the transformer itself -- so it runs BESIDE the previous call's transformer and heads.
        self.step_overlap = False
        self._handoff = None        # event on the calling stream: the previous call's last read of chunk-stream memory is enqueued before it
        self._attn_done = None      # arrival counters of the in-place acting step (ops.masked_attention_step)
        self._rng_state = None      # in-kernel sampler state {seed, step}, created on first stochastic use (rng_state)
        self._rng_src, self._pending_seed = None, None
        self.w: Dict[str, torch.Tensor] = {}
        self.packed = False

    step_overlap, _handoff = False, None      # class defaults (IDMEngine builds its own __init__; it never forks chunk streams)
    _RNG_MARK = 8        # how far the engine advances torch's CUDA generator offset when it derives a seed from it (see rng_state)

    def rng_state(self, device):
        """The device-resident {seed, step} the stochastic heads draw from.  One per engine: a captured acting step holds its address.
        Seeding follows torch's generator the way the reference's th.rand_like does (lib/action_head.py:200): unless seed() was called, the
        sampler's seed is DERIVED from the device generator's (seed, offset), and the engine leaves a mark on that generator (its offset
        advanced by _RNG_MARK).  A later torch.manual_seed() -- even with the same seed -- resets the offset below the mark, and the next
        stochastic call re-derives {seed, step = 0} in place: "seed, run an episode, re-seed, run again" reproduces the draws.  Other consumers
        of the generator only move the offset forward and do not disturb the sampler (a re-seed followed by more than _RNG_MARK foreign draws
        BEFORE the engine's next stochastic call would go unnoticed: call seed() for that)."""
        cur = self._rng_state
        if cur is None or cur.device.type != device.type or (device.index is not None and cur.device.index != device.index):
            pending = getattr(self, "_pending_seed", None)
            self._rng_state = torch.zeros(2, dtype=torch.int64, device=device)
            self._rng_src = None
            if pending is not None:
                self._rng_state.copy_(torch.tensor([pending, 0], dtype=torch.int64))
                self._rng_src = "explicit"
            self._pending_seed = None
        if self._rng_src != "explicit" and not torch.cuda.is_current_stream_capturing():
            gen = torch.cuda.default_generators[self._rng_state.device.index]
            here = (int(gen.initial_seed()), int(gen.get_offset()))
            src = self._rng_src
            # re-derive only when a manual_seed happened: the seed changed, or the offset fell back below the engine's mark.  An offset that merely
            # ADVANCED belongs to another consumer of the generator (torch.rand, dropout, a second engine): the sampler keeps its stream and the
            # generator is left alone (every such consumer used to re-seed the sampler with step = 0 and cost a blocking copy per step)
            if not (isinstance(src, tuple) and src[0] == here[0] and here[1] >= src[1]):
                # splitmix-style mix of (seed, offset), masked to 63 bits (the state is int64)
                z = (here[0] * 0x9E3779B97F4A7C15 + here[1] * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
                z ^= z >> 31
                self._rng_state.copy_(torch.tensor([z & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64))
                gen.set_offset(here[1] + self._RNG_MARK)
                self._rng_src = (here[0], here[1] + self._RNG_MARK)
        return self._rng_state

    def seed(self, seed: int):
        """Re-seed the sampler in place (the captured graph keeps reading the same buffer) and restart its step counter; from here on the
        sampler no longer follows torch's generator."""
        seed = int(seed) & 0x7FFFFFFFFFFFFFFF
        if self._rng_state is not None:
            self._rng_state.copy_(torch.tensor([seed, 0], dtype=torch.int64))
            self._rng_src = "explicit"
        else:
            self._pending_seed = seed

    def adopt_sampler(self, other: "PolicyEngine"):
        """Take over another engine's sampler state (set_precision() builds a new engine: the seed and the step counter carry over)."""
        self._rng_state, self._rng_src = other._rng_state, getattr(other, "_rng_src", None)
        self._pending_seed = getattr(other, "_pending_seed", None)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def pack(self, sd: Dict[str, torch.Tensor]):
        """Re-pack the fp32 state_dict (reference key names) for the kernels."""
        cfg, w = self.cfg, {}
        f32 = lambda t: t.detach().float().contiguous()
        cin = 3
        for s, c in enumerate(cfg["chans"]):
            p = f"net.img_process.cnn.stacks.{s}."
            if s == 0:
                w[p + "firstconv"] = ops.pack_conv_first(f32(sd[p + "firstconv.layer.weight"]), f32(sd[p + "firstconv.layer.bias"]), dtype=self.dtype)
            else:
                w[p + "firstconv"] = ops.pack_conv3x3(f32(sd[p + "firstconv.layer.weight"]), f32(sd[p + "firstconv.norm.weight"]), f32(sd[p + "firstconv.norm.bias"]), dtype=self.dtype)
            w[p + "n.g"], w[p + "n.b"] = f32(sd[p + "n.weight"]), f32(sd[p + "n.bias"])
            for b in range(2):
                for cv in range(2):
                    q = f"{p}blocks.{b}.conv{cv}"
                    w[q] = ops.pack_conv3x3(f32(sd[q + ".layer.weight"]), f32(sd[q + ".norm.weight"]), f32(sd[q + ".norm.bias"]), dtype=self.dtype)
            cin = c
        c2 = cfg["chans"][-1]
        p = "net.img_process.cnn.dense."
        w[p + "g"] = ops.chw_to_blocked(f32(sd[p + "norm.weight"]), c2, 16, 16)
        w[p + "b"] = ops.chw_to_blocked(f32(sd[p + "norm.bias"]), c2, 16, 16)
        w[p + "w"] = ops.pack_linear(ops.chw_to_blocked(f32(sd[p + "layer.weight"]), c2, 16, 16), dtype=self.dtype)
        p = "net.img_process.linear."
        w[p + "g"], w[p + "b"] = f32(sd[p + "norm.weight"]), f32(sd[p + "norm.bias"])
        w[p + "w"] = ops.pack_linear(f32(sd[p + "layer.weight"]), dtype=self.dtype)
        hid = cfg["hidsize"]
        for l in range(cfg["n_layers"]):
            p = f"net.recurrent_layer.blocks.{l}."
            o = p + "r.orc_block."
            w[p + "ln1.g"], w[p + "ln1.b"] = f32(sd[p + "pre_r_ln.weight"]), f32(sd[p + "pre_r_ln.bias"])
            wq = torch.cat([f32(sd[o + "q_layer.weight"]), f32(sd[o + "k_layer.weight"]),
                            f32(sd[o + "v_layer.weight"]), f32(sd[o + "r_layer.weight"])], dim=0)
            nr = sd[o + "r_layer.weight"].shape[0]
            bq = torch.cat([f32(sd[o + "q_layer.bias"]), torch.zeros(2 * hid, device=wq.device), f32(sd[o + "r_layer.bias"])])
            w[p + "qkvr.w"], w[p + "qkvr.b"] = ops.pack_linear(wq, dtype=self.dtype), bq.contiguous()
            w[p + "b_nd"] = f32(sd[o + "b_nd"])
            w[p + "proj.w"], w[p + "proj.b"] = ops.pack_linear(f32(sd[o + "proj_layer.weight"]), dtype=self.dtype), f32(sd[o + "proj_layer.bias"])
            w[p + "ln2.g"], w[p + "ln2.b"] = f32(sd[p + "mlp0.norm.weight"]), f32(sd[p + "mlp0.norm.bias"])
            w[p + "mlp0.w"] = ops.pack_linear(f32(sd[p + "mlp0.layer.weight"]), dtype=self.dtype)
            w[p + "mlp1.w"], w[p + "mlp1.b"] = ops.pack_linear(f32(sd[p + "mlp1.layer.weight"]), dtype=self.dtype), f32(sd[p + "mlp1.layer.bias"])
            self.n_qkvr = 3 * hid + nr
        if cfg["use_pre_lstm_ln"]:    # MinecraftPolicy.pre_lstm_ln (lib/policy.py:186-188,202-203): the reference's default, off in the released models
            w["prelstm.g"], w["prelstm.b"] = f32(sd["net.pre_lstm_ln.weight"]), f32(sd["net.pre_lstm_ln.bias"])
        w["last.g"], w["last.b"] = f32(sd["net.lastlayer.norm.weight"]), f32(sd["net.lastlayer.norm.bias"])
        w["last.w"] = ops.pack_linear(f32(sd["net.lastlayer.layer.weight"]), dtype=self.dtype)
        w["final.g"], w["final.b"] = f32(sd["net.final_ln.weight"]), f32(sd["net.final_ln.bias"])
        wh = torch.cat([f32(sd["pi_head.buttons.linear_layer.weight"]), f32(sd["pi_head.camera.linear_layer.weight"]),
                        f32(sd["value_head.linear.weight"])], dim=0)
        bh = torch.cat([f32(sd["pi_head.buttons.linear_layer.bias"]), f32(sd["pi_head.camera.linear_layer.bias"]),
                        f32(sd["value_head.linear.bias"])])
        w["heads.w"], w["heads.b"] = ops.pack_linear(wh, dtype=self.dtype), bh.contiguous()
        self.w = w
        self.packed = True
        # sources of the n-fold tables (computed on first use after a pack: inference only, nothing on the BC step's path)
        self._fold_src = {s: (sd[f"net.img_process.cnn.stacks.{s}.blocks.0.conv0.layer.weight"], sd[f"net.img_process.cnn.stacks.{s}.blocks.0.conv0.norm.weight"])
                          for s in range(len(cfg["chans"]))}
        self._dense_src = sd["net.img_process.cnn.dense.layer.weight"]
        self._fold_tab = {}
        self._handoff = None        # the next forward orders its chunk streams behind EVERYTHING on the calling stream (the packing kernels above)

    def overlap_steps(self, enable: bool = True):
        """Opt-in pipelining of consecutive forward() calls (throughput path, more than one CNN chunk): with it, the chunk streams of a call wait
        for the previous call's hand-off point (its CNN output consumed) instead of for all earlier work on the calling stream, so this call's
        convolutions run beside the previous call's transformer + heads (7 % of a 2x step, launch-bound GEMMs and small kernels that leave most
        of the chip idle).  Results are unchanged -- the same kernels on the same data in the same per-stream order.
        CONTRACT: the frames passed to forward() must be complete on the device when forward() is called (resident, or uploaded on a copy
        stream whose event the HOST has waited for): work enqueued on the calling stream after the previous forward() returned is NOT waited for
        by the convolutions.  A resident but NON-CONTIGUOUS img (a slice, a permute) is fine: forward() makes the contiguous copy on the calling
        stream and orders the chunk streams behind it.  Off by default: a caller that produces frames on the calling stream right before
        forward() keeps working."""
        self.step_overlap = bool(enable)
        self._handoff = None

    def _prepare_fold_tables(self, tiling: str):
        """The lazily built constant tables of the folded path, built on the CALLING stream before the chunk streams fork: built inside the first
        chunk, the other chunks' streams would read them unordered (first forward after a pack)."""
        if tiling != "throughput":
            return
        if self.fold_dense:
            self._dense_fold()
        if self.fold_n and self.fuse_pool:
            for s in range(len(self.cfg["chans"])):
                self._nfold_tables(s)

    def _dense_fold(self):
        """ImpalaCNN.dense with its LayerNorm(65536) folded into the GEMM (inference): (weights packed from W * gain in blocked order,
        sg[n] = sum_k op16(W gain)[n][k], sb[n] = sum_k W[n][k] bias[k]).  Built on first use after a pack."""
        if "dense" not in self._fold_tab:
            w = self.w
            c2 = self.cfg["chans"][-1]
            wt = self._dense_src.detach().float()                                           # [256, C*16*16] in the reference's C,H,W order
            g, b = w["net.img_process.cnn.dense.g"], w["net.img_process.cnn.dense.b"]       # blocked order
            wb = ops.chw_to_blocked(wt.contiguous(), c2, 16, 16)                            # [256, K] blocked
            wg = wb * g.view(1, -1)
            wgr = wg.to(self.dtype)
            sg = wgr.double().sum(1).float().contiguous()
            sb = (wb.double() * b.double().view(1, -1)).sum(1).float().contiguous()
            self._fold_tab["dense"] = (ops.pack_linear(wg.contiguous(), dtype=self.dtype), sg, sb)
        return self._fold_tab["dense"]

    def _nfold_tables(self, s: int):
        """TB / TG of stack s: the edge-class sums of block 0's conv0 weights W' = op16(W * gain_conv0) with every input channel weighted by
        n.bias / n.weight (edge_sg is the same sum with weight 1) -- fp32 [9, CoutPad], vpt_nfold_coef's `tb` / `tg`."""
        if s not in self._fold_tab:
            wt, g0 = self._fold_src[s]
            p = f"net.img_process.cnn.stacks.{s}."
            gam, bet = self.w[p + "n.g"], self.w[p + "n.b"]
            cout = wt.shape[0]
            wp = (wt.detach().float() * g0.detach().float().view(1, -1, 1, 1)).to(self.dtype).double()        # the packed (rounded) weights
            m = packing.edge_tap_matrix(wp.device, torch.float64)                                            # [9 classes, 9 taps]
            pad = (cout + 127) // 128 * 128
            tabs = []
            for v in (bet, gam):
                tap = (wp * v.double().view(1, -1, 1, 1)).sum(1).view(cout, 9)                                # [Cout, 9]
                t = torch.zeros(9, pad, dtype=torch.float32, device=wp.device)
                t[:, :cout] = (m.unsqueeze(1) * tap.unsqueeze(0)).sum(-1).float()                             # (an index-sum, not a GEMM)
                tabs.append(t.contiguous())
            self._fold_tab[s] = tuple(tabs)
        return self._fold_tab[s]

    # ------------------------------------------------------------------------------------------
    def _cnn_dense(self, img, tiling: str = "throughput", x0=None, s_x0=None) -> torch.Tensor:
        """ImpalaCNN.forward up to the dense layer's pre-activation output, fp32 [F, 256].  Throughput path: the 65536-wide LayerNorm is folded
        into the split-K GEMM (_dense_fold + ops.dense_fold_epilogue: no per-element affine pass); otherwise affine pass + GEMM."""
        w = self.w
        f = img.shape[0] if x0 is None else x0.shape[0]
        if self.fold_dense and tiling == "throughput":
            x, s_x = self._cnn_chunk(img, x0=x0, s_x0=s_x0, tiling=tiling, raw=True)
            wg, sg, sb = self._dense_fold()
            flat = x.view(f, -1)
            part, _ = ops.linear(flat, wg, 256, splitk=DENSE_SPLITK, splitk_raw=True, tiling="throughput")
            return ops.dense_fold_epilogue(part, s_x, flat.shape[1], sg, sb)
        xn = self._cnn_chunk(img, x0=x0, s_x0=s_x0, tiling=tiling)
        d32, _ = ops.linear(xn.view(f, -1), w["net.img_process.cnn.dense.w"], 256, splitk=DENSE_SPLITK, tiling=tiling)
        return d32

    def _cnn_chunk(self, img: torch.Tensor, x0=None, s_x0=None, tiling: str = "throughput", raw: bool = False) -> torch.Tensor:
        """img uint8 [F,128,128,3] -> blocked bf16 [F, C2/32, 16, 16, 32] normalised for the dense layer,
        i.e. everything of ImpalaCNN.forward up to (and including) dense.norm.  With (x0, s_x0) given (IDM:
        the temporal conv's blocked output and its frame statistics) stack 0 uses the normed conv3x3 path."""
        cfg, w = self.cfg, self.w
        f = img.shape[0] if x0 is None else x0.shape[0]
        dev = img.device if x0 is None else x0.device
        fold = self.fold_n and self.fuse_pool and tiling == "throughput"
        # ONE zero-filled fp64 arena per chunk for every statistic its kernels accumulate into: 24 x (sum, sum of squares) per frame + (folded path)
        # the per-channel sums of each stack's pooled tensor -- one fill launch instead of one per tensor (VERDICT r4 "what's weak" 7)
        n_chs = sum(cfg["chans"]) if fold and self.fold_stats_in_producer else 0
        arena = torch.zeros(f * (48 + 2 * n_chs), dtype=torch.float64, device=dev)
        st = arena[:48 * f].view(24, f, 2)
        chs_off = 48 * f
        si = 0

        def nxt():
            nonlocal si
            si += 1
            return st[si - 1]

        x, s_x = x0, s_x0
        for s, c in enumerate(cfg["chans"]):
            p = f"net.img_process.cnn.stacks.{s}."
            s_pool = nxt()
            gn = w[p + "n.g"] if fold else None      # folded: the producer stores Q = n.weight * P (statistics: those of P)
            # ... and the per-channel sums of Q the fold needs (a pass of their own only where the producer cannot: vpt_channel_stats)
            chs = None
            if fold and self.fold_stats_in_producer:
                chs = arena[chs_off:chs_off + 2 * f * c].view(f, c, 2)
                chs_off += 2 * f * c
            if s == 0 and x0 is None:
                if c > 128:
                    chs = None
                pooled = ops.conv_first(img, w[p + "firstconv"], c, stats_out=s_pool, out_gain=gn, chs_out=chs)
            else:
                wpk, sa, sg = w[p + "firstconv"]
                if self.fuse_pool and tiling == "throughput":
                    # firstconv + ReLU + max-pool in ONE pass: the 16 x 16 output tiles are pooled in LDS, a seam kernel completes the windows
                    # that cross tile borders; the pre-pool tensor (2 MB per frame in stack 1) is never written (ops.conv3x3_pool)
                    fsub = self.fuse_pool_sub
                    if fsub and f > fsub:
                        pooled = torch.empty(f, c // 32, x.shape[2] // 2, x.shape[3] // 2, 32, dtype=x.dtype, device=x.device)
                        for i in range(0, f, fsub):
                            j = min(i + fsub, f)
                            ops.conv3x3_pool(x[i:j], wpk, sa, sg, s_x[i:j], c, stats_out=s_pool[i:j], out=pooled[i:j], out_gain=gn,
                                             chs_out=chs[i:j] if chs is not None else None)
                    else:
                        pooled = ops.conv3x3_pool(x, wpk, sa, sg, s_x, c, stats_out=s_pool, out_gain=gn, chs_out=chs)
                else:
                    sub = self.pool_subchunk
                    if sub and f > sub:
                        # conv + pool over sub-chunks of frames: the pre-pool tensor of a sub-chunk (2 MB per frame in stack 1) is pooled while it
                        # is still in the 256 MB Infinity Cache instead of after the whole chunk's 2 GB have gone through HBM
                        pooled = torch.empty(f, c // 32, x.shape[2] // 2, x.shape[3] // 2, 32, dtype=x.dtype, device=x.device)
                        for i in range(0, f, sub):
                            j = min(i + sub, f)
                            pre = ops.conv3x3(x[i:j], wpk, sa, sg, s_x[i:j], c, tiling=tiling)
                            ops.maxpool(pre, stats_out=s_pool[i:j], out=pooled[i:j])
                            del pre
                    else:
                        pre = ops.conv3x3(x, wpk, sa, sg, s_x, c, tiling=tiling)
                        pooled = ops.maxpool(pre, stats_out=s_pool)
                        del pre
            if fold:
                # GroupNorm `n` without a pass of its own: x = n(P) is never written.  Block 0's conv0 convolves Q with its
                # ordinary weights and a per-frame epilogue table, conv1 takes its residual as res_scale * Q + res_bias[c].
                hw = pooled.shape[2] * pooled.shape[3]
                if chs is None:
                    chs = ops.channel_stats(pooled)
                wpk, sa, sg = w[f"{p}blocks.0.conv0"]
                tb, tg = self._nfold_tables(s)
                kk, rs, rsc, rb = ops.nfold_coef(s_pool, chs, w[p + "n.g"], w[p + "n.b"], sa, sg, tb, tg, hw, c)
                s_y = nxt()
                y = ops.conv3x3_folded(pooled, wpk, sa, sg, None, c, kk_frame=kk, rs_frame=rs, stats_out=s_y)
                wpk, sa, sg = w[f"{p}blocks.0.conv1"]
                s_x = nxt()
                x = ops.conv3x3_folded(y, wpk, sa, sg, s_y, c, res=pooled, res_scale=rsc, res_bias=rb, stats_out=s_x)
                del y, pooled, chs, kk
                first_block = 1
            else:
                s_x = nxt()
                x = ops.frame_affine(pooled, w[p + "n.g"], w[p + "n.b"], s_pool, stats_out=s_x, out=pooled)
                first_block = 0
            for b in range(first_block, 2):
                wpk, sa, sg = w[f"{p}blocks.{b}.conv0"]
                s_y = nxt()
                y = ops.conv3x3(x, wpk, sa, sg, s_x, c, stats_out=s_y, tiling=tiling)
                wpk, sa, sg = w[f"{p}blocks.{b}.conv1"]
                s_n = nxt()
                x = ops.conv3x3(y, wpk, sa, sg, s_y, c, res=x, stats_out=s_n, tiling=tiling)
                s_x = s_n
                del y
        if raw:          # (the caller folds dense.norm into the dense GEMM: _cnn_dense)
            return x, s_x
        p = "net.img_process.cnn.dense."
        return ops.frame_affine(x, w[p + "g"], w[p + "b"], s_x, per_element=True)

    def _ln_linear(self, x, g, b, wpk, n, bias=None, res=None, relu=False, relu_in=False, ln_out_f32=False, out_f32=True, out_bf16=False,
                   tiling="throughput"):
        """LayerNorm -> linear.  tiling "latency" (the acting step: <= 8 rows): ONE launch (the normalisation is a prologue of the
        weight-streaming kernel); "throughput": vpt_layernorm_kernel + the MFMA GEMM whatever the row count -- the CALLER's choice, never
        the row count's, so a row's result does not depend on the batch around it.
        -> (normalised rows fp32 | None, fp32 out | None, 16-bit out | None)."""
        if tiling == "latency" and x.shape[0] <= ops.LN_LINEAR_MAX_ROWS and x.shape[1] <= ops.LN_LINEAR_MAX_K:
            return ops.layernorm_linear(x, g, b, wpk, n, bias=bias, res=res, relu=relu, relu_in=relu_in, ln_out_f32=ln_out_f32,
                                        out_f32=out_f32, out_bf16=out_bf16, dtype=self.dtype)
        ln32, ln16 = ops.layernorm(x, g, b, relu_in=relu_in, out_f32=ln_out_f32, dtype=self.dtype)
        o32, o16 = ops.linear(ln16, wpk, n, bias=bias, res=res, relu=relu, out_f32=out_f32, out_bf16=out_bf16, tiling=tiling)
        return ln32, o32, o16

    def _img_process(self, frames: torch.Tensor, tiling: str = "throughput", frames_ready=None) -> torch.Tensor:
        """uint8 [N,128,128,3] -> fp32 [N,hid]  (ImgObsProcess.forward, lib/policy.py:79-80).  tiling: see ops.conv3x3.
        frames_ready: an event on the calling stream behind which `frames` is complete (forward() made a contiguous copy there); the pipelined
        chunk streams wait for it -- they do not wait for the calling stream itself."""
        cfg, w = self.cfg, self.w
        n = frames.shape[0]
        outs = []
        n_chunks = (n + self.cnn_chunk - 1) // self.cnn_chunk
        # Frame chunks are independent: alternate them over `cnn_streams` HIP streams so that one chunk's
        # HBM-bound kernels (pool / affine / first conv) and launch tails overlap the other's MFMA-bound convs.
        n_streams = min(self.cnn_streams, n_chunks)
        main = torch.cuda.current_stream()
        if n_streams > 1:
            self._prepare_fold_tables(tiling)
            while len(self._streams) < n_streams:      # created once, appended to, never replaced
                self._streams.append(torch.cuda.Stream())
            pipelined = self.step_overlap and self._handoff is not None and not torch.cuda.is_current_stream_capturing()
            for st in self._streams[:n_streams]:
                if pipelined:
                    st.wait_event(self._handoff)       # overlap_steps(): behind the previous call's hand-off, beside its transformer
                    if frames_ready is not None:
                        st.wait_event(frames_ready)    # ... and behind the copy kernel that made `frames` on the calling stream
                else:
                    st.wait_stream(main)
        for ci, i in enumerate(range(0, n, self.cnn_chunk)):
            ctx = torch.cuda.stream(self._streams[ci % n_streams]) if n_streams > 1 else _NullCtx()
            with ctx:
                outs.append(self._cnn_dense(frames[i:i + self.cnn_chunk], tiling=tiling))
        if n_streams > 1:
            for st in self._streams[:n_streams]:
                main.wait_stream(st)
        d = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        p = "net.img_process.linear."
        _, x, _ = self._ln_linear(d, w[p + "g"], w[p + "b"], w[p + "w"], cfg["hidsize"], relu=True, relu_in=True, tiling=tiling)
        if n_streams > 1 and self.step_overlap and not torch.cuda.is_current_stream_capturing():
            # the hand-off: every read of memory that belongs to the chunk streams' allocator pools (`outs`) is enqueued on `main` before this event,
            # so the next call's chunks may reuse it once the event has fired; everything behind it on `main` (transformer, heads) they do not wait for
            self._handoff = torch.cuda.Event()
            self._handoff.record(main)
        return x

    @torch.no_grad()
    def forward(self, img_u8: torch.Tensor, first: torch.Tensor, state_in: List, mask: Optional[dict] = None,
                sample: Optional[str] = None, inplace_state: bool = False, act_tail: Optional[tuple] = None):
        """mask: optional {"buttons" / "camera": bool [B,T,1,n]} availability masks (obs["mask"], lib/policy.py:257-266).
        sample: None, "deterministic" or "stochastic" -- CategoricalActionHead.sample + logprob fused into the head kernel
        (lib/action_head.py:176-207); adds out["action"] (int64 [B,T,1] per head) and out["action_log_prob"] ([B,T]).
        inplace_state (T = 1 only): state_out IS state_in, masks included, updated in place (the captured acting graph's static state).
        act_tail = (scale, shift) of the value normaliser (T = 1, sample set): the tail of MinecraftAgentPolicy.act in one launch --
        adds out["_keep"] (ops.act_epilogue's packed record), out["nan_flag"], out["vpred_denorm"]; action / action_log_prob / vpred are
        views of the record."""
        if not self.packed:
            raise RuntimeError("PolicyEngine.pack(state_dict) must be called before forward")
        cfg, w = self.cfg, self.w
        bsz, t = img_u8.shape[:2]
        hid, heads, maxlen = cfg["hidsize"], cfg["heads"], cfg["maxlen"]
        frames = img_u8.reshape(bsz * t, *img_u8.shape[2:]).contiguous()
        frames_ready = None
        if self.step_overlap and frames.data_ptr() != img_u8.data_ptr() and not torch.cuda.is_current_stream_capturing():
            # a non-contiguous (but resident) img: .contiguous() just enqueued a copy kernel on the calling stream, which pipelined chunk streams
            # do not wait for -- hand them an event behind the copy (without it the convolutions could read `frames` before it is written)
            frames_ready = torch.cuda.Event()
            frames_ready.record(torch.cuda.current_stream())
        # the acting step (agent.py:190-206: T = 1, a handful of environments) runs the convolutions on the latency tiling
        tiling = "latency" if (t == 1 and bsz <= ops.LN_LINEAR_MAX_ROWS) else "throughput"     # one choice for every kernel of the step
        x = self._img_process(frames, tiling=tiling, frames_ready=frames_ready)
        if cfg["use_pre_lstm_ln"]:
            x, _ = ops.layernorm(x, w["prelstm.g"], w["prelstm.b"], out_f32=True, out_bf16=False, dtype=self.dtype)

        step = t == 1 and maxlen <= ops.ATTENTION_STEP_MAXLEN     # acting step: attention, memory shift and mask update in one launch
        if inplace_state and not step:
            raise ValueError("inplace_state is the acting step's option (T = 1)")
        if step:
            first8 = first[:, 0].contiguous().view(torch.uint8)
        else:
            not_first = ~first[:, 0].reshape(bsz, 1, 1)
        state_out = []
        for l in range(cfg["n_layers"]):
            p = f"net.recurrent_layer.blocks.{l}."
            state_mask, (kmem, vmem) = state_in[l]
            if state_mask is None:
                state_mask = torch.zeros(bsz, 1, maxlen, dtype=torch.bool, device=x.device)
            x1, qkvr, _ = self._ln_linear(x, w[p + "ln1.g"], w[p + "ln1.b"], w[p + "qkvr.w"], self.n_qkvr, bias=w[p + "qkvr.b"], ln_out_f32=True, tiling=tiling)
            if step:
                if inplace_state and not (kmem.is_contiguous() and vmem.is_contiguous()):
                    raise ValueError("inplace_state needs contiguous state tensors (a .contiguous() copy would receive the update instead of the state)")
                done = None
                if inplace_state:       # the mask too: the last workgroup to arrive writes it (ops.masked_attention_step)
                    if not state_mask.is_contiguous():
                        raise ValueError("inplace_state needs contiguous state masks")
                    if self._attn_done is None or self._attn_done.device != x.device:
                        self._attn_done = torch.zeros(64, dtype=torch.int32, device=x.device)
                    done = self._attn_done
                att, kout, vout, m8 = ops.masked_attention_step(qkvr, kmem.contiguous(), vmem.contiguous(), state_mask.reshape(bsz, maxlen).contiguous(), first8,
                                                               w[p + "b_nd"], bsz, heads, hid, dtype=self.dtype, inplace=inplace_state, done=done)
                new_mask = state_mask if inplace_state else m8.view(torch.bool).view(bsz, 1, maxlen)
            else:
                memvalid = (state_mask & not_first).reshape(bsz, maxlen).to(torch.uint8).contiguous()
                att = ops.masked_attention(qkvr, kmem.contiguous(), vmem.contiguous(), memvalid, w[p + "b_nd"], bsz, t, heads, hid, dtype=self.dtype)
                kout, vout = ops.kv_memory_update(qkvr, kmem.contiguous(), vmem.contiguous(), bsz, t, hid)
                new_mask = torch.cat([state_mask[:, :, t:] & not_first,
                                      torch.ones(bsz, 1, min(t, maxlen), dtype=torch.bool, device=x.device)], dim=-1)
            x2, _ = ops.linear(att, w[p + "proj.w"], hid, bias=w[p + "proj.b"], res=x1, tiling=tiling)
            _, _, h2 = self._ln_linear(x2, w[p + "ln2.g"], w[p + "ln2.b"], w[p + "mlp0.w"], hid * cfg["pointwise_ratio"], relu=True,
                                       out_f32=False, out_bf16=True, tiling=tiling)
            x, _ = ops.linear(h2, w[p + "mlp1.w"], hid, bias=w[p + "mlp1.b"], res=x2, tiling=tiling)
            state_out.append((new_mask, (kout, vout)))

        _, y, _ = self._ln_linear(x, w["last.g"], w["last.b"], w["last.w"], hid, relu=True, relu_in=True, tiling=tiling)
        nb, nc = self.n_buttons, self.n_camera
        latent, logits, _ = self._ln_linear(y, w["final.g"], w["final.b"], w["heads.w"], nb + nc + 1, bias=w["heads.b"], ln_out_f32=True, tiling=tiling)
        temp = cfg["temperature"]
        out = dict(latent=latent.view(bsz, t, hid), state_out=state_out)
        tail = act_tail is not None and sample is not None and t == 1
        rng = self.rng_state(x.device) if sample == "stochastic" else None
        heads_out = action_heads(logits, (("buttons", 0, 1, nb), ("camera", nb, 1, nc)), bsz, t, temp, mask, sample, keep_head_logps=tail, rng_state=rng)
        out.update(heads_out)
        if tail:
            lps = out.pop("_head_logps")
            keep, flag = ops.act_epilogue(out["action"]["buttons"].view(-1), out["action"]["camera"].view(-1), lps["buttons"].reshape(-1),
                                          lps["camera"].reshape(-1), logits, nb + nc, act_tail[0], act_tail[1], rng_state=rng)   # (advances rng's step)
            out["_keep"], out["nan_flag"] = keep, flag
            out.update(unpack_act_tail(keep, bsz))
        else:
            out["vpred"] = logits[:, nb + nc:nb + nc + 1].reshape(bsz, t, 1).clone()
            if rng is not None:
                rng[1:].add_(1)         # both heads have drawn: next call, next counter
        return out


class IDMEngine(PolicyEngine):
    """InverseActionNet.forward (lib/policy.py:374-392) on the same kernels: temporal Conv3d + ReLU, IMPALA CNN
    with a normed first conv, transformer blocks with mask "none" and no memory, ReLU, final_ln (the reference
    computes `lastlayer` and discards it, lib/policy.py:390-391 -- so it is not computed here), two heads."""

    def __init__(self, cfg: dict, button_shape, camera_shape, cnn_chunk: int = 128, precision: Optional[str] = None):
        check_supported(cfg, idm=True)
        self.precision = resolve_precision(precision)
        self.dtype = PRECISIONS[self.precision]
        self.cfg = cfg
        self.button_shape, self.camera_shape = tuple(button_shape), tuple(camera_shape)  # (20, 2), (2, 11)
        self.cnn_chunk = cnn_chunk
        self.cnn_streams = 1
        self.pool_subchunk = 0      # (PolicyEngine._cnn_chunk's option; the IDM's chunks are one 128-frame window)
        self.fuse_pool = os.environ.get("VPT_FUSE_POOL", "1") != "0"
        self.fuse_pool_sub = 0
        self.fold_n = os.environ.get("VPT_FOLD_N", "1") != "0"
        self.fold_stats_in_producer = os.environ.get("VPT_FOLD_STATS", "1") != "0"
        self.fold_dense = os.environ.get("VPT_FOLD_DENSE", "1") != "0"
        # split-K of the trunk linears, named here (a function of the layer's (N, K) only, ops.nk_splitk) -- never chosen from the row count
        self.linear_splitk = "nk"
        self._streams = []
        self._rng_state = None
        self._rng_src, self._pending_seed = None, None
        self.w = {}
        self.packed = False

    @torch.no_grad()
    def pack(self, sd):
        cfg, w = self.cfg, {}
        f32 = lambda t: t.detach().float().contiguous()
        w["conv3d"] = ops.pack_conv3d_t5(f32(sd["net.conv3d_layer.layer.weight"]), f32(sd["net.conv3d_layer.layer.bias"]), dtype=self.dtype)
        self.c3d_out = sd["net.conv3d_layer.layer.weight"].shape[0]
        for s, c in enumerate(cfg["chans"]):
            p = f"net.img_process.cnn.stacks.{s}."
            w[p + "firstconv"] = ops.pack_conv3x3(f32(sd[p + "firstconv.layer.weight"]), f32(sd[p + "firstconv.norm.weight"]), f32(sd[p + "firstconv.norm.bias"]), dtype=self.dtype)
            w[p + "n.g"], w[p + "n.b"] = f32(sd[p + "n.weight"]), f32(sd[p + "n.bias"])
            for b in range(2):
                for cv in range(2):
                    q = f"{p}blocks.{b}.conv{cv}"
                    w[q] = ops.pack_conv3x3(f32(sd[q + ".layer.weight"]), f32(sd[q + ".norm.weight"]), f32(sd[q + ".norm.bias"]), dtype=self.dtype)
        c2 = cfg["chans"][-1]
        p = "net.img_process.cnn.dense."
        w[p + "g"] = ops.chw_to_blocked(f32(sd[p + "norm.weight"]), c2, 16, 16)
        w[p + "b"] = ops.chw_to_blocked(f32(sd[p + "norm.bias"]), c2, 16, 16)
        w[p + "w"] = ops.pack_linear(ops.chw_to_blocked(f32(sd[p + "layer.weight"]), c2, 16, 16), dtype=self.dtype)
        p = "net.img_process.linear."
        w[p + "g"], w[p + "b"] = f32(sd[p + "norm.weight"]), f32(sd[p + "norm.bias"])
        w[p + "w"] = ops.pack_linear(f32(sd[p + "layer.weight"]), dtype=self.dtype)
        hid = cfg["hidsize"]
        for l in range(cfg["n_layers"]):
            p = f"net.recurrent_layer.blocks.{l}."
            o = p + "r.orc_block."
            w[p + "ln1.g"], w[p + "ln1.b"] = f32(sd[p + "pre_r_ln.weight"]), f32(sd[p + "pre_r_ln.bias"])
            wq = torch.cat([f32(sd[o + "q_layer.weight"]), f32(sd[o + "k_layer.weight"]), f32(sd[o + "v_layer.weight"])], dim=0)
            bq = torch.cat([f32(sd[o + "q_layer.bias"]), torch.zeros(2 * hid, device=wq.device)])
            w[p + "qkv.w"], w[p + "qkv.b"] = ops.pack_linear(wq, dtype=self.dtype), bq.contiguous()
            w[p + "proj.w"], w[p + "proj.b"] = ops.pack_linear(f32(sd[o + "proj_layer.weight"]), dtype=self.dtype), f32(sd[o + "proj_layer.bias"])
            w[p + "ln2.g"], w[p + "ln2.b"] = f32(sd[p + "mlp0.norm.weight"]), f32(sd[p + "mlp0.norm.bias"])
            w[p + "mlp0.w"] = ops.pack_linear(f32(sd[p + "mlp0.layer.weight"]), dtype=self.dtype)
            w[p + "mlp1.w"], w[p + "mlp1.b"] = ops.pack_linear(f32(sd[p + "mlp1.layer.weight"]), dtype=self.dtype), f32(sd[p + "mlp1.layer.bias"])
        if cfg["use_pre_lstm_ln"]:
            w["prelstm.g"], w["prelstm.b"] = f32(sd["net.pre_lstm_ln.weight"]), f32(sd["net.pre_lstm_ln.bias"])
        w["final.g"], w["final.b"] = f32(sd["net.final_ln.weight"]), f32(sd["net.final_ln.bias"])
        for h in ("buttons", "camera"):
            w[h + ".w"] = ops.pack_linear(f32(sd[f"pi_head.{h}.linear_layer.weight"]), dtype=self.dtype)
            w[h + ".b"] = f32(sd[f"pi_head.{h}.linear_layer.bias"])
        self.w = w
        self.packed = True
        self._fold_src = {s: (sd[f"net.img_process.cnn.stacks.{s}.blocks.0.conv0.layer.weight"], sd[f"net.img_process.cnn.stacks.{s}.blocks.0.conv0.norm.weight"])
                          for s in range(len(cfg["chans"]))}
        self._dense_src = sd["net.img_process.cnn.dense.layer.weight"]
        self._fold_tab = {}

    @torch.no_grad()
    def forward(self, img_u8: torch.Tensor, mask: Optional[dict] = None, sample: Optional[str] = None):
        if not self.packed:
            raise RuntimeError("IDMEngine.pack(state_dict) must be called before forward")
        cfg, w = self.cfg, self.w
        bsz, t = img_u8.shape[:2]
        if t > 160:
            raise NotImplementedError("the mask='none' attention kernel handles chunks of at most 160 frames")
        hid, heads = cfg["hidsize"], cfg["heads"]
        sk = self.linear_splitk      # "nk": K cut as a function of the layer alone (one <= 160-frame window = one row tile of the MFMA GEMM)
        frames = img_u8.reshape(bsz * t, *img_u8.shape[2:]).contiguous()
        # temporal conv needs whole sequences; the CNN behind it runs in frame chunks to bound the 4 MB/frame tensor
        wfrag, bias = w["conv3d"]
        outs = []
        step = max(1, self.cnn_chunk // t) * t if t <= self.cnn_chunk else t
        for i in range(0, bsz * t, step):
            fr = frames[i:i + step]
            s0 = torch.zeros(fr.shape[0], 2, dtype=torch.float64, device=fr.device)
            x0 = ops.conv3d_t5(fr, wfrag, bias, self.c3d_out, t, stats_out=s0)
            outs.append(self._cnn_dense(None, x0=x0, s_x0=s0))
            del x0
        d = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        p = "net.img_process.linear."
        _, dn = ops.layernorm(d, w[p + "g"], w[p + "b"], relu_in=True, dtype=self.dtype)
        x, _ = ops.linear(dn, w[p + "w"], hid, relu=True, tiling="throughput", splitk=sk)
        if cfg["use_pre_lstm_ln"]:
            x, _ = ops.layernorm(x, w["prelstm.g"], w["prelstm.b"], out_f32=True, out_bf16=False, dtype=self.dtype)
        for l in range(cfg["n_layers"]):
            p = f"net.recurrent_layer.blocks.{l}."
            x1, x1b = ops.layernorm(x, w[p + "ln1.g"], w[p + "ln1.b"], out_f32=True, dtype=self.dtype)
            qkv, _ = ops.linear(x1b, w[p + "qkv.w"], 3 * hid, bias=w[p + "qkv.b"], tiling="throughput", splitk=sk)
            att = ops.full_attention(qkv, bsz, t, heads, hid, dtype=self.dtype)
            x2, _ = ops.linear(att, w[p + "proj.w"], hid, bias=w[p + "proj.b"], res=x1, tiling="throughput", splitk=sk)
            _, hb = ops.layernorm(x2, w[p + "ln2.g"], w[p + "ln2.b"], dtype=self.dtype)
            _, h2 = ops.linear(hb, w[p + "mlp0.w"], hid * cfg["pointwise_ratio"], relu=True, out_f32=False, out_bf16=True, tiling="throughput", splitk=sk)
            x, _ = ops.linear(h2, w[p + "mlp1.w"], hid, bias=w[p + "mlp1.b"], res=x2, tiling="throughput", splitk=sk)
        latent, lb = ops.layernorm(x, w["final.g"], w["final.b"], relu_in=True, out_f32=True, dtype=self.dtype)
        out = {}
        temp = cfg["temperature"]
        logp = None
        rng = self.rng_state(x.device) if sample == "stochastic" else None
        for h, shape in (("buttons", self.button_shape), ("camera", self.camera_shape)):
            n_groups, n = shape
            z, _ = ops.linear(lb, w[h + ".w"], n_groups * n, bias=w[h + ".b"], tiling="throughput", splitk=sk)
            r = action_heads(z, ((h, 0, n_groups, n),), bsz, t, temp, mask, sample, rng_state=rng)
            out[h] = r[h]
            if sample is not None:
                out.setdefault("action", {})[h] = r["action"][h]
                logp = r["action_log_prob"] if logp is None else logp + r["action_log_prob"]
        if sample is not None:
            out["action_log_prob"] = logp
        if rng is not None:
            rng[1:].add_(1)
        out["latent"] = latent.view(bsz, t, hid)
        return out
