"""MinecraftAgentPolicy on MI355X: the reference's policy API (lib/policy.py:227-339) over the HIP engine.

Drop-in surface: same constructor arguments, `nn.Module` protocol, `state_dict()` key names
and shapes (so `.weights` files load with `load_state_dict(..., strict=False)` exactly as agent.py:134 does),
`initial_state`, `forward`, `act`, `get_output_for_observation`, `get_logprob_of_action`,
`get_kl_of_action_dists`, `v`.  The parameter tree below mirrors the reference's module tree only as a
*container* (fp32 masters); all arithmetic of `forward` runs in libvpt_hip.so through engine.PolicyEngine.
There is no CPU path: tensors must live on the GPU and the native library must be present.
"""
import math
from typing import Dict, Optional

import torch
from torch import nn

from ..engine import IDMEngine, PolicyEngine, config_from_policy_kwargs
from .action_head import make_action_head
from .tree_util import tree_map


class _Node(nn.Module):
    """Bare container; children and parameters are attached by name so state_dict keys match the reference."""


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, requires_grad: bool = True):
    parts = dotted.split(".")
    mod = root
    for name in parts[:-1]:
        if name not in mod._modules:
            mod.add_module(name, _Node())
        mod = mod._modules[name]
    mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=requires_grad))


def _fan_in_(w: torch.Tensor, scale: float):
    """FanInInitReLULayer init: each output unit's weight vector gets L2 norm `scale` (lib/util.py:67-70)."""
    flat = w.view(w.shape[0], -1)
    flat.mul_(scale / flat.norm(dim=1, p=2, keepdim=True))
    return w


class MinecraftPolicy(nn.Module):
    """Parameter container with the key names of lib/policy.py:83-224 (transformer recurrence only)."""

    def __init__(self, recurrence_type="lstm", impala_width=1, impala_chans=(16, 32, 32), hidsize=512,
                 img_shape=None, init_norm_kwargs=None, impala_kwargs=None, attention_mask_style="clipped_causal",
                 attention_heads=8, attention_memory_size=2048, pointwise_ratio=4, n_recurrence_layers=1,
                 timesteps=None, use_pre_lstm_ln=True, first_conv_norm=False, first_conv_inchan=3, **unused_kwargs):
        super().__init__()
        if recurrence_type != "transformer":
            raise NotImplementedError("only recurrence_type='transformer' (every released VPT model) is implemented")
        init_norm_kwargs = init_norm_kwargs or {}
        if init_norm_kwargs.get("group_norm_groups", None) != 1 or init_norm_kwargs.get("batch_norm", False):
            raise NotImplementedError("only init_norm_kwargs={'group_norm_groups': 1} (every released model) is implemented")
        if (impala_kwargs or {}).get("post_pool_groups", None) != 1:
            raise NotImplementedError("only impala_kwargs={'post_pool_groups': 1} is implemented")
        self.hidsize = hidsize
        chans = [int(impala_width * c) for c in impala_chans]
        maxlen = attention_memory_size - timesteps
        nblk_scale = math.sqrt(math.sqrt(len(chans)) / math.sqrt(2))  # lib/impala_cnn.py:33,106,169
        cin = first_conv_inchan
        for s, c in enumerate(chans):
            p = f"img_process.cnn.stacks.{s}."
            if s > 0 or first_conv_norm:
                _attach(self, p + "firstconv.norm.weight", torch.ones(cin))
                _attach(self, p + "firstconv.norm.bias", torch.zeros(cin))
            _attach(self, p + "firstconv.layer.weight", _fan_in_(torch.randn(c, cin, 3, 3), 1.0))
            if s == 0 and not first_conv_norm:
                _attach(self, p + "firstconv.layer.bias", torch.zeros(c))
            _attach(self, p + "n.weight", torch.ones(c))
            _attach(self, p + "n.bias", torch.zeros(c))
            for b in range(2):
                for cv in range(2):
                    q = f"{p}blocks.{b}.conv{cv}."
                    _attach(self, q + "norm.weight", torch.ones(c))
                    _attach(self, q + "norm.bias", torch.zeros(c))
                    _attach(self, q + "layer.weight", _fan_in_(torch.randn(c, c, 3, 3), nblk_scale))
            cin = c
        flat = chans[-1] * 16 * 16
        _attach(self, "img_process.cnn.dense.norm.weight", torch.ones(flat))
        _attach(self, "img_process.cnn.dense.norm.bias", torch.zeros(flat))
        _attach(self, "img_process.cnn.dense.layer.weight", _fan_in_(torch.randn(256, flat), 1.4))
        _attach(self, "img_process.linear.norm.weight", torch.ones(256))
        _attach(self, "img_process.linear.norm.bias", torch.zeros(256))
        _attach(self, "img_process.linear.layer.weight", _fan_in_(torch.randn(hidsize, 256), 1.0))
        if use_pre_lstm_ln:
            _attach(self, "pre_lstm_ln.weight", torch.ones(hidsize))
            _attach(self, "pre_lstm_ln.bias", torch.zeros(hidsize))
        s_blk = (n_recurrence_layers ** -0.5) * (2 ** -0.5)  # lib/util.py:107,150-151
        for l in range(n_recurrence_layers):
            p = f"recurrent_layer.blocks.{l}."
            _attach(self, p + "mlp0.norm.weight", torch.ones(hidsize))
            _attach(self, p + "mlp0.norm.bias", torch.zeros(hidsize))
            _attach(self, p + "mlp0.layer.weight", _fan_in_(torch.randn(hidsize * pointwise_ratio, hidsize), 1.0))
            _attach(self, p + "mlp1.layer.weight", _fan_in_(torch.randn(hidsize, hidsize * pointwise_ratio), s_blk))
            _attach(self, p + "mlp1.layer.bias", torch.zeros(hidsize))
            _attach(self, p + "pre_r_ln.weight", torch.ones(hidsize))
            _attach(self, p + "pre_r_ln.bias", torch.zeros(hidsize))
            o = p + "r.orc_block."
            sq = math.sqrt(s_blk)  # lib/xf.py:247-254
            _attach(self, o + "b_nd", torch.randn(10, maxlen) * 0.2)
            _attach(self, o + "q_layer.weight", _fan_in_(torch.randn(hidsize, hidsize), 0.1))
            _attach(self, o + "q_layer.bias", torch.zeros(hidsize))
            _attach(self, o + "k_layer.weight", _fan_in_(torch.randn(hidsize, hidsize), 0.2))
            _attach(self, o + "v_layer.weight", _fan_in_(torch.randn(hidsize, hidsize), sq))
            _attach(self, o + "proj_layer.weight", _fan_in_(torch.randn(hidsize, hidsize), sq))
            _attach(self, o + "proj_layer.bias", torch.zeros(hidsize))
            _attach(self, o + "r_layer.weight", _fan_in_(torch.randn(10 * attention_heads, hidsize), 0.1))
            _attach(self, o + "r_layer.bias", torch.zeros(10 * attention_heads))
        _attach(self, "lastlayer.norm.weight", torch.ones(hidsize))
        _attach(self, "lastlayer.norm.bias", torch.zeros(hidsize))
        _attach(self, "lastlayer.layer.weight", _fan_in_(torch.randn(hidsize, hidsize), 1.0))
        _attach(self, "final_ln.weight", torch.ones(hidsize))
        _attach(self, "final_ln.bias", torch.zeros(hidsize))

    def output_latent_size(self):
        return self.hidsize


class NormalizeEwma(nn.Module):
    """Buffers and denormalisation of lib/normalize_ewma.py:8-31,57-60 (parameters with requires_grad=False)."""

    def __init__(self, input_shape, epsilon=1e-5):
        super().__init__()
        self.epsilon = epsilon
        self.running_mean = nn.Parameter(torch.zeros(input_shape, dtype=torch.float), requires_grad=False)
        self.running_mean_sq = nn.Parameter(torch.zeros(input_shape, dtype=torch.float), requires_grad=False)
        self.debiasing_term = nn.Parameter(torch.tensor(0.0, dtype=torch.float), requires_grad=False)

    def running_mean_var(self):
        deb = self.debiasing_term.clamp(min=self.epsilon)
        mean = self.running_mean / deb
        mean_sq = self.running_mean_sq / deb
        return mean, (mean_sq - mean ** 2).clamp(min=1e-2)

    def denormalize(self, x):
        mean, var = self.running_mean_var()
        return x * torch.sqrt(var)[(None,) * 2] + mean[(None,) * 2]

    def affine(self):
        """denormalize(x) = x * scale + shift with (scale, shift) as host floats, recomputed only when the statistics change
        (lib/normalize_ewma.py:27-31,57-60 with output size 1): the acting step applies it with one fused multiply-add inside
        its graph instead of the half-dozen small kernels of running_mean_var() per step."""
        key = (self.running_mean._version, self.running_mean_sq._version, self.debiasing_term._version,
               self.running_mean.data_ptr(), str(self.running_mean.device))
        if getattr(self, "_affine_key", None) != key:
            mean, var = self.running_mean_var()
            if mean.numel() != 1:
                raise NotImplementedError("NormalizeEwma.affine(): scalar value head only")
            self._affine, self._affine_key = (float(torch.sqrt(var).reshape(-1)[0]), float(mean.reshape(-1)[0])), key
        return self._affine


class ScaledMSEHead(nn.Module):
    """lib/scaled_mse_head.py:11-50: the Linear(hid -> 1) lives in the fused heads GEMM."""

    def __init__(self, input_size: int, output_size: int):
        super().__init__()
        self.linear = nn.Linear(input_size, output_size)
        self.normalizer = NormalizeEwma(output_size)

    def denormalize(self, x):
        return self.normalizer.denormalize(x)


class _PolicyForwardFn(torch.autograd.Function):
    """The differentiable boundary: MinecraftAgentPolicy.forward as ONE autograd node whose backward is the hand-written
    HIP backward of training.py (every layer, CNN included).  That is what lets the reference's own training loop --
    `get_output_for_observation` -> `get_logprob_of_action` -> `loss.backward()` -> `torch.optim.Adam.step()`
    (behavioural_cloning.py:99-122) -- run unchanged: autograd accumulates the returned gradients into `param.grad` exactly
    as it does for the reference's module tree.  The recurrent state is returned non-differentiable (the reference's loop
    detaches it, behavioural_cloning.py:111; gradients never cross chunk boundaries)."""

    @staticmethod
    def forward(ctx, policy, grad_engine, img, first, state_in, mask, names, *params):
        S = grad_engine.forward_saving(img, first, state_in, mask=mask)
        bsz, t = S["bsz"], S["t"]
        nb, nc = grad_engine.engine.n_buttons, grad_engine.engine.n_camera
        ctx.S, ctx.grad_engine, ctx.names = S, grad_engine, names
        ctx.param_shapes = [p.shape for p in params]
        lp_b = S["lp_b"].view(bsz, t, 1, nb)
        lp_c = S["lp_c"].view(bsz, t, 1, nc)
        vpred = S["logits"][:, nb + nc:nb + nc + 1].reshape(bsz, t, 1).clone()
        flat = []
        for m, (k, v) in S["state_out"]:
            flat += [m, k, v]
        ctx.mark_non_differentiable(*flat)
        ctx.set_materialize_grads(False)     # an output the loss does not use arrives as None: no value-head gradient under the BC loss
        return (lp_b, lp_c, vpred, *flat)

    @staticmethod
    def backward(ctx, g_b, g_c, g_v, *g_state):
        from .. import ops
        S, eng = ctx.S, ctx.grad_engine
        if S is None:
            raise RuntimeError("the policy's activations were already consumed by a backward pass (retain_graph is not supported)")
        ctx.S = None
        m = S["m"]
        flat2 = lambda g_, n_: None if g_ is None else g_.reshape(m, n_).float().contiguous()
        nb, nc = eng.engine.n_buttons, eng.engine.n_camera
        gv = None if g_v is None else g_v.reshape(m).float().contiguous()
        with torch.no_grad():
            gb, gc = flat2(g_b, nb), flat2(g_c, nc)
            scale = 1.0
            if eng.scaled:
                # fp16 operands: the 16-bit gradient buffers need the incoming gradients lifted into IEEE half's range (a mean
                # over M frames arrives as 1 / M per element).  Power of two, so un-scaling the fp32 results below is exact.
                # `autograd_lift` (256 to start with) is where the largest incoming element is placed; an overflow halves it.
                gmax = max([float(x.abs().max()) for x in (gb, gc, gv) if x is not None and x.numel()] or [0.0])
                if not math.isfinite(gmax):
                    # the INCOMING gradient is inf / nan (the caller's loss, not our lift): no gradient from this backward, counted, and
                    # the lift stays where it is -- halving it would not have helped
                    import warnings
                    eng.autograd_overflows += 1
                    warnings.warn("fp16 backward received a non-finite gradient: this backward returns no gradients", RuntimeWarning)
                    return (None,) * (7 + len(ctx.names))
                if gmax > 0.0:
                    scale = 2.0 ** math.floor(math.log2(eng.autograd_lift / gmax))
            dz = ops.heads_logprob_backward(S["lp_b"], S["lp_c"], gb, gc, gv, S["ldz"], eng.engine.cfg["temperature"],
                                            mask_buttons=S["mask"]["buttons"], mask_camera=S["mask"]["camera"], dtype=eng.dtype, grad_scale=scale)
            g = eng.backward_from(S, dz, value_grads=gv is not None)
            if eng.scaled:
                # GradScaler semantics at the autograd boundary: the caller's optimizer (th.optim.Adam in the reference's loop,
                # behavioural_cloning.py:117-122) must never see the inf / nan a half overflow in dz / dacc / dx16 leaves behind.
                # On overflow this backward contributes NO gradient (every input gradient None: an optimizer skips parameters
                # whose .grad is None, an accumulation loop simply misses this sample), warns, and the next backward lifts less.
                if int(ops.grads_nonfinite(list(g.values())).item()):
                    import warnings
                    eng.autograd_lift = max(1.0, eng.autograd_lift * 0.5)
                    eng.autograd_overflows += 1
                    warnings.warn(f"fp16 backward overflowed (gradient lift 2^{math.log2(scale):.0f}): this backward returns no gradients; "
                                  f"the next one lifts to {eng.autograd_lift:g}", RuntimeWarning)
                    return (None,) * (7 + len(ctx.names))
            if scale != 1.0:
                for t_ in g.values():
                    t_.mul_(1.0 / scale)
        grads = tuple(g[n].reshape(shape) if n in g else None for n, shape in zip(ctx.names, ctx.param_shapes))
        return (None, None, None, None, None, None, None) + grads


class MinecraftAgentPolicy(nn.Module):
    """`precision` (not a reference argument; also env VPT_PRECISION or set_precision()): "fp16" -- the default: the parity
    mode, IEEE-half MFMA operands -- or "bf16", the north star's "MFMA bf16 tiles" and bench.py's headline: the same kernels,
    same speed, 8x coarser operand rounding (engine.PolicyEngine / engine.resolve_precision)."""

    def __init__(self, action_space, policy_kwargs, pi_head_kwargs, precision: Optional[str] = None):
        super().__init__()
        self.net = MinecraftPolicy(**policy_kwargs)
        self.action_space = action_space
        self.value_head = ScaledMSEHead(self.net.output_latent_size(), 1)
        self.pi_head = make_action_head(self.action_space, self.net.output_latent_size(), **pi_head_kwargs)
        self._cfg = config_from_policy_kwargs(policy_kwargs, pi_head_kwargs)
        self._engine = PolicyEngine(self._cfg, n_buttons=action_space["buttons"].eltype.n,
                                    n_camera=action_space["camera"].eltype.n, precision=precision)
        self._packed_key = None
        self._param_cache = None
        self._step_graph = None
        import os
        self._auto_graph = dict(enabled=os.environ.get("VPT_STEP_GRAPH", "1") != "0", batch=None, count=0)
        self._grad_engines = {}

    @property
    def precision(self) -> str:
        return self._engine.precision

    def set_precision(self, precision: str):
        """Switch the operand format ("bf16" / "fp16"); weights are re-packed on the next forward."""
        if precision != self._engine.precision:
            self._grad_engines = {}
            old = self._engine
            self._engine = PolicyEngine(self._cfg, n_buttons=old.n_buttons, n_camera=old.n_camera, precision=precision)
            self._engine.adopt_sampler(old)         # seed_sampler() / the step counter survive a change of operand format
            self._engine.overlap_steps(old.step_overlap)
            self._packed_key = None
            if self._step_graph is not None:
                self._step_graph = self._fresh_step_graph(self._step_graph)      # (static buffers and graphs are rebuilt lazily)
        return self

    def overlap_steps(self, enable: bool = True):
        """Opt-in pipelining of consecutive multi-chunk forward() calls (a labelling / evaluation loop over resident batches): the convolutions of
        a call run beside the previous call's transformer and heads.  Same results; the CONTRACT is PolicyEngine.overlap_steps's -- the frames
        handed to forward() are already complete on the device.  Off by default (the reference's call semantics on the current stream)."""
        self._engine.overlap_steps(enable)
        return self

    @property
    def grad_overflows(self) -> int:
        """How many backward passes through this policy returned NO gradients because an IEEE-half buffer overflowed (fp16 mode) or the
        incoming gradient was non-finite.  A gradient-accumulation loop (the reference's 8 x backward then optimizer.step(),
        behavioural_cloning.py:107-122) reads it before and after its backwards and skips optimizer.step() when it moved -- what
        torch.cuda.amp.GradScaler.step does.  Always 0 in bf16."""
        return sum(e.autograd_overflows for e in self._grad_engines.values())

    # ---- engine plumbing --------------------------------------------------------------------
    def _apply(self, fn, *args, **kwargs):      # .to() / .cuda() / .half(): parameters may be re-created
        self._param_cache = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):  # assign=True replaces the Parameter objects
        self._param_cache = None
        return super().load_state_dict(*args, **kwargs)

    def _params(self):
        """(name, parameter) list, cached: named_parameters() walks the module tree (~0.1 ms, paid per acting step otherwise).
        Parameters are created in __init__ and only ever updated in place or moved by .to() (same Parameter objects)."""
        if self._param_cache is None:
            self._param_cache = list(self.named_parameters())
        return self._param_cache

    def _device(self):
        return self._params()[0][1].device

    def _ensure_packed(self):
        named = self._params()
        key = (named[0][1].device,) + tuple((p.data_ptr(), p._version) for _, p in named)
        if key != self._packed_key:
            if self._device().type != "cuda":
                raise RuntimeError("MinecraftAgentPolicy (HIP) needs its parameters on the GPU: call .to('cuda')")
            self._engine.pack(dict(named))
            self._packed_key = key
            if self._step_graph is not None:      # the graph holds the old packed weights' addresses: re-capture lazily
                self._step_graph = self._fresh_step_graph(self._step_graph)

    # ---- T = 1 acting path: one hipGraph replay per environment step ---------------------------
    # The reference's entry point is MineRLAgent.get_action -> policy.act(agent_input, first, hidden_state, stochastic=True)
    # (agent.py:190-206): one frame, one environment, every 50 ms.  act() therefore captures that step into a hipGraph BY ITSELF
    # once it has seen the same (B, T = 1) shape twice in a row -- run_agent.py needs no extra call -- with the stochastic sampling
    # inside the graph (the heads draw their uniforms in the kernel from a device-resident {seed, step}, engine.rng_state).
    AUTO_GRAPH_AFTER = 2     # consecutive same-shape act() calls that run eagerly before the step is captured

    @staticmethod
    def _fresh_step_graph(old: dict) -> dict:
        return dict(batch=old["batch"], alias=old.get("alias", False))

    def enable_step_graph(self, batch_size: int = 1, alias_state: bool = True):
        """Capture the T = 1 forward for `batch_size` environments into a hipGraph at the next such call (the ~50 kernel launches of
        one agent step, agent.py:190-206, become one graph launch; one graph per sampling mode, captured on first use).  act() does
        this on its own after AUTO_GRAPH_AFTER same-shape calls; forward() / v() only after this explicit call.  The recurrent state
        lives in static buffers that the graph updates in place.
        alias_state=True (this explicit call's default): the `state_out` of a graphed step ALIASES those buffers and is overwritten by
        the next step -- right for a single acting loop that only ever keeps the latest state (agent.py:201-205), WRONG for two
        environments sharing one policy object or for a caller that keeps an earlier state for rollback; clone to keep a snapshot.
        alias_state=False (what act()'s AUTOMATIC capture uses): every step returns a fresh copy of the state, as the reference does
        (one flat 8 MB copy per step on the 2x model); a state_in that is not the copy handed out by the latest step -- another
        environment's, an older snapshot, initial_state() -- is copied into the static buffers first.
        Other (B, T) shapes keep using eager launches."""
        self._step_graph = dict(batch=int(batch_size), alias=bool(alias_state))
        self._auto_graph = dict(enabled=self._auto_graph["enabled"], batch=None, count=0)

    def disable_step_graph(self):
        """Back to eager launches, and no automatic capture either (env VPT_STEP_GRAPH=0 does the same for a whole process;
        auto_step_graph(True) turns the automatic capture back on)."""
        self._step_graph = None
        self._auto_graph = dict(enabled=False, batch=None, count=0)

    def auto_step_graph(self, enabled: bool = True):
        """Switch act()'s automatic capture of the acting step on / off without touching a graph that is already in use."""
        self._auto_graph = dict(enabled=bool(enabled), batch=None, count=0)

    def seed_sampler(self, seed: int):
        """Re-seed the in-kernel generator behind act(stochastic=True) / predict(deterministic=False).  Default: the sampler follows the
        device's torch generator -- torch.manual_seed() at ANY time restarts the draws reproducibly, as it does for the reference's
        th.rand_like (engine.PolicyEngine.rng_state)."""
        self._engine.seed(seed)

    def _auto_graph_tick(self, batch: int):
        """act() saw an eligible (B, T = 1) call while no graph is enabled: count it; the call after AUTO_GRAPH_AFTER of them in a
        row with the same B turns the graph on."""
        from .. import ops
        ag = self._auto_graph
        if not ag["enabled"] or batch > ops.LN_LINEAR_MAX_ROWS:      # the acting path proper: a handful of environments
            return
        if ag["batch"] == batch:
            ag["count"] += 1
        else:
            ag["batch"], ag["count"] = batch, 1
        if ag["count"] > self.AUTO_GRAPH_AFTER:
            # nobody asked for aliasing: the automatically captured step hands out COPIES of the recurrent state (reference semantics)
            self._step_graph = dict(batch=batch, alias=False)

    def _state_views(self, flat: torch.Tensor, b: int):
        """The recurrent state [(mask, (K, V))] x n_layers as views of ONE flat byte buffer (K / V fp32 first, the bool masks behind
        them): a snapshot of the whole state is one copy kernel."""
        cfg = self._cfg
        kv, mk = b * cfg["maxlen"] * cfg["hidsize"] * 4, b * cfg["maxlen"]
        state, off, moff = [], 0, 2 * cfg["n_layers"] * kv
        for _ in range(cfg["n_layers"]):
            k = flat[off:off + kv].view(torch.float32).view(b, cfg["maxlen"], cfg["hidsize"])
            v = flat[off + kv:off + 2 * kv].view(torch.float32).view(b, cfg["maxlen"], cfg["hidsize"])
            m = flat[moff:moff + mk].view(torch.bool).view(b, 1, cfg["maxlen"])
            state.append((m, (k, v)))
            off += 2 * kv
            moff += mk
        return state

    def _static_step_buffers(self):
        sg, dev, cfg = self._step_graph, self._device(), self._cfg
        if "img" not in sg:
            b = sg["batch"]
            sg["img"] = torch.zeros(b, 1, *cfg["img_shape"], dtype=torch.uint8, device=dev)
            sg["first"] = torch.zeros(b, 1, dtype=torch.bool, device=dev)
            nbytes = cfg["n_layers"] * (2 * b * cfg["maxlen"] * cfg["hidsize"] * 4 + b * cfg["maxlen"])
            sg["flat"] = torch.zeros((nbytes + 15) // 16 * 16, dtype=torch.uint8, device=dev)
            sg["state"] = self._state_views(sg["flat"], b)
            sg["last_copy"] = None      # (alias=False) the flat copy handed out by the latest step, and its version counter
            sg["graphs"] = {}
        return sg

    def _capture_step_graph(self, mode: str):
        """mode: "deterministic" or "stochastic" -- the head kernel's sampling rule is part of the captured launch arguments, so each
        mode is its own graph over the SAME static inputs and recurrent state."""
        sg, eng, cfg = self._static_step_buffers(), self._engine, self._cfg
        from .. import ops
        inplace = cfg["maxlen"] <= ops.ATTENTION_STEP_MAXLEN     # the fused step kernel's limit; longer memories go through copies (below)
        scale, shift = self.value_head.normalizer.affine()
        # the static state may hold a LIVE episode (the other mode's graph, or the eager steps before an automatic capture, copied in by
        # _graphed_forward): the warm-up and the capture itself execute nothing / must change nothing the caller can see
        snap = [(m.clone(), (k.clone(), v.clone())) for m, (k, v) in sg["state"]]
        rng = eng.rng_state(self._device()) if mode == "stochastic" else None      # (a deterministic capture neither creates nor touches the sampler)
        rng_snap = rng.clone() if rng is not None else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm-up outside capture, with the captured call's exact arguments (lazy kernel loading, allocator
            for _ in range(2):                  # pools, the engine's arrival counters); what it advances is restored below
                eng.forward(sg["img"], sg["first"], sg["state"], sample=mode, inplace_state=inplace, act_tail=(scale, shift))
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            # sampling (arg-max, or Gumbel-max on uniforms the head kernel draws from the device-resident {seed, step}) + its log-prob ride
            # in the graph; the recurrent state -- K, V and masks -- is updated in place (no copies back into the static buffers).  act()'s
            # glue rides in the graph too, as ONE launch (ops.act_epilogue): the heads' log-probs summed, the value de-normalised, the NaN
            # check of the action log-prob (lib/policy.py:320-321 asserts it every step) as a flag the host reads with the action, the
            # sampler's step counter advanced, and everything the caller keeps beyond the next replay packed into one record: a single
            # clone per step
            out = eng.forward(sg["img"], sg["first"], sg["state"], sample=mode, inplace_state=inplace, act_tail=(scale, shift))
            for (m_in, (k_in, v_in)), (m_out, (k_out, v_out)) in zip(sg["state"], out["state_out"]):
                if m_out.data_ptr() != m_in.data_ptr():
                    m_in.copy_(m_out)
                if k_out.data_ptr() != k_in.data_ptr():
                    k_in.copy_(k_out); v_in.copy_(v_out)
        for (m_in, (k_in, v_in)), (m_s, (k_s, v_s)) in zip(sg["state"], snap):    # the warm-up runs advanced the static state and the sampler
            m_in.copy_(m_s); k_in.copy_(k_s); v_in.copy_(v_s)
        if rng is not None:
            rng.copy_(rng_snap)
        sg["graphs"][mode] = (graph, out, (scale, shift))

    def _graphed_forward(self, img, first, state_in, mode: str):
        sg = self._static_step_buffers()
        # the caller's state first (a capture below must see -- and preserve -- the live episode).  It is already IN the static buffers
        # only if it is (alias) the static state itself, or (copies) exactly the copy the latest step handed out, unmodified since.
        last = sg["last_copy"]
        live = False
        if not sg["alias"] and last is not None:
            k0 = state_in[0][1][0]
            live = k0.data_ptr() == last[0].data_ptr() and last[0]._version == last[1] and len(state_in) == len(sg["state"])
        if not live:
            for (m_s, (k_s, v_s)), (m, (k, v)) in zip(sg["state"], state_in):
                if sg["alias"] and k.data_ptr() == k_s.data_ptr():   # the aliased state of the previous graphed step
                    continue
                k_s.copy_(k); v_s.copy_(v)
                if m is None:
                    m_s.zero_()
                else:
                    m_s.copy_(m)
        # the captured launch arguments include the value normaliser's (scale, shift): a normaliser update re-captures
        if mode in sg["graphs"] and sg["graphs"][mode][2] != self.value_head.normalizer.affine():
            del sg["graphs"][mode]
        if mode not in sg["graphs"]:
            self._capture_step_graph(mode)
        if mode == "stochastic":
            self._engine.rng_state(self._device())     # (host-side check only: a torch.manual_seed() since the last step re-derives the sampler's seed in place)
        sg["img"].copy_(img)
        sg["first"].copy_(first)
        graph, gout, _ = sg["graphs"][mode]
        graph.replay()
        out = dict(gout)
        if sg["alias"]:
            out["state_out"] = sg["state"]
        else:       # reference semantics: a fresh state every step (ONE copy kernel over the flat buffer)
            c = sg["flat"].clone()
            out["state_out"] = self._state_views(c, sg["batch"])
            sg["last_copy"] = (c, c._version)
        if "_keep" in out:       # handed to the caller: must survive the next replay (the other outputs are consumed at once)
            from ..engine import unpack_act_tail
            out.update(unpack_act_tail(out["_keep"].clone(), sg["batch"]))
        return out

    def initial_state(self, batch_size: int):
        """List (one entry per block) of (None, (K, V)) zeros fp32 [B, maxlen, hid] (lib/masked_attention.py:153-159)."""
        dev = self._device()
        z = lambda: torch.zeros(batch_size, self._cfg["maxlen"], self._cfg["hidsize"], dtype=torch.float32, device=dev)
        return [(None, (z(), z())) for _ in range(self._cfg["n_layers"])]

    # ---- the reference API ---------------------------------------------------------------------
    def forward(self, obs, first: torch.Tensor, state_in):
        (pd, vpred, _), state_out, _ = self._run(obs, first, state_in, sample=None)
        return (pd, vpred, None), state_out

    def _run(self, obs, first, state_in, sample=None, auto_graph=False, keep_pd=True):
        """forward() plus, on request, the fused CategoricalActionHead.sample / logprob of the head kernel (act()).
        auto_graph: the caller is act() -- eligible calls count towards the automatic capture of the acting step.
        keep_pd=False: the caller drops the log-prob tensors at once (act() without return_pd) -- a graphed step then hands out its
        static output buffers instead of copies (everything else a graphed step returns, the recurrent state excepted, is a copy)."""
        if isinstance(obs, dict):
            obs = obs.copy()
            mask = obs.pop("mask", None)        # {"buttons"/"camera": bool [B,T,1,n]}: False -> LOG0 (lib/action_head.py:170-171)
        else:
            mask = None
        assert len(state_in) == self._cfg["n_layers"], (
            f"Length of state {len(state_in)} did not match length of blocks {self._cfg['n_layers']}")
        self._ensure_packed()
        img = obs["img"]
        if img.dtype != torch.uint8:
            raise TypeError("obs['img'] must be uint8 [B,T,128,128,3] (the /255 is fused into the first conv)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # gradient-enabled call, as the reference's forward is: the outputs join the autograd graph (both operand formats;
            # fp16 scales the incoming gradients into half's range inside the node, _PolicyForwardFn.backward)
            pd_v, state_out = self._forward_differentiable(img, first, state_in, mask)
            return pd_v, state_out, {}
        graphable = mask is None and img.shape[1] == 1
        if auto_graph and sample is not None and self._step_graph is None and graphable:
            self._auto_graph_tick(img.shape[0])
        sg = self._step_graph
        if sg is not None and graphable and img.shape[0] == sg["batch"]:
            # (a plain forward() on the graphed shape replays the deterministic graph and ignores its action outputs)
            out = self._graphed_forward(img, first, state_in, sample or "deterministic")
            if sample is None:
                out = {k: v for k, v in out.items() if k not in ("action", "action_log_prob", "vpred_denorm", "nan_flag")}
            if keep_pd:      # the next replay overwrites the graph's output buffers: what the caller may keep is copied
                out["camera"], out["buttons"] = out["camera"].clone(), out["buttons"].clone()
        else:
            out = self._engine.forward(img, first, state_in, mask=mask, sample=sample)
        pi_logits = {"camera": out["camera"], "buttons": out["buttons"]}
        extra = {k: out[k] for k in ("action", "action_log_prob", "vpred_denorm", "nan_flag") if k in out}
        return (pi_logits, out["vpred"], None), out["state_out"], extra

    def _forward_differentiable(self, img, first, state_in, mask=None):
        """Gradient-enabled call (the reference's get_output_for_observation is, lib/policy.py:287-305): same kernels, every
        activation kept for the backward, outputs attached to the autograd graph through _PolicyForwardFn."""
        from ..training import BCTrainer
        named = [(n, p) for n, p in self.named_parameters()]
        train_cnn = any(p.requires_grad for n, p in named if n.startswith("net.img_process.cnn."))
        eng = self._grad_engines.get(train_cnn)
        if eng is None:
            eng = self._grad_engines[train_cnn] = BCTrainer(self, train_cnn=train_cnn, optimizer_state=False)
        names = [n for n, _ in named]
        out = _PolicyForwardFn.apply(self, eng, img, first, state_in, mask, names, *[p for _, p in named])
        lp_b, lp_c, vpred = out[:3]
        flat = out[3:]
        state_out = [(flat[3 * l], (flat[3 * l + 1], flat[3 * l + 2])) for l in range(self._cfg["n_layers"])]
        return ({"camera": lp_c, "buttons": lp_b}, vpred, None), state_out

    def get_logprob_of_action(self, pd, action):
        ac = tree_map(lambda x: x.unsqueeze(1), action)
        log_prob = self.pi_head.logprob(ac, pd)
        assert not torch.isnan(log_prob).any()
        return log_prob[:, 0]

    def get_kl_of_action_dists(self, pd1, pd2):
        return self.pi_head.kl_divergence(pd1, pd2)

    def get_output_for_observation(self, obs, state_in, first):
        obs = tree_map(lambda x: x.unsqueeze(1), obs)
        first = first.unsqueeze(1)
        (pd, vpred, _), state_out = self(obs=obs, first=first, state_in=state_in)
        return pd, self.value_head.denormalize(vpred)[:, 0], state_out

    @torch.no_grad()
    def act(self, obs, first, state_in, stochastic: bool = True, taken_action=None, return_pd=False):
        obs = tree_map(lambda x: x.unsqueeze(1), obs)
        first = first.unsqueeze(1)
        want = None if taken_action is not None else ("stochastic" if stochastic else "deterministic")
        (pd, vpred, _), state_out, extra = self._run(obs, first, state_in, sample=want, auto_graph=True, keep_pd=return_pd or taken_action is not None)
        if taken_action is None and "action" in extra:
            # CategoricalActionHead.sample / logprob (lib/action_head.py:176-207) came out of the head kernel
            ac = {k: extra["action"][k] for k in pd}
            log_prob = extra["action_log_prob"]
        else:
            if taken_action is None:
                ac = self.pi_head.sample(pd, deterministic=not stochastic)
            else:
                ac = tree_map(lambda x: x.unsqueeze(1), taken_action)
            log_prob = self.pi_head.logprob(ac, pd)
        if "nan_flag" in extra and taken_action is None and "action" in extra:
            assert not bool(extra["nan_flag"])            # computed inside the step graph
            vp = extra["vpred_denorm"]
        else:
            assert not torch.isnan(log_prob).any()
            vp = self.value_head.denormalize(vpred)[:, 0]
        result = {"log_prob": log_prob[:, 0], "vpred": vp}
        if return_pd:
            result["pd"] = tree_map(lambda x: x[:, 0], pd)
        ac = tree_map(lambda x: x[:, 0], ac)
        return ac, state_out, result

    @torch.no_grad()
    def v(self, obs, first, state_in):
        obs = tree_map(lambda x: x.unsqueeze(1), obs)
        first = first.unsqueeze(1)
        (pd, vpred, _), state_out = self(obs=obs, first=first, state_in=state_in)
        return self.value_head.denormalize(vpred)[:, 0]


class InverseActionNet(MinecraftPolicy):
    """Parameter container of lib/policy.py:342-372: MinecraftPolicy with a normed first conv plus the temporal
    Conv3d layer in front (`conv3d_layer.layer.{weight,bias}`)."""

    def __init__(self, hidsize=512, conv3d_params=None, **kwargs):
        if conv3d_params is None:
            raise NotImplementedError("the IDM without conv3d_params is not a released configuration")
        if list(conv3d_params.get("kernel_size", [])) != [5, 1, 1] or list(conv3d_params.get("padding", [])) != [2, 0, 0] \
                or conv3d_params.get("inchan") != 3:
            raise NotImplementedError("only Conv3d(3 -> C, kernel (5,1,1), padding (2,0,0)) is implemented")
        kwargs.pop("first_conv_norm", None)
        super().__init__(hidsize=hidsize, first_conv_norm=True, first_conv_inchan=conv3d_params["outchan"], **kwargs)
        oc = conv3d_params["outchan"]
        _attach(self, "conv3d_layer.layer.weight", _fan_in_(torch.randn(oc, 3, 5, 1, 1), 1.0))
        _attach(self, "conv3d_layer.layer.bias", torch.zeros(oc))


class InverseActionPolicy(nn.Module):
    """lib/policy.py:406-467 over the HIP engine: same constructor, `initial_state`, `forward`, `predict`."""

    def __init__(self, action_space, pi_head_kwargs=None, idm_net_kwargs=None, precision: Optional[str] = None):
        super().__init__()
        self.action_space = action_space
        self.net = InverseActionNet(**idm_net_kwargs)
        pi_head_kwargs = {} if pi_head_kwargs is None else pi_head_kwargs
        self.pi_head = make_action_head(self.action_space, self.net.output_latent_size(), **pi_head_kwargs)
        self._cfg = config_from_policy_kwargs(idm_net_kwargs, pi_head_kwargs)
        bt, ct = action_space["buttons"], action_space["camera"]
        self._engine = IDMEngine(self._cfg, (bt.size, bt.eltype.n), (ct.size, ct.eltype.n), precision=precision)
        self._packed_key = None

    @property
    def precision(self) -> str:
        return self._engine.precision

    def set_precision(self, precision: str):
        if precision != self._engine.precision:
            old = self._engine
            self._engine = IDMEngine(self._cfg, old.button_shape, old.camera_shape, precision=precision)
            self._engine.adopt_sampler(old)
            self._packed_key = None
        return self

    def _device(self):
        return next(self.parameters()).device

    def _ensure_packed(self):
        params = dict(self.named_parameters())
        key = (str(self._device()),) + tuple((p.data_ptr(), p._version) for p in params.values())
        if key != self._packed_key:
            if self._device().type != "cuda":
                raise RuntimeError("InverseActionPolicy (HIP) needs its parameters on the GPU: call .to('cuda')")
            self._engine.pack(params)
            self._packed_key = key

    def initial_state(self, batch_size: int):
        """maxlen = 0: empty K/V memories, as the reference returns (lib/masked_attention.py:153-159)."""
        dev = self._device()
        z = lambda: torch.zeros(batch_size, 0, self._cfg["hidsize"], dtype=torch.float32, device=dev)
        return [(None, (z(), z())) for _ in range(self._cfg["n_layers"])]

    def forward(self, obs, first: torch.Tensor, state_in, **kwargs):
        (pd, _, _), state_out, _ = self._run(obs, state_in, sample=None)
        return (pd, None, None), state_out

    def _run(self, obs, state_in, sample=None):
        if isinstance(obs, dict):
            obs = obs.copy()
            mask = obs.pop("mask", None)        # {"buttons": bool [B,T,20,2], "camera": bool [B,T,2,11]}: False -> LOG0
        else:
            mask = None
        self._ensure_packed()
        img = obs["img"]
        if img.dtype != torch.uint8:
            raise TypeError("obs['img'] must be uint8 [B,T,128,128,3]")
        out = self._engine.forward(img, mask=mask, sample=sample)
        pi_logits = {"buttons": out["buttons"], "camera": out["camera"]}
        extra = {k: out[k] for k in ("action", "action_log_prob") if k in out}
        # mask "none" / maxlen 0: the state passes through empty (lib/xf.py:366-391 with cache_keep_len = 0)
        return (pi_logits, None, None), state_in, extra

    @torch.no_grad()
    def predict(self, obs, deterministic: bool = True, **kwargs):
        state_in = kwargs.get("state_in")
        (pd, _, _), state_out, extra = self._run(obs, state_in, sample="deterministic" if deterministic else "stochastic")
        ac = {k: extra["action"][k] for k in pd}          # CategoricalActionHead.sample, fused into the head kernel
        log_prob = extra["action_log_prob"]
        assert not torch.isnan(log_prob).any()
        return ac, state_out, {"log_prob": log_prob, "pd": pd}
