"""Behavioural-cloning step on MI355X (behavioural_cloning.py:86-123 generalised to [B, T] chunks).

    loss = -mean_{b,t}[ log pi(buttons_bt) + log pi(camera_bt) ]      (lib/action_head.py:176-184,252-253)
    th.optim.Adam(lr=1.81e-4, weight_decay=0.039428)                   (behavioural_cloning.py:38-40,63-67)
    gradient clipping: none -- the reference's clip_grad_norm_ receives an exhausted generator and is a no-op
                       (behavioural_cloning.py:60,63,121; SURVEY.md §2 'latent bugs'); reproduced, i.e. NOT applied
    state: the KV memory is carried between chunks detached (behavioural_cloning.py:111)

Every layer's backward runs on the HIP kernels: both action heads, final_ln, lastlayer, the four transformer blocks
(attention incl. the relative-position bias and b_nd, MLPs, LayerNorms), ImgObsProcess.linear + its LayerNorm
(vpt_gemm_kernel as dgrad / wgrad, vpt_attn_bwd_kernel, vpt_ln_bwd_kernel, vpt_nll_bwd_kernel), and -- with
train_cnn=True, the reference's behaviour (behavioural_cloning.py:57-63 trains every parameter) -- the IMPALA CNN:
dense layer GEMMs, per-element / per-channel affine backward, max-pool routing, the folded GroupNorm convolutions
(vpt_conv_bwd_prep_kernel -> vpt_conv3x3_kernel in dgrad mode + vpt_conv_wgrad_kernel -> conv_param_grads) and the
fused first conv (vpt_conv_first_bwd_kernel).  train_cnn=False freezes `net.img_process.cnn.*` and fine-tunes the
trunk and heads only (71 % of the 2x model's parameters).  Data parallelism: one process per GPU, sequences sharded by rank, the
gradients summed over RCCL in buckets -- trunk + heads while the CNN backward runs, the CNN's at the end
(BCTrainer.reduced_loss_and_grads); the 1 / global_frames factor is already in the loss gradient."""
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import distributed as D
from . import ops, packing
from .engine import DENSE_SPLITK


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def linear_backward(dy16: torch.Tensor, n: int, x16: Optional[torch.Tensor], weight: torch.Tensor, need_dx: bool = True,
                    res: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                    dx_f32: bool = True, dx_bf16_ld: Optional[int] = None, need_dw: bool = True):
    """Backward of y = x W^T on the MFMA GEMM kernel.
    dy16: bf16 [M, Np] (Np >= n, Np % 64 == 0, padding zero); x16: bf16 [M, K]; weight fp32 [n, K].
    Returns (dx fp32 or None, dx bf16 or None, dW fp32 [n, K] or None).
    dgrad:  dx = dy W      -> GEMM(A = dy, weights = W^T packed), optional ReLU gate `mask` and skip-sum `res`;
    wgrad:  dW = dy^T x    -> the TN GEMM (ops.linear_wgrad): both operands row-major over the M frames, no transposes."""
    m, np_ = dy16.shape
    k = weight.shape[1]
    dev = dy16.device
    dx32 = dx16 = dw = None
    if need_dx:   # W^T packed straight from the fp32 master by one kernel (was: zero, transpose, cast, permute -- four tensor ops)
        wt = ops.pack_linear(weight.contiguous(), transposed=True, k_pad=np_, dtype=dy16.dtype)
        dx32, dx16 = ops.linear(dy16, wt, k, res=res, mask=mask, out_f32=dx_f32,
                                out_bf16=dx_bf16_ld is not None, out_bf16_ld=dx_bf16_ld)
        del wt
    if need_dw:       # dW = dy^T x straight from the row-major activations (vpt_gemm_tn_kernel: LDS transpose reads, no copies)
        dw = ops.linear_wgrad(dy16, x16, np_)[:n]
    return dx32, dx16, dw


class BCTrainer:
    def __init__(self, policy, lr: float = 0.000181, weight_decay: float = 0.039428, betas=(0.9, 0.999), eps: float = 1e-8,
                 train_cnn: bool = True, optimizer_state: bool = True, loss_scale: Optional[float] = None,
                 scale_growth_interval: int = 200):
        """train_cnn=True (default, as the reference: behavioural_cloning.py:57-63 optimises policy.parameters(), i.e. every
        parameter) or False to freeze `net.img_process.cnn.*` and fine-tune trunk + heads only (no CNN activations kept).
        optimizer_state=False: gradients only (the autograd boundary of lib/policy.py), no Adam moments allocated.

        Both operand formats train.  precision="fp16" (the parity mode) keeps every 16-bit gradient buffer -- dz, the GEMM /
        conv operands dacc, the inter-layer dx -- inside IEEE half's range by LOSS SCALING: the loss gradient is written as
        (loss_scale / temperature) x (softmax - one-hot) per frame, i.e. the 1 / global_frames of the mean is left out as well, so
        the magnitudes in the 16-bit buffers do not depend on the batch size; vpt_adam_step_multi multiplies the fp32 gradients by
        1 / (loss_scale x global_frames).  `loss_scale` (default 256 for fp16, 1 = off for bf16) is dynamic as in
        torch.cuda.amp.GradScaler: vpt_grads_nonfinite_multi checks the (all-reduced) gradients; an overflowed step is skipped on
        the device (the Adam launch reads the flag) and the scale halves, it doubles after `scale_growth_interval` clean steps."""
        self.train_cnn = bool(train_cnn)
        self.policy = policy
        self.engine = policy._engine
        self.dtype = self.engine.dtype
        self.scaled = self.engine.precision == "fp16"
        self.loss_scale = float(loss_scale) if loss_scale is not None else (256.0 if self.scaled else 1.0)
        self.scale_growth_interval, self._clean_steps, self.skipped_steps = int(scale_growth_interval), 0, 0
        # the autograd boundary's counterpart of loss_scale (lib/policy.py:_PolicyForwardFn.backward): where the largest incoming
        # gradient element is lifted to before it enters the 16-bit buffers; halved on every overflow it detects
        self.autograd_lift, self.autograd_overflows = 256.0, 0
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.step_count = 0
        # Frame chunks of the CNN (forward-saving AND backward) alternate over this many HIP streams, chunk ci always on stream
        # ci % n: one chunk's HBM-bound passes (conv_backward_prepare, the affine backward, pool) and launch tails overlap the
        # other chunks' MFMA-bound convolutions, as in the inference engine.  Each stream accumulates its weight-gradient pieces in
        # its own buffers, merged in stream order at the end (the merge order is fixed; WITHIN a buffer the wgrad / first-conv kernels
        # add their per-workgroup pieces with fp32 atomics in arrival order, so the last bits of the stack-0 gradients vary from run to
        # run -- 1e-7 relative in bf16, up to 1e-4 on heavily cancelling fp16 sums, tests/test_gpu_training.py); a chunk's saved
        # activations are allocated, used and released on ONE stream, so the caching allocator's per-stream pools need no cross-stream
        # bookkeeping.  The streams are created ONCE per trainer and only ever appended to (never replaced): the activations a
        # forward_saving() left for a later backward stay tied to the stream object that produced them.
        self.cnn_streams = int(os.environ.get("VPT_BC_STREAMS", self.engine.cnn_streams))
        # a block's conv1 -> conv0 backward through ONE dgrad epilogue (_block_backward); 0 = the round-4 path (A/B)
        self.gated_dgrad = os.environ.get("VPT_BC_GATED_DGRAD", "1") != "0"
        # stacks 1..: firstconv + max-pool as ONE pass that records the arg-max positions (ops.conv3x3_pool_argmax), its backward from the pooled
        # tensors alone (ops.conv_backward_prepare_pooled); 0 = conv -> vpt_pool_kernel with the pre-pool tensor kept (round 4, A/B)
        self.fused_pool = os.environ.get("VPT_BC_FUSED_POOL", "1") != "0"
        # ... and with it the second pass of the GroupNorm-`n` backward of stacks 1.. folded into that consumer (needs fused_pool); 0: two passes (A/B)
        self.fold_n_backward = os.environ.get("VPT_BC_FOLD_N_BWD", "1") != "0"
        # ... and stack 0's inside the first conv's backward kernel (ops.conv_first_backward(nfold=...), round 6); 0: two passes + the plain kernel (A/B)
        self.fold_n_backward0 = os.environ.get("VPT_BC_FOLD_N_BWD0", "1") != "0"
        self.force_exchange = os.environ.get("VPT_DP_FORCE_EXCHANGE", "0") == "1"      # see reduced_loss_and_grads
        self._arenas = None      # (key, (GradArena trunk + heads, GradArena CNN)) of the data-parallel step, built on first use
        self._streams: List[torch.cuda.Stream] = []
        self.params: Dict[str, torch.nn.Parameter] = dict(policy.named_parameters())
        self.trainable = [n for n in self.params if self._is_trainable(n)]
        self.m = {n: torch.zeros_like(self.params[n], dtype=torch.float32) for n in self.trainable} if optimizer_state else {}
        self.v = {n: torch.zeros_like(self.params[n], dtype=torch.float32) for n in self.trainable} if optimizer_state else {}

    # ---- checkpoint / resume ---------------------------------------------------
    def state_dict(self) -> dict:
        """Optimizer state next to the policy's own `.weights` state_dict (behavioural_cloning.py:131-132 saves only the
        latter): Adam moments keyed by the reference's parameter names, the step count and the hyper-parameters."""
        self._need_optimizer_state()
        return dict(step=self.step_count, lr=self.lr, weight_decay=self.wd, betas=tuple(self.betas), eps=self.eps,
                    loss_scale=self.loss_scale, train_cnn=self.train_cnn, exp_avg={n: t.detach().clone() for n, t in self.m.items()},
                    exp_avg_sq={n: t.detach().clone() for n, t in self.v.items()})

    def _need_optimizer_state(self):
        if not self.m and self.trainable:
            raise RuntimeError("this BCTrainer was built with optimizer_state=False (gradients only): step / state_dict / load_state_dict need the Adam moments")

    def load_state_dict(self, sd: dict):
        self._need_optimizer_state()
        if self.scaled and "loss_scale" in sd:
            self.loss_scale = float(sd["loss_scale"])
        if set(sd["exp_avg"]) != set(self.m):
            raise KeyError(f"optimizer state does not match the trainable parameters: {sorted(set(sd['exp_avg']) ^ set(self.m))[:4]} ...")
        self.step_count = int(sd["step"])
        self.lr, self.wd, self.betas, self.eps = sd["lr"], sd["weight_decay"], tuple(sd["betas"]), sd["eps"]
        for n in self.m:
            self.m[n].copy_(sd["exp_avg"][n])
            self.v[n].copy_(sd["exp_avg_sq"][n])

    def _is_trainable(self, name: str) -> bool:
        if name.startswith("net.img_process.cnn.") and not self.train_cnn:
            return False            # train_cnn=False: fine-tune the trunk and heads only
        if name.startswith("value_head."):
            return False            # no gradient under the BC loss; normaliser buffers are not trained
        return True

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def loss_and_grads(self, img_u8, first, state_in, act_buttons, act_camera, global_frames: Optional[int] = None, debug: Optional[dict] = None,
                       on_trunk_grads=None, unscaled: bool = True):
        """Forward (saving activations) + backward.  Returns (loss of this rank's frames, grads dict, state_out).
        global_frames: number of frames in the global (all-rank) batch the mean runs over (default: local).
        unscaled=True: the gradients of the mean loss in both operand formats.  False (what step() uses): in the fp16 mode they are
        left multiplied by 1 / grad_unscale(global_frames) -- the optimiser launch folds that factor in, no extra pass.
        on_trunk_grads(g): called once the gradients of everything behind the CNN (88 % of the parameters) are final and
        before the CNN's backward starts -- the data-parallel step starts their all-reduce there."""
        if on_trunk_grads is not None and unscaled and self.scaled:
            # the callback (an asynchronous all-reduce in the data-parallel step) would see loss-SCALED gradients that this function then
            # multiplies in place while the collective may still be reading them
            raise ValueError("loss_and_grads: on_trunk_grads needs unscaled=False in the fp16 (loss-scaled) mode; apply grad_unscale() after the exchange")
        S = self.forward_saving(img_u8, first, state_in)
        m = S["m"]
        ab = act_buttons.reshape(m).to(torch.int64).contiguous()
        ac = act_camera.reshape(m).to(torch.int64).contiguous()
        loss = -(S["lp_b"].gather(1, ab[:, None]) + S["lp_c"].gather(1, ac[:, None])).mean()
        gf = global_frames or m
        # bf16: the gradient of the global mean.  fp16: loss_scale x the gradient of the SUM over frames (see __init__); grad_unscale()
        # is the factor that turns the returned gradients into those of the mean.
        scale = (self.loss_scale if self.scaled else 1.0 / gf) / self.engine.cfg["temperature"]
        dz = ops.nll_backward(S["lp_b"], S["lp_c"], ab, ac, S["ldz"], scale, dtype=self.dtype)
        g = self.backward_from(S, dz, on_trunk_grads=on_trunk_grads, debug=debug)
        if unscaled and self.scaled:
            f = self.grad_unscale(gf)
            for t_ in g.values():
                t_.mul_(f)
        return loss, g, S["state_out"]

    def grad_unscale(self, global_frames: int) -> float:
        """What loss_and_grads' gradients must be multiplied by to be d(mean loss)/d(parameter): 1 in bf16."""
        return 1.0 / (self.loss_scale * global_frames) if self.scaled else 1.0

    @torch.no_grad()
    def forward_saving(self, img_u8, first, state_in, mask: Optional[dict] = None) -> dict:
        """The policy forward with every activation the backward needs kept alive.  Returns the saved-state dict S:
        S["lp_b"] / S["lp_c"] fp32 [M, n] log-probs, S["logits"] fp32 [M, nb+nc+1] (last column: raw value), S["state_out"]."""
        pol, eng = self.policy, self.engine
        pol._ensure_packed()
        cfg, w = eng.cfg, eng.w
        bsz, t = img_u8.shape[:2]
        m = bsz * t
        hid, heads, maxlen = cfg["hidsize"], cfg["heads"], cfg["maxlen"]
        ratio = cfg["pointwise_ratio"]
        dev = img_u8.device
        nb, nc = eng.n_buttons, eng.n_camera
        frames = img_u8.reshape(m, *img_u8.shape[2:]).contiguous()

        # ---------------- forward, keeping what the backward needs ----------------
        outs, cnn_saved = [], []
        streams, main = self._chunk_streams((m + eng.cnn_chunk - 1) // eng.cnn_chunk)
        for ci, i in enumerate(range(0, m, eng.cnn_chunk)):
            with (torch.cuda.stream(streams[ci % len(streams)]) if streams else _NullCtx()):
                if self.train_cnn:
                    xn, sv = self._cnn_forward_saving(frames[i:i + eng.cnn_chunk])
                    cnn_saved.append(sv)
                else:
                    xn = eng._cnn_chunk(frames[i:i + eng.cnn_chunk])
                d32, _ = ops.linear(xn.view(xn.shape[0], -1), w["net.img_process.cnn.dense.w"], 256, splitk=DENSE_SPLITK)
                outs.append(d32)
                del xn
        for st in streams:
            main.wait_stream(st)
        d = outs[0] if len(outs) == 1 else torch.cat(outs, 0)                     # [M,256] pre-ReLU dense output
        pl = "net.img_process.linear."
        dt = self.dtype
        _, dn = ops.layernorm(d, w[pl + "g"], w[pl + "b"], relu_in=True, dtype=dt)          # 16-bit
        x, x16 = ops.linear(dn, w[pl + "w"], hid, relu=True, out_f32=True, out_bf16=True)
        x_lin16 = x16
        x_pre = None
        if cfg["use_pre_lstm_ln"]:     # MinecraftPolicy.pre_lstm_ln (lib/policy.py:202-203)
            x_pre = x
            x, _ = ops.layernorm(x_pre, w["prelstm.g"], w["prelstm.b"], out_f32=True, out_bf16=False, dtype=dt)
        not_first = ~first[:, 0].reshape(bsz, 1, 1)
        saved: List[dict] = []
        state_out = []
        for l in range(cfg["n_layers"]):
            p = f"net.recurrent_layer.blocks.{l}."
            state_mask, (kmem, vmem) = state_in[l]
            if state_mask is None:
                state_mask = torch.zeros(bsz, 1, maxlen, dtype=torch.bool, device=dev)
            memvalid = (state_mask & not_first).reshape(bsz, maxlen).to(torch.uint8).contiguous()
            kmem, vmem = kmem.contiguous(), vmem.contiguous()
            x1, x1b = ops.layernorm(x, w[p + "ln1.g"], w[p + "ln1.b"], out_f32=True, dtype=dt)
            qkvr, _ = ops.linear(x1b, w[p + "qkvr.w"], eng.n_qkvr, bias=w[p + "qkvr.b"])
            att = ops.masked_attention(qkvr, kmem, vmem, memvalid, w[p + "b_nd"], bsz, t, heads, hid, dtype=dt)
            kout, vout = ops.kv_memory_update(qkvr, kmem, vmem, bsz, t, hid)
            x2, _ = ops.linear(att, w[p + "proj.w"], hid, bias=w[p + "proj.b"], res=x1)
            _, hb = ops.layernorm(x2, w[p + "ln2.g"], w[p + "ln2.b"], dtype=dt)
            _, h2 = ops.linear(hb, w[p + "mlp0.w"], hid * ratio, relu=True, out_f32=False, out_bf16=True)
            xo, _ = ops.linear(h2, w[p + "mlp1.w"], hid, bias=w[p + "mlp1.b"], res=x2)
            saved.append(dict(x=x, x1b=x1b, qkvr=qkvr, kmem=kmem, vmem=vmem, memvalid=memvalid, att=att, x2=x2, hb=hb, h2=h2))
            new_mask = torch.cat([state_mask[:, :, t:] & not_first,
                                  torch.ones(bsz, 1, min(t, maxlen), dtype=torch.bool, device=dev)], dim=-1)
            state_out.append((new_mask, (kout, vout)))
            x = xo
        x_trunk = x
        _, xb = ops.layernorm(x_trunk, w["last.g"], w["last.b"], relu_in=True, dtype=dt)
        y, y16 = ops.linear(xb, w["last.w"], hid, relu=True, out_f32=True, out_bf16=True)
        _, lb = ops.layernorm(y, w["final.g"], w["final.b"], dtype=dt)
        logits, _ = ops.linear(lb, w["heads.w"], nb + nc + 1, bias=w["heads.b"])
        temp = cfg["temperature"]
        mk = {h: (mask[h].reshape(m, n_).to(torch.uint8).contiguous() if mask is not None and mask.get(h) is not None else None)
              for h, n_ in (("buttons", nb), ("camera", nc))}
        lp_b = ops.log_softmax_cols(logits, 0, nb, temp, mask=mk["buttons"])
        lp_c = ops.log_softmax_cols(logits, nb, nc, temp, mask=mk["camera"])
        return dict(m=m, bsz=bsz, t=t, dev=dev, d=d, dn=dn, x_lin16=x_lin16, x_pre=x_pre, saved=saved, x_trunk=x_trunk, xb=xb, y=y, y16=y16, lb=lb,
                    logits=logits, lp_b=lp_b, lp_c=lp_c, cnn_saved=cnn_saved, state_out=state_out, ldz=_round_up(nb + nc + 1, 64), mask=mk)

    @torch.no_grad()
    def backward_from(self, S: dict, dz: torch.Tensor, on_trunk_grads=None, debug: Optional[dict] = None, value_grads: bool = False):
        """Backward of everything below the head logits.  dz: bf16 [M, ldz] = d loss / d (fused head logits).
        Consumes S (the CNN activations are released chunk by chunk).  value_grads: also return the value head's gradients
        (non-zero only when dz's value column is)."""
        pol, eng = self.policy, self.engine
        cfg, w = eng.cfg, eng.w
        P = {n: p.detach() for n, p in self.params.items()}
        m, bsz, t, dev = S["m"], S["bsz"], S["t"], S["dev"]
        hid, heads, maxlen = cfg["hidsize"], cfg["heads"], cfg["maxlen"]
        ratio = cfg["pointwise_ratio"]
        nb, nc = eng.n_buttons, eng.n_camera
        d, dn, x_lin16, saved, x_trunk, xb, y, y16, lb = (S[k] for k in ("d", "dn", "x_lin16", "saved", "x_trunk", "xb", "y", "y16", "lb"))
        cnn_saved = S["cnn_saved"]
        pl = "net.img_process.linear."
        g: Dict[str, torch.Tensor] = {}
        zeros = lambda n_: torch.zeros(n_, dtype=torch.float32, device=dev)
        nh = nb + nc + 1
        # heads (fused [buttons; camera; value] GEMM): dlat, dW, db
        wh = torch.cat([P["pi_head.buttons.linear_layer.weight"], P["pi_head.camera.linear_layer.weight"],
                        P["value_head.linear.weight"]], 0)
        dlat, _, dwh = linear_backward(dz, nh, lb, wh)
        if debug is not None:
            debug['dlatent'] = dlat.clone(); debug['latent'] = lb.float().clone(); debug['y'] = y.clone()
        dbh = zeros(nh)
        ops.column_sum_(dbh, dz, nh)
        g["pi_head.buttons.linear_layer.weight"], g["pi_head.camera.linear_layer.weight"] = dwh[:nb], dwh[nb:nb + nc]
        g["pi_head.buttons.linear_layer.bias"], g["pi_head.camera.linear_layer.bias"] = dbh[:nb], dbh[nb:nb + nc]
        if value_grads:
            g["value_head.linear.weight"], g["value_head.linear.bias"] = dwh[nb + nc:nh].clone(), dbh[nb + nc:nh].clone()
        del dz, dwh
        # final_ln
        g["net.final_ln.weight"], g["net.final_ln.bias"] = zeros(hid), zeros(hid)
        dy = ops.layernorm_backward(y, P["net.final_ln.weight"], dlat, g["net.final_ln.weight"], g["net.final_ln.bias"])
        # lastlayer: y = relu(xb Wl^T); gate dy by y > 0 while casting to the GEMM operand
        dy16 = ops.gate_cast(dy, hid, mask=y16)
        dxb, _, g["net.lastlayer.layer.weight"] = linear_backward(dy16, hid, xb, P["net.lastlayer.layer.weight"])
        g["net.lastlayer.norm.weight"], g["net.lastlayer.norm.bias"] = zeros(hid), zeros(hid)
        dx = ops.layernorm_backward(x_trunk, P["net.lastlayer.norm.weight"], dxb, g["net.lastlayer.norm.weight"],
                                    g["net.lastlayer.norm.bias"], relu_in=True)
        if debug is not None:
            debug['dy'] = dy.clone(); debug['dx_trunk'] = dx.clone(); debug['x_trunk'] = x_trunk.clone()
        del dy, dy16, dxb, dlat
        # transformer blocks, last to first
        for l in reversed(range(cfg["n_layers"])):
            p = f"net.recurrent_layer.blocks.{l}."
            o = p + "r.orc_block."
            s = saved[l]
            dout16 = ops.gate_cast(dx, hid, dtype=self.dtype)
            # mlp1: out = x2 + h2 W1^T + b1
            _, dh16, g[p + "mlp1.layer.weight"] = linear_backward(dout16, hid, s["h2"], P[p + "mlp1.layer.weight"], mask=s["h2"],
                                                                  dx_f32=False, dx_bf16_ld=hid * ratio)
            g[p + "mlp1.layer.bias"] = zeros(hid)
            ops.column_sum_(g[p + "mlp1.layer.bias"], dout16, hid)
            # mlp0: h2 = relu(hb W0^T)  (the ReLU gate was applied by the mask above)
            dhb, _, g[p + "mlp0.layer.weight"] = linear_backward(dh16, hid * ratio, s["hb"], P[p + "mlp0.layer.weight"])
            g[p + "mlp0.norm.weight"], g[p + "mlp0.norm.bias"] = zeros(hid), zeros(hid)
            dx2 = ops.layernorm_backward(s["x2"], P[p + "mlp0.norm.weight"], dhb, g[p + "mlp0.norm.weight"], g[p + "mlp0.norm.bias"], dx_add=dx)
            if debug is not None:
                debug[f'dout16_{l}'] = dout16.clone(); debug[f'dh16_{l}'] = dh16.clone(); debug[f'dhb_{l}'] = dhb.clone()
            del dh16, dhb, dout16
            # proj: x2 = x1 + att Wp^T + bp
            dx2_16 = ops.gate_cast(dx2, hid, dtype=self.dtype)
            datt, _, g[o + "proj_layer.weight"] = linear_backward(dx2_16, hid, s["att"], P[o + "proj_layer.weight"])
            g[o + "proj_layer.bias"] = zeros(hid)
            ops.column_sum_(g[o + "proj_layer.bias"], dx2_16, hid)
            # attention
            g[o + "b_nd"] = torch.zeros(10, maxlen, dtype=torch.float32, device=dev)
            dqkvr = ops.masked_attention_backward(s["qkvr"], s["kmem"], s["vmem"], s["memvalid"], w[p + "b_nd"], datt,
                                                  g[o + "b_nd"], bsz, t, heads, hid)
            nq = eng.n_qkvr
            dq16 = ops.gate_cast(dqkvr, _round_up(nq, 64), dtype=self.dtype)
            wq = torch.cat([P[o + "q_layer.weight"], P[o + "k_layer.weight"], P[o + "v_layer.weight"], P[o + "r_layer.weight"]], 0)
            dx1, _, dwq = linear_backward(dq16, nq, s["x1b"], wq, res=dx2)   # dx1 = dx2 (skip) + dqkvr Wqkvr
            g[o + "q_layer.weight"], g[o + "k_layer.weight"] = dwq[:hid], dwq[hid:2 * hid]
            g[o + "v_layer.weight"], g[o + "r_layer.weight"] = dwq[2 * hid:3 * hid], dwq[3 * hid:]
            dbq = zeros(nq)
            ops.column_sum_(dbq, dq16, nq)
            g[o + "q_layer.bias"], g[o + "r_layer.bias"] = dbq[:hid], dbq[3 * hid:]
            g[p + "pre_r_ln.weight"], g[p + "pre_r_ln.bias"] = zeros(hid), zeros(hid)
            dx = ops.layernorm_backward(s["x"], P[p + "pre_r_ln.weight"], dx1, g[p + "pre_r_ln.weight"], g[p + "pre_r_ln.bias"])
            if debug is not None:
                debug[f'dx_block{l}'] = dx.clone(); debug[f'dx2_block{l}'] = dx2.clone(); debug[f'datt{l}'] = datt.clone(); debug[f'dqkvr{l}'] = dqkvr.clone()
                debug[f'dx1_block{l}'] = dx1.clone(); debug[f'dq16_{l}'] = dq16.clone(); debug[f'dx2_16_{l}'] = dx2_16.clone()
            del dx2, dx2_16, datt, dqkvr, dq16, dx1, dwq
        if cfg["use_pre_lstm_ln"]:
            g["net.pre_lstm_ln.weight"], g["net.pre_lstm_ln.bias"] = zeros(hid), zeros(hid)
            dx = ops.layernorm_backward(S["x_pre"], P["net.pre_lstm_ln.weight"], dx, g["net.pre_lstm_ln.weight"], g["net.pre_lstm_ln.bias"])
        # ImgObsProcess.linear: x = relu(dn Wlin^T)
        dx16 = ops.gate_cast(dx, hid, mask=x_lin16)
        ddn, _, g[pl + "layer.weight"] = linear_backward(dx16, hid, dn, P[pl + "layer.weight"])
        g[pl + "norm.weight"], g[pl + "norm.bias"] = zeros(256), zeros(256)
        dd = ops.layernorm_backward(d, P[pl + "norm.weight"], ddn, g[pl + "norm.weight"], g[pl + "norm.bias"], relu_in=True)
        if on_trunk_grads is not None:
            on_trunk_grads(g)
        if self.train_cnn:
            acc = self._cnn_backward_begin(P)
            n_acc = max(1, min(self.cnn_streams, len(cnn_saved)))
            accs = [acc] + [self._cnn_backward_accumulators(acc) for _ in range(n_acc - 1)]     # operands shared, accumulators per stream
            streams, main = self._chunk_streams(len(cnn_saved))                                   # (after the zero-fills above were enqueued)
            for ci, i in enumerate(range(0, m, eng.cnn_chunk)):
                with (torch.cuda.stream(streams[ci % len(streams)]) if streams else _NullCtx()):
                    self._cnn_backward_chunk(cnn_saved[ci], dd[i:i + eng.cnn_chunk].contiguous(), accs[ci % len(accs)])
                    cnn_saved[ci] = None
            for st in streams:
                main.wait_stream(st)
            self._cnn_backward_merge(accs)
            self._cnn_backward_finish(acc, P, g)
        return g

    def _chunk_streams(self, n_chunks: int):
        """-> (the side streams the CNN chunks alternate over -- [] = everything on the current stream --, the current stream);
        the side streams have been made to wait for the work enqueued so far."""
        main = torch.cuda.current_stream()
        n = min(self.cnn_streams, n_chunks)
        if n <= 1:
            return [], main
        while len(self._streams) < n:        # append only: chunk ci of an earlier forward_saving() must find ITS stream at index ci % n again
            self._streams.append(torch.cuda.Stream())
        for st in self._streams[:n]:
            st.wait_stream(main)
        return self._streams[:n], main

    def _cnn_backward_accumulators(self, acc):
        """A second set of zeroed accumulators next to `acc` (same read-only operands)."""
        z = torch.zeros_like
        other = dict(wt=acc["wt"], dense_wt=acc["dense_wt"], raw={}, n={s: (z(a), z(b)) for s, (a, b) in acc["n"].items()}, dense=None,
                     dense_dwT=z(acc["dense_dwT"]), dense_dg=z(acc["dense_dg"]), dense_db=z(acc["dense_db"]))
        return other

    def _cnn_backward_merge(self, accs):
        """Sum the per-stream accumulators into accs[0], in stream order."""
        a0 = accs[0]
        for a in accs[1:]:
            for q, r in a["raw"].items():
                if q in a0["raw"]:
                    for t0, t1 in zip(a0["raw"][q], r):
                        t0.add_(t1)
                else:
                    a0["raw"][q] = r
            for s, (dg_, db_) in a["n"].items():
                a0["n"][s][0].add_(dg_); a0["n"][s][1].add_(db_)
            for k in ("dense_dwT", "dense_dg", "dense_db"):
                a0[k].add_(a[k])
            if a.get("first") is not None:
                if a0.get("first") is None:
                    a0["first"] = a["first"]
                else:
                    a0["first"][0].add_(a["first"][0]); a0["first"][1].add_(a["first"][1])

    # ------------------------------------------------------------------------------------------
    # IMPALA CNN: forward that keeps every activation (6.2 MB / frame on the 2x model: a 64 x 128 batch is 50 GB of
    # the 288 GB HBM, so nothing is recomputed), and the backward through the folded GroupNorm convolutions.
    # ------------------------------------------------------------------------------------------
    def _cnn_forward_saving(self, img: torch.Tensor):
        """PolicyEngine._cnn_chunk without in-place reuse; returns (xn, saved)."""
        eng = self.engine
        cfg, w = eng.cfg, eng.w
        f = img.shape[0]
        st = torch.zeros(24, f, 2, dtype=torch.float64, device=img.device)
        si = 0

        def nxt():
            nonlocal si
            si += 1
            return st[si - 1]

        sv = dict(img=img, stacks=[])
        x, s_x = None, None
        for s, c in enumerate(cfg["chans"]):
            p = f"net.img_process.cnn.stacks.{s}."
            rec = dict(x_prev=x, s_prev=s_x)
            s_pool = nxt()
            if s == 0:
                pooled = ops.conv_first(img, w[p + "firstconv"], c, stats_out=s_pool)
            else:
                wpk, sa, sg = w[p + "firstconv"]
                if self.fused_pool:    # firstconv + ReLU + max-pool in one pass that also records where each maximum sits (round 5): no pre-pool tensor
                    pooled, rec["mask"] = ops.conv3x3_pool_argmax(x, wpk, sa, sg, s_x, c, stats_out=s_pool)
                else:
                    rec["pre"] = ops.conv3x3(x, wpk, sa, sg, s_x, c)
                    pooled, rec["argmax"] = ops.maxpool(rec["pre"], stats_out=s_pool, want_argmax=True)
            s_x = nxt()
            x = ops.frame_affine(pooled, w[p + "n.g"], w[p + "n.b"], s_pool, stats_out=s_x)
            rec.update(pooled=pooled, s_pool=s_pool, blocks=[])
            for b in range(2):
                wpk, sa, sg = w[f"{p}blocks.{b}.conv0"]
                s_y = nxt()
                y = ops.conv3x3(x, wpk, sa, sg, s_x, c, stats_out=s_y)
                wpk, sa, sg = w[f"{p}blocks.{b}.conv1"]
                s_n = nxt()
                xo = ops.conv3x3(y, wpk, sa, sg, s_y, c, res=x, stats_out=s_n)
                rec["blocks"].append(dict(x_in=x, s_in=s_x, y=y, s_y=s_y, x_out=xo))
                x, s_x = xo, s_n
            sv["stacks"].append(rec)
        sv["x_last"], sv["s_last"] = x, s_x
        p = "net.img_process.cnn.dense."
        return ops.frame_affine(x, w[p + "g"], w[p + "b"], s_x, per_element=True), sv

    def _conv_names(self):
        names = []
        for s in range(len(self.engine.cfg["chans"])):
            p = f"net.img_process.cnn.stacks.{s}."
            if s > 0:
                names.append(p + "firstconv")
            for b in range(2):
                for cv in range(2):
                    names.append(f"{p}blocks.{b}.conv{cv}")
        return names

    def _cnn_backward_begin(self, P):
        """Per-step operands of the CNN backward (transposed conv weights, dense W^T) and zeroed accumulators."""
        cfg = self.engine.cfg
        dev = next(iter(P.values())).device
        c2 = cfg["chans"][-1]
        acc = dict(wt={}, raw={}, n={}, dense=None)
        for q in self._conv_names():
            acc["wt"][q] = packing.pack_conv3x3_dgrad(P[q + ".layer.weight"].float(), P[q + ".norm.weight"].float(), dtype=self.dtype)
        pd = "net.img_process.cnn.dense."
        wd_blk = packing.chw_to_blocked_columns(P[pd + "layer.weight"].float(), c2, 16, 16)      # [256, K] in activation order
        acc["dense_wt"] = packing.pack_linear(wd_blk.t().contiguous(), dtype=self.dtype)                           # dgrad operand: N = K, K = 256
        k = wd_blk.shape[1]
        acc["dense_dwT"] = torch.zeros(k, 256, dtype=torch.float32, device=dev)
        acc["dense_dg"], acc["dense_db"] = torch.zeros(k, dtype=torch.float32, device=dev), torch.zeros(k, dtype=torch.float32, device=dev)
        for s, c in enumerate(cfg["chans"]):
            acc["n"][s] = (torch.zeros(c, dtype=torch.float32, device=dev), torch.zeros(c, dtype=torch.float32, device=dev))
        return acc

    def _conv_layer_backward(self, q, acc, dy, y, res, x_in, s_in, skip, need_dx=True, pool=None):
        """One GN -> conv3x3 -> ReLU (+res) layer: accumulates the raw weight-gradient pieces and returns dx (+skip).
        pool = (dpooled, argmax) when the layer feeds the stack's max-pool (dy is then None)."""
        w = self.engine.w
        _, sa, sg = w[q]
        cin = x_in.shape[1] * 32
        n = cin * x_in.shape[2] * x_in.shape[3]
        r = acc["raw"].get(q)
        if r is None:   # [dw_raw, d_sa, d_sg]: the kernels accumulate into them across the frame chunks
            r = acc["raw"][q] = [torch.zeros(y.shape[1] * 32, 9, cin, dtype=torch.float32, device=y.device), torch.zeros_like(sa), torch.zeros_like(sg)]
        dacc, coef, _, _ = ops.conv_backward_prepare(dy, y, res, s_in, sa, sg, cin, dpooled=pool[0] if pool else None,
                                                     argmax=pool[1] if pool else None, d_sa=r[1], d_sg=r[2])
        ops.conv3x3_wgrad(dacc, x_in, out=r[0])
        if not need_dx:
            return None
        return ops.conv3x3_dgrad(dacc, acc["wt"][q], cin, skip=skip, xin=x_in, coef=coef)

    def _block_backward(self, p, b, blk, acc, dx):
        """CnnBasicBlock backward (lib/impala_cnn.py:50-52: x + conv1(conv0(x))): dx w.r.t. the block output -> dx w.r.t. its input.
        gated_dgrad (round 5, default): conv1's dgrad writes conv0's backward operand directly (ops.conv3x3_dgrad_gated: conv0 has no residual,
        its output is conv1's input, so its ReLU gate and rstd scale fit into that epilogue), and conv0's per-element prepare pass (read dy, read y,
        write dacc) becomes a reduction over the operand (ops.conv_backward_reduce: one read)."""
        q1, q0 = f"{p}blocks.{b}.conv1", f"{p}blocks.{b}.conv0"
        if not self.gated_dgrad:
            dy = self._conv_layer_backward(q1, acc, dx, blk["x_out"], blk["x_in"], blk["y"], blk["s_y"], None)
            return self._conv_layer_backward(q0, acc, dy, blk["y"], None, blk["x_in"], blk["s_in"], dx)
        w = self.engine.w
        x_in, y, x_out = blk["x_in"], blk["y"], blk["x_out"]
        c_in, c_mid = x_in.shape[1] * 32, y.shape[1] * 32
        # conv1 (residual layer): prepare -> wgrad -> gated dgrad
        _, sa1, sg1 = w[q1]
        r1 = self._raw_acc(acc, q1, x_out.shape[1] * 32, c_mid, sa1, sg1)
        dacc1, coef1, _, _ = ops.conv_backward_prepare(dx, x_out, x_in, blk["s_y"], sa1, sg1, c_mid, d_sa=r1[1], d_sg=r1[2])
        ops.conv3x3_wgrad(dacc1, y, out=r1[0])
        dacc0, gate_u = ops.conv3x3_dgrad_gated(dacc1, acc["wt"][q1], c_mid, y, coef1, blk["s_in"], c_in)
        del dacc1
        # conv0 (no residual): its operand exists already; sums only
        _, sa0, sg0 = w[q0]
        r0 = self._raw_acc(acc, q0, c_mid, c_in, sa0, sg0)
        coef0, _, _ = ops.conv_backward_reduce(dacc0, gate_u, blk["s_in"], sa0, sg0, c_in, d_sa=r0[1], d_sg=r0[2])
        ops.conv3x3_wgrad(dacc0, x_in, out=r0[0])
        return ops.conv3x3_dgrad(dacc0, acc["wt"][q0], c_in, skip=dx, xin=x_in, coef=coef0)

    @staticmethod
    def _raw_acc(acc, q, cout, cin, sa, sg):
        r = acc["raw"].get(q)
        if r is None:   # [dw_raw, d_sa, d_sg]: the kernels accumulate into them across the frame chunks
            r = acc["raw"][q] = [torch.zeros(cout, 9, cin, dtype=torch.float32, device=sa.device), torch.zeros_like(sa), torch.zeros_like(sg)]
        return r

    def _cnn_backward_chunk(self, sv, dd, acc):
        """dd: fp32 [f, 256] gradient w.r.t. the dense layer's pre-activation output for this chunk's frames."""
        eng = self.engine
        cfg, w = eng.cfg, eng.w
        pd = "net.img_process.cnn.dense."
        x_last, s_last = sv["x_last"], sv["s_last"]
        f = x_last.shape[0]
        k = x_last[0].numel()
        # dense: d = xn Wd^T.   dxn = dd Wd ;  dWd^T += xn^T dd  (GEMM rows = the K activations, reduction over frames)
        dd16 = ops.gate_cast(dd, 256, dtype=self.dtype)
        _, dxn = ops.linear(dd16, acc["dense_wt"], k, out_f32=False, out_bf16=True)
        xn = ops.frame_affine(x_last, w[pd + "g"], w[pd + "b"], s_last, per_element=True)
        ops.linear_wgrad(xn.view(f, k), dd16, k, out=acc["dense_dwT"])      # dWd^T [K, 256] += xn^T dd
        del xn
        dx = ops.frame_affine_backward(x_last, dxn.view_as(x_last), w[pd + "g"], s_last, acc["dense_dg"], acc["dense_db"], per_element=True)
        del dxn
        for s in reversed(range(len(cfg["chans"]))):
            p = f"net.img_process.cnn.stacks.{s}."
            rec = sv["stacks"][s]
            for b in (1, 0):
                dx = self._block_backward(p, b, rec["blocks"][b], acc, dx)
            dgn, dbn = acc["n"][s]
            nfold = None
            if s == 0 and self.fold_n_backward0:
                # stack 0 (round 6): the reduction pass only; vpt_conv_first_bwd_kernel forms d(pooled) per element itself -- the pooled value it needs is the
                # window maximum its arg-max search finds anyway
                ab = ops.frame_affine_backward_reduce(rec["pooled"], dx, w[p + "n.g"], rec["s_pool"], dgn, dbn)
                c = cfg["chans"][0]
                acc["first"] = ops.conv_first_backward(sv["img"], w[p + "firstconv"], dx, c, out=acc.get("first"), nfold=(w[p + "n.g"], rec["s_pool"], ab))
                continue
            if "mask" in rec and self.fold_n_backward:
                # GroupNorm `n` backward: the reduction pass only; the consumer below forms d(pooled) from (G, pooled) per element itself
                nfold = (w[p + "n.g"], rec["s_pool"], ops.frame_affine_backward_reduce(rec["pooled"], dx, w[p + "n.g"], rec["s_pool"], dgn, dbn))
                dpooled = dx
            else:
                dpooled = ops.frame_affine_backward(rec["pooled"], dx, w[p + "n.g"], rec["s_pool"], dgn, dbn)
            if s == 0:
                c = cfg["chans"][0]
                acc["first"] = ops.conv_first_backward(sv["img"], w[p + "firstconv"], dpooled, c, out=acc.get("first"))
            elif "mask" in rec:
                q = p + "firstconv"
                _, sa, sg = w[q]
                x_prev = rec["x_prev"]
                c_prev = x_prev.shape[1] * 32
                r = self._raw_acc(acc, q, rec["pooled"].shape[1] * 32, c_prev, sa, sg)
                dacc, coef, _, _ = ops.conv_backward_prepare_pooled(dpooled, rec["pooled"], rec["mask"], rec["s_prev"], sa, sg, c_prev, d_sa=r[1], d_sg=r[2], nfold=nfold)
                ops.conv3x3_wgrad(dacc, x_prev, out=r[0])
                dx = ops.conv3x3_dgrad(dacc, acc["wt"][q], c_prev, xin=x_prev, coef=coef)
                del dacc
            else:
                dx = self._conv_layer_backward(p + "firstconv", acc, None, rec["pre"], None, rec["x_prev"], rec["s_prev"], None,
                                               pool=(dpooled, rec["argmax"]))
            del dpooled

    def _cnn_backward_finish(self, acc, P, g):
        cfg = self.engine.cfg
        c2 = cfg["chans"][-1]
        for q, (dw_raw, d_sa, d_sg) in acc["raw"].items():
            dW, dgain, dbias = conv_param_grads(dw_raw, d_sa, d_sg, P[q + ".layer.weight"].float(), P[q + ".norm.weight"].float(),
                                                P[q + ".norm.bias"].float())
            g[q + ".layer.weight"], g[q + ".norm.weight"], g[q + ".norm.bias"] = dW, dgain, dbias
        g["net.img_process.cnn.stacks.0.firstconv.layer.weight"] = ops.conv_first_grad_to_reference(acc["first"][0])
        g["net.img_process.cnn.stacks.0.firstconv.layer.bias"] = acc["first"][1]
        for s in range(len(cfg["chans"])):
            g[f"net.img_process.cnn.stacks.{s}.n.weight"], g[f"net.img_process.cnn.stacks.{s}.n.bias"] = acc["n"][s]
        pd = "net.img_process.cnn.dense."
        unblock = lambda v: v.view(c2 // 32, 16, 16, 32).permute(0, 3, 1, 2).reshape(-1).contiguous()
        g[pd + "norm.weight"], g[pd + "norm.bias"] = unblock(acc["dense_dg"]), unblock(acc["dense_db"])
        dwd = acc["dense_dwT"].t()                                                               # [256, K] in activation order
        g[pd + "layer.weight"] = dwd.reshape(256, c2 // 32, 16, 16, 32).permute(0, 1, 4, 2, 3).reshape(256, -1).contiguous()

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def reduced_loss_and_grads(self, img_u8, first, state_in, act_buttons, act_camera):
        """This rank's shard of the batch -> (global mean loss, gradients of the GLOBAL mean loss summed over ranks, state_out).
        (fp16 mode: the gradients stay loss-scaled, i.e. are those of the mean divided by grad_unscale(global frames); step() hands
        that factor to the optimiser launch.)
        The loss gradient already carries 1 / global_frames, so the exchange is a plain sum, in two bucketed all-reduces:
        the trunk + head gradients (final before the CNN backward starts) travel over RCCL WHILE the CNN backward -- two
        thirds of the step's compute -- runs; the CNN's own gradients (a tenth of the bytes) follow at the end.  A rank
        that fails locally (e.g. out of memory) still joins every collective with zeros and reports it in the final
        (loss, healthy-rank count) reduction, so all ranks raise together instead of blocking in an all-reduce."""
        # (self.exchange = False: the local step only -- bench.py times it beside the full step to report how much of the exchange the CNN backward hides)
        world = dist.get_world_size() if dist.is_initialized() and getattr(self, "exchange", True) else 1
        m_local = img_u8.shape[0] * img_u8.shape[1]
        self._global_frames = m_local
        # force_exchange (VPT_DP_FORCE_EXCHANGE=1): take the data-parallel path in a ONE-rank group too -- the frame-count reduction, both arenas, the early exchange
        # under the CNN backward, the late one, the health reduction -- so that the whole step can be run (and timed) over the real transport on a 1-GPU box
        force = bool(getattr(self, "force_exchange", False)) and dist.is_initialized()
        if world == 1 and not force:
            return self.loss_and_grads(img_u8, first, state_in, act_buttons, act_camera, global_frames=m_local, unscaled=False)
        dev = img_u8.device
        # Shards may differ by one sequence when B % world != 0 (distributed.shard_range): the mean runs over the TRUE global
        # frame count, and the reported loss is the frame-weighted mean of the ranks' losses.
        count = torch.tensor([float(m_local)], dtype=torch.float64, device=dev)
        dist.all_reduce(count)
        m_global = self._global_frames = int(round(float(count.item())))
        early = [n for n in self.trainable if not n.startswith("net.img_process.cnn.")]   # final before the CNN backward
        late = [n for n in self.trainable if n.startswith("net.img_process.cnn.")]
        pending, state = [], dict(early_sent=False)
        # the gradients live in two flat fp32 arenas that persist across steps (distributed.GradArena): each tensor is copied in once when it
        # is final, the collectives run in place on 64 MB slices of the arena, the optimiser reads the views -- no cat, no copy back
        akey = (tuple(early), tuple(late), tuple(tuple(self.params[n].shape) for n in early + late), str(dev))
        if self._arenas is None or self._arenas[0] != akey:      # rebuilt when the trainable set, a shape or the device changes
            self._arenas = (akey, tuple(D.GradArena(ns, [self.params[n].shape for n in ns], dev) for ns in (early, late)))
        arena_early, arena_late = self._arenas[1]

        def start_trunk_exchange(g):
            arena_early.adopt(g)
            pending.extend(arena_early.all_reduce_start(force))
            state["early_sent"] = True

        err, loss, grads, state_out = None, None, None, None
        try:
            loss, grads, state_out = self.loss_and_grads(img_u8, first, state_in, act_buttons, act_camera,
                                                         global_frames=m_global, on_trunk_grads=start_trunk_exchange, unscaled=False)
            arena_late.adopt(grads)
            pending.extend(arena_late.all_reduce_start(force))
        except Exception as e:          # e.g. out of memory on this rank: still take part in every collective (with zeros of
            err = e                     # the same shapes) so that the other ranks are not left blocked, then fail everywhere
            if not state["early_sent"]:
                arena_early.flat.zero_()
                pending.extend(arena_early.all_reduce_start(force))
            arena_late.flat.zero_()
            pending.extend(arena_late.all_reduce_start(force))
        D.bucketed_all_reduce_finish(pending)
        tail = torch.tensor([0.0 if err is not None else float(loss) * m_local, 0.0 if err is not None else 1.0], device=dev)
        dist.all_reduce(tail)           # (sum over ranks of loss x local frames, number of healthy ranks)
        if float(tail[1].item()) != world:
            raise RuntimeError(f"BC step failed on {'this' if err is not None else 'another'} rank: {err!r}")
        return tail[0] / m_global, grads, state_out

    @torch.no_grad()
    def step(self, img_u8, first, state_in, act_buttons, act_camera):
        """One optimiser step on this rank's shard of the batch.  Returns (global mean loss, state_out)."""
        self._need_optimizer_state()
        loss, grads, state_out = self.reduced_loss_and_grads(img_u8, first, state_in, act_buttons, act_camera)
        names = [n for n in self.trainable if n in grads]
        found_inf = torch.zeros(1, dtype=torch.int32, device=img_u8.device) if self.scaled else None
        # one launch for all tensors (th.optim.Adam(policy.parameters()).step(), behavioural_cloning.py:122); in the fp16 mode it
        # un-scales the gradients and is a no-op on the device when the overflow check (same table, one launch before) fired
        ops.adam_step_multi_([self.params[n].data.view(-1) for n in names], [grads[n].contiguous().view(-1) for n in names],
                             [self.m[n].view(-1) for n in names], [self.v[n].view(-1) for n in names], self.step_count + 1,
                             lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=self.wd,
                             grad_scale=self.grad_unscale(self._global_frames), found_inf=found_inf)
        loss = float(loss)                     # (synchronises: the flag below is ready)
        if self.scaled and int(found_inf.item()):
            self.skipped_steps += 1            # the ranks agree: every rank checked the same all-reduced gradients
            self._clean_steps = 0
            self.loss_scale = max(self.loss_scale * 0.5, 1.0)
            return loss, state_out
        self.step_count += 1
        if self.scaled:
            self._clean_steps += 1
            if self._clean_steps >= self.scale_growth_interval:
                self._clean_steps, self.loss_scale = 0, min(self.loss_scale * 2.0, 65536.0)
        self.policy._packed_key = None  # weights changed: re-pack before the next forward
        return loss, state_out


# ---------------------------------------------------------------------------------------------------------
# host mapping of the folded-conv gradients to the reference's parameters (small tensors: [Cout, Cin, 3, 3])
# ---------------------------------------------------------------------------------------------------------
def _tap_sum(tab: torch.Tensor, cout: int) -> torch.Tensor:
    """[9 edge classes, >= cout] -> [cout, 3, 3]: for every tap, the sum over the edge classes in which it is inside the image."""
    # an index-sum, not a matrix product: `tab.t() @ M` dispatched to a Tensile / hipBLASLt GEMM (84 of them per BC step in the round-3
    # PMC survey) -- there is no vendor BLAS on the measured path
    m = packing.edge_tap_matrix(tab.device, tab.dtype)                      # [9 classes, 9 taps] of 0 / 1
    return (tab[:, :cout].t().unsqueeze(2) * m.unsqueeze(0)).sum(dim=1).view(cout, 3, 3)


def conv_param_grads(dw_raw: torch.Tensor, d_sa: torch.Tensor, d_sg: torch.Tensor, weight: torch.Tensor,
                     gain: torch.Tensor, bias: torch.Tensor):
    """Gradients of a GN -> conv layer's parameters from the kernels' outputs.
    Forward (vpt_conv3x3.hip): W' = W * gain[c];  v = rstd conv(W', x) + SA[e,o] - rstd mu SG[e,o],
    SG[e,o] = sum_{taps valid in e, c} W'[o,c,tap],  SA[e,o] = sum_{valid taps, c} W[o,c,tap] bias[c].
      dW'  = dw_raw (wgrad kernel, [Cout,9,Cin])  +  tap_sum(d_sg)  (broadcast over c)
      dW   = dW' * gain  +  bias[c] * tap_sum(d_sa)
      dgain[c] = sum_{o,tap} dW' W ;   dbias[c] = sum_{o,tap} W tap_sum(d_sa)
    Returns (dW [Cout,Cin,3,3], dgain [Cin], dbias [Cin])."""
    cout, cin = weight.shape[:2]
    dwp = dw_raw.view(cout, 3, 3, cin).permute(0, 3, 1, 2)                  # [Cout, Cin, 3, 3]
    dwp = dwp + _tap_sum(d_sg, cout).unsqueeze(1)
    ta = _tap_sum(d_sa, cout).unsqueeze(1)                                  # [Cout, 1, 3, 3]
    dW = dwp * gain.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1) * ta
    dgain = (dwp * weight).sum(dim=(0, 2, 3))
    dbias = (weight * ta).sum(dim=(0, 2, 3))
    return dW.contiguous(), dgain, dbias


def conv_dgrad_coef(stats_in: torch.Tensor, t12: torch.Tensor, n: int) -> torch.Tensor:
    """Per-frame (c0, c1) of the statistics terms of dx:  dx += c0 + c1 x,  c1 = -rstd^2 T1 / n,  c0 = -rstd T2 / n - c1 mu."""
    mu = stats_in[:, 0] / n
    rstd = torch.rsqrt((stats_in[:, 1] / n - mu * mu).clamp(min=0) + 1e-5)
    c1 = -(rstd * rstd) * t12[:, 0] / n
    c0 = -(rstd / n) * t12[:, 1] - c1 * mu
    return torch.stack([c0, c1], 1).float().contiguous()
