"""bf16-rounding emulation of the HIP path on the CPU  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The same restatement as oracle/vpt_oracle.py with a round-to-bf16 at exactly the points where the MI355X
kernels round: MFMA operands (conv / linear inputs and weights, GroupNorm gain folded into the
conv weight before rounding), CNN activations as stored between kernels (incl. the CNN residual stream and the
normalised input of the dense layer), the attention output and the MLP hidden activation.  Statistics,
LayerNorms, softmax, the transformer's residual stream, K/V and the head logits stay fp32.

Two uses: (1) it predicts how far a correct bf16 pipeline sits from the fp32 reference (the tolerances of
tests/test_gpu_policy.py); (2) because ReLU gates flip wherever the forward differs, gradients of a bf16 forward
differ from fp32-forward gradients by 15-30 % relative L2 per tensor even with exact autograd -- so the BC step's
backward is checked against THIS oracle's autograd (same gates), and only loosely (cosine) against the fp32 one.
torch's cast ops are differentiable (straight-through), so autograd through this file is the matched oracle."""
import torch
import torch.nn.functional as F

from . import vpt_oracle as O


def bf(t):
    return t.to(torch.bfloat16).float()


def f16(t):
    return t.to(torch.float16).float()


def bf_split(t):
    """hi + lo bf16 halves (what a 2- or 3-pass split-operand MFMA sees): ~16 mantissa bits."""
    hi = bf(t)
    return hi + bf(t - hi)


def f16_split(t):
    hi = f16(t)
    return hi + f16(t - hi)


def exact(t):
    return t


ROUNDERS = {"fp32": exact, "bf16": bf, "fp16": f16, "bf16x2": bf_split, "fp16x2": f16_split}


class Rounding:
    """Where the pipeline rounds: MFMA weight operands (w), CNN activations as stored between kernels and fed to the
    MFMAs (a), transformer-trunk GEMM operands / stored activations (t).  Default = what the bf16 kernels do."""

    def __init__(self, w="bf16", a="bf16", t="bf16", winograd=False):
        self.names = (w, a, t)
        self.w, self.a, self.t = ROUNDERS[w], ROUNDERS[a], ROUNDERS[t]
        self.winograd = winograd      # feasibility probe: the 3x3 convs as Winograd F(2x2, 3x3) with ROUNDED transformed operands


DEFAULT_ROUNDING = Rounding()


_WG = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])
_WBT = torch.tensor([[1.0, 0.0, -1.0, 0.0], [0.0, 1.0, 1.0, 0.0], [0.0, -1.0, 1.0, 0.0], [0.0, 1.0, 0.0, -1.0]])
_WAT = torch.tensor([[1.0, 1.0, 1.0, 0.0], [0.0, 1.0, -1.0, -1.0]])


def winograd_conv(x, w, rnd):
    """3x3 / pad 1 convolution as Winograd F(2x2, 3x3): what a kernel that feeds the matrix cores with the TRANSFORMED operands
    would compute -- U = G w G^T rounded to the weight format, V = B^T d B (exact in fp32 from the stored activations) rounded to
    the activation format, fp32 accumulation over the channels, output transform in fp32.  H, W even."""
    n, c, h, wd = x.shape
    k = w.shape[0]
    u = rnd.w(torch.einsum("ia,kcab,jb->kcij", _WG, w, _WG))                      # [K,C,4,4]
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                      # [N,C,H/2,W/2,4,4]
    v = rnd.a(torch.einsum("ia,nctuab,jb->nctuij", _WBT, tiles, _WBT))
    m = torch.einsum("kcij,nctuij->nktuij", u, v)
    y = torch.einsum("ia,nktuab,jb->nktuij", _WAT, m, _WAT)                         # [N,K,H/2,W/2,2,2]
    return y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, h, wd)


def conv_fold(sd, pfx, x_bf, res=None, rnd=DEFAULT_ROUNDING):
    """vpt_conv3x3_kernel: raw bf16 activations through conv(bf16(W*gain)), GroupNorm applied as the epilogue fold."""
    g, b, W = sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], sd[pfx + "layer.weight"]
    n = x_bf.shape[0]
    flat = x_bf.reshape(n, -1)
    mu = flat.mean(1)
    rstd = torch.rsqrt(flat.var(1, unbiased=False) + O.NORM_EPS)
    Wg = rnd.w(W * g.view(1, -1, 1, 1))
    acc = winograd_conv(x_bf, W * g.view(1, -1, 1, 1), rnd) if rnd.winograd else F.conv2d(x_bf, Wg, padding=1)
    ones = torch.ones(1, x_bf.shape[1], *x_bf.shape[2:])
    sg = F.conv2d(ones, Wg, padding=1)
    sa = F.conv2d(b.view(1, -1, 1, 1) * ones, W, padding=1)
    out = torch.relu(rstd.view(-1, 1, 1, 1) * acc - (rstd * mu).view(-1, 1, 1, 1) * sg + sa)
    if res is not None:
        out = out + res
    return rnd.a(out)


def policy_forward(sd, cfg, img_u8, first, state_in, grad=False, taps=None, rnd=DEFAULT_ROUNDING):
    bw, ba, bt = rnd.w, rnd.a, rnd.t
    with torch.set_grad_enabled(grad):
        b, t = img_u8.shape[:2]
        x = img_u8.reshape(b * t, 128, 128, 3).float().permute(0, 3, 1, 2)
        p = "net.img_process.cnn.stacks.0."
        # vpt_conv_first.hip: operands = raw bytes (exact in bf16) and bf16(W / 255); bias rides as hi + lo bf16 halves (~fp32)
        y = torch.relu(F.conv2d(x, bw(sd[p + "firstconv.layer.weight"] / 255.0), padding=1) + sd[p + "firstconv.layer.bias"].view(1, -1, 1, 1))
        cur = None
        for s in range(3):
            p = f"net.img_process.cnn.stacks.{s}."
            if s > 0:
                y = conv_fold(sd, p + "firstconv.", cur, rnd=rnd)
            y = ba(F.max_pool2d(ba(y), 3, 2, 1))
            cur = ba(O.group_norm_1(y, sd[p + "n.weight"], sd[p + "n.bias"]))
            for blk in range(2):
                q = f"{p}blocks.{blk}."
                h = conv_fold(sd, q + "conv0.", cur, rnd=rnd)
                cur = conv_fold(sd, q + "conv1.", h, res=cur, rnd=rnd)
        flat = cur.reshape(b * t, -1)
        p = "net.img_process.cnn.dense."
        xn = ba(O.layer_norm(flat, sd[p + "norm.weight"], sd[p + "norm.bias"]))
        d = xn @ bw(sd[p + "layer.weight"]).t()
        p = "net.img_process.linear."
        dn = bt(O.layer_norm(torch.relu(d), sd[p + "norm.weight"], sd[p + "norm.bias"]))
        x = torch.relu(dn @ bw(sd[p + "layer.weight"]).t()).reshape(b, t, -1)
        if taps is not None:
            taps["img_process"] = x
        first_b = first[:, 0]
        state_out = []
        hid, heads, maxlen = cfg["hidsize"], cfg["heads"], cfg["maxlen"]
        dh = hid // heads
        for l in range(cfg["n_layers"]):
            p = f"net.recurrent_layer.blocks.{l}."
            o = p + "r.orc_block."
            x1 = O.layer_norm(x, sd[p + "pre_r_ln.weight"], sd[p + "pre_r_ln.bias"])
            x1b = bt(x1)
            sm, (km, vm) = state_in[l]
            q = x1b @ bw(sd[o + "q_layer.weight"]).t() + sd[o + "q_layer.bias"]
            k = x1b @ bw(sd[o + "k_layer.weight"]).t()
            v = x1b @ bw(sd[o + "v_layer.weight"]).t()
            r = x1b @ bw(sd[o + "r_layer.weight"]).t() + sd[o + "r_layer.bias"]
            kf, vf = torch.cat([km, k], 1), torch.cat([vm, v], 1)
            split = lambda z: z.reshape(b, z.shape[1], heads, dh).permute(0, 2, 1, 3)
            lg = split(q) @ split(kf).transpose(-1, -2) / dh
            vis, nm = O.band_visibility(t, maxlen, first_b, sm)
            lg = lg + (~vis).float().unsqueeze(1) * O.NEG_MASK + O.rel_pos_bias(r.reshape(b, t, heads, -1), sd[o + "b_nd"], t, maxlen)
            a = bt((torch.softmax(lg, -1) @ split(vf)).permute(0, 2, 1, 3).reshape(b, t, hid))
            x2 = x1 + a @ bw(sd[o + "proj_layer.weight"]).t() + sd[o + "proj_layer.bias"]
            hb = bt(O.layer_norm(x2, sd[p + "mlp0.norm.weight"], sd[p + "mlp0.norm.bias"]))
            h2 = bt(torch.relu(hb @ bw(sd[p + "mlp0.layer.weight"]).t()))
            x = x2 + h2 @ bw(sd[p + "mlp1.layer.weight"]).t() + sd[p + "mlp1.layer.bias"]
            state_out.append((nm, (kf[:, -maxlen:], vf[:, -maxlen:])))
            if taps is not None:
                taps[f"block{l}"] = x
        xb = bt(O.layer_norm(torch.relu(x), sd["net.lastlayer.norm.weight"], sd["net.lastlayer.norm.bias"]))
        y = torch.relu(xb @ bw(sd["net.lastlayer.layer.weight"]).t())
        if taps is not None:
            taps["y"] = y
        lat = O.layer_norm(y, sd["net.final_ln.weight"], sd["net.final_ln.bias"])
        latb = bt(lat)
        out = {"latent": lat, "state_out": state_out,
               # value head: column 8762 of the fused heads GEMM (same operand rounding as the policy heads)
               "vpred": latb @ bw(sd["value_head.linear.weight"]).t() + sd["value_head.linear.bias"]}
        for hname in ("buttons", "camera"):
            z = latb @ bw(sd[f"pi_head.{hname}.linear_layer.weight"]).t() + sd[f"pi_head.{hname}.linear_layer.bias"]
            out[hname] = torch.log_softmax(z / cfg["temperature"], -1).unsqueeze(-2)
        return out


def bc_loss_and_grads(sd, cfg, img_u8, first, state_in, act_buttons, act_camera, rnd=DEFAULT_ROUNDING):
    """As vpt_oracle.bc_loss_and_grads, through the bf16-emulating forward."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point}
    state_det = [(m, (k.detach(), v.detach())) for m, (k, v) in state_in]
    out = policy_forward(leaves, cfg, img_u8, first, state_det, grad=True, rnd=rnd)
    with torch.enable_grad():    # (the caller may sit inside torch.no_grad(): the GPU tests' inference fixture)
        lp = out["buttons"][:, :, 0].gather(-1, act_buttons.unsqueeze(-1)).squeeze(-1) \
            + out["camera"][:, :, 0].gather(-1, act_camera.unsqueeze(-1)).squeeze(-1)
        loss = -lp.mean()
        names = list(leaves)
        grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    return float(loss.detach()), {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in zip(names, grads)}
