"""Torch-tensor front ends of the C-ABI entry points (one function per symbol of include/vpt_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; all arithmetic happens in
libvpt_hip.so.  Every function requires CUDA(HIP) tensors and raises if the native library is absent.
"""
import ctypes
import os

import torch

from . import _native
from ._native import ptr


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Optional per-entry-point HIP-event timing (bench.py / profiling only).  Events are recorded on the
    current torch stream, which is the stream every kernel is launched on."""

    def __init__(self):
        self.enabled = False
        self.records = []  # (name, start_event, stop_event, meta)

    def reset(self):
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b, meta in self.records:
            d = out.setdefault(name, dict(ms=0.0, calls=0, flops=0.0, bytes=0.0))
            d["ms"] += a.elapsed_time(b)
            d["calls"] += 1
            d["flops"] += meta.get("flops", 0.0)
            d["bytes"] += meta.get("bytes", 0.0)
        return out

    def by_shape(self, prefix="vpt_conv3x3"):
        """Per (label, work size): the same records split by their FLOP / byte count, i.e. by layer shape (profiling: which shapes lose)."""
        torch.cuda.synchronize()
        out = {}
        for name, a, b, meta in self.records:
            if not name.startswith(prefix):
                continue
            key = (name, meta.get("flops", 0.0) or meta.get("bytes", 0.0))
            d = out.setdefault(key, dict(ms=0.0, calls=0, flops=0.0, bytes=0.0))
            d["ms"] += a.elapsed_time(b)
            d["calls"] += 1
            d["flops"] += meta.get("flops", 0.0)
            d["bytes"] += meta.get("bytes", 0.0)
        return out


TIMER = KernelTimer()


POISON_LDS = os.environ.get("VPT_POISON_LDS", "0") == "1"    # diagnostics: NaN-fill every CU's LDS before each launch (vpt_debug_poison_lds)


def _call(name, meta, *args, fmt="bf16", label=None):
    if POISON_LDS:
        _native.call("vpt_debug_poison_lds", _stream(), fmt=fmt)
    if TIMER.enabled:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _native.call(name, *args, fmt=fmt)
        b.record()
        TIMER.records.append((label or name, a, b, meta or {}))
    else:
        _native.call(name, *args, fmt=fmt)


WS_CONV3X3_WGRAD, WS_CONV_BACKWARD_PREPARE, WS_LINEAR_SPLITK, WS_LAYERNORM_BACKWARD, WS_COLUMN_SUM = 1, 2, 3, 4, 5      # include/vpt_hip.h VPT_WS_*
WS_ATTENTION_BACKWARD_DKV, WS_ATTENTION_BACKWARD_DBND, WS_FRAME_AFFINE_BACKWARD, WS_CONV_FIRST_BACKWARD = 6, 7, 8, 9


def _workspace(op, a=0, b=0, c=0, d=0, e=0, device=None, fmt="bf16"):
    """fp32 scratch of vpt_workspace_bytes(op, ...) bytes (caller-owned: the library never allocates); None when the entry point needs none."""
    nbytes = int(_native.load(fmt).vpt_workspace_bytes(int(op), int(a), int(b), int(c), int(d), int(e)))
    if nbytes < 0:
        raise RuntimeError(f"vpt_workspace_bytes({op}, {a}, {b}, {c}, {d}, {e}) failed")
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device) if nbytes else None


# The 16-bit operand format of a call (bf16: libvpt_hip.so, fp16: libvpt_hip_f16.so, see include/vpt_hip.h
# vpt_operand_format) is the dtype of its 16-bit tensors; every 16-bit tensor of one call must agree.
OP16 = "op16"
_FMT = {torch.bfloat16: "bf16", torch.float16: "fp16"}


def _fmt(*tensors, dtype=None):
    """-> (torch dtype, library format) of the 16-bit tensors among `tensors` (or of `dtype` when there are none)."""
    dt = dtype
    for t in tensors:
        if t is not None and t.dtype in _FMT:
            if dt is not None and t.dtype != dt:
                raise TypeError(f"mixed 16-bit operand formats in one call: {dt} and {t.dtype}")
            dt = t.dtype
    if dt is None:
        dt = torch.bfloat16
    if dt not in _FMT:
        raise TypeError(f"16-bit operand dtype must be bfloat16 or float16, got {dt}")
    return dt, _FMT[dt]


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if dtype is OP16:
        if t.dtype not in _FMT:
            raise TypeError(f"{name}: expected a 16-bit operand tensor (bfloat16 / float16), got {t.dtype}")
    elif t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")


def pack_conv3x3(weight, gain, bias=None, tables=True, dtype=torch.bfloat16):
    """Device re-pack of a GN -> conv3x3 layer (vpt_pack_conv3x3): -> (wpk, edge_sa, edge_sg) exactly as packing.pack_conv3x3."""
    _chk(weight, torch.float32, "weight"); _chk(gain, torch.float32, "gain"); _chk(bias, torch.float32, "bias")
    cout, cin = weight.shape[:2]
    dt, fmt = _fmt(dtype=dtype)
    lib = _native.load(fmt)
    nt = (cout + 127) // 128
    wpk = torch.empty(nt, cin // 32, 9, 128, 32, dtype=dt, device=weight.device)
    assert wpk.numel() == lib.vpt_conv3x3_packed_elems(cout, cin)
    sa = sg = None
    if tables:
        sa = torch.empty(9, nt * 128, dtype=torch.float32, device=weight.device)
        sg = torch.empty(9, nt * 128, dtype=torch.float32, device=weight.device)
    _call("vpt_pack_conv3x3", None, ptr(weight), ptr(gain), ptr(bias), ptr(wpk), ptr(sa), ptr(sg), cout, cin, _stream(), fmt=fmt)
    return wpk, sa, sg


def pack_linear(weight, dtype=torch.bfloat16, transposed=False, k_pad=None):
    """Device re-pack of an nn.Linear weight [N, K] (vpt_pack_linear) -> [ceil(N/128)][K/32][128][32].
    transposed=True: the packed matrix is weight^T, i.e. [K rows][N reduction] with the reduction dimension zero-padded to
    k_pad (a multiple of 64) -- the operand of the input-gradient GEMM dx = dy W (training.linear_backward)."""
    _chk(weight, torch.float32, "weight")
    dt, fmt = _fmt(dtype=dtype)
    rows, cols = weight.shape
    if transposed:
        n_out, k_red, src_rows = cols, (k_pad or rows), rows
    else:
        n_out, k_red, src_rows = rows, cols, rows
    out = torch.empty((n_out + 127) // 128, k_red // 32, 128, 32, dtype=dt, device=weight.device)
    _call("vpt_pack_linear", None, ptr(weight), ptr(out), n_out, k_red, 1 if transposed else 0, cols, src_rows, _stream(), fmt=fmt)
    return out


def pack_conv_first(weight, bias, dtype=torch.bfloat16):
    """Device re-pack of the stack-0 firstconv (vpt_pack_conv_first) -> fragments [NT][4][2][64][8], exactly as packing.pack_conv_first."""
    _chk(weight, torch.float32, "weight"); _chk(bias, torch.float32, "bias")
    cout = weight.shape[0]
    if tuple(weight.shape[1:]) != (3, 3, 3):
        raise ValueError(f"pack_conv_first: weight must be [Cout, 3, 3, 3], got {tuple(weight.shape)}")
    dt, fmt = _fmt(dtype=dtype)
    out = torch.empty((cout + 127) // 128, 4, 2, 64, 8, dtype=dt, device=weight.device)
    assert out.numel() == _native.load(fmt).vpt_conv_first_packed_elems(cout)
    _call("vpt_pack_conv_first", None, ptr(weight), ptr(bias), ptr(out), cout, _stream(), fmt=fmt)
    return out


def pack_conv3d_t5(weight, bias, dtype=torch.bfloat16):
    """Device re-pack of the IDM's temporal conv (vpt_pack_conv3d_t5) -> (fragments [NT][4][64][8], bias fp32 [NT*128])."""
    _chk(weight, torch.float32, "weight"); _chk(bias, torch.float32, "bias")
    o = weight.shape[0]
    if tuple(weight.shape[1:]) != (3, 5, 1, 1):
        raise ValueError(f"pack_conv3d_t5: weight must be [O, 3, 5, 1, 1], got {tuple(weight.shape)}")
    dt, fmt = _fmt(dtype=dtype)
    nt = (o + 127) // 128
    out = torch.empty(nt, 4, 64, 8, dtype=dt, device=weight.device)
    bp = torch.empty(nt * 128, dtype=torch.float32, device=weight.device)
    _call("vpt_pack_conv3d_t5", None, ptr(weight), ptr(bias), ptr(out), ptr(bp), o, _stream(), fmt=fmt)
    return out, bp


def chw_to_blocked(x, c, h, w):
    """fp32 [rows, c*h*w] (or [c*h*w]) in C,H,W flatten order -> the same shape in blocked activation order (vpt_chw_to_blocked)."""
    _chk(x, torch.float32, "x")
    rows = 1 if x.dim() == 1 else x.shape[0]
    if x.numel() != rows * c * h * w:
        raise ValueError(f"chw_to_blocked: {tuple(x.shape)} is not [rows, {c}*{h}*{w}]")
    out = torch.empty_like(x)
    _call("vpt_chw_to_blocked", None, ptr(x), ptr(out), ctypes.c_int64(rows), c, h, w, _stream())
    return out


def conv_first(img_u8, wfrag, cout, stats_out=None, out_gain=None, chs_out=None):
    """img_u8 [F,H,W,3] uint8 -> pooled blocked bf16 [F, cout/32, H/2, W/2, 32].  out_gain fp32 [cout]: stored times it per channel
    (GroupNorm `n`'s gain when the norm is folded into the first block); stats_out are the statistics of the unscaled tensor.
    chs_out fp64 [F, cout, 2] (zeroed; cout <= 128): receives the per-channel sums of the stored tensor (what channel_stats() computes)."""
    _chk(img_u8, torch.uint8, "img"); _chk(wfrag, OP16, "wfrag"); _chk(stats_out, torch.float64, "stats_out"); _chk(out_gain, torch.float32, "out_gain")
    _chk(chs_out, torch.float64, "chs_out")
    f, h, w, _ = img_u8.shape
    dt, fmt = _fmt(wfrag)
    y = torch.empty(f, cout // 32, h // 2, w // 2, 32, dtype=dt, device=img_u8.device)
    _call("vpt_conv_first_forward", dict(flops=2.0 * f * h * w * cout * 27, bytes=f * (h * w * 3 + h * w * cout // 2)), ptr(img_u8), ptr(wfrag), ptr(y), ptr(stats_out), ptr(out_gain), ptr(chs_out), f, h, w, cout, _stream(), fmt=fmt)
    return y


CONV_TILING = {"throughput": 1, "latency": 2, "throughput32": 3}   # throughput32: 32-row / eight-wave tiles where H % 32 == 0 (measured at parity; A/B)


def conv3x3(x, wpk, edge_sa, edge_sg, stats_in, cout, res=None, stats_out=None, out=None, tiling="throughput"):
    """x blocked bf16 [F,Cin/32,H,W,32] -> blocked bf16 [F,cout/32,H,W,32] (GN fold + ReLU [+res]).
    tiling: "throughput" (default: a frame's result must not depend on how many frames share the launch -- chunking, sharding
    over ranks and the batch size are free choices of the caller; the two tilings sum a tile's statistics in different orders),
    or "latency" (the acting step asks for it explicitly) -- vpt_conv3x3_forward_tiled; the library never picks by grid size."""
    _chk(x, OP16, "x"); _chk(wpk, OP16, "wpk"); _chk(edge_sa, torch.float32, "edge_sa")
    _chk(edge_sg, torch.float32, "edge_sg"); _chk(stats_in, torch.float64, "stats_in")
    _chk(res, OP16, "res"); _chk(stats_out, torch.float64, "stats_out")
    f, cb, h, w, _ = x.shape
    dt, fmt = _fmt(x, wpk, res, out)
    if out is None:
        out = torch.empty(f, cout // 32, h, w, 32, dtype=dt, device=x.device)
    meta = dict(flops=2.0 * f * h * w * cout * 9 * cb * 32, bytes=2.0 * f * h * w * (cb * 32 + cout * (2 if res is not None else 1)))
    if tiling not in CONV_TILING:
        raise ValueError(f"conv3x3: tiling must be one of {sorted(CONV_TILING)}, got {tiling!r}")
    _call("vpt_conv3x3_forward_tiled", meta, ptr(x), ptr(wpk), ptr(edge_sa), ptr(edge_sg), ptr(stats_in), ptr(res),
          ptr(out), ptr(stats_out), f, h, w, cb * 32, cout, CONV_TILING[tiling], _stream(), fmt=fmt,
          label="vpt_conv3x3_forward_latency" if tiling == "latency" else "vpt_conv3x3_forward")
    return out


def conv3x3_pool(x, wpk, edge_sa, edge_sg, stats_in, cout, stats_out=None, out=None, out_gain=None, chs_out=None):
    """GN fold + conv3x3 + ReLU + max_pool2d(3, 2, 1) in one pass (vpt_conv3x3_pool_forward): x blocked [F,Cin/32,H,W,32] -> pooled
    [F,cout/32,H/2,W/2,32] (+ its frame statistics into stats_out) -- what conv3x3() followed by maxpool() returns, bit for bit, without
    the pre-pool tensor's round trip through HBM.  Inference path of the stacks' firstconv (the BC step keeps the pre-pool tensor)."""
    _chk(x, OP16, "x"); _chk(wpk, OP16, "wpk"); _chk(edge_sa, torch.float32, "edge_sa"); _chk(edge_sg, torch.float32, "edge_sg")
    _chk(stats_in, torch.float64, "stats_in"); _chk(stats_out, torch.float64, "stats_out"); _chk(out, OP16, "out"); _chk(out_gain, torch.float32, "out_gain"); _chk(chs_out, torch.float64, "chs_out")
    f, cb, h, w, _ = x.shape
    dt, fmt = _fmt(x, wpk, out)
    if out is not None and (tuple(out.shape) != (f, cout // 32, h // 2, w // 2, 32) or not out.is_contiguous()):
        raise ValueError("conv3x3_pool: out must be a contiguous [F, cout/32, H/2, W/2, 32] tensor")
    y = out if out is not None else torch.empty(f, cout // 32, h // 2, w // 2, 32, dtype=dt, device=x.device)
    seam = torch.empty(_native.load(fmt).vpt_conv3x3_pool_seam_elems(f, h, w, cout), dtype=dt, device=x.device)
    meta = dict(flops=2.0 * f * h * w * cout * 9 * cb * 32, bytes=2.0 * f * h * w * (cb * 32 + cout * 0.25))
    args = (ptr(x), ptr(wpk), ptr(edge_sa), ptr(edge_sg), ptr(stats_in), ptr(y), ptr(seam), ptr(stats_out), ptr(out_gain), ptr(chs_out), f, h, w, cb * 32, cout)
    if TIMER.enabled:    # the two launches timed apart: the convolution is the roofline kernel (vpt_conv3x3_kernel, pool-fused mode: same FLOPs)
        _call("vpt_conv3x3_pool_forward", meta, *args, 1, _stream(), fmt=fmt, label="vpt_conv3x3_pool_forward")
        _call("vpt_conv3x3_pool_forward", dict(bytes=2.0 * y.numel() * 0.4), *args, 2, _stream(), fmt=fmt, label="vpt_pool_seam")
    else:
        _call("vpt_conv3x3_pool_forward", meta, *args, 3, _stream(), fmt=fmt)
    return y


def conv3x3_pool_argmax(x, wpk, edge_sa, edge_sg, stats_in, cout, stats_out=None):
    """conv3x3_pool() for the TRAINING forward (vpt_conv3x3_pool_argmax_forward): -> (pooled [F,cout/32,H/2,W/2,32], mask int16 of the same shape).
    mask: per pooled value the 9-bit "window position differs from the maximum" word (bit 8 - k for scan position k; outside the image: 1) --
    conv_backward_prepare_pooled() routes the pooled gradient to the first zero bit; the pre-pool tensor never exists."""
    _chk(x, OP16, "x"); _chk(wpk, OP16, "wpk"); _chk(edge_sa, torch.float32, "edge_sa"); _chk(edge_sg, torch.float32, "edge_sg")
    _chk(stats_in, torch.float64, "stats_in"); _chk(stats_out, torch.float64, "stats_out")
    f, cb, h, w, _ = x.shape
    dt, fmt = _fmt(x, wpk)
    y = torch.empty(f, cout // 32, h // 2, w // 2, 32, dtype=dt, device=x.device)
    mask = torch.empty(f, cout // 32, h // 2, w // 2, 32, dtype=torch.int16, device=x.device)
    seam = torch.empty(_native.load(fmt).vpt_conv3x3_pool_seam_elems(f, h, w, cout), dtype=dt, device=x.device)
    meta = dict(flops=2.0 * f * h * w * cout * 9 * cb * 32, bytes=2.0 * f * h * w * (cb * 32 + cout * 0.5))
    args = (ptr(x), ptr(wpk), ptr(edge_sa), ptr(edge_sg), ptr(stats_in), ptr(y), ptr(mask), ptr(seam), ptr(stats_out), f, h, w, cb * 32, cout)
    if TIMER.enabled:    # the two launches timed apart, as in conv3x3_pool()
        _call("vpt_conv3x3_pool_argmax_forward", meta, *args, 1, _stream(), fmt=fmt, label="vpt_conv3x3_pool_forward")
        _call("vpt_conv3x3_pool_argmax_forward", dict(bytes=2.0 * y.numel() * 0.8), *args, 2, _stream(), fmt=fmt, label="vpt_pool_seam")
    else:
        _call("vpt_conv3x3_pool_argmax_forward", meta, *args, 3, _stream(), fmt=fmt)
    return y, mask


def conv_backward_prepare_pooled(dpooled, pooled, mask, stats_in, edge_sa, edge_sg, cin, d_sa=None, d_sg=None, want_t12=False, nfold=None):
    """conv_backward_prepare() for the layer in front of the max-pool when its forward was conv3x3_pool_argmax(): (dpooled, pooled, mask) at the
    pooled resolution -> (dacc blocked [F,Cout/32,2h,2w,32], coef, d_sa, d_sg[, t12]) (vpt_conv_backward_prepare_pooled).
    nfold = (n_gain fp32 [Cout], pool_stats fp64 [F,2], ab fp64 [F,2] from frame_affine_backward_reduce): `dpooled` is the gradient w.r.t.
    n(pooled) and the GroupNorm `n` backward is applied on the fly (no second affine-backward pass)."""
    ng, pst, pab = nfold if nfold is not None else (None, None, None)
    _chk(ng, torch.float32, "n_gain"); _chk(pst, torch.float64, "pool_stats"); _chk(pab, torch.float64, "pool_ab")
    _chk(dpooled, OP16, "dpooled"); _chk(pooled, OP16, "pooled"); _chk(mask, torch.int16, "mask")
    _chk(stats_in, torch.float64, "stats_in"); _chk(edge_sa, torch.float32, "edge_sa"); _chk(edge_sg, torch.float32, "edge_sg")
    _chk(d_sa, torch.float32, "d_sa"); _chk(d_sg, torch.float32, "d_sg")
    if dpooled.shape != pooled.shape or mask.shape != pooled.shape:
        raise ValueError("conv_backward_prepare_pooled: dpooled, pooled and mask must have one shape")
    f, cb, ph, pw, _ = pooled.shape
    h, w = 2 * ph, 2 * pw
    dev = pooled.device
    dt, fmt = _fmt(dpooled, pooled)
    dacc = torch.empty(f, cb, h, w, 32, dtype=dt, device=dev)
    coef = torch.empty(f, 2, dtype=torch.float32, device=dev)
    t12 = torch.empty(f, 2, dtype=torch.float64, device=dev) if want_t12 else None
    if d_sa is None:
        d_sa, d_sg = torch.zeros_like(edge_sa), torch.zeros_like(edge_sg)
    scratch = _workspace(WS_CONV_BACKWARD_PREPARE, f, 0, 0, 0, cb * 32, device=dev)
    _call("vpt_conv_backward_prepare_pooled", dict(bytes=2.0 * dacc.numel() + 6.0 * pooled.numel()), ptr(dpooled), ptr(pooled), ptr(mask), ptr(stats_in), ptr(edge_sa), ptr(edge_sg),
          ptr(dacc), ptr(t12), ptr(coef), ptr(d_sa), ptr(d_sg), ptr(scratch), ptr(ng), ptr(pst), ptr(pab), f, h, w, cin, cb * 32, _stream(), fmt=fmt, label="vpt_conv_backward_prepare")
    return (dacc, coef, d_sa, d_sg, t12) if want_t12 else (dacc, coef, d_sa, d_sg)


def conv3x3_folded(x, wpk, edge_sa, edge_sg, stats_in, cout, kk_frame=None, rs_frame=None, res=None, res_scale=None, res_bias=None, stats_out=None, out=None):
    """conv3x3() with the GroupNorm `n` of the stack folded in (vpt_conv3x3_forward_folded): (kk_frame [F,9,CoutPad], rs_frame [F]) from
    nfold_coef() replace edge_sa and the statistics of x (conv0 on the gain-scaled pooled tensor Q); (res_scale [F], res_bias [F,cout])
    make the residual res_scale * res + res_bias (conv1 with res = Q)."""
    _chk(x, OP16, "x"); _chk(wpk, OP16, "wpk"); _chk(edge_sa, torch.float32, "edge_sa"); _chk(edge_sg, torch.float32, "edge_sg")
    _chk(stats_in, torch.float64, "stats_in"); _chk(kk_frame, torch.float32, "kk_frame"); _chk(rs_frame, torch.float32, "rs_frame")
    _chk(res, OP16, "res"); _chk(res_scale, torch.float32, "res_scale"); _chk(res_bias, torch.float32, "res_bias"); _chk(stats_out, torch.float64, "stats_out")
    f, cb, h, w, _ = x.shape
    dt, fmt = _fmt(x, wpk, res)
    if kk_frame is not None and tuple(kk_frame.shape) != (f, 9, edge_sg.shape[1]):
        raise ValueError(f"conv3x3_folded: kk_frame must be [F, 9, {edge_sg.shape[1]}], got {tuple(kk_frame.shape)}")
    if res_bias is not None and tuple(res_bias.shape) != (f, cout):
        raise ValueError(f"conv3x3_folded: res_bias must be [F, {cout}], got {tuple(res_bias.shape)}")
    if out is None:
        out = torch.empty(f, cout // 32, h, w, 32, dtype=dt, device=x.device)
    meta = dict(flops=2.0 * f * h * w * cout * 9 * cb * 32, bytes=2.0 * f * h * w * (cb * 32 + cout * (2 if res is not None else 1)))
    _call("vpt_conv3x3_forward_folded", meta, ptr(x), ptr(wpk), ptr(edge_sa), ptr(edge_sg), ptr(stats_in), ptr(kk_frame), ptr(rs_frame), ptr(res),
          ptr(res_scale), ptr(res_bias), ptr(out), ptr(stats_out), f, h, w, cb * 32, cout, _stream(), fmt=fmt, label="vpt_conv3x3_forward")
    return out


def channel_stats(x, out=None):
    """Blocked [F,C/32,H,W,32] -> fp64 [F, C, 2] per-frame, per-channel (sum, sum of squares) (vpt_channel_stats)."""
    _chk(x, OP16, "x"); _chk(out, torch.float64, "out")
    f, cb, h, w, _ = x.shape
    chs = out if out is not None else torch.zeros(f, cb * 32, 2, dtype=torch.float64, device=x.device)
    _call("vpt_channel_stats", dict(bytes=2.0 * x.numel()), ptr(x), ptr(chs), f, cb * 32, h * w, _stream(), fmt=_fmt(x)[1])
    return chs


def nfold_coef(tot, chs, gain, bias, sa, sg, tb, tg, hw, cout):
    """Per-frame coefficients of the folded GroupNorm `n` (vpt_nfold_coef) -> (kk_frame [F,9,CoutPad], rs_frame [F], res_scale [F], res_bias [F,C])."""
    _chk(tot, torch.float64, "tot"); _chk(chs, torch.float64, "chs")
    for t_, nme in ((gain, "gain"), (bias, "bias"), (sa, "sa"), (sg, "sg"), (tb, "tb"), (tg, "tg")):
        _chk(t_, torch.float32, nme)
    f, c, _ = chs.shape
    dev = chs.device
    kk = torch.empty(f, 9, sa.shape[1], dtype=torch.float32, device=dev)
    rs = torch.empty(f, dtype=torch.float32, device=dev)
    rsc = torch.empty(f, dtype=torch.float32, device=dev)
    rb = torch.empty(f, c, dtype=torch.float32, device=dev)
    _call("vpt_nfold_coef", dict(bytes=4.0 * kk.numel()), ptr(tot), ptr(chs), ptr(gain), ptr(bias), ptr(sa), ptr(sg), ptr(tb), ptr(tg),
          ptr(kk), ptr(rs), ptr(rsc), ptr(rb), f, c, int(hw), int(cout), _stream())
    return kk, rs, rsc, rb


def maxpool(x, stats_out=None, want_argmax=False, out=None):
    """-> pooled, or (pooled, argmax uint8) with want_argmax (training: vpt_conv_backward_prepare routes through it).
    out: pooled tensor to fill (e.g. a slice of a larger batch's)."""
    _chk(x, OP16, "x"); _chk(stats_out, torch.float64, "stats_out"); _chk(out, OP16, "out")
    f, cb, h, w, _ = x.shape
    dt, fmt = _fmt(x, out)
    if out is not None and (tuple(out.shape) != (f, cb, h // 2, w // 2, 32) or not out.is_contiguous()):
        raise ValueError("maxpool: out must be a contiguous [F, C/32, H/2, W/2, 32] tensor")
    y = out if out is not None else torch.empty(f, cb, h // 2, w // 2, 32, dtype=dt, device=x.device)
    am = torch.empty(f, cb, h // 2, w // 2, 32, dtype=torch.uint8, device=x.device) if want_argmax else None
    _call("vpt_maxpool_forward", dict(bytes=2.0 * f * cb * 32 * h * w * 1.25), ptr(x), ptr(y), ptr(stats_out), ptr(am), f, cb * 32, h, w, _stream(), fmt=fmt)
    return (y, am) if want_argmax else y


def frame_affine(x, gain, bias, stats_in, stats_out=None, per_element=False, out=None):
    _chk(x, OP16, "x"); _chk(gain, torch.float32, "gain"); _chk(bias, torch.float32, "bias")
    _chk(stats_in, torch.float64, "stats_in"); _chk(stats_out, torch.float64, "stats_out")
    f, cb, h, w, _ = x.shape
    if out is None:
        out = torch.empty_like(x)
    _call("vpt_frame_affine_forward", dict(bytes=4.0 * f * cb * 32 * h * w), ptr(x), ptr(out), ptr(gain), ptr(bias), ptr(stats_in), ptr(stats_out),
                 f, cb * 32, h * w, 1 if per_element else 0, _stream(), fmt=_fmt(x, out)[1])
    return out


def dense_fold_epilogue(part, stats, count, sg, sb):
    """Split-K partial slices [S, M, N] of x @ op16(W gain)^T + the frame statistics of x -> LayerNorm(x) @ W^T [M, N] (vpt_dense_fold_epilogue)."""
    _chk(part, torch.float32, "part"); _chk(stats, torch.float64, "stats"); _chk(sg, torch.float32, "sg"); _chk(sb, torch.float32, "sb")
    s_, m, n = part.shape
    out = torch.empty(m, n, dtype=torch.float32, device=part.device)
    _call("vpt_dense_fold_epilogue", dict(bytes=4.0 * (s_ + 1) * m * n), ptr(part), s_, ptr(stats), int(count), ptr(sg), ptr(sb), ptr(out), m, n, _stream())
    return out


_GEMM256 = os.environ.get("VPT_LINEAR_256", "0") == "1"      # A/B switch: "throughput" takes the 256 x 256 kernel (bit-identical results either way)
LINEAR_TILING = {"auto": 0, "throughput": 1, "latency": 2, "throughput256": 4}   # throughput256: vpt_gemm256_kernel where its grid fills the chip (measured neutral in the engine: not the default)


def nk_splitk(n: int, k: int) -> int:
    """The split-K factor of a mid-size-M linear as a function of the LAYER (N, K) alone: enough K slices that one row tile's N/128 column tiles
    fill the chip.  Named by the caller (`splitk="nk"`: IDMEngine, whose workload is one <= 160-frame window), never derived from M."""
    tiles = (n + 127) // 128
    if k < 2048 or tiles >= 128:
        return 1
    return max(1, min(16, k // 512, 256 // tiles))


def _every_split_has_k_steps(k: int, sk: int) -> bool:
    """vpt_gemm_kernel cuts the K / 64 steps into `sk` runs of ceil(steps / sk): when the last run is not empty every [M, N] slice of the partial buffer
    is written in full and needs no zero-fill (the dense layer: 1024 steps in 32 runs)."""
    steps = k // 64
    return sk >= 1 and (sk - 1) * ((steps + sk - 1) // sk) < steps


def linear(a_bf16, wpk, n, bias=None, res=None, relu=False, out_f32=True, out_bf16=False, splitk=1, mask=None,
           out_bf16_ld=None, splitk_raw=False, tiling="auto"):
    """a [M,K] bf16 (row stride = K) x packed weight -> ([M,n] fp32 or None, [M,ld] bf16 or None).
    mask: optional bf16 [M, >=n] gate (output zeroed where mask <= 0).  out_bf16_ld: row stride of the bf16
    output (>= n, extra columns zero) so it can feed the next GEMM as an A operand with K padded to 64.
    tiling: "throughput" = the MFMA GEMM whatever M (the inference engine's batch path: a row's result must not depend on how many rows
    share the call), "latency" = the weight-streaming kernel (M <= 8: the acting step), "auto" = by M (vpt_linear_forward).
    splitk: an int (the caller's explicit factor: the dense layer), or "nk" = nk_splitk(n, k) finished by the epilogue kernel.  Under the
    named tilings ("throughput", "latency") the summation order over K is a function of (tiling, splitk, N, K) ONLY -- never of M:
    the M-based automatic split below applies to tiling="auto" alone (the BC step's small test shapes)."""
    _chk(a_bf16, OP16, "A"); _chk(wpk, OP16, "wpk"); _chk(bias, torch.float32, "bias")
    _chk(res, torch.float32, "res"); _chk(mask, OP16, "mask")
    m, k = a_bf16.shape
    dev = a_bf16.device
    dt, fmt = _fmt(a_bf16, wpk, mask)
    ld16 = (out_bf16_ld or n) if out_bf16 else n
    if tiling == "throughput" and _GEMM256:
        tiling = "throughput256"
    tl = LINEAR_TILING[tiling]
    auto_sk = 1
    if splitk == "nk":
        splitk = 1
        if tl != 2:
            auto_sk = nk_splitk(n, k)
    elif splitk == 1 and tl == 0 and 8 < m <= 512 and k >= 2048:
        # tiling "auto" only.  Mid-size M: the 256 x 128 tiling alone gives N/128 workgroups for 256 CUs, so cut K as well and finish
        # with the epilogue kernel (fixed summation order: deterministic)
        tiles = ((m + 255) // 256) * ((n + 127) // 128)
        if tiles < 128:
            auto_sk = max(1, min(16, k // 512, 256 // tiles))
    if auto_sk > 1:
        part = (torch.empty if _every_split_has_k_steps(k, auto_sk) else torch.zeros)(auto_sk, m, n, dtype=torch.float32, device=dev)      # (this branch is always the MFMA GEMM)
        _call("vpt_linear_forward_tiled", dict(flops=2.0 * m * n * k, bytes=2.0 * (m * k + n * k) + 4.0 * m * n), ptr(a_bf16), ptr(wpk), None, None, ptr(part), None,
              m, n, k, k, n, n, n, 0, auto_sk, None, 0, tl, _stream(), fmt=fmt, label="vpt_linear_forward")
        o32 = torch.empty(m, n, dtype=torch.float32, device=dev) if out_f32 else None
        o16 = None
        if out_bf16:
            o16 = torch.zeros(m, ld16, dtype=dt, device=dev) if ld16 > n else torch.empty(m, n, dtype=dt, device=dev)
        _call("vpt_linear_splitk_epilogue", dict(bytes=4.0 * (auto_sk + 1) * m * n), ptr(part), auto_sk, ptr(bias), ptr(res), ptr(o32), ptr(o16),
              m, n, n, n, ld16, 1 if relu else 0, ptr(mask), mask.shape[1] if mask is not None else 0, _stream(), fmt=fmt)
        return o32, o16
    o32 = None
    if out_f32:   # split-K: one [m, n] slice per split (zeroed: a split without k-steps writes nothing), summed below
        if splitk > 1:
            gemm = tl in (1, 4) or (tl == 0 and m > 8)          # the MFMA GEMM's K partition (the weight-streaming kernel cuts K its own way: keep its zero-fill)
            o32 = (torch.empty if gemm and _every_split_has_k_steps(k, splitk) else torch.zeros)(splitk, m, n, dtype=torch.float32, device=dev)
        else:
            o32 = torch.empty(m, n, dtype=torch.float32, device=dev)
    o16 = None
    if out_bf16:
        o16 = torch.zeros(m, ld16, dtype=dt, device=dev) if ld16 > n else torch.empty(m, n, dtype=dt, device=dev)
    _call("vpt_linear_forward_tiled", dict(flops=2.0 * m * n * k, bytes=2.0 * (m * k + n * k) + 4.0 * m * n), ptr(a_bf16), ptr(wpk), ptr(bias), ptr(res), ptr(o32), ptr(o16),
          m, n, k, k, n, n, ld16, 1 if relu else 0, splitk, ptr(mask), mask.shape[1] if mask is not None else 0, tl, _stream(), fmt=fmt, label="vpt_linear_forward")
    if splitk > 1 and o32 is not None and not splitk_raw:
        o32 = o32.sum(0)           # fixed summation order: the result does not depend on scheduling
    return o32, o16                # (splitk_raw: the [splitk, M, n] partial slices, for an epilogue of the caller's)


def linear_wgrad(dy16, x16, n, out=None):
    """dW [n, K] fp32 = dy16[:, :n]^T @ x16 (sum over the M rows), straight from the row-major activations; added to `out` when given."""
    _chk(dy16, OP16, "dy"); _chk(x16, OP16, "x"); _chk(out, torch.float32, "out")
    m, k = x16.shape
    assert dy16.shape[0] == m and n % 8 == 0 and k % 8 == 0
    dw = out if out is not None else torch.empty(n, k, dtype=torch.float32, device=x16.device)
    _call("vpt_linear_wgrad", dict(flops=2.0 * m * n * k), ptr(dy16), ptr(x16), ptr(dw), m, n, k, dy16.shape[1], k, k, 1 if out is not None else 0, _stream(), fmt=_fmt(dy16, x16)[1])
    return dw


def layernorm(x, gain, bias, relu_in=False, out_f32=False, out_bf16=True, dtype=torch.bfloat16):
    """dtype: format of the 16-bit output (the A operand of the next GEMM)."""
    _chk(x, torch.float32, "x"); _chk(gain, torch.float32, "gain"); _chk(bias, torch.float32, "bias")
    m, d = x.shape
    o32 = torch.empty_like(x) if out_f32 else None
    dt, fmt = _fmt(dtype=dtype)
    o16 = torch.empty(m, d, dtype=dt, device=x.device) if out_bf16 else None
    _call("vpt_layernorm_forward", dict(bytes=6.0 * m * d), ptr(x), ptr(gain), ptr(bias), ptr(o32), ptr(o16), m, d, 1 if relu_in else 0, _stream(), fmt=fmt)
    return o32, o16


LN_LINEAR_MAX_ROWS, LN_LINEAR_MAX_K = 8, 3072


def layernorm_linear(x, gain, ln_bias, wpk, n, bias=None, res=None, relu=False, relu_in=False, ln_out_f32=False, out_f32=True, out_bf16=False,
                     dtype=torch.bfloat16):
    """LayerNorm (optionally of relu(x)) fused into the linear layer it feeds, for the acting path (M <= 8 rows, K <= 3072):
    (normalised rows fp32 or None, [M,n] fp32 or None, [M,n] 16-bit or None) -- bit-identical to layernorm() + linear()."""
    _chk(x, torch.float32, "x"); _chk(gain, torch.float32, "gain"); _chk(ln_bias, torch.float32, "ln_bias"); _chk(wpk, OP16, "wpk")
    _chk(bias, torch.float32, "bias"); _chk(res, torch.float32, "res")
    m, k = x.shape
    if m > LN_LINEAR_MAX_ROWS or k > LN_LINEAR_MAX_K:
        raise ValueError(f"layernorm_linear: M <= {LN_LINEAR_MAX_ROWS}, K <= {LN_LINEAR_MAX_K} (got {m} x {k}); call layernorm() and linear()")
    dt, fmt = _fmt(wpk, dtype=dtype)
    ln32 = torch.empty_like(x) if ln_out_f32 else None
    o32 = torch.empty(m, n, dtype=torch.float32, device=x.device) if out_f32 else None
    o16 = torch.empty(m, n, dtype=dt, device=x.device) if out_bf16 else None
    _call("vpt_layernorm_linear_forward", dict(flops=2.0 * m * n * k, bytes=2.0 * n * k + 4.0 * m * (n + 2 * k)), ptr(x), ptr(gain), ptr(ln_bias), 1 if relu_in else 0, ptr(ln32),
          ptr(wpk), ptr(bias), ptr(res), ptr(o32), ptr(o16), m, n, k, n, n, n, 1 if relu else 0, _stream(), fmt=fmt)
    return ln32, o32, o16


def conv3d_t5(img_u8, wfrag, bias, cout, t, stats_out=None):
    """img_u8 [F = B*t, H, W, 3] uint8 -> blocked bf16 [F, cout/32, H, W, 32] (IDM temporal conv + ReLU)."""
    _chk(img_u8, torch.uint8, "img"); _chk(wfrag, OP16, "wfrag"); _chk(bias, torch.float32, "bias")
    _chk(stats_out, torch.float64, "stats_out")
    f, h, w, _ = img_u8.shape
    dt, fmt = _fmt(wfrag)
    y = torch.empty(f, cout // 32, h, w, 32, dtype=dt, device=img_u8.device)
    _call("vpt_conv3d_t5_forward", dict(flops=2.0 * f * h * w * cout * 15, bytes=f * h * w * (3 + 2 * cout)),
          ptr(img_u8), ptr(wfrag), ptr(bias), ptr(y), ptr(stats_out), f, t, h, w, cout, _stream(), fmt=fmt)
    return y


def full_attention(qkv, batch, t, heads, hid, dtype=torch.bfloat16):
    """Mask "none", no memory (IDM): every query attends to all t rows of its chunk.  qkv [B*t, 3*hid] fp32."""
    _chk(qkv, torch.float32, "qkv")
    dt, fmt = _fmt(dtype=dtype)
    out = torch.empty(batch * t, hid, dtype=dt, device=qkv.device)
    _call("vpt_masked_attention_forward", dict(flops=4.0 * batch * t * t * hid), ptr(qkv), None, None, None, None, ptr(out),
          batch, t, heads, hid, qkv.shape[1], 0, 0, _stream(), fmt=fmt)
    return out


def masked_attention(qkvr, kmem, vmem, memvalid, b_nd, batch, t, heads, hid, dtype=torch.bfloat16):
    _chk(qkvr, torch.float32, "qkvr"); _chk(kmem, torch.float32, "kmem"); _chk(vmem, torch.float32, "vmem")
    _chk(memvalid, torch.uint8, "memvalid"); _chk(b_nd, torch.float32, "b_nd")
    maxlen = kmem.shape[1]
    dt, fmt = _fmt(dtype=dtype)
    out = torch.empty(batch * t, hid, dtype=dt, device=qkvr.device)
    _call("vpt_masked_attention_forward", dict(flops=4.0 * batch * t * (t + maxlen) * hid), ptr(qkvr), ptr(kmem), ptr(vmem), ptr(memvalid), ptr(b_nd), ptr(out),
                 batch, t, heads, hid, qkvr.shape[1], maxlen, 1, _stream(), fmt=fmt)
    return out


def kv_memory_update(qkvr, kmem, vmem, batch, t, hid):
    _chk(qkvr, torch.float32, "qkvr"); _chk(kmem, torch.float32, "kmem"); _chk(vmem, torch.float32, "vmem")
    kout, vout = torch.empty_like(kmem), torch.empty_like(vmem)
    _call("vpt_kv_memory_update", dict(bytes=16.0 * batch * kmem.shape[1] * hid), ptr(qkvr), ptr(kmem), ptr(vmem), ptr(kout), ptr(vout),
                 batch, t, hid, qkvr.shape[1], kmem.shape[1], _stream())
    return kout, vout


ATTENTION_STEP_MAXLEN = 128


def masked_attention_step(qkvr, kmem, vmem, state_mask, first, b_nd, batch, heads, hid, dtype=torch.bfloat16, inplace=False, done=None):
    """Acting step (t = 1): masked_attention(), kv_memory_update() and the state-mask bookkeeping in one launch.
    state_mask bool/uint8 [batch, maxlen], first bool/uint8 [batch] -> (out 16-bit [batch, hid], kout, vout, new mask uint8 [batch, maxlen]).
    inplace: kout / vout ARE kmem / vmem (updated in place: the captured acting graph's static state; a workgroup owns its head's
    columns and holds the rows in registers across a barrier).  The mask is a new tensor unless `done` (int32 [>= batch], zero; the
    kernel leaves it zero) is given with inplace: then the workgroup that arrives last writes it over state_mask."""
    _chk(qkvr, torch.float32, "qkvr"); _chk(kmem, torch.float32, "kmem"); _chk(vmem, torch.float32, "vmem"); _chk(b_nd, torch.float32, "b_nd")
    if state_mask.dtype == torch.bool:
        state_mask = state_mask.view(torch.uint8)
    if first.dtype == torch.bool:
        first = first.view(torch.uint8)
    _chk(state_mask, torch.uint8, "state_mask"); _chk(first, torch.uint8, "first")
    maxlen = kmem.shape[1]
    if qkvr.shape[0] != batch or maxlen > ATTENTION_STEP_MAXLEN or tuple(state_mask.shape) != (batch, maxlen) or first.numel() != batch:
        raise ValueError(f"masked_attention_step: one token per sequence, maxlen <= {ATTENTION_STEP_MAXLEN}, state_mask [batch, maxlen], first [batch] "
                         f"(got {qkvr.shape[0]} rows for batch {batch}, maxlen {maxlen}, mask {tuple(state_mask.shape)})")
    dt, fmt = _fmt(dtype=dtype)
    out = torch.empty(batch, hid, dtype=dt, device=qkvr.device)
    meta = dict(flops=4.0 * batch * maxlen * hid, bytes=16.0 * batch * maxlen * hid)
    if done is not None:
        if not inplace:
            raise ValueError("masked_attention_step: `done` is the in-place variant's counter")
        if done.dtype != torch.int32 or done.numel() < batch or not done.is_contiguous():
            raise ValueError("masked_attention_step: done must be a contiguous int32 tensor of at least `batch` zeros")
        _call("vpt_masked_attention_step_inplace", meta, ptr(qkvr), ptr(kmem), ptr(vmem), ptr(state_mask), ptr(first), ptr(b_nd), ptr(out), ptr(done),
              batch, heads, hid, qkvr.shape[1], maxlen, _stream(), fmt=fmt, label="vpt_masked_attention_step")
        return out, kmem, vmem, state_mask
    mout = torch.empty_like(state_mask)
    kout, vout = (kmem, vmem) if inplace else (torch.empty_like(kmem), torch.empty_like(vmem))
    _call("vpt_masked_attention_step", meta, ptr(qkvr), ptr(kmem), ptr(vmem), ptr(state_mask), ptr(first),
          ptr(b_nd), ptr(out), ptr(kout), ptr(vout), ptr(mout), batch, heads, hid, qkvr.shape[1], maxlen, _stream(), fmt=fmt)
    return out, kout, vout, mout


def new_rng_state(device, seed=None):
    """Device-resident state of the in-kernel generator: int64 [2] = {seed, step}.  The seed comes from torch's default CPU generator
    unless given, so torch.manual_seed() makes a run's sampled actions reproducible, as it does for the reference's th.rand_like."""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return torch.tensor([int(seed), 0], dtype=torch.int64, device=device)


def _chk_rng(rng_state):
    if rng_state is not None and (rng_state.dtype != torch.int64 or rng_state.numel() != 2 or not rng_state.is_cuda or not rng_state.is_contiguous()):
        raise ValueError("rng_state must be a contiguous int64 [2] GPU tensor {seed, step} (ops.new_rng_state)")


def uniform_noise(rng_state, rng_stream, m, n):
    """fp32 [m, n]: exactly the uniforms log_softmax_cols(..., rng=(rng_state, rng_stream)) draws (vpt_uniform_noise); state not advanced."""
    _chk_rng(rng_state)
    out = torch.empty(m, n, dtype=torch.float32, device=rng_state.device)
    _call("vpt_uniform_noise", dict(bytes=4.0 * m * n), ptr(rng_state), ctypes.c_uint32(rng_stream), ptr(out), m, n, _stream())
    return out


def act_epilogue(action_buttons, action_camera, logp_buttons, logp_camera, logits, value_col, scale, shift, rng_state=None):
    """Tail of MinecraftAgentPolicy.act on the acting path in one launch (include/vpt_hip.h: vpt_act_epilogue) ->
    (keep int64 [M, 4], nan_flag uint8 [1]); see unpack_act_keep().  rng_state: the generator state whose step counter this launch
    advances (the stochastic heads of the step drew from it)."""
    _chk_rng(rng_state)
    _chk(action_buttons, torch.int64, "action_buttons"); _chk(action_camera, torch.int64, "action_camera")
    _chk(logp_buttons, torch.float32, "logp_buttons"); _chk(logp_camera, torch.float32, "logp_camera"); _chk(logits, torch.float32, "logits")
    m = logits.shape[0]
    if m > 64 or action_buttons.numel() != m or action_camera.numel() != m or logp_buttons.numel() != m or logp_camera.numel() != m:
        raise ValueError("act_epilogue: one action / log-prob per row of logits, at most 64 rows")
    keep = torch.empty(m, 4, dtype=torch.int64, device=logits.device)
    flag = torch.empty(1, dtype=torch.uint8, device=logits.device)
    _call("vpt_act_epilogue", dict(bytes=64.0 * m), ptr(action_buttons), ptr(action_camera), ptr(logp_buttons), ptr(logp_camera), ptr(logits),
          logits.shape[1], int(value_col), ctypes.c_float(scale), ctypes.c_float(shift), ptr(keep), ptr(flag), ptr(rng_state), m, _stream())
    return keep, flag


def unpack_act_keep(keep):
    """Views (no kernels) of act_epilogue's packed record: (buttons int64 [M], camera int64 [M], log_prob fp32 [M], vpred de-normalised
    fp32 [M], raw vpred fp32 [M])."""
    kf = keep.view(torch.float32)       # [M, 8]
    return keep[:, 0], keep[:, 1], kf[:, 4], kf[:, 6], kf[:, 7]


def log_softmax_cols(logits, col0, n, temperature, mask=None, noise=None, want_action=False, rng=None):
    """log_softmax(logits[:, col0:col0+n] / T) -> fp32 [M, n]; mask uint8 [M, n] (0 -> LOG0).  want_action: also
    CategoricalActionHead.sample + logprob in the same kernel -> (lp, action int64 [M], action_logp fp32 [M]).  Stochastic
    (Gumbel-max) when uniforms are given -- noise fp32 [M, n], or rng = (rng_state, stream id): drawn inside the kernel from the
    device-resident generator state (new_rng_state; the caller advances its step) --, deterministic arg-max otherwise."""
    _chk(logits, torch.float32, "logits"); _chk(mask, torch.uint8, "mask"); _chk(noise, torch.float32, "noise")
    rng_state, rng_stream = rng if rng is not None else (None, 0)
    _chk_rng(rng_state)
    if noise is not None and rng_state is not None:
        raise ValueError("log_softmax_cols: give noise or rng, not both")
    m = logits.shape[0]
    out = torch.empty(m, n, dtype=torch.float32, device=logits.device)
    if mask is None and not want_action:
        _call("vpt_log_softmax_forward", dict(bytes=8.0 * m * n), ptr(logits), ptr(out), m, logits.shape[1], col0, n,
              ctypes.c_float(temperature), _stream())
        return out
    action = torch.empty(m, dtype=torch.int64, device=logits.device) if want_action else None
    alp = torch.empty(m, dtype=torch.float32, device=logits.device) if want_action else None
    _call("vpt_action_head_forward", dict(bytes=8.0 * m * n), ptr(logits), ptr(mask), ptr(noise), ptr(rng_state), ctypes.c_uint32(rng_stream),
          ptr(out), ptr(action), ptr(alp), m, logits.shape[1], col0, n, ctypes.c_float(temperature), _stream())
    return (out, action, alp) if want_action else out


def adam_step_(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    """In-place fused Adam on flat fp32 tensors (torch.optim.Adam semantics, L2 weight decay)."""
    for t, nme in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, torch.float32, nme)
    n = param.numel()
    assert grad.numel() == n and exp_avg.numel() == n and exp_avg_sq.numel() == n
    _call("vpt_adam_step", dict(bytes=28.0 * n), ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), ctypes.c_uint64(n), int(step),
          ctypes.c_float(lr), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps), ctypes.c_float(weight_decay),
          ctypes.c_float(grad_scale), _stream())


def adam_step_multi_(params, grads, exp_avgs, exp_avg_sqs, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0,
                     found_inf=None):
    """The Adam update of adam_step_ for a list of tensors in ONE launch (flat fp32 views; same arithmetic per element).
    found_inf: optional int32 [1] device tensor (loss-scaled fp16 step).  It is zeroed, set by vpt_grads_nonfinite_multi when any
    gradient is inf / nan, and read by the Adam launch, which then leaves every tensor untouched -- no host round trip."""
    import numpy as np
    n_t = len(params)
    if n_t == 0:
        return
    _chk(found_inf, torch.int32, "found_inf")
    rec = np.zeros(n_t, dtype=[("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<u8"), ("fb", "<i8")])
    blk = 0
    for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avgs, exp_avg_sqs)):
        for t, nme in ((p, "param"), (g, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
            _chk(t, torch.float32, nme)
            assert t.is_contiguous()
        n = p.numel()
        assert g.numel() == n and m.numel() == n and v.numel() == n
        rec[i] = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, blk)
        blk += (n + 1023) // 1024
    table = torch.from_numpy(rec.view(np.uint8)).to(params[0].device, non_blocking=False)
    if found_inf is not None:
        found_inf.zero_()
        _call("vpt_grads_nonfinite_multi", dict(bytes=4.0 * sum(p.numel() for p in params)), ptr(table), n_t, blk, ptr(found_inf), _stream())
    _call("vpt_adam_step_multi", dict(bytes=28.0 * sum(p.numel() for p in params)), ptr(table), n_t, blk, int(step),
          ctypes.c_float(lr), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps), ctypes.c_float(weight_decay),
          ctypes.c_float(grad_scale), ptr(found_inf), _stream())
    return table   # keep alive until the launch has been enqueued (the caller may drop it afterwards: stream-ordered free)


def grads_nonfinite(grads):
    """int32 [1] device flag: 1 if any element of the fp32 tensors in `grads` is inf / nan (vpt_grads_nonfinite_multi, one launch; what
    torch.cuda.amp.GradScaler.unscale_ computes).  No host synchronisation here: the caller reads the flag when it needs it."""
    import numpy as np
    grads = [g if g.is_contiguous() else g.contiguous() for g in grads if g is not None and g.numel()]
    if not grads:
        raise ValueError("grads_nonfinite: no gradient tensors to check")
    flag = torch.zeros(1, dtype=torch.int32, device=grads[0].device)
    rec = np.zeros(len(grads), dtype=[("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<u8"), ("fb", "<i8")])
    blk = 0
    for i, g in enumerate(grads):
        _chk(g, torch.float32, "grad")
        rec[i] = (0, g.data_ptr(), 0, 0, g.numel(), blk)
        blk += (g.numel() + 1023) // 1024
    table = torch.from_numpy(rec.view(np.uint8)).to(grads[0].device, non_blocking=False)
    _call("vpt_grads_nonfinite_multi", dict(bytes=4.0 * sum(g.numel() for g in grads)), ptr(table), len(grads), blk, ptr(flag), _stream())
    return flag


# ---- backward (behavioural-cloning step) ---------------------------------------------------------------
def nll_backward(lp_buttons, lp_camera, act_buttons, act_camera, ldz, scale, dtype=torch.bfloat16):
    """16-bit [M, ldz] gradient of the BC loss w.r.t. the fused head logits (value column and padding zero).  `scale` carries
    1 / (frames x temperature) and, in the fp16 mode, the loss scale (training.BCTrainer)."""
    _chk(lp_buttons, torch.float32, "lp_buttons"); _chk(lp_camera, torch.float32, "lp_camera")
    _chk(act_buttons, torch.int64, "act_buttons"); _chk(act_camera, torch.int64, "act_camera")
    m, nb = lp_buttons.shape
    nc = lp_camera.shape[1]
    dt, fmt = _fmt(dtype=dtype)
    dz = torch.empty(m, ldz, dtype=dt, device=lp_buttons.device)
    _call("vpt_bc_nll_backward", dict(bytes=6.0 * m * ldz), ptr(lp_buttons), ptr(lp_camera), ptr(act_buttons), ptr(act_camera),
          ptr(dz), m, nb, nc, ldz, ctypes.c_float(scale), _stream(), fmt=fmt)
    return dz


def heads_logprob_backward(lp_buttons, lp_camera, g_buttons, g_camera, g_value, ldz, temperature, mask_buttons=None, mask_camera=None,
                           dtype=torch.bfloat16, grad_scale=1.0):
    """16-bit [M, ldz] gradient w.r.t. the fused head logits for arbitrary incoming gradients of the two log-prob tensors and of the
    raw value output (any of them None = zero).  The autograd boundary of lib/policy.py uses this; the BC fast path uses nll_backward."""
    _chk(lp_buttons, torch.float32, "lp_buttons"); _chk(lp_camera, torch.float32, "lp_camera")
    _chk(g_buttons, torch.float32, "g_buttons"); _chk(g_camera, torch.float32, "g_camera"); _chk(g_value, torch.float32, "g_value")
    _chk(mask_buttons, torch.uint8, "mask_buttons"); _chk(mask_camera, torch.uint8, "mask_camera")
    m, nb = lp_buttons.shape
    nc = lp_camera.shape[1]
    dt, fmt = _fmt(dtype=dtype)
    dz = torch.empty(m, ldz, dtype=dt, device=lp_buttons.device)
    # grad_scale (fp16 loss scaling): d/dz is linear in the incoming gradients and they enter as g / temperature
    _call("vpt_heads_logprob_backward", dict(bytes=10.0 * m * ldz), ptr(lp_buttons), ptr(lp_camera), ptr(g_buttons), ptr(g_camera), ptr(g_value),
          ptr(mask_buttons), ptr(mask_camera), ptr(dz), m, nb, nc, ldz, ctypes.c_float(temperature), ctypes.c_float(grad_scale), _stream(), fmt=fmt)
    return dz


def layernorm_backward(x, gain, dy, dgain, dbias, relu_in=False, dx_add=None):
    for t, nme in ((x, "x"), (gain, "gain"), (dy, "dy"), (dgain, "dgain"), (dbias, "dbias"), (dx_add, "dx_add")):
        _chk(t, torch.float32, nme)
    m, d = x.shape
    dx = torch.empty_like(x)
    part = _workspace(WS_LAYERNORM_BACKWARD, m, d, device=x.device)
    _call("vpt_layernorm_backward", dict(bytes=20.0 * m * d), ptr(x), ptr(gain), ptr(dy), ptr(dx_add), ptr(dx), ptr(dgain), ptr(dbias), ptr(part),
          m, d, 1 if relu_in else 0, _stream())
    return dx


def gate_cast(x, ldo, mask=None, dtype=torch.bfloat16):
    """fp32 [M, N] -> 16-bit [M, ldo] (zero padded), zeroed where mask <= 0."""
    _chk(x, torch.float32, "x"); _chk(mask, OP16, "mask")
    m, n = x.shape
    dt, fmt = _fmt(mask, dtype=dtype if mask is None else None)
    out = torch.empty(m, ldo, dtype=dt, device=x.device)
    _call("vpt_gate_cast", dict(bytes=6.0 * m * ldo), ptr(x), ptr(mask), ptr(out), m, n, n, mask.shape[1] if mask is not None else 0, ldo, _stream(), fmt=fmt)
    return out


def column_sum_(out, x_bf16, n):
    _chk(x_bf16, OP16, "x"); _chk(out, torch.float32, "out")
    part = _workspace(WS_COLUMN_SUM, x_bf16.shape[0], n, device=out.device)
    _call("vpt_column_sum", dict(bytes=2.0 * x_bf16.numel()), ptr(x_bf16), ptr(out), ptr(part), x_bf16.shape[0], n, x_bf16.shape[1], _stream(), fmt=_fmt(x_bf16)[1])


def masked_attention_backward(qkvr, kmem, vmem, memvalid, b_nd, dout, db_nd, batch, t, heads, hid):
    for tt, nme in ((qkvr, "qkvr"), (kmem, "kmem"), (vmem, "vmem"), (b_nd, "b_nd"), (dout, "dout"), (db_nd, "db_nd")):
        _chk(tt, torch.float32, nme)
    _chk(memvalid, torch.uint8, "memvalid")
    dqkvr = torch.empty_like(qkvr) if qkvr.shape[1] == 3 * hid + 10 * heads else torch.zeros_like(qkvr)     # (every projection column is written)
    dkv = _workspace(WS_ATTENTION_BACKWARD_DKV, batch, t, hid, device=qkvr.device)
    dbnd = _workspace(WS_ATTENTION_BACKWARD_DBND, batch, t, heads, kmem.shape[1], device=qkvr.device)
    _call("vpt_masked_attention_backward", dict(flops=10.0 * batch * t * (t + kmem.shape[1]) * hid), ptr(qkvr), ptr(kmem), ptr(vmem),
          ptr(memvalid), ptr(b_nd), ptr(dout), ptr(dqkvr), ptr(db_nd), ptr(dkv), ptr(dbnd), batch, t, heads, hid, qkvr.shape[1], kmem.shape[1], _stream())
    return dqkvr


def conv_backward_prepare(dy, y, res, stats_in, edge_sa, edge_sg, cin, dpooled=None, argmax=None, d_sa=None, d_sg=None, want_t12=False):
    """-> (dacc bf16 blocked, coef fp32 [F,2], d_sa fp32 [9,CoutPad], d_sg fp32 [9,CoutPad][, t12 double [F,2]]).
    coef = (c0, c1) of the statistics terms conv3x3_dgrad adds (dx += c0 + c1 x).  d_sa / d_sg are accumulated into when
    given.  dy=None with (dpooled, argmax): the layer feeds a max-pool whose backward is applied on the fly."""
    for t, nme in ((dy, "dy"), (y, "y"), (res, "res"), (dpooled, "dpooled")):
        _chk(t, OP16, nme)
    _chk(argmax, torch.uint8, "argmax")
    _chk(stats_in, torch.float64, "stats_in"); _chk(edge_sa, torch.float32, "edge_sa"); _chk(edge_sg, torch.float32, "edge_sg")
    _chk(d_sa, torch.float32, "d_sa"); _chk(d_sg, torch.float32, "d_sg")
    if (dy is None) == (dpooled is None or argmax is None):
        raise ValueError("conv_backward_prepare: give either dy or (dpooled, argmax)")
    f, cb, h, w, _ = y.shape
    dev = y.device
    dacc = torch.empty_like(y)
    coef = torch.empty(f, 2, dtype=torch.float32, device=dev)
    t12 = torch.empty(f, 2, dtype=torch.float64, device=dev) if want_t12 else None
    if d_sa is None:
        d_sa, d_sg = torch.zeros_like(edge_sa), torch.zeros_like(edge_sg)
    scratch = _workspace(WS_CONV_BACKWARD_PREPARE, f, 0, 0, 0, cb * 32, device=dev)
    _call("vpt_conv_backward_prepare", dict(bytes=(6.0 if dy is not None else 4.75) * y.numel() + (2.0 * y.numel() if res is not None else 0)),
          ptr(dy), ptr(dpooled), ptr(argmax), ptr(y), ptr(res), ptr(stats_in), ptr(edge_sa), ptr(edge_sg),
          ptr(dacc), ptr(t12), ptr(coef), ptr(d_sa), ptr(d_sg), ptr(scratch), f, h, w, cin, cb * 32, _stream(), fmt=_fmt(dy, y, res, dpooled)[1])
    return (dacc, coef, d_sa, d_sg, t12) if want_t12 else (dacc, coef, d_sa, d_sg)


def conv3x3_dgrad(dacc, wpk_t, cin, skip=None, xin=None, coef=None):
    """dacc blocked [F,Cout/32,H,W,32] -> dx blocked [F,cin/32,H,W,32] (transposed conv + skip + c0 + c1*xin)."""
    _chk(dacc, OP16, "dacc"); _chk(wpk_t, OP16, "wpk_t"); _chk(skip, OP16, "skip")
    _chk(xin, OP16, "xin"); _chk(coef, torch.float32, "coef")
    f, cb, h, w, _ = dacc.shape
    dt, fmt = _fmt(dacc, wpk_t, skip, xin)
    dx = torch.empty(f, cin // 32, h, w, 32, dtype=dt, device=dacc.device)
    _call("vpt_conv3x3_dgrad", dict(flops=2.0 * f * h * w * cin * 9 * cb * 32), ptr(dacc), ptr(wpk_t), ptr(skip), ptr(xin), ptr(coef), ptr(dx),
          f, h, w, cb * 32, cin, _stream(), fmt=fmt)
    return dx


def conv3x3_dgrad_gated(dacc, wpk_t, cin, xin, coef, gate_stats, gate_cin):
    """A block's conv1 -> conv0 backward in one epilogue (vpt_conv3x3_dgrad_gated): the transposed convolution of `dacc` (conv1's operand) whose
    output is conv0's OPERAND  rstd0 * (conv^T + c0 + c1 xin) * [xin > 0]  (xin = conv0's output, gate_stats = the frame statistics of conv0's
    input over gate_cin * H * W elements) instead of the plain gradient.  -> (dacc0 blocked [F, cin/32, H, W, 32], gate_u fp64 [F])."""
    _chk(dacc, OP16, "dacc"); _chk(wpk_t, OP16, "wpk_t"); _chk(xin, OP16, "xin"); _chk(coef, torch.float32, "coef"); _chk(gate_stats, torch.float64, "gate_stats")
    f, cb, h, w, _ = dacc.shape
    dt, fmt = _fmt(dacc, wpk_t, xin)
    out = torch.empty(f, cin // 32, h, w, 32, dtype=dt, device=dacc.device)
    gate_u = torch.zeros(f, dtype=torch.float64, device=dacc.device)
    _call("vpt_conv3x3_dgrad_gated", dict(flops=2.0 * f * h * w * cin * 9 * cb * 32), ptr(dacc), ptr(wpk_t), ptr(xin), ptr(coef), ptr(gate_stats), int(gate_cin),
          ptr(out), ptr(gate_u), f, h, w, cb * 32, cin, _stream(), fmt=fmt, label="vpt_conv3x3_dgrad")
    return out, gate_u


def conv_backward_reduce(dacc, gate_u, stats_in, edge_sa, edge_sg, cin, d_sa=None, d_sg=None, want_t12=False):
    """What conv_backward_prepare returns besides dacc, for an operand that conv3x3_dgrad_gated already wrote: (coef fp32 [F,2], d_sa, d_sg[, t12])
    from one read of dacc (vpt_conv_backward_reduce).  stats_in: the layer's input statistics (cin * H * W elements)."""
    _chk(dacc, OP16, "dacc"); _chk(gate_u, torch.float64, "gate_u"); _chk(stats_in, torch.float64, "stats_in")
    _chk(edge_sa, torch.float32, "edge_sa"); _chk(edge_sg, torch.float32, "edge_sg"); _chk(d_sa, torch.float32, "d_sa"); _chk(d_sg, torch.float32, "d_sg")
    f, cb, h, w, _ = dacc.shape
    dev = dacc.device
    coef = torch.empty(f, 2, dtype=torch.float32, device=dev)
    t12 = torch.empty(f, 2, dtype=torch.float64, device=dev) if want_t12 else None
    if d_sa is None:
        d_sa, d_sg = torch.zeros_like(edge_sa), torch.zeros_like(edge_sg)
    scratch = _workspace(WS_CONV_BACKWARD_PREPARE, f, 0, 0, 0, cb * 32, device=dev)
    _call("vpt_conv_backward_reduce", dict(bytes=2.0 * dacc.numel()), ptr(dacc), ptr(gate_u), ptr(stats_in), ptr(edge_sa), ptr(edge_sg),
          ptr(t12), ptr(coef), ptr(d_sa), ptr(d_sg), ptr(scratch), f, h, w, cin, cb * 32, _stream(), fmt=_fmt(dacc)[1], label="vpt_conv_backward_prepare")
    return (coef, d_sa, d_sg, t12) if want_t12 else (coef, d_sa, d_sg)


def maxpool_backward(pre, pooled, dpooled):
    for t, nme in ((pre, "pre"), (pooled, "pooled"), (dpooled, "dpooled")):
        _chk(t, OP16, nme)
    f, cb, h, w, _ = pre.shape
    dpre = torch.empty_like(pre)
    _call("vpt_maxpool_backward", dict(bytes=5.0 * pre.numel()), ptr(pre), ptr(pooled), ptr(dpooled), ptr(dpre), f, cb * 32, h, w, _stream(), fmt=_fmt(pre, pooled, dpooled)[1])
    return dpre


def frame_affine_backward_reduce(x, dy, gain, stats_in, dgain, dbias):
    """Pass 1 of frame_affine_backward alone (per-channel gain): dgain / dbias accumulated in place, returns ab fp64 [F,2] = (sum dy g, sum dy g xhat)
    for a consumer that applies dx = rstd (dy g - ab0/n - xhat ab1/n) itself (conv_backward_prepare_pooled(nfold=...))."""
    _chk(x, OP16, "x"); _chk(dy, OP16, "dy"); _chk(gain, torch.float32, "gain"); _chk(stats_in, torch.float64, "stats_in")
    _chk(dgain, torch.float32, "dgain"); _chk(dbias, torch.float32, "dbias")
    f, cb, h, w, _ = x.shape
    ab = torch.zeros(f, 2, dtype=torch.float64, device=x.device)
    part = _workspace(WS_FRAME_AFFINE_BACKWARD, f, h * w, 0, 1, cb * 32, device=x.device)
    _call("vpt_frame_affine_backward", dict(bytes=4.0 * x.numel()), ptr(x), ptr(dy), None, None, ptr(gain), ptr(stats_in), ptr(ab), ptr(dgain), ptr(dbias), ptr(part),
          f, cb * 32, h * w, 0, 1, _stream(), fmt=_fmt(x, dy)[1])
    return ab


def frame_affine_backward(x, dy, gain, stats_in, dgain, dbias, per_element=False, dx_add=None):
    """Backward of frame_affine: returns dx (bf16 blocked); dgain / dbias accumulated in place."""
    _chk(x, OP16, "x"); _chk(dy, OP16, "dy"); _chk(dx_add, OP16, "dx_add")
    _chk(gain, torch.float32, "gain"); _chk(stats_in, torch.float64, "stats_in"); _chk(dgain, torch.float32, "dgain"); _chk(dbias, torch.float32, "dbias")
    f, cb, h, w, _ = x.shape
    ab = torch.zeros(f, 2, dtype=torch.float64, device=x.device)
    dx = torch.empty_like(x)
    pe = 1 if per_element else 0
    head = (ptr(x), ptr(dy), ptr(dx_add), ptr(dx), ptr(gain), ptr(stats_in), ptr(ab), ptr(dgain), ptr(dbias))
    tail = (f, cb * 32, h * w, pe)
    fmt = _fmt(x, dy, dx_add)[1]
    part1 = _workspace(WS_FRAME_AFFINE_BACKWARD, f, h * w, pe, 1, cb * 32, device=x.device)      # (None for the per-element variant: pass 3 reduces)
    _call("vpt_frame_affine_backward", dict(bytes=4.0 * x.numel()), *head, ptr(part1), *tail, 1, _stream(), fmt=fmt)
    _call("vpt_frame_affine_backward", dict(bytes=6.0 * x.numel()), *head, None, *tail, 2, _stream(), fmt=fmt)
    if per_element:
        part3 = _workspace(WS_FRAME_AFFINE_BACKWARD, f, h * w, pe, 3, cb * 32, device=x.device)
        _call("vpt_frame_affine_backward", dict(bytes=4.0 * x.numel()), *head, ptr(part3), *tail, 3, _stream(), fmt=fmt)
    return dx


def conv3x3_wgrad(dacc, x, out=None):
    """-> fp32 [Cout, 9, Cin]: sum over frames and pixels of dacc (x) shifted x (added to `out` when given)."""
    _chk(dacc, OP16, "dacc"); _chk(x, OP16, "x"); _chk(out, torch.float32, "out")
    f, cbo, h, w, _ = dacc.shape
    cbi = x.shape[1]
    dw = out if out is not None else torch.zeros(cbo * 32, 9, cbi * 32, dtype=torch.float32, device=x.device)
    fmt = _fmt(dacc, x)[1]
    scratch = torch.empty(_native.load(fmt).vpt_conv3x3_wgrad_scratch_floats(f, cbi * 32, cbo * 32), dtype=torch.float32, device=x.device)
    _call("vpt_conv3x3_wgrad", dict(flops=2.0 * f * h * w * cbo * 32 * 9 * cbi * 32), ptr(dacc), ptr(x), ptr(dw), ptr(scratch), f, h, w, cbi * 32, cbo * 32, _stream(), fmt=fmt)
    return dw


def conv_first_backward(img_u8, wfrag, dpooled, cout, out=None, nfold=None):
    """-> (dW fp32 [cout, 27] in (kh, kw, ch) tap order, db fp32 [cout]); accumulated into out=(dW, db) when given.
    conv_first_grad_to_reference() maps dW to the reference's [cout, 3, 3, 3] (o, ch, kh, kw).
    nfold = (n_gain fp32 [cout], pool_stats fp64 [F,2], ab fp64 [F,2] from frame_affine_backward_reduce): `dpooled` is the gradient w.r.t. n(pooled) and the
    GroupNorm `n` backward is applied on the fly (vpt_conv_first_backward_nfold: no second affine-backward pass over stack 0)."""
    _chk(img_u8, torch.uint8, "img"); _chk(wfrag, OP16, "wfrag"); _chk(dpooled, OP16, "dpooled")
    f, h, w, _ = img_u8.shape
    if out is None:
        out = (torch.zeros(cout, 27, dtype=torch.float32, device=img_u8.device), torch.zeros(cout, dtype=torch.float32, device=img_u8.device))
    dw, db = out
    _chk(dw, torch.float32, "dw"); _chk(db, torch.float32, "db")
    part = _workspace(WS_CONV_FIRST_BACKWARD, f, h, w, 0, cout, device=img_u8.device)
    if nfold is not None:
        ng, pst, pab = nfold
        _chk(ng, torch.float32, "n_gain"); _chk(pst, torch.float64, "pool_stats"); _chk(pab, torch.float64, "pool_ab")
        _call("vpt_conv_first_backward_nfold", dict(flops=2.0 * f * (h // 2) * (w // 2) * cout * 27), ptr(img_u8), ptr(wfrag), ptr(dpooled), ptr(ng), ptr(pst), ptr(pab),
              ptr(dw), ptr(db), ptr(part), f, h, w, cout, _stream(), fmt=_fmt(wfrag, dpooled)[1], label="vpt_conv_first_backward")
        return dw, db
    _call("vpt_conv_first_backward", dict(flops=2.0 * f * (h // 2) * (w // 2) * cout * 27), ptr(img_u8), ptr(wfrag), ptr(dpooled), ptr(dw), ptr(db), ptr(part),
          f, h, w, cout, _stream(), fmt=_fmt(wfrag, dpooled)[1])
    return dw, db


def conv_first_grad_to_reference(dw):
    cout = dw.shape[0]
    return dw.view(cout, 3, 3, 3).permute(0, 3, 1, 2).contiguous()


# ---- action codec (lib/actions.py, lib/action_mapping.py on the device) ----------------------------------------
def camera_discretize(xy, maxval, binsize, mu, mu_law):
    """fp64 tensor (any shape) -> int64 bins of the same shape."""
    _chk(xy, torch.float64, "xy")
    out = torch.empty(xy.shape, dtype=torch.int64, device=xy.device)
    if xy.numel():
        _call("vpt_camera_discretize", dict(bytes=16.0 * xy.numel()), ptr(xy), ptr(out), xy.numel(), float(maxval), float(binsize), float(mu), int(bool(mu_law)), _stream())
    return out


def camera_undiscretize(bins, maxval, binsize, mu, mu_law):
    """int64 bins -> fp64 camera angles."""
    _chk(bins, torch.int64, "bins")
    out = torch.empty(bins.shape, dtype=torch.float64, device=bins.device)
    if bins.numel():
        _call("vpt_camera_undiscretize", dict(bytes=16.0 * bins.numel()), ptr(bins), ptr(out), bins.numel(), float(maxval), float(binsize), float(mu), int(bool(mu_law)), _stream())
    return out


def action_from_factored(buttons, camera, n_camera_bins=11):
    """buttons int64 [N,20], camera int64 [N,2] -> (joint buttons int64 [N], joint camera int64 [N])."""
    _chk(buttons, torch.int64, "buttons"); _chk(camera, torch.int64, "camera")
    n = buttons.shape[0]
    jb = torch.empty(n, dtype=torch.int64, device=buttons.device)
    jc = torch.empty(n, dtype=torch.int64, device=buttons.device)
    if n:
        # camera bins outside 0..n_bins-1 are a KeyError in the reference's table lookup (buttons are truthy flags there: any value)
        if int(camera.min()) < 0 or int(camera.max()) >= n_camera_bins:
            raise IndexError(f"factored action out of range: camera bins must lie in 0..{n_camera_bins - 1}")
        _call("vpt_action_from_factored", dict(bytes=192.0 * n), ptr(buttons), ptr(camera), ptr(jb), ptr(jc), n, int(n_camera_bins), _stream())
    return jb, jc


def action_to_factored(joint_buttons, joint_camera, n_camera_bins=11):
    """joint indices int64 [N] -> (buttons int64 [N,20], camera int64 [N,2])."""
    _chk(joint_buttons, torch.int64, "joint_buttons"); _chk(joint_camera, torch.int64, "joint_camera")
    n = joint_buttons.shape[0]
    b = torch.empty(n, 20, dtype=torch.int64, device=joint_buttons.device)
    c = torch.empty(n, 2, dtype=torch.int64, device=joint_buttons.device)
    if n:
        # the kernel decodes blindly; the reference raises IndexError / KeyError on indices outside its tables
        lo_b, hi_b = int(joint_buttons.min()), int(joint_buttons.max())
        lo_c, hi_c = int(joint_camera.min()), int(joint_camera.max())
        if lo_b < 0 or hi_b > 8640 or lo_c < 0 or hi_c >= n_camera_bins * n_camera_bins:
            raise IndexError(f"joint action index out of range: buttons [{lo_b}, {hi_b}] (0..8640), camera [{lo_c}, {hi_c}] (0..{n_camera_bins * n_camera_bins - 1})")
        _call("vpt_action_to_factored", dict(bytes=192.0 * n), ptr(joint_buttons), ptr(joint_camera), ptr(b), ptr(c), n, int(n_camera_bins), _stream())
    return b, c


# ---- clip data path (data_loader.py:113-122 on the device) --------------------------------------------------------
def clip_frames(frames_bgr, cursor_state=None, cursor_bgr=None, cursor_alpha=None, out_hw=(128, 128), out=None):
    """uint8 BGR [F,H,W,3] (+ int32 [F,3] (gui open, x, y), uint8 cursor [h,w,3] BGR, fp64 alpha [h,w]) -> uint8 RGB [F,oh,ow,3]:
    cursor compositing, BGR->RGB and the cv2.INTER_LINEAR resize in one launch.  The compositing is bit-identical to the live
    reference's function (golden vectors); the resize restates OpenCV's 8-bit fixed-point algorithm and is bit-identical to
    oracle/clip_oracle.py, which is NOT pinned against real cv2 (absent from this image)."""
    _chk(frames_bgr, torch.uint8, "frames_bgr")
    if frames_bgr.dim() != 4 or frames_bgr.shape[3] != 3:
        raise ValueError("frames_bgr must be [F, H, W, 3]")
    f, h, w, _ = frames_bgr.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    if out is None:
        out = torch.empty(f, oh, ow, 3, dtype=torch.uint8, device=frames_bgr.device)
    else:
        _chk(out, torch.uint8, "out")
        if tuple(out.shape) != (f, oh, ow, 3) or out.device != frames_bgr.device:
            raise ValueError(f"clip_frames: out must be uint8 [{f}, {oh}, {ow}, 3] on {frames_bgr.device}, got {tuple(out.shape)} on {out.device}")
    ch = cw = 0
    if cursor_state is not None:
        if cursor_bgr is None or cursor_alpha is None:
            raise ValueError("cursor_state needs cursor_bgr and cursor_alpha")
        _chk(cursor_state, torch.int32, "cursor_state"); _chk(cursor_bgr, torch.uint8, "cursor_bgr"); _chk(cursor_alpha, torch.float64, "cursor_alpha")
        if tuple(cursor_state.shape) != (f, 3) or cursor_bgr.dim() != 3 or cursor_bgr.shape[2] != 3 or tuple(cursor_alpha.shape) != tuple(cursor_bgr.shape[:2]):
            raise ValueError("cursor_state must be [F, 3], cursor_bgr [h, w, 3], cursor_alpha [h, w]")
        ch, cw = int(cursor_bgr.shape[0]), int(cursor_bgr.shape[1])
    if f:
        _call("vpt_clip_frames", dict(bytes=float(f) * (h * w * 3 + oh * ow * 3)), ptr(frames_bgr), f, h, w, ptr(cursor_state), ptr(cursor_bgr), ptr(cursor_alpha),
              ch, cw, ptr(out), oh, ow, _stream())
    return out
