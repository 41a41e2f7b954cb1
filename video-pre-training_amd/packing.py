"""Host-side weight re-packing into the layouts the gfx950 kernels stream (see DESIGN.md §Data layout).

All functions take fp32 torch tensors with the reference's shapes (the `.weights` state_dict,
SURVEY.md §8b) and return new tensors on the same device; the fp32 masters are left untouched.
"""
import math

import torch


def _ceil_div(a, b):
    return (a + b - 1) // b


def pack_conv3x3(weight, gain, bias, tables=True, dtype=torch.bfloat16):
    """Conv2d weight [Cout,Cin,3,3] with the preceding GroupNorm(1,Cin) affine (gain, bias [Cin]) folded.

    Returns (wpk bf16 [NT][Cin/32][9][128][32], edge_sa fp32 [9][NT*128], edge_sg fp32 [9][NT*128]).
    wpk holds bf16(W * gain), each 64-byte row (one cout, 32 cin) stored with its four 16-byte chunks
    XOR-swizzled by ((cout >> 2) & 3): the tile is DMA'd to LDS verbatim and read conflict-free; edge_sg[e][o] = sum over the taps valid for edge class e and over Cin of that
    rounded value; edge_sa[e][o] = same sum of W * bias (fp32).  e = 3*ey + ex with ey/ex in
    {0: first row/col, 1: interior, 2: last row/col} (vpt_conv3x3.hip epilogue)."""
    cout, cin = weight.shape[:2]
    assert cin % 32 == 0 and cout % 32 == 0
    nt = _ceil_div(cout, 128)
    cp = nt * 128
    wg = (weight * gain.view(1, -1, 1, 1)).to(dtype)
    wp = torch.zeros(cp, cin, 3, 3, dtype=dtype, device=weight.device)
    wp[:cout] = wg
    wpk = wp.view(nt, 128, cin // 32, 32, 9).permute(0, 2, 4, 1, 3).contiguous()
    wpk = swizzle_rows64(wpk)
    if not tables:
        return wpk, None, None
    sg_tap = torch.zeros(cp, 9, dtype=torch.float64, device=weight.device)
    sa_tap = torch.zeros(cp, 9, dtype=torch.float64, device=weight.device)
    sg_tap[:cout] = wg.double().sum(dim=1).view(cout, 9)
    sa_tap[:cout] = (weight.double() * bias.double().view(1, -1, 1, 1)).sum(dim=1).view(cout, 9)
    m = edge_tap_matrix(weight.device)                    # [9 edge classes, 9 taps]
    return wpk, (m @ sa_tap.t()).float().contiguous(), (m @ sg_tap.t()).float().contiguous()


_EDGE_TAP = {}


def edge_tap_matrix(device, dtype=torch.float64):
    """0/1 matrix [9 edge classes e = 3*ey + ex, 9 taps kh*3 + kw]: the tap reads inside the image for pixels of that class."""
    key = (str(device), dtype)
    if key not in _EDGE_TAP:
        valid = {0: [1, 2], 1: [0, 1, 2], 2: [0, 1]}
        m = torch.zeros(9, 9, dtype=dtype)
        for ey in range(3):
            for ex in range(3):
                for kh in valid[ey]:
                    for kw in valid[ex]:
                        m[ey * 3 + ex, kh * 3 + kw] = 1
        _EDGE_TAP[key] = m.to(device)
    return _EDGE_TAP[key]


def pack_conv3x3_dgrad(weight, gain, dtype=torch.bfloat16):
    """Weights of the input-gradient convolution of a normed conv layer: Wd[c, o, kh, kw] = op16(W * gain)[o, c, 2-kh, 2-kw]
    in the same packed format (a conv with Cin' = Cout, Cout' = Cin, no further gain)."""
    wg = (weight * gain.view(1, -1, 1, 1)).to(dtype).float()
    wd = wg.permute(1, 0, 2, 3).flip(2, 3).contiguous()
    wpk, _, _ = pack_conv3x3(wd, torch.ones(wd.shape[1], device=weight.device), None, tables=False, dtype=dtype)
    return wpk


def swizzle_rows64(t):
    """[..., rows, 32] bf16 -> same shape with chunk c (8 elements) of row r stored at chunk c ^ ((r >> 2) & 3).
    The permutation is an involution, so applying it twice restores the logical order."""
    rows = t.shape[-2]
    v = t.reshape(*t.shape[:-1], 4, 8)
    r = torch.arange(rows, device=t.device)
    idx = (torch.arange(4, device=t.device).view(1, 4) ^ ((r >> 2) & 3).view(rows, 1))  # [rows, 4]: source chunk per slot
    idx = idx.view(*([1] * (v.dim() - 3)), rows, 4, 1).expand(*v.shape)
    return torch.gather(v, -2, idx).reshape(t.shape).contiguous()


# K slot -> k of the first conv's two MFMA k-steps (vpt_conv_first_tile.h): values 0..7 of kernel row 0, 1, 2 (each one 16-byte
# read of the 16-bit input tile), then the ninth value of the three rows, the two bias slots and three zero slots.
CONV_FIRST_SLOT_K = [9 * (s >> 3) + (s & 7) for s in range(24)] + [8, 17, 26, 27, 28, 29, 30, 31]


def pack_conv_first(weight, bias, dtype=torch.bfloat16):
    """Stack-0 firstconv weight [Cout,3,3,3] + bias [Cout] -> MFMA A-operand fragments
    bf16 [NT][4][2][64][8] (vpt_conv_first.hip).  k = (kh*3+kw)*3 + ch for k < 27 holds W / 255 (the pixel operand is
    the raw byte 0..255); k = 27 / 28 carry the hi / lo bf16 halves of the bias (the pixel operand holds 1.0 there);
    the 32 values of a row are stored in the slot order CONV_FIRST_SLOT_K."""
    cout = weight.shape[0]
    assert weight.shape[1:] == (3, 3, 3) and cout % 32 == 0
    nt = _ceil_div(cout, 128)
    cp = nt * 128
    wk = torch.zeros(cp, 32, dtype=torch.float32, device=weight.device)
    wk[:cout, :27] = weight.permute(0, 2, 3, 1).reshape(cout, 27) / 255.0
    hi = bias.to(dtype).float()
    lo = (bias - hi).to(dtype).float()
    wk[:cout, 27] = hi
    wk[:cout, 28] = lo
    wk = wk[:, torch.tensor(CONV_FIRST_SLOT_K, device=wk.device)].contiguous()
    # [nt][cs][l31][ks][hi][e] -> [nt][cs][ks][hi][l31][e]
    frag = wk.view(nt, 4, 32, 2, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous().view(nt, 4, 2, 64, 8)
    return frag.to(dtype).contiguous()


def pack_conv3d_t5(weight, bias, dtype=torch.bfloat16):
    """IDM Conv3d weight [O,3,5,1,1] + bias [O] -> (MFMA A-operand fragments bf16 [NT][4][64][8], bias fp32 [NT*128]).
    k = dt*3 + ch for k < 15 (dt = temporal tap 0..4), k = 15 is zero padding (vpt_conv3d.hip)."""
    o = weight.shape[0]
    assert tuple(weight.shape[1:]) == (3, 5, 1, 1) and o % 32 == 0
    nt = _ceil_div(o, 128)
    cp = nt * 128
    wk = torch.zeros(cp, 16, dtype=torch.float32, device=weight.device)
    wk[:o, :15] = weight.reshape(o, 3, 5).permute(0, 2, 1).reshape(o, 15)
    frag = wk.view(nt, 4, 32, 2, 8).permute(0, 1, 3, 2, 4).contiguous().view(nt, 4, 64, 8)
    bp = torch.zeros(cp, dtype=torch.float32, device=weight.device)
    bp[:o] = bias
    return frag.to(dtype).contiguous(), bp


def pack_linear(weight, dtype=torch.bfloat16):
    """nn.Linear weight [N,K] -> bf16 [ceil(N/128)][K/32][128][32] (vpt_gemm.hip B operand)."""
    n, k = weight.shape
    assert k % 64 == 0
    nt = _ceil_div(n, 128)
    wp = torch.zeros(nt * 128, k, dtype=dtype, device=weight.device)
    wp[:n] = weight.to(dtype)
    return wp.view(nt, 128, k // 32, 32).permute(0, 2, 1, 3).contiguous()


def chw_to_blocked_columns(weight, c, h, w):
    """Permute the K axis of a [N, c*h*w] matrix from the reference's C,H,W flatten order
    (lib/impala_cnn.py:192-193) to the blocked activation order [c/32][h][w][32]."""
    n = weight.shape[0]
    return weight.view(n, c // 32, 32, h, w).permute(0, 1, 3, 4, 2).reshape(n, c * h * w).contiguous()


def chw_to_blocked_vector(v, c, h, w):
    return v.view(c // 32, 32, h, w).permute(0, 2, 3, 1).reshape(-1).contiguous()


def blocked_to_nchw(x_blocked, c, h, w):
    """bf16 [F][c/32][h][w][32] -> fp32 [F][c][h][w] (tests / debugging only)."""
    f = x_blocked.shape[0]
    return x_blocked.view(f, c // 32, h, w, 32).permute(0, 1, 4, 2, 3).reshape(f, c, h, w).float()


def nchw_to_blocked(x, dtype=torch.bfloat16):
    f, c, h, w = x.shape
    return x.view(f, c // 32, 32, h, w).permute(0, 1, 3, 4, 2).contiguous().to(dtype)
