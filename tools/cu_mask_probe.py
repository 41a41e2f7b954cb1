"""Does partitioning the CUs between the MFMA-bound convolution and an HBM-bound pass pay on this chip?  (round 4 probe)
Two streams created with hipExtStreamCreateWithCUMask: `mat` (256 - R CUs) runs vpt_conv3x3 launches, `mem` (R CUs) runs an HBM-bound
torch copy / the max-pool kernel.  Prints each alone (full chip and masked) and both together."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd  # noqa: F401  (package alias)
from vpt_amd import ops, packing

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.zeros(1, device=dev)


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask rc={rc}"
    return torch.cuda.ExternalStream(s.value, device=dev)


def reserve_mask(per_xcd):
    """Clear `per_xcd` CUs of every XCD under BOTH plausible bit layouts (blocked: word k = XCD k; interleaved: bit i -> XCD i % 8)."""
    mem = 0
    for k in range(8):
        for j in range(per_xcd):
            mem |= 1 << (32 * k + 8 * (j % 4) + k + (0 if j < 4 else 0))
    return mem


ALL = (1 << 256) - 1
f, h, cin, cout = 2048, 32, 256, 256
g = torch.Generator().manual_seed(0)
W = torch.randn(cout, cin, 3, 3, generator=g) * 0.02
wpk, sa, sg = packing.pack_conv3x3(W.to(dev), torch.ones(cin, device=dev), torch.zeros(cin, device=dev))
x = (torch.randn(f, cin // 32, h, h, 32, device=dev)).to(torch.bfloat16)
st_in = torch.stack([x.float().reshape(f, -1).sum(1).double(), (x.float() ** 2).reshape(f, -1).sum(1).double()], 1).contiguous()
out = torch.empty(f, cout // 32, h, h, 32, dtype=torch.bfloat16, device=dev)
src = torch.empty(1 << 29, dtype=torch.uint8, device=dev)   # 512 MB
dst = torch.empty_like(src)
NCONV, NCOPY = 20, 8


def run_conv(stream):
    with torch.cuda.stream(stream):
        for _ in range(NCONV):
            ops.conv3x3(x, wpk, sa, sg, st_in, cout, out=out)


def run_copy(stream):
    with torch.cuda.stream(stream):
        for _ in range(NCOPY):
            dst.copy_(src, non_blocking=True)


def timed(fns):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    for fn, s in fns:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); fn(s); e1.record(s)
        evs.append((e0, e1))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    return wall, [a.elapsed_time(b) for a, b in evs]


flops = 2.0 * f * h * h * cout * 9 * cin * NCONV
gbytes = 2.0 * src.numel() * NCOPY / 1e9
for per_xcd in (2, 4, 6, 8):
    memb = reserve_mask(min(per_xcd, 4)) if per_xcd <= 4 else 0
    if per_xcd > 4:   # second set of four: shift the pattern by one CU slot
        memb = reserve_mask(4)
        for k in range(8):
            for j in range(per_xcd - 4):
                memb |= 1 << (32 * k + 8 * j + (k + 1) % 8)
    R = bin(memb).count("1")
    s_full_a, s_full_b = masked_stream(ALL), masked_stream(ALL)
    s_mat, s_mem = masked_stream(ALL & ~memb), masked_stream(memb)
    for s in (s_full_a, s_mat):
        run_conv(s)
    torch.cuda.synchronize()
    w, (tc,) = timed([(run_conv, s_full_a)])
    print(f"R={R:3d}: conv alone, full chip      : {tc:7.2f} ms  {flops / tc / 1e9:7.1f} TF/s")
    w, (tc,) = timed([(run_conv, s_mat)])
    print(f"R={R:3d}: conv alone, {256 - R} CUs        : {tc:7.2f} ms  {flops / tc / 1e9:7.1f} TF/s")
    w, (tm,) = timed([(run_copy, s_full_b)])
    print(f"R={R:3d}: copy alone, full chip      : {tm:7.2f} ms  {gbytes / tm:7.2f} TB/s")
    w, (tm,) = timed([(run_copy, s_mem)])
    print(f"R={R:3d}: copy alone, {R} CUs          : {tm:7.2f} ms  {gbytes / tm:7.2f} TB/s")
    w, (tc, tm) = timed([(run_conv, s_full_a), (run_copy, s_full_b)])
    print(f"R={R:3d}: both, unmasked streams     : wall {w:7.2f} ms  conv {tc:7.2f}  copy {tm:7.2f}")
    w, (tc, tm) = timed([(run_conv, s_mat), (run_copy, s_mem)])
    print(f"R={R:3d}: both, partitioned          : wall {w:7.2f} ms  conv {tc:7.2f}  copy {tm:7.2f}")
