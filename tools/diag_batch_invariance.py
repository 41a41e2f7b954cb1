"""Where does a sequence's result start to depend on the batch around it?  python tools/diag_batch_invariance.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from vpt_amd import ops, configs
from vpt_amd.lib.policy import MinecraftAgentPolicy
from vpt_amd.lib.types import minecraft_action_space

torch.set_grad_enabled(False)
pk = configs.policy_kwargs_for("1x")
pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision=os.environ.get("VPT_PRECISION", "bf16"))
if os.environ.get("DIAG_SYNTH") == "1":
    from oracle import vpt_oracle as O
    pol.load_state_dict(O.synthetic_state_dict(O.config_from_policy_kwargs(pk, dict(temperature=2.0)), seed=0), strict=False)
else:
    configs.randomize_(pol, 0)
pol = pol.to("cuda")
pol._ensure_packed()
eng = pol._engine
t = 5
a = torch.randint(0, 256, (2, t, 128, 128, 3), generator=torch.Generator().manual_seed(301), dtype=torch.uint8)
other = torch.randint(0, 256, (4, t, 128, 128, 3), generator=torch.Generator().manual_seed(302), dtype=torch.uint8)
big = torch.stack([other[0], a[0], other[1], other[2], a[1], other[3]])
fa, fb = a.reshape(-1, 128, 128, 3).cuda(), big.reshape(-1, 128, 128, 3).cuda()
rows = torch.cat([torch.arange(t) + t * 1, torch.arange(t) + t * 4]).cuda()


first2, first6 = torch.zeros(2, t, dtype=torch.bool, device="cuda"), torch.zeros(6, t, dtype=torch.bool, device="cuda")
idx = torch.tensor([1, 4], device="cuda")


def eq(x, y, name):
    same = torch.equal(x, y)
    d = (x.float() - y.float()).abs().max().item()
    print(f"{name:40s} {'IDENTICAL' if same else 'DIFFERENT'}  max |d| {d:.3e}")
    return same

# COLD: the very first launches of the process, through the public API (what a test sees)
(pa0, _, _), _ = pol({"img": a.cuda()}, first2, pol.initial_state(2))
(pb0, _, _), _ = pol({"img": big.cuda()}, first6, pol.initial_state(6))
(pa1, _, _), _ = pol({"img": a.cuda()}, first2, pol.initial_state(2))
eq(pa0["buttons"], pb0["buttons"][idx], "COLD policy.forward B=2 (first call) vs B=6")
eq(pa1["buttons"], pb0["buttons"][idx], "policy.forward B=2 (third call) vs B=6")
eq(pa0["buttons"], pa1["buttons"], "policy.forward B=2 first vs third call")
# run-to-run
d1 = eng._img_process(fa); d2 = eng._img_process(fa)
eq(d1, d2, "img_process, same batch twice")
db = eng._img_process(fb)
eq(d1, db[rows], "img_process, B=2 vs inside B=6")
# stage by stage in the CNN (single stream)
w = eng.w
c0 = eng.cfg["chans"][0]
for name, fr in (("a", fa), ("b", fb)):
    st = torch.zeros(fr.shape[0], 2, dtype=torch.float64, device="cuda")
    p0 = ops.conv_first(fr, w["net.img_process.cnn.stacks.0.firstconv"], c0, stats_out=st)
    globals()["p0_" + name], globals()["st_" + name] = p0, st
eq(p0_a, p0_b[rows], "conv_first output")
eq(st_a, st_b[rows], "conv_first frame statistics (fp64)")
xa = eng._cnn_chunk(fa); xb = eng._cnn_chunk(fb)
eq(xa, xb[rows], "_cnn_chunk (normalised CNN output)")
da, _ = ops.linear(xa.view(xa.shape[0], -1), w["net.img_process.cnn.dense.w"], 256, splitk=16)
dbb, _ = ops.linear(xb.view(xb.shape[0], -1), w["net.img_process.cnn.dense.w"], 256, splitk=16)
eq(da, dbb[rows], "dense GEMM (split-K 16) on own outputs")
dc, _ = ops.linear(xb[rows].contiguous().view(10, -1), w["net.img_process.cnn.dense.w"], 256, splitk=16)
eq(da, dc, "dense GEMM, same rows, M = 10 both")
# full forward
oa = eng.forward(a.cuda(), first2, pol.initial_state(2))
ob = eng.forward(big.cuda(), first6, pol.initial_state(6))
eq(oa["latent"], ob["latent"][idx], "latent")
eq(oa["buttons"], ob["buttons"][idx], "buttons log-probs")

# through the public API, as the test does
(pa, va, _), sa = pol({"img": a.cuda()}, first2, pol.initial_state(2))
(pb, vb, _), sb = pol({"img": big.cuda()}, first6, pol.initial_state(6))
eq(pa["buttons"], pb["buttons"][idx], "policy.forward buttons")
eq(sa[0][1][0], sb[0][1][0][idx], "policy.forward K memory of block 0")
for l in range(4):
    eq(oa["state_out"][l][1][0], ob["state_out"][l][1][0][idx], f"engine K memory block {l}")
