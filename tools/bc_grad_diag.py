"""Locate where the GPU backward diverges from the oracle's autograd: compares gradients w.r.t. intermediate
activations (run on the GPU box).  python tools/bc_grad_diag.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from vpt_amd.training import BCTrainer
from vpt_amd.lib.policy import MinecraftAgentPolicy
from vpt_amd.lib.types import minecraft_action_space
from oracle import vpt_oracle as O
from oracle import vpt_oracle_bf16 as OB

torch.set_num_threads(32)
DEV = "cuda"
pk = O.policy_kwargs_for("1x"); cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
sd = O.synthetic_state_dict(cfg, 0)
pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision=__import__("os").environ.get("VPT_PRECISION", "bf16")); pol.load_state_dict(sd, strict=False); pol = pol.to(DEV)
tr = BCTrainer(pol)
b, t = 2, 6
g = torch.Generator().manual_seed(5)
img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
first = torch.zeros(b, t, dtype=torch.bool)
ab = torch.randint(0, 8641, (b, t), generator=g); ac = torch.randint(0, 121, (b, t), generator=g)
# oracle with retained intermediate grads
leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
taps = {}
EM = os.environ.get('EMUL', '1') == '1'
out = OB.policy_forward(leaves, cfg, img, first, O.initial_state(cfg, b), taps=taps, grad=True) if EM else O.policy_forward(leaves, cfg, img, first, O.initial_state(cfg, b), taps=taps, grad=True)
lp = out["buttons"][:, :, 0].gather(-1, ab.unsqueeze(-1)).squeeze(-1) + out["camera"][:, :, 0].gather(-1, ac.unsqueeze(-1)).squeeze(-1)
loss = -lp.mean()
names = ["latent", "y"] + [f"block{l}" for l in range(4)] + ["img_process"]
tens = [out["latent"], taps["y"]] + [taps[f"block{l}"] for l in range(4)] + [taps["img_process"]]
gr = torch.autograd.grad(loss, tens, retain_graph=True)
ref = dict(zip(names, gr))
dbg = {}
loss_g, grads, _ = tr.loss_and_grads(img.to(DEV), first.to(DEV), pol.initial_state(b), ab.to(DEV), ac.to(DEV), debug=dbg)
torch.cuda.synchronize()
l2 = lambda a, r: float((a - r).norm() / r.norm())
m = b * t
print("loss", float(loss_g), float(loss))
print("latent fwd err", l2(dbg["latent"].cpu(), out["latent"].detach().reshape(m, -1)))
print("x_trunk fwd err", l2(dbg["x_trunk"].cpu(), taps["block3"].detach().reshape(m, -1)))
print("d latent", l2(dbg["dlatent"].cpu(), ref["latent"].reshape(m, -1)))
print("y fwd err", l2(dbg["y"].cpu(), taps["y"].detach().reshape(m, -1)), "gate mismatch frac", float(((dbg["y"].cpu() > 0) != (taps["y"].detach().reshape(m, -1) > 0)).float().mean()))
print("d y", l2(dbg["dy"].cpu(), ref["y"].reshape(m, -1)))
print("d x_trunk (= d block3 out)", l2(dbg["dx_trunk"].cpu(), ref["block3"].reshape(m, -1)))
for l in (3, 2, 1):
    print(f"d block{l-1} out", l2(dbg[f"dx_block{l}"].cpu(), ref[f"block{l-1}"].reshape(m, -1)))
print("d img_process", l2(dbg["dx_block0"].cpu(), ref["img_process"].reshape(m, -1)))
