"""Where do the BC gradients of the pool-fused training path (round 5) part from the conv -> pool path?  Same inputs, same weights, both paths in one
process: saved activations of the forward first (pooled tensors, their statistics, block outputs), then every gradient tensor.
python tools/diag_fused_pool.py [fp16|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from oracle import vpt_oracle as O
from vpt_amd.lib.policy import MinecraftAgentPolicy
from vpt_amd.lib.types import minecraft_action_space
from vpt_amd.training import BCTrainer
from vpt_amd import ops, packing

mode = sys.argv[1] if len(sys.argv) > 1 else "fp16"
DEV = "cuda"
pk = O.policy_kwargs_for("1x")
cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
sd = O.synthetic_state_dict(cfg, seed=0)
pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision=mode)
pol.load_state_dict(sd, strict=False)
pol = pol.to(DEV)
b, t = 2, 6
g = torch.Generator().manual_seed(5)
img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)
first = torch.zeros(b, t, dtype=torch.bool, device=DEV)
ab = torch.randint(0, 8641, (b, t), generator=g).to(DEV)
ac = torch.randint(0, 121, (b, t), generator=g).to(DEV)
l2 = lambda a, r: float((a.double() - r.double()).norm() / r.double().norm().clamp(min=1e-30))
tr = BCTrainer(pol, train_cnn=True)
res = {}
for name, fp, fn in (("conv->pool", False, False), ("fused", True, False), ("fused+nfold", True, True)):
    tr.fused_pool, tr.fold_n_backward = fp, fn
    S = tr.forward_saving(img, first, pol.initial_state(b))
    sv = S["cnn_saved"][0]
    fw = {}
    for s, rec in enumerate(sv["stacks"]):
        fw[f"s{s}.pooled"] = rec["pooled"].float().clone()
        fw[f"s{s}.s_pool"] = rec["s_pool"].clone()
        for bi, blk in enumerate(rec["blocks"]):
            fw[f"s{s}.b{bi}.y"] = blk["y"].float().clone()
            fw[f"s{s}.b{bi}.x_out"] = blk["x_out"].float().clone()
        if "mask" in rec and "argmax_ref" in res:
            m = rec["mask"].to(torch.int32) & 0x1ff
            inv = (~m) & 0x1ff
            code = 8 - torch.floor(torch.log2(inv.float())).to(torch.int32)
            am = res["argmax_ref"][s].to(torch.int32)
            live = rec["pooled"].float() > 0
            print(f"  [{name}] stack {s}: decoded arg-max == vpt_maxpool_forward's on {float((code[live] == am[live]).float().mean()):.6f} of the live positions ({int(live.sum())})")
    if not fp:
        res["argmax_ref"] = {s: rec["argmax"].clone() for s, rec in enumerate(sv["stacks"]) if "argmax" in rec}
    del S
    loss, grads, _ = tr.loss_and_grads(img, first, pol.initial_state(b), ab, ac)
    torch.cuda.synchronize()
    res[name] = (fw, {k: v.float().clone() for k, v in grads.items()}, float(loss))
ref_fw, ref_g, ref_loss = res["conv->pool"]
for name in ("fused", "fused+nfold"):
    fw, gr, loss = res[name]
    print(f"== {name} vs conv->pool ({mode}): loss {loss:.6f} vs {ref_loss:.6f}")
    for k in ref_fw:
        print(f"   forward {k:14s} rel-L2 {l2(fw[k], ref_fw[k]):.3e}  identical {bool(torch.equal(fw[k], ref_fw[k]))}")
    rows = sorted(((l2(gr[k], ref_g[k]), k) for k in ref_g if float(ref_g[k].norm()) > 0), reverse=True)
    print("   gradients, largest differences:", [(k, f"{e:.3e}") for e, k in rows[:8]])
    cnn = [e for e, k in rows if "cnn" in k]
    print(f"   mean rel-L2 over CNN tensors {sum(cnn) / len(cnn):.3e}, over the others {sum(e for e, k in rows if 'cnn' not in k) / max(1, len(rows) - len(cnn)):.3e}")
    for e, k in rows:
        if "firstconv" in k or ".n." in k:
            print(f"      {k:70s} {e:.3e}")

# ---- is either path closer to the fp32 oracle?  Several input seeds, mean rel-L2 over all tensors to the CPU oracle's gradients (the figure
# test_bc_gradients_vs_oracle prints): two noise realisations of the same arithmetic should scatter around the same value
if os.environ.get("VPT_DIAG_ORACLE", "1") == "1":
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    for seed in (5, 6, 7, 8):
        g = torch.Generator().manual_seed(seed)
        img_c = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
        first_c = torch.zeros(b, t, dtype=torch.bool)
        ab_c, ac_c = torch.randint(0, 8641, (b, t), generator=g), torch.randint(0, 121, (b, t), generator=g)
        _, grads_ref = O.bc_loss_and_grads(sd, cfg, img_c, first_c, O.initial_state(cfg, b), ab_c, ac_c)[:2]
        out = []
        for name, fp in (("conv->pool", False), ("fused", True)):
            tr.fused_pool, tr.fold_n_backward = fp, fp
            _, grads, _ = tr.loss_and_grads(img_c.to(DEV), first_c.to(DEV), pol.initial_state(b), ab_c.to(DEV), ac_c.to(DEV))
            torch.cuda.synchronize()
            ds, cs = [], []
            for k in tr.trainable:
                r = grads_ref[k]
                if float(r.norm()) == 0.0:
                    continue
                m = grads[k].cpu().reshape(r.shape).float()
                ds.append(l2(m, r)); cs.append(float((m * r).sum() / (m.norm() * r.norm())))
            out.append(f"{name}: mean rel-L2 {sum(ds) / len(ds):.4f}, mean cosine {sum(cs) / len(cs):.4f}, worst {min(cs):.3f}")
        print(f"seed {seed} [{mode}] vs the fp32 oracle -- " + " | ".join(out), flush=True)
