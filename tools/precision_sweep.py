"""Operand-format sweep on the CPU emulator (oracle/vpt_oracle_bf16.py): what each rounding point of the HIP
pipeline costs against the fp32 oracle, on the log-prob metric AND on the centred logits (the constant -log N of a
log-softmax carries no information but dominates the norms of the former), plus latent / value error and
deterministic-action agreement.  TEST INFRASTRUCTURE; writes profiles/r02_precision_sweep.md.

  python tools/precision_sweep.py [model=1x] [B=2] [T=6] [seed=0]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import vpt_oracle as O
from oracle import vpt_oracle_bf16 as E

model = sys.argv[1] if len(sys.argv) > 1 else "1x"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T = int(sys.argv[3]) if len(sys.argv) > 3 else 6
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
torch.set_num_threads(8)
pk = O.policy_kwargs_for(model)
cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
sd = O.synthetic_state_dict(cfg, seed=seed)
g = torch.Generator().manual_seed(1)
img = torch.randint(0, 256, (B, T, 128, 128, 3), generator=g, dtype=torch.uint8)
first = torch.zeros(B, T, dtype=torch.bool)
ref = O.policy_forward(sd, cfg, img, first, O.initial_state(cfg, B))


def l2(a, r):
    return float((a - r).norm() / r.norm())


def mx(a, r):
    return float((a - r).abs().max() / r.abs().max())


def centred(x):
    return x - x.mean(-1, keepdim=True)


def metrics(out):
    m = {}
    for h in ("buttons", "camera"):
        a, r = out[h], ref[h]
        m[h + "_lp_l2"], m[h + "_lp_max"] = l2(a, r), mx(a, r)
        m[h + "_c_l2"], m[h + "_c_max"] = l2(centred(a), centred(r)), mx(centred(a), centred(r))
        top2 = r.topk(2, -1).values
        margin = (top2[..., 0] - top2[..., 1])
        agree = (a.argmax(-1) == r.argmax(-1))
        err = (a - r).abs().max()
        safe = margin > 4 * err
        m[h + "_agree"] = float(agree.float().mean())
        m[h + "_safe_frac"] = float(safe.float().mean())
        m[h + "_safe_agree"] = float(agree[safe].float().mean()) if safe.any() else float("nan")
    m["latent_l2"] = l2(out["latent"], ref["latent"])
    m["vpred_rel"] = float((out["vpred"] - ref["vpred"]).abs().max() / ref["vpred"].abs().max())
    k_ref = ref["state_out"][-1][1][0][:, -T:]
    m["K3_l2"] = l2(out["state_out"][-1][1][0][:, -T:], k_ref)
    return m


CASES = [
    ("bf16 everywhere (round-1 kernels)", ("bf16", "bf16", "bf16")),
    ("only weights bf16", ("bf16", "fp32", "fp32")),
    ("only CNN activations bf16", ("fp32", "bf16", "fp32")),
    ("only trunk bf16", ("fp32", "fp32", "bf16")),
    ("fp16 everywhere", ("fp16", "fp16", "fp16")),
    ("only weights fp16", ("fp16", "fp32", "fp32")),
    ("only CNN activations fp16", ("fp32", "fp16", "fp32")),
    ("only trunk fp16", ("fp32", "fp32", "fp16")),
    ("bf16 hi+lo weights, bf16 acts (2 passes)", ("bf16x2", "bf16", "bf16")),
    ("bf16 hi+lo weights + CNN acts (3 passes), bf16 trunk", ("bf16x2", "bf16x2", "bf16")),
    ("bf16 hi+lo everywhere (3 passes)", ("bf16x2", "bf16x2", "bf16x2")),
    ("fp16 hi+lo weights, fp16 acts (2 passes)", ("fp16x2", "fp16", "fp16")),
    ("fp16 hi+lo everywhere (3 passes)", ("fp16x2", "fp16x2", "fp16x2")),
    ("fp32 everywhere (emulator == oracle check)", ("fp32", "fp32", "fp32")),
    # feasibility probe for DESIGN.md section 10: the fourteen 3x3 convs as Winograd F(2x2, 3x3), transformed operands rounded
    ("bf16 everywhere, 3x3 convs as Winograd F(2x2,3x3)", ("bf16", "bf16", "bf16", True)),
    ("fp16 everywhere, 3x3 convs as Winograd F(2x2,3x3)", ("fp16", "fp16", "fp16", True)),
    ("fp32 Winograd (transform algebra check)", ("fp32", "fp32", "fp32", True)),
]
rows = []
for name, fmt in CASES:
    w, a, t = fmt[:3]
    out = E.policy_forward(sd, cfg, img, first, O.initial_state(cfg, B), rnd=E.Rounding(w, a, t, winograd=len(fmt) > 3 and fmt[3]))
    m = metrics(out)
    rows.append((name, m))
    print(f"{name:55s} lp_l2 b/c {m['buttons_lp_l2']:.2e}/{m['camera_lp_l2']:.2e}  centred_l2 b/c {m['buttons_c_l2']:.2e}/{m['camera_c_l2']:.2e} "
          f"centred_max {m['buttons_c_max']:.2e}/{m['camera_c_max']:.2e} latent {m['latent_l2']:.2e} v {m['vpred_rel']:.2e} "
          f"agree {m['buttons_agree']:.3f}/{m['camera_agree']:.3f} safe {m['buttons_safe_frac']:.2f}/{m['camera_safe_frac']:.2f}", flush=True)

path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"r02_precision_sweep_{model}.md")
with open(path, "w") as f:
    f.write(f"# Operand-format sweep (CPU emulator vs fp32 oracle), model {model}, B={B} T={T}, synthetic weights seed {seed}\n\n")
    f.write("`lp` = log-probabilities as returned by the policy; `c` = centred logits (x - mean over the head). l2 = relative L2, max = max|d|/max|ref|.\n"
            "`agree` = deterministic action equal to the oracle's; `safe` = fraction of positions whose oracle top-2 margin exceeds 4x the max error.\n\n")
    f.write("| rounding (weights / CNN acts / trunk) | lp l2 buttons | lp l2 camera | lp max b | lp max c | c l2 b | c l2 c | c max b | c max c | latent l2 | vpred rel | K mem l2 | agree b | agree c | safe b | safe c |\n")
    f.write("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for name, m in rows:
        f.write(f"| {name} | {m['buttons_lp_l2']:.2e} | {m['camera_lp_l2']:.2e} | {m['buttons_lp_max']:.2e} | {m['camera_lp_max']:.2e} | {m['buttons_c_l2']:.2e} | {m['camera_c_l2']:.2e} | "
                f"{m['buttons_c_max']:.2e} | {m['camera_c_max']:.2e} | {m['latent_l2']:.2e} | {m['vpred_rel']:.2e} | {m['K3_l2']:.2e} | {m['buttons_agree']:.3f} | {m['camera_agree']:.3f} | "
                f"{m['buttons_safe_frac']:.2f} | {m['camera_safe_frac']:.2f} |\n")
print("wrote", path)
