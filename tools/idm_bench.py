"""Throughput of the 4x inverse-dynamics model forward (BASELINE.json config 3: seq = 128) on one GPU.
python tools/idm_bench.py [--batch 1] [--seq 128] [--steps 3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.set_grad_enabled(False)
import __graft_entry__ as ge
ge.build()
from vpt_amd import ops
from vpt_amd.lib.policy import InverseActionPolicy
from vpt_amd.lib.types import idm_action_space
from vpt_amd import configs

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1); ap.add_argument("--seq", type=int, default=128); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--model", default="4x")
a = ap.parse_args()
kw = configs.idm_kwargs_for(a.model)
pol = InverseActionPolicy(idm_action_space(), pi_head_kwargs=dict(temperature=2.0), idm_net_kwargs=kw, precision=__import__("os").environ.get("VPT_PRECISION", "bf16"))
configs.randomize_(pol, 0)
pol = pol.to("cuda")
g = torch.Generator().manual_seed(1)
img = torch.randint(0, 256, (a.batch, a.seq, 128, 128, 3), generator=g, dtype=torch.uint8).to("cuda")
st = pol.initial_state(a.batch)
for _ in range(2):
    pol({"img": img}, first=None, state_in=st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    pol({"img": img}, first=None, state_in=st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
n = a.batch * a.seq
print(f"IDM {a.model} forward B={a.batch} T={a.seq}: {dt*1e3:.2f} ms/window  {n/dt:.0f} frames/s  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GB")
ops.TIMER.enabled = True; ops.TIMER.reset()
pol({"img": img}, first=None, state_in=st)
torch.cuda.synchronize()
summ = ops.TIMER.summary()
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] and v["ms"] else 0
    print(f"  {k:36s} {v['ms']:9.3f} ms {v['calls']:5d} calls  {tf:8.1f} TF/s")
