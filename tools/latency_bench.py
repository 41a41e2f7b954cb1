"""T = 1 acting latency (agent.py:get_action -> policy.act) on one GPU: eager launches vs the captured hipGraph.
python tools/latency_bench.py [--model 2x] [--steps 200]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from vpt_amd.lib.policy import MinecraftAgentPolicy
from vpt_amd.lib.types import minecraft_action_space
from vpt_amd import configs

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="2x"); ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
pk = configs.policy_kwargs_for(a.model)
pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision=__import__("os").environ.get("VPT_PRECISION", "bf16")); configs.randomize_(pol, 0); pol = pol.to("cuda")
g = torch.Generator().manual_seed(1)
frames = torch.randint(0, 256, (a.steps, 1, 128, 128, 3), generator=g, dtype=torch.uint8).to("cuda")
first = torch.zeros(1, dtype=torch.bool, device="cuda")


def run(n):
    st = pol.initial_state(1)
    acts = []
    for i in range(n):
        ac, st, _ = pol.act({"img": frames[i]}, first, st, stochastic=False)
        acts.append(ac)
    torch.cuda.synchronize()
    return acts


for mode in ("eager", "graph"):
    if mode == "graph":
        if not hasattr(pol, "enable_step_graph"):
            break
        pol.enable_step_graph()
    run(10)
    t0 = time.perf_counter()
    acts = run(a.steps)
    dt = (time.perf_counter() - t0) / a.steps
    print(f"{mode:6s}: {dt*1e3:.3f} ms / step  ({1/dt:.0f} steps/s)   buttons[0..5] = {[int(x['buttons']) for x in acts[:6]]}")

if getattr(pol, "_step_graph", None) and "graph" in pol._step_graph:      # the GPU side alone: back-to-back replays, no host glue
    gph = pol._step_graph["graph"]
    for _ in range(10):
        gph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        gph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"replay: {dt*1e3:.3f} ms / graph replay (no act() glue: input copies, output clones, python)")
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable(); run(200); pr.disable()
    sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats("cumulative").print_stats(18)
    print("\n".join(l[:150] for l in sio.getvalue().splitlines()[:40]))

from vpt_amd import ops
pol.disable_step_graph()
ops.TIMER.enabled = True; ops.TIMER.reset()
run(1)
summ = ops.TIMER.summary()
print(f"per-kernel GPU time of one eager T=1 step: total {sum(v['ms'] for v in summ.values()):.3f} ms")
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:36s} {v['ms']*1e3:9.1f} us {v['calls']:4d} calls")
