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


def run(n, stochastic=False):
    st = pol.initial_state(1)
    acts = []
    for i in range(n):
        ac, st, _ = pol.act({"img": frames[i % frames.shape[0]]}, first, st, stochastic=stochastic)
        acts.append(ac)
    torch.cuda.synchronize()
    return acts


for mode, stoch in (("eager", False), ("eager", True), ("graph", False), ("graph", True)):
    if mode == "graph":
        pol.auto_step_graph(True)       # what act() does by itself: the step is captured from the third same-shape call on
    else:
        pol.disable_step_graph()
    run(10, stoch)
    t0 = time.perf_counter()
    acts = run(a.steps, stoch)
    dt = (time.perf_counter() - t0) / a.steps
    print(f"{mode:6s} {'stochastic   ' if stoch else 'deterministic'}: {dt*1e3:.3f} ms / step  ({1/dt:.0f} steps/s)   buttons[0..5] = {[int(x['buttons']) for x in acts[:6]]}")

# ---- the reference's own wrapper, unmodified: MineRLAgent.get_action on a 640x360 observation (agent.py:190-206: resize, H2D copy,
# policy.act(..., stochastic=True), action mapping on the host) over the HIP policy -- what run_agent.py's loop pays per step
try:
    from tests import ref_env
    R = ref_env.reference()
    if R is not None:
        import numpy as np
        ag_mod = R.agent
        ref_class = ag_mod.MinecraftAgentPolicy
        from vpt_amd.lib import policy as hip_policy
        ag_mod.MinecraftAgentPolicy = hip_policy.MinecraftAgentPolicy           # the one-line swap of INTEGRATION.md
        try:
            agent = ag_mod.MineRLAgent(ref_env.FakeEnv(ag_mod), device="cuda", policy_kwargs=pk, pi_head_kwargs=dict(temperature=2.0))
        finally:
            ag_mod.MinecraftAgentPolicy = ref_class
        agent.policy.load_state_dict(pol.state_dict(), strict=False)
        agent.policy.set_precision(pol.precision)
        rng = np.random.default_rng(0)
        obs = [{"pov": rng.integers(0, 256, (360, 640, 3), dtype=np.uint8)} for _ in range(16)]
        obs128 = [{"pov": rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)} for _ in range(16)]   # already at AGENT_RESOLUTION: cv2.resize is the identity
        for variant in ("auto graph (default)", "eager (VPT_STEP_GRAPH=0)"):
            if variant.startswith("eager"):
                agent.policy.disable_step_graph()
            for what, ob in (("640x360 obs (the image's cv2 STUB resizes in numpy: test infrastructure, ~2 ms)", obs), ("128x128 obs (no resize: policy + the wrapper's own host code)", obs128)):
                agent.reset()
                for i in range(12):
                    agent.get_action(ob[i % 16])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(a.steps):
                    act = agent.get_action(ob[i % 16])
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / a.steps
                print(f"wrapper: unmodified MineRLAgent.get_action (stochastic=True), {variant}, {what}: {dt*1e3:.3f} ms / step  ({1/dt:.0f} steps/s)")
        # where the wrapper's time goes besides policy.act: its own host code
        agent.policy.auto_step_graph(True)
        t0 = time.perf_counter()
        for i in range(a.steps):
            agent._env_obs_to_agent(obs[i % 16])
        torch.cuda.synchronize()
        print(f"wrapper: _env_obs_to_agent alone (resize stub + from_numpy + H2D): {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms / step")
except Exception as e:
    print("wrapper measurement unavailable:", type(e).__name__, e)

pol.auto_step_graph(True)
run(10, True)
sg = getattr(pol, "_step_graph", None)
if sg and "graphs" in sg and "stochastic" in sg["graphs"]:      # the GPU side alone: back-to-back replays, no host glue
    gph = sg["graphs"]["stochastic"][0]
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
    pr = cProfile.Profile(); pr.enable(); run(200, True); pr.disable()
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
