"""Time the BC step (forward + backward + Adam) on one GPU and print the per-kernel breakdown.
python tools/bc_bench.py [--model 2x] [--batch 64] [--seq 128] [--steps 2] [--no-cnn]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from vpt_amd import ops
from vpt_amd.training import BCTrainer
from vpt_amd.lib.policy import MinecraftAgentPolicy
from vpt_amd.lib.types import minecraft_action_space
from vpt_amd import configs

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="2x"); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--seq", type=int, default=128)
ap.add_argument("--steps", type=int, default=2); ap.add_argument("--no-cnn", action="store_true")
ap.add_argument("--streams1", action="store_true", help="the instrumented step on ONE stream (per-kernel durations without cross-stream overlap, as bench.py reports them)")
a = ap.parse_args()
dev = "cuda"
pk = configs.policy_kwargs_for(a.model)
pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision=__import__("os").environ.get("VPT_PRECISION", "bf16")); configs.randomize_(pol, 0); pol = pol.to(dev)
tr = BCTrainer(pol, train_cnn=not a.no_cnn)
if os.environ.get("VPT_DP_FORCE_EXCHANGE") == "1":      # the data-parallel step over RCCL in a ONE-rank group (what the exchange plumbing costs on the real transport)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29571"), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    print("data-parallel path forced over", dist.get_backend(), "with", dist.get_world_size(), "rank")
g = torch.Generator().manual_seed(1)
B, T = a.batch, a.seq
img = torch.randint(0, 256, (B, T, 128, 128, 3), generator=g, dtype=torch.uint8).to(dev)
first = torch.zeros(B, T, dtype=torch.bool, device=dev)
ab = torch.randint(0, 8641, (B, T), generator=g).to(dev); ac = torch.randint(0, 121, (B, T), generator=g).to(dev)
st = pol.initial_state(B)
loss, st = tr.step(img, first, st, ab, ac)
torch.cuda.synchronize()
print("warm-up loss", loss, "peak mem GB", torch.cuda.max_memory_allocated() / 2**30)
t0 = time.perf_counter()
for _ in range(a.steps):
    loss, st = tr.step(img, first, st, ab, ac)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"BC step {dt*1e3:.1f} ms  ({B*T/dt:.0f} frames/s)  loss {loss:.4f}")
if a.streams1:
    tr.cnn_streams = 1
ops.TIMER.enabled = True; ops.TIMER.reset()
t0 = time.perf_counter()
tr.step(img, first, st, ab, ac)
torch.cuda.synchronize()
print(f"instrumented step wall {1e3*(time.perf_counter()-t0):.1f} ms")
summ = ops.TIMER.summary()
tot = sum(v["ms"] for v in summ.values())
print(f"kernel time total {tot:.1f} ms")
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] and v["ms"] else 0
    print(f"  {k:36s} {v['ms']:9.2f} ms {v['calls']:5d} calls  {tf:8.1f} TF/s")
if a.streams1:       # per layer shape: which of the three convolution passes loses where
    for pre in ("vpt_conv3x3", "vpt_conv_backward_prepare"):
        for (k, work), v in sorted(ops.TIMER.by_shape(pre).items(), key=lambda kv: (kv[0][0], -kv[1]["ms"])):
            rate = f"{v['flops'] / (v['ms'] * 1e-3) / 1e12:8.1f} TF/s" if v["flops"] else f"{v['bytes'] / (v['ms'] * 1e-3) / 1e12:8.2f} TB/s"
            print(f"    {k:30s} work/call {work:10.3e}  {v['ms']:8.2f} ms {v['calls']:4d} calls  {rate}")
