"""Why did bench.py's first ingest leg show no copy / compute overlap?  Variants of ONE pinned-host -> device copy stream beside the forward:
   default-priority copy stream, high-priority copy stream (its own hardware queue), fewer engine side streams (<= 4 HIP streams in total:
   the runtime maps streams onto GPU_MAX_HW_QUEUES = 4 hardware queues round-robin, a copy that shares a queue with a compute stream
   serialises behind its kernels).  python tools/ingest_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from vpt_amd.lib.policy import MinecraftAgentPolicy
from vpt_amd.lib.types import minecraft_action_space
from vpt_amd import configs

dev = torch.device("cuda", 0)
pol = MinecraftAgentPolicy(minecraft_action_space(), configs.policy_kwargs_for("2x"), dict(temperature=2.0), precision="bf16")
configs.randomize_(pol, 0)
pol = pol.to(dev)
B, T, steps = 64, 128, 4
img = torch.randint(0, 256, (B, T, 128, 128, 3), dtype=torch.uint8, device=dev)
first = torch.zeros(B, T, dtype=torch.bool, device=dev)
host = torch.empty(img.shape, dtype=torch.uint8).pin_memory()
host.copy_(img)
bufs = [torch.empty_like(img), torch.empty_like(img)]


def fwd(frames, st):
    with torch.no_grad():
        (_, _, _), st = pol({"img": frames}, first, st)
    return st


def timeit(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def run(copy_s, label):
    main = torch.cuda.current_stream()

    def copies_only():
        for k in range(steps):
            with torch.cuda.stream(copy_s):
                bufs[k % 2].copy_(host, non_blocking=True)

    def compute_only():
        st = pol.initial_state(B)
        for k in range(steps):
            st = fwd(bufs[k % 2], st)

    def pipelined():
        st = pol.initial_state(B)
        done = {}
        with torch.cuda.stream(copy_s):
            bufs[0].copy_(host, non_blocking=True)
            ready = torch.cuda.Event(); ready.record(copy_s)
        for k in range(steps):
            main.wait_event(ready)
            if k + 1 < steps:
                with torch.cuda.stream(copy_s):
                    if k - 1 in done:
                        copy_s.wait_event(done[k - 1])
                    bufs[(k + 1) % 2].copy_(host, non_blocking=True)
                    ready = torch.cuda.Event(); ready.record(copy_s)
            st = fwd(bufs[k % 2], st)
            done[k] = torch.cuda.Event(); done[k].record(main)

    copies_only(); compute_only()
    tc, tf, tp = timeit(copies_only), timeit(compute_only), timeit(pipelined)
    print(f"{label:58s} copy {tc:7.2f} ms  forward {tf:7.2f} ms  pipelined {tp:7.2f} ms  hidden {1 - max(0.0, tp - tf) / tc:6.3f}", flush=True)


print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), " engine cnn_streams =", pol._engine.cnn_streams)
run(torch.cuda.Stream(), "default-priority copy stream")
run(torch.cuda.Stream(priority=-1), "high-priority copy stream")
for n in (2, 1):
    pol._engine.cnn_streams = n
    run(torch.cuda.Stream(), f"default-priority copy stream, engine cnn_streams = {n}")
    run(torch.cuda.Stream(priority=-1), f"high-priority copy stream, engine cnn_streams = {n}")
