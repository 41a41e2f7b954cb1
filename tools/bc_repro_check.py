"""Bit-reproducibility of the BC gradients at the BENCHMARK size (2x model, 64 x 128 frames, CNN chunks of 1024 on three streams): the same batch
`reps` times in one process, every gradient tensor and the loss compared with the first run bit for bit.
    python tools/bc_repro_check.py [reps=3] [model=2x] [batch=64] [seq=128]
(tests/test_gpu_training.py::test_bc_gradients_bitwise_reproducible is the small-shape version that runs in the suite.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from vpt_amd import configs  # noqa: E402
from vpt_amd.lib.policy import MinecraftAgentPolicy  # noqa: E402
from vpt_amd.lib.types import minecraft_action_space  # noqa: E402
from vpt_amd.training import BCTrainer  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
model = sys.argv[2] if len(sys.argv) > 2 else "2x"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
T = int(sys.argv[4]) if len(sys.argv) > 4 else 128
for precision in ("bf16", "fp16"):
    pol = MinecraftAgentPolicy(minecraft_action_space(), configs.policy_kwargs_for(model), dict(temperature=2.0), precision=precision)
    configs.randomize_(pol, 0)
    pol = pol.to("cuda")
    tr = BCTrainer(pol, train_cnn=True)
    g = torch.Generator().manual_seed(1)
    img = torch.randint(0, 256, (B, T, 128, 128, 3), generator=g, dtype=torch.uint8).cuda()
    first = torch.zeros(B, T, dtype=torch.bool, device="cuda")
    ab, ac = torch.randint(0, 8641, (B, T), generator=g).cuda(), torch.randint(0, 121, (B, T), generator=g).cuda()
    outs = []
    for r in range(reps):      # the inference forward (three chunk streams, folded path) first
        with torch.no_grad():
            (pd, v, _), st = pol({"img": img}, first, pol.initial_state(B))
        outs.append((pd["buttons"].clone(), pd["camera"].clone(), v.clone(), st[-1][1][0].clone()))
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for o in outs[1:] for a, b in zip(o, outs[0]))
    print(f"[{precision}] {model} {B}x{T} inference forward, {reps} runs: log-probs, value and KV memory {'bit-identical' if same else 'DIFFER'}", flush=True)
    del outs
    ref, t0 = None, time.time()
    for r in range(reps):
        loss, grads, _ = tr.loss_and_grads(img, first, pol.initial_state(B), ab, ac)
        torch.cuda.synchronize()
        if ref is None:
            ref = (loss.clone(), {k: v.clone() for k, v in grads.items()})
            continue
        bad = [k for k in ref[1] if not torch.equal(grads[k], ref[1][k])]
        print(f"[{precision}] {model} {B}x{T} run {r} vs run 0: loss {'equal' if torch.equal(loss, ref[0]) else 'DIFFERS'}, {len(ref[1]) - len(bad)} of {len(ref[1])} gradient tensors bit-identical"
              + (f"; differing: {bad[:5]}" if bad else ""), flush=True)
    print(f"[{precision}] {reps} runs in {time.time() - t0:.1f} s, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GB", flush=True)
    del pol, tr, ref, grads
    torch.cuda.empty_cache()
