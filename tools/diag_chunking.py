"""Diagnostic: BC gradients of the same 30 frames -- twice with one CNN chunk (run-to-run noise), then with chunks of 10 on 1 and on 3 streams.
python tools/diag_chunking.py [bf16|fp16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from vpt_amd.training import BCTrainer
from vpt_amd.lib.policy import MinecraftAgentPolicy
from vpt_amd.lib.types import minecraft_action_space
from vpt_amd import configs
mode = sys.argv[1] if len(sys.argv) > 1 else "fp16"
pol = MinecraftAgentPolicy(minecraft_action_space(), configs.policy_kwargs_for("1x"), dict(temperature=2.0), precision=mode)
configs.randomize_(pol, 0); pol = pol.to("cuda")
g = torch.Generator().manual_seed(51)
b, t = 3, 10
img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8).cuda()
first = torch.zeros(b, t, dtype=torch.bool, device="cuda")
ab, ac = torch.randint(0, 8641, (b, t), generator=g).cuda(), torch.randint(0, 121, (b, t), generator=g).cuda()
tr = BCTrainer(pol, train_cnn=True)
def run(chunk, streams):
    pol._engine.cnn_chunk = chunk; tr.cnn_streams = streams
    l, gr, _ = tr.loss_and_grads(img, first, pol.initial_state(b), ab, ac)
    torch.cuda.synchronize()
    return float(l), {k: v.clone() for k, v in gr.items()}
def cmp(a, bb, what):
    errs = {k: float((bb[k].float() - v.float()).norm() / v.float().norm().clamp(min=1e-30)) for k, v in a.items() if float(v.norm()) > 0}
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(f"[{mode}] {what}: worst {top[0][1]:.2e}  " + "  ".join(f"{k.replace('net.img_process.cnn.', '')}={e:.1e}" for k, e in top))
l1, g1 = run(1024, 1); l1b, g1b = run(1024, 1)
cmp(g1, g1b, "one chunk, run twice        ")
l2, g2 = run(10, 1); cmp(g1, g2, "chunks of 10, 1 stream      ")
l3, g3 = run(10, 3); cmp(g1, g3, "chunks of 10, 3 streams     ")
l4, g4 = run(10, 3); cmp(g3, g4, "chunks of 10, 3 streams, x2 ")
print(l1, l1b, l2, l3)
