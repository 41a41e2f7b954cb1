"""End-to-end parity of the HIP policy (through the reference's MinecraftAgentPolicy API) against the golden
vectors of the live reference and against the oracle.  Needs an MI355X.

Tolerance: the conv / linear kernels take bf16 operands with fp32 accumulation; statistics, softmax, the
residual stream of the transformer and the KV memory stay fp32.  A CPU emulation of exactly these rounding
points predicts, for the synthetic 1x weights: log-prob relative-L2 error ~1e-3
(max|d|/max|ref| ~4e-3) and ~3.5e-2 relative-L2 on the K/V memory (error grows ~0.15 %/layer through the 14
normalised CNN layers).  Bounds used here: log-probs rel-L2 < 3e-3 and max|d|/max|ref| < 1e-2; K/V memory
rel-L2 < 6e-2; masks exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd.lib.policy import MinecraftAgentPolicy  # noqa: E402
from vpt_amd.lib.types import minecraft_action_space  # noqa: E402
from oracle import vpt_oracle as O  # noqa: E402

DEV = "cuda"
TOL = 1e-2      # max|d| / max|ref| on log-probs
L2_TOL = 3e-3   # relative L2 on log-probs
KV_TOL = 6e-2   # relative L2 on the K/V memory rows written by the chunk
V_TOL = 0.25    # absolute, on the raw value-head output (std ~1.3 with the synthetic weights; latent rel. error ~3e-2)


def _inputs(seed, b, t):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)


@pytest.fixture(scope="module")
def pol_1x():
    pk = O.policy_kwargs_for("1x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0))
    pol.load_state_dict(sd, strict=False)
    return pol.to(DEV), cfg, sd


def _rel(a, ref):
    return float(np.abs(a - ref).max() / np.abs(ref).max())


def _l2(a, ref):
    return float(np.linalg.norm((a - ref).ravel()) / max(np.linalg.norm(ref.ravel()), 1e-30))


def test_policy_chunks_vs_golden(pol_1x, golden_1x):
    pol, cfg, sd = pol_1x
    G = golden_1x
    b = 2
    state = pol.initial_state(b)
    report = {}
    for tag, t, first0 in [("A", 4, [False, True]), ("B", 3, [False, False]), ("C", 1, [False, False])]:
        img = _inputs(100 + ord(tag), b, t)
        first = torch.zeros(b, t, dtype=torch.bool)
        first[:, 0] = torch.tensor(first0)
        (pd, vpred, _), state = pol({"img": img.to(DEV)}, first.to(DEV), state)
        torch.cuda.synchronize()
        assert pd["buttons"].shape == (b, t, 1, 8641) and pd["camera"].shape == (b, t, 1, 121) and vpred.shape == (b, t, 1)
        assert list(pd.keys()) == ["camera", "buttons"]
        report[tag] = (_rel(pd["buttons"].cpu().numpy(), G[f"{tag}_buttons"]), _rel(pd["camera"].cpu().numpy(), G[f"{tag}_camera"]),
                       float(np.abs(vpred.cpu().numpy() - G[f"{tag}_vpred"]).max()),
                       _l2(pd["buttons"].cpu().numpy(), G[f"{tag}_buttons"]), _l2(pd["camera"].cpu().numpy(), G[f"{tag}_camera"]))
        for l, (m, (k, v)) in enumerate(state):
            assert m.dtype == torch.bool and m.shape == (b, 1, 128)
            assert np.array_equal(m.cpu().numpy(), G[f"{tag}_mask{l}"])
            assert k.dtype == torch.float32 and k.shape == (b, 128, 1024)
            assert _l2(k[:, -t:].cpu().numpy(), G[f"{tag}_Ktail{l}"][:, -t:]) < KV_TOL
            assert _l2(v[:, -t:].cpu().numpy(), G[f"{tag}_Vtail{l}"][:, -t:]) < KV_TOL
        assert torch.allclose(pd["buttons"].exp().sum(-1).cpu(), torch.ones(b, t, 1), atol=1e-3)
    print("PARITY vs golden (max/max buttons, max/max camera, |dv|, relL2 buttons, relL2 camera) per chunk:", report)
    for tag, (eb, ec, ev, lb, lc) in report.items():
        assert eb < TOL and ec < TOL and ev < V_TOL and lb < L2_TOL and lc < L2_TOL, report
    # act(): API shapes/dtypes + agreement of the deterministic action with the reference where the
    # reference's top-2 margin exceeds the bf16 noise (bit-exactness is only defined away from near-ties)
    img = _inputs(999, b, 1)[:, 0]
    ac, state2, res = pol.act({"img": img.to(DEV)}, torch.zeros(b, dtype=torch.bool, device=DEV), state, stochastic=False)
    assert ac["buttons"].dtype == torch.int64 and ac["buttons"].shape == (b, 1)
    assert res["log_prob"].shape == (b,) and res["vpred"].shape == (b, 1)
    assert np.abs(res["log_prob"].cpu().numpy() - G["act_log_prob"]).max() < 0.1


def test_policy_vs_oracle_long_chunk(pol_1x):
    """T = 40 with `first` on one sequence, then a second chunk with carried memory: exercises multiple
    query tiles of the attention kernel and the ring update at t < maxlen."""
    pol, cfg, sd = pol_1x
    b = 2
    so, sg = O.initial_state(cfg, b), pol.initial_state(b)
    for seed, t, first0 in [(21, 40, [True, False]), (22, 33, [False, False])]:
        img = _inputs(seed, b, t)
        first = torch.zeros(b, t, dtype=torch.bool)
        first[:, 0] = torch.tensor(first0)
        ref = O.policy_forward(sd, cfg, img, first, so)
        so = ref["state_out"]
        (pd, vpred, _), sg = pol({"img": img.to(DEV)}, first.to(DEV), sg)
        torch.cuda.synchronize()
        eb = _rel(pd["buttons"].cpu().numpy(), ref["buttons"].numpy())
        ec = _rel(pd["camera"].cpu().numpy(), ref["camera"].numpy())
        agree = (pd["buttons"].argmax(-1).cpu() == ref["buttons"].argmax(-1)).float().mean().item()
        lb = _l2(pd["buttons"].cpu().numpy(), ref["buttons"].numpy())
        lc = _l2(pd["camera"].cpu().numpy(), ref["camera"].numpy())
        print(f"PARITY vs oracle t={t}: max/max buttons {eb:.3e} camera {ec:.3e}; relL2 buttons {lb:.3e} camera {lc:.3e}; "
              f"argmax agreement {agree:.3f}")
        assert eb < TOL and ec < TOL and lb < L2_TOL and lc < L2_TOL
        for (m1, (k1, v1)), (m2, (k2, v2)) in zip(sg, so):
            assert torch.equal(m1.cpu(), m2)
            assert _l2(k1.cpu().numpy(), k2.numpy()) < KV_TOL and _l2(v1.cpu().numpy(), v2.numpy()) < KV_TOL


def test_first_resets_memory(pol_1x):
    """first[:,0]=True makes the output independent of the incoming memory."""
    pol, cfg, sd = pol_1x
    img = _inputs(31, 1, 3).to(DEV)
    first = torch.tensor([[True, False, False]], device=DEV)
    junk = [(torch.ones(1, 1, 128, dtype=torch.bool, device=DEV), (torch.randn(1, 128, 1024, device=DEV), torch.randn(1, 128, 1024, device=DEV)))
            for _ in range(4)]
    (pd1, _, _), _ = pol({"img": img}, first, junk)
    (pd2, _, _), _ = pol({"img": img}, first, pol.initial_state(1))
    torch.cuda.synchronize()
    assert torch.equal(pd1["buttons"], pd2["buttons"])


def test_policy_full_chunk_t128(pol_1x):
    """One full training-size chunk (T = 128, B = 2) against the oracle: all four query tiles of the band, the
    memory fully replaced by the chunk, then a T = 1 step on the carried state (the run_agent.py shape)."""
    pol, cfg, sd = pol_1x
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    b, t = 2, 128
    img = _inputs(41, b, t)
    first = torch.zeros(b, t, dtype=torch.bool)
    ref = O.policy_forward(sd, cfg, img, first, O.initial_state(cfg, b))
    (pd, vpred, _), sg = pol({"img": img.to(DEV)}, first.to(DEV), pol.initial_state(b))
    torch.cuda.synchronize()
    lb = _l2(pd["buttons"].cpu().numpy(), ref["buttons"].numpy())
    lc = _l2(pd["camera"].cpu().numpy(), ref["camera"].numpy())
    print(f"PARITY vs oracle t=128: relL2 buttons {lb:.3e} camera {lc:.3e}")
    assert lb < L2_TOL and lc < L2_TOL
    for (m1, (k1, v1)), (m2, (k2, v2)) in zip(sg, ref["state_out"]):
        assert torch.equal(m1.cpu(), m2) and bool(m2.all())
        assert _l2(k1.cpu().numpy(), k2.numpy()) < KV_TOL
    img1 = _inputs(42, b, 1)
    f1 = torch.zeros(b, 1, dtype=torch.bool)
    ref1 = O.policy_forward(sd, cfg, img1, f1, ref["state_out"])
    (pd1, _, _), _ = pol({"img": img1.to(DEV)}, f1.to(DEV), sg)
    torch.cuda.synchronize()
    assert _l2(pd1["buttons"].cpu().numpy(), ref1["buttons"].numpy()) < L2_TOL


def test_step_graph_matches_eager(pol_1x):
    """The captured T = 1 hipGraph (enable_step_graph) must reproduce the eager act() loop step by step, including a
    `first` reset in the middle of the episode, a state handed in from outside, and a re-capture after a weight update."""
    pol, cfg, sd = pol_1x
    n = 7
    frames = _inputs(77, n, 1).to(DEV)       # [n, 1, 128, 128, 3]: frame i is frames[i] with B = 1
    firsts = [False, False, False, True, False, False, False]

    def rollout():
        st = pol.initial_state(1)
        outs = []
        for i in range(n):
            ac, st, res = pol.act({"img": frames[i]}, torch.tensor([firsts[i]], device=DEV), st, stochastic=False, return_pd=True)
            outs.append((int(ac["buttons"]), int(ac["camera"]), res["pd"]["buttons"].clone(), res["pd"]["camera"].clone(), float(res["vpred"])))
        torch.cuda.synchronize()
        return outs, [(m.clone(), (k.clone(), v.clone())) for m, (k, v) in st]

    eager, st_e = rollout()
    pol.enable_step_graph(1)
    try:
        graphed, st_g = rollout()
        graphed2, _ = rollout()                 # second episode: initial_state copied over the aliased buffers
        for e, g_, g2 in zip(eager, graphed, graphed2):
            assert e[0] == g_[0] == g2[0] and e[1] == g_[1] == g2[1]
            assert torch.allclose(e[2], g_[2], atol=2e-4) and torch.allclose(e[3], g_[3], atol=2e-4) and abs(e[4] - g_[4]) < 1e-3
            assert torch.allclose(e[2], g2[2], atol=2e-4)
        for (me, (ke, ve)), (mg, (kg, vg)) in zip(st_e, st_g):
            assert torch.equal(me, mg) and torch.allclose(ke, kg, atol=1e-4) and torch.allclose(ve, vg, atol=1e-4)
        # other shapes still take the eager path
        (pd, _, _), _ = pol({"img": _inputs(78, 2, 3).to(DEV)}, torch.zeros(2, 3, dtype=torch.bool, device=DEV), pol.initial_state(2))
        assert pd["buttons"].shape == (2, 3, 1, 8641)
        # a parameter update invalidates the captured weights: the next step re-captures
        with torch.no_grad():
            pol.pi_head.buttons.linear_layer.bias[:200].add_(2.0)     # (a uniform shift would cancel in the log-softmax)
        after, _ = rollout()
        pol.disable_step_graph()
        eager_after, _ = rollout()
        assert torch.allclose(after[-1][2], eager_after[-1][2], atol=2e-4)
        assert not torch.allclose(after[-1][2], eager[-1][2], atol=1e-3)
    finally:
        pol.disable_step_graph()
        with torch.no_grad():
            pol.pi_head.buttons.linear_layer.bias[:200].sub_(2.0)


@pytest.mark.parametrize("b,ts,firsts", [
    (3, (1, 1, 1), ([False, True, False], [False, False, False], [True, False, False])),   # the acting shape, three envs, resets
    (1, (129, 2), ([False], [False])),                                                     # one frame more than the memory
    (2, (257, 64), ([True, False], [False, True])),                                        # > 2 memories long, then a reset
])
def test_policy_vs_oracle_ragged_shapes(pol_1x, b, ts, firsts):
    """Chunk lengths that are not multiples of anything in the kernels (query tile 32, memory 128), several chunks with
    the state carried, `first` raised on different sequences at different chunks."""
    pol, cfg, sd = pol_1x
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    so, sg = O.initial_state(cfg, b), pol.initial_state(b)
    for i, (t, first0) in enumerate(zip(ts, firsts)):
        img = _inputs(100 + 7 * i + t, b, t)
        first = torch.zeros(b, t, dtype=torch.bool)
        first[:, 0] = torch.tensor(first0)
        ref = O.policy_forward(sd, cfg, img, first, so)
        so = ref["state_out"]
        (pd, vpred, _), sg = pol({"img": img.to(DEV)}, first.to(DEV), sg)
        torch.cuda.synchronize()
        for k in ("buttons", "camera"):
            e, l = _rel(pd[k].cpu().numpy(), ref[k].numpy()), _l2(pd[k].cpu().numpy(), ref[k].numpy())
            assert e < TOL and l < L2_TOL, (k, t, e, l)
        assert float((vpred.cpu() - ref["vpred"]).abs().max()) < 0.25
        for (m1, (k1, v1)), (m2, (k2, v2)) in zip(sg, so):
            assert torch.equal(m1.cpu(), m2)
            assert _l2(k1.cpu().numpy(), k2.numpy()) < KV_TOL and _l2(v1.cpu().numpy(), v2.numpy()) < KV_TOL
