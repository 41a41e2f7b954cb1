"""End-to-end parity of the HIP policy (through the reference's MinecraftAgentPolicy API) against the golden
vectors of the live reference and against the oracle, in BOTH precision modes.  Needs an MI355X.

What is gated and why is in tests/parity.py: log-probs as returned AND centred logits, latent, value (relative), KV
memory, and exact deterministic actions outside the noise band.  precision="fp16" is held to the north star's 1e-3 on the
log-probs; "bf16" (the benchmarked default) to bounds calibrated by the CPU emulator of its rounding points."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd.lib.policy import MinecraftAgentPolicy  # noqa: E402
from vpt_amd.lib.types import minecraft_action_space  # noqa: E402
from oracle import vpt_oracle as O  # noqa: E402
from tests import parity as P  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _inference_mode():
    """These tests exercise the inference engine (what act() / bench.py run); a grad-enabled call takes the autograd
    boundary instead (same kernels, activations kept) -- tests/test_gpu_training.py covers that path."""
    with torch.no_grad():
        yield


def _inputs(seed, b, t):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)


@pytest.fixture(scope="module", params=["bf16", "fp16"])
def pol_1x(request):
    pk = O.policy_kwargs_for("1x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision=request.param)
    pol.load_state_dict(sd, strict=False)
    return pol.to(DEV), cfg, sd


def _rel(a, ref):
    return P.rel_max(a, ref)


def _l2(a, ref):
    return P.rel_l2(a, ref)


def test_policy_chunks_vs_golden(pol_1x, golden_1x):
    pol, cfg, sd = pol_1x
    mode, B = pol.precision, P.BOUNDS[pol.precision]
    G = golden_1x
    b = 2
    state = pol.initial_state(b)
    for tag, t, first0 in [("A", 4, [False, True]), ("B", 3, [False, False]), ("C", 1, [False, False])]:
        img = _inputs(100 + ord(tag), b, t)
        first = torch.zeros(b, t, dtype=torch.bool)
        first[:, 0] = torch.tensor(first0)
        (pd, vpred, _), state = pol({"img": img.to(DEV)}, first.to(DEV), state)
        torch.cuda.synchronize()
        assert pd["buttons"].shape == (b, t, 1, 8641) and pd["camera"].shape == (b, t, 1, 121) and vpred.shape == (b, t, 1)
        assert list(pd.keys()) == ["camera", "buttons"]
        m = P.policy_metrics(dict(buttons=pd["buttons"], camera=pd["camera"], vpred=vpred),
                             dict(buttons=G[f"{tag}_buttons"], camera=G[f"{tag}_camera"], vpred=G[f"{tag}_vpred"]))
        print(f"PARITY[{mode}] vs golden chunk {tag}: {P.fmt(m)}")
        P.check(m, mode, f"golden chunk {tag}", model="1x")
        for l, (mk, (k, v)) in enumerate(state):
            assert mk.dtype == torch.bool and mk.shape == (b, 1, 128)
            assert np.array_equal(mk.cpu().numpy(), G[f"{tag}_mask{l}"])
            assert k.dtype == torch.float32 and k.shape == (b, 128, 1024)
            assert _l2(k[:, -t:], G[f"{tag}_Ktail{l}"][:, -t:]) < B["kv_l2"]
            assert _l2(v[:, -t:], G[f"{tag}_Vtail{l}"][:, -t:]) < B["kv_l2"]
        assert torch.allclose(pd["buttons"].exp().sum(-1).cpu(), torch.ones(b, t, 1), atol=1e-3)
    # act(): API shapes / dtypes; log_prob of the deterministic action against the reference's
    img = _inputs(999, b, 1)[:, 0]
    ac, state2, res = pol.act({"img": img.to(DEV)}, torch.zeros(b, dtype=torch.bool, device=DEV), state, stochastic=False)
    assert ac["buttons"].dtype == torch.int64 and ac["buttons"].shape == (b, 1)
    assert res["log_prob"].shape == (b,) and res["vpred"].shape == (b, 1)
    assert np.abs(res["log_prob"].cpu().numpy() - G["act_log_prob"]).max() < (0.02 if mode == "fp16" else 0.1)


def test_deterministic_actions_equal_reference(pol_1x):
    """act(stochastic=False) over 64 frames with carried state: integer action indices must EQUAL the oracle's argmax
    wherever the oracle's top-2 margin exceeds 4x the measured max log-prob error of the head (a16: lib/action_head.py:195-197)."""
    pol, cfg, sd = pol_1x
    mode = pol.precision
    b, t = 4, 16
    img = P.structured_frames(b, t, torch.Generator().manual_seed(555))     # low-frequency content: the arg-max varies from frame to frame
    first = torch.zeros(b, t, dtype=torch.bool)
    ref = O.policy_forward(sd, cfg, img, first, O.initial_state(cfg, b))
    st = pol.initial_state(b)
    acts = {"buttons": [], "camera": []}
    pds = {"buttons": [], "camera": []}
    for i in range(t):
        ac, st, res = pol.act({"img": img[:, i].to(DEV)}, first[:, i].to(DEV), st, stochastic=False, return_pd=True)
        for h in acts:
            acts[h].append(ac[h].cpu()); pds[h].append(res["pd"][h].cpu())
    torch.cuda.synchronize()
    for h in acts:
        got = torch.stack(acts[h], 1)[:, :, 0]              # [b, t]
        logp = torch.stack(pds[h], 1)                        # [b, t, 1, n]
        m = P.head_metrics(logp, ref[h])
        want = ref[h].argmax(-1)[:, :, 0]
        err = m["max_abs_err"]
        top2 = ref[h].topk(2, -1).values[:, :, 0]
        safe = (top2[..., 0] - top2[..., 1]) > 4 * err
        print(f"ACTIONS[{mode}] {h}: agree {float((got == want).float().mean()):.3f} overall, {int(safe.sum())}/{safe.numel()} positions outside "
              f"the noise band (4 x {err:.2e}), mismatches there: {int(((got != want) & safe).sum())}; distinct oracle actions {len(set(want.flatten().tolist()))}")
        assert bool((got[safe] == want[safe]).all())
        if mode == "fp16":
            assert float(safe.float().mean()) > 0.5, "fp16 mode should resolve most positions"


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
        B = P.BOUNDS[pol.precision]
        m = P.policy_metrics(dict(buttons=pd["buttons"], camera=pd["camera"], vpred=vpred), ref)
        print(f"PARITY[{pol.precision}] vs oracle t={t}: {P.fmt(m)}")
        P.check(m, pol.precision, f"long chunk t={t}", model="1x")
        for (m1, (k1, v1)), (m2, (k2, v2)) in zip(sg, so):
            assert torch.equal(m1.cpu(), m2)
            assert _l2(k1, k2) < B["kv_l2"] and _l2(v1, v2) < B["kv_l2"]


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


def test_a_sequence_result_does_not_depend_on_the_batch_around_it(pol_1x):
    """A frame's output must not depend on how many frames share its launches: every statistic is fp32 inside a
    tile / tile group whose extent is fixed and fp64 across them, whatever the batch, the CNN chunk or the stream it lands on.  The same
    two sequences alone (B = 2), inside a batch of six at other positions, and with the CNN cut into uneven chunks: BIT-identical
    log-probs and KV memory.  (Covers the pool-fused convolution, its seam kernel and the folded GroupNorm `n` with its in-kernel
    per-channel sums -- the inference path's newest pieces.)"""
    pol, cfg, sd = pol_1x
    t = 5
    a, other = _inputs(301, 2, t), _inputs(302, 4, t)
    big = torch.stack([other[0], a[0], other[1], other[2], a[1], other[3]])
    f2, f6 = torch.zeros(2, t, dtype=torch.bool, device=DEV), torch.zeros(6, t, dtype=torch.bool, device=DEV)
    (pd_a, v_a, _), st_a = pol({"img": a.to(DEV)}, f2, pol.initial_state(2))
    (pd_b, v_b, _), st_b = pol({"img": big.to(DEV)}, f6, pol.initial_state(6))
    eng = pol._engine
    saved = eng.cnn_chunk
    try:
        eng.cnn_chunk = 7                       # 30 frames as 7 + 7 + 7 + 7 + 2 on three streams (chunks of <= 8 frames: still the MFMA kernels)
        (pd_c, v_c, _), st_c = pol({"img": big.to(DEV)}, f6, pol.initial_state(6))
    finally:
        eng.cnn_chunk = saved
    torch.cuda.synchronize()
    idx = torch.tensor([1, 4], device=DEV)
    for pd_x, v_x, st_x in ((pd_b, v_b, st_b), (pd_c, v_c, st_c)):
        for h in ("buttons", "camera"):
            assert torch.equal(pd_a[h], pd_x[h][idx]), (h, float((pd_a[h] - pd_x[h][idx]).abs().max()), [float((pd_a[h][:, j] - pd_x[h][idx][:, j]).abs().max()) for j in range(t)])
        assert torch.equal(v_a, v_x[idx])
        for (m1, (k1, v1)), (m2, (k2, v2)) in zip(st_a, st_x):
            assert torch.equal(k1, k2[idx]) and torch.equal(v1, v2[idx])


def test_a_sequence_result_does_not_depend_on_the_row_count_of_the_linears(pol_1x):
    """ADVICE r4 / VERDICT r4 item 6: the trunk linears' split-K used to be chosen from M (<= 256, 257..512, > 512 rows summed K in three different
    orders), so a sequence alone (B = 1, T = 128: M = 128) and the same sequence inside a larger batch (M = 640) were not bit-identical.  The K
    order is now a function of the named tiling and the layer only (ops.linear: splitk is the caller's, "nk" or an int, never M's).  Also two CNN
    chunk sizes straddling 512 rows (256 and 1024): the dense layer's split-K is the engine's constant."""
    pol, cfg, sd = pol_1x
    t = 128
    a, other = _inputs(311, 1, t), _inputs(312, 4, t)
    big = torch.cat([other[:2], a, other[2:]])
    f1, f5 = torch.zeros(1, t, dtype=torch.bool, device=DEV), torch.zeros(5, t, dtype=torch.bool, device=DEV)
    eng = pol._engine
    saved = eng.cnn_chunk
    res = []
    try:
        for chunk, img, f in ((1024, a, f1), (1024, big, f5), (256, big, f5), (256, a, f1)):
            eng.cnn_chunk = chunk
            (pd, v, _), st = pol({"img": img.to(DEV)}, f, pol.initial_state(img.shape[0]))
            i = 0 if img.shape[0] == 1 else 2
            res.append(({h: pd[h][i].clone() for h in ("buttons", "camera")}, v[i].clone(), [(k[i].clone(), vv[i].clone()) for _, (k, vv) in st]))
    finally:
        eng.cnn_chunk = saved
    torch.cuda.synchronize()
    for pd_x, v_x, st_x in res[1:]:
        for h in ("buttons", "camera"):
            assert torch.equal(res[0][0][h], pd_x[h]), (h, float((res[0][0][h] - pd_x[h]).abs().max()))
        assert torch.equal(res[0][1], v_x)
        for (k1, v1), (k2, v2) in zip(res[0][2], st_x):
            assert torch.equal(k1, k2) and torch.equal(v1, v2)


def test_overlapped_steps_equal_serial_steps(pol_1x):
    """PolicyEngine.overlap_steps(): with it the chunk streams of call i + 1 wait for call i's hand-off event (CNN output consumed) instead of
    for the calling stream, so they run beside call i's transformer.  Same kernels, same data, same per-stream order: five consecutive calls with
    the state carried -- different frames each, three ragged chunks on three streams -- must equal the serial ordering BIT FOR BIT, in either
    order of running the two modes, and a re-pack in between (new weights) must be waited for."""
    pol, cfg, sd = pol_1x
    b, t, n_calls = 2, 20, 5                             # 40 frames per call: chunks of 16, 16, 8
    imgs = [_inputs(700 + i, b, t).to(DEV) for i in range(n_calls)]
    first = torch.zeros(b, t, dtype=torch.bool, device=DEV)
    first[1, 7] = True
    eng = pol._engine
    saved = eng.cnn_chunk

    def run(overlap):
        pol.overlap_steps(overlap)
        st, rec = pol.initial_state(b), []
        for i in range(n_calls):
            (pd, v, _), st = pol({"img": imgs[i]}, first, st)
            rec.append((pd["buttons"].clone(), pd["camera"].clone(), v.clone()))
        torch.cuda.synchronize()
        return rec, [(k.clone(), vv.clone()) for _, (k, vv) in st]

    try:
        eng.cnn_chunk = 16
        assert min(eng.cnn_streams, 3) > 1, "the test needs more than one chunk stream"
        serial, st_s = run(False)
        piped, st_p = run(True)
        assert pol._engine._handoff is not None          # the pipelined ordering was actually taken
        serial2, _ = run(False)
        for a, c, d in zip(serial, piped, serial2):
            for x, y, z in zip(a, c, d):
                assert torch.equal(x, y) and torch.equal(x, z)
        for (k1, v1), (k2, v2) in zip(st_s, st_p):
            assert torch.equal(k1, k2) and torch.equal(v1, v2)
        # a resident but NON-contiguous img under overlap: forward() makes the contiguous copy on the calling stream, which the pipelined
        # chunk streams do not wait for by themselves -- they are handed an event behind the copy
        wide = [torch.cat([im, torch.flip(im, dims=[-1])], dim=-1) for im in imgs]       # [b, t, 128, 128, 6]: the frames are the slice [..., :3]
        pol.overlap_steps(True)
        st, rec = pol.initial_state(b), []
        for i in range(n_calls):
            view = wide[i][..., :3]
            assert not view.is_contiguous()
            (pd, v, _), st = pol({"img": view}, first, st)
            rec.append((pd["buttons"].clone(), pd["camera"].clone(), v.clone()))
        torch.cuda.synchronize()
        for a, c in zip(serial, rec):
            for x, y in zip(a, c):
                assert torch.equal(x, y)
        # new weights between two overlapped calls: the re-pack runs on the calling stream and the next call must wait for it
        pol.overlap_steps(True)
        (pd0, _, _), _ = pol({"img": imgs[0]}, first, pol.initial_state(b))
        w = dict(pol.named_parameters())["net.img_process.cnn.stacks.1.blocks.0.conv0.layer.weight"]
        orig = w.detach().clone()
        try:
            with torch.no_grad():
                w.mul_(1.5)
            (pd1, _, _), _ = pol({"img": imgs[0]}, first, pol.initial_state(b))
            pol.overlap_steps(False)
            (pd2, _, _), _ = pol({"img": imgs[0]}, first, pol.initial_state(b))
            torch.cuda.synchronize()
            assert torch.equal(pd1["buttons"], pd2["buttons"]) and not torch.equal(pd0["buttons"], pd1["buttons"])
        finally:
            with torch.no_grad():
                w.copy_(orig)                          # (the fixture is shared by the module: restore the exact bits)
    finally:
        pol._engine.cnn_chunk = saved
        pol.overlap_steps(False)


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
    B = P.BOUNDS[pol.precision]
    m = P.policy_metrics(dict(buttons=pd["buttons"], camera=pd["camera"], vpred=vpred), ref)
    print(f"PARITY[{pol.precision}] vs oracle t=128: {P.fmt(m)}")
    P.check(m, pol.precision, "t=128", model="1x")
    for (m1, (k1, v1)), (m2, (k2, v2)) in zip(sg, ref["state_out"]):
        assert torch.equal(m1.cpu(), m2) and bool(m2.all())
        assert _l2(k1, k2) < B["kv_l2"]
    img1 = _inputs(42, b, 1)
    f1 = torch.zeros(b, 1, dtype=torch.bool)
    ref1 = O.policy_forward(sd, cfg, img1, f1, ref["state_out"])
    (pd1, _, _), _ = pol({"img": img1.to(DEV)}, f1.to(DEV), sg)
    torch.cuda.synchronize()
    assert _l2(pd1["buttons"], ref1["buttons"]) < B["lp_l2"]


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

    pol.disable_step_graph()                    # (act() would capture the step by itself from the third call on)
    eager, st_e = rollout()
    pol.enable_step_graph(1)
    try:
        graphed, st_g = rollout()
        graphed2, _ = rollout()                 # second episode: initial_state copied over the aliased buffers
        # the graph against the ORACLE (not only against our own eager launches): the same 7 steps, T = 1 each, state carried,
        # `first` raised at step 3 -- log-probs within the mode's bounds, value de-normalised as lib/normalize_ewma.py does
        so = O.initial_state(cfg, 1)
        for i, g_ in enumerate(graphed):
            ref = O.policy_forward(sd, cfg, frames[i].cpu()[None], torch.tensor([[firsts[i]]]), so)
            so = ref["state_out"]
            m = P.policy_metrics(dict(buttons=g_[2][None], camera=g_[3][None]), dict(buttons=ref["buttons"], camera=ref["camera"]))
            P.check(m, pol.precision, f"graphed step {i} vs oracle", model="1x")
            v_ref = float(O.denormalize_value(sd, "value_head.", ref["vpred"]).reshape(-1)[0])
            assert abs(g_[4] - v_ref) < P.BOUNDS[pol.precision]["v_rel_1x"] * max(1.0, abs(v_ref)) * 1.3, (i, g_[4], v_ref)
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
        pol.auto_step_graph(True)
        with torch.no_grad():
            pol.pi_head.buttons.linear_layer.bias[:200].sub_(2.0)


def test_auto_captured_step_returns_fresh_state_two_environments_share_one_policy(pol_1x):
    """act() captures the acting step BY ITSELF, so the graphed step must keep the reference's contract -- a fresh state_out
    every step (lib/policy.py:307-328).  Two environments that share one policy object and alternate B = 1 act() calls, and a caller that keeps
    an older state for rollback, must get what the eager path gives them.  (The explicit enable_step_graph() keeps its documented aliasing.)
    Also: a value-normaliser update after the capture re-captures (the (scale, shift) pair is a captured launch argument)."""
    pol, cfg, sd = pol_1x
    n = 6
    fa, fb = _inputs(91, n, 1).to(DEV), _inputs(92, n, 1).to(DEV)
    first = torch.zeros(1, dtype=torch.bool, device=DEV)

    def run():
        sa, sb, outs, keep = pol.initial_state(1), pol.initial_state(1), [], None
        for i in range(n):
            aa, sa, ra = pol.act({"img": fa[i]}, first, sa, stochastic=False, return_pd=True)
            ab, sb, rb = pol.act({"img": fb[i]}, first, sb, stochastic=False, return_pd=True)
            if i == 2:
                keep = sa                                  # a snapshot the caller holds on to (no clone: the reference returns fresh tensors)
            outs.append((ra["pd"]["buttons"].clone(), rb["pd"]["buttons"].clone(), float(ra["vpred"]), float(rb["vpred"])))
        # roll environment A back to the state after step 2 and replay step 3
        a3, _, r3 = pol.act({"img": fa[3]}, first, keep, stochastic=False, return_pd=True)
        torch.cuda.synchronize()
        return outs, r3["pd"]["buttons"].clone()

    pol.disable_step_graph()
    eager, eager_rb = run()
    pol.auto_step_graph(True)
    try:
        auto, auto_rb = run()
        assert pol._step_graph is not None and pol._step_graph["alias"] is False and "deterministic" in pol._step_graph["graphs"]
        for i, (e, a) in enumerate(zip(eager, auto)):
            assert torch.allclose(e[0], a[0], atol=2e-4) and torch.allclose(e[1], a[1], atol=2e-4), i
            assert abs(e[2] - a[2]) < 1e-3 * max(1.0, abs(e[2])) and abs(e[3] - a[3]) < 1e-3 * max(1.0, abs(e[3]))
        assert torch.allclose(eager_rb, auto_rb, atol=2e-4)
        assert torch.allclose(auto_rb, auto[3][0], atol=1e-6)          # the rollback reproduces step 3 of environment A
        assert not torch.allclose(auto[3][0], auto[3][1], atol=1e-3)   # (the two environments do see different things)
        # normaliser update -> vpred follows (re-capture), as the eager path does
        st = pol.initial_state(1)
        _, st, r0 = pol.act({"img": fa[0]}, first, st, stochastic=False)
        with torch.no_grad():
            pol.value_head.normalizer.running_mean.add_(3.0 * pol.value_head.normalizer.debiasing_term)
        _, _, r1 = pol.act({"img": fa[0]}, first, pol.initial_state(1), stochastic=False)
        assert pol._step_graph is not None and "deterministic" in pol._step_graph["graphs"]      # (still a graphed step)
        pol.disable_step_graph()
        _, _, r1e = pol.act({"img": fa[0]}, first, pol.initial_state(1), stochastic=False)
        assert abs(float(r1["vpred"]) - float(r1e["vpred"])) < 1e-3 * max(1.0, abs(float(r1e["vpred"]))), (float(r1["vpred"]), float(r1e["vpred"]))
        assert abs(float(r1["vpred"]) - float(r0["vpred"])) > 0.5, (float(r0["vpred"]), float(r1["vpred"]))
    finally:
        with torch.no_grad():
            pol.value_head.normalizer.running_mean.sub_(3.0 * pol.value_head.normalizer.debiasing_term)
        pol.disable_step_graph()
        pol.auto_step_graph(True)


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
        B = P.BOUNDS[pol.precision]
        P.check(P.policy_metrics(dict(buttons=pd["buttons"], camera=pd["camera"], vpred=vpred), ref), pol.precision, f"ragged t={t}", model="1x")
        for (m1, (k1, v1)), (m2, (k2, v2)) in zip(sg, so):
            assert torch.equal(m1.cpu(), m2)
            assert _l2(k1, k2) < B["kv_l2"] and _l2(v1, v2) < B["kv_l2"]


def test_logit_mask_vs_live_reference_golden(pol_1x):
    """obs["mask"] (lib/policy.py:257-266): masked actions get LOG0 before the softmax; golden from the unmodified reference."""
    import os
    pol, cfg, sd = pol_1x
    mode = pol.precision
    G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_1x_seed0.npz")))
    b, t = 2, 3
    img = _inputs(321, b, t)
    gm = torch.Generator().manual_seed(5)
    mb = torch.rand(b, t, 1, 8641, generator=gm) > 0.3
    mc = torch.rand(b, t, 1, 121, generator=gm) > 0.5
    mb[..., 0] = True
    mc[..., 60] = True
    obs = {"img": img.to(DEV), "mask": {"buttons": mb.to(DEV), "camera": mc.to(DEV)}}
    (pd, _, _), _ = pol(obs, torch.zeros(b, t, dtype=torch.bool, device=DEV), pol.initial_state(b))
    torch.cuda.synchronize()
    assert "mask" in obs                                             # the caller's dict is not mutated (lib/policy.py:255)
    for k, mk in (("buttons", mb), ("camera", mc)):
        got = pd[k].cpu()
        assert float(got[~mk].max()) < -90.0                         # LOG0 - logsumexp
        hm = P.head_metrics(got[mk], torch.from_numpy(G[k])[mk])     # on the available actions
        print(f"PARITY[{mode}] masked {k}: {P.fmt(hm)}")
        assert hm["lp_max"] < P.BOUNDS[mode]["lp_max"] and hm["lp_l2"] < P.BOUNDS[mode]["lp_l2"]
        assert bool(mk.gather(-1, got.argmax(-1, keepdim=True)).all())   # the arg-max is always an available action


def test_act_uses_fused_sampling(pol_1x):
    """act(): the action and its log-prob come out of the head kernel (a16); they must agree with the generic
    CategoricalActionHead algebra on the returned pd, in deterministic mode exactly."""
    pol, cfg, sd = pol_1x
    b = 3
    img = _inputs(808, b, 1)[:, 0].to(DEV)
    first = torch.zeros(b, dtype=torch.bool, device=DEV)
    ac, st, res = pol.act({"img": img}, first, pol.initial_state(b), stochastic=False, return_pd=True)
    for k in ("buttons", "camera"):
        assert ac[k].shape == (b, 1) and ac[k].dtype == torch.int64
        assert torch.equal(ac[k], res["pd"][k].argmax(-1))
    lp = sum(res["pd"][k].gather(-1, ac[k].unsqueeze(-1))[:, 0, 0] for k in ("buttons", "camera"))
    assert torch.allclose(res["log_prob"], lp, atol=1e-5)
    torch.manual_seed(0)
    ac_s, _, res_s = pol.act({"img": img}, first, pol.initial_state(b), stochastic=True, return_pd=True)
    lp_s = sum(res_s["pd"][k].gather(-1, ac_s[k].unsqueeze(-1))[:, 0, 0] for k in ("buttons", "camera"))
    assert torch.allclose(res_s["log_prob"], lp_s, atol=1e-5)
    # taken_action: log-prob of a given action through the generic path
    ac_t, _, res_t = pol.act({"img": img}, first, pol.initial_state(b), taken_action=ac)
    assert torch.allclose(res_t["log_prob"], res["log_prob"], atol=1e-5)
