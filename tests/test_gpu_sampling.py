"""Stochastic action sampling on the device: CategoricalActionHead.sample with the uniforms drawn INSIDE the
head kernel (Philox4x32-10 on a device-resident {seed, step}), which is what lets the unmodified MineRLAgent.get_action --
policy.act(..., stochastic=True), agent.py:201-204 -- run as one hipGraph replay per step.  Needs an MI355X.

  * the kernel's uniforms == oracle/philox.py (pinned on Random123's known answers by tests/test_philox_cpu.py), bit for bit;
  * the in-kernel draw == the same head kernel fed those uniforms through its `noise` argument (the eager Gumbel path), exactly;
  * the sampled indices follow exp(log_prob): chi-square over 4096 draws on a small head, and == the oracle's Gumbel arg-max;
  * act(stochastic=True): captured automatically from the third same-shape call on, same actions as the eager loop on the same seed,
    a fresh draw per replay, the step counter advanced once per step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd import ops  # noqa: E402
from vpt_amd.lib.policy import MinecraftAgentPolicy  # noqa: E402
from vpt_amd.lib.types import minecraft_action_space  # noqa: E402
from oracle import philox as PH  # noqa: E402
from oracle import vpt_oracle as O  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


@pytest.mark.parametrize("seed,step,stream,rows,n", [(0x1234567890ABCDEF, 0, 0, 3, 8641), (7, 2 ** 33 + 5, 1, 64, 121), (2 ** 62 - 1, 1, 1, 1, 5)])
def test_kernel_uniforms_equal_the_philox_oracle(seed, step, stream, rows, n):
    st = ops.new_rng_state(DEV, seed=seed)
    st[1] = step
    u = ops.uniform_noise(st, stream, rows, n).cpu().numpy()
    assert np.array_equal(u, PH.head_uniforms(seed, step, stream, rows, n))
    assert int(st[1]) == step                      # drawing does not advance the state


def test_in_kernel_draw_equals_the_noise_argument_path_and_the_distribution():
    rows, n, ld = 4096, 16, 24
    g = torch.Generator().manual_seed(3)
    logits_row = torch.randn(n, generator=g) * 1.5
    logits = torch.zeros(rows, ld)
    logits[:, 4:4 + n] = logits_row                 # the same distribution on every row; every row draws its own uniforms
    logits = logits.to(DEV)
    st = ops.new_rng_state(DEV, seed=99)
    lp, ac, alp = ops.log_softmax_cols(logits, 4, n, 2.0, want_action=True, rng=(st, 1))
    u = ops.uniform_noise(st, 1, rows, n)
    lp2, ac2, alp2 = ops.log_softmax_cols(logits, 4, n, 2.0, want_action=True, noise=u)
    torch.cuda.synchronize()
    assert torch.equal(ac, ac2) and torch.equal(alp, alp2) and torch.equal(lp, lp2)
    assert torch.equal(alp, lp.gather(1, ac[:, None])[:, 0])
    # the oracle's Gumbel arg-max on the kernel's own log-probs and the oracle's uniforms
    want = PH.gumbel_argmax(lp.cpu().numpy(), PH.head_uniforms(99, 0, 1, rows, n))
    agree = float((ac.cpu().numpy() == want).mean())
    assert agree >= 0.999, agree                    # (logf / expf last-bit differences can flip a near-tie)
    p = torch.softmax(logits_row.double() / 2.0, 0).numpy()
    counts = np.bincount(ac.cpu().numpy(), minlength=n)
    chi2 = float(((counts - rows * p) ** 2 / (rows * p)).sum())
    print(f"SAMPLING: chi-square of {rows} in-kernel Gumbel-max draws over {n} classes vs exp(log_prob): {chi2:.1f} (0.1 % point 37.7)")
    assert chi2 < 37.7, (counts, chi2)              # 15 degrees of freedom
    # another step / another stream: different draws
    st[1] += 1
    _, ac3, _ = ops.log_softmax_cols(logits, 4, n, 2.0, want_action=True, rng=(st, 1))
    _, ac4, _ = ops.log_softmax_cols(logits, 4, n, 2.0, want_action=True, rng=(st, 0))
    assert not torch.equal(ac3, ac) and not torch.equal(ac4, ac3)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_act_stochastic_is_captured_automatically_and_matches_the_eager_draws(mode):
    pk = O.policy_kwargs_for("1x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision=mode)
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(DEV)
    n = 9
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (n, 1, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)
    first = torch.zeros(1, dtype=torch.bool, device=DEV)

    def rollout():
        pol.seed_sampler(4242)
        st, outs = pol.initial_state(1), []
        for i in range(n):
            ac, st, res = pol.act({"img": frames[i]}, first, st, stochastic=True, return_pd=True)   # what MineRLAgent.get_action calls
            lp = sum(res["pd"][k].gather(-1, ac[k].unsqueeze(-1))[:, 0, 0] for k in ("buttons", "camera"))
            assert torch.allclose(res["log_prob"], lp, atol=1e-5)
            outs.append((int(ac["buttons"]), int(ac["camera"]), float(res["log_prob"]), float(res["vpred"])))
        torch.cuda.synchronize()
        return outs

    pol.disable_step_graph()
    eager = rollout()
    assert pol._step_graph is None
    assert int(pol._engine._rng_state[1]) == n            # one step of the counter per act()
    pol.auto_step_graph(True)
    auto = rollout()                                                      # calls 1-2 eager, 3.. replay the captured stochastic step
    assert pol._step_graph is not None and "stochastic" in pol._step_graph["graphs"] and "deterministic" not in pol._step_graph["graphs"]
    assert int(pol._engine._rng_state[1]) == n
    again = rollout()                                                     # every step a replay; a new episode's state copied in
    same = sum(e[:2] == a[:2] == b[:2] for e, a, b in zip(eager, auto, again))
    print(f"SAMPLING[{mode}]: act(stochastic=True), auto-captured graph vs eager on the same seed: {same}/{n} identical action pairs; "
          f"distinct button actions {len(set(e[0] for e in eager))}")
    assert same == n
    for e, a in zip(eager, auto):
        assert abs(e[2] - a[2]) < 1e-3 and abs(e[3] - a[3]) < 1e-3 * max(1.0, abs(e[3]))
    assert len(set(e[0] for e in eager)) >= n - 1                        # near-uniform 8641-way head: the draws differ from step to step
    # a different seed draws differently; the deterministic mode is its own graph over the same state buffers
    pol.seed_sampler(1)
    st = pol.initial_state(1)
    ac, st, _ = pol.act({"img": frames[0]}, first, st, stochastic=True)
    assert int(ac["buttons"]) != eager[0][0] or int(ac["camera"]) != eager[0][1]
    acd, _, resd = pol.act({"img": frames[1]}, first, st, stochastic=False, return_pd=True)
    assert "deterministic" in pol._step_graph["graphs"]
    assert torch.equal(acd["buttons"], resd["pd"]["buttons"].argmax(-1))


def test_torch_manual_seed_at_any_time_restarts_the_draws():
    """the reference's th.rand_like (lib/action_head.py:200) is made reproducible by torch.manual_seed() at ANY time -- seed, run an
    episode, re-seed, run again.  The in-kernel sampler follows the device generator: a re-seed (even with the same value) since the last
    stochastic call re-derives {seed, step = 0} in place, eagerly and under the auto-captured graph; seed_sampler() detaches it from torch's
    generator; set_precision() keeps the sampler; seeds >= 2^63 are masked instead of overflowing."""
    pk = O.policy_kwargs_for("1x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision="fp16")
    pol.load_state_dict(O.synthetic_state_dict(cfg, seed=0), strict=False)
    pol = pol.to(DEV)
    n = 6
    frames = torch.randint(0, 256, (n, 1, 128, 128, 3), generator=torch.Generator().manual_seed(6), dtype=torch.uint8).to(DEV)
    first = torch.zeros(1, dtype=torch.bool, device=DEV)

    def episode():
        st, outs = pol.initial_state(1), []
        for i in range(n):
            ac, st, _ = pol.act({"img": frames[i]}, first, st, stochastic=True)
            outs.append((int(ac["buttons"]), int(ac["camera"])))
        return outs

    # a deterministic capture must not create (or consume anything for) the sampler
    st = pol.initial_state(1)
    for i in range(4):
        _, st, _ = pol.act({"img": frames[i]}, first, st, stochastic=False)
    assert pol._step_graph is not None and pol._engine._rng_state is None
    torch.manual_seed(1234)
    a = episode()                      # (calls 1-2 eager, then the captured stochastic step)
    b = episode()                      # no re-seed: the draws go on
    torch.manual_seed(1234)
    c = episode()                      # re-seeded with the SAME value between two graphed steps: the episode repeats
    torch.manual_seed(99)
    d = episode()
    assert a == c and a != b and a != d, (a, b, c, d)
    pol.disable_step_graph()
    torch.manual_seed(1234)
    e = episode()                      # eager launches draw what the graph drew
    assert e == a
    pol.seed_sampler(2 ** 63 + 5)      # masked to 63 bits, and from here on independent of torch's generator
    f = episode()
    torch.manual_seed(1234)
    pol.seed_sampler(2 ** 63 + 5)
    g = episode()
    assert f == g and f != a
    pol.seed_sampler(7)
    h = episode()
    pol.seed_sampler(7)
    pol.set_precision("bf16")          # a new engine: seed and step counter carry over
    assert pol._engine._rng_state is not None and int(pol._engine._rng_state[0]) == 7
    pol.auto_step_graph(True)
