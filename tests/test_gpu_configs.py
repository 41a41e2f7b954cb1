"""The BASELINE.json configurations against the fp32 oracle, in both precision modes (bounds: tests/parity.py):
  - 2x / 3x policy forward at reduced B x T (same architecture, widths and code paths as the full sizes);
  - config 2's real sequence shape: 2x, T = 128, then a second T = 128 chunk on the carried KV memory;
  - config 3: the 4x inverse-dynamics model on a 16-frame and on a full 128-frame window, actions gated;
  - config 5: 3x BC loss / gradients on two consecutive T = 256 chunks with the detached KV memory carried;
  - a 3x BC step at small B x T."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd.lib.policy import InverseActionPolicy, MinecraftAgentPolicy  # noqa: E402
from vpt_amd.lib.types import idm_action_space, minecraft_action_space  # noqa: E402
from vpt_amd.training import BCTrainer  # noqa: E402
from oracle import vpt_oracle as O  # noqa: E402
from tests import parity as P  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _inference_mode():
    """These tests exercise the inference engine (what act() / bench.py run); a grad-enabled call takes the autograd
    boundary instead (same kernels, activations kept) -- tests/test_gpu_training.py covers that path."""
    with torch.no_grad():
        yield


def _l2(a, ref):
    return P.rel_l2(a, ref)


def _threads():
    import os
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))


def _policy(model, precision="bf16"):
    pk = O.policy_kwargs_for(model)
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision=precision)
    missing, unexpected = pol.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    return pol.to(DEV), cfg, sd


@pytest.mark.parametrize("model,t", [("2x", 6), ("3x", 5)])
def test_policy_forward_wide_models(model, t):
    _threads()
    pol, cfg, sd = _policy(model)
    g = torch.Generator().manual_seed(21)
    b = 2
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(b, t, dtype=torch.bool)
    first[1, 0] = True
    ref = O.policy_forward(sd, cfg, img, first, O.initial_state(cfg, b))
    for mode in ("bf16", "fp16"):
        pol.set_precision(mode)
        (pd, vpred, _), state = pol({"img": img.to(DEV)}, first.to(DEV), pol.initial_state(b))
        torch.cuda.synchronize()
        m = P.policy_metrics(dict(buttons=pd["buttons"], camera=pd["camera"], vpred=vpred), ref)
        print(f"PARITY[{mode}] {model} forward: {P.fmt(m)}")
        P.check(m, mode, f"{model} forward")
        assert _l2(state[-1][1][0], ref["state_out"][-1][1][0]) < P.BOUNDS[mode]["kv_l2"]
        assert torch.equal(state[0][0].cpu(), ref["state_out"][0][0])


def test_config2_sequence_shape_2x_t128_with_carry():
    """BASELINE.json configs[1]'s sequence shape: the 2x model on a full T = 128 chunk, then a second T = 128 chunk that
    attends into the carried 128-frame KV memory (lib/xf.py:366-391) -- one sequence of the 64 the bench runs."""
    _threads()
    pol, cfg, sd = _policy("2x")
    g = torch.Generator().manual_seed(31)
    t = 128
    imgs = [torch.randint(0, 256, (1, t, 128, 128, 3), generator=g, dtype=torch.uint8) for _ in range(2)]
    first = torch.zeros(1, t, dtype=torch.bool)
    refs, so = [], O.initial_state(cfg, 1)
    for img in imgs:
        r = O.policy_forward(sd, cfg, img, first, so)
        so = r["state_out"]
        refs.append(r)
    for mode in ("bf16", "fp16"):
        pol.set_precision(mode)
        st = pol.initial_state(1)
        for i, (img, ref) in enumerate(zip(imgs, refs)):
            (pd, vpred, _), st = pol({"img": img.to(DEV)}, first.to(DEV), st)
            torch.cuda.synchronize()
            m = P.policy_metrics(dict(buttons=pd["buttons"], camera=pd["camera"], vpred=vpred), ref)
            print(f"PARITY[{mode}] config 2 (2x, T=128) chunk {i}: {P.fmt(m)}")
            P.check(m, mode, f"2x T=128 chunk {i}")
            for (m1, (k1, v1)), (m2, (k2, v2)) in zip(st, ref["state_out"]):
                assert torch.equal(m1.cpu(), m2)
                assert _l2(k1, k2) < P.BOUNDS[mode]["kv_l2"] and _l2(v1, v2) < P.BOUNDS[mode]["kv_l2"]


def test_prior_dominated_actions_at_config2_size_on_peaked_heads():
    """a16 at config-2 scale, the PRIOR / bias path: the 2x model over B x T = 8 x 128 frames (1024 positions per head) with
    "peaked" heads -- oracle/vpt_oracle.py:peak_heads: one class 4 nat ahead through its bias -- so the top-2 margin exceeds the noise
    band at >= 95 % of the positions in BOTH operand formats, and there the integer indices of deterministic act() (128 T = 1 steps,
    KV memory carried inside the loop, agent.py:190-206) must EQUAL the fp32 oracle's arg-max (lib/action_head.py:195-197).
    What this proves is the bias add, the log-softmax and the arg-max plumbing at scale: the prior decides (nearly) every position --
    the printed distinct-action count is 1-2.  INPUT-driven decisions are test_input_driven_actions_on_competitive_heads below."""
    _threads()
    pk = O.policy_kwargs_for("2x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0, heads="peaked")
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0))
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(DEV)
    b, t = 8, 128
    img = P.structured_frames(b, t, torch.Generator().manual_seed(808))
    first = torch.zeros(b, t, dtype=torch.bool)
    first[3, 0] = True
    ref = O.policy_forward(sd, cfg, img, first, O.initial_state(cfg, b))
    img_d, first_d = img.to(DEV), first.to(DEV)
    for mode in ("bf16", "fp16"):
        pol.set_precision(mode)
        st = pol.initial_state(b)
        acts = {"buttons": [], "camera": []}
        pds = {"buttons": [], "camera": []}
        for i in range(t):
            ac, st, res = pol.act({"img": img_d[:, i]}, first_d[:, i], st, stochastic=False, return_pd=True)
            for h in acts:
                acts[h].append(ac[h]); pds[h].append(res["pd"][h])
        torch.cuda.synchronize()
        for h in acts:
            got = torch.stack(acts[h], 1)[:, :, 0].cpu()              # [b, t] int64
            logp = torch.stack(pds[h], 1).cpu()                        # [b, t, 1, n]
            m = P.head_metrics(logp, ref[h])
            want = ref[h].argmax(-1)[:, :, 0]
            top2 = ref[h].topk(2, -1).values[:, :, 0]
            safe = (top2[..., 0] - top2[..., 1]) > 4 * m["max_abs_err"]
            agree = got == want
            print(f"ACTIONS[{mode}] 2x, {b}x{t} positions, peaked {h}: equal to the oracle at {int(agree.sum())}/{agree.numel()}; outside the noise band "
                  f"(4 x {m['max_abs_err']:.2e}): {int(safe.sum())}/{safe.numel()}, mismatches there {int((~agree & safe).sum())}; "
                  f"median top-2 margin {float((top2[..., 0] - top2[..., 1]).median()):.2f} nat; distinct actions {len(set(want.flatten().tolist()))}; {P.fmt(m)}")
            assert got.dtype == torch.int64 and bool((got[safe] == want[safe]).all())
            assert float(safe.float().mean()) >= 0.95, (mode, h, float(safe.float().mean()))
            assert float(agree.float().mean()) >= 0.95, (mode, h, float(agree.float().mean()))
            assert m["lp_l2"] < {"fp16": 2e-3, "bf16": 2e-2}[mode]     # peaked log-probs carry no -log N offset: this IS the centred error


def test_input_driven_actions_on_competitive_heads():
    """a16 where the INPUT decides: the 2x model over B x T = 8 x 128 frames drawn from 12 scenes (oracle.scene_frames), with the
    "competitive" head family (oracle.fit_scene_heads): 16 live classes per head, every other class 12 nat down with zero weights, the
    live ones a ridge-regression read-out of the scene the current frame shows, fitted on the oracle's latents of these very frames
    (target margin 4 nat; equal biases up to centring: nothing but the latent tells the classes apart).  The oracle's arg-max then
    takes >= 8 distinct values over the run, its top-2 margin has a median >= 0.5 nat, and in BOTH operand formats >= 90 % of the
    positions lie outside the noise band (4 x the head's measured max log-prob error) with ZERO mismatches there -- bit-exact integer
    actions on input-driven decisions under 16-bit noise, deterministic act() through 128 acting steps (auto-captured graph, KV
    memory carried), lib/action_head.py:195-197."""
    _threads()
    pk = O.policy_kwargs_for("2x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd0 = O.synthetic_state_dict(cfg, seed=0)
    b, t, n_scenes = 8, 128, 12
    img, scene = O.scene_frames(b, t, n_scenes, torch.Generator().manual_seed(606))
    first = torch.zeros(b, t, dtype=torch.bool)
    first[5, 0] = True
    trunk = O.policy_forward(sd0, cfg, img, first, O.initial_state(cfg, b))          # the heads do not feed back: one trunk pass serves both
    sd = O.fit_scene_heads(sd0, trunk["latent"].reshape(b * t, -1), scene.reshape(-1), 2.0)
    ref = {h: O.categorical_head(sd, f"pi_head.{h}.", trunk["latent"], 2.0).reshape(b, t, 1, -1) for h in ("buttons", "camera")}
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0))
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(DEV)
    img_d, first_d = img.to(DEV), first.to(DEV)
    for mode in ("bf16", "fp16"):
        pol.set_precision(mode)
        st = pol.initial_state(b)
        acts, pds = {"buttons": [], "camera": []}, {"buttons": [], "camera": []}
        for i in range(t):
            ac, st, res = pol.act({"img": img_d[:, i]}, first_d[:, i], st, stochastic=False, return_pd=True)
            for h in acts:
                acts[h].append(ac[h].clone()); pds[h].append(res["pd"][h].clone())
        torch.cuda.synchronize()
        for h in acts:
            got = torch.stack(acts[h], 1)[:, :, 0].cpu()
            logp = torch.stack(pds[h], 1).cpu()
            m = P.head_metrics(logp, ref[h])
            want = ref[h].argmax(-1)[:, :, 0]
            top2 = ref[h].topk(2, -1).values[:, :, 0]
            margin = top2[..., 0] - top2[..., 1]
            safe = margin > 4 * m["max_abs_err"]
            agree = got == want
            distinct = len(set(want.flatten().tolist()))
            print(f"ACTIONS[{mode}] 2x, {b}x{t} positions, competitive {h}: distinct oracle actions {distinct}; equal at {int(agree.sum())}/{agree.numel()}; outside the noise "
                  f"band (4 x {m['max_abs_err']:.2e}): {int(safe.sum())}/{safe.numel()}, mismatches there {int((~agree & safe).sum())}; top-2 margin median "
                  f"{float(margin.median()):.2f} nat, min {float(margin.min()):.2f}; {P.fmt(m)}")
            assert distinct >= 8 and float(margin.median()) >= 0.5
            assert got.dtype == torch.int64 and bool((got[safe] == want[safe]).all())
            assert float(safe.float().mean()) >= 0.9, (mode, h, float(safe.float().mean()))
            assert float(agree.float().mean()) >= 0.9, (mode, h, float(agree.float().mean()))
            # the log-probs of the one head family whose logits are driven by the latent with an O(1) dynamic range are GATED,
            # not only printed.  Bounds = 1.5 x the round-4 measurement (fp16 lp_l2 2.4-2.8e-4, lp_max 3.2-3.3e-3, c_l2 4.1-4.5e-4; bf16 2.0-2.3e-3,
            # 1.9-2.0e-2, 3.4-3.7e-3): fp16 meets the north star's 1e-3 in relative L2 with 3.5x margin here, its max-norm does not (3e-3) -- head
            # weights amplify the latent's error, "within 1e-3" is head-weight dependent.
            gate = {"fp16": dict(lp_l2=5e-4, lp_max=5e-3, c_l2=7e-4), "bf16": dict(lp_l2=3.5e-3, lp_max=3e-2, c_l2=5.5e-3)}[mode]
            for k_, bound in gate.items():
                assert m[k_] < bound, (mode, h, k_, m[k_], bound)


def _idm(precision="bf16", heads="uniform"):
    kw = O.idm_kwargs_for("4x")
    cfg = O.idm_config_from_kwargs(kw, dict(temperature=2.0))
    sd = O.idm_synthetic_state_dict(cfg, seed=0, heads=heads)
    pol = InverseActionPolicy(idm_action_space(), pi_head_kwargs=dict(temperature=2.0), idm_net_kwargs=kw, precision=precision)
    missing, unexpected = pol.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    return pol.to(DEV), cfg, sd


# log-probs of 2- / 11-way softmaxes are O(1): absolute bounds on them, relative on the centred logits
IDM_BOUNDS = {"bf16": dict(max_abs=3e-2, l2=1.5e-2), "fp16": dict(max_abs=4e-3, l2=2e-3)}


def test_idm_4x_prior_dominated_actions_on_peaked_heads():
    """config 3 with "peaked" heads (the prior / bias path, as test_prior_dominated_actions_at_config2_size_on_peaked_heads): predict() over
    the full 128-frame window must return the oracle's integer actions at >= 95 % of the 128 x (20 + 2) softmax groups outside the noise
    band, in both operand formats -- and equal them there.  The distinct-action count per softmax group is printed (1-2: the bias decides)."""
    _threads()
    pol, cfg, sd = _idm(heads="peaked")
    t = 128
    img = P.structured_frames(1, t, torch.Generator().manual_seed(23))
    ref = O.idm_forward(sd, cfg, img)
    for mode in ("bf16", "fp16"):
        pol.set_precision(mode)
        ac, _, res = pol.predict({"img": img.to(DEV)}, first=None, state_in=pol.initial_state(1), deterministic=True)
        torch.cuda.synchronize()
        for head in ("buttons", "camera"):
            hm = P.head_metrics(res["pd"][head].cpu(), ref[head])
            am = ref[head].argmax(-1).reshape(t, -1)
            print(f"ACTIONS[{mode}] 4x IDM T={t} peaked {head}: distinct oracle actions per softmax group (max over groups) {max(len(set(am[:, j].tolist())) for j in range(am.shape[1]))}; {P.fmt(hm)}")
            assert hm["argmax_safe_mismatch"] == 0 and hm["argmax_safe_frac"] >= 0.95 and hm["argmax_agree"] >= 0.95
            assert torch.equal(ac[head].cpu(), ref[head].argmax(-1)) or hm["argmax_agree"] < 1.0


@pytest.mark.parametrize("t", [16, 128])
def test_idm_4x_forward(t):
    """BASELINE.json config 3 (hid 4096, 32 heads, channels 256/512/512, 2 unmasked layers): a 16-frame window and the
    full 128-frame window run_inverse_dynamics_model.py feeds; predicted actions must equal the oracle's outside the noise band."""
    _threads()
    pol, cfg, sd = _idm()
    g = torch.Generator().manual_seed(22)
    img = torch.randint(0, 256, (1, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    ref = O.idm_forward(sd, cfg, img)
    for mode in ("bf16", "fp16"):
        pol.set_precision(mode)
        ac, _, res = pol.predict({"img": img.to(DEV)}, first=None, state_in=pol.initial_state(1), deterministic=True)
        torch.cuda.synchronize()
        for head in ("buttons", "camera"):
            got, want = res["pd"][head].cpu(), ref[head]
            hm = P.head_metrics(got, want)
            print(f"PARITY[{mode}] 4x IDM T={t} {head}: {P.fmt(hm)}")
            assert hm["max_abs_err"] < IDM_BOUNDS[mode]["max_abs"] and hm["lp_l2"] < IDM_BOUNDS[mode]["l2"]
            assert hm["argmax_safe_mismatch"] == 0                        # exact actions outside the noise band ...
            assert torch.equal(ac[head].cpu(), got.argmax(-1))            # ... and predict() returns the argmax of its own pd
            if mode == "fp16":
                assert hm["argmax_safe_frac"] > 0.5 and hm["argmax_agree"] > 0.97


def test_config5_bc_3x_two_chunks_with_kv_carry():
    """BASELINE.json configs[4]: the 3x model trained on T = 256 chunks with the KV memory carried (detached,
    behavioural_cloning.py:111) from one chunk to the next.  Loss and gradients of BOTH chunks against the oracle's autograd
    with ITS carried state; the state handed over must match too."""
    _threads()
    pol, cfg, sd = _policy("3x")
    g = torch.Generator().manual_seed(24)
    b, t = 1, 256
    so = O.initial_state(cfg, b)
    sgs = {mode: None for mode in ("bf16", "fp16")}
    for chunk in range(2):
        img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
        first = torch.zeros(b, t, dtype=torch.bool)
        ab, ac = torch.randint(0, 8641, (b, t), generator=g), torch.randint(0, 121, (b, t), generator=g)
        loss_ref, grads_ref, so = O.bc_loss_and_grads(sd, cfg, img, first, so, ab, ac)
        for mode in sgs:                                   # both operand formats against the same oracle gradients
            pol.set_precision(mode)
            tr = BCTrainer(pol, train_cnn=True, optimizer_state=False)
            sg = sgs[mode] if sgs[mode] is not None else pol.initial_state(b)
            loss, grads, sg = tr.loss_and_grads(img.to(DEV), first.to(DEV), sg, ab.to(DEV), ac.to(DEV))
            torch.cuda.synchronize()
            sgs[mode] = sg
            assert abs(float(loss) - loss_ref) < 2e-2, (mode, chunk, float(loss), loss_ref)
            worst, cos_all = 1.0, []
            for name in tr.trainable:
                ref = grads_ref[name]
                if float(ref.norm()) == 0.0:
                    continue
                mine = grads[name].cpu().reshape(ref.shape)
                assert torch.isfinite(mine).all(), name
                cos = float((mine * ref).sum() / (mine.norm() * ref.norm()))
                worst = min(worst, cos)
                cos_all.append(cos)
                assert cos > P.GRAD_BOUNDS[mode]["cos_min"], (mode, chunk, name, cos)
                assert 0.7 < float(mine.norm() / ref.norm()) < 1.4, (mode, chunk, name)      # (also: the fp16 loss scale was taken out again)
            assert sum(cos_all) / len(cos_all) > P.GRAD_BOUNDS[mode]["cos_mean"]
            for (m1, (k1, v1)), (m2, (k2, v2)) in zip(sg, so):
                assert torch.equal(m1.cpu(), m2) and not k1.requires_grad
                assert _l2(k1, k2) < P.BOUNDS[mode]["kv_l2"] and _l2(v1, v2) < P.BOUNDS[mode]["kv_l2"]
            print(f"PARITY[{mode}] config 5 (3x BC, T=256) chunk {chunk}: loss {float(loss):.4f} vs {loss_ref:.4f}, gradient cosine vs the fp32 oracle: "
                  f"mean {sum(cos_all) / len(cos_all):.4f}, worst {worst:.3f}")
            del grads, tr
        del grads_ref


def test_bc_step_3x():
    """Config 5's model: one 3x BC step (all parameters) -- loss and the gradient direction of a few tensors vs the fp32 oracle."""
    _threads()
    pk = O.policy_kwargs_for("3x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision="bf16")
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(DEV)
    b, t = 2, 4
    g = torch.Generator().manual_seed(23)
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(b, t, dtype=torch.bool)
    ab, ac = torch.randint(0, 8641, (b, t), generator=g), torch.randint(0, 121, (b, t), generator=g)
    loss_ref, grads_ref, _ = O.bc_loss_and_grads(sd, cfg, img, first, O.initial_state(cfg, b), ab, ac)
    for mode in ("bf16", "fp16"):
        pol.set_precision(mode)
        tr = BCTrainer(pol, train_cnn=True)
        loss, grads, _ = tr.loss_and_grads(img.to(DEV), first.to(DEV), pol.initial_state(b), ab.to(DEV), ac.to(DEV))
        torch.cuda.synchronize()
        assert abs(float(loss) - loss_ref) < 2e-2
        worst, cos_all = 1.0, []
        for name in tr.trainable:
            ref = grads_ref[name]
            if float(ref.norm()) == 0.0:
                continue
            mine = grads[name].cpu().reshape(ref.shape)
            assert torch.isfinite(mine).all(), name
            cos = float((mine * ref).sum() / (mine.norm() * ref.norm()))
            worst = min(worst, cos)
            cos_all.append(cos)
            assert cos > P.GRAD_BOUNDS[mode]["cos_min"], (mode, name, cos)     # per-format bound: tests/parity.py
        assert sum(cos_all) / len(cos_all) > P.GRAD_BOUNDS[mode]["cos_mean"]
        print(f"PARITY[{mode}] 3x BC gradients: cosine vs the fp32 oracle mean {sum(cos_all) / len(cos_all):.4f}, worst {worst:.3f}")
        # and one optimiser step (fp16: loss-scaled, overflow check on the device)
        l0, _ = tr.step(img.to(DEV), first.to(DEV), pol.initial_state(b), ab.to(DEV), ac.to(DEV))
        assert abs(l0 - loss_ref) < 2e-2 and tr.step_count == 1 and tr.skipped_steps == 0
        pol.load_state_dict(sd, strict=False)


def test_pre_lstm_ln_option_forward_and_bc():
    """use_pre_lstm_ln=True -- the reference's constructor DEFAULT (lib/policy.py:186-188,202-203), switched off by every released
    model file: forward (a chunk, then a T = 1 acting step on the carried state) and BC gradients against the oracle."""
    _threads()
    pk = dict(O.policy_kwargs_for("1x"))
    pk["use_pre_lstm_ln"] = True
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    assert cfg["use_pre_lstm_ln"]
    sd = O.synthetic_state_dict(cfg, seed=0)
    assert "net.pre_lstm_ln.weight" in sd
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision="bf16")
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(DEV)
    b = 2
    g = torch.Generator().manual_seed(11)
    so, sg = O.initial_state(cfg, b), pol.initial_state(b)
    for t in (5, 1):
        img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
        first = torch.zeros(b, t, dtype=torch.bool)
        ref = O.policy_forward(sd, cfg, img, first, so)
        so = ref["state_out"]
        (pd, vpred, _), sg = pol({"img": img.to(DEV)}, first.to(DEV), sg)
        torch.cuda.synchronize()
        P.check(P.policy_metrics(dict(buttons=pd["buttons"], camera=pd["camera"], vpred=vpred), ref), "bf16", f"pre_lstm_ln t={t}", model="1x")
    # the option must matter: without the extra LayerNorm the same weights give different log-probs
    cfg_off = dict(cfg); cfg_off["use_pre_lstm_ln"] = False
    off = O.policy_forward(sd, cfg_off, img, first, O.initial_state(cfg, b))
    assert float((off["buttons"] - ref["buttons"]).abs().max()) > 1e-3
    # BC gradients (trunk + heads), incl. the new LayerNorm's gain / bias
    t = 4
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(b, t, dtype=torch.bool)
    ab = torch.randint(0, 8641, (b, t), generator=g)
    ac = torch.randint(0, 121, (b, t), generator=g)
    with torch.enable_grad():
        loss_ref, grads_ref = O.bc_loss_and_grads(sd, cfg, img, first, O.initial_state(cfg, b), ab, ac)[:2]
    tr = BCTrainer(pol, train_cnn=False)
    loss, grads, _ = tr.loss_and_grads(img.to(DEV), first.to(DEV), pol.initial_state(b), ab.to(DEV), ac.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 2e-2
    assert "net.pre_lstm_ln.weight" in grads and "net.pre_lstm_ln.bias" in grads
    for name in ("net.pre_lstm_ln.weight", "net.pre_lstm_ln.bias", "net.img_process.linear.layer.weight", "net.lastlayer.layer.weight"):
        mine, ref_g = grads[name].cpu().reshape(grads_ref[name].shape), grads_ref[name]
        cos = float((mine * ref_g).sum() / (mine.norm() * ref_g.norm()))
        assert cos > 0.9, (name, cos)
        assert 0.7 < float(mine.norm() / ref_g.norm()) < 1.4, name
