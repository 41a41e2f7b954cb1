"""bench.py -- VPT policy forward throughput on MI355X (BASELINE.json metric: frames/sec, 2x policy,
128x128x3 frames, seq = 128).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (MinecraftAgentPolicy.forward through libvpt_hip.so) over one batch
of B x T = 64 x 128 synthetic uint8 frames that are already resident in HBM, with the KV memory carried
from the previous step (BASELINE.json configs[1]: foundation-model-2x forward on 1 MI355X).  Forward has no
exchange step, so N > 1 runs N data-parallel replicas on disjoint batches (weak scaling, no collective on
the data path); rank 0 prints ONE JSON line with the whole-job frames/s.

Extra objects on the line:
  roofline     : dominant kernel (vpt_conv3x3_kernel, 72 % of the forward FLOPs): algorithmic FLOPs of all
                 its launches in one step / their summed HIP-event durations, vs the 2.5 PFLOP/s dense bf16
                 MFMA peak (MI355X_MICROARCH.md).  `e2e_frac` = frames/s x 15.1016 GFLOP / peak.
  cpu_baseline : the UNMODIFIED reference (packaged by oracle/make_ref.py into the git-ignored oracle/_ref/, which travels
                 to the GPU box) timed on 8 host cores on a bounded sample (B=1, T=16) of the same workload, plus
                 BASELINE.json configs[0] (1x); kind "port" (the oracle) only if the archive is missing.
  parity       : log-prob / centred-logit / value errors of the bf16 default and of the fp16 parity mode vs the oracle.
  fp16_mode    : the same workload with precision="fp16" (the parity mode) as a full record: timed steps, roofline, per-kernel table.
  bc_step[_fp16]: BC step (forward + backward + Adam [+ all-reduce]) time, its fraction of the MFMA peak, the three conv passes' TF/s;
                 N > 1: `allreduce_detail` (the gradient exchange alone: ms, bytes, collectives, backend, ranks).
  parity.timed_batch / parity.competitive_heads / parity_mode: parity measured ON two sequences of the timed batch (both formats), on the
                 input-driven head family, and which throughput has earned the 1e-3 gate (fp16).
  ingest       : pinned host frames -> device through a two-deep copy pipeline under the forward (128 x 128 frames; 640 x 360 BGR through
                 vpt_clip_frames): frames/s, ms per step, overlap fraction, the PCIe crossover.  Never the headline value.

`--gpus N` is honoured by the script itself: with N > 1 and no launcher environment it starts its own N ranks (torch.distributed.run);
under a launcher it refuses a WORLD_SIZE that differs from N, and it refuses N > visible GPUs (RCCL: one device per rank) unless
VPT_DIST_BACKEND=gloo (tests).  The N > 1 line carries `collective_ranks` from an actual all-reduce of device tensors.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_FRAME = {"1x": 3.8235e9, "2x": 15.1016e9, "3x": 33.8409e9}  # BASELINE.md §3 (t = 128; 3x at t = 256)
MFMA_BF16_PEAK = 2.5e15


class BoxSampler:
    """Clock and power of the box WHILE a timed region runs (a thread polling torch.cuda.clock_rate / power_draw = amdsmi every 100 ms), so that two
    bench lines can be compared: the round-5 boxes differed by +-2.5 % and nothing on the line said which one ran at what clock.  Host-side
    driver queries only: nothing is enqueued on the GPU."""

    def __init__(self, dev_index, period=0.1):
        import threading
        self.dev, self.period, self.clk, self.pw, self.err = dev_index, period, [], [], None
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            try:
                self.clk.append(float(torch.cuda.clock_rate(self.dev)))
                w = float(torch.cuda.power_draw(self.dev))
                self.pw.append(w / 1e3 if w > 5000.0 else w)      # (documented as mW; the amdsmi path of this ROCm build returns W)
            except Exception as e:       # no amdsmi on this box: say so once and stop polling
                self.err = f"{type(e).__name__}: {e}"[:120]
                return
            self._stop.wait(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join(timeout=2.0)
        return False

    def record(self):
        if not self.clk:
            return dict(error=self.err or "no samples")
        c, w = sorted(self.clk), sorted(self.pw)
        rec = dict(sclk_mhz_sustained=round(c[len(c) // 2]), sclk_mhz_min=round(c[0]), sclk_mhz_max=round(c[-1]), power_w_avg=round(sum(w) / len(w)),
                   power_w_max=round(w[-1]), samples=len(c), source="torch.cuda.clock_rate / power_draw (amdsmi), polled every 100 ms during the timed steps")
        rec["power_cap_w"] = _power_cap_w(self.dev)
        return rec


def _power_cap_w(dev_index):
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[dev_index]
        info = amdsmi.amdsmi_get_power_cap_info(h)
        cap = float(info.get("power_cap", 0))
        return round(cap / 1e6) if cap > 1e5 else round(cap)        # (microwatts in amdsmi >= 6, watts before)
    except Exception:
        return None


def _host_threads(limit):
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, limit))


def _median_time(fn, budget_s, max_reps=5):
    fn()  # warm-up
    times, t_start = [], time.time()
    while len(times) < max_reps and (not times or time.time() - t_start < budget_s):
        t0 = time.time()
        fn()
        times.append(time.time() - t0)
    times.sort()
    return times[len(times) // 2], len(times)


def cpu_baseline(model: str):
    """the UNMODIFIED reference (oracle/_ref/vpt_reference.zip, packaged by oracle/make_ref.py) on the host
    cores, fp32, torch.set_num_threads(8), B=1 T=16 of the same synthetic frames, median of 5 after 1 warm-up -- for the
    benchmarked model AND for BASELINE.json configs[0] (1x).  Falls back to the oracle port when the archive is absent."""
    from oracle import vpt_oracle as O
    from oracle import make_ref
    cores = _host_threads(8)
    torch.set_num_threads(cores)
    t = 16
    g = torch.Generator().manual_seed(1)
    img = torch.randint(0, 256, (1, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(1, t, dtype=torch.bool)
    ref_zip = make_ref.reference_path()
    rates, kind = {}, "port"
    if ref_zip is not None:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_stubs"))
            sys.path.insert(0, ref_zip)
            import lib.torch_util as tu          # the reference's own modules (zipimport)
            tu.set_default_torch_device("cpu")
            from gym3.types import DictType
            from lib.action_mapping import CameraHierarchicalMapping
            from lib.policy import MinecraftAgentPolicy as RefPolicy
            space = DictType(**CameraHierarchicalMapping(n_camera_bins=11).get_action_space_update())
            for name in dict.fromkeys([model, "1x"]):
                cfg = O.config_from_policy_kwargs(O.policy_kwargs_for(name), dict(temperature=2.0))
                pol = RefPolicy(space, O.policy_kwargs_for(name), dict(temperature=2.0))
                pol.load_state_dict(O.synthetic_state_dict(cfg, seed=0), strict=False)
                pol.eval()
                st = pol.initial_state(1)
                with torch.no_grad():
                    med, n = _median_time(lambda: pol({"img": img}, first, st), 12.0)
                rates[name] = (t / med, n)
                del pol
            kind = "reference"
        except Exception as e:     # a broken archive must not take the bench line down
            rates, kind = {}, "port"
            sys.stderr.write(f"cpu_baseline: reference archive unusable ({type(e).__name__}: {e}); timing the oracle port\n")
    if not rates:
        for name in dict.fromkeys([model, "1x"]):
            cfg = O.config_from_policy_kwargs(O.policy_kwargs_for(name), dict(temperature=2.0))
            sd = O.synthetic_state_dict(cfg, seed=0)
            st = O.initial_state(cfg, 1)
            med, n = _median_time(lambda: O.policy_forward(sd, cfg, img, first, st), 12.0)
            rates[name] = (t / med, n)
    bc = None
    if kind == "reference":
        try:
            bc = _cpu_bc_step(RefPolicy, space, O, model, cores)
        except Exception as e:
            bc = dict(error=f"{type(e).__name__}: {e}")
    what = "the unmodified reference MinecraftAgentPolicy.forward (oracle/_ref/vpt_reference.zip)" if kind == "reference" else "oracle/vpt_oracle.py (fp32 port)"
    out = dict(value=round(rates[model][0], 2), unit="frames/s", cores=cores, kind=kind,
               sample=f"{what}, fp32, {model} model, B=1 T={t} synthetic frames, median of {rates[model][1]} after 1 warm-up, torch.set_num_threads({cores})")
    if "1x" in rates and model != "1x":
        out["config1_1x_frames_per_s"] = round(rates["1x"][0], 2)     # BASELINE.json configs[0]: 1x, B=1, T=16 on CPU
    if bc is not None:
        out["bc_step"] = bc
    return out


def _cpu_bc_step(RefPolicy, space, O, model, cores):
    """The BC half of the metric on the host: the reference's OWN training loop (behavioural_cloning.py:86-122) -- BATCH_SIZE = 8
    samples, each one frame (B = 1, T = 1) through get_output_for_observation -> get_logprob_of_action -> (-log_prob / 8).backward()
    with the hidden state carried detached, then th.optim.Adam(lr 1.81e-4, weight_decay 0.039428).step() -- on the unmodified
    reference policy, fp32, `cores` threads.  One warm-up step, then 2 timed steps (bounded: ~10-20 s for the 2x model)."""
    import numpy as np
    import torch as th
    from lib.tree_util import tree_map
    cfg = O.config_from_policy_kwargs(O.policy_kwargs_for(model), dict(temperature=2.0))
    policy = RefPolicy(space, O.policy_kwargs_for(model), dict(temperature=2.0))
    policy.load_state_dict(O.synthetic_state_dict(cfg, seed=0), strict=False)
    optimizer = th.optim.Adam(policy.parameters(), lr=0.000181, weight_decay=0.039428)
    g = th.Generator().manual_seed(2)
    n = 8
    frames = th.randint(0, 256, (n, 1, 128, 128, 3), generator=g, dtype=th.uint8)
    ab, ac = th.randint(0, 8641, (n, 1, 1), generator=g), th.randint(0, 121, (n, 1, 1), generator=g)
    dummy_first = th.from_numpy(np.array((False,)))
    state = policy.initial_state(1)

    def one_step(state):
        for i in range(n):
            pi_distribution, _, new_state = policy.get_output_for_observation({"img": frames[i]}, state, dummy_first)
            log_prob = policy.get_logprob_of_action(pi_distribution, {"buttons": ab[i], "camera": ac[i]})
            state = tree_map(lambda x: x.detach(), new_state)
            (-log_prob / n).backward()
        optimizer.step()
        optimizer.zero_grad()
        return state

    state = one_step(state)
    times = []
    for _ in range(2):
        t0 = time.time()
        state = one_step(state)
        times.append(time.time() - t0)
    sec = min(times)
    return dict(s_per_step=round(sec, 3), frames_per_step=n, s_per_frame=round(sec / n, 4), frames_per_s=round(n / sec, 2), cores=cores, kind="reference",
                sample=f"the reference's own BC loop (behavioural_cloning.py:86-122: 8 x (B=1, T=1) forward + backward, state carried, then Adam), {model} model, "
                       f"fp32, best of 2 steps after 1 warm-up, torch.set_num_threads({cores})")


def parity_block(model: str, dev):
    """Parity of what was just benchmarked, on a bounded sample (B=1, T=8 of the same model and weights family) against the
    CPU oracle: the bf16 default and the fp16 parity mode, metrics of tests/parity.py."""
    from oracle import vpt_oracle as O
    from tests import parity as P
    from vpt_amd.lib.policy import MinecraftAgentPolicy
    from vpt_amd.lib.types import minecraft_action_space
    torch.set_num_threads(_host_threads(32))
    pk = O.policy_kwargs_for(model)
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (1, 8, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(1, 8, dtype=torch.bool)
    ref = O.policy_forward(sd, cfg, img, first, O.initial_state(cfg, 1))
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0))
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(dev)
    out = {"sample": f"{model} model, B=1 T=8, synthetic weights seed 0, vs oracle/vpt_oracle.py (fp32, pinned to the live reference's golden vectors)"}
    for mode in ("bf16", "fp16"):
        pol.set_precision(mode)
        with torch.no_grad():
            (pd, vpred, _), _ = pol({"img": img.to(dev)}, first.to(dev), pol.initial_state(1))
        torch.cuda.synchronize()
        m = P.policy_metrics(dict(buttons=pd["buttons"], camera=pd["camera"], vpred=vpred), ref)
        out[mode] = {"logprob_rel_l2": round(max(m["buttons.lp_l2"], m["camera.lp_l2"]), 6), "logprob_max_rel": round(max(m["buttons.lp_max"], m["camera.lp_max"]), 6),
                     "centred_logits_rel_l2": round(max(m["buttons.c_l2"], m["camera.c_l2"]), 5), "value_rel": round(m["v_rel"], 5),
                     "argmax_mismatch_outside_noise_band": m["buttons.argmax_safe_mismatch"] + m["camera.argmax_safe_mismatch"],
                     "within_bounds": all(m[f"{h}.{k}"] < P.BOUNDS[mode][k] for h in ("buttons", "camera") for k in ("lp_l2", "lp_max", "c_l2", "c_max"))}
    # trained-policy-like ("peaked") heads on frames with low-frequency content: integer actions against the oracle's arg-max
    sdp = O.synthetic_state_dict(cfg, seed=0, heads="peaked")
    imgp = P.structured_frames(2, 16, torch.Generator().manual_seed(4))
    firstp = torch.zeros(2, 16, dtype=torch.bool)
    refp = O.policy_forward(sdp, cfg, imgp, firstp, O.initial_state(cfg, 2))
    pol.load_state_dict(sdp, strict=False)
    out["actions_peaked_heads"] = {"sample": f"{model} model, B=2 T=16 structured frames, oracle.peak_heads weights; deterministic actions of the fused head kernel vs the oracle's arg-max"}
    for mode in ("bf16", "fp16"):
        pol.set_precision(mode)
        pol._ensure_packed()
        with torch.no_grad():
            o = pol._engine.forward(imgp.to(dev), firstp.to(dev), pol.initial_state(2), sample="deterministic")
        torch.cuda.synchronize()
        rec = {}
        for h in ("buttons", "camera"):
            hm = P.head_metrics(o[h], refp[h])
            eq = float((o["action"][h][:, :, 0].cpu() == refp[h].argmax(-1)[:, :, 0]).float().mean())
            rec[h] = dict(equal_frac=round(eq, 4), outside_noise_band_frac=round(hm["argmax_safe_frac"], 4), mismatches_outside_band=hm["argmax_safe_mismatch"],
                          logprob_rel_l2=round(hm["lp_l2"], 6))
        out["actions_peaked_heads"][mode] = rec
    return out


def _roofline(ops, pol, step, state, args, B, T, mode):
    """One extra, instrumented step OUTSIDE the timed region on a single CNN stream (per-kernel durations are only meaningful
    without cross-stream overlap): HIP events around every launch, on the stream the kernels run on (torch's current stream)."""
    streams_saved = pol._engine.cnn_streams
    pol._engine.cnn_streams = 1
    ops.TIMER.enabled = True
    ops.TIMER.reset()
    step(state)
    summ = ops.TIMER.summary()
    ops.TIMER.enabled = False
    pol._engine.cnn_streams = streams_saved
    # every launch of the roofline kernel vpt_conv3x3_kernel: its plain / residual modes ("vpt_conv3x3_forward") AND its pool-fused mode
    # ("vpt_conv3x3_pool_forward": the stacks' firstconv with the max-pool in the epilogue; the seam kernel behind it is a different kernel)
    c = None
    for lab in ("vpt_conv3x3_forward", "vpt_conv3x3_pool_forward"):
        if lab in summ:
            c = dict(summ[lab]) if c is None else {k: c[k] + summ[lab][k] for k in c}
    total_ms = sum(v["ms"] for v in summ.values())
    kernels = {k: dict(ms=round(v["ms"], 3), calls=v["calls"],
                       tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] and v["ms"] else None)
               for k, v in summ.items()}
    roof = None
    if c and c["ms"] > 0:
        ach = c["flops"] / (c["ms"] * 1e-3) / 1e12
        # HBM traffic cannot be counted live (PMC needs rocprofv3): the committed PMC measurement of this same workload
        # (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2 per the gfx950 correction), per launch, labelled as such
        traffic, tsrc, talg = None, None, 152.4e9 / c["calls"]      # 18.6 MB / frame x 8192 (tools/profile_summarize.py has the derivation)
        for tp in TRAFFIC_FILES.get(mode, ()):
            tpath = os.path.join(ROOT, "profiles", tp)
            if args.model == "2x" and B * T == 8192 and os.path.exists(tpath):
                try:      # per-step totals of the committed measurement over THIS run's launch count (sub-chunking changes the count, not the bytes)
                    tj = json.load(open(tpath))
                    per_step = tj.get("hbm_bytes_per_step", tj["hbm_bytes_per_launch"] * 112.0)
                    alg_step = tj.get("algorithmic_bytes_per_step", 152.4e9)
                    traffic, talg = round(per_step / c["calls"]), alg_step / c["calls"]
                    tsrc = f"committed PMC (profiles/{tp}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, tools/profile_round.sh: {per_step / 1e9:.1f} GB per step / {c['calls']} launches), not measured by this run"
                    break
                except Exception:
                    traffic = None
        roof = dict(bound="mfma", kernel="vpt_conv3x3_kernel", achieved=round(ach, 1), peak=2500.0, unit="TFLOP/s",
                    frac=round(ach / 2500.0, 4), traffic=traffic, traffic_unit=f"HBM bytes per launch (algorithmic {talg:.3g})", traffic_source=tsrc,
                    launches=c["calls"], avg_launch_ms=round(c["ms"] / c["calls"], 4), share_of_step_time=round(c["ms"] / total_ms, 3),
                    by_mode={lab: dict(ms=round(summ[lab]["ms"], 3), calls=summ[lab]["calls"], tflops=round(summ[lab]["flops"] / (summ[lab]["ms"] * 1e-3) / 1e12, 1))
                             for lab in ("vpt_conv3x3_forward", "vpt_conv3x3_pool_forward") if lab in summ and summ[lab]["ms"] > 0},
                    flop_accounting="direct-convolution FLOPs (2 x H x W x Cout x 9 x Cin per frame and layer) / summed HIP-event durations of the launches")
    return roof, kernels


def headline_batch_parity(pol, model, img, first, dev, modes, rows=(0, -1)):
    """parity measured ON the benchmarked batch.  Two sequences (the first and the last) of the TIMED 64 x 128 batch, with the
    benchmarked weights, go through the CPU oracle from initial_state (what the first timed step computed for them); the GPU side is the whole-batch
    forward of each operand format -- the rows are read out of the 64-sequence result, so batch effects would show."""
    from oracle import vpt_oracle as O
    from tests import parity as P
    torch.set_num_threads(_host_threads(32))
    B, T = img.shape[:2]
    rows = sorted({r % B for r in rows})
    cfg = O.config_from_policy_kwargs(O.policy_kwargs_for(model), dict(temperature=2.0))
    sd = {k: v.detach().float().cpu() for k, v in pol.state_dict().items()}
    t0 = time.time()
    ref = O.policy_forward(sd, cfg, img[rows].cpu(), first[rows].cpu(), O.initial_state(cfg, len(rows)))
    out = {"sample": f"sequences {rows} of the timed {B} x {T} batch (benchmarked weights, initial_state), whole-batch GPU forward vs oracle/vpt_oracle.py on those rows "
                     f"({time.time() - t0:.1f} s of CPU)"}
    for mode in modes:
        pol.set_precision(mode)
        with torch.no_grad():
            (pd, vpred, _), _ = pol({"img": img}, first, pol.initial_state(B))
        torch.cuda.synchronize()
        m = P.policy_metrics(dict(buttons=pd["buttons"][rows], camera=pd["camera"][rows], vpred=vpred[rows]), ref)
        out[mode] = {"logprob_rel_l2": round(max(m["buttons.lp_l2"], m["camera.lp_l2"]), 6), "logprob_max_rel": round(max(m["buttons.lp_max"], m["camera.lp_max"]), 6),
                     "centred_logits_rel_l2": round(max(m["buttons.c_l2"], m["camera.c_l2"]), 5), "value_rel": round(m["v_rel"], 5),
                     "argmax_mismatch_outside_noise_band": m["buttons.argmax_safe_mismatch"] + m["camera.argmax_safe_mismatch"],
                     "within_bounds": all(m[f"{h}.{k}"] < P.BOUNDS[mode][k] for h in ("buttons", "camera") for k in ("lp_l2", "lp_max", "c_l2", "c_max")),
                     "meets_1e-3_logprob_rel_l2": max(m["buttons.lp_l2"], m["camera.lp_l2"]) < 1e-3,
                     "meets_1e-3_logprob_max_rel": max(m["buttons.lp_max"], m["camera.lp_max"]) < 1e-3}
        out[mode]["meets_1e-3"] = out[mode]["meets_1e-3_logprob_rel_l2"] and out[mode]["meets_1e-3_logprob_max_rel"]      # relative L2 AND max-norm, both sequences
        del pd, vpred
    return out


def competitive_heads_parity(model, dev, modes):
    """the head family whose logits are DRIVEN by the latent with an O(1) dynamic range (oracle.fit_scene_heads: 16 live classes,
    margins of ~4 nat decided by the input) next to the near-uniform one above.  The log-prob error here is the latent's error amplified by the
    head weights -- "logits within 1e-3" depends on the head weights, and no trained .weights exist in this image."""
    from oracle import vpt_oracle as O
    from tests import parity as P
    from vpt_amd.lib.policy import MinecraftAgentPolicy
    from vpt_amd.lib.types import minecraft_action_space
    pk = O.policy_kwargs_for(model)
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd0 = O.synthetic_state_dict(cfg, seed=0)
    b, t, n_scenes = 2, 64, 8
    img, scene = O.scene_frames(b, t, n_scenes, torch.Generator().manual_seed(606))
    first = torch.zeros(b, t, dtype=torch.bool)
    trunk = O.policy_forward(sd0, cfg, img, first, O.initial_state(cfg, b))
    sd = O.fit_scene_heads(sd0, trunk["latent"].reshape(b * t, -1), scene.reshape(-1), 2.0)
    ref = {h: O.categorical_head(sd, f"pi_head.{h}.", trunk["latent"], 2.0).reshape(b, t, 1, -1) for h in ("buttons", "camera")}
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0))
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(dev)
    out = {"sample": f"{model} model, B={b} T={t} frames of {n_scenes} scenes, oracle.fit_scene_heads weights (input-driven, top-2 margin ~4 nat): the head weights "
                     "amplify the latent's error -- 'logits within 1e-3' is head-weight dependent; no trained .weights exist here"}
    for mode in modes:
        pol.set_precision(mode)
        pol._ensure_packed()
        with torch.no_grad():
            o = pol._engine.forward(img.to(dev), first.to(dev), pol.initial_state(b), sample="deterministic")
        torch.cuda.synchronize()
        rec = {}
        for h in ("buttons", "camera"):
            hm = P.head_metrics(o[h], ref[h])
            want = ref[h].argmax(-1)[:, :, 0]
            rec[h] = dict(logprob_rel_l2=round(hm["lp_l2"], 6), logprob_max_rel=round(hm["lp_max"], 6), centred_logits_rel_l2=round(hm["c_l2"], 6),
                          max_abs_err_nat=round(hm["max_abs_err"], 5), actions_equal_frac=round(float((o["action"][h][:, :, 0].cpu() == want).float().mean()), 4),
                          distinct_oracle_actions=len(set(want.flatten().tolist())))
        out[mode] = rec
    return out


def ingest_leg(pol, img, first, dev, copy_s=None, steps=5):
    """host -> device ingest, bounded, NOT the headline (whose frames are resident in HBM when the timed region starts).
    The reference uploads every observation from host memory (agent.py:147-148: th.from_numpy(...).to(device) per frame; data_loader.py:113-122
    hands out host frames).  Here: pinned host uint8 frames -> device through a two-deep, stream-ordered pipeline (copy stream + events; the
    copy of step k + 1 runs under the forward of step k), for
      (i)  agent-resolution frames [B, T, 128, 128, 3] (49 KB / frame), and
      (ii) decoded 640 x 360 BGR frames (691 KB / frame) in 2048-frame chunks through vpt_clip_frames (BGR -> RGB + cv2.INTER_LINEAR resize on
           the device, no cursor) into the batch the forward then consumes.
    Reported per case: frames/s and ms per step of the pipeline, of the forward alone and of the copies alone, the H2D rate, and
    overlap_frac = the share of the copy time hidden under compute."""
    from vpt_amd import ops
    B, T = img.shape[:2]
    n = B * T
    # the copy stream is created FIRST in the process (main()): HIP maps streams onto a few hardware queues in creation order, and a copy queued
    # behind a compute stream's kernels on the same hardware queue does not overlap with anything (measured: tools/ingest_probe.py, round 5)
    main = torch.cuda.current_stream()
    copy_s = copy_s if copy_s is not None else torch.cuda.Stream()
    out = {}

    def fwd(frames, st):
        with torch.no_grad():
            (_, _, _), st = pol({"img": frames}, first, st)
        return st

    def timeit(fn, prime=None):
        """prime: un-timed pipeline fill (the first step's frames are uploaded before the clock starts: the figure is the steady state, in which
        every timed step's upload ran under the previous step's forward)."""
        ctx = prime() if prime is not None else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(ctx) if prime is not None else fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    # ---- (i) 128 x 128 frames
    host = torch.empty(img.shape, dtype=torch.uint8).pin_memory()
    host.copy_(img)
    bufs = [torch.empty_like(img), torch.empty_like(img)]

    def copies_only():
        for k in range(steps):
            with torch.cuda.stream(copy_s):
                bufs[k % 2].copy_(host, non_blocking=True)

    def compute_only():
        st = pol.initial_state(B)
        for k in range(steps):
            st = fwd(bufs[k % 2], st)

    def prime1():
        with torch.cuda.stream(copy_s):
            bufs[0].copy_(host, non_blocking=True)
            ready = torch.cuda.Event(); ready.record(copy_s)
        return ready

    def pipelined(ready):
        st = pol.initial_state(B)
        done = {}
        for k in range(steps):
            main.wait_event(ready)
            with torch.cuda.stream(copy_s):      # (the last step uploads a batch nobody consumes: every timed step carries one upload)
                if k - 1 in done:
                    copy_s.wait_event(done[k - 1])          # the buffer about to be overwritten was read by step k - 1
                bufs[(k + 1) % 2].copy_(host, non_blocking=True)
                ready = torch.cuda.Event(); ready.record(copy_s)
            st = fwd(bufs[k % 2], st)
            done[k] = torch.cuda.Event(); done[k].record(main)

    copies_only(); compute_only()          # warm-up
    t_copy, t_comp, t_pipe = timeit(copies_only), timeit(compute_only), timeit(pipelined, prime1)
    nbytes = img.numel()
    out["frames_128x128"] = dict(bytes_per_step=nbytes, frames_per_s=round(n / t_pipe, 1), ms_per_step=round(1e3 * t_pipe, 3), forward_alone_ms=round(1e3 * t_comp, 3),
                                 copy_alone_ms=round(1e3 * t_copy, 3), h2d_gb_per_s=round(nbytes / t_copy / 1e9, 1),
                                 overlap_frac=round(1.0 - max(0.0, t_pipe - t_comp) / t_copy, 3), steps=steps)
    del host, bufs
    # ---- (ii) decoded 640 x 360 BGR frames through the device clip kernel
    H, W, chunk = 360, 640, 2048
    n_chunks = (n + chunk - 1) // chunk
    # device staging for TWO whole steps of raw chunks (2 x 5.7 GB of the 288 GB): the uploads of step k + 1 run under the whole forward of step k
    nbuf = 2 * n_chunks
    raw_dev = [torch.randint(0, 256, (chunk, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    host = torch.empty(chunk, H, W, 3, dtype=torch.uint8).pin_memory()
    host.copy_(raw_dev[0])
    batch = torch.empty(n, 128, 128, 3, dtype=torch.uint8, device=dev)

    def clip_chunk(c, src):
        lo, hi = c * chunk, min(n, (c + 1) * chunk)
        ops.clip_frames(src[:hi - lo], out_hw=(128, 128), out=batch[lo:hi])

    def copies_only2():
        for g in range(steps * n_chunks):
            with torch.cuda.stream(copy_s):
                raw_dev[g % nbuf].copy_(host, non_blocking=True)

    def compute_only2():
        st = pol.initial_state(B)
        for k in range(steps):
            for c in range(n_chunks):
                clip_chunk(c, raw_dev[c % nbuf])
            st = fwd(batch.view(B, T, 128, 128, 3), st)

    ready, used = {}, {}

    def issue(g):
        with torch.cuda.stream(copy_s):
            if g - nbuf in used:
                copy_s.wait_event(used[g - nbuf])          # the staging buffer about to be overwritten was consumed by chunk g - nbuf's clip launch
            raw_dev[g % nbuf].copy_(host, non_blocking=True)
            ready[g] = torch.cuda.Event(); ready[g].record(copy_s)

    def prime2():
        ready.clear(); used.clear()
        for g in range(n_chunks):            # the first step's chunks, before the clock starts
            issue(g)
        return None

    def pipelined2(_):
        st = pol.initial_state(B)
        total = steps * n_chunks
        for g in range(total):
            if g % n_chunks == 0:            # a step begins: queue the NEXT step's uploads (they run under this step's clip launches and forward)
                for g2 in range(g + n_chunks, g + 2 * n_chunks):
                    issue(g2)
            main.wait_event(ready[g])
            clip_chunk(g % n_chunks, raw_dev[g % nbuf])
            used[g] = torch.cuda.Event(); used[g].record(main)
            if g % n_chunks == n_chunks - 1:
                st = fwd(batch.view(B, T, 128, 128, 3), st)

    copies_only2(); compute_only2()
    t_copy, t_comp, t_pipe = timeit(copies_only2), timeit(compute_only2), timeit(pipelined2, prime2)
    raw_bytes = n * H * W * 3
    rate = raw_bytes / t_copy
    out["frames_640x360_bgr_via_vpt_clip_frames"] = dict(bytes_per_step=raw_bytes, frames_per_s=round(n / t_pipe, 1), ms_per_step=round(1e3 * t_pipe, 3),
                                                         clip_plus_forward_alone_ms=round(1e3 * t_comp, 3), copy_alone_ms=round(1e3 * t_copy, 3), h2d_gb_per_s=round(rate / 1e9, 1),
                                                         overlap_frac=round(1.0 - max(0.0, t_pipe - t_comp) / t_copy, 3), pcie_bound=bool(t_copy > t_comp),
                                                         crossover_frames_per_s=round(rate / (H * W * 3), 1), chunk_frames=chunk, steps=steps,
                                                         note="crossover = H2D rate / 691200 B: above that forward rate the raw-frame upload, not the GPU, bounds the loader")
    out["note"] = ("pinned host memory, hipMemcpyAsync on a copy stream created before every other stream, two device buffers (two whole steps of raw chunks in case ii), "
                   "events both ways; steady state: the first step's upload is outside the timed region, every timed step carries one upload; frames resident in HBM remain the headline's definition")
    return out


TRAFFIC_FILES = {"bf16": ("r06_bench_conv3x3_traffic.json", "r05_bench_conv3x3_traffic.json"), "fp16": ("r06_bench_conv3x3_traffic_fp16.json", "r05_bench_conv3x3_traffic_fp16.json")}


def _bc_leg(pol, args, img, first, g, dev, world, distributed, barrier, dist, mode):
    """Second half of BASELINE.json's metric: BC-step time (forward + backward through every layer + gradient all-reduce over
    RCCL when N > 1 + fused Adam), same batch per GPU, KV memory carried between steps; plus one instrumented step (outside the
    timed region) for the three convolution passes' TF/s."""
    from vpt_amd import ops
    from vpt_amd.training import BCTrainer
    B, T = img.shape[:2]
    pol.set_precision(mode)
    tr = BCTrainer(pol, train_cnn=True)
    ab = torch.randint(0, pol._engine.n_buttons, (B, T), generator=g).to(dev)
    ac = torch.randint(0, pol._engine.n_camera, (B, T), generator=g).to(dev)
    st_bc = pol.initial_state(B)
    losses = []
    for _ in range(args.bc_warmup):
        l, st_bc = tr.step(img, first, st_bc, ab, ac)
        losses.append(l)
    barrier()
    with BoxSampler(dev.index if dev.index is not None else 0) as box_bc:
        t0 = time.perf_counter()
        for _ in range(args.bc_steps):
            l, st_bc = tr.step(img, first, st_bc, ab, ac)
            losses.append(l)
        barrier()
        bc_el = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([bc_el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        bc_el = float(tt.item())
    fl = FLOP_PER_FRAME.get(args.model, 0)
    sec = bc_el / args.bc_steps
    bc = dict(ms_per_step=round(1e3 * sec, 2), frames_per_s=round(world * B * T / sec, 1),
              steps=args.bc_steps, warmup=args.bc_warmup, global_batch=world * B, seq_len=T, trained="all parameters (CNN + trunk + heads)",
              precision=mode, optimizer="Adam lr 1.81e-4 wd 0.039428 (behavioural_cloning.py:38-40)" + (f"; dynamic loss scale {tr.loss_scale:g}, {tr.skipped_steps} skipped steps" if tr.scaled else ""),
              allreduce=("bucketed all-reduce of the fp32 gradients, trunk + heads overlapped with the CNN backward" if distributed else "none (1 GPU)"),
              loss_first=round(losses[0], 4), loss_last=round(losses[-1], 4),
              tflops=round(3 * fl * B * T / sec / 1e12, 1), frac_of_mfma_peak=round(3 * fl * B * T / sec / MFMA_BF16_PEAK, 4),
              flop_accounting="3 x forward FLOPs (SURVEY 8d) x frames / step time, per GPU",
              peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 1), box=box_bc.record(),
              gradients="bit-reproducible: every cross-workgroup sum through a partial slab added in a fixed order (vpt_reduce.hip)")
    if distributed and tr._arenas is not None:
        # the gradient exchange by itself (nothing to overlap with): what the step would pay if none of it were hidden
        from vpt_amd import distributed as D
        arenas = tr._arenas[1]
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            D.bucketed_all_reduce_finish(arenas[0].all_reduce_start() + arenas[1].all_reduce_start())
            torch.cuda.synchronize()
        ar_ms = (time.perf_counter() - t0) / 3 * 1e3
        # ... and the same step WITHOUT the exchange (each rank on its own shard): step - local = the exposed part, the rest was hidden under the CNN backward
        tr.exchange = False
        tr.step(img, first, st_bc, ab, ac)
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            tr.step(img, first, st_bc, ab, ac)
        barrier()
        local_ms = (time.perf_counter() - t0) / 3 * 1e3
        tr.exchange = True
        exposed = max(0.0, 1e3 * sec - local_ms)
        bc["allreduce_detail"] = dict(ms_standalone=round(ar_ms, 3), step_ms_without_exchange=round(local_ms, 2), exposed_ms=round(exposed, 3),
                                      overlap_frac=round(max(0.0, min(1.0, 1.0 - exposed / ar_ms)), 3) if ar_ms > 0 else None,
                                      overlap_note="overlap_frac = 1 - (step - step without exchange) / ms_standalone: the share of the exchange hidden under the CNN backward "
                                                   "(DESIGN.md section 5 models 0.995 at 8 GPUs over xGMI); the replicas' weights diverge in the un-exchanged steps, which run last",
                                      bytes=int(4 * (arenas[0].flat.numel() + arenas[1].flat.numel())),
                                      collectives=len(arenas[0].buckets) + len(arenas[1].buckets), backend=dist.get_backend(), ranks=dist.get_world_size(),
                                      early_wave_bytes=int(4 * arenas[0].flat.numel()), note="early wave (trunk + heads) is launched before the CNN backward and overlaps it; "
                                      "the late wave (CNN) is exposed; ms_standalone = both waves back to back with nothing to hide behind")
    if not distributed and not getattr(args, "no_dp_probe", False):
        # The data-parallel path of the SAME step over RCCL in a one-rank group (BCTrainer.force_exchange): frame-count reduction, arenas, early exchange under
        # the CNN backward, late exchange, health reduction.  One rank measures what the plumbing costs on the real transport, not link bandwidth.
        try:
            import socket
            import torch.distributed as dist1
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            dist1.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            try:
                tr.force_exchange = True
                tr.step(img, first, st_bc, ab, ac)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    _, st_bc = tr.step(img, first, st_bc, ab, ac)
                torch.cuda.synchronize()
                dp_ms = (time.perf_counter() - t0) / 3 * 1e3
                bc["dp_path_one_rank"] = dict(ms_per_step=round(dp_ms, 2), extra_ms=round(dp_ms - 1e3 * sec, 2), backend=dist1.get_backend(), ranks=1,
                                              note="the data-parallel BC step (all collectives issued on RCCL, arenas reduced in place) in a one-rank group on this GPU")
            finally:
                tr.force_exchange = False
                dist1.destroy_process_group()
        except Exception as e:
            bc["dp_path_one_rank"] = dict(error=f"{type(e).__name__}: {e}"[:200])
    if int(os.environ.get("RANK", "0")) == 0 and not distributed:
        tr.cnn_streams = 1          # per-kernel durations are only meaningful without cross-stream overlap
        ops.TIMER.enabled = True
        ops.TIMER.reset()
        tr.step(img, first, st_bc, ab, ac)
        summ = ops.TIMER.summary()
        ops.TIMER.enabled = False
        tf = lambda k: round(summ[k]["flops"] / (summ[k]["ms"] * 1e-3) / 1e12, 1) if k in summ and summ[k]["ms"] > 0 else None
        bc["conv_passes_tflops"] = dict(forward=tf("vpt_conv3x3_forward"), dgrad=tf("vpt_conv3x3_dgrad"), wgrad=tf("vpt_conv3x3_wgrad"))
        bc["kernels_ms"] = {k: float(f"{v['ms']:.3g}") for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:16]}
    del tr
    return bc


def configs_block(dev):
    """The other 1-GPU configurations of BASELINE.json, timed by this run (bounded: ~15 s of GPU time):
       idm_4x_t128    -- configs[2]: the 4x inverse-dynamics model on one 128-frame window (non-causal attention, lib/policy.py:374-403),
                         ms per window and frames/s, + action parity against the oracle on a 16-frame window of the same model;
       bc_3x_b32_t256 -- configs[4]'s model and per-GPU shape: a 3x BC step (forward + backward + Adam) at B = 32, T = 256 on two consecutive
                         chunks with the detached KV memory carried (behavioural_cloning.py:86-123 generalised), ms per step, peak memory."""
    from vpt_amd import configs
    from vpt_amd.lib.policy import InverseActionPolicy, MinecraftAgentPolicy
    from vpt_amd.lib.types import idm_action_space, minecraft_action_space
    from vpt_amd.training import BCTrainer
    out = {}
    # ---- config 3
    try:
        kw = configs.idm_kwargs_for("4x")
        pol = InverseActionPolicy(idm_action_space(), pi_head_kwargs=dict(temperature=2.0), idm_net_kwargs=kw, precision="bf16")
        configs.randomize_(pol, 0)
        pol = pol.to(dev)
        g = torch.Generator().manual_seed(1)
        img = torch.randint(0, 256, (1, 128, 128, 128, 3), generator=g, dtype=torch.uint8).to(dev)
        rec = {}
        for mode in ("bf16", "fp16"):
            pol.set_precision(mode)
            with torch.no_grad():
                for _ in range(2):
                    pol.predict({"img": img}, first=None, state_in=pol.initial_state(1), deterministic=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    pol.predict({"img": img}, first=None, state_in=pol.initial_state(1), deterministic=True)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            rec[mode] = dict(ms_per_window=round(1e3 * dt, 3), frames_per_s=round(128 / dt, 1))
        try:       # parity on a bounded sample: 16-frame window, synthetic oracle weights, predicted actions vs the oracle's arg-max
            from oracle import vpt_oracle as O
            from tests import parity as P
            torch.set_num_threads(_host_threads(32))
            cfg = O.idm_config_from_kwargs(O.idm_kwargs_for("4x"), dict(temperature=2.0))
            sd = O.idm_synthetic_state_dict(cfg, seed=0)
            pol.load_state_dict(sd, strict=False)
            im16 = torch.randint(0, 256, (1, 16, 128, 128, 3), generator=torch.Generator().manual_seed(22), dtype=torch.uint8)
            ref = O.idm_forward(sd, cfg, im16)
            for mode in ("bf16", "fp16"):
                pol.set_precision(mode)
                with torch.no_grad():
                    ac, _, res = pol.predict({"img": im16.to(dev)}, first=None, state_in=pol.initial_state(1), deterministic=True)
                torch.cuda.synchronize()
                hm = {h: P.head_metrics(res["pd"][h].cpu(), ref[h]) for h in ("buttons", "camera")}
                rec[mode]["parity"] = dict(sample="4x IDM, T=16 window, vs oracle/vpt_oracle.py:idm_forward",
                                           logprob_rel_l2=round(max(v["lp_l2"] for v in hm.values()), 6), max_abs_err=round(max(v["max_abs_err"] for v in hm.values()), 5),
                                           action_agreement=round(min(v["argmax_agree"] for v in hm.values()), 4),
                                           mismatches_outside_noise_band=sum(v["argmax_safe_mismatch"] for v in hm.values()))
        except Exception as e:
            rec["parity_error"] = f"{type(e).__name__}: {e}"
        rec["workload"] = "4x_idm forward, B=1, seq=128 (BASELINE.json configs[2]), random-init weights, frames resident in HBM"
        out["idm_4x_t128"] = rec
        del pol, img
    except Exception as e:
        out["idm_4x_t128"] = dict(error=f"{type(e).__name__}: {e}")
    torch.cuda.empty_cache()
    # ---- config 5's model at its per-GPU shape
    try:
        torch.cuda.reset_peak_memory_stats()
        pol = MinecraftAgentPolicy(minecraft_action_space(), configs.policy_kwargs_for("3x"), dict(temperature=2.0), precision="bf16")
        configs.randomize_(pol, 0)
        pol = pol.to(dev)
        B, T = 32, 256
        g = torch.Generator().manual_seed(5)
        imgs = [torch.randint(0, 256, (B, T, 128, 128, 3), generator=g, dtype=torch.uint8).to(dev) for _ in range(2)]
        first = torch.zeros(B, T, dtype=torch.bool, device=dev)
        ab = torch.randint(0, pol._engine.n_buttons, (B, T), generator=g).to(dev)
        ac = torch.randint(0, pol._engine.n_camera, (B, T), generator=g).to(dev)
        tr = BCTrainer(pol, train_cnn=True)
        st = pol.initial_state(B)
        losses = []
        l, st = tr.step(imgs[0], first, st, ab, ac)          # warm-up (chunk 0)
        losses.append(l)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in (1, 0, 1):                                  # three timed steps, every one on the memory the previous chunk left
            l, st = tr.step(imgs[c], first, st, ab, ac)
            losses.append(l)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / 3
        out["bc_3x_b32_t256"] = dict(ms_per_step=round(1e3 * sec, 1), frames_per_s=round(B * T / sec, 1), steps=3, warmup=1, batch=B, seq_len=T,
                                     kv_carry="consecutive T=256 chunks, detached KV memory carried between steps", precision="bf16",
                                     tflops=round(3 * FLOP_PER_FRAME["3x"] * B * T / sec / 1e12, 1),
                                     frac_of_mfma_peak=round(3 * FLOP_PER_FRAME["3x"] * B * T / sec / MFMA_BF16_PEAK, 4),
                                     peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 1), loss_first=round(losses[0], 4), loss_last=round(losses[-1], 4),
                                     workload="foundation-model-3x BC step (forward + backward + Adam), B=32 seq=256 per GPU (BASELINE.json configs[4], single GPU: no all-reduce)")
        del tr, pol, imgs
    except Exception as e:
        out["bc_3x_b32_t256"] = dict(error=f"{type(e).__name__}: {e}")
    torch.cuda.empty_cache()
    return out


def _trim_kernels(kernels):
    """Per-kernel table with 3 significant figures (the line's bulk is these tables)."""
    if not isinstance(kernels, dict):
        return kernels
    out = {}
    for k, v in kernels.items():
        out[k] = {q: (float(f"{x:.3g}") if isinstance(x, float) else x) for q, x in v.items()} if isinstance(v, dict) else v
    return out


def _parity_status(parity, head):
    """One sentence a reader of the parsed line cannot miss: which operand format is inside the north star's 1e-3 and which is not."""
    try:
        tb = parity.get("timed_batch") or {}
        h, f = tb.get(head, {}), tb.get("fp16", {})
        hm = h.get("meets_1e-3")
        if head == "fp16":
            return (f"fp16 headline {'inside' if hm else 'OUTSIDE'} 1e-3 (timed batch: log-prob rel-L2 {h.get('logprob_rel_l2')}, max {h.get('logprob_max_rel')}, "
                    f"centred logits {h.get('centred_logits_rel_l2')})"), None
        return (f"{head} headline {'inside' if hm else 'OUTSIDE'} 1e-3 (rel-L2 {h.get('logprob_rel_l2')}, max {h.get('logprob_max_rel')}, centred "
                f"{h.get('centred_logits_rel_l2')}); fp16 {'inside' if f.get('meets_1e-3') else 'outside'} ({f.get('logprob_rel_l2')} / {f.get('logprob_max_rel')}): parity_mode"), None
    except Exception as e:
        return f"parity status unavailable: {type(e).__name__}", None


def launch_plan(gpus: int, env, n_dev: int, backend: str):
    """What `bench.py --gpus N` does, as a pure function of (N, launcher environment, visible devices, transport) -- no GPU needed to test it:
         ("refuse", message)  never an N-GPU line from fewer ranks or devices than N, never a silent 1-GPU line for N > 1;
         ("spawn", N)         N > 1 and no launcher environment: start N ranks ourselves (torch.distributed.run) and return their exit code;
         ("run", world)       this process is one of `world` == N ranks (or the single process of N = 1)."""
    if n_dev <= 0:
        return "refuse", "bench.py needs a GPU (the HIP path has no CPU fallback)"
    if gpus < 1:
        return "refuse", f"bench.py: --gpus must be >= 1, got {gpus}"
    if gpus > n_dev and backend != "gloo":
        # one rank per GPU over RCCL (VPT_DIST_BACKEND=gloo is the tests' way to run the N > 1 branch with every rank on one device; it claims no scaling figure)
        return "refuse", (f"bench.py: --gpus {gpus} but only {n_dev} GPU(s) visible (RCCL needs one device per rank; VPT_DIST_BACKEND=gloo runs the branch "
                          "on one device for tests)")
    if "WORLD_SIZE" not in env:
        return ("spawn", gpus) if gpus > 1 else ("run", 1)
    world = int(env["WORLD_SIZE"])
    if world != gpus:
        return "refuse", f"bench.py: --gpus {gpus} does not match the launcher's WORLD_SIZE={world} (the line's n_gpus is the number of ranks that ran)"
    return "run", world


_LINE_OUT = None


def _claim_stdout():
    """stdout carries the JSON line and NOTHING else.  Libraries that write to descriptor 1 themselves -- RCCL prints a five-line version banner through C stdio, which a
    pipe delivers at process exit, i.e. BEHIND the line -- are pointed at stderr for the life of the process; the line goes to the saved descriptor."""
    global _LINE_OUT
    sys.stdout.flush()
    _LINE_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)


def _emit(obj):
    out = _LINE_OUT or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="2x")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bc-steps", type=int, default=10, help="timed behavioural-cloning steps after the forward measurement (0: skip)")
    ap.add_argument("--bc-warmup", type=int, default=1)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16"], help="operand format of the HEADLINE value (north star: bf16 tiles); the other format is reported beside it")
    ap.add_argument("--no-ingest", action="store_true", help="skip the host -> device ingest leg")
    ap.add_argument("--step-overlap", type=int, default=0, help="1: PolicyEngine.overlap_steps() during the timed forward (A/B)")
    ap.add_argument("--ingest-only", action="store_true", help="(profiling) only the timed forward and the ingest leg; prints the ingest record")
    ap.add_argument("--value-blocks", type=int, default=1, help="0: skip the repeated short forward blocks (`value_blocks` on the line)")
    ap.add_argument("--no-dp-probe", action="store_true", help="skip bc_step.dp_path_one_rank (the data-parallel step over RCCL in a one-rank group)")
    args = ap.parse_args()

    backend = os.environ.get("VPT_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm; "gloo" lets the N > 1 branch run with every rank on one GPU (tests)
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    what, arg = launch_plan(args.gpus, os.environ, n_dev, backend)
    if what == "refuse":
        sys.exit(arg)
    if what == "spawn":
        # `python bench.py --gpus N` by itself: start the N ranks (what `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
        # does when the driver launches it) and hand their exit code back -- a plain --gpus N never degrades to a 1-GPU line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(arg), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    _claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = arg
    distributed = world > 1
    dist = None
    dev_index = local_rank % n_dev
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    copy_stream = torch.cuda.Stream()      # the ingest leg's host -> device copy stream: the process's first side stream (see ingest_leg)
    collective_ranks = 1
    if distributed:      # an ACTUAL collective on device tensors: every rank contributes 1, the sum is the number of ranks the transport reached
        one = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(one)
        torch.cuda.synchronize()
        collective_ranks = int(round(float(one.item())))
        if collective_ranks != world:
            sys.exit(f"bench.py: all-reduce over {dist.get_backend()} summed {collective_ranks} ranks, launcher says {world}")

    import __graft_entry__ as ge
    if distributed:  # one rank compiles (a no-op when the in-tree .so is current), the others wait
        if rank == 0:
            ge.build()
        dist.barrier()
        if rank != 0:
            ge.build()
    else:
        ge.build()
    from vpt_amd import ops
    from vpt_amd.lib.policy import MinecraftAgentPolicy
    from vpt_amd.lib.types import minecraft_action_space
    from vpt_amd import configs   # (oracle/ is only touched by the parity / cpu_baseline legs below)

    head, other = args.precision, ("fp16" if args.precision == "bf16" else "bf16")
    pol = MinecraftAgentPolicy(minecraft_action_space(), configs.policy_kwargs_for(args.model), dict(temperature=2.0), precision=head)
    configs.randomize_(pol, seed=0)
    pol = pol.to(dev)

    B, T = args.batch, args.seq
    g = torch.Generator().manual_seed(1 + rank)
    img = torch.randint(0, 256, (B, T, 128, 128, 3), generator=g, dtype=torch.uint8).to(dev)
    first = torch.zeros(B, T, dtype=torch.bool, device=dev)

    def step(st):
        with torch.no_grad():      # inference path (a grad-enabled call would go through the autograd boundary and keep activations)
            (pd, vpred, _), st = pol({"img": img}, first, st)
        return st

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_forward(steps, warmup):
        state = pol.initial_state(B)
        for _ in range(warmup):
            state = step(state)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            state = step(state)
        barrier()
        el = time.perf_counter() - t0
        if distributed:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, state

    if args.step_overlap:
        pol.overlap_steps(True)
    with BoxSampler(dev_index) as box_fwd:
        elapsed, state = timed_forward(args.steps, args.warmup)
    pol.overlap_steps(False)
    # the same measurement again in short blocks spread over the run (after the headline steps, after the instrumented step, after the BC leg): a
    # box that throttles as it warms up, or a +-2.5 % box effect, shows as the spread of these blocks on the line itself
    blocks = []

    def value_block(tag):
        if args.ingest_only or args.value_blocks <= 0:
            return
        pol.set_precision(head)
        k = max(2, args.steps // 2)
        el_b, _ = timed_forward(k, 1)
        blocks.append(dict(after=tag, steps=k, ms_per_step=round(1e3 * el_b / k, 3)))

    value_block("headline steps")
    if args.ingest_only:
        _emit(dict(forward_ms=round(1e3 * elapsed / args.steps, 3), ingest=ingest_leg(pol, img, first, dev, copy_stream)))
        return
    roof = kernels = None
    if rank == 0:
        roof, kernels = _roofline(ops, pol, step, state, args, B, T, head)
    value_block("instrumented step")

    # ---- the same workload in the OTHER operand format, as a full record (timed steps, roofline of the same kernel, per-kernel
    # table): with --precision bf16 (default) this is the parity mode, the one that meets the north star's 1e-3 ----
    other_rec = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            pol.set_precision(other)
            el_o, st_o = timed_forward(args.steps, args.warmup)
            roof_o, kern_o = _roofline(ops, pol, step, st_o, args, B, T, other)
            fps_o = B * T * args.steps / el_o
            other_rec = dict(precision=other, frames_per_s=round(fps_o, 1), ms_per_step=round(1e3 * el_o / args.steps, 3), steps=args.steps, warmup=args.warmup,
                             e2e_frac_of_mfma_peak=round(fps_o * FLOP_PER_FRAME.get(args.model, 0) / MFMA_BF16_PEAK, 4), roofline=roof_o, kernels=kern_o,
                             note=f"precision='{other}' ({'libvpt_hip_f16.so' if other == 'fp16' else 'libvpt_hip.so'}): identical sources and workload, timed like the headline; not the headline value")
            del st_o
        except Exception as e:
            other_rec = dict(precision=other, error=f"{type(e).__name__}: {e}")
        finally:
            pol.set_precision(head)

    timed_parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:      # parity ON the timed batch, with the weights the timed steps ran (before the BC leg's Adam steps move them), both formats
            timed_parity = headline_batch_parity(pol, args.model, img, first, dev, (head, other))
        except Exception as e:
            timed_parity = dict(error=f"{type(e).__name__}: {e}")
        pol.set_precision(head)

    bc = bc_other = None
    if args.bc_steps > 0:
        try:
            bc = _bc_leg(pol, args, img, first, g, dev, world, distributed, barrier, dist, head)
        except Exception as e:  # the forward line must survive a failure of the training leg
            bc = dict(error=f"{type(e).__name__}: {e}")
        if world == 1 and not args.no_cpu_baseline:
            try:
                import copy
                a2 = copy.copy(args)
                a2.bc_steps = max(2, args.bc_steps // 2)
                a2.no_dp_probe = True          # (the one-rank RCCL probe once, in the headline format)
                bc_other = _bc_leg(pol, a2, img, first, g, dev, world, distributed, barrier, dist, other)
            except Exception as e:
                bc_other = dict(precision=other, error=f"{type(e).__name__}: {e}")
        pol.set_precision(head)
    value_block("BC leg")

    frames_total = world * B * T * args.steps
    fps = frames_total / elapsed
    if rank == 0:
        par = f"dp{world} replicas (no collective in forward; BC step: gradient all-reduce)"
        if distributed:
            par += f"; torch.distributed backend={dist.get_backend()} world_size={dist.get_world_size()}" + (" (RCCL)" if dist.get_backend() == "nccl" else "")
            par += f"; collective_ranks={collective_ranks} (sum of ones over an all-reduce of device tensors)"
        bms = sorted(b["ms_per_step"] for b in blocks)
        vblocks = None
        if bms:
            med = bms[len(bms) // 2]
            vblocks = dict(blocks=blocks, median_ms_per_step=med, spread_pct=round(100.0 * (bms[-1] - bms[0]) / med, 2),
                           headline_ms_per_step=round(1e3 * elapsed / args.steps, 3),
                           note="short repeats of the timed forward spread over the run; `value` itself comes from the K contract steps only")
        # ---- key order: the contract keys, then the bulky detail, then -- LAST, so that the tail of stdout holds them -- parity_status, parity_mode,
        # bc_step, box, value_blocks, roofline, cpu_baseline (the driver keeps the last few KB of stdout; round 5's bf16 bc_step was cut off) ----
        line = {
            "metric": "frames/sec (fwd) [+ bc_step.ms_per_step], 2x policy, 128x128x3 seq=128",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": head, "data": "synthetic",
            "config": {"workload": f"foundation-model-{args.model} policy forward (IMPALA CNN + 4-layer banded transformer + heads), "
                                   f"batch={B} seq={T} uint8 128x128x3 frames per GPU, KV memory carried, random-init weights",
                       "global_batch": world * B, "seq_len": T, "parallelism": par},
            "e2e_tflops": round(fps * FLOP_PER_FRAME.get(args.model, 0) / 1e12, 1),
            "e2e_frac_of_mfma_peak": round(fps * FLOP_PER_FRAME.get(args.model, 0) / MFMA_BF16_PEAK, 4),
            "kernels": _trim_kernels(kernels),
        }
        tail = {"bc_step": bc, "box": box_fwd.record(), "value_blocks": vblocks, "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            if isinstance(other_rec, dict) and "kernels" in other_rec:
                other_rec["kernels"] = _trim_kernels(other_rec["kernels"])
            line[f"{other}_mode"] = other_rec
            try:
                line["parity"] = parity_block(args.model, dev)
            except Exception as e:
                line["parity"] = dict(error=f"{type(e).__name__}: {e}")
            line["parity"]["timed_batch"] = timed_parity       # (measured before the BC leg, see above)
            try:
                line["parity"]["competitive_heads"] = competitive_heads_parity(args.model, dev, (head, other))
            except Exception as e:
                line["parity"]["competitive_heads"] = dict(error=f"{type(e).__name__}: {e}")
            # which throughput has EARNED the north star's 1e-3 gate (log-prob relative L2 on the timed batch): the fp16 record
            pm = other_rec if other == "fp16" else dict(frames_per_s=round(fps, 1), roofline=roof)
            status, pmode = _parity_status(line["parity"], head)
            if isinstance(pm, dict) and "error" not in pm:
                tb = (line["parity"].get("timed_batch") or {}).get("fp16", {})
                pmode = {"dtype": "fp16", "value": pm.get("frames_per_s"), "unit": "frames/s",
                         "roofline_frac": (pm.get("roofline") or {}).get("frac"),
                         "logprob_rel_l2_on_timed_batch": tb.get("logprob_rel_l2"), "logprob_max_rel_on_timed_batch": tb.get("logprob_max_rel"),
                         "centred_logits_rel_l2_on_timed_batch": tb.get("centred_logits_rel_l2"), "meets_1e-3": tb.get("meets_1e-3"),
                         "note": "precision='fp16' (the policy classes' default) is the format whose log-probs are within 1e-3 relative L2 of the fp32 reference; the headline "
                                 "`value` is the north star's bf16-tile format, whose parity figures are in parity.* (1.5-4x the tolerance by construction: bf16 operands carry 8 mantissa bits)"}
            del pol, img
            torch.cuda.empty_cache()
            if not args.no_ingest:
                # The ingest leg runs in a process of its own (`bench.py --ingest-only`, same batch / model / precision): a loader process holds the
                # forward's streams and ONE copy stream, and that is what decides whether the copy gets a hardware queue to itself -- measured inside this
                # process, after the BC legs had created a dozen streams, the same pipeline showed no overlap at all (profiles/r05_experiments.md section 5)
                try:
                    import subprocess
                    cmd = [sys.executable, os.path.abspath(__file__), "--ingest-only", "--model", args.model, "--batch", str(B), "--seq", str(T), "--steps", "3", "--warmup", "1",
                           "--precision", head]
                    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
                    recs = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                    if p.returncode != 0 or not recs:
                        raise RuntimeError(f"rc {p.returncode}: {p.stderr[-400:]}")
                    sub = json.loads(recs[-1])
                    line["ingest"] = dict(sub["ingest"], forward_ms_in_that_process=sub["forward_ms"], process="separate (bench.py --ingest-only)")
                except Exception as e:
                    line["ingest"] = dict(error=f"{type(e).__name__}: {e}")
            try:
                line["configs"] = configs_block(dev)
            except Exception as e:
                line["configs"] = dict(error=f"{type(e).__name__}: {e}")
            if bc_other is not None:
                line["bc_step_" + other] = bc_other
            line["parity_status"] = status
            line["parity_mode"] = pmode
            tail["cpu_baseline"] = cpu_baseline(args.model)
        # scalars of the BC half of the metric and of the box ON the roofline object as well: the driver's record keeps `roofline` / `cpu_baseline` whole
        if isinstance(roof, dict):
            if isinstance(bc, dict) and "ms_per_step" in bc:
                roof.update(bc_step_ms_per_step=bc["ms_per_step"], bc_step_frac_of_mfma_peak=bc.get("frac_of_mfma_peak"), bc_step_dtype=bc.get("precision"))
            bx = tail["box"]
            if isinstance(bx, dict) and "sclk_mhz_sustained" in bx:
                roof.update(sclk_mhz_sustained=bx["sclk_mhz_sustained"], power_w_avg=bx["power_w_avg"], power_cap_w=bx.get("power_cap_w"))
            if vblocks:
                roof.update(value_blocks_spread_pct=vblocks["spread_pct"])
            if line.get("parity_status"):
                roof.update(parity_status=line["parity_status"][:118])
        line.update(tail)
        _emit(line)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
