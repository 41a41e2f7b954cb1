"""Turn the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/prof_<R>/) into the summaries committed under profiles/.
python tools/profile_summarize.py [R=r02] [tag=r02]  ->  profiles/<tag>_bench_*_kernel_stats.csv, <tag>_conv3x3_pmc.json,
<tag>_bench_conv3x3_traffic.json (+ the bench lines printed by the profiled runs)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

R = sys.argv[1] if len(sys.argv) > 1 else "r02"
TAG = sys.argv[2] if len(sys.argv) > 2 else R
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
DST = os.path.join(ROOT, "profiles")


def one(pattern):
    g = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return g[0] if g else None


def last_json_line(path):
    if not path or not os.path.exists(path):
        return None
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return line
    return None


# (1) / (2) kernel-trace stats + the bench lines of the profiled runs
for sub, name in (("fwd1", "fwd_singlestream"), ("fwd1_fp16", "fwd_singlestream_fp16"), ("fwdbc", "fwd_bc")):
    f = one(f"{sub}/**/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(DST, f"{TAG}_bench_{name}_kernel_stats.csv"))
    line = last_json_line(os.path.join(SRC, f"{sub}_bench.json"))
    if line:
        open(os.path.join(DST, f"{TAG}_bench_{name}_under_rocprof.json"), "w").write(line + "\n")
    if f:
        rows = [r for r in csv.DictReader(open(f)) if "vpt_conv3x3_kernel" in r["Name"]]
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        n = sum(int(r["Calls"]) for r in rows)
        print(f"{sub}: vpt_conv3x3_kernel {n} launches, {tot / 1e6:.1f} ms total, {tot / max(n, 1) / 1e3:.1f} us average")

# (2b) single-stream BC step (round 5)
f = one("bc1/**/*_kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(DST, f"{TAG}_bc_singlestream_kernel_stats.csv"))
    if os.path.exists(os.path.join(SRC, "bc1.log")):
        shutil.copy(os.path.join(SRC, "bc1.log"), os.path.join(DST, f"{TAG}_bc_singlestream_bc_bench.log"))
if os.path.exists(os.path.join(SRC, "latency.log")):
    shutil.copy(os.path.join(SRC, "latency.log"), os.path.join(DST, f"{TAG}_latency_bench.log"))

# (3) PMC of the conv micro-benchmark: per-launch averages by (grid size, kernel instantiation)
SHAPES = {1048576: "s0.block 64x64 128->128 (256 frames)", 2097152: "256-cout layers (s1.first / s1.block / s2.first)",
          524288: "s2.block 16x16 256->256"}
import glob as _g
_variants = [("", "bf16", "")] if _g.glob(os.path.join(SRC, "pmc_sq", "**", "*_counter_collection.csv"), recursive=True) else []
_variants += [("_bf16", "bf16", ""), ("_fp16", "fp16", "_fp16"), ("_bf16_t32", "bf16", "_tiles32")]   # _tiles32: the 32-row / eight-wave tile variant (tiling 3)
for SFX, FMT, FSFX in _variants:
    if not _g.glob(os.path.join(SRC, f"pmc_sq{SFX}", "**", "*_counter_collection.csv"), recursive=True):
        continue
    pmc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(lambda: defaultdict(list))
    for sub in (f"pmc_sq{SFX}", f"pmc_grbm{SFX}"):
        f = one(f"{sub}/**/*_counter_collection.csv")
        if not f:
            continue
        seen = set()
        for r in csv.DictReader(open(f)):
            if "vpt_conv3x3_kernel" not in r["Kernel_Name"]:
                continue
            res = "res" if (", 1>" in r["Kernel_Name"] or "<false, 1," in r["Kernel_Name"] or "<false, 5," in r["Kernel_Name"]) else "nores"   # template <trace, mode[, tile rows]>: modes 1 / 5 carry a residual
            key = f"grid{r['Grid_Size']}_{res}"
            pmc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if (sub, r["Dispatch_Id"]) not in seen:
                seen.add((sub, r["Dispatch_Id"]))
                dur[key][sub].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {"kernel": f"vpt_conv3x3_kernel ({TAG}, operand format {FMT})", "counters": {}}
    for key, cs in sorted(pmc.items()):
        d = {k: sum(v) / len(v) for k, v in cs.items()}
        d["_dur_ns_sq_pass"] = sum(dur[key][f"pmc_sq{SFX}"]) / max(len(dur[key][f"pmc_sq{SFX}"]), 1)
        d["_dur_ns_grbm_pass"] = sum(dur[key][f"pmc_grbm{SFX}"]) / max(len(dur[key][f"pmc_grbm{SFX}"]), 1)
        grid = int(key[4:].split("_")[0])
        d["shape"] = SHAPES.get(grid, f"grid {grid}")
        d["residual"] = key.endswith("_res")
        if d.get("GRBM_GUI_ACTIVE") and d["_dur_ns_grbm_pass"]:
            d["effective_clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / d["_dur_ns_grbm_pass"]   # the counter is summed over the 8 XCDs
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (256 CUs x 4); cycles of the SQ pass = GRBM cycles per XCD scaled by
        # the two passes' durations
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES") and d.get("GRBM_GUI_ACTIVE"):
            d["mfma_busy_frac_of_cycles"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] * d["_dur_ns_sq_pass"] / d["_dur_ns_grbm_pass"] * 128.0)
        if d.get("SQ_WAVE_CYCLES"):
            d["wait_inst_any_frac_of_wave_cycles"] = d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
            d["wait_any_frac_of_wave_cycles"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        out["counters"][key] = d
    if out["counters"]:
        json.dump(out, open(os.path.join(DST, f"{TAG}_conv3x3_pmc{FSFX}.json"), "w"), indent=1)
        for k, d in out["counters"].items():
            print(f"pmc[{FMT}] {k}: clock {d.get('effective_clock_ghz', 0):.2f} GHz  MFMA busy {100 * d.get('mfma_busy_frac_of_cycles', 0):.1f} %  "
                  f"wait_inst_any {100 * d.get('wait_inst_any_frac_of_wave_cycles', 0):.0f} %  LDS conflicts {100 * d.get('lds_conflict_frac', 0):.1f} %")


# (4) HBM traffic of the conv kernel over the bench workload, per operand format
# Algorithmic bytes of all vpt_conv3x3_kernel launches per frame of the 2x model, 16-bit activations, since round 4
# (pool-fused firstconvs: the pre-pool tensor is not written; pooled tensor + seam rows / columns instead):
#   stack-0 blocks: 4 x (1 MB in + 1 MB out) + 2 x 1 MB residual = 10 MB;  s1.first: 1 in + 0.5 pooled + 0.19 seams;  stack-1 blocks: 5 MB;
#   s2.first: 0.5 + 0.125 + 0.03;  stack-2 blocks: 1.25 MB   ->  18.6 MB / frame, 152.4 GB per step of 8192 frames (round 3: 20.25 MB, 173.9 GB)
ALG_PER_STEP = 18.6e6 * 8192
for FSFX in ("", "_fp16"):
    tot = {}
    launches = 0
    for sub, cname in ((f"pmc_fetch{FSFX}", "FETCH_SIZE"), (f"pmc_write{FSFX}", "WRITE_SIZE")):
        f = one(f"{sub}/**/*_counter_collection.csv")
        if not f:
            continue
        s, n = 0.0, 0
        for r in csv.DictReader(open(f)):
            if "vpt_conv3x3_kernel" in r["Kernel_Name"] and r["Counter_Name"] == cname:
                s += float(r["Counter_Value"]); n += 1
        tot[cname] = s
        launches = n
    if len(tot) == 2 and launches:
        hbm = (tot["FETCH_SIZE"] * 2.0 + tot["WRITE_SIZE"]) * 1024.0
        per_pass = 112.0
        try:
            per_pass = float(json.loads(open(os.path.join(SRC, f"pmc_fetch{FSFX}.json")).read().strip().splitlines()[-1])["roofline"]["launches"])
        except Exception:
            pass
        passes = launches / per_pass
        t = dict(fetch_size_kb_sum=tot["FETCH_SIZE"], write_size_kb_sum=tot["WRITE_SIZE"], launches=launches, launches_per_step=per_pass,
                 hbm_bytes_total_fetch_x2_plus_write=hbm, hbm_bytes_per_step=hbm / passes, algorithmic_bytes_per_step=ALG_PER_STEP,
                 hbm_bytes_per_launch=hbm / launches, algorithmic_bytes_per_launch=ALG_PER_STEP / per_pass, ratio_measured_over_algorithmic=hbm / passes / ALG_PER_STEP,
                 note="forward pass(es) of the bench workload (2x, 64x128 frames), `rocprofv3 --kernel-trace --pmc "
                      "FETCH_SIZE` / `--pmc WRITE_SIZE` in separate passes; FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); KB units x1024; "
                      "all vpt_conv3x3_kernel launches incl. the pool-fused mode")
        json.dump(t, open(os.path.join(DST, f"{TAG}_bench_conv3x3_traffic{FSFX}.json"), "w"), indent=1)
        print(f"traffic{FSFX}: {hbm / launches / 1e9:.3f} GB per launch over {launches} launches (x{t['ratio_measured_over_algorithmic']:.3f} of algorithmic)")
