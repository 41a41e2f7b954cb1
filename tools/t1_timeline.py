"""Timeline of the captured T = 1 acting step from a rocprofv3 --kernel-trace CSV: per kernel (in launch order inside one step) the median
duration and the median gap since the previous kernel's end, over the last N replays.   python tools/t1_timeline.py <kernel_trace.csv> [steps]"""
import csv, statistics, sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:46]))
rows.sort()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
# a step starts with the first conv kernel
starts = [i for i, r in enumerate(rows) if r[2].startswith("vpt_conv_first_kernel")]
starts = starts[-nsteps - 1:]
per = len(set(b - a for a, b in zip(starts, starts[1:])))
L = statistics.mode(b - a for a, b in zip(starts, starts[1:]))
steps = [rows[a:b] for a, b in zip(starts, starts[1:]) if b - a == L]
print(f"{len(steps)} steps of {L} kernels (distinct lengths seen: {per})")
tot_d = tot_g = 0.0
for k in range(L):
    d = statistics.median((s[k][1] - s[k][0]) / 1e3 for s in steps)
    g = statistics.median((s[k][0] - (s[k - 1][1] if k else s[k][0])) / 1e3 for s in steps)
    tot_d += d; tot_g += g
    print(f"{k:3d} {steps[0][k][2]:46s} dur {d:7.2f} us   gap before {g:6.2f} us")
span = statistics.median((s[-1][1] - s[0][0]) / 1e3 for s in steps)
period = statistics.median((b[0][0] - a[0][0]) / 1e3 for a, b in zip(steps, steps[1:]))
print(f"sum of durations {tot_d:.1f} us, sum of gaps {tot_g:.1f} us, first start -> last end {span:.1f} us, step period {period:.1f} us")
