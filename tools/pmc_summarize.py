"""Sum rocprofv3 --pmc counter_collection.csv files per kernel name (all passes under a directory) and print one line per kernel:
launches, total duration, counters.  python tools/pmc_summarize.py <dir>"""
import collections, csv, glob, sys

root = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
n = collections.defaultdict(int)
for f in sorted(glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        did = (f, r["Dispatch_Id"])
        if did not in seen and "pmc1" in f:
            seen.add(did)
            n[k] += 1
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
    c = tot[k]
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    la = c.get("SQ_LDS_IDX_ACTIVE", 0) or 1
    print(f"{k:44s} n={n[k]:4d} ms={dur[k]:8.2f} active={c.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} valu={c.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} "
          f"lds={c.get('SQ_ACTIVE_INST_LDS',0)/wc:.2f} wait_any={c.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={c.get('SQ_WAIT_INST_ANY',0)/wc:.2f} "
          f"lds_conf={c.get('SQ_LDS_BANK_CONFLICT',0)/la:.2f} lds_unal={c.get('SQ_LDS_UNALIGNED_STALL',0)/la:.2f} ldsMcyc={la/1e6:.0f} "
          f"valuM={c.get('SQ_INSTS_VALU',0)/1e6:.0f} saluM={c.get('SQ_INSTS_SALU',0)/1e6:.0f} ldsM={c.get('SQ_INSTS_LDS',0)/1e6:.1f} vmemM={c.get('SQ_INSTS_VMEM',0)/1e6:.1f} "
          f"mfmaMcyc={c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1e6:.0f} busyM={c.get('SQ_BUSY_CYCLES',0)/1e6:.0f} fetchGB={c.get('FETCH_SIZE',0)*2*1024/1e9:.2f} writeGB={c.get('WRITE_SIZE',0)*1024/1e9:.2f}"
          f" hbmTBs={(c.get('FETCH_SIZE',0)*2+c.get('WRITE_SIZE',0))*1024/1e9/max(dur[k],1e-9):.2f}")   # FETCH_SIZE / WRITE_SIZE are in KB; FETCH x2 on gfx950 (MI355X_MICROARCH.md)
