"""Median per-launch value of every counter collected by scripts/pmc_collect.sh for the step kernel; derived ratios."""
import csv, glob, json, sys
import numpy as np
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_env_step"
vals = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
med = {k: float(np.median(v[len(v) // 4:])) for k, v in vals.items()}       # skip the settling launches
for k in sorted(med):
    print(f"{k:28s} n={len(vals[k]):4d} median={med[k]:16.1f}")
if dur:
    print(f"kernel duration under the profiler: median {np.median(dur[len(dur) // 4:]):.1f} us over {len(dur)} launches")
g = med.get
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    print(f"wave cycles: active {100 * g('SQ_ACTIVE_INST_ANY', 0) / wc:.1f} %  wait_any {100 * g('SQ_WAIT_ANY', 0) / wc:.1f} %  wait_inst {100 * g('SQ_WAIT_INST_ANY', 0) / wc:.1f} %")
out = {k: med[k] for k in med}
json.dump(out, open(d.rstrip("/") + "_summary.json", "w"), indent=1)
