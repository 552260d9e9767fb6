#!/bin/bash
# Lane utilisation and fp32 operation counts of the step kernel, per stage (SURVEY 8(d) secondary roofline): scripts/pmclanes.sh (ON the GPU
# box) -> gpurun_out/pmclanes/summary.txt. One rocprofv3 --pmc pass per stage mask of the -DNM_MEASURE build (scripts/pmcmask.py), with
# --kernel-trace only. SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = active lanes per VALU instruction; ADD + MUL + 2 FMA + TRANS wave
# instructions x active lanes = fp32 FLOPs actually executed.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmclanes
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for M in 0 1 4 13; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_WAVES -d "$OUT/m$M" -o run -- python3 "$ROOT/scripts/pmcmask.py" $M > "$OUT/m$M.log" 2>&1 || echo "mask $M failed"
  echo "mask $M done"
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU SQ_WAVES -d "$OUT/k0" -o run -- python3 "$ROOT/scripts/pmcmask.py" 0 > "$OUT/k0.log" 2>&1 || echo "kinds pass failed"
python3 - "$OUT" <<'P' | tee "$OUT/summary.txt"
import sys, glob, csv, collections
out = sys.argv[1]
names = {0: "full step", 1: "no collision (=> no contacts, no constraints)", 4: "no constraint stage", 13: "load + integrate + epilogue only"}
def avg_of(d):
    f = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    if not f: return None
    rows = [r for r in csv.DictReader(open(f[0])) if "k_env_step" in r["Kernel_Name"]]
    per = collections.defaultdict(dict)
    for r in rows: per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[-10:]
    return {c: sum(per[i][c] for i in ids) / len(ids) for c in per[ids[0]]}
res = {}
for m in (0, 1, 4, 13):
    a = avg_of(f"m{m}")
    if a is None: print(m, "no csv"); continue
    res[m] = a
    w = a.get("SQ_WAVES", 2048)
    lanes = a["SQ_THREAD_CYCLES_VALU"] / max(a["SQ_ACTIVE_INST_VALU"], 1)
    fl_inst = a["SQ_INSTS_VALU_ADD_F32"] + a["SQ_INSTS_VALU_MUL_F32"] + 2 * a["SQ_INSTS_VALU_FMA_F32"] + a["SQ_INSTS_VALU_TRANS_F32"]
    print(f"mask {m:2d} {names[m]:46s}: VALU/wave {a['SQ_INSTS_VALU'] / w:7.0f}  active lanes per VALU instruction {lanes:5.1f} of 64 ({lanes / 64:.1%})  "
          f"f32 add {a['SQ_INSTS_VALU_ADD_F32'] / w:6.0f} mul {a['SQ_INSTS_VALU_MUL_F32'] / w:6.0f} fma {a['SQ_INSTS_VALU_FMA_F32'] / w:6.0f} trans {a['SQ_INSTS_VALU_TRANS_F32'] / w:5.0f} per wave  "
          f"-> {fl_inst * lanes / 1e6:8.2f} MFLOP per launch (fp32, active lanes)")
if 0 in res and 1 in res and 4 in res and 13 in res:
    def stage(hi, lo, label):
        a, b = res[hi], res[lo]
        tc, ai = a["SQ_THREAD_CYCLES_VALU"] - b["SQ_THREAD_CYCLES_VALU"], a["SQ_ACTIVE_INST_VALU"] - b["SQ_ACTIVE_INST_VALU"]
        print(f"   stage {label:44s}: {(a['SQ_INSTS_VALU'] - b['SQ_INSTS_VALU']) / a.get('SQ_WAVES', 2048):7.0f} VALU/wave, active lanes {tc / max(ai, 1):5.1f} ({tc / max(ai, 1) / 64:.1%})")
    stage(0, 4, "constraints (mask 0 - mask 4)")
    stage(4, 1, "collision (mask 4 - mask 1)")
    stage(1, 13, "smooth dynamics (mask 1 - mask 13)")
    a = res[13]
    print(f"   stage {'load + 2 x integrate + epilogue (mask 13)':44s}: {a['SQ_INSTS_VALU'] / a.get('SQ_WAVES', 2048):7.0f} VALU/wave, active lanes {a['SQ_THREAD_CYCLES_VALU'] / max(a['SQ_ACTIVE_INST_VALU'], 1):5.1f}")
import json
if 0 in res:
    a = res[0]
    lanes = a["SQ_THREAD_CYCLES_VALU"] / max(a["SQ_ACTIVE_INST_VALU"], 1)
    fl = (a["SQ_INSTS_VALU_ADD_F32"] + a["SQ_INSTS_VALU_MUL_F32"] + 2 * a["SQ_INSTS_VALU_FMA_F32"] + a["SQ_INSTS_VALU_TRANS_F32"]) * lanes
    json.dump({"source": "rocprofv3 --kernel-trace --pmc pass of scripts/pmcmask.py 0 (4096 envs, the 10 last launches), scripts/pmclanes.sh",
               "kernel": "k_env_step<float,2>", "envs": 4096,
               "active_lanes_per_valu_instruction": lanes, "valu_insts_per_wave": a["SQ_INSTS_VALU"] / a.get("SQ_WAVES", 2048),
               "f32_wave_instructions_per_launch": {"add": a["SQ_INSTS_VALU_ADD_F32"], "mul": a["SQ_INSTS_VALU_MUL_F32"], "fma": a["SQ_INSTS_VALU_FMA_F32"], "trans": a["SQ_INSTS_VALU_TRANS_F32"]},
               "fp32_flop_per_launch": fl,
               "definition": "(add + mul + 2 fma + trans wave-instructions) x active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU); "
                             "'active' = exec-mask lanes: the leg-lane stages run exact duplicates on lanes that hold no leg, which this counts as work"},
              open(out + "/lanes.json", "w"), indent=1)
k = avg_of("k0")
if k:
    w = k.get("SQ_WAVES", 2048)
    print("instruction kinds per wave (full step): " + "  ".join(f"{c[14:]} {k[c] / w:.0f}" for c in sorted(k) if c.startswith("SQ_INSTS_VALU_")) + f"  all VALU {k['SQ_INSTS_VALU'] / w:.0f}")
P
