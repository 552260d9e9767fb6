#!/bin/bash
# Dynamic instruction counts of the step kernel per stage: scripts/pmcmask.sh  (ON the GPU box) -> gpurun_out/pmcmask/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcmask
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for M in 0 1 4 13; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES -d "$OUT/m$M" -o run -- python3 "$ROOT/scripts/pmcmask.py" $M > "$OUT/m$M.log" 2>&1 || echo "mask $M failed"
  echo "mask $M done"
done
python3 - "$OUT" <<'P' | tee "$OUT/summary.txt"
import sys, glob, csv, collections
out = sys.argv[1]
names = {0: "full", 1: "no collision (=> no contacts, no constraints)", 4: "no constraint stage", 9: "no collision, no smooth stage", 13: "load + integrate + epilogue only", 32: "constraint stage one env at a time"}
for m in (0, 1, 4, 13):
    f = glob.glob(f"{out}/m{m}/**/*counter_collection.csv", recursive=True)
    if not f: print(m, "no csv"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "k_env_step" in r["Kernel_Name"]]
    per = collections.defaultdict(dict)
    for r in rows: per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[-10:]
    avg = {c: sum(per[i][c] for i in ids) / len(ids) for c in per[ids[0]]}
    w = avg.get("SQ_WAVES", 2048)
    print(f"mask {m:2d} {names[m]:48s}: per wave VALU {avg['SQ_INSTS_VALU'] / w:8.0f}  SALU {avg['SQ_INSTS_SALU'] / w:7.0f}  LDS {avg['SQ_INSTS_LDS'] / w:7.0f}  VMEM {avg['SQ_INSTS_VMEM'] / w:6.0f}")
P
