#!/bin/bash
# Round-5 extras ON the GPU box (scripts/measure_extra.sh <tag>): batch-size scaling of the step kernel, stage ablation, a 300-iteration training
# run, the step kernel under a trained policy, per-stage instruction counts in the standing regime. Outputs under gpurun_out/<tag>_*.
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
echo "== batch scaling (kernel only, random actions | standing)"
for n in 1024 2048 4096 8192 16384 32768; do python scripts/quickbench.py $n 0 1.0 2>&1 | grep "step kernel"; done > $OUT/${TAG}_batch_scaling.txt
for n in 4096 32768; do python scripts/quickbench.py $n 0 0.12 2>&1 | grep "step kernel"; done >> $OUT/${TAG}_batch_scaling.txt
cat $OUT/${TAG}_batch_scaling.txt
echo "== stage ablation"; python scripts/ablate.py > $OUT/${TAG}_ablation.txt 2>&1; grep "kernel avg" $OUT/${TAG}_ablation.txt
echo "== train 300 iterations"; rm -rf logs; python train.py -e 4096 --iters 300 > $OUT/${TAG}_train300.log 2>&1; grep -E "^it +(1|40|100|200|299)/" $OUT/${TAG}_train300.log | cut -c1-150
echo "== step kernel under a trained policy (150 iterations)"; python scripts/nconhist_policy.py 150 2>&1 | grep -v amdgpu > $OUT/${TAG}_trained_policy_regime.txt; cat $OUT/${TAG}_trained_policy_regime.txt
echo "== per-stage instruction counts, standing regime (0.12 x actions)"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmcmask_stand; mkdir -p $OUT/pmcmask_stand
for M in 0 1 4 13; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES -d "$OUT/pmcmask_stand/m$M" -o run -- python3 "$ROOT/scripts/pmcmask.py" $M 0.12 > "$OUT/pmcmask_stand/m$M.log" 2>&1 || echo "mask $M failed"
done
python3 - "$OUT/pmcmask_stand" <<'P' | tee $OUT/${TAG}_stage_instruction_counts_standing.txt
import sys, glob, csv, collections
out = sys.argv[1]
names = {0: "full", 1: "no collision (=> no contacts, no constraints)", 4: "no constraint stage", 13: "load + integrate + epilogue only"}
res = {}
for m in (0, 1, 4, 13):
    f = glob.glob(f"{out}/m{m}/**/*counter_collection.csv", recursive=True)
    if not f: print(m, "no csv"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "k_env_step" in r["Kernel_Name"]]
    per = collections.defaultdict(dict)
    for r in rows: per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[-10:]
    avg = {c: sum(per[i][c] for i in ids) / len(ids) for c in per[ids[0]]}
    w = avg.get("SQ_WAVES", 2048)
    res[m] = avg["SQ_INSTS_VALU"] / w
    print(f"standing regime, mask {m:2d} {names[m]:48s}: per wave VALU {avg['SQ_INSTS_VALU'] / w:8.0f}  SALU {avg['SQ_INSTS_SALU'] / w:7.0f}  LDS {avg['SQ_INSTS_LDS'] / w:7.0f}  VMEM {avg['SQ_INSTS_VMEM'] / w:6.0f}")
if len(res) == 4:
    print(f"   constraints {res[0] - res[4]:.0f}  collision {res[4] - res[1]:.0f}  smooth dynamics {res[1] - res[13]:.0f}  load + 2 x integrate + epilogue {res[13]:.0f}   VALU per wave and step")
P
echo done
