#!/bin/bash
# rocprofv3 --pmc passes over the one-launch rollout kernel k_env_rollout (scripts/pmcrollout.py: 4096 envs x 80 steps per launch, policy in the wave).
# Run ON the GPU box:  scripts/pmc_rollout.sh <tag>   -> gpurun_out/pmc_rollout_<tag>/p*/ ; summary printed by scripts/pmc_rollout_summary.py
# Counters are collected in separate passes with --kernel-trace only (no other trace domain), as the counter guide prescribes.
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_rollout_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"
P3="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
P4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
P5="FETCH_SIZE"
P6="WRITE_SIZE"
P7="GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6" "$P7"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $P -d "$OUT/p$i" -o run -- python3 "$ROOT/scripts/pmcrollout.py" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
  echo "pass $i done"
done
python3 "$ROOT/scripts/pmc_rollout_summary.py" "$OUT"
