#!/bin/bash
# rocprofv3 --pmc passes over the step kernel (scripts/pmcrun.py: 4096 envs, 360 steps). Run ON the GPU box:
#   scripts/pmc_collect.sh <tag> [mask]      -> gpurun_out/pmc_<tag>/<pass>/...   then  python scripts/pmc_summary.py gpurun_out/pmc_<tag>
# Counters are collected in passes of <= 8 SQ counters, each in its own run with --kernel-trace only (no other trace domain).
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_TRANS_F32"
P4="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL"
P5="FETCH_SIZE"
P6="WRITE_SIZE"
P7="GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6" "$P7"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $P -d "$OUT/p$i" -o run -- python3 "$ROOT/scripts/pmcrun.py" 4096 > "$OUT/p$i.log" 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
  echo "pass $i done"
done
