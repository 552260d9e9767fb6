#!/bin/bash
# rocprofv3 --pmc passes over the PPO forward / backward kernel (scripts/ppostamps.py: 81 920-row mini-batches of the reference's network).
# Run ON the GPU box:  scripts/pmc_ppo.sh <tag>   (NM_PPO_FAST4=1 in the environment selects the four-wave kernel)
#   -> gpurun_out/pmc_ppo_<tag>/p*/ ; summary: python scripts/pmc_summary.py gpurun_out/pmc_ppo_<tag> k_ppo_fwdbwd
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_ppo_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
P3="GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $P -d "$OUT/p$i" -o run -- python3 "$ROOT/scripts/ppostamps.py" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
python3 "$ROOT/scripts/pmc_summary.py" "$OUT" k_ppo_fwdbwd
