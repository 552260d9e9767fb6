#!/bin/bash
# memory-side counters of the PPO forward / backward kernel (see scripts/pmc_ppo.sh). Run ON the GPU box: scripts/pmc_ppo_mem.sh <tag>
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_ppomem_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
P2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
P3="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TCP_TA_TCP_STATE_READ_sum"
P4="TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $P -d "$OUT/p$i" -o run -- python3 "$ROOT/scripts/ppostamps.py" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python3 "$ROOT/scripts/pmc_summary.py" "$OUT" k_ppo_fwdbwd
