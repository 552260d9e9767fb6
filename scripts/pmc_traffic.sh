#!/bin/bash
# HBM-side traffic of the step kernel only (FETCH_SIZE, WRITE_SIZE: one counter per pass), for A/B-ing builds on the GPU box:
#   scripts/pmc_traffic.sh <tag>          (NM_HIP_LIB in the environment selects the library)
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmct_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $P -d "$OUT/$P" -o run -- python3 "$ROOT/scripts/pmcrun.py" 4096 > "$OUT/$P.log" 2>&1 || echo "pass $P failed"
done
cd $ROOT && python scripts/pmc_summary.py $OUT | grep -E "FETCH|WRITE|duration"
