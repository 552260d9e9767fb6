#!/bin/bash
# Everything the round's numbers come from, in one go ON the GPU box (scripts/measure_round.sh <tag>); outputs under gpurun_out/<tag>_*.
# Each rocprofv3 run has the program itself after "--"; counters are collected in separate --pmc passes with --kernel-trace only.
TAG=${1:-rXX}
PART=${2:-all}      # a: bench / train / curves / rocprof kernel stats; b: PMC passes, wave lifetimes, stage stamps, parity report (each fits one gpurun call)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
if [ "$PART" != "b" ]; then
echo "== bench"; python bench.py > $OUT/${TAG}_bench.json.log 2>&1; tail -c 600 $OUT/${TAG}_bench.json.log; echo
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_driverargs.json.log 2>&1
echo "== train 40 iterations + play"; rm -rf logs; python train.py -e 4096 --iters 40 > $OUT/${TAG}_train40.log 2>&1; grep -E "^it +(1|20|39)/" $OUT/${TAG}_train40.log
python scripts/play.py --log-root logs/nightmare_v3 -e 64 --steps 400 > $OUT/${TAG}_play.log 2>&1; tail -4 $OUT/${TAG}_play.log
echo "== curves (HIP side; the CPU side of the same size: profiles/r05_curve_cpu_runs.json (round-5 tables: see DESIGN 6.3), tests/tools/curve_vs_cpu.py --kinds cpu in the build container)"
python tests/tools/curve_vs_cpu.py --kinds hip --envs 4096 --iters 150 --seeds 3 --merge profiles/r05_curve_cpu_runs.json --out $OUT/${TAG}_curve_vs_cpu.json > $OUT/${TAG}_curve.log 2>&1; grep -E "^it +[0-9]+  mean" $OUT/${TAG}_curve.log
echo "== one-launch rollout"; python scripts/rolloutbench.py 4096 80 > $OUT/${TAG}_rolloutbench.txt 2>&1; grep -v amdgpu $OUT/${TAG}_rolloutbench.txt
python scripts/rolloutwaves.py 4096 80 > $OUT/${TAG}_rolloutwaves.txt 2>&1; grep -v amdgpu $OUT/${TAG}_rolloutwaves.txt
cd /tmp && export TMPDIR=/tmp
echo "== rocprof kernel stats: bench"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_bench -o run -- python3 $ROOT/bench.py --steps 200 --warmup 50 --no-cpu-baseline > $OUT/${TAG}_prof_bench.log 2>&1
echo "== rocprof kernel stats: train"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_train -o run -- python3 $ROOT/train.py -e 4096 --iters 12 > $OUT/${TAG}_prof_train.log 2>&1
echo "== rocprof kernel trace: train down the MULTI-RANK path on one rank (NM_FORCE_DATA_PARALLEL=1: RCCL all-reduce per mini-batch inside the update graph, pipelined logging with its all-reduce)"
NM_FORCE_DATA_PARALLEL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_train_dp -o run -- python3 $ROOT/train.py -e 4096 --iters 12 > $OUT/${TAG}_prof_train_dp.log 2>&1
python3 $ROOT/scripts/trace_gaps.py $OUT/${TAG}_prof_train_dp > $OUT/${TAG}_train_dp_gaps.txt 2>&1; python3 $ROOT/scripts/trace_gaps.py $OUT/${TAG}_prof_train >> $OUT/${TAG}_train_dp_gaps.txt 2>&1; cat $OUT/${TAG}_train_dp_gaps.txt
echo "== rocprof kernel stats + MFMA counters: mlp"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_mlp -o run -- python3 $ROOT/scripts/mlpbench.py > $OUT/${TAG}_prof_mlp.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmc_mlp -o run -- python3 $ROOT/scripts/mlpbench.py 4096 big > $OUT/${TAG}_pmc_mlp.log 2>&1
for d in bench train train_dp mlp; do f=$(ls $OUT/${TAG}_prof_$d/*kernel_stats.csv $OUT/${TAG}_prof_$d/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_${d}_kernel_stats.csv && head -6 $OUT/${TAG}_${d}_kernel_stats.csv | cut -c1-200; done
python $ROOT/scripts/pmc_summary.py $OUT/${TAG}_pmc_mlp k_mlp_fused > $OUT/${TAG}_pmc_mlp_summary.txt 2>&1; cat $OUT/${TAG}_pmc_mlp_summary.txt | head -12
echo "== PPO forward / backward kernel: timing, SQ counters, memory-side counters"
cd $ROOT; python scripts/ppostamps.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_ppobench.txt; cat $OUT/${TAG}_ppobench.txt
[ -f nightmare_rl_amd/csrc/libnightmare_hip_ppostamps.so ] && NM_HIP_LIB=nightmare_rl_amd/csrc/libnightmare_hip_ppostamps.so python scripts/ppostamps.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_ppo_stage_stamps.txt
bash scripts/pmc_ppo.sh ${TAG} > $OUT/${TAG}_pmc_ppo_summary.txt 2>&1; tail -4 $OUT/${TAG}_pmc_ppo_summary.txt
bash scripts/pmc_ppo_mem.sh ${TAG} > $OUT/${TAG}_pmc_ppo_mem_summary.txt 2>&1; tail -3 $OUT/${TAG}_pmc_ppo_mem_summary.txt
fi
if [ "$PART" != "a" ]; then
cd /tmp && export TMPDIR=/tmp
echo "== PMC passes: one-launch rollout kernel"; cd $ROOT; bash scripts/pmc_rollout.sh ${TAG} > $OUT/${TAG}_pmc_rollout_summary.txt 2>&1; tail -12 $OUT/${TAG}_pmc_rollout_summary.txt; cp $OUT/pmc_rollout_${TAG}_summary.json $OUT/${TAG}_pmc_rollout.json
echo "== PMC passes: step kernel"; cd $ROOT; scripts/pmc_collect.sh ${TAG} > /dev/null 2>&1; python scripts/pmc_summary.py $OUT/pmc_${TAG} > $OUT/${TAG}_pmc_step_summary.txt 2>&1; tail -8 $OUT/${TAG}_pmc_step_summary.txt
echo "== wave lifetimes + stage stamps + parity report"
python scripts/wavetimes.py > $OUT/${TAG}_wavetimes.txt 2>&1; tail -3 $OUT/${TAG}_wavetimes.txt
[ -f nightmare_rl_amd/csrc/libnightmare_hip_stamps.so ] && NM_HIP_LIB=nightmare_rl_amd/csrc/libnightmare_hip_stamps.so python scripts/stamps.py > $OUT/${TAG}_stage_stamps.txt 2>&1
python tests/tools/parity_report.py > $OUT/${TAG}_parity_report.txt 2>&1; tail -5 $OUT/${TAG}_parity_report.txt
echo "== lane utilisation / fp32 operation counts per stage"; bash scripts/pmclanes.sh > $OUT/${TAG}_pmclanes_run.log 2>&1; cp $OUT/pmclanes/summary.txt $OUT/${TAG}_pmc_lanes_summary.txt; cp $OUT/pmclanes/lanes.json $OUT/${TAG}_pmc_lanes.json; cat $OUT/${TAG}_pmc_lanes_summary.txt
fi
echo done
