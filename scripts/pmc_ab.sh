# instruction counts of two builds of the step kernel, random and standing regimes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_ab
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-base ""}; do
  for sc in ${SCALES:-1.0 0.12}; do
    L=$ROOT/nightmare_rl_amd/csrc/libnightmare_hip${lib:+_$lib}.so
    export NM_HIP_LIB=$L
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -d $OUT/${lib:-new}_$sc -o run -- python3 $ROOT/scripts/pmcrun.py 4096 $sc > $OUT/${lib:-new}_$sc.log 2>&1
    rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_BUSY_CYCLES -d $OUT/${lib:-new}_${sc}_ic -o run -- python3 $ROOT/scripts/pmcrun.py 4096 $sc > $OUT/${lib:-new}_${sc}_ic.log 2>&1
    echo "== ${lib:-new} scale $sc"; python3 $ROOT/scripts/pmc_summary.py $OUT/${lib:-new}_$sc | grep -v "^kernel\|summary" ; python3 $ROOT/scripts/pmc_summary.py $OUT/${lib:-new}_${sc}_ic | grep "ICACHE\|IFETCH\|duration"
  done
done
