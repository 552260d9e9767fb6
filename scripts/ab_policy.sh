# A/B of step-kernel builds under a TRAINED policy's actions (scripts/nconhist_policy.py per build): bash scripts/ab_policy.sh variantA variantB ...
for rep in 1 2; do
  for lib in "$@"; do
    L=nightmare_rl_amd/csrc/libnightmare_hip${lib:+_$lib}.so
    echo -n "lib=$(basename $L)  "; NM_HIP_LIB=$PWD/$L python scripts/nconhist_policy.py 150 2>&1 | grep "step kernel under"
  done
done
