# A/B of step-kernel builds on one GPU box: bash scripts/ab_quick.sh variantA variantB ...   ("" = the shipped library; name = libnightmare_hip_<name>.so)
set -e
for rep in 1 2; do
  for lib in "$@"; do
    L=nightmare_rl_amd/csrc/libnightmare_hip${lib:+_$lib}.so
    NM_HIP_LIB=$PWD/$L python scripts/quickbench.py 4096 0 1.0 2>&1 | grep "step kernel"
    NM_HIP_LIB=$PWD/$L python scripts/quickbench.py 4096 0 0.12 2>&1 | grep "step kernel"
  done
done
