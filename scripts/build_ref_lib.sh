# Build the HIP library of another commit as nightmare_rl_amd/csrc/libnightmare_hip_<name>.so, for A/B runs on one GPU box
# (scripts/ab_quick.sh): bash scripts/build_ref_lib.sh <git rev> <name>
set -e
rev=$1; name=$2
d=$(mktemp -d)
git archive "$rev" nightmare_rl_amd/csrc nightmare_rl_amd/model/nm_model_data.h include | tar -x -C "$d"
make -C "$d/nightmare_rl_amd/csrc" -s
cp "$d/nightmare_rl_amd/csrc/libnightmare_hip.so" "nightmare_rl_amd/csrc/libnightmare_hip_${name}.so"
rm -rf "$d"
