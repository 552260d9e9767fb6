/* nightmare_hip_measure.h - entry points that exist ONLY in the measurement build of the library
 * (make -C nightmare_rl_amd/csrc measure: -DNM_MEASURE -> libnightmare_hip_measure.so). The shipped libnightmare_hip.so does not
 * export them and does not read the NM_MEASURE_* environment variables; nothing in the product path loads the measurement build
 * (users: scripts/ablate.py, scripts/quickbench.py, and the device bit-equality test of the two-env constraint pass, which needs
 * the "one env at a time" switch). No upstream counterpart. */
#ifndef NIGHTMARE_HIP_MEASURE_H
#define NIGHTMARE_HIP_MEASURE_H
#include "nightmare_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Results become wrong (except bit5): skip kernel stages to attribute time. bit0 collision, bit1 solver sweeps, bit2 whole
 * constraint stage, bit3 smooth-dynamics stage, bit4 tibia pairs, bit5 constraint stage one env at a time (same results),
 * bit7 env epilogue, bit8 observation, bit9 the empty launch (every wave returns at once), bit10 load stage only, bit11 no substeps
 * (load + epilogue). 0 = normal. Also only in this build: NM_MEASURE_PGS_ITERS / NM_MEASURE_NOSLIP_ITERS
 * (environment, read by nm_create) override the solver sweep counts. */
int nm_set_ablation(nm_env* env, int32_t mask);
#ifdef __cplusplus
}
#endif
#endif
