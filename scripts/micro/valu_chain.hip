// What does one instruction of the step kernel's kinds cost a wave that issues it in a DEPENDENT chain (the step kernel is 9.9 k VALU
// instructions per wave at 2 waves per SIMD, most of them in short dependent chains)?  s_memtime ticks per instruction, one workgroup
// of 64 threads per SIMD slot (grid 1024 or 2048 one-wave workgroups = 1 or 2 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o valu_chain valu_chain.hip && ./valu_chain
#include <hip/hip_runtime.h>
#include <cstdio>
enum { FMA_IND, FMA_DEP, DPP_DEP, READLANE_DEP, LDS_RT, LDS_BCAST, CNDMASK_DEP, RCP_DEP, NKIND };
static const char* kNames[NKIND] = {"v_fma_f32, 8 independent chains", "v_fma_f32, one dependent chain", "v_add_f32 row_shr:1 (DPP), dependent", "v_readlane_b32 -> v_fma_f32 (SGPR operand), dependent",
                                    "ds_write_b32 + ds_read_b32 round trip (wave-private LDS)", "ds_read_b32 broadcast (uniform address), dependent address", "v_cmp + v_cndmask, dependent", "v_rcp_f32, dependent"};
template <int KIND>
__global__ void __launch_bounds__(64) k(int iters, float* out, unsigned long long* ticks) {
  __shared__ float lds[128];
  float x = threadIdx.x * 1e-3f + 1.0f, y = 0.5f, z[8];
  for (int i = 0; i < 8; i++) z[i] = x + i;
  lds[threadIdx.x] = x; lds[64 + threadIdx.x] = 0.0f;
  int idx = threadIdx.x & 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if (KIND == FMA_IND) {
#pragma unroll
        for (int c = 0; c < 8; c++) z[c] = z[c] * 1.0001f + y;
      } else if (KIND == FMA_DEP) {
        x = x * 1.0001f + y;
      } else if (KIND == DPP_DEP) {
        x = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, false));
      } else if (KIND == READLANE_DEP) {
        const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 5));
        x = x * 0.5f + s;
      } else if (KIND == LDS_RT) {
        lds[threadIdx.x] = x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        x = lds[threadIdx.x ^ 1] * 0.999f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      } else if (KIND == LDS_BCAST) {
        const int a = __builtin_amdgcn_readfirstlane(idx);
        const float v = lds[64 + a];
        idx = (int)v;                    // 0: the next address depends on the value read
        x += v;
      } else if (KIND == CNDMASK_DEP) {
        x = x > y ? x * 0.5f : x + 1.0f;
      } else if (KIND == RCP_DEP) {
        x = __builtin_amdgcn_rcpf(x) + 1.0f;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = x + idx;
  for (int i = 0; i < 8; i++) s += z[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int KIND> void run(int grid, float* out, unsigned long long* ticks, int per_iter) {
  const int iters = 500;
  k<KIND><<<grid, 64>>>(iters, out, ticks); (void)hipDeviceSynchronize();
  k<KIND><<<grid, 64>>>(iters, out, ticks); (void)hipDeviceSynchronize();
  static unsigned long long h[2048];
  (void)hipMemcpy(h, ticks, grid * 8, hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < grid; i++) m += (double)h[i]; m /= grid;
  printf("%-66s waves/SIMD %d: %6.1f ticks per instruction (group)\n", kNames[KIND], grid / 1024, m / (iters * 16.0 * per_iter));
}
int main() {
  float* out; unsigned long long* ticks;
  (void)hipMalloc(&out, 2048 * 64 * 4); (void)hipMalloc(&ticks, 2048 * 8);
  for (int grid = 1024; grid <= 2048; grid += 1024) {
    run<FMA_IND>(grid, out, ticks, 8); run<FMA_DEP>(grid, out, ticks, 1); run<DPP_DEP>(grid, out, ticks, 1); run<READLANE_DEP>(grid, out, ticks, 1);
    run<LDS_RT>(grid, out, ticks, 1); run<LDS_BCAST>(grid, out, ticks, 1); run<CNDMASK_DEP>(grid, out, ticks, 1); run<RCP_DEP>(grid, out, ticks, 1);
  }
  return 0;
}
