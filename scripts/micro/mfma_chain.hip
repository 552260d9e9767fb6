// How fast does ONE wave issue v_mfma_f32_16x16x4_f32 as a function of the number of independent accumulation chains, and what do two
// waves on a SIMD make of it? (Question behind k_ppo_fwdbwd_split: its pairs of output tiles are 2 chains per wave.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
// Prints s_memtime ticks per MFMA for K = 1, 2, 3, 4, 8 chains at 1 and 2 waves per SIMD (one workgroup per CU, 4 or 8 waves), with and
// without a filler of 4 VALU instructions per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int K, int FILL>
__global__ void __launch_bounds__(512) chain(int iters, float* out, unsigned long long* ticks) {
  f32x4 acc[K];
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f, x = a, y = b;
#pragma unroll
  for (int k = 0; k < K; k++) acc[k] = f32x4{0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int k = 0; k < K; k++) {
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
        if (FILL) { x = x * 1.0001f + y; y = y * 0.9999f + x; x = x * 1.0001f + y; y = y * 0.9999f + x; }
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = x + y;
#pragma unroll
  for (int k = 0; k < K; k++) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int K, int FILL> void run(int waves, float* out, unsigned long long* ticks) {
  const int iters = 2000, nb = 256;
  chain<K, FILL><<<nb, 64 * waves>>>(iters, out, ticks);
  hipDeviceSynchronize();
  chain<K, FILL><<<nb, 64 * waves>>>(iters, out, ticks);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < nb; i++) m += (double)h[i];
  m /= nb;
  const double per_wave = m / (iters * 8.0 * K);
  printf("chains %d  fill %d  waves/SIMD %d: %6.1f ticks per MFMA of a wave = %6.1f ticks per MFMA of the SIMD\n", K, FILL, waves / 4, per_wave, per_wave / (waves / 4));
}
int main() {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 256 * 8);
  for (int waves = 4; waves <= 8; waves += 4) {
    run<1, 0>(waves, out, ticks); run<2, 0>(waves, out, ticks); run<3, 0>(waves, out, ticks); run<4, 0>(waves, out, ticks); run<8, 0>(waves, out, ticks);
    run<1, 1>(waves, out, ticks); run<2, 1>(waves, out, ticks); run<4, 1>(waves, out, ticks);
  }
  return 0;
}
