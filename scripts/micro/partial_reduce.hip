// Is the partial-gradient reduction of k_ppo_step slow because of its access pattern?  256 writer blocks each write a row of 27 944 floats
// (row-major: rows 112 KB apart) or tiles of 64 floats (tile-major: the 256 rows' pieces of a tile are adjacent); then 236 reader blocks
// of 8 waves add the 256 rows for 64 consecutive positions each (what k_ppo_reduce does).  Time of the reader, both layouts.
//   hipcc --offload-arch=gfx950 -O3 -o partial_reduce partial_reduce.hip && ./partial_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kRows = 256, kStride = 27944, kUsed = 15104;     // 236 x 64 positions are read
__device__ __forceinline__ size_t at(int layout, int m, int row) { return layout == 0 ? (size_t)row * kStride + m : (size_t)(m >> 6) * (kRows * 64) + row * 64 + (m & 63); }
__global__ void writer(float* p, int layout) {
  const int row = blockIdx.x;
  for (int m = threadIdx.x; m < kUsed; m += blockDim.x) p[at(layout, m, row)] = (float)(row + m % 7);
}
__global__ void __launch_bounds__(512) reader(const float* __restrict__ p, int layout, float* __restrict__ out) {
  __shared__ float part[8][64];
  const int lane = threadIdx.x & 63, w0 = threadIdx.x >> 6, m = blockIdx.x * 64 + lane;
  float g0 = 0, g1 = 0, g2 = 0, g3 = 0;
  for (int w = w0; w + 24 < kRows; w += 32) { g0 += p[at(layout, m, w)]; g1 += p[at(layout, m, w + 8)]; g2 += p[at(layout, m, w + 16)]; g3 += p[at(layout, m, w + 24)]; }
  part[w0][lane] = (g0 + g1) + (g2 + g3);
  __syncthreads();
  if (w0 == 0) { float g = 0; for (int w = 0; w < 8; w++) g += part[w][lane]; out[m] = g; }
}
int main() {
  float *p, *out;
  hipMalloc(&p, (size_t)kRows * kStride * 4 + (1 << 20)); hipMalloc(&out, kUsed * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int layout = 0; layout < 2; layout++)
    for (int rep = 0; rep < 3; rep++) {
      float tot = 0;
      for (int i = 0; i < 20; i++) {
        writer<<<kRows, 256>>>(p, layout);
        hipEventRecord(e0);
        reader<<<kUsed / 64, 512>>>(p, layout, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
      }
      printf("layout %s: reader %.1f us (event pair around the launch, after a writer launch)\n", layout ? "tile-major" : "row-major ", tot / 20 * 1e3);
    }
  return 0;
}
