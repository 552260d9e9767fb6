// Micro-benchmark: latency of a dependent 16 B/lane gather from an L2-resident 1 MB table, with the occupancy of the step kernel
// (2 waves per SIMD, every CU busy). build: hipcc -O3 --offload-arch=gfx950 gather_latency.hip -o gather_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64, 2) k(const f4* __restrict__ t, int nvert, int iters, int rowlen, float* out) {
  __shared__ float pad[4600];   // 18.4 KB: 8 waves per CU like the step kernel
  int lane = threadIdx.x;
  pad[lane] = 0;
  unsigned v = (blockIdx.x * 2654435761u) % nvert;
  float acc = 0;
  for (int i = 0; i < iters; i++) {
    f4 x = t[(size_t)v * rowlen + (lane < rowlen ? lane : rowlen - 1)];
    acc += x.x;
    unsigned w = __builtin_amdgcn_readlane(__float_as_uint(x.w), 5);   // next vertex depends on the loaded data
    v = (w + i * 7919u + blockIdx.x) % nvert;
  }
  if (acc == 12345.f) out[0] = acc + pad[lane];
}
int main() {
  const int nvert = 1734, rowlen = 35, iters = 200;
  std::vector<f4> h((size_t)nvert * rowlen);
  for (size_t i = 0; i < h.size(); i++) { h[i].x = 1; h[i].y = 2; h[i].z = 3; unsigned r = (unsigned)(i * 2246822519u); unsigned rr = r >> 8; float fw; memcpy(&fw, &rr, 4); h[i].w = fw; }
  f4* d; float* o;
  hipMalloc(&d, h.size() * sizeof(f4)); hipMalloc(&o, 4);
  hipMemcpy(d, h.data(), h.size() * sizeof(f4), hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int waves : {256, 1024, 2048, 4096}) {
    hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d, nvert, iters, rowlen, o);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d, nvert, iters, rowlen, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("waves %5d: %.1f ns per dependent gather (kernel %.1f us)\n", waves, ms * 1e6 / iters, ms * 1e3);
  }
  return 0;
}
