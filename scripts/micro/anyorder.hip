// Do two back-to-back dispatches on ONE stream overlap when the second is launched with hipExtAnyOrderLaunch (AQL packet without the
// barrier bit)? A kernel of 64 one-wave workgroups that spins ~50 us; 10 launches in order vs any-order.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long ticks, int* out) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) atomicAdd(out, 1);
}
int main() {
  int* out; hipMalloc(&out, 4); hipMemset(out, 0, 4);
  hipStream_t s; hipStreamCreate(&s);
  for (int flags = 0; flags < 2; flags++)
    for (int rep = 0; rep < 3; rep++) {
      hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 10; i++) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, flags, 100000LL, out);
      hipError_t e = hipStreamSynchronize(s);
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("flags=%d rep %d: 10 launches in %.1f us (%s)\n", flags, rep, us, hipGetErrorString(e));
    }
  int h; hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost); printf("waves counted %d (expect %d)\n", h, 64 * 60);
  return 0;
}
