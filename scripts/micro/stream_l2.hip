// stream_l2.hip - how fast can ONE workgroup (8 waves, as in k_mlp_fused) stream a small, L2-resident array through dwordx4 loads,
// as a function of the number of loads each wave keeps in flight? Answers whether the policy kernel's weight stream is bound by
// per-CU delivery rate or by latency x bytes in flight.   hipcc --offload-arch=gfx950 -O3 -o stream_l2 stream_l2.hip && ./stream_l2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int D> __global__ void __launch_bounds__(512) k_stream(const f32x4* __restrict__ w, int nper, int reps, float* out, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const f32x4* base = w + (size_t)wave * nper * 64;      // this wave's slice: nper fragments of 1 KiB
  f32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
    f32x4 ring[D];
#pragma unroll
    for (int j = 0; j < D; j++) ring[j] = base[(unsigned)((j % nper) * 64 + lane)];
    for (int g = 0; g < nper; g += D) {
#pragma unroll
      for (int j = 0; j < D; j++) {
        acc += ring[j];
        const int gn = g + j + D < nper ? g + j + D : 0;
        ring[j] = base[(unsigned)(gn * 64 + lane)];
      }
    }
#pragma unroll
    for (int j = 0; j < D; j++) acc += ring[j];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int D> void run(const f32x4* w, int nper, int grid, float* out, unsigned long long* cyc) {
  const int reps = 20;
  hipLaunchKernelGGL(k_stream<D>, dim3(grid), dim3(512), 0, 0, w, nper, reps, out, cyc);
  hipLaunchKernelGGL(k_stream<D>, dim3(grid), dim3(512), 0, 0, w, nper, reps, out, cyc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s = 0; for (auto x : h) s += (double)x;
  const double c = s / grid / reps, bytes = 8.0 * nper * 1024;
  printf("grid %4d  in flight/wave %2d x 1 KiB: %8.0f cycles per pass of %3.0f KiB  = %5.1f B/clk/CU (s_memtime ticks at 100 MHz: x%.0f shader clocks)\n", grid, D, c, bytes / 1024, bytes / c, 1.0);
}
int main() {
  const int nper = 44;    // 8 waves x 44 KiB = 352 KiB ~ the 66-256-256-18 network
  f32x4* w; float* out; unsigned long long* cyc;
  hipMalloc(&w, (size_t)8 * nper * 1024); hipMemset(w, 0, (size_t)8 * nper * 1024);
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  for (int grid : {1, 16, 256}) {
    run<2>(w, nper, grid, out, cyc); run<4>(w, nper, grid, out, cyc); run<8>(w, nper, grid, out, cyc); run<16>(w, nper, grid, out, cyc); run<32>(w, nper, grid, out, cyc);
  }
  return 0;
}
