// The constraint solver's Gauss-Seidel sweep as the step kernel runs it for two envs on half-waves (nm_core.h stage_constraint2): row i of
// env 0 lives in lane i, row i of env 1 in lane 32 + i, every lane keeps its row of A in 32 registers, and row i's change dl_i must reach
// every other row's residual before that row's own turn: g_j += A[j][i] dl_i. What bounds a sweep is the chain row i -> row i + 1.
// Variants (ticks per row update, s_memtime, one-wave workgroups at 1 or 2 waves per SIMD):
//   V0  the round-4 kernel: res = g + R f; dl = max(-res / AR, -f); select(run); broadcast = 2 v_readlane + 2 v_mov + v_cndmask; g += A dl
//   V1  per-sweep constants k0 = -1/AR, k1 = -(R f)/AR, nf = -f folded with the run mask: dl = max(fma(g, k0, k1), nf); same broadcast
//   V2  V1 + the NEXT row served by a DPP shift (wave_shr:1 moves lane i -> i + 1 in BOTH halves at once, no SGPR round trip):
//       gt = fma(shr1(dl), Asub, g) is what row i + 1 computes from, the readlane broadcast feeds everybody else off the critical path
//   V4  V2 with the shift folded into the multiply-add (v_fmac_f32_dpp, inline assembly)
//   V5  V1 with the broadcast as ONE ds_bpermute_b32 (LDS crossbar: lane l reads lane (l & 32) + i; no VALU slot, no SGPR round trip) and
//       the capture of a row's own dl under a literal lane mask in an SGPR pair (no v_cmp): 4 VALU + 1 DS per row instead of 13 VALU
//   V6  V5 + the wave_shr:1 fast path of V2 (the DS latency off the row-to-row chain)
//   V8  V1 + the SGPR-mask capture: what the step kernel runs since round 5 (9 VALU per row)
//   V7  V8 with the half-wave broadcast as two v_fmac_f32 with a scalar operand each, under the halves' exec masks (7 VALU + 3 SALU per row)
//   V3  V2 with row_shr:1 + row_bcast:15 at the row-of-16 boundaries instead of wave_shr:1 (if wave_shr were not available)
// V1, V2, V3 give bit-identical f and g (checked here); V0 differs from them by the rounding of the folded constants.
//   hipcc --offload-arch=gfx950 -O3 -o pgs_chain pgs_chain.hip && ./pgs_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define DEV __device__ __forceinline__
DEV float rdl(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
template <int CTRL> DEV float dppmov(float old, float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false)); }
constexpr int kRows = 32;
template <int V>
__global__ void __launch_bounds__(64) k(int nrows, int sweeps, const float* __restrict__ Ain, float* out, unsigned long long* ticks) {
  const int lane = threadIdx.x, hl = lane & 31;
  const bool h1 = lane >= 32;
  float A[kRows];
#pragma unroll
  for (int i = 0; i < kRows; i++) A[i] = Ain[(blockIdx.x & 7) * 64 * kRows + lane * kRows + i];
  const float Ajj = 2.0f + 0.01f * hl, Rr = 0.05f + 0.001f * hl, ARinv = 1.0f / (Ajj + Rr);
  float Asub = 0.0f;                       // A[lane][lane - 1]: the entry that couples this row to the one before it
#pragma unroll
  for (int i = 0; i < kRows; i++) Asub = (hl == i + 1) ? A[i] : Asub;
  float f = 0.1f + 0.01f * hl, g = -0.5f + 0.03f * hl + (h1 ? 0.2f : 0.0f);
  int ln = lane; asm volatile("" : "+v"(ln));
  const int lv = ln & 31;
  const int hb4 = (lane & 32) * 4;
  const bool run = true;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < sweeps; s++) {
    float dcap = 0.0f;
    if (V == 0) {
#pragma unroll
      for (int i = 0; i < kRows; i++) {
        if ((i & 3) == 0 && i >= nrows) break;
        const float res = g + Rr * f;
        float dl = fmaxf(-res * ARinv, -f);
        dl = run ? dl : 0.0f;
        const float s0 = rdl(dl, i), s1 = rdl(dl, 32 + i);
        const float b = h1 ? s1 : s0;
        g += A[i] * b;
        dcap = (lv == i) ? dl : dcap;
      }
    } else {
      const float k0 = run ? -ARinv : 0.0f, k1 = run ? (Rr * f) * -ARinv : 0.0f, nf = run ? -f : 0.0f;
      float gt = g;
#pragma unroll
      for (int i = 0; i < kRows; i++) {
        if ((i & 3) == 0 && i >= nrows) break;
        const float dl = fmaxf(__builtin_fmaf((V == 1 || V == 5 || V == 7 || V == 8) ? g : gt, k0, k1), nf);
        if (V == 2) gt = __builtin_fmaf(dppmov<0x138>(dl, dl), Asub, g);                      // wave_shr:1
        if (V == 4) {      // the shift folded into the multiply-add: v_fmac_f32 with a DPP source (VOP2), one instruction on the chain
          gt = g;
          asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(gt) : "v"(dl), "v"(Asub));
        }
        if (V == 7 || V == 8) {      // what the kernel does since round 5: capture the row's g under an SGPR literal mask (V8), and the broadcast as two
                                     // multiply-adds with a scalar operand each under the halves' exec masks (V7)
          asm("s_mov_b32 vcc_lo, %3\n\ts_mov_b32 vcc_hi, %3\n\tv_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(dcap) : "v"(dcap), "v"(dl), "n"(1u << i) : "vcc");
          const float s0 = rdl(dl, i), s1 = rdl(dl, 32 + i);
          if (V == 8) { const float b = h1 ? s1 : s0; g = __builtin_fmaf(A[i], b, g); }
          else asm("s_nop 1\n\ts_mov_b32 exec_hi, 0\n\tv_fmac_f32_e32 %0, %2, %1\n\ts_not_b64 exec, exec\n\tv_fmac_f32_e32 %0, %3, %1\n\ts_mov_b64 exec, -1"
                   : "+v"(g) : "v"(A[i]), "s"(s0), "s"(s1) : "scc");
          continue;
        }
        if (V == 5 || V == 6) {      // broadcast through the LDS crossbar (ds_bpermute_b32: one DS instruction, no VALU slot, no SGPR), capture under an SGPR literal mask
          if (V == 6) gt = __builtin_fmaf(dppmov<0x138>(dl, dl), Asub, g);
          const float b = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(hb4 + 4 * i, __builtin_bit_cast(int, dl)));
          g = __builtin_fmaf(A[i], b, g);
          const unsigned long long m = (1ull << i) | (1ull << (32 + i));
          asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(dcap) : "v"(dcap), "v"(dl), "s"(m));
          continue;
        }
        if (V == 3) {
          float sh = dppmov<0x111>(dl, dl);                                                   // row_shr:1
          if ((i & 15) == 15) sh = dppmov<0x142>(sh, dl);                                     // row_bcast:15: lane 15 of a row -> the next row
          gt = __builtin_fmaf(sh, Asub, g);
        }
        const float s0 = rdl(dl, i), s1 = rdl(dl, 32 + i);
        const float b = h1 ? s1 : s0;
        g = __builtin_fmaf(A[i], b, g);
        dcap = (lv == i) ? dl : dcap;
      }
    }
    f += dcap;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 128 + lane] = f;
  out[blockIdx.x * 128 + 64 + lane] = g;
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}
static float hA[8 * 64 * kRows];
template <int V> double run(int grid, int nrows, int sweeps, const float* A, float* out, unsigned long long* ticks, float* res) {
  k<V><<<grid, 64>>>(nrows, sweeps, A, out, ticks); (void)hipDeviceSynchronize();
  k<V><<<grid, 64>>>(nrows, sweeps, A, out, ticks); (void)hipDeviceSynchronize();
  static unsigned long long h[2048];
  (void)hipMemcpy(h, ticks, grid * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(res, out, 8 * 128 * 4, hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < grid; i++) m += (double)h[i]; m /= grid;
  return m / ((double)sweeps * nrows);
}
int main() {
  float *A, *out; unsigned long long* ticks;
  (void)hipMalloc(&A, sizeof(hA)); (void)hipMalloc(&out, 2048 * 128 * 4); (void)hipMalloc(&ticks, 2048 * 8);
  unsigned s = 12345u;
  for (size_t i = 0; i < sizeof(hA) / 4; i++) { s = s * 1664525u + 1013904223u; hA[i] = ((s >> 8) & 0xffff) / 65536.0f * 0.06f - 0.03f; }
  (void)hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice);
  static float r[9][8 * 128];
  for (int grid = 1024; grid <= 2048; grid += 1024)
    for (int nrows = 8; nrows <= 32; nrows += 8) {
      const double t0 = run<0>(grid, nrows, 300, A, out, ticks, r[0]), t1 = run<1>(grid, nrows, 300, A, out, ticks, r[1]);
      const double t2 = run<2>(grid, nrows, 300, A, out, ticks, r[2]), t3 = run<3>(grid, nrows, 300, A, out, ticks, r[3]);
      const double t4 = run<4>(grid, nrows, 300, A, out, ticks, r[4]);
      const double t5 = run<5>(grid, nrows, 300, A, out, ticks, r[5]), t6 = run<6>(grid, nrows, 300, A, out, ticks, r[6]);
      printf("    V5 ds_bpermute broadcast + SGPR-mask capture %.1f | V6 = V5 + wave_shr:1 fast path %.1f  [V5 == V1: %s, V6 == V1: %s]\n", t5, t6,
             memcmp(r[1], r[5], sizeof(r[1])) ? "NO" : "yes", memcmp(r[1], r[6], sizeof(r[1])) ? "NO" : "yes");
      const double t7 = run<7>(grid, nrows, 300, A, out, ticks, r[7]), t8 = run<8>(grid, nrows, 300, A, out, ticks, r[8]);
      printf("    V8 = V1 + SGPR-mask capture (the round-5 kernel) %.1f | V7 = V8 with the broadcast as two exec-masked v_fmac with a scalar operand %.1f  [V8 == V1: %s, V7 == V1: %s]\n", t8, t7,
             memcmp(r[1], r[8], sizeof(r[1])) ? "NO" : "yes", memcmp(r[1], r[7], sizeof(r[1])) ? "NO" : "yes");
      printf("waves/SIMD %d, %2d rows per env: ticks per row update  V0 %.1f | V1 folded constants %.1f | V2 + wave_shr:1 fast path %.1f | V3 row_shr + row_bcast %.1f | V4 v_fmac_dpp %.1f"
             "   [V2 == V1 bitwise: %s, V3 == V1: %s, V4 == V1: %s, max |V0 - V1| %.2e]\n", grid / 1024, nrows, t0, t1, t2, t3, t4,
             memcmp(r[1], r[2], sizeof(r[1])) ? "NO" : "yes", memcmp(r[1], r[3], sizeof(r[1])) ? "NO" : "yes", memcmp(r[1], r[4], sizeof(r[1])) ? "NO" : "yes",
             [&] { double m = 0; for (int i = 0; i < 8 * 128; i++) { double d = r[0][i] - r[1][i]; m = d < 0 ? (m > -d ? m : -d) : (m > d ? m : d); } return m; }());
    }
  return 0;
}
