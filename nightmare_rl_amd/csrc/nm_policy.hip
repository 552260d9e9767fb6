// nm_policy.hip - ActorCritic MLP forward (rsl_rl v1.0.2 ActorCritic: Linear -> ELU ... -> Linear; reference call sites
// play.py:122, train.py:40) as a batched GEMM chain on the matrix cores: exact-f32 MFMA (v_mfma_f32_16x16x4_f32), bias + ELU fused
// into the accumulator epilogue, all layers in ONE launch.
//
// Shape of the problem: M = num_envs rows (4096), K, N <= 256: 0.7 GFLOP for 66->256->256->18 - a small GEMM chain whose floor is
// the matrix pipe (16 rows per CU on 256 CUs: 1360 MFMAs per CU = 10.9k cycles). Design:
//   * one workgroup of 8 waves (two per SIMD) owns 16 rows; activations ping-pong between two LDS tiles;
//   * weights are packed ONCE per parameter update (nm_policy_load) into the exact per-lane order of the MFMA B operand, so a
//     lane fetches 4 k-steps with one 16-byte load (coalesced 1 KiB per wave instruction, L2-resident: 0.35 MB per network);
//     K is padded to 16 (not 64): 66 -> 80;
//   * activations sit in LDS as [row][k mod 4][k div 4], so a lane's 4 k-steps are one ds_read_b128;
//   * a wave runs two accumulator chains (two 16-column tiles) to hide the 40-cycle dependent latency behind the 32-cycle issue;
//   * narrow layers (the 18-wide head: 2 tiles for 8 waves) split K across the waves and reduce through LDS.
// State lives in a handle (nm_policy): packed weights, layer table, scratch - one per network, no process-wide statics.
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/nightmare_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

extern "C" int nm_policy_set_error(const char* m);

constexpr int kTileRows = 16, kMaxDim = 256, kMaxLayers = 4;
constexpr int kMlpThreads = 512, kMlpWaves = kMlpThreads / 64;
constexpr int kMaxSteps = kMaxDim / 4;            // k-steps of 4 per layer (one MFMA 16x16x4 each)
constexpr int kQStride = kMaxSteps + 4;           // floats between the four (k mod 4) planes of a row: 16 B aligned, and the 16 columns a
                                                  // wave's epilogue stores per instruction land in 16 different banks (64 would put 4 in one)
#ifndef NM_MLP_RING
#define NM_MLP_RING 4
#endif
constexpr int kRing = NM_MLP_RING;                // weight fragment pairs in flight per wave (k-groups of 16 inputs, 2 KiB each); a power of two.
                                                  // An L2 hit is back in 300-500 cycles (scripts/micro/stream_l2.hip), a group is 256 matrix-pipe cycles
constexpr int kMaxGroups = kMaxDim / 16;          // k-groups per layer at most
constexpr int kActLd = 4 * kQStride + 4;          // floats per activation row in LDS (16 B aligned; 8 consecutive rows cover all banks)

struct MlpArgs {
  const f32x4* w[kMaxLayers];   // packed: [tile][group of 4 k-steps][lane] -> 4 consecutive k-steps of that lane
  const float* b[kMaxLayers];
  int dims[kMaxLayers + 1];
  int n_layers, N;
  const f32x4* wt_last;         // the last layer once more, packed in the k order of the previous layer's accumulators (fuse_last)
  int fuse_last;                // 1: the last layer is computed from the previous layer's registers (see k_mlp_fused)
};

// W [O,K] row-major (torch.nn.Linear) -> packed B operand: lane (r = l & 15, q = l >> 4) of tile t needs, at k-step s,
// W[16 t + r][4 s + q]; zero beyond K or O.
// korder 1: lane (r, q) of group gs holds k = 16 gs + 4 q + j (j = 0..3) - the order in which a lane of the PREVIOUS layer's transposed
// product holds its features (16 t + 4 q + reg), so those accumulators can be this layer's B operand without leaving the registers.
__global__ void k_pack_weights(const float* __restrict__ W, f32x4* __restrict__ P, int O, int K, int ngrp, int ntile, int korder) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntile * ngrp * 64) return;
  const int lane = i & 63, gs = (i >> 6) % ngrp, t = (i >> 6) / ngrp;
  const int r = lane & 15, q = lane >> 4, col = 16 * t + r;
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int k = korder ? 16 * gs + 4 * q + j : 4 * (4 * gs + j) + q;
    v[j] = (col < O && k < K) ? W[(size_t)col * K + k] : 0.0f;
  }
  P[i] = v;
}

// fragment `idx` (units of 16 bytes) of a wave-uniform array: the byte offset is formed in 32 bits, so the load can take the
// `global_load_dwordx4 v, v_offset, s[base]` form (one VALU instruction per load) instead of a 64-bit per-lane address
__device__ __forceinline__ f32x4 ldfrag(const f32x4* base, int idx) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + (unsigned)(idx * 16));
}

// f(0) && f(1) && ... with compile-time indices: a fully unrolled loop whose `break` is a forward branch
template <class F, int... I> __device__ __forceinline__ void unroll_while(std::integer_sequence<int, I...>, F&& f) {
  (void)(f(std::integral_constant<int, I>{}) && ...);
}

// Workgroup barrier for data exchanged through LDS only: wait for this wave's LDS traffic, then s_barrier. __syncthreads() also
// drains the vector-memory queue (s_waitcnt vmcnt(0): a workgroup-scope release fence has to assume global memory), which would wait
// for every weight fragment prefetched across the layer boundary before the barrier instead of underneath it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int act_pos(int k) { return (k & 3) * kQStride + (k >> 2); }   // position of input k inside a row

// what one wave does in one layer: a pair of 16-column tiles and a range of k-groups (narrow layers split K over the waves)
struct WavePlan {
  int ntile, ngrp, split, part, g0, g1, t0, t1;
  bool work, two;
  const f32x4 *w0, *w1;
};
__device__ __forceinline__ WavePlan plan_layer(const MlpArgs& a, int l, int wave, int lane) {
  WavePlan p;
  const int K = a.dims[l], O = a.dims[l + 1];
  p.ntile = (O + 15) >> 4; p.ngrp = (K + 15) >> 4;
  const int npair = (p.ntile + 1) >> 1;
  // split K over 2^lg waves per tile pair when the layer is narrow (fewer pairs than waves) AND deep enough to be worth a
  // reduction round (>= 4 k-groups per part); shifts only, no integer division
  int lg = 0;
  while ((npair << (lg + 1)) <= kMlpWaves && (p.ngrp >> (lg + 1)) >= 4) lg++;
  p.split = 1 << lg;
  const int pr = wave >> lg;
  p.part = wave & (p.split - 1);
  p.g0 = (p.ngrp * p.part) >> lg; p.g1 = (p.ngrp * (p.part + 1)) >> lg;
  p.work = pr < npair;
  p.t0 = 2 * pr; p.t1 = min(2 * pr + 1, p.ntile - 1);
  p.two = 2 * pr + 1 < p.ntile;
  // wave-uniform bases (SGPRs); the lane is added as a 32-bit offset at the access (global_load ... v_off, s[base])
  p.w0 = a.w[l] + ((size_t)(p.work ? p.t0 : 0) * p.ngrp) * 64;
  p.w1 = a.w[l] + ((size_t)(p.work ? p.t1 : 0) * p.ngrp) * 64;
  return p;
}

constexpr int kMaxWavesStamp = 8;
#ifdef NM_MLP_STAMPS   // measurement build: cycle stamps of workgroup 0 (scripts/mlpstamps.py)
__device__ unsigned long long g_mlp_stamps[16];
#define MLP_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mlp_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_mlp_steps[4][kMaxWavesStamp][20];
#define MLP_STEP(l, k) do { if (blockIdx.x == 0 && lane == 0 && (l) < 4) g_mlp_steps[l][wave][k] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int nm_mlp_read_steps(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_steps), sizeof(g_mlp_steps)) == hipSuccess ? 0 : 1;
}
extern "C" int nm_mlp_read_stamps(unsigned long long* out16) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_mlp_stamps), sizeof(g_mlp_stamps)) == hipSuccess ? 0 : 1;
}
#else
#define MLP_STAMP(k)
#define MLP_STEP(l, k)
#endif

__global__ void __launch_bounds__(kMlpThreads) k_mlp_fused(const float* __restrict__ obs, float* __restrict__ out, MlpArgs a) {
  __shared__ __attribute__((aligned(16))) float act[2][kTileRows * kActLd];
  __shared__ __attribute__((aligned(16))) float red[kMlpWaves][8][64];     // split-K partial accumulators of narrow layers
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index as a scalar: everything planned from it (tile pair, k-range, weight base addresses, trip counts) is then
  // wave-uniform to the compiler as well - scalar branches, SGPR base addresses - instead of "divergent" VGPR arithmetic
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * kTileRows;
  MLP_STAMP(0);
  // Weight stream: a ring of kRing fragment pairs per wave, filled kRing k-groups ahead of their use. Every ring slot is a fixed
  // register (all indices below are compile-time after unrolling), so the compiler's s_waitcnt before a group's MFMAs waits for
  // THAT group's two loads only (vmcnt = the loads issued since), not for everything in flight: the rotating-variable version of
  // this loop (c <- n <- f by moves) compiled to `s_waitcnt vmcnt(0)` at the top of every iteration, i.e. a prefetch distance of ONE
  // group = 256 matrix-pipe cycles against an L2 round trip of 500+ (wave active 20 %, matrix pipe busy 30 %).
  f32x4 r0[kRing], r1[kRing];
  // Every ring load is UNCONDITIONAL (a group that does not exist re-reads the wave's first fragment, a line that is hot in the
  // CU's vector cache): only then is the number of loads issued between a fragment's load and its use a compile-time constant. With
  // `if (group exists) load` the compiler has to assume the fewest, and waits with vmcnt(0) again.
  auto fill = [&](const WavePlan& pl) {   // the first kRing groups of a layer: they depend on nothing but the plan, so they are issued
#pragma unroll                            // before the observation is staged / before the previous layer's epilogue and barrier
    for (int j = 0; j < kRing; j++) {
      const int gj = pl.g0 + j < pl.g1 ? pl.g0 + j : pl.g0;
      r0[j] = ldfrag(pl.w0, gj * 64 + lane); r1[j] = ldfrag(pl.w1, gj * 64 + lane);
    }
  };
  WavePlan p = plan_layer(a, 0, wave, lane);
  {  // stage the observation tile: wave w takes rows 2w and 2w+1 (coalesced), permuted to the [k mod 4][k div 4] layout; the
     // k positions between K and the next multiple of 16 are zeroed (the packed weights are zero there, LDS garbage may be NaN).
     // The observation comes from HBM / the other XCDs' writes (>= 1 us) and is needed first: its loads go out BEFORE the weight ring
     // (vector-memory loads return in order), the ring fill is issued underneath them.
    const int K = a.dims[0], Kp = (K + 15) & ~15;
    float ov[2][kMaxDim / 64];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int rr = 2 * wave + h;
#pragma unroll
      for (int c = 0; c < kMaxDim / 64; c++) {
        const int kk = lane + 64 * c;
        ov[h][c] = (kk < K && row0 + rr < a.N) ? obs[(size_t)(row0 + rr) * K + kk] : 0.0f;
      }
    }
    fill(p);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int rr = 2 * wave + h;
#pragma unroll
      for (int c = 0; c < kMaxDim / 64; c++) {
        const int kk = lane + 64 * c;
        if (kk < Kp) act[0][rr * kActLd + act_pos(kk)] = ov[h][c];
      }
    }
  }
  lds_barrier();
  MLP_STAMP(1);
  for (int l = 0; l < a.n_layers; l++) {
    const int O = a.dims[l + 1], Op = (O + 15) & ~15;
    const bool last = l == a.n_layers - 1;
    const float* __restrict__ B = a.b[l];
    const float* x = act[l & 1] + r * kActLd + q * kQStride;     // this lane's A operand: row r, inputs k = 4 s + q
    float* y = act[(l + 1) & 1];
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    // biases of this wave's two column tiles: requested now, needed in the epilogue
    // unconditional (index clamped; columns >= O are masked where they are stored): a conditional load compiles to `v = 0; if (..) v = load`,
    // and the write of the 0 into a register that a load of the previous layer targeted costs an s_waitcnt vmcnt(0) at the loop head -
    // i.e. a wait for the whole weight ring that was prefetched across the barrier
    // In the transposed product a lane's four accumulator registers are four FEATURES (16 t + 4 q + reg) of ONE batch row (lane & 15).
    float bias0[4], bias1[4];
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      bias0[reg] = B[min(16 * p.t0 + 4 * q + reg, O - 1)];
      bias1[reg] = B[min(16 * p.t1 + 4 * q + reg, O - 1)];
    }
    // The layer as kRing-aligned steps: n real ones (8 MFMAs each), padded with MFMA-free steps to nv = 8 or 16. Every step refills the
    // ring slot it has just consumed: with group i + kRing of this layer while there is one, then with the NEXT layer's groups - the
    // padding makes next-layer group j land in slot j % kRing, where that layer's step j will look for it. So the first kRing groups
    // of a layer are requested during the previous layer's last kRing steps, not after its MFMAs (where the first step then sat out
    // an L2 round trip under load: ~2 us per layer boundary in the stamps), and the fill in front of layer 0 is the only other one.
    // Fused last layer: this is the layer before it and the handle found the shapes fit (nm_policy_create). Its weights for this wave's
    // two feature tiles (2 output tiles x 2 k-groups, packed in the accumulators' k order) are requested now; the loads are issued in
    // every layer (a hot line when there is nothing to fetch) so that the ring's wait counts stay compile-time constants.
    const bool fuse = a.fuse_last != 0 && l == a.n_layers - 2;
    f32x4 wl[2][2];
    {
      const int OL = a.dims[a.n_layers], ngL = (a.dims[a.n_layers - 1] + 15) >> 4, ntoL = (OL + 15) >> 4;
#pragma unroll
      for (int to = 0; to < 2; to++) {
        wl[to][0] = ldfrag(a.wt_last, fuse ? (min(to, ntoL - 1) * ngL + min(p.t0, ngL - 1)) * 64 + lane : lane);
        wl[to][1] = ldfrag(a.wt_last, fuse ? (min(to, ntoL - 1) * ngL + min(p.t1, ngL - 1)) * 64 + lane : lane);
      }
    }
    WavePlan pn = p;
    if (!last) pn = plan_layer(a, l + 1, wave, lane);
    {
      const int n = p.work ? p.g1 - p.g0 : 0;                    // k-groups of this wave in this layer: <= kMaxGroups
      const int nv = (last || fuse) ? n : (n + kRing - 1 > kRing ? ((n + kRing - 1) & ~(kRing - 1)) : kRing);   // the last computed layer prefetches for nobody
      const int nn = (!last && !fuse && pn.work) ? pn.g1 - pn.g0 : 0;
      f32x4 av = *reinterpret_cast<const f32x4*>(x + 4 * (n > 0 ? p.g0 : 0));
      unroll_while(std::make_integer_sequence<int, kMaxGroups>{}, [&](auto iT) {
        constexpr int i = decltype(iT)::value;
        if (i >= nv) return false;                               // forward exits only: the chain of executed steps is straight-line code
        if (i < n) {
          const int gs = p.g0 + i;
#ifdef NM_MLP_NOLDS
          const f32x4 an = av;
#else
          const f32x4 an = *reinterpret_cast<const f32x4*>(x + 4 * min(gs + 1, p.g1 - 1));           // activations: one group ahead
#endif
          const f32x4 c0 = r0[i % kRing], c1 = r1[i % kRing];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[j], av[j], acc0, 0, 0, 0);   // Y' = W X': weights are the A operand
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[j], av[j], acc1, 0, 0, 0);
          }
          av = an;
        }
#ifndef NM_MLP_NOLOAD
        {  // refill (always two loads, see above); every choice below is wave-uniform
          const int j = i + kRing - nv;                          // the next layer's group that belongs into this slot
          const bool mine = i + kRing < n, next = !mine && j >= 0 && j < nn;
          const f32x4* s0 = next ? pn.w0 : p.w0;
          const f32x4* s1 = next ? pn.w1 : p.w1;
          const int g = mine ? p.g0 + i + kRing : (next ? pn.g0 + j : 0);       // neither: the layer's first fragment, a hot line
          r0[i % kRing] = ldfrag(s0, g * 64 + lane);
          r1[i % kRing] = ldfrag(s1, g * 64 + lane);
        }
#endif
        return true;
      });
    }
    MLP_STAMP(2 + 3 * l);
    const WavePlan cur = p;
    p = pn;
    if (cur.split > 1) {   // reduce the k-parts: both accumulators parked at once, the first wave of a pair adds them up
      if (cur.work && cur.part != 0) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) { red[wave][reg][lane] = acc0[reg]; red[wave][4 + reg][lane] = acc1[reg]; }
      }
      lds_barrier();
      if (cur.work && cur.part == 0) {
        for (int pp = 1; pp < cur.split; pp++)
#pragma unroll
          for (int reg = 0; reg < 4; reg++) { acc0[reg] += red[wave + pp][reg][lane]; acc1[reg] += red[wave + pp][4 + reg][lane]; }
      }
    }
    MLP_STAMP(3 + 3 * l);
    // epilogue. C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg; with Y' = W X' the column is the batch row and the
    // row the output feature: lane (r, q) holds features 16 t + 4 q + reg of batch row r - which is exactly what the B operand of the
    // next layer's k-steps wants (see the fused last layer below), and 16 contiguous bytes of the output row.
    if (fuse) {
      // ---- the last layer from registers: Y2' = W2 Y1'. Lane (r, q) holds, for each of its two feature tiles t, the features
      // 16 t + 4 q + reg of batch row r after bias + ELU - as a B operand that is k-step `reg` of k-group t in the packing of wt_last.
      // A wave contributes the partial sums over ITS features; the eight waves' partials are added through LDS (split-k by wave).
      const int OL = a.dims[a.n_layers], ntoL = (OL + 15) >> 4;
      f32x4 a2[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      if (cur.work) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
          if (half == 0 || cur.two) {
            const int tile = half ? cur.t1 : cur.t0;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
              const int f = 16 * tile + 4 * q + reg;
              float v = (half ? acc1[reg] : acc0[reg]) + (half ? bias1[reg] : bias0[reg]);
              v = v > 0.0f ? v : __expf(v) - 1.0f;
              v = f < O ? v : 0.0f;
              a2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[0][half][reg], v, a2[0], 0, 0, 0);
              a2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[1][half][reg], v, a2[1], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int reg = 0; reg < 4; reg++) { red[wave][reg][lane] = a2[0][reg]; red[wave][4 + reg][lane] = a2[1][reg]; }
      lds_barrier();
      if (wave < ntoL) {                 // wave `to` finishes output tile `to`: sum of the eight partials, bias, store
        const float* __restrict__ BL = a.b[a.n_layers - 1];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          float v = 0.0f;
#pragma unroll
          for (int ww = 0; ww < kMlpWaves; ww++) v += red[ww][4 * wave + reg][lane];
          const int f = 16 * wave + 4 * q + reg;
          if (f < OL && row0 + r < a.N) out[(size_t)(row0 + r) * OL + f] = v + BL[f];
        }
      }
      return;
    }
    if (cur.work && cur.part == 0) {
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int tile = half ? cur.t1 : cur.t0;
        if (half == 0 || cur.two) {
#pragma unroll
          for (int reg = 0; reg < 4; reg++) {
            const int f = 16 * tile + 4 * q + reg;
            float v = (half ? acc1[reg] : acc0[reg]) + (half ? bias1[reg] : bias0[reg]);
            if (!last) {
              v = v > 0.0f ? v : __expf(v) - 1.0f;   // ELU; v_exp_f32 (1 ulp) - 1: absolute error < 1.2e-7, no libm call in the epilogue
              y[r * kActLd + reg * kQStride + (4 * tile + q)] = f < O ? v : 0.0f;   // = act_pos(f); features up to the next multiple of 16 feed zero weights: keep them finite
            } else if (f < O && row0 + r < a.N) {
              out[(size_t)(row0 + r) * O + f] = v;
            }
          }
        }
      }
    }
    lds_barrier();
    MLP_STAMP(4 + 3 * l);
  }
}

// y[N,O] = act(x[N,K] W[O,K]^T + b[O]);  grid = (ceil(N/32), ceil(O/32)), block = 64   (fallback for layers wider than 256)
__global__ void __launch_bounds__(64) k_linear_mfma(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
                                                    float* __restrict__ y, int N, int K, int O, int elu) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
  const int ar = row0 + r, bc = col0 + r;
  const bool aok = ar < N, bok = bc < O;
  const float* xa = x + (size_t)(aok ? ar : 0) * K;
  const float* wb = W + (size_t)(bok ? bc : 0) * K;
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 2) {
    const int k = k0 + h;
    float a = (aok && k < K) ? xa[k] : 0.0f;
    float w = (bok && k < K) ? wb[k] : 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w, acc, 0, 0, 0);
  }
  const int col = col0 + r;
  if (col < O) {
    const float bias = b[col];
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      int row = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (row < N) {
        float v = acc[reg] + bias;
        if (elu) v = v > 0.0f ? v : expm1f(v);
        y[(size_t)row * O + col] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ handle
struct nm_policy {
  int device = 0, n_layers = 0;
  std::vector<int> dims;
  bool fused = false, loaded = false;
  f32x4* packed = nullptr;            // fused path: packed weights of all layers
  f32x4* packed_t = nullptr;          // the last layer in the accumulator k order (fuse_last)
  bool fuse_last = false;
  float* bias = nullptr;              // own copy of the biases (the handle never points into caller memory)
  float* wcopy = nullptr;             // per-layer path: own copy of the weights, torch layout
  std::vector<size_t> w_off, b_off, p_off;
  float* scratch[2] = {nullptr, nullptr};
  size_t scratch_n = 0;
  MlpArgs args;
};

static bool mlp_fits(const int32_t* dims, int32_t n_layers) {
  bool fits = n_layers <= kMaxLayers;
  for (int l = 0; l <= n_layers && fits; l++) fits = dims[l] > 0 && dims[l] <= kMaxDim;
  return fits;
}

extern "C" int nm_policy_create(const int32_t* dims, int32_t n_layers, int32_t device, nm_policy** out) {
  if (!out) return nm_policy_set_error("nm_policy_create: out is NULL");
  *out = nullptr;
  if (!dims || n_layers <= 0) return nm_policy_set_error("nm_policy_create: bad argument");
  for (int l = 0; l <= n_layers; l++)
    if (dims[l] <= 0) return nm_policy_set_error("nm_policy_create: layer sizes must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return nm_policy_set_error("nm_policy_create: no such HIP device");
  if (hipSetDevice(device) != hipSuccess) return nm_policy_set_error("nm_policy_create: hipSetDevice failed");
  nm_policy* h = new nm_policy();
  h->device = device; h->n_layers = n_layers;
  h->dims.assign(dims, dims + n_layers + 1);
  h->fused = mlp_fits(dims, n_layers);
  size_t wtot = 0, btot = 0, ptot = 0;
  for (int l = 0; l < n_layers; l++) {
    h->w_off.push_back(wtot); h->b_off.push_back(btot); h->p_off.push_back(ptot);
    wtot += (size_t)dims[l] * dims[l + 1];
    btot += (size_t)dims[l + 1];
    ptot += (size_t)((dims[l + 1] + 15) / 16) * ((dims[l] + 15) / 16) * 64;
  }
  bool ok = hipMalloc((void**)&h->bias, btot * sizeof(float)) == hipSuccess;
  if (ok && h->fused) ok = hipMalloc((void**)&h->packed, ptot * sizeof(f32x4)) == hipSuccess;
  // The last layer can take the previous layer's accumulators straight from the registers when it is narrow (<= 32 outputs: two
  // accumulator tiles) and the previous layer is not split over k (every wave then holds finished features): no LDS round trip, no
  // barrier, no separate phase for it. The split rule is plan_layer's.
  if (ok && h->fused && n_layers >= 2 && dims[n_layers] <= 32) {
    const int Kp = dims[n_layers - 2], Op = dims[n_layers - 1];
    const int npair = ((Op + 15) / 16 + 1) / 2, ngrp = (Kp + 15) / 16;
    const bool split = npair * 2 <= kMlpWaves && ngrp / 2 >= 4;
    h->fuse_last = !split;
    if (h->fuse_last) ok = hipMalloc((void**)&h->packed_t, (size_t)((dims[n_layers] + 15) / 16) * ((Op + 15) / 16) * 64 * sizeof(f32x4)) == hipSuccess;
  }
  if (ok && !h->fused) ok = hipMalloc((void**)&h->wcopy, wtot * sizeof(float)) == hipSuccess;
  if (!ok) { nm_policy_destroy(h); return nm_policy_set_error("nm_policy_create: hipMalloc failed"); }
  *out = h;
  return 0;
}

extern "C" int nm_policy_destroy(nm_policy* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  if (h->packed) (void)hipFree(h->packed);
  if (h->packed_t) (void)hipFree(h->packed_t);
  if (h->bias) (void)hipFree(h->bias);
  if (h->wcopy) (void)hipFree(h->wcopy);
  for (int i = 0; i < 2; i++) if (h->scratch[i]) (void)hipFree(h->scratch[i]);
  delete h;
  return 0;
}

// copy (and, on the fused path, repack) the parameters: call after every optimiser step / load_state_dict. Stream-ordered.
extern "C" int nm_policy_load(nm_policy* h, const float* const* weights, const float* const* bias, void* stream) {
  if (!h || !weights || !bias) return nm_policy_set_error("nm_policy_load: bad argument");
  if (hipSetDevice(h->device) != hipSuccess) return nm_policy_set_error("nm_policy_load: hipSetDevice failed");
  hipStream_t s = (hipStream_t)stream;
  for (int l = 0; l < h->n_layers; l++) {
    const int K = h->dims[l], O = h->dims[l + 1];
    if (!weights[l] || !bias[l]) return nm_policy_set_error("nm_policy_load: NULL parameter pointer");
    if (hipMemcpyAsync(h->bias + h->b_off[l], bias[l], (size_t)O * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
      return nm_policy_set_error("nm_policy_load: bias copy failed");
    if (h->fused) {
      const int ntile = (O + 15) / 16, ngrp = (K + 15) / 16, n = ntile * ngrp * 64;
      hipLaunchKernelGGL(k_pack_weights, dim3((n + 255) / 256), dim3(256), 0, s, weights[l], h->packed + h->p_off[l], O, K, ngrp, ntile, 0);
      h->args.w[l] = h->packed + h->p_off[l];
      if (l == h->n_layers - 1 && h->fuse_last) {
        hipLaunchKernelGGL(k_pack_weights, dim3((n + 255) / 256), dim3(256), 0, s, weights[l], h->packed_t, O, K, ngrp, ntile, 1);
        h->args.wt_last = h->packed_t;
      }
    } else if (hipMemcpyAsync(h->wcopy + h->w_off[l], weights[l], (size_t)O * K * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
      return nm_policy_set_error("nm_policy_load: weight copy failed");
    }
    h->args.b[l] = h->bias + h->b_off[l];
  }
  if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_policy_load: launch failed");
  for (int l = 0; l <= h->n_layers && l <= kMaxLayers; l++) h->args.dims[l] = h->dims[l];
  h->args.n_layers = h->n_layers;
  h->args.fuse_last = h->fuse_last ? 1 : 0;
  if (!h->fuse_last) h->args.wt_last = h->packed;
  h->loaded = true;
  return 0;
}

extern "C" int nm_policy_forward(nm_policy* h, const float* obs, int32_t N, float* out, void* stream) {
  if (!h || !obs || !out || N <= 0) return nm_policy_set_error("nm_policy_forward: bad argument");
  if (!h->loaded) return nm_policy_set_error("nm_policy_forward: no parameters loaded (nm_policy_load)");
  if (hipSetDevice(h->device) != hipSuccess) return nm_policy_set_error("nm_policy_forward: hipSetDevice failed");
  hipStream_t s = (hipStream_t)stream;
  if (h->fused) {
    MlpArgs a = h->args;
    a.N = N;
    hipLaunchKernelGGL(k_mlp_fused, dim3((N + kTileRows - 1) / kTileRows), dim3(kMlpThreads), 0, s, obs, out, a);
    if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_policy_forward: launch failed");
    return 0;
  }
  size_t maxh = 1;
  for (int l = 1; l < h->n_layers; l++) maxh = (size_t)h->dims[l] > maxh ? (size_t)h->dims[l] : maxh;
  const size_t need = (size_t)N * maxh;
  if (h->n_layers > 1 && need > h->scratch_n) {
    for (int i = 0; i < 2; i++) {
      if (h->scratch[i]) (void)hipFree(h->scratch[i]);
      h->scratch[i] = nullptr;
      if (hipMalloc((void**)&h->scratch[i], need * sizeof(float)) != hipSuccess) return nm_policy_set_error("nm_policy_forward: hipMalloc failed");
    }
    h->scratch_n = need;
  }
  const float* in = obs;
  for (int l = 0; l < h->n_layers; l++) {
    const bool last = l == h->n_layers - 1;
    float* o = last ? out : h->scratch[l & 1];
    dim3 grid((N + 31) / 32, (h->dims[l + 1] + 31) / 32);
    hipLaunchKernelGGL(k_linear_mfma, grid, dim3(64), 0, s, in, h->wcopy + h->w_off[l], h->bias + h->b_off[l], o, N, h->dims[l], h->dims[l + 1],
                       last ? 0 : 1);
    if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_policy_forward: launch failed");
    in = o;
  }
  return 0;
}
