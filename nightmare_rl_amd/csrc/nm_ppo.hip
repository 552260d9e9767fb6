// nm_ppo.hip - PPO mini-batch update of rsl_rl v1.0.2 (`algorithms/ppo.py` PPO.update; caller reference train.py:54) on hand-written
// kernels: forward of actor and critic, the clipped-surrogate / clipped-value / entropy losses, the whole backward pass, gradient-norm
// clipping, the KL-adaptive learning rate and Adam - TWO launches per mini-batch (k_ppo_fwdbwd / k_ppo_fwdbwd_split, k_ppo_step), no host synchronisation, instead of ~60 framework
// launches (elementwise chains on [81920 x 54] activations + library GEMMs with K = 82 k).
//
// The two MLPs (reference envs/nightmare_v3_config.py:107-109: 66 -> 54 -> 42 -> 30 -> 18 and -> 1, ELU) run as ONE merged network:
// actor and critic side by side, block-diagonal hidden layers, the bias of a layer stored as one more weight column fed by a constant
// 1 - so forward, dX and dW are three plain GEMM shapes per layer, all on exact-f32 MFMA (v_mfma_f32_16x16x4_f32):
//     forward   a_l  = ELU(a_{l-1} Wm_l')              A = activations [16 rows x K],  B = packed Wm_l   (k-major per lane)
//     dX        d_{l-1} = (d_l Wm_l) * ELU'(a_{l-1})   A = deltas      [16 rows x O],  B = packed Wm_l'  (o-major per lane)
//     dW        G_l += d_l' a_{l-1}                    A = deltas^T, B = activations, reduction over the 16 rows of the tile
// k_ppo_fwdbwd: a workgroup of 8 waves walks its share of the mini-batch in tiles of 16 rows; activations and deltas live in LDS
//   ([row][k mod 4][k div 4]: the A operand of 4 k-steps is one ds_read_b128); the dW accumulators stay in registers for the whole
//   walk (<= 16 tiles of 16x16 per wave) and are written once per workgroup as a partial gradient. The loss head (one wave) turns the
//   network output into d(loss)/d(mean), d(loss)/d(value), and accumulates d(loss)/d(std), the KL to the behaviour policy and the
//   loss values.
// k_ppo_step   : the rest of a mini-batch in one launch with one grid barrier - partial gradients -> gradient of every real parameter,
//   gradient norm -> clip coefficient, KL mean -> learning rate (rsl_rl's adaptive schedule), Adam, the two packed copies of the weights.
// k_ppo_reduce / k_ppo_scalars / k_ppo_adam / k_ppo_pack: the same as four launches (data-parallel phase 1 uses k_ppo_reduce alone; the
//   whole chain is the NM_PPO_UNFUSED_STEP=1 path for GPUs shared between processes).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <utility>
#include <vector>

#include "../../include/nightmare_hip.h"

extern "C" int nm_policy_set_error(const char* m);

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kL = 4;                      // layers of the merged network (<=)
constexpr int kW = 128;                    // widest padded layer
constexpr int kQS = kW / 4 + 4;            // stride between the four (k mod 4) planes of a row
constexpr int kLD = 4 * kQS + 4;           // floats per LDS row
constexpr int kRows = 16;
constexpr int kThreads = 512, kWaves = 8;
constexpr int kSlots = 16;                 // dW tiles a wave can own
constexpr int kMaxA = 32;                  // actions (<=)
constexpr int kNS = 40;                    // per-workgroup scalars: dstd[32], kl, surrogate, value loss, count, pad

struct PpoNet {
  int n_layers;
  int Kr[kL], Or[kL];          // real inputs (without the bias column) / outputs of the merged layers
  int Kp[kL], Op[kL];          // padded to 16 (Kp includes the bias column)
  int goff[kL];                // offset of layer l's [Op x Kp] block in a merged gradient vector
  int gtotal;                  // floats per merged gradient (without the scalars)
  int A;                       // actions; the critic value is output column A
  const f32x4* pf[kL];         // forward packing   [O tile][k group][lane]
  const f32x4* pb[kL];         // backward packing  [K tile][o group][lane]
  int slot[kWaves][kSlots];    // dW tiles of wave w: layer | o tile << 4 | k tile << 8, -1 = none (tile g of the network goes to wave g % 8)
  int slot4[4][16];                    // split kernel (k_ppo_fwdbwd_split): dW tiles of row-group wave g of either net, layer by layer: o tile | k tile << 4, -1 = none
  int woff[2][kL], boff[2][kL];        // split kernel: flat-parameter index of W[0][0] / b[0] of net (actor, critic), layer l; nparam_flat: all of them
  int nparam_flat;
  const f32x4* sf[2];                  // split kernel: per net (actor, critic), forward fragments in consumption order
  const f32x4* sb[2];                  // ... and the dX fragments
};
struct PpoBatch {
  const float *obs, *actions, *old_mu, *old_sigma, *old_logp, *adv, *ret, *tval, *std;
  int B, n_obs;
  float clip, vcoef, inv_B;
  int clip_value;
  const int* rows;             // null: the mini-batch is rows 0..B-1 of the arrays; else row i of the mini-batch is row rows[i] (gather in the kernel)
};

__device__ __forceinline__ int pos(int c) { return (c & 3) * kQS + (c >> 2); }

__global__ void __launch_bounds__(kThreads) k_ppo_fwdbwd(PpoNet net, PpoBatch bt, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float act[kL][kRows * kLD];      // inputs of layer l (a_0 = observation), with the 1-column
  __shared__ __attribute__((aligned(16))) float outb[kRows * kLD];         // network output: means | value
  __shared__ __attribute__((aligned(16))) float dl[2][kRows * kLD];        // deltas, ping-pong
  __shared__ float hs[kRows][8];                                           // per-row scalars of the loss head
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int ntiles = (bt.B + kRows - 1) / kRows;
  // ---- this wave's dW tiles (table built on the host)
  int s_l[kSlots], s_to[kSlots], s_tk[kSlots];
#pragma unroll
  for (int s = 0; s < kSlots; s++) {
    const int code = net.slot[wave][s];
    s_l[s] = code < 0 ? -1 : (code & 15); s_to[s] = (code >> 4) & 15; s_tk[s] = (code >> 8) & 15;
  }
  f32x4 gw[kSlots];
#pragma unroll
  for (int s = 0; s < kSlots; s++) gw[s] = f32x4{0, 0, 0, 0};
  float a_dstd[kMaxA / 4], a_kl = 0.0f, a_surr = 0.0f, a_vl = 0.0f;       // head accumulators (wave 0; lane-local partials)
#pragma unroll
  for (int it = 0; it < kMaxA / 4; it++) a_dstd[it] = 0.0f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * kRows;
    // ---- stage the observation tile (+ the constant 1 that carries the bias); wave w takes rows 2w, 2w+1
    {
      const int K = net.Kr[0], Kp = net.Kp[0];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int rr = 2 * wave + h, row = row0 + rr;
        const size_t srow = row < bt.B ? (bt.rows ? (size_t)bt.rows[row] : (size_t)row) : 0;
        for (int kk = lane; kk < Kp; kk += 64) {
          float v = 0.0f;
          if (row < bt.B) v = kk < K ? bt.obs[srow * bt.n_obs + kk] : (kk == K ? 1.0f : 0.0f);
          act[0][rr * kLD + pos(kk)] = v;
        }
      }
    }
    __syncthreads();
    // ---- forward
    for (int l = 0; l < net.n_layers; l++) {
      const bool last = l == net.n_layers - 1;
      const int ngrp = net.Kp[l] >> 4, ntile = net.Op[l] >> 4, Or = net.Or[l];
      const float* x = act[l] + r * kLD + q * kQS;
      float* y = last ? outb : act[l + 1];
      for (int t = wave; t < ntile; t += kWaves) {
        const f32x4* __restrict__ w = net.pf[l] + ((size_t)t * ngrp) * 64 + lane;
        f32x4 acc = {0, 0, 0, 0};
        for (int g = 0; g < ngrp; g++) {
          const f32x4 wv = w[(size_t)g * 64];
          const f32x4 av = *reinterpret_cast<const f32x4*>(x + 4 * g);
#pragma unroll
          for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[j], acc, 0, 0, 0);
        }
        const int col = 16 * t + r;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const int rr = q * 4 + reg;
          float v = acc[reg];
          if (!last) v = col < Or ? (v > 0.0f ? v : __expf(v) - 1.0f) : (col == Or ? 1.0f : 0.0f);   // ELU | bias carrier | padding
          y[rr * kLD + pos(col)] = v;
        }
      }
      __syncthreads();
    }
    // ---- loss head (wave 0): lane (row r, q) handles actions q, q+4, ...; d(loss)/d(output) -> dl[0]
    if (wave == 0) {
      const int row = row0 + r, A = net.A;
      const bool live = row < bt.B;
      const size_t srow = live ? (bt.rows ? (size_t)bt.rows[row] : (size_t)row) : 0;
      float lp = 0.0f, kl = 0.0f;
      for (int j = q; j < A; j += 4) {
        if (live) {
          const float mu = outb[r * kLD + pos(j)], sd = bt.std[j];
          const float a = bt.actions[srow * A + j], omu = bt.old_mu[srow * A + j], osd = bt.old_sigma[srow * A + j];
          const float z = (a - mu) / sd;
          lp += -0.5f * z * z - __logf(sd) - 0.9189385332046727f;
          kl += __logf(sd / osd + 1.0e-5f) + (osd * osd + (omu - mu) * (omu - mu)) / (2.0f * sd * sd) - 0.5f;
        }
      }
      lp += __shfl_xor(lp, 16); lp += __shfl_xor(lp, 32);
      kl += __shfl_xor(kl, 16); kl += __shfl_xor(kl, 32);
      float dlogp = 0.0f;      // d(surrogate)/d(logp) * (1/B)
      if (live) {
        const float adv = bt.adv[srow], ratio = __expf(lp - bt.old_logp[srow]);
        const float s1 = -adv * ratio, rc = fminf(fmaxf(ratio, 1.0f - bt.clip), 1.0f + bt.clip), s2 = -adv * rc;
        const bool inside = ratio > 1.0f - bt.clip && ratio < 1.0f + bt.clip;
        // torch.max(s1, s2).backward(): the larger branch gets the gradient, a tie (ratio inside the clip range) splits it between
        // two identical branches; the clipped branch has zero slope outside the range
        const float g1 = s1 > s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f), g2 = 1.0f - g1;
        dlogp = (g1 * (-adv * ratio) + g2 * (inside ? -adv * ratio : 0.0f)) * bt.inv_B;
        if (q == 0) {
          a_surr += fmaxf(s1, s2);
          a_kl += kl;
          const float v = outb[r * kLD + pos(A)], R = bt.ret[srow], tv = bt.tval[srow];
          float dv, vl;
          if (bt.clip_value) {
            const float dvt = v - tv, vc = tv + fminf(fmaxf(dvt, -bt.clip), bt.clip);
            const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
            const bool in2 = dvt > -bt.clip && dvt < bt.clip;
            const float h1 = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f), h2 = 1.0f - h1;
            vl = fmaxf(l1, l2);
            dv = h1 * 2.0f * (v - R) + h2 * (in2 ? 2.0f * (vc - R) : 0.0f);
          } else {
            vl = (R - v) * (R - v);
            dv = 2.0f * (v - R);
          }
          a_vl += vl;
          hs[r][0] = dv * bt.vcoef * bt.inv_B;
        }
      }
      __builtin_amdgcn_wave_barrier();
      // deltas of the output layer: columns j < A: dlogp * (a - mu)/sd^2, column A: value, the rest 0; and this tile's part of
      // d(loss)/d(std_j) = sum over rows of dlogp * ((a-mu)^2/sd^3 - 1/sd): lane (r, q) owns j = q + 4 it, rows summed over the 16-lane row
#pragma unroll
      for (int it = 0; it < kMaxA / 4 + 1; it++) {
        const int c = q + 4 * it;
        if (c < net.Op[net.n_layers - 1]) {
          float d = 0.0f, ds = 0.0f;
          if (live && c < A) {
            const float mu = outb[r * kLD + pos(c)], sd = bt.std[c], a = bt.actions[srow * A + c];
            d = dlogp * (a - mu) / (sd * sd);
            ds = dlogp * ((a - mu) * (a - mu) / (sd * sd * sd) - 1.0f / sd);
          } else if (live && c == A) {
            d = hs[r][0];
          }
          dl[0][r * kLD + pos(c)] = d;
          ds += __shfl_xor(ds, 1); ds += __shfl_xor(ds, 2); ds += __shfl_xor(ds, 4); ds += __shfl_xor(ds, 8);
          if (it < kMaxA / 4) a_dstd[it] += ds;
        }
      }
    }
    __syncthreads();
    // ---- backward
    for (int l = net.n_layers - 1; l >= 0; l--) {
      const float* D = dl[(net.n_layers - 1 - l) & 1];
      float* Dn = dl[(net.n_layers - l) & 1];
      // dW_l += D' a_{l-1}: this wave's tiles of layer l
#pragma unroll
      for (int s = 0; s < kSlots; s++) {
        if (s_l[s] == l) {
          const int co = 16 * s_to[s] + r, ck = 16 * s_tk[s] + r;
          const float* dp = D + q * kLD + pos(co);
          const float* ap = act[l] + q * kLD + pos(ck);
#pragma unroll
          for (int st = 0; st < 4; st++)
            gw[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(dp[4 * st * kLD], ap[4 * st * kLD], gw[s], 0, 0, 0);
        }
      }
      // d_{l-1} = (D Wm_l) * ELU'(a_{l-1})
      if (l > 0) {
        const int ngrp = net.Op[l] >> 4, ntile = net.Kp[l] >> 4, Kr = net.Kr[l];
        const float* x = D + r * kLD + q * kQS;
        for (int t = wave; t < ntile; t += kWaves) {
          const f32x4* __restrict__ w = net.pb[l] + ((size_t)t * ngrp) * 64 + lane;
          f32x4 acc = {0, 0, 0, 0};
          for (int g = 0; g < ngrp; g++) {
            const f32x4 wv = w[(size_t)g * 64];
            const f32x4 av = *reinterpret_cast<const f32x4*>(x + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[j], acc, 0, 0, 0);
          }
          const int col = 16 * t + r;
#pragma unroll
          for (int reg = 0; reg < 4; reg++) {
            const int rr = q * 4 + reg;
            const float a = act[l][rr * kLD + pos(col)];
            Dn[rr * kLD + pos(col)] = col < Kr ? acc[reg] * (a > 0.0f ? 1.0f : a + 1.0f) : 0.0f;   // ELU'(z) from ELU(z); no gradient into the 1-column
          }
        }
      }
      __syncthreads();
    }
  }
  // ---- this workgroup's partial gradient
  float* P = partial + (size_t)blockIdx.x * (net.gtotal + kNS);
#pragma unroll
  for (int s = 0; s < kSlots; s++) {
    if (s_l[s] >= 0) {
      const int l = s_l[s], Kp = net.Kp[l];
      const int k = 16 * s_tk[s] + r;
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int o = 16 * s_to[s] + q * 4 + reg;
        P[net.goff[l] + o * Kp + k] = gw[s][reg];
      }
    }
  }
  if (wave == 0) {
    a_kl += __shfl_xor(a_kl, 1); a_kl += __shfl_xor(a_kl, 2); a_kl += __shfl_xor(a_kl, 4); a_kl += __shfl_xor(a_kl, 8);
    a_surr += __shfl_xor(a_surr, 1); a_surr += __shfl_xor(a_surr, 2); a_surr += __shfl_xor(a_surr, 4); a_surr += __shfl_xor(a_surr, 8);
    a_vl += __shfl_xor(a_vl, 1); a_vl += __shfl_xor(a_vl, 2); a_vl += __shfl_xor(a_vl, 4); a_vl += __shfl_xor(a_vl, 8);
    if (r == 0) {
#pragma unroll
      for (int it = 0; it < kMaxA / 4; it++) P[net.gtotal + q + 4 * it] = a_dstd[it];
    }
    if (lane == 0) { P[net.gtotal + 32] = a_kl; P[net.gtotal + 33] = a_surr; P[net.gtotal + 34] = a_vl; }
  }
}


// ------------------------------------------------------------------------------------------------ fast path
// For the reference's network shape (three hidden layers, known at compile time) the mini-batch runs through k_ppo_fwdbwd_split and
// the collection step through k_ppo_act_fast. Common to both:
//   * a wave owns 16 rows from observation to deltas; activations and deltas never leave its registers. Every product is computed
//     transposed, Y' = W X': the weights are the A operand (straight from L2, 16 B per lane per 4 MFMAs), the activations the B
//     operand, and the result lands as lane (row r, q) <- features 16 t + 4 q + reg of tile t - exactly the B-operand layout of the
//     next layer (feature order inside a k-step is free as long as the packed weights use the same one) and of dX = W' D'.
//     No LDS, no barrier, no cross-wave dependency in forward, loss head and dX.
//   * only dW needs the batch on the k axis: per layer the four waves of a workgroup park their transposed delta / activation tiles
//     in LDS (conflict-free b32 writes, b128 reads), one barrier, then each wave accumulates its share of the layer's dW tiles over
//     the workgroup's 64 rows; two buffers alternate between layers, so that barrier is the only one.
// (Round 4 replaced the merged four-wave kernel of rounds 2-3 - one 512-register wave per SIMD, 146 us per 81 920-row mini-batch - by
// the per-net kernel below: 133 us. k_ppo_act_fast still runs the merged network: Shape4::nz skips its actor x critic tiles.)
// Workgroup barrier for data exchanged through LDS only: wait for this wave's LDS traffic, then s_barrier. __syncthreads() also drains
// the vector-memory queue (a workgroup-scope release has to assume global memory): at the layer-1 barrier that is the next pass's rows
// (an HBM gather) and at every barrier the weight fragments prefetched across it - the ring would be waited for, not run under.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N, class F, int... Is> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <int I_, int A1, int A2, int A3, int AO, int C1, int C2, int C3>
struct Shape4 {
  static constexpr int NL = 4;
  static constexpr __host__ __device__ int ain(int l) { return l == 0 ? I_ : l == 1 ? A1 : l == 2 ? A2 : A3; }
  static constexpr __host__ __device__ int cin(int l) { return l == 0 ? I_ : l == 1 ? C1 : l == 2 ? C2 : C3; }
  static constexpr __host__ __device__ int aout(int l) { return l == 0 ? A1 : l == 1 ? A2 : l == 2 ? A3 : AO; }
  static constexpr __host__ __device__ int cout(int l) { return l == 0 ? C1 : l == 1 ? C2 : l == 2 ? C3 : 1; }
  static constexpr __host__ __device__ int Kr(int l) { return l == 0 ? I_ : ain(l) + cin(l); }
  static constexpr __host__ __device__ int Or(int l) { return aout(l) + cout(l); }
  static constexpr __host__ __device__ int up16(int x) { return (x + 15) & ~15; }
  // padded width of the input of layer l (l = NL: of the network output) - the rule of nm_ppo_create
  static constexpr __host__ __device__ int P(int l) { return l == NL ? up16(Or(NL - 1)) : up16(Kr(l) + 1); }
  static constexpr __host__ __device__ int maxT() { int m = 0; for (int l = 0; l <= NL; l++) m = P(l) / 16 > m ? P(l) / 16 : m; return m; }
  // does tile (o tile, k tile) of merged layer l hold a parameter? (positions as nm_ppo_create's map)
  static constexpr __host__ __device__ bool nz(int l, int to, int tk) {
    const int o0 = 16 * to, o1 = o0 + 16, k0 = 16 * tk, k1 = k0 + 16;
    const int cc0 = l == 0 ? 0 : ain(l), cc1 = cc0 + cin(l);
    const bool actor = o0 < aout(l) && k0 < ain(l);
    const bool critic = o0 < Or(l) && o1 > aout(l) && k0 < cc1 && k1 > cc0;
    const bool bias = o0 < Or(l) && k0 <= Kr(l) && Kr(l) < k1;
    return actor || critic || bias;
  }
  static constexpr __host__ __device__ int ntiles(int l) { int n = 0; for (int to = 0; to < P(l + 1) / 16; to++) for (int tk = 0; tk < P(l) / 16; tk++) n += nz(l, to, tk); return n; }
};
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
// -DNM_PPO_STAMPS (measurement builds only): lane 0 of wave 0 adds the s_memtime ticks since its previous stamp to g_ppo_stamps[k]
#ifdef NM_PPO_STAMPS
__device__ unsigned long long g_ppo_stamps[16];
#define PPO_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
                          if (tid == 0) atomicAdd(&g_ppo_stamps[k], n_ - t_last); t_last = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PPO_STAMP(k)
#endif

// Order in which a wave consumes the packed weight fragments (one fragment = the A operand of 4 MFMAs), so that they can stream
// through a ring of registers `kRing` fragments ahead of their use, across layer boundaries (the compiler left to itself issued
// load -> wait -> 4 MFMAs, one L2 round trip per fragment):
//   forward: layer 0..NL-1 | pairs of output tiles (2p, 2p+1) | k tile ascending | even, odd output tile
//   dX     : layer NL-1..1 | pairs of k tiles (2p, 2p+1)      | o tile ascending | even, odd k tile
// The two output tiles of a pair are independent accumulation chains that share the B operand: their MFMAs alternate, which also
// covers the 40-cycle dependent-accumulator latency of v_mfma_f32_16x16x4_f32 (issue 32).
template <class S> struct Seq {
  static constexpr int NL = S::NL;
  // mode 0: number of fragments; mode 1: entry number `arg` as l | to << 4 | tk << 8; mode 2: position of (l, to, tk) packed in arg
  static constexpr __host__ __device__ int walk(bool fwd, int mode, int arg) {
    int i = 0;
    for (int step = 0; step < (fwd ? NL : NL - 1); step++) {
      const int l = fwd ? step : NL - 1 - step;
      const int nto = S::P(l + 1) / 16, nkt = S::P(l) / 16;
      const int nouter = fwd ? nto : nkt, ninner = fwd ? nkt : nto;
      for (int p = 0; p < (nouter + 1) / 2; p++)
        for (int in = 0; in < ninner; in++)
          for (int h = 0; h < 2; h++) {
            const int o = 2 * p + h;
            if (o >= nouter) continue;
            const int to = fwd ? o : in, tk = fwd ? in : o;
            if (!S::nz(l, to, tk)) continue;
            const int e = l | (to << 4) | (tk << 8);
            if (mode == 1 && i == arg) return e;
            if (mode == 2 && e == arg) return i;
            i++;
          }
    }
    return mode == 0 ? i : -1;
  }
  static constexpr __host__ __device__ int count(bool fwd) { return walk(fwd, 0, 0); }
  static constexpr __host__ __device__ int entry(bool fwd, int i) { return walk(fwd, 1, i); }
  static constexpr __host__ __device__ int index(bool fwd, int l, int to, int tk) { return walk(fwd, 2, l | (to << 4) | (tk << 8)); }
};
constexpr int kRing = 12;     // weight fragments in flight in k_ppo_act_fast

// k_ppo_fwdbwd_split: the same mini-batch pass with TWO waves per SIMD. The merged kernel of rounds 2-3 kept one 512-register wave per SIMD, so every
// stall of that wave - the ELU / loss-head VALU between MFMA chains, the LDS round trips of the dW operands, barriers, the weight ring's
// L2 latency - is idle matrix-pipe time (measured: 50 % MFMA busy, 2x the MFMA floor). Giving each SIMD a second wave needs the per-wave
// state to fit 256 registers; rows cannot be split further (an MFMA tile is 16 rows), but the NETWORK can: actor and critic are independent
// chains that share only the observation. A workgroup is 4 waves = 4 row groups of 16 rows of ONE net (even blocks: actor, odd blocks:
// critic); it runs that net's forward, its part of the loss head, dX and dW - half the activations, deltas, accumulators and weight
// fragments of a merged wave, the same total MFMA work per row (no block-diagonal padding tiles at all) - and TWO workgroups share a CU
// (80 KB of LDS and 256 registers each). They are independent: no common barrier, so they drift out of phase and one's stalls are the
// other's matrix-pipe time. (The same split inside ONE eight-wave workgroup was measured first: its common barriers keep the two waves
// of a SIMD in lock step - both in their MFMA phases, then both in their waits - and it gained nothing: 45 % MFMA busy.)
// Both nets have the same tile shape (the critic's last layer is padded to the actor's), so both roles run ONE code path on different
// weights; only the loss head branches on the role (workgroup-uniform).
//   * weights: per net, fragments stored in the order the wave consumes them (nm_ppo_create: sfi / sbi), so the stream is linear;
//   * exchange buffers: per layer the net's delta and input tiles of the 4 row groups; two buffers alternate between layers as in the
//     four-wave kernel: one barrier per layer;
//   * the partial gradient of blocks 2 v and 2 v + 1 goes to partial row v at the merged layout's positions (the two nets' entries are
//     disjoint), so k_ppo_step / k_ppo_reduce and the parameter map do not change.
template <class S> struct Split {
  static constexpr int NL = S::NL, NG = 4;                     // row groups (waves per net)
  static constexpr __host__ __device__ int up16(int x) { return (x + 15) & ~15; }
  static constexpr __host__ __device__ int in(int l) { return S::ain(l); }                          // real inputs of layer l (both nets)
  static constexpr __host__ __device__ int out(int l, int net) { return net ? S::cout(l) : S::aout(l); }
  static constexpr __host__ __device__ int P(int l) { return l == NL ? up16(S::aout(NL - 1)) : up16(in(l) + 1); }   // padded input width of layer l / output width
  static constexpr __host__ __device__ int nkt(int l) { return P(l) / 16; }
  static constexpr __host__ __device__ int nto(int l) { return P(l + 1) / 16; }
  static constexpr __host__ __device__ int maxT() { int m = 0; for (int l = 0; l <= NL; l++) m = P(l) / 16 > m ? P(l) / 16 : m; return m; }
  static constexpr __host__ __device__ bool same() {
    for (int l = 1; l < NL; l++) if (S::ain(l) != S::cin(l)) return false;
    return S::cout(NL - 1) <= S::aout(NL - 1);
  }
  // consumption order of the weight fragments. forward: layer 0..NL-1 | pairs of output tiles | k tile | even, odd output tile;
  // dX: layer NL-1..1 | pairs of k tiles | o tile | even, odd k tile  (the two tiles of a pair are independent accumulation chains)
  static constexpr __host__ __device__ int nf() { int n = 0; for (int l = 0; l < NL; l++) n += nto(l) * nkt(l); return n; }
  static constexpr __host__ __device__ int nb() { int n = 0; for (int l = 1; l < NL; l++) n += nto(l) * nkt(l); return n; }
  static constexpr __host__ __device__ int fidx(int l, int to, int tk) {
    int n = 0;
    for (int i = 0; i < l; i++) n += nto(i) * nkt(i);
    const int p = to / 2, width = (2 * p + 1 < nto(l)) ? 2 : 1;
    return n + 2 * p * nkt(l) + tk * width + (to & 1);
  }
  static constexpr __host__ __device__ int bidx(int l, int to, int tk) {
    int n = 0;
    for (int i = NL - 1; i > l; i--) n += nto(i) * nkt(i);
    const int p = tk / 2, width = (2 * p + 1 < nkt(l)) ? 2 : 1;
    return n + 2 * p * nto(l) + to * width + (tk & 1);
  }
  static constexpr __host__ __device__ int slots(int l) { return (nto(l) * nkt(l) + NG - 1) / NG; }
  static constexpr __host__ __device__ int slotbase(int l) { int n = 0; for (int i = 0; i < l; i++) n += slots(i); return n; }
  static constexpr __host__ __device__ int nslots() { return slotbase(NL); }
  // exchange tiles of layer l: nto deltas + nkt inputs
  static constexpr __host__ __device__ int xtiles(int l) { return nto(l) + nkt(l); }
  static constexpr __host__ __device__ int xfloats(int b) { int m = 0; for (int l = b; l < NL; l += 2) { const int n = xtiles(l) * NG * 320; m = n > m ? n : m; } return m; }
};
constexpr int kSplitWaves = 4, kSplitSlots = 16;
#ifndef NM_PPO_SRING
#define NM_PPO_SRING 10      // weight fragments in flight per wave (6 .. 14 measure the same)
#endif
#ifndef NM_PPO_ABL           // measurement builds only (wrong results): 1 no weight stream, 2 no dW, 4 no parking, 8 no loss head, 16 no ELU, 32 no barriers
#define NM_PPO_ABL 0
#endif
// The weight ring relies on vmcnt retiring in order: fragment i is waited for with vmcnt(number of fragments requested after it). The
// scheduler treats the loads as independent and moves them away from the MFMAs they were written next to (measured: the waits counted
// DOWN 10, 9, .. 0 through a layer while the refills sat in a cluster behind it - the ring drained); a scheduling barrier after every
// request keeps each request at its place in the MFMA stream.
#define NM_PPO_PIN() __builtin_amdgcn_sched_barrier(0)

template <class S>
__global__ void __launch_bounds__(64 * kSplitWaves, 2) k_ppo_fwdbwd_split(PpoNet net, PpoBatch bt, float* __restrict__ partial) {
  typedef Split<S> X;
  constexpr int NL = S::NL, NG = X::NG, MT = X::maxT(), PO = X::P(NL) / 16, AO = S::aout(NL - 1), I = X::in(0), T0 = X::nkt(0);
  constexpr int NF = X::nf(), NB = X::nb(), kR = NM_PPO_SRING;
  constexpr int kXT = 20, kTileF = 16 * kXT;       // an exchange tile: 16 features x (16 rows + 4 pad) floats
  static_assert(X::same() && X::nslots() <= kSplitSlots && AO <= 32 && PO <= 2 && NF >= kR && NB >= kR, "shape outside the split kernel's limits");
  static_assert((X::xfloats(0) + X::xfloats(1)) * 4 <= 80 * 1024 && X::xfloats(0) >= kSplitWaves * 40, "two workgroups must fit the LDS of a CU");
  __shared__ __attribute__((aligned(16))) float xb0[X::xfloats(0)];
  __shared__ __attribute__((aligned(16))) float xb1[X::xfloats(1)];
  float (*hsum)[40] = reinterpret_cast<float (*)[40]>(xb0);      // the closing sums reuse the exchange buffer (the two buffers are exactly 80 KB)
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  const int g = w, nt = blockIdx.x & 1;            // row group; net (0 actor, 1 critic): workgroup-uniform
  const int vb = blockIdx.x >> 1, nvb = gridDim.x >> 1;      // blocks 2 v and 2 v + 1 walk the same passes v, v + nvb, ...
  const bool critic = nt != 0;
  const int prow = (r & 3) * 4 + (r >> 2);         // rows are parked as [row mod 4][row div 4]: one b128 read = rows 4 s + q', s = 0..3
  f32x4 gw[X::nslots()];
  sfor<X::nslots()>([&](auto i) { gw[i] = f32x4{0, 0, 0, 0}; });
  float a_dstd[PO][4], a_kl = 0.0f, a_surr = 0.0f, a_vl = 0.0f;
  float sdv[PO][4];
  sfor<PO>([&](auto T) {
    constexpr int t = T;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) { a_dstd[t][reg] = 0.0f; const int f = 16 * t + 4 * q + reg; sdv[t][reg] = f < AO ? bt.std[f] : 1.0f; }
  });
  const int npass = (bt.B + 16 * NG - 1) / (16 * NG);
#ifdef NM_PPO_STAMPS
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
  // per-row data of a pass. The observation tile (B operand of layer 0) is loaded one pass ahead; what the role's loss head needs is
  // requested inside the forward pass, at the first layer boundary by which every forward weight fragment has been requested (vmcnt
  // retires in order: an HBM gather issued earlier would hold up every wait for a fragment behind it; held a whole pass ahead it costs
  // 28 registers that the compiler spills and reloads one by one, each reload behind a full vmcnt(0))
  struct RowData { f32x4 obs[T0]; };
  struct HeadData { f32x4 act[PO], omu[PO], osd[PO]; float adv, olp, ret, tv; };
  // (row data is addressed as a uniform base + a 32-bit byte offset per lane - global_load with an SGPR base - instead of 64-bit
  // per-lane pointers: every VALU instruction of a wave costs its full latency here, the MFMA stream does not hide it)
  auto row_of = [&](int pass) -> unsigned {
    const int row = (pass * NG + g) * 16 + r;
    return row < bt.B ? (bt.rows ? (unsigned)bt.rows[row] : (unsigned)row) : 0u;
  };
  auto ldf = [](const float* base, unsigned byte_off) -> float { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
  auto load_rows = [&](int pass, RowData& rd) {
    const unsigned ob = row_of(pass) * (unsigned)(I * 4);
    sfor<T0>([&](auto T) {
      constexpr int t = T;
      if constexpr (16 * t + 16 <= I) {
        rd.obs[t] = *reinterpret_cast<const f32x4u*>(reinterpret_cast<const char*>(bt.obs) + (ob + 16u * q + 64u * t));
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) { const int f = 16 * t + 4 * q + reg; const float v = ldf(bt.obs, ob + 4u * (f < I ? f : 0)); rd.obs[t][reg] = f < I ? v : (f == I ? 1.0f : 0.0f); }
      }
    });
  };
  auto load_head = [&](int pass, HeadData& rd) {
    const unsigned lrow = row_of(pass);
    if (!critic) {
      const unsigned ab = lrow * (unsigned)(AO * 4);
      sfor<PO>([&](auto T) {
        constexpr int t = T;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const int f = 16 * t + 4 * q + reg;
          const unsigned at = ab + 4u * (f < AO ? f : 0);
          rd.act[t][reg] = ldf(bt.actions, at); rd.omu[t][reg] = ldf(bt.old_mu, at); rd.osd[t][reg] = ldf(bt.old_sigma, at);
        }
      });
      rd.adv = ldf(bt.adv, 4u * lrow); rd.olp = ldf(bt.old_logp, 4u * lrow);
    } else {
      rd.ret = ldf(bt.ret, 4u * lrow); rd.tv = ldf(bt.tval, 4u * lrow);
    }
  };
  // the first layer at whose start all forward fragments are in flight or consumed
  constexpr int LH = [] { int n = 0; for (int l = 0; l < NL; l++) { if (n + kR >= NF) return l; n += X::nto(l) * X::nkt(l); } return NL - 1; }();
  RowData nx;
  f32x4 ring[kR];
  const f32x4* __restrict__ WF = net.sf[nt];
  const f32x4* __restrict__ WB = net.sb[nt];
  unsigned wlane = 16u * lane; // this lane's byte offset inside a fragment; made opaque once per pass: the weight fragments are re-read from L2 every pass, not hoisted out of the loop
  asm volatile("" : "+v"(wlane));
  auto wfrag = [&](auto FWD, auto IDX) -> f32x4 {
    constexpr bool fwd = FWD;
    constexpr int i = IDX;
#if NM_PPO_ABL & 1      // measurement only: no weight stream (wrong results). A different constant per fragment: with ONE constant the two
                        // output tiles of a pair are the same chain and the compiler drops one of them (260 MFMAs instead of 432)
    return f32x4{__int_as_float(wlane + 64 * i), 1.0f + i, 0.5f + i, 0.25f + i};
#else
    // A wave-uniform base + a 32-bit byte offset per lane: global_load with an SGPR base, no 64-bit address arithmetic per load. One base
    // per FOUR fragments (4 KB) and the instruction's immediate offset inside it: a base per fragment is 64 SGPR pairs that the compiler
    // hoists out of the pass loop, spills into VGPR lanes and fetches back with two v_readlane + s_nop in front of every load.
    constexpr int grp = i >> 2, sub = i & 3;
    const char* base = reinterpret_cast<const char*>((fwd ? WF : WB) + grp * 256);
    return *reinterpret_cast<const f32x4*>((base + wlane) + sub * 1024);
#endif
  };
  if (vb < npass) load_rows(vb, nx);
  sfor<kR>([&](auto i) { ring[i] = wfrag(std::true_type{}, i); NM_PPO_PIN(); });
  for (int pass = vb; pass < npass; pass += nvb) {
    PPO_STAMP(15);
    const int row = (pass * NG + g) * 16 + r;
    const bool live = row < bt.B;
    HeadData cu;
    f32x4 a[NL][MT], d[2][MT], out[PO];
    sfor<T0>([&](auto T) { a[0][T] = live ? nx.obs[T] : f32x4{0, 0, 0, 0}; });
    PPO_STAMP(0);
    // ---- forward
    sfor<NL>([&](auto L) {
      constexpr int l = L, nkt = X::nkt(l), nto = X::nto(l);
      constexpr bool last = l == NL - 1;
      if constexpr (l > 0) PPO_STAMP(l);
      if constexpr (l == LH) load_head(pass, cu);
      sfor<(nto + 1) / 2>([&](auto PP) {
        constexpr int to0 = 2 * PP, to1 = to0 + 1;
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        sfor<nkt>([&](auto TK) {
          constexpr int tk = TK;
          f32x4 w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
          {
            constexpr int i = X::fidx(l, to0, tk);
            w0 = ring[i % kR];
            if constexpr (i + kR < NF) { ring[i % kR] = wfrag(std::true_type{}, std::integral_constant<int, (i + kR < NF ? i + kR : 0)>{}); NM_PPO_PIN(); }
          }
          if constexpr (to1 < nto) {
            constexpr int i = X::fidx(l, to1 < nto ? to1 : to0, tk);
            w1 = ring[i % kR];
            if constexpr (i + kR < NF) { ring[i % kR] = wfrag(std::true_type{}, std::integral_constant<int, (i + kR < NF ? i + kR : 0)>{}); NM_PPO_PIN(); }
          }
#pragma unroll
          for (int j = 0; j < 4; j++) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[j], a[l][tk][j], acc0, 0, 0, 0);
            if constexpr (to1 < nto) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[j], a[l][tk][j], acc1, 0, 0, 0);
          }
        });
        auto finish = [&](auto TO, const f32x4& acc) {
          constexpr int to = TO;
          if constexpr (last) {
            out[to] = acc;
          } else {
            constexpr int NO = X::in(l + 1);      // real outputs of this layer = real inputs of the next (both nets)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
              const float v = acc[reg];
#if NM_PPO_ABL & 16     // measurement only: no ELU
              const float y = v;
#else
              // ELU(v) = v > 0 ? v : e with e = exp(v) - 1 >= v for every v: the median of (v, e, 0) - one v_med3_f32 instead of compare + select
              const float e = __expf(v) - 1.0f, y = __builtin_amdgcn_fmed3f(v, e, 0.0f);
#endif
              if constexpr (16 * to + 16 <= NO) {
                a[l + 1][to][reg] = y;
              } else {
                const int col = 16 * to + 4 * q + reg;
                const float pad = col == NO ? 1.0f : 0.0f;
                a[l + 1][to][reg] = col < NO ? y : pad;   // ELU | bias carrier | padding
              }
            }
          }
        };
        finish(std::integral_constant<int, to0>{}, acc0);
        if constexpr (to1 < nto) finish(std::integral_constant<int, to1>{}, acc1);
      });
    });
    PPO_STAMP(4);
    // ---- loss head, by role. Actor: lane (row r, q) holds means 16 t + 4 q + reg, per-row sums cross the four q lanes. Critic: the
    // value is output column 0 (tile 0, register 0 of the q = 0 lanes).
    sfor<PO>([&](auto T) { d[0][T] = f32x4{0, 0, 0, 0}; });
#if NM_PPO_ABL & 8      // measurement only: no loss head
    sfor<PO>([&](auto T) { d[0][T] = out[T]; });
    if (bt.B < 0)
#endif
    if (!critic) {
      // Written on 1 / sigma and log sigma (one v_rcp_f32 and one v_log_f32 per column and pass, from an opaque copy of sigma: left to
      // itself the compiler hoists every function of the loop-invariant sigmas out of the pass loop and spills them), so that a column
      // costs ~25 VALU instructions instead of ~60 (the divisions' scaling sequences): with z = (a - mu) / sigma,
      //   log p = -z^2 / 2 - log sigma - c,  d log p / d mu = z / sigma,  d log p / d sigma = (z^2 - 1) / sigma.
      // On this chip the f32 MFMA does not hide a wave's own VALU work (scripts/micro/mfma_chain.hip: 4 dependent v_fma behind every MFMA
      // cost their full 22 cycles on top of its 32), and the head runs on the actor blocks only: it is on the launch's critical path.
      float lp = 0.0f, kl = 0.0f;
      float is_[PO][4], zz[PO][4];
      sfor<PO>([&](auto T) {
        constexpr int t = T;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          float sd = sdv[t][reg];
          asm volatile("" : "+v"(sd));
          const int f = 16 * t + 4 * q + reg;
          is_[t][reg] = 0.0f; zz[t][reg] = 0.0f;
          if (f < AO) {
            const float is = __builtin_amdgcn_rcpf(sd), ls = __logf(sd);
            const float mu = out[t][reg], osd = cu.osd[t][reg], omu = cu.omu[t][reg];
            const float z = (cu.act[t][reg] - mu) * is;
            lp += -0.5f * z * z - ls - 0.9189385332046727f;
            kl += __logf(sd * __builtin_amdgcn_rcpf(osd) + 1.0e-5f) + (osd * osd + (omu - mu) * (omu - mu)) * (0.5f * is * is) - 0.5f;
            is_[t][reg] = is; zz[t][reg] = z;
          }
        }
      });
      lp += __shfl_xor(lp, 16); lp += __shfl_xor(lp, 32);
      float dlogp = 0.0f;      // d(surrogate)/d(logp) * (1/B)
      if (live) {
        const float ratio = __expf(lp - cu.olp);
        const float s1 = -cu.adv * ratio, rc = fminf(fmaxf(ratio, 1.0f - bt.clip), 1.0f + bt.clip), s2 = -cu.adv * rc;
        const bool inside = ratio > 1.0f - bt.clip && ratio < 1.0f + bt.clip;
        // torch.max(s1, s2).backward(): the larger branch gets the gradient, a tie splits it; the clipped branch has zero slope outside the range
        const float g1 = s1 > s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f), g2 = 1.0f - g1;
        dlogp = (g1 * (-cu.adv * ratio) + g2 * (inside ? -cu.adv * ratio : 0.0f)) * bt.inv_B;
        a_kl += kl;
        if (q == 0) a_surr += fmaxf(s1, s2);
      }
      sfor<PO>([&](auto T) {      // (dlogp = 0 on rows past the end of the mini-batch, is_ = z = 0 on columns past the actions: they add 0)
        constexpr int t = T;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const float gq = dlogp * is_[t][reg], z = zz[t][reg];
          d[0][t][reg] = gq * z;
          a_dstd[t][reg] += gq * (z * z - 1.0f);
        }
      });
    } else if (live && q == 0) {
      const float v = out[0][0];
      float dv, vl;
      if (bt.clip_value) {
        const float dvt = v - cu.tv, vc = cu.tv + fminf(fmaxf(dvt, -bt.clip), bt.clip);
        const float l1 = (v - cu.ret) * (v - cu.ret), l2 = (vc - cu.ret) * (vc - cu.ret);
        const bool in2 = dvt > -bt.clip && dvt < bt.clip;
        const float h1 = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f), h2 = 1.0f - h1;
        vl = fmaxf(l1, l2);
        dv = h1 * 2.0f * (v - cu.ret) + h2 * (in2 ? 2.0f * (vc - cu.ret) : 0.0f);
      } else {
        vl = (cu.ret - v) * (cu.ret - v);
        dv = 2.0f * (v - cu.ret);
      }
      a_vl += vl;
      d[0][0][0] = dv * bt.vcoef * bt.inv_B;
    }
    PPO_STAMP(5);
    // the dX fragments are requested only now: in flight across the loss head they are 4 x kR registers that the head's divisions and
    // logarithms have no room for (the spill reloads each waited for the whole ring)
    sfor<kR>([&](auto i) { ring[i] = wfrag(std::false_type{}, i); NM_PPO_PIN(); });
    // ---- backward
    sfor<NL>([&](auto LL) {
      constexpr int l = NL - 1 - LL, cur = LL & 1, nxt = cur ^ 1;
      constexpr int nkt = X::nkt(l), nto = X::nto(l);
      // layout of the layer's exchange buffer: [delta tiles][row group] then [input tiles][row group]
      float* xb = (l & 1) ? xb1 : xb0;
      constexpr int dbase = 0, abase = nto;
      if constexpr (LL > 0) PPO_STAMP(5 + 2 * LL);
      // park this wave's deltas and layer inputs, transposed, for the workgroup's dW
#if NM_PPO_ABL & 4      // measurement only: no parking
      if (bt.B < 0)
#endif
      {
      sfor<nto>([&](auto T) {
        constexpr int t = T;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) xb[((dbase + t) * NG + g) * kTileF + (4 * q + reg) * kXT + prow] = d[cur][t][reg];
      });
      if constexpr (l == 0) {       // the observation tile is not kept across the pass (20 registers): read it again (L2) for layer 0's dW
        RowData ob;
        load_rows(pass, ob);
        sfor<T0>([&](auto T) { a[0][T] = live ? ob.obs[T] : f32x4{0, 0, 0, 0}; });
      }
      sfor<nkt>([&](auto T) {
        constexpr int t = T;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) xb[((abase + t) * NG + g) * kTileF + (4 * q + reg) * kXT + prow] = a[l][t][reg];
      });
      }
      // d_{l-1} = (W' d_l) * ELU'(a_l): registers only
      if constexpr (l > 0) {
        sfor<(nkt + 1) / 2>([&](auto PP) {
          constexpr int tk0 = 2 * PP, tk1 = tk0 + 1;
          f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
          sfor<nto>([&](auto TO) {
            constexpr int to = TO;
            f32x4 w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
            {
              constexpr int i = X::bidx(l, to, tk0);
              w0 = ring[i % kR];
              if constexpr (i + kR < NB) { ring[i % kR] = wfrag(std::false_type{}, std::integral_constant<int, (i + kR < NB ? i + kR : 0)>{}); NM_PPO_PIN(); }
            }
            if constexpr (tk1 < nkt) {
              constexpr int i = X::bidx(l, to, tk1 < nkt ? tk1 : tk0);
              w1 = ring[i % kR];
              if constexpr (i + kR < NB) { ring[i % kR] = wfrag(std::false_type{}, std::integral_constant<int, (i + kR < NB ? i + kR : 0)>{}); NM_PPO_PIN(); }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[j], d[cur][to][j], acc0, 0, 0, 0);
              if constexpr (tk1 < nkt) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[j], d[cur][to][j], acc1, 0, 0, 0);
            }
          });
          auto finish = [&](auto TK, const f32x4& acc) {
            constexpr int tk = TK;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
              const float av = a[l][tk][reg];
#if NM_PPO_ABL & 16
              const float gr = acc[reg] + av;
#else
              const float gr = acc[reg] * fminf(av + 1.0f, 1.0f);                                // ELU'(z) from ELU(z): 1 if ELU > 0, else ELU + 1 (<= 1)
#endif
              if constexpr (16 * tk + 16 <= X::in(l)) d[nxt][tk][reg] = gr;
              else d[nxt][tk][reg] = 16 * tk + 4 * q + reg < X::in(l) ? gr : 0.0f;              // no gradient into the 1-column / padding
            }
          };
          finish(std::integral_constant<int, tk0>{}, acc0);
          if constexpr (tk1 < nkt) finish(std::integral_constant<int, tk1>{}, acc1);
        });
      }
      if constexpr (l == 1) {
        // the last weight fragment of this pass has been consumed; what follows (dW of layers 1 and 0) waits on LDS only. Fetch the next
        // pass's rows (HBM) and the first forward fragments now: they retire in order long before the next pass waits on vmcnt
        asm volatile("" : "+v"(wlane));
        if (pass + nvb < npass) load_rows(pass + nvb, nx);
        sfor<kR>([&](auto i) { ring[i] = wfrag(std::true_type{}, i); NM_PPO_PIN(); });
      }
      PPO_STAMP(6 + 2 * LL);
#if !(NM_PPO_ABL & 32)  // measurement only: no barriers
      lds_barrier();
#endif
#if NM_PPO_ABL & 2      // measurement only: no dW
      if (bt.B < 0)
#endif
      // dW_l += D' a_l over the 64 rows of the workgroup: this wave's tiles of its net. An unused slot of the table computes on tile (0, 0)
      // into an accumulator that is never stored - no branch; the operands of slot i + 1 are read from LDS while slot i multiplies.
      {
        constexpr int NS = X::slots(l), SB = X::slotbase(l);
        f32x4 fd[2][NG], fa[2][NG];
        auto frags = [&](auto SI, auto BUF) {
          const int code = net.slot4[g][SB + SI], c = code < 0 ? 0 : code;
          const float* dp = xb + ((dbase + (c & 15)) * NG) * kTileF + r * kXT + 4 * q;
          const float* ap = xb + ((abase + (c >> 4)) * NG) * kTileF + r * kXT + 4 * q;
#pragma unroll
          for (int gg = 0; gg < NG; gg++) {
            fd[BUF][gg] = *reinterpret_cast<const f32x4*>(dp + gg * kTileF);
            fa[BUF][gg] = *reinterpret_cast<const f32x4*>(ap + gg * kTileF);
          }
        };
        frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        sfor<NS>([&](auto SI) {
          constexpr int si = SI, buf = si & 1;
          if constexpr (si + 1 < NS) frags(std::integral_constant<int, (si + 1 < NS ? si + 1 : 0)>{}, std::integral_constant<int, buf ^ 1>{});
#pragma unroll
          for (int gg = 0; gg < NG; gg++)
#pragma unroll
            for (int j = 0; j < 4; j++) gw[SB + si] = __builtin_amdgcn_mfma_f32_16x16x4f32(fd[buf][gg][j], fa[buf][gg][j], gw[SB + si], 0, 0, 0);
        });
      }
    });
  }
  PPO_STAMP(13);
  // ---- this workgroup's partial gradient, in FLAT parameter order (actor W0 b0 ..., critic ..., std): row v of `partial` is then read by
  // k_ppo_step / k_ppo_reduce as it stands - 256 contiguous bytes per wave and row, no map indirection, no gaps (in the merged layout
  // the 15 k parameters lie spread over 28 k positions: half again as many cache lines per row). The scalars keep their place behind gtotal.
  float* P = partial + (size_t)vb * (net.gtotal + kNS);
  sfor<NL>([&](auto L) {
    constexpr int l = L, in = X::in(l);
    const int no = critic ? S::cout(l) : S::aout(l), wo = net.woff[nt][l], bo = net.boff[nt][l];
    sfor<X::slots(l)>([&](auto SI) {
      constexpr int slot = X::slotbase(l) + SI;
      const int code = net.slot4[g][slot];
      if (code >= 0) {
        const int kk = 16 * (code >> 4) + r;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const int o = 16 * (code & 15) + 4 * q + reg;
          if (o < no && kk <= in) P[kk < in ? wo + o * in + kk : bo + o] = gw[slot][reg];
        }
      }
    });
  });
  PPO_STAMP(13);
  for (int o = 1; o < 64; o <<= 1) { a_kl += __shfl_xor(a_kl, o); a_surr += __shfl_xor(a_surr, o); a_vl += __shfl_xor(a_vl, o); }
  __syncthreads();                       // every wave has left its last dW: the exchange buffer is free for the closing sums
  sfor<PO>([&](auto T) {
    constexpr int t = T;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      float v = a_dstd[t][reg];
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
      const int f = 16 * t + 4 * q + reg;
      if (r == 0 && f < 32) hsum[w][f] = v;
    }
  });
  if (lane == 0) { hsum[w][32] = a_kl; hsum[w][33] = a_surr; hsum[w][34] = a_vl; }
  __syncthreads();
  // the actor block owns d loss / d sigma (flat parameters nparam - A ..), KL and surrogate (scalars 32, 33 behind gtotal), the critic block
  // the value loss (scalar 34)
  if (critic ? tid == 34 : tid < 34) {
    float v = 0.0f;
    for (int gg = 0; gg < kSplitWaves; gg++) v += hsum[gg][tid];
    if (tid >= 32) P[net.gtotal + tid] = v;
    else if (tid < AO) P[net.nparam_flat - AO + tid] = v;
  }
  PPO_STAMP(14);
}

// ------------------------------------------------------------------------------------------------ collection step on the same packing
// PPO.act for the reference-shaped network in ONE launch: the register-resident forward of the merged network (one wave = 16 rows,
// weights from the update's own packed copy - no separate repack for the collector), then the sampling head on the output registers:
// lane (row r, q) holds outputs 16 t + 4 q + reg, i.e. whole action pairs, so Box-Muller pairs, log-probability partials and the
// storage writes need no exchange; the log-probability crosses the four q lanes with two shuffles. Same counter generator and keys
// as nm_ppo_sample (seed, iteration, step, env, action pair).
__device__ __forceinline__ float ppo_u24(uint64_t seed, uint64_t a, uint64_t b) {   // = u24 of nm_rl.hip
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * (a + 1) + 0xD1B54A32D192ED03ull * b;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return ((float)(uint32_t)(x >> 40) + 1.0f) * (1.0f / 16777216.0f);
}
// the bookkeeping of the PREVIOUS step (what nm_ppo_record does), done at the head of the next PPO.act launch: rew == nullptr = nothing to do
struct PpoRecord {
  const float* rew; const int64_t* done; const float* time_outs; const float* values;
  float gamma;
  float* rewards_store; unsigned char* dones_store;
  float *cur_ret, *cur_len, *fin3;
  const float* ep_stats; const int* ep_idx; int n_ep; float* ep_acc;
};
template <class S>
__global__ void __launch_bounds__(64) k_ppo_act_fast(PpoNet net, const float* __restrict__ obs, const float* __restrict__ std, int N, uint64_t seed,
                                                     const int64_t* __restrict__ iter_dev, int step, float* __restrict__ actions, float* __restrict__ logp,
                                                     float* __restrict__ values, float* __restrict__ mu, float* __restrict__ sigma, float* __restrict__ obs_store,
                                                     PpoRecord rec) {
  typedef Seq<S> Q;
  constexpr int NL = S::NL, MT = S::maxT(), PO = S::P(NL) / 16, AO = S::aout(NL - 1), I = S::Kr(0), T0 = S::P(0) / 16, NF = Q::count(true);
  static_assert(AO % 2 == 0, "action pairs");
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const int row = blockIdx.x * 16 + r;
  const bool live = row < N;
  const size_t lrow = live ? row : 0;
  if (rec.rew) {     // k_ppo_record for the step before this one (same arithmetic, env = row): the launch it would have been is this one
    if (blockIdx.x == 0)
      for (int e = lane; e < rec.n_ep; e += 64) rec.ep_acc[e] += rec.ep_stats[rec.ep_idx[e]];
    if (q == 0 && live) {
      const float rw = rec.rew[row];
      const bool d = rec.done[row] > 0;
      rec.rewards_store[row] = rw + (rec.time_outs ? rec.gamma * rec.values[row] * rec.time_outs[row] : 0.0f);
      rec.dones_store[row] = d ? 1 : 0;
      const float cr = rec.cur_ret[row] + rw, cl = rec.cur_len[row] + 1.0f;
      if (d) { atomicAdd(rec.fin3, cr); atomicAdd(rec.fin3 + 1, cl); atomicAdd(rec.fin3 + 2, 1.0f); }
      rec.cur_ret[row] = d ? 0.0f : cr;
      rec.cur_len[row] = d ? 0.0f : cl;
    }
  }
  auto wfrag = [&](auto IDX) -> f32x4 {
    constexpr int e = Q::entry(true, IDX), l = e & 15, to = (e >> 4) & 15, tk = e >> 8;
    return net.pf[l][(to * (S::P(l) / 16) + tk) * 64 + lane];
  };
  f32x4 ring[kRing];
  f32x4 a[NL][MT], out[PO];
  sfor<T0>([&](auto T) {
    constexpr int t = T;
    f32x4 v;
    if constexpr (16 * t + 16 <= I) {
      v = *reinterpret_cast<const f32x4u*>(obs + lrow * I + 16 * t + 4 * q);
      if (obs_store && live) *reinterpret_cast<f32x4u*>(obs_store + lrow * I + 16 * t + 4 * q) = v;
    } else {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int f = 16 * t + 4 * q + reg;
        const float x = obs[lrow * I + (f < I ? f : 0)];
        if (obs_store && live && f < I) obs_store[lrow * I + f] = x;
        v[reg] = f < I ? x : (f == I ? 1.0f : 0.0f);
      }
    }
    a[0][t] = v;
  });
  sfor<kRing>([&](auto i) { ring[i] = wfrag(i); });
  sfor<NL>([&](auto L) {
    constexpr int l = L, nkt = S::P(l) / 16, nto = S::P(l + 1) / 16;
    constexpr bool last = l == NL - 1;
    sfor<(nto + 1) / 2>([&](auto PP) {
      constexpr int to0 = 2 * PP, to1 = to0 + 1;
      f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      sfor<nkt>([&](auto TK) {
        constexpr int tk = TK;
        constexpr bool z0 = S::nz(l, to0, tk), z1 = to1 < nto && S::nz(l, to1, tk);
        f32x4 w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
        if constexpr (z0) {
          constexpr int i = Q::index(true, l, to0, tk);
          w0 = ring[i % kRing];
          if constexpr (i + kRing < NF) ring[i % kRing] = wfrag(std::integral_constant<int, (i + kRing < NF ? i + kRing : 0)>{});
        }
        if constexpr (z1) {
          constexpr int i = Q::index(true, l, to1, tk);
          w1 = ring[i % kRing];
          if constexpr (i + kRing < NF) ring[i % kRing] = wfrag(std::integral_constant<int, (i + kRing < NF ? i + kRing : 0)>{});
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if constexpr (z0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[j], a[l][tk][j], acc0, 0, 0, 0);
          if constexpr (z1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[j], a[l][tk][j], acc1, 0, 0, 0);
        }
      });
      auto finish = [&](auto TO, const f32x4& acc) {
        constexpr int to = TO;
        if constexpr (last) {
          out[to] = acc;
        } else {
#pragma unroll
          for (int reg = 0; reg < 4; reg++) {
            const int col = 16 * to + 4 * q + reg;
            const float v = acc[reg];
            a[l + 1][to][reg] = col < S::Or(l) ? (v > 0.0f ? v : __expf(v) - 1.0f) : (col == S::Or(l) ? 1.0f : 0.0f);
          }
        }
      };
      finish(std::integral_constant<int, to0>{}, acc0);
      if constexpr (to1 < nto) finish(std::integral_constant<int, to1>{}, acc1);
    });
  });
  // ---- sampling head
  const uint64_t ctr = (uint64_t)iter_dev[0] * 4096ull + (uint64_t)step;
  float lp = 0.0f;
  sfor<PO>([&](auto T) {
    constexpr int t = T;
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {
      const int f0 = 16 * t + 4 * q + 2 * pr;            // even action index: the pair (f0, f0 + 1) shares one Box-Muller draw
      if (f0 < AO) {
        const float u1 = ppo_u24(seed, (uint64_t)lrow * 64 + f0, ctr), u2 = ppo_u24(seed, (uint64_t)lrow * 64 + f0 + 1, ctr);
        const float rad = sqrtf(-2.0f * __logf(u1));
        float sn, cs;
        __sincosf(6.283185307179586f * u2, &sn, &cs);
        const float z[2] = {rad * cs, rad * sn};
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
          const float m = out[t][2 * pr + hh], sd = std[f0 + hh];
          if (live) {
            actions[lrow * AO + f0 + hh] = m + sd * z[hh];
            mu[lrow * AO + f0 + hh] = m;
            sigma[lrow * AO + f0 + hh] = sd;
          }
          lp += -0.5f * z[hh] * z[hh] - __logf(sd) - 0.9189385332046727f;
        }
      } else if (f0 == AO && live) {
        values[lrow] = out[t][2 * pr];
      }
    }
  });
  lp += __shfl_xor(lp, 16); lp += __shfl_xor(lp, 32);
  if (q == 0 && live) logp[lrow] = lp;
}
// the reference's networks (envs/nightmare_v3_config.py:107-109): 66 -> 54 -> 42 -> 30 -> 18 | 1
typedef Shape4<66, 54, 42, 30, 18, 54, 42, 30> RefShape;

// merged [Op x Kp] matrices (bias in column Kr) -> the two MFMA packings of every layer (blockIdx.y = layer).
//   generic kernel: forward [O tile t][k group g][lane] = Wm[16 t + r][4 (4 g + j) + q], backward [K tile t][o group g][lane] = Wm[4 (4 g + j) + q][16 t + r]
//   fast kernel   : forward [O tile t][K tile g][lane] = Wm[16 t + r][16 g + 4 q + j],   backward [K tile t][O tile g][lane] = Wm[16 g + 4 q + j][16 t + r]
__global__ void k_ppo_pack(PpoNet net, const float* __restrict__ Wm_all, int fast) {
  const int l = blockIdx.y, Op = net.Op[l], Kp = net.Kp[l];
  const float* __restrict__ Wm = Wm_all + net.goff[l];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nf = (Op / 16) * (Kp / 16) * 64;
  if (i >= 2 * nf) return;
  const int which = i / nf, ii = i - which * nf;
  const int lane = ii & 63, r = lane & 15, q = lane >> 4;
  f32x4 v;
  if (which == 0) {
    const int ngrp = Kp / 16, g = (ii >> 6) % ngrp, t = (ii >> 6) / ngrp;
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = Wm[(size_t)(16 * t + r) * Kp + (fast ? 16 * g + 4 * q + j : 4 * (4 * g + j) + q)];
    const_cast<f32x4*>(net.pf[l])[ii] = v;
  } else {
    const int ngrp = Op / 16, g = (ii >> 6) % ngrp, t = (ii >> 6) / ngrp;
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = Wm[(size_t)(fast ? 16 * g + 4 * q + j : 4 * (4 * g + j) + q) * Kp + 16 * t + r];
    const_cast<f32x4*>(net.pb[l])[ii] = v;
  }
}
// flat real parameters -> merged matrices (zero elsewhere: the off-diagonal blocks and the padding never change)
// the split kernel's packings from the merged matrices: parameter i sits at Wm[map[i]] and goes to sf[sfi[i]] and (layers >= 1) sb[sbi[i]]
__global__ void k_ppo_pack_split(const float* __restrict__ Wm_all, const int* __restrict__ map, const int* __restrict__ sfi, const int* __restrict__ sbi, int n,
                                 float* __restrict__ sf, float* __restrict__ sb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || map[i] < 0) return;
  const float p = Wm_all[map[i]];
  sf[sfi[i]] = p;
  if (sbi[i] >= 0) sb[sbi[i]] = p;
}
__global__ void k_ppo_scatter(const float* __restrict__ flat, const int* __restrict__ map, int n, float* __restrict__ Wm_all) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && map[i] >= 0) Wm_all[map[i]] = flat[i];
}
// gradient of every real parameter: sum of the workgroups' partial gradients at its merged position (std: from the scalar tail,
// plus the entropy bonus -c_e * d(sum_j log std_j)/d(std_j))
constexpr int kRedParams = 64, kRedWaves = 8;    // (16 waves per block measured no faster: 31.0 vs 29.8 us for k_ppo_step)
__global__ void __launch_bounds__(kRedParams * kRedWaves) k_ppo_reduce(const float* __restrict__ partial, int nwg, int stride, const int* __restrict__ map, int n, int gtotal,
                             const float* __restrict__ flat, float ent_coef, float inv_B, float* __restrict__ grad, int compact) {
  // entry n (one past the parameters) = this mini-batch's mean KL to the behaviour policy: it travels with the gradient through a
  // multi-GPU all-reduce. A block sums 64 consecutive parameters: wave w takes workgroups w, w + 8, ... (independent loads, 256 B runs), LDS joins the waves
  // in a fixed order - the result does not depend on scheduling
  __shared__ float part[kRedWaves][kRedParams];
  const int lane = threadIdx.x & 63, w0 = threadIdx.x >> 6;
  const int i = blockIdx.x * kRedParams + lane;
  float g = 0.0f;
  int m = 0;
  if (i <= n) {
    m = i < n ? map[i] : -33;
    // compact: the rows are in flat parameter order (k_ppo_fwdbwd_split), else in the merged layout (map)
    const float* P = partial + (compact ? (i < n ? i : gtotal + 32) : (m >= 0 ? m : gtotal + (-m - 1)));
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f;
    int w = w0;
    for (; w + 3 * kRedWaves < nwg; w += 4 * kRedWaves) {
      g0 += P[(size_t)w * stride]; g1 += P[(size_t)(w + kRedWaves) * stride];
      g2 += P[(size_t)(w + 2 * kRedWaves) * stride]; g3 += P[(size_t)(w + 3 * kRedWaves) * stride];
    }
    for (; w < nwg; w += kRedWaves) g0 += P[(size_t)w * stride];
    g = (g0 + g1) + (g2 + g3);
  }
  part[w0][lane] = g;
  __syncthreads();
  if (w0 == 0 && i <= n) {
    g = 0.0f;
#pragma unroll
    for (int w = 0; w < kRedWaves; w++) g += part[w][lane];
    if (i == n) g *= inv_B;
    else if (m < 0) g -= ent_coef / flat[i];
    grad[i] = g;
  }
}
// state: [0] lr [1] step [2] last kl [3] sum of value losses [4] sum of surrogate losses [5] mini-batches [6] clip coefficient [7] grad norm
__global__ void k_ppo_scalars(const float* __restrict__ partial, int nwg, int stride, int gtotal, const float* __restrict__ grad, int n, float inv_B,
                              float desired_kl, int adaptive, float max_norm, float kl_override, int kl_from_grad, float* __restrict__ state) {
  __shared__ float red[4][32];
  const int tid = threadIdx.x;
  float kl = 0, su = 0, vl = 0, n2 = 0;
  for (int w = tid; w < nwg; w += blockDim.x) {
    const float* P = partial + (size_t)w * stride + gtotal;
    kl += P[32]; su += P[33]; vl += P[34];
  }
  for (int i = tid; i < n; i += blockDim.x) n2 += grad[i] * grad[i];
  for (int o = 32; o > 0; o >>= 1) { kl += __shfl_xor(kl, o); su += __shfl_xor(su, o); vl += __shfl_xor(vl, o); n2 += __shfl_xor(n2, o); }
  if ((tid & 63) == 0) { red[0][tid >> 6] = kl; red[1][tid >> 6] = su; red[2][tid >> 6] = vl; red[3][tid >> 6] = n2; }
  __syncthreads();
  if (tid == 0) {
    kl = su = vl = n2 = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) { kl += red[0][w]; su += red[1][w]; vl += red[2][w]; n2 += red[3][w]; }
    float klm = kl * inv_B;
    if (kl_from_grad) klm = grad[n];                   // multi-GPU: the caller all-reduced gradient | KL (nm_ppo_copy_grad) and divided by the ranks
    if (kl_override >= 0.0f) klm = kl_override;
    float lr = state[0];
    if (adaptive) {                                    // rsl_rl v1.0.2 PPO.update: schedule == 'adaptive'
      if (klm > desired_kl * 2.0f) lr = fmaxf(1e-5f, lr / 1.5f);
      else if (klm < desired_kl / 2.0f && klm > 0.0f) lr = fminf(1e-2f, lr * 1.5f);
    }
    state[0] = lr;
    state[1] += 1.0f;
    state[2] = klm;
    state[3] += vl * inv_B;
    state[4] += su * inv_B;
    state[5] += 1.0f;
    const float norm = sqrtf(n2);
    state[6] = fminf(1.0f, max_norm / (norm + 1e-6f));   // torch.nn.utils.clip_grad_norm_
    state[7] = norm;
  }
}
__global__ void k_ppo_adam(float* __restrict__ flat, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ grad, int n,
                           const float* __restrict__ state, float b1, float b2, float eps, const int* __restrict__ map, float* __restrict__ Wm_all) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float lr = state[0], t = state[1], g = grad[i] * state[6];
  const float mi = b1 * m[i] + (1.0f - b1) * g, vi = b2 * v[i] + (1.0f - b2) * g * g;
  m[i] = mi; v[i] = vi;
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  const float p = flat[i] - (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);      // torch.optim.Adam (no amsgrad, no weight decay)
  flat[i] = p;
  if (map[i] >= 0) Wm_all[map[i]] = p;
}

// The step of a mini-batch - k_ppo_reduce, k_ppo_scalars, k_ppo_adam and the repacking of k_ppo_pack - as ONE launch:
// block b owns parameters 64 b .. 64 b + 63 from the sum of the partial gradients to the Adam update and writes the new value straight
// into its places of the two MFMA packings (index tables built by nm_ppo_create). The only thing a block needs from the others is the
// gradient norm: every block publishes the sum of squares of its 64 gradients (device-scope atomic), one grid barrier, then every block
// adds the nblk partial sums in the same fixed order - so all blocks (and, after an all-reduce, all ranks) get the same clip coefficient
// bit for bit - and evaluates the KL-adaptive learning rate itself from the workgroups' scalar tails; block 0 files the state.
// nblk (~236) blocks of 512 threads are co-resident on the 256 CUs, which the spinning barrier needs: nm_ppo_create asks the runtime
// (hipOccupancyMaxActiveBlocksPerMultiprocessor x the CU count) and keeps the four-launch step when the grid does not fit. The spin is
// bounded; a time-out turns the step into a no-op for EVERY block (state[8] = 1, nm_ppo_get_state / nm_ppo_snapshot_state report it).
struct StepArgs {
  const float* partial; int nwg, stride, gtotal;
  const int* map; int n;
  float *flat, *m, *v, *grad, *state, *Wm, *n2part;
  float *pf, *pb; const int *pfi, *pbi;        // packings as float arrays + per-parameter positions in them (-1: none)
  float *sf, *sb; const int *sfi, *sbi;        // the split kernel's per-net packings (null: the network has none)
  unsigned* bar;                                // [0] arrivals, [1] generation
  float ent_coef, inv_B, desired_kl, max_norm, kl_override, b1, b2, eps;
  int adaptive, kl_from_grad, do_reduce;
  int compact;                                  // the partial rows are in flat parameter order (k_ppo_fwdbwd_split)
};
__global__ void __launch_bounds__(kRedParams * kRedWaves) k_ppo_step(StepArgs a) {
  __shared__ float part[kRedWaves][kRedParams];
  const int lane = threadIdx.x & 63, w0 = threadIdx.x >> 6, b = blockIdx.x;
  const int i = b * kRedParams + lane, n = a.n;
  const float lr0 = a.state[0], t0 = a.state[1], s3 = a.state[3], s4 = a.state[4], s5 = a.state[5];   // read before the barrier: block 0 rewrites them behind it
  const unsigned gen = __hip_atomic_load(a.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float g = 0.0f;
  int mp = 0;
  if (i <= n) mp = i < n ? a.map[i] : -33;
  if (a.do_reduce) {       // k_ppo_reduce's arithmetic and summation order
    if (i <= n) {
      const float* P = a.partial + (a.compact ? (i < n ? i : a.gtotal + 32) : (mp >= 0 ? mp : a.gtotal + (-mp - 1)));
      float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f;
      int w = w0;
      for (; w + 3 * kRedWaves < a.nwg; w += 4 * kRedWaves) {
        g0 += P[(size_t)w * a.stride]; g1 += P[(size_t)(w + kRedWaves) * a.stride];
        g2 += P[(size_t)(w + 2 * kRedWaves) * a.stride]; g3 += P[(size_t)(w + 3 * kRedWaves) * a.stride];
      }
      for (; w < a.nwg; w += kRedWaves) g0 += P[(size_t)w * a.stride];
      g = (g0 + g1) + (g2 + g3);
    }
    part[w0][lane] = g;
    __syncthreads();
    if (w0 == 0 && i <= n) {
      g = 0.0f;
#pragma unroll
      for (int w = 0; w < kRedWaves; w++) g += part[w][lane];
      if (i == n) g *= a.inv_B;
      else if (mp < 0) g -= a.ent_coef / a.flat[i];
      a.grad[i] = g;
    }
  } else if (w0 == 0 && i <= n) {
    g = a.grad[i];           // the caller's (all-reduced) gradient | KL
  }
  if (w0 == 0) {
    float s = i < n ? g * g : 0.0f;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    // published by a RETURNING device-scope atomic whose value this lane consumes: it has been performed at the coherence point before the
    // block's arrival below is issued - no release fence (on this chip that is an L2 write-back: the first version of this kernel, with
    // release / acquire on the barrier flag, took 110 us instead of 20)
    if (lane == 0) { const float was = atomicExch(a.n2part + b, s); asm volatile("" ::"v"(was) : "memory"); }
  }
  // ---- grid barrier (one per launch): the last block to arrive re-arms the counter and opens the next generation.
  // bar[1] is the generation AND the verdict: a launch that starts at the even value `gen` ends at gen + 2 (every block arrived) or at
  // gen + 1 (a block gave up waiting: the workgroups were not co-resident). Both transitions are a compare-and-swap from `gen`, so exactly
  // one of them happens and every block reads the same verdict. A FAILED barrier makes the whole step a no-op - no block writes flat, m, v,
  // the packings or the state (a step taken on an incomplete norm must not reach the parameters, let alone a checkpoint) - and
  // the odd generation is sticky: every later launch returns here without arriving, until nm_ppo_get_state has reported the failure and
  // reset the barrier (it then switches the handle to the four-launch step).
  __shared__ unsigned s_fail;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned fail = gen & 1u;
    if (!fail) {
      const unsigned old = atomicAdd(a.bar, 1u);
      if (old == gridDim.x - 1) {
        __hip_atomic_store(a.bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fail = atomicCAS(a.bar + 1, gen, gen + 2u) != gen;       // a block that timed out has decided first
      } else {
        int spins = 0;
        unsigned v;
        while ((v = __hip_atomic_load(a.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == gen) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1 << 22)) {      // never on an idle GPU; a hung barrier must not hang the device
            v = atomicCAS(a.bar + 1, gen, gen + 1u);
            if (v == gen) v = gen + 1u;   // this block's verdict stands; otherwise the last block arrived (or another one gave up) just before
            break;
          }
        }
        fail = v & 1u;
      }
    }
    if (fail) __hip_atomic_store(a.state + 8, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_fail = fail;
  }
  __syncthreads();
  if (s_fail) return;
  if (w0 != 0) return;
  // ---- the scalars of k_ppo_scalars, by every block in the same order
  float n2 = 0.0f, kl = 0.0f, su = 0.0f, vl = 0.0f;
  for (int j = lane; j < (int)gridDim.x; j += 64) n2 += __hip_atomic_load(a.n2part + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int w = lane; w < a.nwg; w += 64) {
    const float* P = a.partial + (size_t)w * a.stride + a.gtotal;
    kl += P[32]; su += P[33]; vl += P[34];
  }
  for (int o = 32; o > 0; o >>= 1) { n2 += __shfl_xor(n2, o); kl += __shfl_xor(kl, o); su += __shfl_xor(su, o); vl += __shfl_xor(vl, o); }
  float klm = kl * a.inv_B;
  if (a.kl_from_grad) klm = a.grad[n];
  if (a.kl_override >= 0.0f) klm = a.kl_override;
  float lr = lr0;
  if (a.adaptive) {                                    // rsl_rl v1.0.2 PPO.update: schedule == 'adaptive'
    if (klm > a.desired_kl * 2.0f) lr = fmaxf(1e-5f, lr / 1.5f);
    else if (klm < a.desired_kl / 2.0f && klm > 0.0f) lr = fminf(1e-2f, lr * 1.5f);
  }
  const float norm = sqrtf(n2), clipc = fminf(1.0f, a.max_norm / (norm + 1e-6f)), t = t0 + 1.0f;
  if (b == 0 && lane == 0) {
    a.state[0] = lr; a.state[1] = t; a.state[2] = klm; a.state[3] = s3 + vl * a.inv_B; a.state[4] = s4 + su * a.inv_B; a.state[5] = s5 + 1.0f;
    a.state[6] = clipc; a.state[7] = norm;
  }
  if (i < n) {     // k_ppo_adam + this parameter's two packed copies
    const float gc = g * clipc;
    const float mi = a.b1 * a.m[i] + (1.0f - a.b1) * gc, vi = a.b2 * a.v[i] + (1.0f - a.b2) * gc * gc;
    a.m[i] = mi; a.v[i] = vi;
    const float bc1 = 1.0f - powf(a.b1, t), bc2 = 1.0f - powf(a.b2, t);
    const float p = a.flat[i] - (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + a.eps);      // torch.optim.Adam (no amsgrad, no weight decay)
    a.flat[i] = p;
    if (mp >= 0) {
      a.Wm[mp] = p; a.pf[a.pfi[i]] = p; a.pb[a.pbi[i]] = p;
      if (a.sf) { a.sf[a.sfi[i]] = p; const int bi = a.sbi[i]; if (bi >= 0) a.sb[bi] = p; }
    }
  }
}

// A random permutation of 0..n-1 without a sort (rsl_rl draws torch.randperm per update: storage/rollout_storage.py mini_batch_generator):
// a 4-round Feistel network on the next even power of two, cycle-walked back into [0, n) - a bijection for every key. out[i] = image of i.
__device__ __forceinline__ uint32_t perm_mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
__global__ void k_ppo_perm(int* __restrict__ out, int n, int half, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t mask = (1u << half) - 1u, key[4] = {k0, k1, k2, k3};
  uint32_t x = (uint32_t)i;
  do {
    uint32_t L = x >> half, R = x & mask;
#pragma unroll
    for (int r = 0; r < 4; r++) { const uint32_t F = perm_mix(R * 0x9E3779B1u + key[r]) & mask; const uint32_t nr = L ^ F; L = R; R = nr; }
    x = (L << half) | R;
  } while (x >= (uint32_t)n);
  out[i] = (int)x;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ handle + C ABI
struct nm_ppo {
  int device = 0, n_layers = 0, A = 0, nparam = 0, nwg = 0, wm_total = 0;
  int nrows = 0;               // partial rows the last forward / backward launch wrote (<= nwg): what the reductions add up
  bool fast = false;           // the network has the shape k_ppo_fwdbwd_split / k_ppo_act_fast are compiled for
  float *sf = nullptr, *sb = nullptr;       // split kernel: both nets' forward / dX fragments (actor first)
  int *sfi = nullptr, *sbi = nullptr;       // per flat parameter: its float position in sf / sb (-1: none)
  PpoNet net;
  std::vector<int> map_host;
  int* map = nullptr;
  float* Wm = nullptr;         // merged matrices, all layers
  f32x4 *pf = nullptr, *pb = nullptr;
  float *partial = nullptr, *grad = nullptr, *state = nullptr;
  float* grad_ext = nullptr;   // caller-owned gradient | KL buffer (nm_ppo_set_grad_buffer): what a data-parallel update all-reduces in place
  int *pfi = nullptr, *pbi = nullptr;       // per flat parameter: its float position in pf / pb (k_ppo_step writes the packings itself)
  float* n2part = nullptr;
  unsigned* bar = nullptr;
  bool fused_step = true;                   // NM_PPO_UNFUSED_STEP=1: the four launches (reduce, scalars, adam, pack) - A/B timing and tests
  int64_t storage_rows = 0;                 // rows of the [T*N, .] arrays nm_ppo_minibatch_rows gathers from (nm_ppo_set_storage_rows)
  int step_grid_capacity = 0;               // workgroups of k_ppo_step the device holds at once (occupancy query at creation)
  std::vector<size_t> pf_off, pb_off;
};

#ifdef NM_PPO_STAMPS
extern "C" int nm_ppo_read_stamps(unsigned long long* out16, int clear) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ppo_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (clear) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ppo_stamps), z, sizeof z) != hipSuccess) return 1; }
  return 0;
}
#endif
#define PPO_CHK(x) do { if ((x) != hipSuccess) return nm_policy_set_error("nm_ppo: " #x " failed"); } while (0)

extern "C" int nm_ppo_destroy(nm_ppo* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : {(void*)h->map, (void*)h->Wm, (void*)h->pf, (void*)h->pb, (void*)h->partial, (void*)h->grad, (void*)h->state, (void*)h->pfi, (void*)h->pbi, (void*)h->n2part,
                  (void*)h->bar, (void*)h->sf, (void*)h->sb, (void*)h->sfi, (void*)h->sbi})
    if (p) (void)hipFree(p);
  delete h;
  return 0;
}

// actor_dims / critic_dims: {n_obs, h1, ..., n_out} with the same number of layers and the same input; critic output 1.
// Flat parameter order: actor W0 b0 W1 b1 ..., critic W0 b0 ..., std[A] (W row-major [out, in] as torch.nn.Linear).
extern "C" int nm_ppo_create(const int32_t* actor_dims, const int32_t* critic_dims, int32_t n_layers, int32_t device, nm_ppo** out) {
  if (!out) return nm_policy_set_error("nm_ppo_create: out is NULL");
  *out = nullptr;
  if (!actor_dims || !critic_dims || n_layers < 1 || n_layers > kL) return nm_policy_set_error("nm_ppo_create: 1..4 layers");
  if (actor_dims[0] != critic_dims[0] || critic_dims[n_layers] != 1 || actor_dims[n_layers] > kMaxA || actor_dims[n_layers] < 1)
    return nm_policy_set_error("nm_ppo_create: actor and critic must share the observation, critic output 1, at most 32 actions");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return nm_policy_set_error("nm_ppo_create: no such HIP device");
  PPO_CHK(hipSetDevice(device));
  nm_ppo* h = new nm_ppo();
  h->device = device; h->n_layers = n_layers; h->A = actor_dims[n_layers];
  PpoNet& n = h->net;
  n.n_layers = n_layers; n.A = h->A;
  int goff = 0;
  for (int l = 0; l < n_layers; l++) {
    n.Kr[l] = l == 0 ? actor_dims[0] : actor_dims[l] + critic_dims[l];
    n.Or[l] = actor_dims[l + 1] + critic_dims[l + 1];
    n.Kp[l] = (n.Kr[l] + 1 + 15) & ~15;
    n.Op[l] = (n.Or[l] + 15) & ~15;
    if (l + 1 < n_layers) n.Op[l] = (n.Or[l] + 1 + 15) & ~15;      // room for the 1-column of the next layer's input
    if (n.Kp[l] > kW || n.Op[l] > kW) { delete h; return nm_policy_set_error("nm_ppo_create: merged layers wider than 128 are not supported"); }
    n.goff[l] = goff;
    goff += n.Op[l] * n.Kp[l];
  }
  for (int l = 0; l + 1 < n_layers; l++)
    if (n.Op[l] != n.Kp[l + 1]) n.Op[l] = n.Kp[l + 1] = std::max(n.Op[l], n.Kp[l + 1]);
  goff = 0;
  for (int l = 0; l < n_layers; l++) { n.goff[l] = goff; goff += n.Op[l] * n.Kp[l]; }
  n.gtotal = goff; h->wm_total = goff;
  // dW tiles -> waves
  for (int w = 0; w < kWaves; w++) for (int s = 0; s < kSlots; s++) n.slot[w][s] = -1;
  {
    int g = 0, cnt[kWaves] = {0};
    for (int l = 0; l < n_layers; l++)
      for (int to = 0; to < n.Op[l] / 16; to++)
        for (int tk = 0; tk < n.Kp[l] / 16; tk++, g++) {
          const int w = g % kWaves;
          if (cnt[w] >= kSlots) { delete h; return nm_policy_set_error("nm_ppo_create: network too large (dW tiles per wave)"); }
          n.slot[w][cnt[w]++] = l | (to << 4) | (tk << 8);
        }
  }
  // the compiled fast path (NM_PPO_GENERIC=1 keeps such a network on the generic kernel: tests, A/B timing)
  {
    typedef RefShape S;
    bool same = n_layers == S::NL && !(std::getenv("NM_PPO_GENERIC") && std::atoi(std::getenv("NM_PPO_GENERIC")) != 0);
    for (int l = 0; same && l < S::NL; l++)
      same = actor_dims[l] == S::ain(l) && critic_dims[l] == S::cin(l) && actor_dims[l + 1] == S::aout(l) && critic_dims[l + 1] == S::cout(l) &&
             n.Kp[l] == S::P(l) && n.Op[l] == S::P(l + 1);
    h->fast = same;
    for (int w = 0; w < 4; w++) for (int i = 0; i < 16; i++) n.slot4[w][i] = -1;
    n.sf[0] = n.sf[1] = n.sb[0] = n.sb[1] = nullptr;
    if (same) {
      typedef Split<S> X;
      for (int l = 0; l < S::NL; l++)      // a net's tiles of layer l in (o tile, k tile) order, a contiguous run per row-group wave
        for (int idx = 0; idx < X::nto(l) * X::nkt(l); idx++) n.slot4[idx / X::slots(l)][X::slotbase(l) + idx % X::slots(l)] = (idx / X::nkt(l)) | ((idx % X::nkt(l)) << 4);
    }
  }
  // flat parameter -> merged position
  std::vector<int>& map = h->map_host;
  for (int net_i = 0; net_i < 2; net_i++) {
    const int32_t* d = net_i ? critic_dims : actor_dims;
    for (int l = 0; l < n_layers; l++) {
      const int o0 = net_i ? actor_dims[l + 1] : 0, k0 = (net_i && l > 0) ? actor_dims[l] : 0;
      for (int o = 0; o < d[l + 1]; o++)
        for (int k = 0; k < d[l]; k++) map.push_back(n.goff[l] + (o0 + o) * n.Kp[l] + k0 + k);
      for (int o = 0; o < d[l + 1]; o++) map.push_back(n.goff[l] + (o0 + o) * n.Kp[l] + n.Kr[l]);
    }
  }
  for (int j = 0; j < h->A; j++) map.push_back(-(j + 1));
  h->nparam = (int)map.size();
  {   // where each net's layers start in the flat vector (the order of the loop above)
    int at = 0;
    for (int net_i = 0; net_i < 2; net_i++) {
      const int32_t* d = net_i ? critic_dims : actor_dims;
      for (int l = 0; l < n_layers; l++) { n.woff[net_i][l] = at; at += d[l + 1] * d[l]; n.boff[net_i][l] = at; at += d[l + 1]; }
    }
    n.nparam_flat = h->nparam;
  }
  // from here on a failure must release the handle (and what it already owns)
#define PPO_CHK_H(x) do { if ((x) != hipSuccess) { nm_ppo_destroy(h); return nm_policy_set_error("nm_ppo_create: " #x " failed"); } } while (0)
  hipDeviceProp_t prop;
  PPO_CHK_H(hipGetDeviceProperties(&prop, device));
  h->nwg = prop.multiProcessorCount;
  size_t pft = 0, pbt = 0;
  for (int l = 0; l < n_layers; l++) {
    h->pf_off.push_back(pft); h->pb_off.push_back(pbt);
    pft += (size_t)(n.Op[l] / 16) * (n.Kp[l] / 16) * 64;
    pbt += (size_t)(n.Kp[l] / 16) * (n.Op[l] / 16) * 64;
  }
  bool ok = hipMalloc((void**)&h->map, map.size() * sizeof(int)) == hipSuccess && hipMalloc((void**)&h->Wm, (size_t)h->wm_total * sizeof(float)) == hipSuccess &&
            hipMalloc((void**)&h->pf, pft * sizeof(f32x4)) == hipSuccess && hipMalloc((void**)&h->pb, pbt * sizeof(f32x4)) == hipSuccess &&
            hipMalloc((void**)&h->partial, (size_t)h->nwg * (n.gtotal + kNS) * sizeof(float)) == hipSuccess &&
            hipMalloc((void**)&h->grad, (map.size() + 1) * sizeof(float)) == hipSuccess && hipMalloc((void**)&h->state, 12 * sizeof(float)) == hipSuccess &&
            hipMalloc((void**)&h->pfi, map.size() * sizeof(int)) == hipSuccess && hipMalloc((void**)&h->pbi, map.size() * sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&h->n2part, ((map.size() + 1 + kRedParams - 1) / kRedParams) * sizeof(float)) == hipSuccess && hipMalloc((void**)&h->bar, 2 * sizeof(unsigned)) == hipSuccess;
  if (!ok) { nm_ppo_destroy(h); return nm_policy_set_error("nm_ppo_create: hipMalloc failed"); }
  PPO_CHK_H(hipMemcpy(h->map, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
  PPO_CHK_H(hipMemset(h->Wm, 0, (size_t)h->wm_total * sizeof(float)));
  PPO_CHK_H(hipMemset(h->state, 0, 12 * sizeof(float)));
  PPO_CHK_H(hipMemset(h->bar, 0, 2 * sizeof(unsigned)));
  {   // where parameter i sits in the forward / backward packing (the layouts of k_ppo_pack), as float offsets into pf / pb
    std::vector<int> pfi(map.size(), -1), pbi(map.size(), -1);
    for (size_t i = 0; i < map.size(); i++) {
      const int m = map[i];
      if (m < 0) continue;
      int l = 0;
      while (l + 1 < n_layers && m >= n.goff[l + 1]) l++;
      const int o = (m - n.goff[l]) / n.Kp[l], k = (m - n.goff[l]) % n.Kp[l];
      int t, g, q, j, r;
      if (h->fast) { t = o / 16; r = o % 16; g = k / 16; q = (k % 16) / 4; j = k % 4; }
      else { t = o / 16; r = o % 16; g = k / 16; j = (k % 16) / 4; q = k % 4; }
      pfi[i] = (int)((h->pf_off[l] + ((size_t)t * (n.Kp[l] / 16) + g) * 64 + q * 16 + r) * 4 + j);
      if (h->fast) { t = k / 16; r = k % 16; g = o / 16; q = (o % 16) / 4; j = o % 4; }
      else { t = k / 16; r = k % 16; g = o / 16; j = (o % 16) / 4; q = o % 4; }
      pbi[i] = (int)((h->pb_off[l] + ((size_t)t * (n.Op[l] / 16) + g) * 64 + q * 16 + r) * 4 + j);
    }
    PPO_CHK_H(hipMemcpy(h->pfi, pfi.data(), pfi.size() * sizeof(int), hipMemcpyHostToDevice));
    PPO_CHK_H(hipMemcpy(h->pbi, pbi.data(), pbi.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  if (h->fast) {   // the split kernel's packings: per net its own [out x in + 1] matrices, fragments in the order a wave consumes them
    typedef Split<RefShape> X;
    const size_t nfw = (size_t)X::nf() * 64 * 4, nbw = (size_t)X::nb() * 64 * 4;      // floats per net
    std::vector<int> sfi(map.size(), -1), sbi(map.size(), -1);
    size_t i = 0;
    for (int net_i = 0; net_i < 2; net_i++) {
      const int32_t* d = net_i ? critic_dims : actor_dims;
      for (int l = 0; l < n_layers; l++) {
        auto place = [&](int o, int k) {      // element (o, k) of the net's layer-l matrix (k = d[l]: the bias column)
          { const int t = o / 16, r = o % 16, g = k / 16, q = (k % 16) / 4, j = k % 4;      // forward: lane (r, q) of fragment (t, g), component j
            sfi[i] = (int)(net_i * nfw + ((size_t)X::fidx(l, t, g) * 64 + q * 16 + r) * 4 + j); }
          if (l > 0) { const int t = k / 16, r = k % 16, g = o / 16, q = (o % 16) / 4, j = o % 4;      // dX: lane (r, q) of fragment (k tile t, o tile g)
            sbi[i] = (int)(net_i * nbw + ((size_t)X::bidx(l, g, t) * 64 + q * 16 + r) * 4 + j); }
          i++;
        };
        for (int o = 0; o < d[l + 1]; o++) for (int k = 0; k < d[l]; k++) place(o, k);
        for (int o = 0; o < d[l + 1]; o++) place(o, d[l]);
      }
    }
    bool ok2 = hipMalloc((void**)&h->sf, 2 * nfw * sizeof(float)) == hipSuccess && hipMalloc((void**)&h->sb, 2 * nbw * sizeof(float)) == hipSuccess &&
               hipMalloc((void**)&h->sfi, map.size() * sizeof(int)) == hipSuccess && hipMalloc((void**)&h->sbi, map.size() * sizeof(int)) == hipSuccess;
    if (!ok2) { nm_ppo_destroy(h); return nm_policy_set_error("nm_ppo_create: hipMalloc failed"); }
    PPO_CHK_H(hipMemset(h->sf, 0, 2 * nfw * sizeof(float)));
    PPO_CHK_H(hipMemset(h->sb, 0, 2 * nbw * sizeof(float)));
    PPO_CHK_H(hipMemcpy(h->sfi, sfi.data(), sfi.size() * sizeof(int), hipMemcpyHostToDevice));
    PPO_CHK_H(hipMemcpy(h->sbi, sbi.data(), sbi.size() * sizeof(int), hipMemcpyHostToDevice));
    for (int net_i = 0; net_i < 2; net_i++) { n.sf[net_i] = reinterpret_cast<const f32x4*>(h->sf + net_i * nfw); n.sb[net_i] = reinterpret_cast<const f32x4*>(h->sb + net_i * nbw); }
  }
  h->fused_step = !(std::getenv("NM_PPO_UNFUSED_STEP") && std::atoi(std::getenv("NM_PPO_UNFUSED_STEP")) != 0);
  if (h->fused_step) {      // the grid barrier of k_ppo_step needs every workgroup of its grid resident at once: ask, do not assume
    int per_cu = 0, cus = 0;
    const int nbr = (h->nparam + 1 + kRedParams - 1) / kRedParams;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_ppo_step, kRedParams * kRedWaves, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || (long)per_cu * cus < nbr)
      h->fused_step = false;
    h->step_grid_capacity = per_cu * cus;
  }
  PPO_CHK_H(hipMemset(h->partial, 0, (size_t)h->nwg * (n.gtotal + kNS) * sizeof(float)));
#undef PPO_CHK_H
  for (int l = 0; l < n_layers; l++) { n.pf[l] = h->pf + h->pf_off[l]; n.pb[l] = h->pb + h->pb_off[l]; }
  *out = h;
  return 0;
}
extern "C" int32_t nm_ppo_num_params(const nm_ppo* h) { return h ? h->nparam : 0; }
static inline float* ppo_grad(nm_ppo* h) { return h->grad_ext ? h->grad_ext : h->grad; }
// The gradient | mean-KL vector of a mini-batch ([num_params + 1] floats, device) in a buffer the CALLER owns: phase 1 of nm_ppo_minibatch
// writes it there, phase 2 reads it from there - a data-parallel update all-reduces that buffer in place (one collective per mini-batch,
// no copies around it: nm_ppo_copy_grad is then not needed). NULL returns to the handle's own buffer.
extern "C" int nm_ppo_set_grad_buffer(nm_ppo* h, float* grad_kl_dev) {
  if (!h) return nm_policy_set_error("nm_ppo_set_grad_buffer: bad argument");
  h->grad_ext = grad_kl_dev;
  return 0;
}

static int ppo_pack(nm_ppo* h, hipStream_t s) {
  int most = 0;
  for (int l = 0; l < h->n_layers; l++) most = std::max(most, 2 * (h->net.Op[l] / 16) * (h->net.Kp[l] / 16) * 64);
  hipLaunchKernelGGL(k_ppo_pack, dim3((most + 255) / 256, h->n_layers), dim3(256), 0, s, h->net, h->Wm, h->fast ? 1 : 0);
  if (h->sf) hipLaunchKernelGGL(k_ppo_pack_split, dim3((h->nparam + 255) / 256), dim3(256), 0, s, h->Wm, h->map, h->sfi, h->sbi, h->nparam, h->sf, h->sb);
  return hipGetLastError() == hipSuccess ? 0 : nm_policy_set_error("nm_ppo: pack launch failed");
}
// (re)load the parameters from the flat vector (after load_state_dict, or a step taken elsewhere) and set learning rate / step count
extern "C" int nm_ppo_sync_params(nm_ppo* h, const float* flat_dev, float lr, int64_t step, void* stream) {
  if (!h || !flat_dev) return nm_policy_set_error("nm_ppo_sync_params: bad argument");
  PPO_CHK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ppo_scatter, dim3((h->nparam + 255) / 256), dim3(256), 0, s, flat_dev, h->map, h->nparam, h->Wm);
  const float st[12] = {lr, (float)step, 0, 0, 0, 0, 1.0f, 0, 0, 0, 0, 0};
  PPO_CHK(hipMemcpyAsync(h->state, st, sizeof st, hipMemcpyHostToDevice, s));
  PPO_CHK(hipStreamSynchronize(s));      // st lives on this stack frame
  return ppo_pack(h, s);
}
// one mini-batch of PPO.update: forward, losses, backward, (optional: stop after the gradient for a multi-GPU all-reduce), clip, lr, Adam
extern "C" int nm_ppo_minibatch_rows(nm_ppo* h, float* flat_dev, float* exp_avg_dev, float* exp_avg_sq_dev, const float* obs, const float* actions,
                                     const float* old_mu, const float* old_sigma, const float* old_logp, const float* adv, const float* ret, const float* tval,
                                     const int32_t* rows_dev, int32_t B, int32_t n_obs, float clip, float value_coef, float entropy_coef, int32_t clip_value, float desired_kl,
                                     int32_t adaptive, float max_grad_norm, float beta1, float beta2, float eps, int32_t phase, float kl_override, void* stream);
extern "C" int nm_ppo_minibatch(nm_ppo* h, float* flat_dev, float* exp_avg_dev, float* exp_avg_sq_dev, const float* obs, const float* actions,
                                const float* old_mu, const float* old_sigma, const float* old_logp, const float* adv, const float* ret, const float* tval,
                                int32_t B, int32_t n_obs, float clip, float value_coef, float entropy_coef, int32_t clip_value, float desired_kl,
                                int32_t adaptive, float max_grad_norm, float beta1, float beta2, float eps, int32_t phase, float kl_override, void* stream) {
  return nm_ppo_minibatch_rows(h, flat_dev, exp_avg_dev, exp_avg_sq_dev, obs, actions, old_mu, old_sigma, old_logp, adv, ret, tval, nullptr, B, n_obs, clip, value_coef,
                               entropy_coef, clip_value, desired_kl, adaptive, max_grad_norm, beta1, beta2, eps, phase, kl_override, stream);
}
// rows of the arrays a row list indexes: row numbers are in [0, n_rows); n_rows * max(n_obs, n_actions) * 4 must stay below 2^32
extern "C" int nm_ppo_set_storage_rows(nm_ppo* h, int64_t n_rows) {
  if (!h || n_rows <= 0) return nm_policy_set_error("nm_ppo_set_storage_rows: bad argument");
  if (n_rows * std::max(h->net.Kr[0], h->A) * 4 >= ((int64_t)1 << 32))
    return nm_policy_set_error("nm_ppo_set_storage_rows: the in-kernel row gather uses 32-bit byte offsets: n_rows * max(n_obs, n_actions) * 4 must stay below 2^32");
  h->storage_rows = n_rows;
  return 0;
}
// the same with the mini-batch given as row numbers into the (unpermuted) arrays: row i of the mini-batch is row rows_dev[i] - the gather
// of rsl_rl's mini_batch_generator (obs[batch_idx], ...) happens inside the forward / backward kernel. rows_dev == NULL: rows 0..B-1.
extern "C" int nm_ppo_minibatch_rows(nm_ppo* h, float* flat_dev, float* exp_avg_dev, float* exp_avg_sq_dev, const float* obs, const float* actions,
                                     const float* old_mu, const float* old_sigma, const float* old_logp, const float* adv, const float* ret, const float* tval,
                                     const int32_t* rows_dev, int32_t B, int32_t n_obs, float clip, float value_coef, float entropy_coef, int32_t clip_value, float desired_kl,
                                     int32_t adaptive, float max_grad_norm, float beta1, float beta2, float eps, int32_t phase, float kl_override, void* stream) {
  if (!h || !flat_dev || !exp_avg_dev || !exp_avg_sq_dev || !obs || !actions || !old_mu || !old_sigma || !old_logp || !adv || !ret || !tval || B <= 0)
    return nm_policy_set_error("nm_ppo_minibatch: bad argument");
  if (n_obs != h->net.Kr[0]) return nm_policy_set_error("nm_ppo_minibatch: observation width does not match the network");
  {  // the kernels address a row as a uniform base + a 32-bit BYTE offset per lane (row * width * 4): the arrays a row number can reach
     // must stay under 4 GiB - B rows without a row list, the declared storage (nm_ppo_set_storage_rows) with one
    const int64_t reach = rows_dev ? h->storage_rows : (int64_t)B, widest = std::max(n_obs, h->A);
    if (rows_dev && h->storage_rows <= 0) return nm_policy_set_error("nm_ppo_minibatch_rows: declare the row count of the arrays with nm_ppo_set_storage_rows first");
    if (reach * widest * 4 >= ((int64_t)1 << 32)) return nm_policy_set_error("nm_ppo_minibatch: more than 2^32 bytes per array (32-bit row offsets): split the storage");
  }
  PPO_CHK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const int stride = h->net.gtotal + kNS, nb = (h->nparam + 255) / 256, nbr = (h->nparam + 1 + kRedParams - 1) / kRedParams;
  if (phase == 0 || phase == 1) {     // 1: gradient only
    PpoBatch bt{obs, actions, old_mu, old_sigma, old_logp, adv, ret, tval, flat_dev + (h->nparam - h->A), B, n_obs, clip, value_coef, 1.0f / (float)B, clip_value, rows_dev};
    const int rows = h->fast ? 16 * kSplitWaves : kRows;
    const int ntiles = (B + rows - 1) / rows, grid = ntiles < h->nwg ? ntiles : h->nwg;
    // the reductions add the `grid` rows this launch writes and no others (the rows beyond used to be cleared by a 28 MB memset per
    // mini-batch; inside the update's captured graph that memset node corrupted the loss sums of small batches: grid < number of CUs)
    h->nrows = grid;
    if (h->fast) hipLaunchKernelGGL(k_ppo_fwdbwd_split<RefShape>, dim3(2 * grid), dim3(64 * kSplitWaves), 0, s, h->net, bt, h->partial);      // two blocks (actor, critic) per partial row
    else hipLaunchKernelGGL(k_ppo_fwdbwd, dim3(grid), dim3(kThreads), 0, s, h->net, bt, h->partial);
    if (phase == 1 || !h->fused_step)
      hipLaunchKernelGGL(k_ppo_reduce, dim3(nbr), dim3(kRedParams * kRedWaves), 0, s, h->partial, h->nrows, stride, h->map, h->nparam, h->net.gtotal, flat_dev, entropy_coef, 1.0f / (float)B, ppo_grad(h), h->fast ? 1 : 0);
  }
  if ((phase == 0 || phase == 2) && h->fused_step) {     // reduce (phase 0) + scalars + Adam + both packings: one launch
    StepArgs a;
    a.partial = h->partial; a.nwg = h->nrows; a.stride = stride; a.gtotal = h->net.gtotal; a.map = h->map; a.n = h->nparam;
    a.flat = flat_dev; a.m = exp_avg_dev; a.v = exp_avg_sq_dev; a.grad = ppo_grad(h); a.state = h->state; a.Wm = h->Wm; a.n2part = h->n2part;
    a.pf = reinterpret_cast<float*>(h->pf); a.pb = reinterpret_cast<float*>(h->pb); a.pfi = h->pfi; a.pbi = h->pbi; a.bar = h->bar;
    a.sf = h->sf; a.sb = h->sb; a.sfi = h->sfi; a.sbi = h->sbi;
    a.compact = h->fast ? 1 : 0;
    a.ent_coef = entropy_coef; a.inv_B = 1.0f / (float)B; a.desired_kl = desired_kl; a.max_norm = max_grad_norm; a.kl_override = kl_override;
    a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.adaptive = adaptive; a.kl_from_grad = phase == 2 ? 1 : 0; a.do_reduce = phase == 0 ? 1 : 0;
    hipLaunchKernelGGL(k_ppo_step, dim3(nbr), dim3(kRedParams * kRedWaves), 0, s, a);
  } else if (phase == 0 || phase == 2) {     // 2: the step, after the caller has all-reduced gradient | KL; the KL is then read from there
    hipLaunchKernelGGL(k_ppo_scalars, dim3(1), dim3(1024), 0, s, h->partial, h->nrows, stride, h->net.gtotal, ppo_grad(h), h->nparam, 1.0f / (float)B, desired_kl,
                       adaptive, max_grad_norm, kl_override, phase == 2 ? 1 : 0, h->state);
    hipLaunchKernelGGL(k_ppo_adam, dim3(nb), dim3(256), 0, s, flat_dev, exp_avg_dev, exp_avg_sq_dev, ppo_grad(h), h->nparam, h->state, beta1, beta2, eps, h->map, h->Wm);
    if (ppo_pack(h, s)) return 1;
  }
  if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_ppo_minibatch: launch failed");
  return 0;
}
// does this network run on the compiled fast kernels (k_ppo_fwdbwd_split, k_ppo_act_fast)?
extern "C" int32_t nm_ppo_has_fast_path(const nm_ppo* h) { return h && h->fast ? 1 : 0; }
// PPO.act in one launch (fast-path networks only): forward of the merged network from the update's packed weights, sampling,
// log-probability, value, and the rollout-storage writes of nm_ppo_sample (same generator and keys)
extern "C" int nm_ppo_act(nm_ppo* h, const float* flat_dev, const float* obs, int32_t N, uint64_t seed, const int64_t* iter_dev, int32_t step,
                          float* actions, float* logp, float* values, float* mu, float* sigma, float* obs_store, void* stream) {
  if (!h || !flat_dev || !obs || !iter_dev || !actions || !logp || !values || !mu || !sigma || N <= 0) return nm_policy_set_error("nm_ppo_act: bad argument");
  if (!h->fast) return nm_policy_set_error("nm_ppo_act: this network shape has no compiled fast path (use nm_policy_forward + nm_ppo_sample)");
  PPO_CHK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_ppo_act_fast<RefShape>, dim3((N + 15) / 16), dim3(64), 0, (hipStream_t)stream, h->net, obs, flat_dev + (h->nparam - h->A), N, seed, iter_dev,
                     step, actions, logp, values, mu, sigma, obs_store, PpoRecord{});
  if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_ppo_act: launch failed");
  return 0;
}
// nm_ppo_record of the previous step and nm_ppo_act of this one in ONE launch (the record part first; same arguments and meaning as the
// two calls): a rollout of T steps is T + 1 launches besides the env's instead of 2 T
extern "C" int nm_ppo_record_act(nm_ppo* h, const float* rew, const int64_t* done, const float* time_outs, const float* prev_values, float gamma,
                                 float* rewards_store, unsigned char* dones_store, float* cur_ret, float* cur_len, float* fin3,
                                 const float* ep_stats, const int32_t* ep_idx, int32_t n_ep, float* ep_acc,
                                 const float* flat_dev, const float* obs, int32_t N, uint64_t seed, const int64_t* iter_dev, int32_t step,
                                 float* actions, float* logp, float* values, float* mu, float* sigma, float* obs_store, void* stream) {
  if (!h || !flat_dev || !obs || !iter_dev || !actions || !logp || !values || !mu || !sigma || N <= 0) return nm_policy_set_error("nm_ppo_record_act: bad argument");
  if (!rew || !done || !prev_values || !rewards_store || !dones_store || !cur_ret || !cur_len || !fin3) return nm_policy_set_error("nm_ppo_record_act: bad record argument");
  if (n_ep < 0 || n_ep > 256 || (n_ep > 0 && (!ep_stats || !ep_idx || !ep_acc))) return nm_policy_set_error("nm_ppo_record_act: bad episode-statistics arguments");
  if (!h->fast) return nm_policy_set_error("nm_ppo_record_act: this network shape has no compiled fast path");
  PPO_CHK(hipSetDevice(h->device));
  PpoRecord rec{rew, done, time_outs, prev_values, gamma, rewards_store, dones_store, cur_ret, cur_len, fin3, ep_stats, ep_idx, n_ep, ep_acc};
  hipLaunchKernelGGL(k_ppo_act_fast<RefShape>, dim3((N + 15) / 16), dim3(64), 0, (hipStream_t)stream, h->net, obs, flat_dev + (h->nparam - h->A), N, seed, iter_dev,
                     step, actions, logp, values, mu, sigma, obs_store, rec);
  if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_ppo_record_act: launch failed");
  return 0;
}
// gradient of the last mini-batch in flat order followed by its mean KL, nparam + 1 floats: direction 0 copies them into grad_dev,
// 1 replaces them by grad_dev (after an all-reduce and the division by the number of ranks)
extern "C" int nm_ppo_copy_grad(nm_ppo* h, float* grad_dev, int32_t direction, void* stream) {
  if (!h || !grad_dev) return nm_policy_set_error("nm_ppo_copy_grad: bad argument");
  PPO_CHK(hipSetDevice(h->device));
  PPO_CHK(hipMemcpyAsync(direction ? ppo_grad(h) : grad_dev, direction ? grad_dev : ppo_grad(h), ((size_t)h->nparam + 1) * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
// out_dev[i] = image of i under a random permutation of 0..n-1 keyed by (seed, counter): the mini-batch order of one PPO.update
// (rsl_rl v1.0.2 storage/rollout_storage.py mini_batch_generator: torch.randperm) without a sort or any library kernel
extern "C" int nm_ppo_permutation(int32_t* out_dev, int32_t n, uint64_t seed, uint64_t counter, void* stream) {
  if (!out_dev || n <= 0) return nm_policy_set_error("nm_ppo_permutation: bad argument");
  int bits = 2;
  while ((1ll << bits) < (long long)n) bits += 2;          // an even number of bits: two equal halves
  uint32_t key[4];
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + counter * 0xD1B54A32D192ED03ull + 0x2545F4914F6CDD1Dull;
  for (int r = 0; r < 4; r++) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; key[r] = (uint32_t)(x >> 16); x += 0x9E3779B97F4A7C15ull; }
  hipLaunchKernelGGL(k_ppo_perm, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, out_dev, n, bits / 2, key[0], key[1], key[2], key[3]);
  if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_ppo_permutation: launch failed");
  return 0;
}
// HOST out[8]: lr, Adam steps, last KL, sum of value losses, sum of surrogate losses, mini-batches, clip coefficient, gradient norm.
// reset_sums != 0 clears the loss sums and the mini-batch count afterwards. Synchronises the stream.
// the same nine values (out[8] != 0: k_ppo_step's grid barrier timed out) copied to DEVICE memory, stream-ordered and without a host
// synchronisation: the caller reads them (e.g. through a pinned host copy) whenever it likes - the runner does so one iteration later, while
// the next rollout is already running
// after a failed grid barrier: re-arm it (arrivals 0, even generation), clear the flag, and keep away from the fused step for good
static int ppo_reset_barrier(nm_ppo* h, hipStream_t s) {
  PPO_CHK(hipMemsetAsync(h->bar, 0, 2 * sizeof(unsigned), s));
  PPO_CHK(hipMemsetAsync(h->state + 8, 0, sizeof(float), s));
  h->fused_step = false;
  return 0;
}
// 1 if this handle's mini-batch step is the one-launch k_ppo_step, 0 if it is the four-launch chain (asked for, the grid would not be
// co-resident, or a barrier has failed)
extern "C" int32_t nm_ppo_step_is_fused(const nm_ppo* h) { return h && h->fused_step ? 1 : 0; }
// TEST HOOK: make the NEXT fused step's barrier time out (the arrival count is moved out of reach), to prove the no-op behaviour on a GPU
// that would otherwise never show it. Costs that launch ~1 s of spinning.
extern "C" int nm_ppo_debug_break_barrier(nm_ppo* h, void* stream) {
  if (!h) return nm_policy_set_error("nm_ppo_debug_break_barrier: bad argument");
  PPO_CHK(hipSetDevice(h->device));
  const unsigned far = 1u << 30;
  PPO_CHK(hipMemcpyAsync(h->bar, &far, sizeof(unsigned), hipMemcpyHostToDevice, (hipStream_t)stream));
  PPO_CHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
extern "C" int nm_ppo_snapshot_state(nm_ppo* h, float* out9_dev, int32_t reset_sums, void* stream) {
  if (!h || !out9_dev) return nm_policy_set_error("nm_ppo_snapshot_state: bad argument");
  PPO_CHK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  PPO_CHK(hipMemcpyAsync(out9_dev, h->state, 9 * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (reset_sums) PPO_CHK(hipMemsetAsync(h->state + 3, 0, 3 * sizeof(float), s));
  PPO_CHK(hipMemsetAsync(h->state + 8, 0, sizeof(float), s));
  return 0;
}
extern "C" int nm_ppo_get_state(nm_ppo* h, float* out8_host, int32_t reset_sums, void* stream) {
  if (!h || !out8_host) return nm_policy_set_error("nm_ppo_get_state: bad argument");
  PPO_CHK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  float st[9];
  PPO_CHK(hipMemcpyAsync(st, h->state, 9 * sizeof(float), hipMemcpyDeviceToHost, s));
  PPO_CHK(hipStreamSynchronize(s));
  for (int i = 0; i < 8; i++) out8_host[i] = st[i];
  if (reset_sums) PPO_CHK(hipMemsetAsync(h->state + 3, 0, 3 * sizeof(float), s));
  if (st[8] != 0.0f) {      // k_ppo_step's grid barrier gave up waiting (its blocks were not co-resident): that step and every one since was a no-op
    if (ppo_reset_barrier(h, s)) return 1;
    return nm_policy_set_error("nm_ppo: the grid barrier of the fused mini-batch step timed out (GPU shared with another job?); the mini-batch steps "
                               "since then were skipped (no parameter written); this handle now takes the four-launch step (NM_PPO_UNFUSED_STEP=1)");
  }
  return 0;
}
