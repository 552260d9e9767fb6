// nm_rollout.h - the policy inside the env's wavefront: rsl_rl v1.0.2 `PPO.act` (actor mean, Normal sample, log-probability, critic
// value, transition record) and `PPO.process_env_step` (time-out bootstrap, dones, episode bookkeeping) evaluated BY THE WAVE THAT OWNS
// THE ENV, between two physics steps - the rollout loop `act -> env.step -> process_env_step` that the reference drives from
// train.py:54 (OnPolicyRunner.learn, num_steps_per_env = 80: envs/nightmare_v3_config.py:135) without a kernel boundary per step.
//
// A wave owns two envs, so its policy batch is two rows. The matrix instruction that fits that is v_mfma_f32_4x4x1_16B_f32: sixteen
// independent 4x4 outer products per instruction. Block b of an instruction computes output neurons 4b..4b+3 (A operand: one weight
// per lane, lane = 4 b + i) for four batch columns (B operand: lane 4 b + j holds the input of column j; columns 0/1 = the wave's two
// envs, 2/3 duplicate them), accumulated over k in a 4-register tile per lane (register i, lane 4 b + j = neuron 4 b + i of column j).
// Because every block has its own B value, actor blocks read the actor's activations and critic blocks the critic's: the merged
// actor | critic network needs no block-diagonal zero tiles. 64 neurons x 1 k per instruction at 50 % column use - against 12.5 % for
// a 16x16x4 tile with two live columns. Weights stream from L2 in consumption order through a register ring (one 16-byte load per lane
// = the A operands of 4 k steps), activations go from layer to layer through 2 KB of the wave's LDS.
//
// Arithmetic: exact f32 FMA chains in k order, bias as the accumulator's initial value, ELU with the hardware exponential - the same
// network as nm_ppo_act (16x16x4 tiles), rounded in another order (agreement ~1e-6; tests/test_gpu_rollout.py states the tolerance).
// The action noise uses the generator and keys of nm_ppo_sample / nm_ppo_act (seed, iteration, step, env, action pair): the same z.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

namespace nmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

template <int N, class F, int... Is> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// Network shape, compile time: observation I, three hidden layers of the actor (A1..A3) and of the critic (C1..C3), AO actions, value 1.
// Layer l of the merged network: inputs ka(l) | kc(l) (layer 0: both read the observation), outputs padded to whole 4-neuron blocks:
// oa4(l) actor neurons first, then oc4(l) critic neurons.
template <int I_, int A1, int A2, int A3, int AO_, int C1, int C2, int C3>
struct Shape {
  static constexpr int NL = 4, I = I_, AO = AO_;
  static constexpr __host__ __device__ int up4(int x) { return (x + 3) & ~3; }
  static constexpr __host__ __device__ int ka(int l) { return l == 0 ? I_ : l == 1 ? A1 : l == 2 ? A2 : A3; }
  static constexpr __host__ __device__ int kc(int l) { return l == 0 ? I_ : l == 1 ? C1 : l == 2 ? C2 : C3; }
  static constexpr __host__ __device__ int aout(int l) { return l == 0 ? A1 : l == 1 ? A2 : l == 2 ? A3 : AO_; }
  static constexpr __host__ __device__ int cout(int l) { return l == 0 ? C1 : l == 1 ? C2 : l == 2 ? C3 : 1; }
  static constexpr __host__ __device__ int oa4(int l) { return up4(aout(l)); }
  static constexpr __host__ __device__ int oc4(int l) { return up4(cout(l)); }
  static constexpr __host__ __device__ int nblk(int l) { return (oa4(l) + oc4(l)) / 4; }
  static constexpr __host__ __device__ int ng(int l) { return (nblk(l) + 15) / 16; }                  // instructions per k step
  static constexpr __host__ __device__ int kq(int l) { return (up4(ka(l) > kc(l) ? ka(l) : kc(l))) / 4; }   // k quads
  static constexpr __host__ __device__ int foff(int l) { int n = 0; for (int i = 0; i < l; i++) n += ng(i) * kq(i); return n; }   // first fragment of layer l
  static constexpr __host__ __device__ int nfrag() { return foff(NL); }
  static constexpr __host__ __device__ int boff(int l) { int n = 0; for (int i = 0; i < l; i++) n += ng(i) * 64; return n; }       // first bias float of layer l
  static constexpr __host__ __device__ int nbias() { return boff(NL); }
  // flat parameter vector (rsl_rl ActorCritic order: actor W0 b0 W1 b1 ..., critic W0 b0 ..., std[AO]): offsets of W_l / b_l
  static constexpr __host__ __device__ int aw(int l) { int n = 0; for (int i = 0; i < l; i++) n += aout(i) * ka(i) + aout(i); return n; }
  static constexpr __host__ __device__ int ab(int l) { return aw(l) + aout(l) * ka(l); }
  static constexpr __host__ __device__ int cw(int l) { int n = aw(NL); for (int i = 0; i < l; i++) n += cout(i) * kc(i) + cout(i); return n; }
  static constexpr __host__ __device__ int cb(int l) { return cw(l) + cout(l) * kc(l); }
  static constexpr __host__ __device__ int stdoff() { return cw(NL); }
  static constexpr __host__ __device__ int nparam() { return stdoff() + AO_; }
  // weight of merged output neuron o (block order: actor, then critic), input k of layer l - index into the flat vector or -1 (padding)
  static constexpr __host__ __device__ int widx(int l, int o, int k) {
    if (o < oa4(l)) return (o < aout(l) && k < ka(l)) ? aw(l) + o * ka(l) + k : -1;
    const int oc = o - oa4(l);
    return (oc < cout(l) && k < kc(l)) ? cw(l) + oc * kc(l) + k : -1;
  }
  static constexpr __host__ __device__ int bidx(int l, int o) {
    if (o < oa4(l)) return o < aout(l) ? ab(l) + o : -1;
    const int oc = o - oa4(l);
    return oc < cout(l) ? cb(l) + oc : -1;
  }
  static_assert(A1 <= 64 && A2 <= 64 && A3 <= 64 && C1 <= 64 && C2 <= 64 && C3 <= 64 && I_ <= 124 && AO_ <= 32 && AO_ % 2 == 0, "shape limits of the LDS activation rows");
};
// the reference's networks (envs/nightmare_v3_config.py:107-109): 66 -> 54 -> 42 -> 30 -> 18 | 1
typedef Shape<66, 54, 42, 30, 18, 54, 42, 30> RefShape;

constexpr int kXW = 128;                 // floats per env in an activation row: actor part at 0, critic part at 64 (layer 0: the observation at 0)
constexpr int kXFloats = 2 * 2 * kXW;    // two rows (ping-pong) x two envs
constexpr int kRing = 10;                // weight fragments in flight per wave

// flat parameters -> packed: fragment f = foff(l) + q * ng(l) + g, lane, 4 floats = W(o = 4 (16 g + lane / 4) + lane % 4, k = 4 q + 0..3);
// bias of layer l, instruction g: boff(l) + g * 64 + (lane / 4) * 4 + i = bias of neuron 4 (16 g + lane / 4) + i
template <class S> __global__ void k_roll_pack(const float* __restrict__ flat, float* __restrict__ wp, float* __restrict__ bp) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < S::nfrag() * 256) {
    const int f = t >> 8, lane = (t >> 2) & 63, kk = t & 3;
    int l = 0;
    while (l + 1 < S::NL && f >= S::foff(l + 1)) l++;
    const int r = f - S::foff(l), q = r / S::ng(l), g = r % S::ng(l);
    const int o = 4 * (16 * g + lane / 4) + lane % 4, k = 4 * q + kk;
    const int i = o < S::oa4(l) + S::oc4(l) ? S::widx(l, o, k) : -1;
    wp[t] = i >= 0 ? flat[i] : 0.0f;
  }
  if (t < S::nbias()) {
    int l = 0;
    while (l + 1 < S::NL && t >= S::boff(l + 1)) l++;
    const int r = t - S::boff(l), g = r / 64, o = 64 * g + (r & 63);
    const int i = o < S::oa4(l) + S::oc4(l) ? S::bidx(l, o) : -1;
    bp[t] = i >= 0 ? flat[i] : 0.0f;
  }
}

__device__ __forceinline__ float u24(uint64_t seed, uint64_t a, uint64_t b) {   // = u24 of nm_rl.hip / ppo_u24 of nm_ppo.hip
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * (a + 1) + 0xD1B54A32D192ED03ull * b;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return ((float)(uint32_t)(x >> 40) + 1.0f) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ void roll_sync() {     // LDS traffic of one wave is executed in issue order: keep the compiler's order, nothing else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// What one collection step files (rsl_rl RolloutStorage.add_transitions): rows of step s, all [N, .] device pointers
struct ActOut {
  float *actions, *logp, *values, *mu, *sigma;
  float* obs_store;      // a copy of the observation the policy saw, or null (the rollout kernel's step writes it there itself)
};

// PPO.act for the wave's two envs (env = 2 wave + column). xb: kXFloats floats of the wave's LDS. obs: [N, I] device memory that this wave
// may have written itself a moment ago (the caller has waited for those stores). `env_id0` = global id of env 0 (noise key).
// VALUE_ONLY: the same forward, but nothing is sampled or filed except the critic's value into o.values[env] (the last observation of a rollout).
template <class S, bool VALUE_ONLY = false>
__device__ __forceinline__ void policy_wave(float* xb, const f32x4* __restrict__ wp, const float* __restrict__ bp, const float* __restrict__ stdv,
                                            const float* obs, int N, int wave, uint64_t seed, uint64_t ctr, const ActOut& o) {
  constexpr int NL = S::NL, NF = S::nfrag(), I = S::I, AO = S::AO;
  // Inside the rollout kernel the pointers come out of an LDS copy of the launch arguments: the compiler no longer knows that they are
  // global memory and would emit flat_load / flat_store - which count on vmcnt AND lgkmcnt, out of order against LDS traffic, so every
  // wait for a weight fragment became vmcnt(0) lgkmcnt(0): the ring was drained at each use. With the address space stated the ring
  // is waited for with counted vmcnt, and a scheduling barrier after every request keeps the requests at their places in the MFMA
  // stream (the scheduler otherwise moves them into clusters - same finding as in k_ppo_fwdbwd_split).
  typedef const __attribute__((address_space(1))) f32x4* gf4p;
  typedef const __attribute__((address_space(1))) float* gfp;
  typedef __attribute__((address_space(1))) float* gwp;
  const gf4p wg = (gf4p)wp;
  const gfp bg = (gfp)bp, sg = (gfp)stdv, og = (gfp)obs;
  const int lane = threadIdx.x & 63, blk = lane >> 2, col = lane & 3, envl = col & 1;
  const int env = min(wave * 2 + envl, N - 1);
  const bool live = col < 2 && wave * 2 + envl < N;
  f32x4 ring[kRing];
  sfor<kRing>([&](auto i) { ring[i] = wg[(size_t)(i < NF ? (int)i : 0) * 64 + lane]; __builtin_amdgcn_sched_barrier(0); });
  // ---- observation rows -> LDS (row 0), zero padded to the next k quad
  {
    const int e0 = min(wave * 2, N - 1), e1 = min(wave * 2 + 1, N - 1);
#pragma unroll
    for (int i = 0; i < (2 * I + 63) / 64; i++) {
      const int idx = lane + 64 * i;
      if (idx < 2 * I) {
        const int e = idx >= I, k = idx - e * I;
        const float v = og[(size_t)(e ? e1 : e0) * I + k];
        xb[e * kXW + k] = v;
        if (o.obs_store && wave * 2 + e < N) ((gwp)o.obs_store)[(size_t)(wave * 2 + e) * I + k] = v;
      }
    }
    if (lane < 2 * (S::up4(I) - I)) xb[(lane & 1) * kXW + I + (lane >> 1)] = 0.0f;
  }
  roll_sync();
  f32x4 out = {0, 0, 0, 0};
  sfor<NL>([&](auto L) {
    constexpr int l = L, ng = S::ng(l), kq = S::kq(l), oa4 = S::oa4(l), nb = S::nblk(l);
    constexpr bool last = l == NL - 1;
    const float* xin = xb + (l & 1) * 2 * kXW + envl * kXW;
    float* xout = xb + ((l + 1) & 1) * 2 * kXW + envl * kXW;
    f32x4 acc[ng];
    const float* xl[ng];
    sfor<ng>([&](auto G) {
      constexpr int g = G;
      const int o0 = 4 * (16 * g + blk);                                     // first neuron of this lane's block
      acc[g] = *reinterpret_cast<gf4p>(bg + S::boff(l) + g * 64 + blk * 4);
      xl[g] = xin + ((l > 0 && o0 >= oa4) ? 64 : 0);                          // critic blocks read the critic's activations
    });
    sfor<kq>([&](auto Q) {
      constexpr int q = Q;
      sfor<ng>([&](auto G) {
        constexpr int g = G, f = S::foff(l) + q * ng + g;
        const f32x4 w = ring[f % kRing];
        if constexpr (f + kRing < NF) { ring[f % kRing] = wg[(size_t)(f + kRing) * 64 + lane]; __builtin_amdgcn_sched_barrier(0); }
        const f32x4 x = *reinterpret_cast<const f32x4*>(xl[g] + 4 * q);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[kk], x[kk], acc[g], 0, 0, 0);
      });
    });
    if constexpr (last) {
      out = acc[0];
    } else {
      sfor<ng>([&](auto G) {
        constexpr int g = G;
        const int b = 16 * g + blk, o0 = 4 * b;
        f32x4 v = acc[g];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = v[i] > 0.0f ? v[i] : __expf(v[i]) - 1.0f;     // ELU; padding neurons: bias 0, weights 0 -> 0
        if (col < 2 && b < nb) *reinterpret_cast<f32x4*>(xout + (o0 < oa4 ? o0 : 64 + o0 - oa4)) = v;
      });
      roll_sync();
    }
  });
  // ---- sampling head on the output tile: lane (block b, column j) holds action means 4 b .. 4 b + 3 of env j (b < oa4 / 4) or the value (first critic block)
  constexpr int nab = S::oa4(NL - 1) / 4;
  float lp = 0.0f;
  if constexpr (VALUE_ONLY) {
    if (blk == nab && live) ((gwp)o.values)[env] = out[0];
    return;
  }
  if (blk < nab) {
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {
      const int f0 = 4 * blk + 2 * pr;                 // even action index: (f0, f0 + 1) share one Box-Muller draw, keyed like nm_ppo_sample
      if (f0 < AO) {
        const float u1 = u24(seed, (uint64_t)env * 64 + f0, ctr), u2 = u24(seed, (uint64_t)env * 64 + f0 + 1, ctr);
        const float rad = sqrtf(-2.0f * __logf(u1));
        float sn, cs;
        __sincosf(6.283185307179586f * u2, &sn, &cs);
        const float z[2] = {rad * cs, rad * sn};
        f32x2u a2, m2, s2;
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
          const float m = out[2 * pr + hh], sd = sg[f0 + hh];
          a2[hh] = m + sd * z[hh]; m2[hh] = m; s2[hh] = sd;
          lp += -0.5f * z[hh] * z[hh] - __logf(sd) - 0.9189385332046727f;
        }
        if (live) {
          typedef __attribute__((address_space(1))) f32x2u* g2p;
          *(g2p)((gwp)o.actions + (size_t)env * AO + f0) = a2;
          *(g2p)((gwp)o.mu + (size_t)env * AO + f0) = m2;
          *(g2p)((gwp)o.sigma + (size_t)env * AO + f0) = s2;
        }
      }
    }
  } else if (blk == nab && live) {
    ((gwp)o.values)[env] = out[0];
  }
  lp += __shfl_xor(lp, 4); lp += __shfl_xor(lp, 8); lp += __shfl_xor(lp, 16);     // over the action blocks (lane bits 2..4; blocks >= nab hold 0)
  if (blk == 0 && live) ((gwp)o.logp)[env] = lp;
}

// ---- the K-step launch (kernels in nm_rollout.hip - a translation unit of its own, so that the code generation of k_env_step in
// nm_hip.hip is not touched by a second kernel around the same physics; host launchers below)
struct RollArgs {
  int K;
  const f32x4* wp; const float* bp; const float* stdv;       // packed policy (k_roll_pack), std[AO]
  uint64_t seed; const int64_t* iter_dev;                      // action-noise key: (seed, *iter_dev, step, env, action pair) = nm_ppo_sample's
  const float* obs0; float* obs_final;                         // [N,66]: what the first act sees / the observation after the last step
  float *s_obs, *s_actions, *s_logp, *s_values, *s_mu, *s_sigma, *s_rewards;   // rollout storage, [K,N,.]
  unsigned char* s_dones;
  float *cur_ret, *cur_len, *fin3;                             // the runner's episode bookkeeping (nm_ppo_record's)
  float* st_sum; int* st_cnt;                                  // [K,kNREW], [K,4]: per-step accumulators (Args::stat_sum / stat_cnt of that step)
  int* to_step;                                                // [N]: the step at which the env timed out in this rollout, or -1
  float* last_values;                                          // [N] or null: the critic's value of the observation after the last step (PPO.compute_returns)
  unsigned long long* wave_clock;                              // measurement (nm_set_debug_buffer on): [waves][2] s_memtime at the wave's start / end, else null
};
struct TailArgs {
  int N, K;
  float* st_sum; int* st_cnt; const int* to_step;
  float *ep_stats, *time_outs;
  float ep_len_s;
  long long* counters;
  float gamma;                 // < 0: no time-out bootstrap (the env does not send time_outs)
  const float* s_values; float* s_rewards;
  const int* ep_idx; int n_ep; float* ep_acc;
  unsigned long long* to_owner;
};
}  // namespace nmr
namespace nm { template <class real> struct Model; template <class real> struct Args; }
namespace nmr {
int launch_pack(const float* flat, float* wp, float* bp, hipStream_t s);
int launch_act(const float* wp, const float* bp, const float* stdv, const float* obs, int N, uint64_t seed, const int64_t* iter_dev, int step, const ActOut& o, hipStream_t s);
int launch_rollout(const nm::Model<float>* M_dev, const nm::Args<float>& a, const RollArgs& R, const TailArgs& t, hipStream_t s);

}  // namespace nmr
