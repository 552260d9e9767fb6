// nm_rollout.hip - K env steps per launch with the policy inside the env's wavefront (C ABI: nm_rollout / nm_rollout_act in nm_hip.hip).
//
// The on-policy rollout of rsl_rl v1.0.2 (OnPolicyRunner.learn: `for i in range(num_steps_per_env): actions = alg.act(obs, critic_obs);
// obs, _, rewards, dones, infos = env.step(actions); alg.process_env_step(rewards, dones, infos)`, the loop reference train.py:54 drives,
// horizon envs/nightmare_v3_config.py:135) as ONE launch: every wave keeps its two envs for all K steps - policy forward + sampling
// (nm_rollout.h), the physics step and epilogue of k_env_step (nm::wave_step, the same code), the transition record - and goes straight
// from step t into step t + 1: no launch boundary, no wait for the slowest wave of the step. What crosses waves is collected per step
// (episode-sum atomics, reset counts) and closed by k_rollout_tail afterwards.
//
// A translation unit of its own: a second kernel around nm::wave_step in nm_hip.hip changed the register allocation of k_env_step
// (250 VGPRs / no spill -> 256 / 1 spill); here the two kernels cannot see each other.
#include <hip/hip_runtime.h>

#include "nm_core.h"
#include "nm_rollout.h"

#ifndef NM_WAVES_PER_SIMD
#define NM_WAVES_PER_SIMD 2
#endif

namespace nmr {

// PPO.process_env_step + the runner's bookkeeping for this wave's envs (k_ppo_record's arithmetic), in two halves so that its loads travel
// with the next policy step's observation loads (one L2 round trip instead of two): `load` right after the step's stores have landed,
// `file` whenever the values are needed. The time-out bootstrap needs the step's extras['time_outs'], a cross-wave quantity: k_rollout_tail adds it.
struct RecordRegs { float rw, to, cr, cl; long long d; };
__device__ __forceinline__ void record_load(RecordRegs& r, const RollArgs* Rs, const nm::Args<float>* As, int wave) {
  const int lane = threadIdx.x, e = min(wave * 2 + (lane & 1), As->N - 1);
  // (global-memory accessors of simt.h: the pointers come out of LDS copies of the arguments - plain dereferences would be flat_load)
  r.rw = simt::gld1(As->rew, e); r.d = simt::gld1(As->done, e); r.to = simt::gld1(As->timeout_now, e);
  r.cr = simt::gld1((const float*)Rs->cur_ret, e); r.cl = simt::gld1((const float*)Rs->cur_len, e);
}
__device__ __forceinline__ void record_file(const RecordRegs& r, const RollArgs* Rs, const nm::Args<float>* As, int t, int wave) {
  const int lane = threadIdx.x, N = As->N, e = wave * 2 + lane;
  if (lane < 2 && e < N) {
    const size_t so = (size_t)t * N;
    const bool d = r.d > 0;
    float cr = r.cr + r.rw, cl = r.cl + 1.0f;
    simt::gst1(Rs->s_rewards, so + e, r.rw);
    simt::gst1(Rs->s_dones, so + e, (unsigned char)(d ? 1 : 0));
    if (d) { atomicAdd(Rs->fin3, cr); atomicAdd(Rs->fin3 + 1, cl); atomicAdd(Rs->fin3 + 2, 1.0f); cr = 0.f; cl = 0.f; }
    simt::gst1(Rs->cur_ret, (size_t)e, cr); simt::gst1(Rs->cur_len, (size_t)e, cl);
    if (r.to != 0.f) simt::gst1(Rs->to_step, (size_t)e, t);
  }
}
// The record of step t - 1 (t > 0) and PPO.act of step t for the wave's envs + the launch arguments of the env step that follows.
// Out of line: its registers (weight ring, accumulators) are not live across the physics, and the physics' are not live here.
template <class S>
__device__ __noinline__ void policy_step(float* xb, const RollArgs* Rs, nm::Args<float>* As, int t, int wave, uint64_t noise0) {
  const int N = As->N;
  const size_t so = (size_t)t * N;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // state rows, observation, reward / done / time-out of the previous step: stored
  RecordRegs rec;
  if (t > 0) record_load(rec, Rs, As, wave);         // issued before the observation loads below: they return together
  ActOut o{Rs->s_actions + so * nm::kNU, Rs->s_logp + so, Rs->s_values + so, Rs->s_mu + so * nm::kNU, Rs->s_sigma + so * nm::kNU, t == 0 ? Rs->s_obs : nullptr};
  // the observation is the one this wave's previous step wrote into the storage row of step t
  policy_wave<S>(xb, Rs->wp, Rs->bp, Rs->stdv, t == 0 ? Rs->obs0 : Rs->s_obs + so * nm::kNOBS, N, wave, Rs->seed, (uint64_t)simt::gld1(Rs->iter_dev, 0) * 4096ull + (uint64_t)t, o);
  if (t > 0) record_file(rec, Rs, As, t - 1, wave);
  if (threadIdx.x == 0) {
    As->actions = o.actions;
    As->obs = t + 1 < Rs->K ? Rs->s_obs + (so + N) * nm::kNOBS : Rs->obs_final;     // the step files its observation where the next act reads it
    As->stat_sum = Rs->st_sum + (size_t)t * nm::kNREW;
    As->stat_cnt = Rs->st_cnt + (size_t)t * 4;
    As->noise_step = noise0 + (uint64_t)t;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the actions are in L2 before the load stage asks for them (other lanes of this wave)
  nm::wave_sync();
}
// the record of the rollout's last step (no policy step follows it)
__device__ __noinline__ void record_last(const RollArgs* Rs, const nm::Args<float>* As, int t, int wave) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RecordRegs rec;
  record_load(rec, Rs, As, wave);
  record_file(rec, Rs, As, t, wave);
}
// PPO.compute_returns' `last_values = actor_critic.evaluate(last_critic_obs)` for the wave's envs: the forward once more, on the observation the
// last step left in obs_final (this wave's own stores: waited for by record_last), value head only - instead of nine framework launches
template <class S>
__device__ __noinline__ void value_last(float* xb, const RollArgs* Rs, const nm::Args<float>* As, int wave) {
  ActOut o{nullptr, nullptr, Rs->last_values, nullptr, nullptr, nullptr};
  nm::wave_sync();
  policy_wave<S, true>(xb, Rs->wp, Rs->bp, Rs->stdv, Rs->obs_final, As->N, wave, 0, 0, o);
}

template <class S>
__global__ void __launch_bounds__(64, NM_WAVES_PER_SIMD) k_env_rollout(const nm::Model<float>* __restrict__ Mp, nm::Args<float> A, RollArgs R) {
  __shared__ nm::ShW<float, 2> sh;
  __shared__ nm::Model<float> Ms;
  __shared__ nm::Args<float> As;
  __shared__ RollArgs Rs;
  static_assert(sizeof(sh) >= kXFloats * sizeof(float), "the policy's activation rows alias the env images");
  int wave = blockIdx.x;
#ifndef NM_NO_XCD_MAP
  {
    const int nwx = (int)gridDim.x >> 3;
    if (A.nxcd == 8 && wave < (nwx << 3)) wave = (wave & 7) * nwx + (wave >> 3);
  }
#endif
  if (wave * 2 >= A.N) return;
  As = A;
  Rs = R;
  __syncthreads();
  {  // the model constants: L2 -> LDS, once for the whole rollout
    const uint32_t* src = reinterpret_cast<const uint32_t*>(Mp);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&Ms);
    constexpr int kWords = (int)(sizeof(nm::Model<float>) / 4);
    for (int i = threadIdx.x; i < kWords; i += 64) dst[i] = src[i];
    __syncthreads();
  }
  float* xb = reinterpret_cast<float*>(&sh);          // between two steps the env images hold nothing that is needed (env_load2 rewrites them)
  if ((int)threadIdx.x < 2 && wave * 2 + (int)threadIdx.x < A.N) R.to_step[wave * 2 + threadIdx.x] = -1;
  const uint64_t noise0 = A.noise_step;
  const int K = R.K;
  if (R.wave_clock && threadIdx.x == 0) R.wave_clock[2 * wave] = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < K; t++) {
    policy_step<S>(xb, &Rs, &As, t, wave, noise0);    // (+ the record of step t - 1)
    nm::wave_step<float, 2>(sh, Ms, As, wave);        // env.step: load, decimation x mj_step, epilogue - the code of k_env_step
  }
  record_last(&Rs, &As, K - 1, wave);
  if (Rs.last_values) value_last<S>(xb, &Rs, &As, wave);
  if (Rs.wave_clock && threadIdx.x == 0) Rs.wave_clock[2 * wave + 1] = __builtin_amdgcn_s_memtime();
}
// One env's policy step as a launch of its own: PPO.act on the same wave code (the step-by-step counterpart of k_env_rollout and its
// bit-exact reference in tests/test_gpu_rollout.py). One wave = two envs.
template <class S>
__global__ void __launch_bounds__(64) k_roll_act(const f32x4* __restrict__ wp, const float* __restrict__ bp, const float* __restrict__ stdv, const float* __restrict__ obs,
                                                 int N, uint64_t seed, const int64_t* __restrict__ iter_dev, int step, ActOut o) {
  __shared__ float xb[kXFloats];
  policy_wave<S>(xb, wp, bp, stdv, obs, N, (int)blockIdx.x, seed, (uint64_t)iter_dev[0] * 4096ull + (uint64_t)step, o);
}
// Closes a rollout of K steps: what the closing wave of every k_env_step launch does per step (step_tail in nm_hip.hip), for all K steps in
// order, plus the two things of PPO.process_env_step / the runner that need them.
//   * extras['episode'] (ep_stats) is refreshed by every step in which >= 1 env reset (env.py:344-371), and the runner adds the CURRENT
//     extras to its running sum after every step (ep_acc[i] += ep_stats[ep_idx[i]]) - also the stale one of a step without resets;
//   * extras['time_outs'] likewise keeps the flags of the last step that had a reset, and PPO.process_env_step adds gamma * value *
//     time_outs at EVERY step - so an env's bootstrap term is applied from its time-out step until the next step with a reset.
//     `time_outs` on entry = the flags left by the steps before this rollout; on exit = those of the last refreshing step.
constexpr int kTailSteps = 4096;     // nm_rollout's limit on K
__global__ void __launch_bounds__(256) k_rollout_tail(TailArgs a) {
  __shared__ int cnts[kTailSteps];      // resets of step t (> 0: the step refreshed the extras) - read K times by every thread below
  const int tid = threadIdx.x, N = a.N, K = a.K;
  for (int t = tid; t < K; t += 256) cnts[t] = a.st_cnt[t * 4];
  __syncthreads();
  if (blockIdx.x == 0) {
    // extras['episode'][k] after every step, and the runner's running sum of it: thread k < 16 follows reward k, thread 16 + i the i-th
    // summed key - each walks the K steps on its own (no exchange: a summed key recomputes its reward's value), additions in step order
    if (tid < nm::kNREW + a.n_ep) {
      const bool sums = tid >= nm::kNREW;
      const int k = sums ? a.ep_idx[tid - nm::kNREW] : tid;
      float e = a.ep_stats ? a.ep_stats[k] : 0.f, acc = sums ? a.ep_acc[tid - nm::kNREW] : 0.f;
      for (int t = 0; t < K; t++) {
        const int cnt = cnts[t];
        if (cnt > 0) e = (float)(a.st_sum[(size_t)t * nm::kNREW + k] / (float)cnt / a.ep_len_s);
        acc += e;
      }
      if (sums) a.ep_acc[tid - nm::kNREW] = acc;
      else if (a.ep_stats) a.ep_stats[k] = e;
    }
    if (tid == 255) {
      long long c1 = 0, c2 = 0;
      for (int t = 0; t < K; t++) { c1 += a.st_cnt[t * 4 + 1]; c2 += a.st_cnt[t * 4 + 2]; }
      a.counters[0] += c1; a.counters[1] += c2;
      *a.to_owner = 0ull;          // the next nm_step rewrites extras['time_outs'] in full
    }
  }
  // time-out bootstrap and the final extras['time_outs'], one thread per env
  int tl = -1;
  for (int t = 0; t < K; t++) if (cnts[t] > 0) tl = t;
  for (int e = blockIdx.x * blockDim.x + tid; e < N; e += gridDim.x * blockDim.x) {
    const int ts = a.to_step[e];
    // k_ppo_record's `rew + gamma * value * time_out` with time_out = 1: the product rounded, then the sum rounded (bit for bit)
    auto boot = [&](int t) {
#pragma clang fp contract(off)      // HIP's __fmul_rn / __fadd_rn are plain operators: without this the pair becomes one fma (one rounding)
      const size_t i = (size_t)t * N + e;
      const float gv = a.gamma * a.s_values[i];
      a.s_rewards[i] = a.s_rewards[i] + gv;
    };
    const bool boots = a.gamma >= 0.f;
    if (boots && a.time_outs && a.time_outs[e] != 0.f)         // flags from before the rollout hold until the first refresh
      for (int t = 0; t < K && cnts[t] == 0; t++) boot(t);
    if (boots && ts >= 0 && a.time_outs)
      for (int t = ts; t < K && (t == ts || cnts[t] == 0); t++) boot(t);
    if (a.time_outs && tl >= 0) a.time_outs[e] = ts == tl ? 1.f : 0.f;
  }
}
// the per-step accumulators of a rollout back to zero (a launch of its own: after EVERY block of k_rollout_tail has read them)
__global__ void k_rollout_clear(int K, float* st_sum, int* st_cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K * nm::kNREW) st_sum[i] = 0.f;
  if (i < K * 4) st_cnt[i] = 0;
}

int launch_pack(const float* flat, float* wp, float* bp, hipStream_t s) {
  const int n = RefShape::nfrag() * 256;
  hipLaunchKernelGGL(k_roll_pack<RefShape>, dim3((n + 255) / 256), dim3(256), 0, s, flat, wp, bp);
  return hipGetLastError() != hipSuccess;
}
int launch_act(const float* wp, const float* bp, const float* stdv, const float* obs, int N, uint64_t seed, const int64_t* iter_dev, int step, const ActOut& o,
               hipStream_t s) {
  hipLaunchKernelGGL(k_roll_act<RefShape>, dim3((N + 1) / 2), dim3(64), 0, s, (const f32x4*)wp, bp, stdv, obs, N, seed, iter_dev, step, o);
  return hipGetLastError() != hipSuccess;
}
int launch_rollout(const nm::Model<float>* M_dev, const nm::Args<float>& a, const RollArgs& R, const TailArgs& t, hipStream_t s) {
  const int N = a.N;
  hipLaunchKernelGGL(k_env_rollout<RefShape>, dim3((N + 1) / 2), dim3(64), 0, s, M_dev, a, R);
  if (hipGetLastError() != hipSuccess) return 1;
  hipLaunchKernelGGL(k_rollout_tail, dim3(min((N + 255) / 256, 64)), dim3(256), 0, s, t);
  if (hipGetLastError() != hipSuccess) return 1;
  hipLaunchKernelGGL(k_rollout_clear, dim3((t.K * nm::kNREW + 255) / 256), dim3(256), 0, s, t.K, t.st_sum, t.st_cnt);
  return hipGetLastError() != hipSuccess;
}

}  // namespace nmr
