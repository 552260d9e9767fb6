// nm_rl.hip - small native pieces of the on-policy loop that sit next to the env on the device.
// nm_gae: backward GAE(lambda) scan of rsl_rl v1.0.2 RolloutStorage.compute_returns (caller: reference train.py:54
// -> OnPolicyRunner.learn -> PPO.compute_returns), one thread per env, coalesced across envs.
#include <hip/hip_runtime.h>

#include "../../include/nightmare_hip.h"

extern "C" int nm_policy_set_error(const char* m);

__global__ void k_gae(const float* __restrict__ rewards, const float* __restrict__ values, const unsigned char* __restrict__ dones,
                      const float* __restrict__ last_values, int T, int N, float gamma, float lam, float* __restrict__ returns) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  float adv = 0.0f, next = last_values[e];
  for (int s = T - 1; s >= 0; s--) {
    size_t i = (size_t)s * N + e;
    float live = 1.0f - (float)dones[i];
    float v = values[i];
    float delta = rewards[i] + live * gamma * next - v;
    adv = delta + live * gamma * lam * adv;
    returns[i] = adv + v;
    next = v;
  }
}

extern "C" int nm_gae(const float* rewards, const float* values, const unsigned char* dones, const float* last_values, int32_t T, int32_t N,
                      float gamma, float lam, float* returns, void* stream) {
  if (!rewards || !values || !dones || !last_values || !returns || T <= 0 || N <= 0) return nm_policy_set_error("nm_gae: bad argument");
  hipLaunchKernelGGL(k_gae, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, rewards, values, dones, last_values, T, N, gamma, lam, returns);
  if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_gae: launch failed");
  return 0;
}

// GAE + what RolloutStorage.compute_returns does next (rsl_rl v1.0.2: advantages = returns - values, then (adv - mean) / (std + 1e-8) over all
// envs x steps, unbiased std) in two launches instead of ~18 framework launches whose host-side issue left the GPU idle for ~0.3 ms per
// iteration: k_gae_adv walks its env like k_gae, writes returns and raw advantages and leaves per-block partial sums (sum, sum of
// squares; fixed summation order: a thread adds its env's steps last-to-first, the block adds its threads by a fixed tree); k_adv_norm
// adds the partials in block order in every block (same mean / std everywhere, reproducible) and normalises in place.
constexpr int kGaeThreads = 256;
__global__ void __launch_bounds__(kGaeThreads) k_gae_adv(const float* __restrict__ rewards, const float* __restrict__ values, const unsigned char* __restrict__ dones,
                                                         const float* __restrict__ last_values, int T, int N, float gamma, float lam, float* __restrict__ returns,
                                                         float* __restrict__ adv_out, float* __restrict__ partial) {
  __shared__ float red[2][kGaeThreads];
  const int e = blockIdx.x * kGaeThreads + threadIdx.x;
  float s1 = 0.0f, s2 = 0.0f;
  if (e < N) {
    float adv = 0.0f, next = last_values[e];
    // (the loads of a step do not depend on the recurrence: unrolled, eight steps' loads are in flight at once instead of one HBM round
    // trip per step - 80 of them in a row were the kernel's 29 us)
#pragma unroll 8
    for (int s = T - 1; s >= 0; s--) {
      const size_t i = (size_t)s * N + e;
      const float live = 1.0f - (float)dones[i];
      const float v = values[i];
      const float delta = rewards[i] + live * gamma * next - v;
      adv = delta + live * gamma * lam * adv;
      const float ret = adv + v;
      returns[i] = ret;
      const float a = ret - v;                  // torch.sub(returns, values): what the framework path stores (not `adv` itself)
      adv_out[i] = a;
      s1 += a; s2 += a * a;
      next = v;
    }
  }
  red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = kGaeThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = red[0][0]; partial[2 * blockIdx.x + 1] = red[1][0]; }
}
__global__ void __launch_bounds__(256) k_adv_norm(float* __restrict__ adv, size_t n, const float* __restrict__ partial, int nblk) {
  float s1 = 0.0f, s2 = 0.0f;
  for (int b = 0; b < nblk; b++) { s1 += partial[2 * b]; s2 += partial[2 * b + 1]; }      // the same order in every thread
  const float cnt = (float)n, mean = s1 / cnt;
  const float var = fmaxf(s2 / cnt - mean * mean, 0.0f) * (cnt / fmaxf(cnt - 1.0f, 1.0f));   // unbiased, like torch.std (distributed.global_advantage_stats)
  const float inv = 1.0f / (sqrtf(var) + 1e-8f);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) adv[i] = (adv[i] - mean) * inv;
}
extern "C" int nm_gae_advantages(const float* rewards, const float* values, const unsigned char* dones, const float* last_values, int32_t T, int32_t N,
                                 float gamma, float lam, float* returns, float* advantages, float* scratch, int32_t normalize, void* stream) {
  if (!rewards || !values || !dones || !last_values || !returns || !advantages || !scratch || T <= 0 || N <= 0) return nm_policy_set_error("nm_gae_advantages: bad argument");
  const int nblk = (N + kGaeThreads - 1) / kGaeThreads;
  hipLaunchKernelGGL(k_gae_adv, dim3(nblk), dim3(kGaeThreads), 0, (hipStream_t)stream, rewards, values, dones, last_values, T, N, gamma, lam, returns, advantages, scratch);
  if (normalize) {
    const size_t n = (size_t)T * N;
    const int grid = (int)((n + 1023) / 1024 < 1024 ? (n + 1023) / 1024 : 1024);
    hipLaunchKernelGGL(k_adv_norm, dim3(grid), dim3(256), 0, (hipStream_t)stream, advantages, n, scratch, nblk);
  }
  if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_gae_advantages: launch failed");
  return 0;
}

// ------------------------------------------------------------------------------------------------ rollout collection
// PPO.act of rsl_rl v1.0.2 (`algorithms/ppo.py`: actor mean, Normal(mean, std).sample(), log_prob, critic value, transition record)
// after the fused actor+critic forward (nm_policy_forward on the merged network: out[N, A+1] = action means | value):
// one pass per env row that draws the action, evaluates its log-probability and files everything the update needs directly into
// step s of the rollout storage - instead of ~25 elementwise launches per step.
__device__ __forceinline__ float u24(uint64_t seed, uint64_t a, uint64_t b) {   // counter-based uniform in (0,1], 24 bits
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * (a + 1) + 0xD1B54A32D192ED03ull * b;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return ((float)(uint32_t)(x >> 40) + 1.0f) * (1.0f / 16777216.0f);
}
// 16 lanes per env: lane j < ceil(A/2) draws the action pair (2j, 2j+1) (Box-Muller: two normals per pair of uniforms), the
// log-probability is summed over the 16 lanes, and the block copies its 16 envs' observations as one contiguous run.
constexpr int kSampleEnvs = 16;    // envs per 256-thread block
__global__ void __launch_bounds__(256) k_ppo_sample(const float* __restrict__ net_out, const float* __restrict__ std, const float* __restrict__ obs, int N, int A, int n_obs,
                             uint64_t seed, const int64_t* __restrict__ iter_dev, int step, float* __restrict__ actions, float* __restrict__ logp,
                             float* __restrict__ values, float* __restrict__ mu, float* __restrict__ sigma, float* __restrict__ obs_store) {
  const int e0 = blockIdx.x * kSampleEnvs;
  if (obs_store) {
    const size_t base = (size_t)e0 * n_obs;
    const int cnt = min(kSampleEnvs, N - e0) * n_obs;
    for (int i = threadIdx.x; i < cnt; i += 256) obs_store[base + i] = obs[base + i];
  }
  const int e = e0 + (threadIdx.x >> 4), j = 2 * (threadIdx.x & 15);
  float lp = 0.0f;
  if (e < N && j < A) {
    const uint64_t ctr = (uint64_t)iter_dev[0] * 4096ull + (uint64_t)step;
    const float* o = net_out + (size_t)e * (A + 1);
    const float u1 = u24(seed, (uint64_t)e * 64 + j, ctr), u2 = u24(seed, (uint64_t)e * 64 + j + 1, ctr);
    const float rad = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.283185307179586f * u2, &sn, &cs);
    const float z[2] = {rad * cs, rad * sn};
    for (int h = 0; h < 2 && j + h < A; h++) {
      const float m = o[j + h], sd = std[j + h];
      actions[(size_t)e * A + j + h] = m + sd * z[h];
      mu[(size_t)e * A + j + h] = m;
      sigma[(size_t)e * A + j + h] = sd;
      lp += -0.5f * z[h] * z[h] - __logf(sd) - 0.9189385332046727f;   // Normal.log_prob: -(a-m)^2/(2 sd^2) - log sd - log sqrt(2 pi)
    }
    if (j == 0) values[e] = o[A];
  }
  lp += __shfl_xor(lp, 1); lp += __shfl_xor(lp, 2); lp += __shfl_xor(lp, 4); lp += __shfl_xor(lp, 8);
  if (e < N && j == 0) logp[e] = lp;
}
extern "C" int nm_ppo_sample(const float* net_out, const float* std, const float* obs, int32_t N, int32_t A, int32_t n_obs, uint64_t seed,
                             const int64_t* iter_dev, int32_t step, float* actions, float* logp, float* values, float* mu, float* sigma, float* obs_store,
                             void* stream) {
  if (!net_out || !std || !iter_dev || !actions || !logp || !values || !mu || !sigma || N <= 0 || A <= 0 || A > 32) return nm_policy_set_error("nm_ppo_sample: bad argument (1..32 actions)");
  hipLaunchKernelGGL(k_ppo_sample, dim3((N + kSampleEnvs - 1) / kSampleEnvs), dim3(256), 0, (hipStream_t)stream, net_out, std, obs, N, A, n_obs, seed, iter_dev, step, actions,
                     logp, values, mu, sigma, obs_store);
  if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_ppo_sample: launch failed");
  return 0;
}

// PPO.process_env_step + the runner's episode bookkeeping for one step: reward with the time-out bootstrap
// (rewards += gamma * value * time_out), done flag, running episode return / length, and the sums over episodes that ended.
__global__ void k_ppo_record(const float* __restrict__ rew, const int64_t* __restrict__ done, const float* __restrict__ time_outs,
                             const float* __restrict__ values, float gamma, int N, float* __restrict__ rewards_store, unsigned char* __restrict__ dones_store,
                             float* __restrict__ cur_ret, float* __restrict__ cur_len, float* __restrict__ fin3,
                             const float* __restrict__ ep_stats, const int* __restrict__ ep_idx, int n_ep, float* __restrict__ ep_acc) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_ep) ep_acc[e] += ep_stats[ep_idx[e]];      // the runner's running sum of extras['episode'] over the rollout (n_ep <= 256: block 0)
  if (e >= N) return;
  const float r = rew[e];
  const bool d = done[e] > 0;
  rewards_store[e] = r + (time_outs ? gamma * values[e] * time_outs[e] : 0.0f);
  dones_store[e] = d ? 1 : 0;
  const float cr = cur_ret[e] + r, cl = cur_len[e] + 1.0f;
  if (d) { atomicAdd(fin3, cr); atomicAdd(fin3 + 1, cl); atomicAdd(fin3 + 2, 1.0f); }
  cur_ret[e] = d ? 0.0f : cr;
  cur_len[e] = d ? 0.0f : cl;
}
extern "C" int nm_ppo_record(const float* rew, const int64_t* done, const float* time_outs, const float* values, float gamma, int32_t N,
                             float* rewards_store, unsigned char* dones_store, float* cur_ret, float* cur_len, float* fin3,
                             const float* ep_stats, const int32_t* ep_idx, int32_t n_ep, float* ep_acc, void* stream) {
  if (!rew || !done || !values || !rewards_store || !dones_store || !cur_ret || !cur_len || !fin3 || N <= 0) return nm_policy_set_error("nm_ppo_record: bad argument");
  if (n_ep < 0 || n_ep > 256 || (n_ep > 0 && (!ep_stats || !ep_idx || !ep_acc))) return nm_policy_set_error("nm_ppo_record: bad episode-statistics arguments");
  hipLaunchKernelGGL(k_ppo_record, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, rew, done, time_outs, values, gamma, N, rewards_store,
                     dones_store, cur_ret, cur_len, fin3, ep_stats, ep_idx, n_ep, ep_acc);
  if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_ppo_record: launch failed");
  return 0;
}
