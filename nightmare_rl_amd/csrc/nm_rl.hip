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
