// nm_policy.hip - ActorCritic actor forward (rsl_rl v1.0.2 ActorCritic.act mean path; reference call sites
// play.py:122, train.py:40) as a batched GEMM chain on the matrix cores: exact-f32 MFMA (v_mfma_f32_32x32x2_f32),
// bias + ELU fused into the accumulator epilogue. One wave computes a 32(envs) x 32(outputs) tile.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/nightmare_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// y[N,O] = act(x[N,K] W[O,K]^T + b[O]);  grid = (ceil(N/32), ceil(O/32)), block = 64
__global__ void __launch_bounds__(64) k_linear_mfma(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
                                                    float* __restrict__ y, int N, int K, int O, int elu) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
  const int ar = row0 + r, bc = col0 + r;  // A row (env), B column (output unit) owned by this lane
  const bool aok = ar < N, bok = bc < O;
  const float* xa = x + (size_t)(aok ? ar : 0) * K;
  const float* wb = W + (size_t)(bok ? bc : 0) * K;
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // lane l feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31] of each 32x32x2 step
  for (int k0 = 0; k0 < K; k0 += 2) {
    const int k = k0 + h;
    float a = (aok && k < K) ? xa[k] : 0.0f;
    float w = (bok && k < K) ? wb[k] : 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w, acc, 0, 0, 0);
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
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

static thread_local std::string g_perr;
static float* g_scratch[2] = {nullptr, nullptr};
static size_t g_scratch_n = 0;
static int g_scratch_dev = -1;
extern "C" const char* nm_last_error(void);
extern "C" int nm_policy_set_error(const char* m);

extern "C" int nm_policy_forward(const float* obs, int32_t N, const float* const* weights, const float* const* bias, const int32_t* dims,
                                 int32_t n_layers, float* actions, void* stream) {
  if (!obs || !weights || !bias || !dims || !actions || N <= 0 || n_layers <= 0) return nm_policy_set_error("nm_policy_forward: bad argument");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nm_policy_set_error("nm_policy_forward: no HIP device");
  size_t maxh = 0;
  for (int l = 1; l < n_layers; l++) maxh = (size_t)dims[l] > maxh ? (size_t)dims[l] : maxh;
  size_t need = (size_t)N * (maxh ? maxh : 1);
  if (n_layers > 1 && (need > g_scratch_n || dev != g_scratch_dev)) {
    for (int i = 0; i < 2; i++) {
      if (g_scratch[i]) (void)hipFree(g_scratch[i]);
      if (hipMalloc((void**)&g_scratch[i], need * sizeof(float)) != hipSuccess) return nm_policy_set_error("nm_policy_forward: hipMalloc failed");
    }
    g_scratch_n = need;
    g_scratch_dev = dev;
  }
  const float* in = obs;
  for (int l = 0; l < n_layers; l++) {
    const bool last = l == n_layers - 1;
    float* out = last ? actions : g_scratch[l & 1];
    dim3 grid((N + 31) / 32, (dims[l + 1] + 31) / 32);
    hipLaunchKernelGGL(k_linear_mfma, grid, dim3(64), 0, (hipStream_t)stream, in, weights[l], bias[l], out, N, dims[l], dims[l + 1], last ? 0 : 1);
    if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_policy_forward: launch failed");
    in = out;
  }
  return 0;
}
