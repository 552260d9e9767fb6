// nm_policy.hip - ActorCritic actor forward (rsl_rl v1.0.2 ActorCritic.act mean path; reference call sites
// play.py:122, train.py:40) as a batched GEMM chain on the matrix cores: exact-f32 MFMA (v_mfma_f32_32x32x2_f32),
// bias + ELU fused into the accumulator epilogue.
//
// Fused path (all dims <= 256, <= 4 layers): one workgroup of 8 waves owns 16 envs; activations ping-pong between two
// padded LDS tiles, weights stream from L2 as one 16-byte load per lane per 4 k-steps; a wave computes 32x32 output
// tiles. One launch per policy step. Layers that do not fit use the per-layer kernel.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/nightmare_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileRows = 16, kMaxDim = 256, kLd = kMaxDim + 1, kMaxLayers = 4;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Wt[k][o] = W[o][k] for k < K, 0 for K <= k < Kpad: torch.nn.Linear layout -> what the MFMA B operand wants
// (coalesced over output units), K padded to the 64-deep chunk of the main loop so that loop needs no predicates.
__global__ void k_transpose_pad(const float* __restrict__ W, float* __restrict__ Wt, int O, int K, int Kpad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;   // i = k * O + o over the padded matrix (coalesced writes)
  if (i < O * Kpad) { int k = i / O, o = i - k * O; Wt[i] = k < K ? W[(size_t)o * K + k] : 0.0f; }
}

struct MlpArgs {
  const float* w[kMaxLayers];   // transposed + padded
  const float* b[kMaxLayers];
  int dims[kMaxLayers + 1];
  int n_layers, N;
};

// One workgroup (8 waves, two per SIMD) owns 16 envs; activations ping-pong between two padded LDS tiles. A wave computes 16x16 output
// tiles with v_mfma_f32_16x16x4_f32 (exact f32), TWO tiles at a time so the two accumulator chains hide the 40-cycle
// dependent latency behind the 32-cycle issue interval. Lane (r = l&15, q = l>>4) feeds A[i=r][k=k0+q], B[k=k0+q][j=r].
constexpr int kMlpThreads = 512;   // 8 waves: one pass over the 16 column tiles of a 256-wide layer, two waves per SIMD
__global__ void __launch_bounds__(kMlpThreads) k_mlp_fused(const float* __restrict__ obs, float* __restrict__ out, MlpArgs a) {
  __shared__ float act[2][kTileRows * kLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * kTileRows;
  for (int i = tid; i < 2 * kTileRows * kLd; i += kMlpThreads) (&act[0][0])[i] = 0.0f;   // padded k-columns must be finite
  __syncthreads();
  {  // stage the observation tile (coalesced rows)
    const int K = a.dims[0];
    for (int i = tid; i < kTileRows * K; i += kMlpThreads) {
      int rr = i / K, kk = i - rr * K;
      act[0][rr * kLd + kk] = (row0 + rr < a.N) ? obs[(size_t)(row0 + rr) * K + kk] : 0.0f;
    }
  }
  __syncthreads();
  for (int l = 0; l < a.n_layers; l++) {
    const int K = a.dims[l], O = a.dims[l + 1];
    const bool last = l == a.n_layers - 1;
    const float* __restrict__ Wt = a.w[l];
    const float* __restrict__ B = a.b[l];
    const float* xrow = act[l & 1] + r * kLd;
    float* y = act[(l + 1) & 1];
    const int ntile = (O + 15) / 16, nch = (K + 63) >> 6;
    for (int t = wave * 2; t < ntile; t += 2 * (kMlpThreads / 64)) {   // this wave: tiles t and t+1
      const int c0 = t * 16 + r, c1 = c0 + 16;
      const bool ok0 = c0 < O, ok1 = c1 < O;
      const float* w0 = Wt + (ok0 ? c0 : 0);
      const float* w1 = Wt + (ok1 ? c1 : 0);
      f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      float cur0[16], cur1[16], nx0[16], nx1[16];
#pragma unroll
      for (int j = 0; j < 16; j++) { cur0[j] = w0[(size_t)(4 * j + q) * O]; cur1[j] = w1[(size_t)(4 * j + q) * O]; }
      for (int c = 0; c < nch; c++) {
        const int kb = (c << 6) + q;
        if (c + 1 < nch) {
#pragma unroll
          for (int j = 0; j < 16; j++) { nx0[j] = w0[(size_t)(kb + 64 + 4 * j) * O]; nx1[j] = w1[(size_t)(kb + 64 + 4 * j) * O]; }
        }
        float av[16];
#pragma unroll
        for (int j = 0; j < 16; j++) av[j] = xrow[kb + 4 * j];
#pragma unroll
        for (int j = 0; j < 16; j++) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], cur0[j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], cur1[j], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) { cur0[j] = nx0[j]; cur1[j] = nx1[j]; }
      }
      // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int col = half ? c1 : c0;
        if (half ? ok1 : ok0) {
          const float bias = B[col];
#pragma unroll
          for (int reg = 0; reg < 4; reg++) {
            const int rr = q * 4 + reg;
            float v = (half ? acc1[reg] : acc0[reg]) + bias;
            if (!last) {
              v = v > 0.0f ? v : expm1f(v);
              y[rr * kLd + col] = v;
            } else if (row0 + rr < a.N) {
              out[(size_t)(row0 + rr) * O + col] = v;
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

// y[N,O] = act(x[N,K] W[O,K]^T + b[O]);  grid = (ceil(N/32), ceil(O/32)), block = 64   (fallback for wide layers)
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

static float* g_wt = nullptr;   // transposed weights of the fused path (repacked every call: weights change between PPO updates)
static size_t g_wt_n = 0;
static int g_wt_dev = -1;
static float* g_scratch[2] = {nullptr, nullptr};
static size_t g_scratch_n = 0;
static int g_scratch_dev = -1;
extern "C" int nm_policy_set_error(const char* m);

static MlpArgs g_packed;          // layer table of the last nm_policy_pack (device pointers into g_wt / the caller's biases)
static bool g_packed_ok = false;
static int g_packed_dev = -1;

static bool mlp_fits(const int32_t* dims, int32_t n_layers) {
  bool fits = n_layers <= kMaxLayers;
  for (int l = 0; l <= n_layers && fits; l++) fits = dims[l] > 0 && dims[l] <= kMaxDim;
  return fits;
}

extern "C" int nm_policy_pack(const float* const* weights, const float* const* bias, const int32_t* dims, int32_t n_layers, void* stream) {
  if (!weights || !bias || !dims || n_layers <= 0) return nm_policy_set_error("nm_policy_pack: bad argument");
  if (!mlp_fits(dims, n_layers)) return nm_policy_set_error("nm_policy_pack: network too large for the fused kernel (<= 4 layers of <= 256 units)");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nm_policy_set_error("nm_policy_pack: no HIP device");
  g_packed_ok = false;
  MlpArgs a;
  a.n_layers = n_layers; a.N = 0;
  size_t tot = 0;
  for (int l = 0; l < n_layers; l++) tot += (size_t)((dims[l] + 63) & ~63) * dims[l + 1];
  if (tot > g_wt_n || dev != g_wt_dev) {
    if (g_wt) (void)hipFree(g_wt);
    g_wt = nullptr; g_wt_n = 0;
    if (hipMalloc((void**)&g_wt, tot * sizeof(float)) != hipSuccess) return nm_policy_set_error("nm_policy_pack: hipMalloc failed");
    g_wt_n = tot; g_wt_dev = dev;
  }
  size_t off = 0;
  for (int l = 0; l < n_layers; l++) {
    int kpad = (dims[l] + 63) & ~63, n = kpad * dims[l + 1];
    hipLaunchKernelGGL(k_transpose_pad, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, weights[l], g_wt + off, dims[l + 1], dims[l], kpad);
    a.w[l] = g_wt + off; a.b[l] = bias[l];
    off += n;
  }
  for (int l = 0; l <= n_layers; l++) a.dims[l] = dims[l];
  if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_policy_pack: launch failed");
  g_packed = a; g_packed_ok = true; g_packed_dev = dev;
  return 0;
}

extern "C" int nm_policy_forward_packed(const float* obs, int32_t N, float* actions, void* stream) {
  if (!obs || !actions || N <= 0) return nm_policy_set_error("nm_policy_forward_packed: bad argument");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nm_policy_set_error("nm_policy_forward_packed: no HIP device");
  if (!g_packed_ok || dev != g_packed_dev) return nm_policy_set_error("nm_policy_forward_packed: no packed network on this device (call nm_policy_pack)");
  MlpArgs a = g_packed;
  a.N = N;
  hipLaunchKernelGGL(k_mlp_fused, dim3((N + kTileRows - 1) / kTileRows), dim3(kMlpThreads), 0, (hipStream_t)stream, obs, actions, a);
  if (hipGetLastError() != hipSuccess) return nm_policy_set_error("nm_policy_forward_packed: launch failed");
  return 0;
}

extern "C" int nm_policy_forward(const float* obs, int32_t N, const float* const* weights, const float* const* bias, const int32_t* dims,
                                 int32_t n_layers, float* actions, void* stream) {
  if (!obs || !weights || !bias || !dims || !actions || N <= 0 || n_layers <= 0) return nm_policy_set_error("nm_policy_forward: bad argument");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nm_policy_set_error("nm_policy_forward: no HIP device");
  if (mlp_fits(dims, n_layers)) {
    if (nm_policy_pack(weights, bias, dims, n_layers, stream)) return 1;
    return nm_policy_forward_packed(obs, N, actions, stream);
  }
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
