// nm_hip.hip - gfx950 kernels + the C ABI of include/nightmare_hip.h. Device-only: there is no CPU path.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nightmare_hip.h"
#include "nm_host_model.h"
#include "nm_rollout.h"

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }
#define HIPCHK(x)                                                                                        \
  do {                                                                                                   \
    hipError_t e_ = (x);                                                                                 \
    if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_));                  \
  } while (0)

// ------------------------------------------------------------------------------------------------ kernels
// One 64-lane wavefront per NM_ENVS_PER_WAVE (2) envs, one wave per workgroup: LDS image private to the wave, no inter-wave sync.
#ifndef NM_WAVES_PER_SIMD
#define NM_WAVES_PER_SIMD 2  /* the step kernel needs ~250 VGPRs (row-per-lane A matrix): 2 waves per SIMD; a 4-wave budget spills */
#endif
#ifndef NM_ENVS_PER_WAVE
#define NM_ENVS_PER_WAVE 2
#endif
// extras: like the reference, 'episode' and 'time_outs' are refreshed only by a step in which >= 1 env reset (env.py:344-371);
// then the per-step accumulators are cleared for the next launch. One wave, after all others have published.
template <class real> __device__ __noinline__ void step_tail(const nm::Args<real>& A, real ep_len_s) {
  const int lane = NM_TID;
  // This runs behind the launch's last wave, alone: every global access is a full round trip that nothing hides. So the reads are
  // issued in two batches - everything that is addressed by the lane alone, then the two lists those counts index - not one by one.
#define TAIL_LD(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
  const int cnt = TAIL_LD(A.stat_cnt);
  const int c1 = TAIL_LD(A.stat_cnt + 1), c2 = TAIL_LD(A.stat_cnt + 2);
  const real ssum = TAIL_LD(A.stat_sum + (lane < nm::kNREW ? lane : 0));
  const int n = TAIL_LD(A.nto);
  const int np = TAIL_LD(A.nprev);
  // decided on the device, so that it also holds for a launch replayed from a graph: is this the buffer the last refresh wrote?
  const bool to_full = A.time_outs && TAIL_LD(A.to_owner) != (unsigned long long)(uintptr_t)A.time_outs;
  const long long k0 = lane == 0 ? A.counters[0] : 0, k1 = lane == 0 ? A.counters[1] : 0;
  if (cnt > 0) {
    if (lane < nm::kNREW && A.ep_stats) A.ep_stats[lane] = (float)(ssum / (real)cnt / ep_len_s);
    if (A.time_outs) {
      // extras['time_outs'] = this step's time-out flags (a subset of the resets). The buffer still holds what the last refresh
      // wrote, so only the entries that were 1 are cleared and the new ones set: O(time-outs) instead of N stores.
      // A buffer other than the one the last refresh wrote (to_full) is rewritten completely.
      const int e_new = lane < n ? TAIL_LD(A.to_list + lane) : -1;
      const int e_old = (!to_full && lane < np) ? A.to_prev[lane] : -1;
      if (to_full) {
        for (int i = lane; i < A.N; i += 64) A.time_outs[i] = 0.f;
      } else {
        if (e_old >= 0) A.time_outs[e_old] = 0.f;
        for (int j = lane + 64; j < np; j += 64) A.time_outs[A.to_prev[j]] = 0.f;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zeros have reached L2 before a one goes to the same address
      if (e_new >= 0) { A.time_outs[e_new] = 1.f; A.to_prev[lane] = e_new; }
      for (int j = lane + 64; j < n; j += 64) {
        const int e = TAIL_LD(A.to_list + j);
        A.time_outs[e] = 1.f;
        A.to_prev[j] = e;
      }
      if (lane == 0) { *A.nprev = n; *A.to_owner = (unsigned long long)(uintptr_t)A.time_outs; }
    }
  }
#undef TAIL_LD
  if (lane < nm::kNREW) A.stat_sum[lane] = real(0);
  if (lane == 0) {
    A.counters[0] = k0 + c1;
    A.counters[1] = k1 + c2;
    A.stat_cnt[0] = 0; A.stat_cnt[1] = 0; A.stat_cnt[2] = 0; A.stat_cnt[3] = 0;
    *A.nto = 0;
    A.wave_done[nm::kTicketTop] = 0;
  }
}

// Waves per workgroup: 1 by default (LDS image private to the wave, no inter-wave synchronisation at all). -DNM_WG_WAVES=w (measurement,
// DESIGN.md 6.1) packs w waves into one workgroup - w times fewer workgroups for the dispatcher to start - with wave-private env images and
// ONE shared copy of the model constants and of the launch arguments; the two barriers of the start-up are then real workgroup barriers.
#ifdef NM_WG_WAVES
constexpr int kWG = NM_WG_WAVES;
#else
constexpr int kWG = 1;
#endif
template <class real, int G>
__global__ void __launch_bounds__(64 * (sizeof(real) == 8 ? 1 : kWG), NM_WAVES_PER_SIMD) k_env_step(const nm::Model<real>* __restrict__ Mp, nm::Args<real> A) {
  constexpr int kWG = sizeof(real) == 8 ? 1 : ::kWG;     // the fp64 verification build keeps one wave per workgroup
  __shared__ nm::ShW<real, G> shs[kWG];
  __shared__ nm::Model<real> Ms;   // this workgroup's copy of the model constants
  __shared__ nm::Args<real> As;    // ... and of the launch arguments: ~30 pointers would otherwise pin 60 SGPRs for the whole kernel
  nm::ShW<real, G>& sh = shs[kWG == 1 ? 0 : (int)(threadIdx.x >> 6)];
  // XCD-aware block -> wave mapping: the dispatcher deals workgroups round-robin over the 8 XCDs (block b runs on XCD b % 8), each with
  // its own L2. Consecutive envs share cache lines (rows of 100 / 96 / 72 bytes), so each XCD takes a CONTIGUOUS eighth of the waves:
  // a line's bytes are then written through one L2 instead of being merged in memory from two.
  int wave = blockIdx.x;
#ifndef NM_NO_XCD_MAP
  {
    const int nwx = (int)gridDim.x >> 3;          // workgroups per XCD (the remainder, if any, keeps the identity mapping)
    if (A.nxcd == 8 && wave < (nwx << 3)) wave = (wave & 7) * nwx + (wave >> 3);   // other partition modes (CPX, DPX): identity
  }
#endif
  if (kWG > 1) wave = wave * kWG + (int)(threadIdx.x >> 6);
#ifdef NM_MEASURE
  if (A.ablate & 512) return;      // measurement only: the empty launch
#endif
  const unsigned long long t_start = A.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
  if (kWG == 1 || threadIdx.x < 64) As = A;
  __syncthreads();
  if (wave * G >= A.N) return;     // (a wave that has left does not count for the barrier of the model copy below)
  // the model copy (L2 -> LDS) runs inside the load stage, after the env rows' HBM reads have been issued: one start-up round trip, not two
  auto copy_model = [&]() {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(Mp);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&Ms);
    constexpr int kWords = (int)(sizeof(nm::Model<real>) / 4);
    uint32_t tmp[(kWords + 63) / 64];
#pragma unroll
    for (int k = 0; k < (kWords + 63) / 64; k++) { const int i = NM_TID + 64 * k; tmp[k] = i < kWords ? src[i] : 0u; }
    if (kWG == 1 || threadIdx.x < 64) {     // one wave fills the workgroup's copy
#pragma unroll
      for (int k = 0; k < (kWords + 63) / 64; k++) { const int i = NM_TID + 64 * k; if (i < kWords) dst[i] = tmp[k]; }
    }
    __syncthreads();
  };
  // Two-level ticket for "which wave closes the step": waves draw from their group's counter, the last wave of a group from the top
  // counter - at most 64 + 32 same-address atomics in a row instead of gridDim.x. A wave draws its group ticket as soon as everything
  // it contributes to the bookkeeping is published (inside the epilogue, before rewards and observation), so the round trip of that
  // atomic is off the critical path of the wave that finishes last.
  const int nw = (As.N + G - 1) / G, grp = wave / nm::kTicketGroup, ngrp = (nw + nm::kTicketGroup - 1) / nm::kTicketGroup;
  const int gsize = min(nm::kTicketGroup, nw - grp * nm::kTicketGroup);
  int ticket = 0, top = 0;
  int stage = 0;                       // 0: nothing drawn, 1: group ticket drawn, 2: group ticket resolved (and the top one drawn if this wave closes its group)
  bool closes_group = false;
  auto draw = [&](int phase) {
#ifdef NM_NO_TICKETS   // measurement only (results without extras): what the end-of-step tickets cost
    stage = 2; (void)phase; return;
#endif
    if (phase == 0) {
      if (NM_TID == 0) ticket = atomicAdd(As.wave_done + (grp + 1) * nm::kTicketStride, 1);
      stage = 1;
    } else {
      closes_group = __builtin_amdgcn_readfirstlane(ticket) == gsize - 1;
      if (closes_group && NM_TID == 0) {
        As.wave_done[(grp + 1) * nm::kTicketStride] = 0;     // every member has drawn: re-arm for the next launch
        top = atomicAdd(As.wave_done + nm::kTicketTop, 1);
      }
      stage = 2;
    }
  };
  nm::wave_step<real, G>(sh, Ms, As, wave, copy_model, draw);
  if (As.dbg && NM_TID == 0) {   // debug buffer only: start / end clock of this wave as exact 24-bit pieces (scripts/wavetimes.py)
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    real* d = As.dbg + (size_t)(wave * G) * nm::kDbgN + 250;
    d[0] = (real)(unsigned)(t_start & 0xFFFFFF); d[1] = (real)(unsigned)((t_start >> 24) & 0xFFFFFF);
    d[2] = (real)(unsigned)(t_end & 0xFFFFFF); d[3] = (real)(unsigned)((t_end >> 24) & 0xFFFFFF);
    // where the wave ran: HW_ID (hwreg 4: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]) and XCC_ID (hwreg 20)
    d[4] = (real)(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)) & 0xFFFF);
    d[5] = (real)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xF);
  }
  if (As.physics_only) return;
  // The wave that finishes last closes the step (what used to be a second launch). What it needs from the others went through
  // device-scope atomics whose results each wave has already consumed (nm_consume), so the tickets need no fence.
  if (stage < 1) draw(0);              // paths that do not run the two-env epilogue draw here
  if (stage < 2) draw(1);
  if (!closes_group) return;
  if (__builtin_amdgcn_readfirstlane(top) != ngrp - 1) return;
#ifndef NM_SKIP_TAIL2
  step_tail<real>(As, Ms.ep_len_s);
#endif
}


// reset_idx (reference envs/nightmare_v3_env.py:335-371): one thread per env to reset
template <class real>
__global__ void k_reset(nm::Model<real> M, nm::Args<real> A, const int32_t* ids, int n) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int env = ids ? ids[j] : j;
  for (int k = 0; k < nm::kNQ; k++) A.qpos[env * nm::kNQ + k] = M.qpos0[k];
  for (int k = 0; k < nm::kNV; k++) A.qvel[env * nm::kNV + k] = real(0);
  real ux, uy;
  uint32_t ctr = A.rngctr[env];
  if (A.cmd_u) { ux = A.cmd_u[env * 4 + 2]; uy = A.cmd_u[env * 4 + 3]; }
  else {
    ux = (real)nm::rand_u24_bits(A.seed, (uint64_t)(A.env_offset + env), ctr) * real(1.0 / 16777216.0);
    uy = (real)nm::rand_u24_bits(A.seed, (uint64_t)(A.env_offset + env), ctr + 1) * real(1.0 / 16777216.0);
    ctr += 2;
  }
  real c0 = ux * real(2) * M.max_lin_x - M.max_lin_x, c2 = uy * real(2) * M.max_ang - M.max_ang;
  real keep = sqrt(c0 * c0) > real(0.02) ? real(1) : real(0);
  A.cmd[env * 3] = c0 * keep; A.cmd[env * 3 + 1] = real(0); A.cmd[env * 3 + 2] = c2;
  A.rngctr[env] = ctr;
  A.eplen[env] = 0;
  for (int k = 0; k < nm::kNLEG; k++) A.feetair[env * nm::kNLEG + k] = real(0);   // env.py:359 (last_contacts* are not cleared)
  for (int k = 0; k < nm::kNREW; k++) {
    atomicAdd(A.stat_sum + k, A.epsum[env * nm::kNREW + k]);
    A.epsum[env * nm::kNREW + k] = real(0);
  }
  atomicAdd(A.stat_cnt, 1);
}

// the same closing step as a kernel of its own, for reset_idx() (k_reset fills the accumulators, nothing else is in flight)
template <class real>
__global__ void k_finalize(int N, real* stat_sum, int* stat_cnt, float* ep_stats, const float* timeout_now, float* time_outs,
                           real ep_len_s, long long* counters) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = stat_cnt[0];
  __syncthreads();
  if (cnt > 0) {
    if (threadIdx.x < nm::kNREW && ep_stats) ep_stats[threadIdx.x] = (float)(stat_sum[threadIdx.x] / (real)cnt / ep_len_s);
    if (time_outs && timeout_now)
      for (int i = threadIdx.x; i < N; i += blockDim.x) time_outs[i] = timeout_now[i];
  }
  __syncthreads();
  if (threadIdx.x < nm::kNREW) stat_sum[threadIdx.x] = real(0);
  if (threadIdx.x == 0) {
    stat_cnt[0] = 0;
    counters[0] += stat_cnt[1];
    counters[1] += stat_cnt[2];
    counters[2] += stat_cnt[3];
    stat_cnt[1] = 0;
    stat_cnt[2] = 0;
    stat_cnt[3] = 0;
  }
}

// ------------------------------------------------------------------------------------------------ host object
struct nm_env {
  int N = 0, device = 0, dtype = 0;
  virtual ~nm_env() {}
  virtual int reset(const int32_t* ids, int n, int64_t* eplen, float* ep_stats, hipStream_t s) = 0;
  virtual int step(const float* actions, int64_t* eplen, float* obs, float* rew, int64_t* done, float* time_outs, float* ep_stats,
                   int physics_only, hipStream_t s) = 0;
  virtual int get_state(double* qpos, double* qvel, double* qw) = 0;
  virtual int set_state(const double* qpos, const double* qvel, const double* qw) = 0;
  virtual int get_buffers(double* dp, double* dv, double* act, double* cmd, double* es) = 0;
  virtual int set_buffers(const double* dp, const double* dv, const double* act, const double* cmd, const double* es) = 0;
  virtual int set_cmd_u(const double* u) = 0;
  virtual int get_feet(double* air, unsigned char* last, unsigned char* filt) = 0;
  virtual int set_feet(const double* air, const unsigned char* last, const unsigned char* filt) = 0;
  virtual int counters(int64_t* out) = 0;
  virtual void set_dbg(void* p) = 0;
  virtual void set_ret_acc(float* p) = 0;
  virtual int invalidate_time_outs(hipStream_t s) = 0;
  virtual int rollout(const nm_rollout_args* r, hipStream_t s) = 0;
  virtual int rollout_act(const float* flat, const float* obs, uint64_t seed, const int64_t* iter_dev, int step, float* actions, float* logp, float* values,
                          float* mu, float* sigma, float* obs_store, hipStream_t s) = 0;
  virtual int profiling(int on, double* sum_ms, int64_t* count) = 0;
  virtual void set_ablate(int m) = 0;
  virtual int set_noise(const double* vec) = 0;
  virtual int set_noise_u(const double* u) = 0;
  virtual int set_record(int idx) = 0;
  virtual int get_record(double* qpos, double* qvel, int32_t* nbad) = 0;
};

template <class real> struct Env : nm_env {
  nmhost::Tables<real> T;
  nm::Model<real> M;
  nm::Args<real> A;
  std::vector<void*> allocs;
  real* cmd_u_dev = nullptr;
  bool cmd_u_on = false;
  float* timeout_now = nullptr;
  int32_t* ids_dev = nullptr;
  long long* counters_dev = nullptr;
  nm::Model<real>* M_dev = nullptr;
  real* noise_vec_dev = nullptr;
  real* noise_u_dev = nullptr;
  bool noise_on = false, noise_u_on = false;
  uint64_t noise_step = 0;
  real* rec_dev = nullptr;
  int rec_env = -1;
  bool prof_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
  size_t prof_used = 0;

  // All device memory of an env comes out of ONE slab (a multiple of 2 MiB): the driver maps large allocations with 2 MiB
  // page-table fragments, so the shared hull tables and the per-env rows stay within a handful of TLB entries instead of one
  // 4 KiB-granular mapping per small hipMalloc.
  char* slab = nullptr;
  size_t slab_size = 0, slab_used = 0;
  int slab_init(size_t bytes) {
    slab_size = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    HIPCHK(hipMalloc((void**)&slab, slab_size));
    HIPCHK(hipMemset(slab, 0, slab_size));
    allocs.push_back(slab);
    return 0;
  }
  template <class X> int dalloc(X** p, size_t n) {
    const size_t bytes = (n * sizeof(X) + 255) & ~(size_t)255;
    if (slab && slab_used + bytes <= slab_size) {
      *p = (X*)(slab + slab_used);
      slab_used += bytes;
      return 0;
    }
    HIPCHK(hipMalloc((void**)p, n * sizeof(X)));
    HIPCHK(hipMemset(*p, 0, n * sizeof(X)));
    allocs.push_back(*p);
    return 0;
  }
  template <class X> int upload(const X** p, const std::vector<X>& v) {
    X* d;
    if (dalloc(&d, v.size())) return 1;
    HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(X), hipMemcpyHostToDevice));
    *p = d;
    return 0;
  }
  int init(const nmhost::EnvConfig& cfg, int n, int dev, uint64_t seed, int64_t off) {
    N = n; device = dev; dtype = sizeof(real) == 8;
    HIPCHK(hipSetDevice(dev));
    T.build();
    memset(&M, 0, sizeof M);
    memset(&A, 0, sizeof A);
    T.fill_scalars(M, cfg);
#ifdef NM_MEASURE   // measurement builds only (results change): cost per solver sweep
    if (const char* e = getenv("NM_MEASURE_PGS_ITERS")) M.pgs_iters = atoi(e);
    if (const char* e = getenv("NM_MEASURE_NOSLIP_ITERS")) M.noslip_iters = atoi(e);
#endif
    if (slab_init((T.hullv.size() + T.hullnv.size()) * sizeof(real) + (size_t)n * 300 * sizeof(real) + (size_t)n * 64 + (1u << 20))) return 1;
    if (upload(&M.hullv, T.hullv) || upload(&M.hullnv, T.hullnv)) return 1;
    A.N = N; A.seed = seed; A.env_offset = off; A.nsub = cfg.decimation;
    {
      int nx = 0;     // the dispatcher's round-robin width: 8 on an MI355X in SPX mode; anything else switches the remap off
      if (hipDeviceGetAttribute(&nx, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess) { (void)hipGetLastError(); nx = 0; }
      A.nxcd = nx;
    }
    size_t n_ = (size_t)N;
    if (dalloc(&A.qpos, n_ * 25) || dalloc(&A.qvel, n_ * 24) || dalloc(&A.qwarm, n_ * 24) || dalloc(&A.dofpos, n_ * 18) ||
        dalloc(&A.dofvel, n_ * 18) || dalloc(&A.act, n_ * 18) || dalloc(&A.cmd, n_ * 3) || dalloc(&A.epsum, n_ * nm::kNREW) || dalloc(&A.feetair, n_ * nm::kNLEG) || dalloc(&A.feetflags, n_) ||
        dalloc(&A.rngctr, n_) || dalloc(&A.hullcache, n_ * 8) || dalloc(&A.stat_sum, nm::kNREW) || dalloc(&A.stat_cnt, 4) || dalloc(&cmd_u_dev, n_ * 4) ||
        dalloc(&timeout_now, n_) || dalloc(&ids_dev, n_) || dalloc(&counters_dev, 4) || dalloc(&A.wave_done, ((n_ + nm::kTicketGroup - 1) / nm::kTicketGroup + 2) * nm::kTicketStride) || dalloc(&A.nto, 1) || dalloc(&A.to_list, n_) || dalloc(&A.nprev, 1) || dalloc(&A.to_prev, n_) || dalloc(&A.to_owner, 1))
      return 1;
    if (dalloc(&M_dev, 1)) return 1;
    HIPCHK(hipMemcpy(M_dev, &M, sizeof M, hipMemcpyHostToDevice));
    std::vector<real> q0(n_ * 25);
    for (size_t i = 0; i < n_; i++)
      for (int k = 0; k < 25; k++) q0[i * 25 + k] = T.qpos0[k];
    HIPCHK(hipMemcpy(A.qpos, q0.data(), q0.size() * sizeof(real), hipMemcpyHostToDevice));
    return 0;
  }
  ~Env() override {
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();   // kernels of this env may still be in flight on the caller's stream
    for (auto& ev : prof_ev) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (void* p : allocs) (void)hipFree(p);
  }
  int finalize(float* ep_stats, float* time_outs, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize<real>, dim3(1), dim3(1024), 0, s, N, A.stat_sum, A.stat_cnt, ep_stats, timeout_now, time_outs, M.ep_len_s,
                       counters_dev);
    HIPCHK(hipGetLastError());
    return 0;
  }
  int reset(const int32_t* ids, int n, int64_t* eplen, float* ep_stats, hipStream_t s) override {
    HIPCHK(hipSetDevice(device));
    if (!eplen) return fail("nm_reset: episode_length_dev is NULL");
    const int32_t* idp = nullptr;
    if (ids) {
      if (n <= 0) return 0;  // reset_idx returns early on an empty id list (env.py:344)
      for (int i = 0; i < n; i++)
        if (ids[i] < 0 || ids[i] >= N) return fail("nm_reset: env id out of range");
      HIPCHK(hipMemcpyAsync(ids_dev, ids, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
      HIPCHK(hipStreamSynchronize(s));   // the caller's host array may go away as soon as we return
      idp = ids_dev;
    } else n = N;
    nm::Args<real> a = A;
    a.eplen = eplen;
    a.cmd_u = cmd_u_on ? cmd_u_dev : nullptr;
    hipLaunchKernelGGL(k_reset<real>, dim3((n + 255) / 256), dim3(256), 0, s, M, a, idp, n);
    HIPCHK(hipGetLastError());
    return finalize(ep_stats, nullptr, s);
  }
  int step(const float* actions, int64_t* eplen, float* obs, float* rew, int64_t* done, float* time_outs, float* ep_stats, int physics_only,
           hipStream_t s) override {
    HIPCHK(hipSetDevice(device));
    if (!actions) return fail("nm_step: actions_dev is NULL");
    if (!physics_only && (!eplen || !obs || !rew || !done)) return fail("nm_step: NULL output pointer");
    nm::Args<real> a = A;
    a.actions = actions; a.eplen = eplen; a.obs = obs; a.rew = rew; a.done = done; a.timeout_now = timeout_now;
    a.cmd_u = cmd_u_on ? cmd_u_dev : nullptr;
    a.physics_only = physics_only;
    a.ep_stats = ep_stats; a.time_outs = time_outs; a.counters = counters_dev;
    if (!physics_only) {
      a.noise_vec = noise_on ? noise_vec_dev : nullptr;
      a.noise_u = noise_on && noise_u_on ? noise_u_dev : nullptr;
      a.noise_step = noise_step++;
      a.rec = rec_env >= 0 ? rec_dev : nullptr;
      a.rec_env = rec_env;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof_on) {  // HIP events on the launch stream around the dominant kernel only (bench.py's roofline leg)
      if (prof_used == prof_ev.size()) {
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        prof_ev.push_back({e0, e1});
      }
      e0 = prof_ev[prof_used].first; e1 = prof_ev[prof_used].second;
      prof_used++;
      HIPCHK(hipEventRecord(e0, s));
    }
    constexpr int G = sizeof(real) == 8 ? 1 : NM_ENVS_PER_WAVE;  // the fp64 verification build keeps one env per wave (LDS)
    constexpr int W = sizeof(real) == 8 ? 1 : kWG;
    hipLaunchKernelGGL((k_env_step<real, G>), dim3((N + G * W - 1) / (G * W)), dim3(64 * W), 0, s, (const nm::Model<real>*)M_dev, a);
    HIPCHK(hipGetLastError());
    if (prof_on) HIPCHK(hipEventRecord(e1, s));
    return 0;
  }
  int d2h(const real* dev, double* host, size_t n) {
    if (!host) return 0;
    std::vector<real> tmp(n);
    HIPCHK(hipMemcpy(tmp.data(), dev, n * sizeof(real), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) host[i] = (double)tmp[i];
    return 0;
  }
  int h2d(real* dev, const double* host, size_t n) {
    if (!host) return 0;
    std::vector<real> tmp(n);
    for (size_t i = 0; i < n; i++) tmp[i] = (real)host[i];
    HIPCHK(hipMemcpy(dev, tmp.data(), n * sizeof(real), hipMemcpyHostToDevice));
    return 0;
  }
  int get_state(double* qpos, double* qvel, double* qw) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    return d2h(A.qpos, qpos, (size_t)N * 25) || d2h(A.qvel, qvel, (size_t)N * 24) || d2h(A.qwarm, qw, (size_t)N * 24);
  }
  int set_state(const double* qpos, const double* qvel, const double* qw) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    return h2d(A.qpos, qpos, (size_t)N * 25) || h2d(A.qvel, qvel, (size_t)N * 24) || h2d(A.qwarm, qw, (size_t)N * 24);
  }
  int get_buffers(double* dp, double* dv, double* act, double* cmd, double* es) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    return d2h(A.dofpos, dp, (size_t)N * 18) || d2h(A.dofvel, dv, (size_t)N * 18) || d2h(A.act, act, (size_t)N * 18) ||
           d2h(A.cmd, cmd, (size_t)N * 3) || d2h(A.epsum, es, (size_t)N * nm::kNREW);
  }
  int set_buffers(const double* dp, const double* dv, const double* act, const double* cmd, const double* es) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    return h2d(A.dofpos, dp, (size_t)N * 18) || h2d(A.dofvel, dv, (size_t)N * 18) || h2d(A.act, act, (size_t)N * 18) ||
           h2d(A.cmd, cmd, (size_t)N * 3) || h2d(A.epsum, es, (size_t)N * nm::kNREW);
  }
  int get_feet(double* air, unsigned char* last, unsigned char* filt) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    if (d2h(A.feetair, air, (size_t)N * 6)) return 1;
    std::vector<int> fl(N);
    HIPCHK(hipMemcpy(fl.data(), A.feetflags, sizeof(int) * N, hipMemcpyDeviceToHost));
    for (int i = 0; i < N; i++)
      for (int k = 0; k < 6; k++) {
        if (last) last[i * 6 + k] = (fl[i] >> k) & 1;
        if (filt) filt[i * 6 + k] = (fl[i] >> (6 + k)) & 1;
      }
    return 0;
  }
  int set_feet(const double* air, const unsigned char* last, const unsigned char* filt) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    if (h2d(A.feetair, air, (size_t)N * 6)) return 1;
    if (last || filt) {
      std::vector<int> fl(N);
      HIPCHK(hipMemcpy(fl.data(), A.feetflags, sizeof(int) * N, hipMemcpyDeviceToHost));
      for (int i = 0; i < N; i++)
        for (int k = 0; k < 6; k++) {
          if (last) fl[i] = (fl[i] & ~(1 << k)) | ((last[i * 6 + k] ? 1 : 0) << k);
          if (filt) fl[i] = (fl[i] & ~(1 << (6 + k))) | ((filt[i * 6 + k] ? 1 : 0) << (6 + k));
        }
      HIPCHK(hipMemcpy(A.feetflags, fl.data(), sizeof(int) * N, hipMemcpyHostToDevice));
    }
    return 0;
  }
  int set_cmd_u(const double* u) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    cmd_u_on = u != nullptr;
    return h2d(cmd_u_dev, u, (size_t)N * 4);
  }
  int counters(int64_t* out) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    long long c[3];
    HIPCHK(hipMemcpy(c, counters_dev, sizeof c, hipMemcpyDeviceToHost));
    std::vector<int> hc((size_t)N * 8);
    HIPCHK(hipMemcpy(hc.data(), A.hullcache, hc.size() * sizeof(int), hipMemcpyDeviceToHost));
    long long fb = 0;
    for (int i = 0; i < N; i++) fb += hc[(size_t)i * 8 + 7];    // per-env running counts (wrap after 2^31 fallbacks of one env)
    out[0] = c[0]; out[1] = c[1]; out[2] = fb;
    return 0;
  }
  int set_noise(const double* vec) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    noise_on = vec != nullptr;
    if (vec && !noise_vec_dev && dalloc(&noise_vec_dev, NM_NUM_OBS)) return 1;
    return vec ? h2d(noise_vec_dev, vec, NM_NUM_OBS) : 0;
  }
  int set_noise_u(const double* u) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    noise_u_on = u != nullptr;
    if (u && !noise_u_dev && dalloc(&noise_u_dev, (size_t)N * NM_NUM_OBS)) return 1;
    return u ? h2d(noise_u_dev, u, (size_t)N * NM_NUM_OBS) : 0;
  }
  int set_record(int idx) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    if (idx >= N) return fail("nm_set_state_record: env index out of range");
    if (idx >= 0 && !rec_dev && dalloc(&rec_dev, 64)) return 1;
    rec_env = idx < 0 ? -1 : idx;
    return 0;
  }
  int get_record(double* qpos, double* qvel, int32_t* nbad) override {
    HIPCHK(hipSetDevice(device));
    if (rec_env < 0) return fail("nm_get_state_record: recording is off (nm_set_state_record)");
    HIPCHK(hipDeviceSynchronize());
    double tmp[50];
    if (d2h(rec_dev, tmp, 50)) return 1;
    if (qpos) for (int i = 0; i < 25; i++) qpos[i] = tmp[i];
    if (qvel) for (int i = 0; i < 24; i++) qvel[i] = tmp[25 + i];
    if (nbad) *nbad = (int32_t)tmp[49];
    return 0;
  }

  // ---- K-step rollout with the policy in the wave (nm_rollout; fp32 two-env waves and the reference's network shape only)
  float *roll_wp = nullptr, *roll_bp = nullptr;
  real* roll_sum = nullptr;
  int *roll_cnt = nullptr, *roll_to = nullptr;
  int roll_cap = 0;
  typedef nmr::RefShape RS;
  int roll_pack(const float* flat, hipStream_t s) {
    if (!roll_wp && (dalloc(&roll_wp, (size_t)RS::nfrag() * 256) || dalloc(&roll_bp, (size_t)RS::nbias()))) return 1;
    if (nmr::launch_pack(flat, roll_wp, roll_bp, s)) return fail("nm_rollout: packing the policy failed to launch");
    return 0;
  }
  int rollout_act(const float* flat, const float* obs, uint64_t seed, const int64_t* iter_dev, int step, float* actions, float* logp, float* values,
                  float* mu, float* sigma, float* obs_store, hipStream_t s) override {
    HIPCHK(hipSetDevice(device));
    if (!flat || !obs || !iter_dev || !actions || !logp || !values || !mu || !sigma) return fail("nm_rollout_act: NULL pointer");
    if (roll_pack(flat, s)) return 1;
    nmr::ActOut o{actions, logp, values, mu, sigma, obs_store};
    if (nmr::launch_act(roll_wp, roll_bp, flat + RS::stdoff(), obs, N, seed, iter_dev, step, o, s)) return fail("nm_rollout_act: launch failed");
    return 0;
  }
  int rollout(const nm_rollout_args* r, hipStream_t s) override {
    HIPCHK(hipSetDevice(device));
    if constexpr (sizeof(real) == 8 || NM_ENVS_PER_WAVE != 2) {
      return fail("nm_rollout: the fused rollout runs on the fp32 kernel (two envs per wave) only");
    } else {
      if (!r) return fail("nm_rollout: args is NULL");
      const int K = r->steps;
      if (K < 1 || K > 4096) return fail("nm_rollout: steps must be in 1..4096");
      if ((double)K > (double)M.max_ep_len) return fail("nm_rollout: more steps than an episode has (an env may time out once per rollout)");
      if (!r->params_flat_dev || !r->iter_dev || !r->obs0_dev || !r->obs_final_dev || !r->episode_length_dev || !r->rew_dev || !r->done_dev || !r->s_obs ||
          !r->s_actions || !r->s_logp || !r->s_values || !r->s_mu || !r->s_sigma || !r->s_rewards || !r->s_dones || !r->cur_ret || !r->cur_len || !r->fin3)
        return fail("nm_rollout: NULL pointer");
      if (r->n_ep < 0 || r->n_ep > nm::kNREW || (r->n_ep > 0 && (!r->ep_idx_dev || !r->ep_acc_dev || !r->ep_stats_dev))) return fail("nm_rollout: bad episode-statistics arguments");
      if (rec_env >= 0) return fail("nm_rollout: the state log (nm_set_state_record) needs one launch per step");
      if (K > roll_cap) {   // per-step accumulators (zero: the slab and hipMalloc'ed blocks are cleared, k_rollout_clear keeps them so)
        const int cap = (K + 127) & ~127;
        if (dalloc(&roll_sum, (size_t)cap * nm::kNREW) || dalloc(&roll_cnt, (size_t)cap * 4)) return 1;
        if (!roll_to && dalloc(&roll_to, (size_t)N)) return 1;
        roll_cap = cap;
      }
      if (roll_pack(r->params_flat_dev, s)) return 1;
      nm::Args<real> a = A;
      a.actions = nullptr; a.eplen = r->episode_length_dev; a.obs = r->obs_final_dev; a.rew = r->rew_dev; a.done = r->done_dev; a.timeout_now = timeout_now;
      a.cmd_u = cmd_u_on ? cmd_u_dev : nullptr;
      a.physics_only = 0;
      a.ep_stats = nullptr; a.time_outs = nullptr; a.counters = counters_dev;     // extras are closed by k_rollout_tail
      a.to_list = nullptr;
      a.noise_vec = noise_on ? noise_vec_dev : nullptr;
      a.noise_u = noise_on && noise_u_on ? noise_u_dev : nullptr;
      a.noise_step = noise_step;
      noise_step += (uint64_t)K;
      a.rec = nullptr; a.rec_env = -1; a.dbg = nullptr; a.ret_acc = nullptr;
      nmr::RollArgs R;
      R.K = K; R.wp = (const nmr::f32x4*)roll_wp; R.bp = roll_bp; R.stdv = r->params_flat_dev + RS::stdoff();
      R.seed = r->seed; R.iter_dev = r->iter_dev; R.obs0 = r->obs0_dev; R.obs_final = r->obs_final_dev;
      R.s_obs = r->s_obs; R.s_actions = r->s_actions; R.s_logp = r->s_logp; R.s_values = r->s_values; R.s_mu = r->s_mu; R.s_sigma = r->s_sigma;
      R.s_rewards = r->s_rewards; R.s_dones = r->s_dones; R.cur_ret = r->cur_ret; R.cur_len = r->cur_len; R.fin3 = r->fin3;
      R.st_sum = roll_sum; R.st_cnt = roll_cnt; R.to_step = roll_to;
      R.last_values = r->last_values_dev;
      R.wave_clock = A.dbg ? reinterpret_cast<unsigned long long*>(A.dbg) : nullptr;   // measurement: the debug buffer ([N,256] reals) takes the waves' clocks instead
      nmr::TailArgs ta{N, K, roll_sum, roll_cnt, roll_to, r->ep_stats_dev, r->time_outs_dev, M.ep_len_s, counters_dev, r->bootstrap_time_outs ? r->gamma : -1.0f, r->s_values, r->s_rewards,
                       r->ep_idx_dev, r->n_ep, r->ep_acc_dev, A.to_owner};
      if (nmr::launch_rollout(M_dev, a, R, ta, s)) return fail("nm_rollout: launch failed");
      return 0;
    }
  }
  void set_dbg(void* p) override { A.dbg = (real*)p; }
  void set_ret_acc(float* p) override { A.ret_acc = p; }
  int invalidate_time_outs(hipStream_t s) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemsetAsync(A.to_owner, 0, sizeof(unsigned long long), s));   // no buffer is "the one the last refresh wrote" any more
    return 0;
  }
  void set_ablate(int m) override { A.ablate = m; }
  int profiling(int on, double* sum_ms, int64_t* count) override {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    double sum = 0;
    for (size_t i = 0; i < prof_used; i++) {
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, prof_ev[i].first, prof_ev[i].second));
      sum += ms;
    }
    if (sum_ms) *sum_ms = sum;
    if (count) *count = (int64_t)prof_used;
    prof_used = 0;
    prof_on = on != 0;
    return 0;
  }
};

// ------------------------------------------------------------------------------------------------ C ABI
static const char* kRewardNames[NM_NUM_REWARDS] = {"action_rate", "ang_vel_xy", "base_height", "body_contact_forces", "default_position", "dof_acc",
                                                   "dof_vel", "feet_air_time", "feet_contact_forces", "lin_vel_z", "orientation", "stand_still",
                                                   "torques", "tracking_ang_vel", "tracking_lin_vel", "termination"};
static_assert(NM_NUM_REWARDS == nm::kNREW, "reward table size");
extern "C" {
const char* nm_last_error(void) { return g_err.c_str(); }
int nm_policy_set_error(const char* m) { return fail(m); }
const char* nm_reward_name(int i) { return (i >= 0 && i < NM_NUM_REWARDS) ? kRewardNames[i] : ""; }
void nm_default_config(nm_config* c) {
  nmhost::EnvConfig d;
  c->decimation = d.decimation; c->p_gain = d.p_gain; c->action_scale = d.action_scale;
  for (int i = 0; i < 3; i++) c->default_pos[i] = d.default_pos[i];
  c->clip_actions = d.clip_actions; c->clip_observations = d.clip_observations;
  c->obs_lin_vel = d.obs_lin_vel; c->obs_ang_vel = d.obs_ang_vel; c->obs_dof_pos = d.obs_dof_pos; c->obs_dof_vel = d.obs_dof_vel;
  c->episode_length_s = d.episode_length_s; c->resampling_time = d.resampling_time;
  c->max_lin_vel_x = d.max_lin_vel_x; c->max_ang_vel = d.max_ang_vel;
  c->termination_contact_force = d.termination_contact_force; c->tracking_sigma = d.tracking_sigma;
  for (int i = 0; i < NM_NUM_REWARDS; i++) c->reward_scales[i] = d.rew_scales[i];
  c->tibia_contact_mode = d.tibia_contact_mode; c->tibia_max_contact_force = d.tibia_max_contact_force;
  c->body_contact_mode = d.body_contact_mode; c->body_max_contact_force = d.body_max_contact_force;
  c->base_height_target = d.base_height_target; c->max_contact_force = d.max_contact_force;
}
int nm_create(const nm_config* cfg, int32_t num_envs, int32_t device, uint64_t seed, int64_t env_id_offset, int32_t dtype, nm_env** out) {
  if (!out) return fail("nm_create: out is NULL");
  *out = nullptr;
  if (num_envs <= 0) return fail("nm_create: num_envs must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail("nm_create: no HIP device available (this backend has no CPU path)");
  if (device < 0 || device >= ndev) return fail("nm_create: bad device index");
  nmhost::EnvConfig c;
  if (cfg) {
    if (cfg->decimation < 1) return fail("nm_create: decimation must be >= 1");
    if (!(cfg->episode_length_s > 0)) return fail("nm_create: episode_length_s must be positive");
    // env.py:235 takes episode_length % int(resampling_time / dt): a period below one step is a division by zero upstream
    if ((int)(cfg->resampling_time / (NM_TIMESTEP * cfg->decimation)) < 1) return fail("nm_create: resampling_time must be at least one env step (dt)");
    if (cfg->tibia_contact_mode < 0 || cfg->tibia_contact_mode > 2 || cfg->body_contact_mode < 0 || cfg->body_contact_mode > 2)
      return fail("nm_create: contact modes are 0 (ignore), 1 (penalise) or 2 (terminate)");
    c.decimation = cfg->decimation; c.p_gain = cfg->p_gain; c.action_scale = cfg->action_scale;
    for (int i = 0; i < 3; i++) c.default_pos[i] = cfg->default_pos[i];
    c.clip_actions = cfg->clip_actions; c.clip_observations = cfg->clip_observations;
    c.obs_lin_vel = cfg->obs_lin_vel; c.obs_ang_vel = cfg->obs_ang_vel; c.obs_dof_pos = cfg->obs_dof_pos; c.obs_dof_vel = cfg->obs_dof_vel;
    c.episode_length_s = cfg->episode_length_s; c.resampling_time = cfg->resampling_time;
    c.max_lin_vel_x = cfg->max_lin_vel_x; c.max_ang_vel = cfg->max_ang_vel;
    c.termination_contact_force = cfg->termination_contact_force; c.tracking_sigma = cfg->tracking_sigma;
    for (int i = 0; i < NM_NUM_REWARDS; i++) c.rew_scales[i] = cfg->reward_scales[i];
    c.tibia_contact_mode = cfg->tibia_contact_mode; c.tibia_max_contact_force = cfg->tibia_max_contact_force;
    c.body_contact_mode = cfg->body_contact_mode; c.body_max_contact_force = cfg->body_max_contact_force;
    c.base_height_target = cfg->base_height_target; c.max_contact_force = cfg->max_contact_force;
  }
  nm_env* e;
  int rc;
  if (dtype == NM_DTYPE_F64) { auto* p = new Env<double>(); rc = p->init(c, num_envs, device, seed, env_id_offset); e = p; }
  else if (dtype == NM_DTYPE_F32) { auto* p = new Env<float>(); rc = p->init(c, num_envs, device, seed, env_id_offset); e = p; }
  else return fail("nm_create: dtype must be NM_DTYPE_F32 or NM_DTYPE_F64");
  if (rc) { delete e; return 1; }
  *out = e;
  return 0;
}
int nm_destroy(nm_env* env) { delete env; return 0; }
int32_t nm_num_envs(const nm_env* env) { return env ? env->N : 0; }
int32_t nm_dtype(const nm_env* env) { return env ? env->dtype : -1; }
#define NEED(e) if (!(e)) return fail("null nm_env")
int nm_reset(nm_env* env, const int32_t* ids, int32_t n, int64_t* eplen, float* ep_stats, void* stream) {
  NEED(env); return env->reset(ids, n, eplen, ep_stats, (hipStream_t)stream);
}
int nm_step(nm_env* env, const float* actions, int64_t* eplen, float* obs, float* rew, int64_t* done, float* time_outs, float* ep_stats,
            void* stream) {
  NEED(env); return env->step(actions, eplen, obs, rew, done, time_outs, ep_stats, 0, (hipStream_t)stream);
}
int nm_step_physics(nm_env* env, const float* actions, void* stream) {
  NEED(env); return env->step(actions, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, (hipStream_t)stream);
}
int nm_get_state(nm_env* env, double* qpos, double* qvel, double* qw) { NEED(env); return env->get_state(qpos, qvel, qw); }
int nm_set_state(nm_env* env, const double* qpos, const double* qvel, const double* qw) { NEED(env); return env->set_state(qpos, qvel, qw); }
int nm_get_buffers(nm_env* env, double* dp, double* dv, double* act, double* cmd, double* es) { NEED(env); return env->get_buffers(dp, dv, act, cmd, es); }
int nm_set_buffers(nm_env* env, const double* dp, const double* dv, const double* act, const double* cmd, const double* es) {
  NEED(env); return env->set_buffers(dp, dv, act, cmd, es);
}
int nm_get_feet_state(nm_env* env, double* air, unsigned char* last, unsigned char* filt) { NEED(env); return env->get_feet(air, last, filt); }
int nm_set_feet_state(nm_env* env, const double* air, const unsigned char* last, const unsigned char* filt) { NEED(env); return env->set_feet(air, last, filt); }
int nm_set_command_uniforms(nm_env* env, const double* u) { NEED(env); return env->set_cmd_u(u); }
int nm_get_counters(nm_env* env, int64_t* out2) { NEED(env); return env->counters(out2); }
int nm_set_debug_buffer(nm_env* env, void* dbg) { NEED(env); env->set_dbg(dbg); return 0; }
int nm_set_return_accumulator(nm_env* env, float* acc) { NEED(env); env->set_ret_acc(acc); return 0; }
int nm_invalidate_time_outs(nm_env* env, void* stream) { NEED(env); return env->invalidate_time_outs((hipStream_t)stream); }
int nm_rollout_supported(const int32_t* actor_dims, const int32_t* critic_dims, int32_t n_layers) {
  typedef nmr::RefShape RS;
  if (!actor_dims || !critic_dims || n_layers != RS::NL) return 0;
  if (actor_dims[0] != RS::I || critic_dims[0] != RS::I) return 0;
  for (int l = 0; l < RS::NL; l++)
    if (actor_dims[l + 1] != RS::aout(l) || critic_dims[l + 1] != RS::cout(l)) return 0;
  return 1;
}
int nm_rollout(nm_env* env, const nm_rollout_args* args, void* stream) { NEED(env); return env->rollout(args, (hipStream_t)stream); }
int nm_rollout_act(nm_env* env, const float* flat, const float* obs, uint64_t seed, const int64_t* iter_dev, int32_t step, float* actions, float* logp,
                   float* values, float* mu, float* sigma, float* obs_store, void* stream) {
  NEED(env); return env->rollout_act(flat, obs, seed, iter_dev, step, actions, logp, values, mu, sigma, obs_store, (hipStream_t)stream);
}
#ifdef NM_MEASURE   // include/nightmare_hip_measure.h: not part of the shipped ABI
int nm_set_ablation(nm_env* env, int32_t mask) { NEED(env); env->set_ablate(mask); return 0; }
#endif
int nm_set_observation_noise(nm_env* env, const double* vec) { NEED(env); return env->set_noise(vec); }
int nm_set_noise_uniforms(nm_env* env, const double* u) { NEED(env); return env->set_noise_u(u); }
int nm_set_state_record(nm_env* env, int32_t idx) { NEED(env); return env->set_record(idx); }
int nm_get_state_record(nm_env* env, double* qpos, double* qvel, int32_t* nbad) { NEED(env); return env->get_record(qpos, qvel, nbad); }
#ifdef NM_STAMPS
int nm_read_stamps(unsigned long long* out16, int reset) {   // measurement builds only
  (void)hipDeviceSynchronize();
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(nm::g_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return fail("nm_read_stamps: copy failed");
  unsigned long long z[16] = {0};
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(nm::g_stamps), z, sizeof z) != hipSuccess) return fail("nm_read_stamps: reset failed");
  return 0;
}
#endif
int nm_profile(nm_env* env, int32_t enable, double* sum_ms, int64_t* count) { NEED(env); return env->profiling(enable, sum_ms, count); }
}
