// simt.h - the thin SIMT vocabulary the wave-per-env kernels are written in.
//
// Device build (hipcc, gfx950): V<T> is simply T (one value per lane of the 64-wide wavefront), masks are
// bool, cross-lane ops are CDNA wave intrinsics (v_readlane, DPP/ds_bpermute butterflies, s_ballot).
// Host build (-DNM_EMUL, used only by tests/emul): V<T> is an explicit 64-element vector and every
// statement runs for all 64 lanes in lockstep, so the *same kernel source* can be checked against the CPU
// oracle in a container without a GPU. The emulation is test scaffolding; the product library is device-only.
//
// Rules for kernel code written against this header:
//   * no per-lane `if`: use sel(mask, a, b) and masked stores; uniform control flow only
//   * masks combine with & | ! (never && ||)
//   * literals are wrapped: real(0.5)
#pragma once
#include <stdint.h>

#define NM_WAVE 64

#ifndef NM_EMUL
// =====================================================================================  DEVICE (gfx950)
#include <hip/hip_runtime.h>
#define NM_FN __device__ __forceinline__
#define NM_COLD __device__ __noinline__   /* rarely executed paths: keep them out of the hot instruction stream */
namespace simt {
template <class T> using V = T;
using VB = bool;

// lane of the thread inside its wavefront. One wave per workgroup (the default): the thread index itself. -DNM_WG_WAVES=w packs w waves
// into a workgroup (fewer workgroups for the dispatcher to start; LDS slices stay wave-private): the low six bits.
#ifdef NM_WG_WAVES
#define NM_TID ((int)(threadIdx.x & 63))
#else
#define NM_TID ((int)threadIdx.x)
#endif
NM_FN int lane_id() { return NM_TID; }
template <class T> NM_FN T sel(bool c, T a, T b) { return c ? a : b; }
NM_FN float vsqrt(float x) { return sqrtf(x); }
NM_FN double vsqrt(double x) { return sqrt(x); }
NM_FN float vabs(float x) { return fabsf(x); }
NM_FN double vabs(double x) { return fabs(x); }
NM_FN float vmax(float a, float b) { return fmaxf(a, b); }
NM_FN double vmax(double a, double b) { return fmax(a, b); }
NM_FN float vmin(float a, float b) { return fminf(a, b); }
NM_FN double vmin(double a, double b) { return fmin(a, b); }
NM_FN int vmin(int a, int b) { return a < b ? a : b; }
NM_FN int vmax(int a, int b) { return a > b ? a : b; }
// sin and cos of a joint angle: Cody-Waite reduction by pi/2 (two fma terms, exact products) + the cephes single-precision
// polynomials on [-pi/4, pi/4]: max error 9e-8 absolute (checked against float64 on 2 M points in [-12, 12]; the reduction adds
// < 1e-8 up to the clamp), 25 branch-free instructions instead of the library's general-range sincosf (1.3 us of the step kernel;
// a fallback BRANCH to the library for large arguments cost all of that again). Angles are clamped to +-2^22 rad - a joint would
// have to wind 600 000 turns; NaN never gets here (mj_checkPos runs first).
NM_FN void vsincos(float x0, float* s, float* c) {
  const float x = fminf(fmaxf(x0, -4194304.0f), 4194304.0f);
  const float k = rintf(x * 0.636619772f);
  float r = fmaf(-k, 1.5707963705062866f, x);
  r = fmaf(-k, -4.371139000186241e-8f, r);
  const float r2 = r * r;
  const float ps = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float pc = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(r2, -0.5f, 1.0f));
  const int n = (int)k;
  const float sa = (n & 1) ? pc : ps, ca = (n & 1) ? ps : pc;
  *s = (n & 2) ? -sa : sa;
  *c = ((n + 1) & 2) ? -ca : ca;
}
NM_FN void vsincos(double x, double* s, double* c) { sincos(x, s, c); }
NM_FN float vexp(float x) { return expf(x); }
NM_FN double vexp(double x) { return exp(x); }
NM_FN float vacos(float x) { return acosf(x); }
NM_FN double vacos(double x) { return acos(x); }
NM_FN float vpow(float x, float p) { return powf(x, p); }
NM_FN double vpow(double x, double p) { return pow(x, p); }
NM_FN bool visbad(float x) { return !(fabsf(x) <= 1e10f); }
NM_FN bool visbad(double x) { return !(fabs(x) <= 1e10); }
template <class T> NM_FN T to_real(int i) { return (T)i; }
template <class T, class S> NM_FN T vcvt(S x) { return (T)x; }   // per-lane value conversion (fp32 state -> fp64 MPR arithmetic)
// 1 / x. fp32: v_rcp_f32 (1 ulp) - the operands here (pivots of positive definite blocks, norms, impedances) are far from the
// denormal range, which is all that the compiler's division sequence (2.5 ulp + frexp / ldexp range scaling, ~8 instructions) adds;
// ~90 divisions per wave and step. The fp64 verification build divides exactly.
NM_FN float vrcp(float x) { return __builtin_amdgcn_rcpf(x); }
NM_FN double vrcp(double x) { return 1.0 / x; }

// a * b + c in ONE rounding, wherever it is written (the solver's update rules are stated once and must give the same bits in every
// row layout: the contraction is not left to the compiler's mood)
NM_FN float vfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
NM_FN double vfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// a in the lanes of the wave-uniform 64-bit lane mask m (an SGPR pair), b elsewhere: ONE v_cndmask. The sweeps walk their rows in lane
// order, so the mask of row i + 1 is the mask of row i shifted by one (s_lshl_b64): two instructions per kept value instead of three.
NM_FN float sel_mask(uint64_t m, float a, float b) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
NM_FN double sel_mask(uint64_t m, double a, double b) {
  int rl, rh;
  asm("v_cndmask_b32_e64 %0, %2, %3, %6\n\tv_cndmask_b32_e64 %1, %4, %5, %6" : "=&v"(rl), "=v"(rh)
      : "v"(__double2loint(b)), "v"(__double2loint(a)), "v"(__double2hiint(b)), "v"(__double2hiint(a)), "s"(m));
  return __hiloint2double(rh, rl);
}
// m << k as a value the optimiser cannot fold (the rows are unrolled, so it would otherwise turn the walking mask into one 64-bit literal
// per row and keep them all in scalar registers: 334 SGPR spills)
NM_FN uint64_t mask_shl(uint64_t m, int k) { m <<= k; asm("" : "+s"(m)); return m; }
// g + a * (value of x in lane I of THIS lane's half-wave): how the two-env constraint pass hands a row's change to the other rows of
// its env (lanes 0..31 = env 0, 32..63 = env 1): two v_readlane, two moves and a select build the per-half operand, then the
// multiply-add (6 VALU). Both lanes are read BEFORE the select (a `h1 ? rdlane : rdlane` compiles to exec-mask branches: +6 us on the
// step kernel). Measured and not adopted (round 5): two multiply-adds with a scalar operand each under the halves' exec masks (4 VALU
// + 3 SALU: +1.4 us on the kernel, 77 vs 58 ticks per row in scripts/micro/pgs_chain.hip - every instruction of a wave costs an issue
// slot, scalar ones included); the broadcast through ds_bpermute_b32 (one DS instruction, but 70 ticks of LDS latency per row).
template <int I> NM_FN float fma_half_lane(float a, float x, float g, bool h1);
template <int I> NM_FN double fma_half_lane(double a, double x, double g, bool h1);
// value of lane l (l wave-uniform) as a wave-uniform scalar
NM_FN int rdlane(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
NM_FN float rdlane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
NM_FN double rdlane(double x, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}
template <int I> NM_FN float fma_half_lane(float a, float x, float g, bool h1) {
  const float s0 = rdlane(x, I), s1 = rdlane(x, 32 + I);
  return vfma(a, h1 ? s1 : s0, g);
}
template <int I> NM_FN double fma_half_lane(double a, double x, double g, bool h1) {
  const double s0 = rdlane(x, I), s1 = rdlane(x, 32 + I);
  return vfma(a, h1 ? s1 : s0, g);
}
// x with lane l (wave-uniform) replaced by the wave-uniform value v. The lane id goes through an empty asm so
// the `lane == l` mask is recomputed here (v_cmp + v_cndmask) instead of being hoisted out of the solver's
// iteration loops by LICM, where 64+ live masks (2 SGPRs each) spill the scalar register file.
template <class T> NM_FN T wrlane(T x, T v, int l) {
  int ln = NM_TID;
  asm volatile("" : "+v"(ln));
  return ln == l ? v : x;
}
// lane id that the optimiser cannot see through: compares against it stay inside the loop they are written in (hoisting
// 64 loop-invariant `lane == i` masks out of the solver sweeps costs 128 SGPRs and spills the scalar file)
NM_FN int opaque_lane() { int ln = NM_TID; asm volatile("" : "+v"(ln)); return ln; }
template <class T> NM_FN T uniform(T x) { return rdlane(x, 0); }
NM_FN bool uniform(bool x) { return __builtin_amdgcn_readfirstlane((int)x) != 0; }
// ---- DPP lane permutations inside a row of 16 lanes (no LDS traffic, ~VALU latency)
#define NM_DPP_QUAD_XOR1 0xB1     /* quad_perm [1,0,3,2] */
#define NM_DPP_QUAD_XOR2 0x4E     /* quad_perm [2,3,0,1] */
#define NM_DPP_HALF_MIRROR 0x141  /* lane i <-> 7-i inside each 8 */
#define NM_DPP_ROW_MIRROR 0x140   /* lane i <-> 15-i inside each 16 */
template <int CTRL> NM_FN int dpp(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true); }
template <int CTRL> NM_FN float dpp(float x) { return __int_as_float(dpp<CTRL>(__float_as_int(x))); }
template <int CTRL> NM_FN double dpp(double x) { return __hiloint2double(dpp<CTRL>(__double2hiint(x)), dpp<CTRL>(__double2loint(x))); }
// quad_perm inside every aligned group of 4 lanes: lane q of a quad takes the value of lane ((P >> 2q) & 3) of its quad
template <int P, class T> NM_FN T quad(T x) { return dpp<P>(x); }
// value of lane (lane ^ 1) / (lane ^ 2)
template <class T> NM_FN T shfl_xor1(T x) { return dpp<NM_DPP_QUAD_XOR1>(x); }
template <class T> NM_FN T shfl_xor2(T x) { return dpp<NM_DPP_QUAD_XOR2>(x); }
// sum of lanes 0..7 (wave-uniform): pairs, quads, then the two quads of the first 8
template <class T> NM_FN T wsum8(T x) {
  x = x + dpp<NM_DPP_QUAD_XOR1>(x);
  x = x + dpp<NM_DPP_QUAD_XOR2>(x);
  x = x + dpp<NM_DPP_HALF_MIRROR>(x);
  return rdlane(x, 0);
}
// sum inside every aligned group of 8 lanes; each lane of the group gets the group total (no broadcast needed)
template <class T> NM_FN T gsum8(T x) {
  x = x + dpp<NM_DPP_QUAD_XOR1>(x);
  x = x + dpp<NM_DPP_QUAD_XOR2>(x);
  x = x + dpp<NM_DPP_HALF_MIRROR>(x);
  return x;
}
// sum of all 64 lanes (wave-uniform); fixed association order: 2,4,8,16 inside rows, then (r0+r1)+(r2+r3)
template <class T> NM_FN T wsum(T x) {
  x = x + dpp<NM_DPP_QUAD_XOR1>(x);
  x = x + dpp<NM_DPP_QUAD_XOR2>(x);
  x = x + dpp<NM_DPP_HALF_MIRROR>(x);
  x = x + dpp<NM_DPP_ROW_MIRROR>(x);
  return (rdlane(x, 0) + rdlane(x, 16)) + (rdlane(x, 32) + rdlane(x, 48));
}
// sum inside each half of the wave (lanes 0..31 | 32..63); every lane gets its half's total. Same association order as wsum
template <class T> NM_FN T hsum32(T x) {
  x = x + dpp<NM_DPP_QUAD_XOR1>(x);
  x = x + dpp<NM_DPP_QUAD_XOR2>(x);
  x = x + dpp<NM_DPP_HALF_MIRROR>(x);
  x = x + dpp<NM_DPP_ROW_MIRROR>(x);
  const T s0 = rdlane(x, 0) + rdlane(x, 16), s1 = rdlane(x, 32) + rdlane(x, 48);
  return (NM_TID & 32) ? s1 : s0;
}
NM_FN bool wany(bool c) { return __ballot(c) != 0ull; }
// number of set bits of the wave-uniform mask m below this lane (v_mbcnt)
NM_FN int lane_rank(uint64_t m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
NM_FN int popc64(uint64_t m) { return __popcll(m); }
NM_FN uint64_t ballot(bool c) { return __ballot(c); }
NM_FN bool in_mask(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }   // this lane's bit of a wave-uniform mask
// scheduling fence: what is computed from `a` afterwards cannot be issued before `dep` exists (keeps short-lived lane masks short-lived)
template <class T> NM_FN void order_after(int& a, const T& dep) { asm volatile("" : "+v"(a) : "v"(dep)); }
// argmax with lowest-index tie break; returns uniform (value, index)
template <class T> NM_FN void wargmax(T val, int idx, T* best, int* ibest) {
#define NM_AM_STEP(CTRL)                                                   \
  {                                                                        \
    T ov = dpp<CTRL>(val);                                                 \
    int oi = dpp<CTRL>(idx);                                               \
    bool take = (ov > val) | ((ov == val) & (oi < idx));                   \
    val = take ? ov : val;                                                 \
    idx = take ? oi : idx;                                                 \
  }
  NM_AM_STEP(NM_DPP_QUAD_XOR1) NM_AM_STEP(NM_DPP_QUAD_XOR2) NM_AM_STEP(NM_DPP_HALF_MIRROR) NM_AM_STEP(NM_DPP_ROW_MIRROR)
#undef NM_AM_STEP
  T bv = rdlane(val, 0);
  int bi = rdlane(idx, 0);
#pragma unroll
  for (int r = 16; r < 64; r += 16) {
    T ov = rdlane(val, r);
    int oi = rdlane(idx, r);
    bool take = (ov > bv) | ((ov == bv) & (oi < bi));
    bv = take ? ov : bv;
    bi = take ? oi : bi;
  }
  *best = bv;
  *ibest = bi;
}
// wave maximum and the FIRST lane that holds it (no index carried through the reduction: 4 DPP max + 3 readlane merges + a ballot;
// wargmax costs three times that). For callers to whom any lane of the maximum is as good as another.
template <class T> NM_FN void wmaxfirst(T val, T* best, int* lane_of_best) {
  T m = val;
  m = vmax(m, dpp<NM_DPP_QUAD_XOR1>(m));
  m = vmax(m, dpp<NM_DPP_QUAD_XOR2>(m));
  m = vmax(m, dpp<NM_DPP_HALF_MIRROR>(m));
  m = vmax(m, dpp<NM_DPP_ROW_MIRROR>(m));
  const T b = vmax(vmax(rdlane(m, 0), rdlane(m, 16)), vmax(rdlane(m, 32), rdlane(m, 48)));
  *best = b;
  *lane_of_best = (int)__builtin_ctzll(__ballot(val == b));
}
template <class T> NM_FN T ldsv(const T* a, int i) { return a[i]; }
template <class T> NM_FN void stsv(T* a, int i, T v, bool m) { if (m) a[i] = v; }
// masked LDS store without a branch: lanes outside the mask write their value to their own word of a scratch row instead. A masked
// store compiles to s_and_saveexec / s_cbranch_execz / ds_write / s_or - the end of a scheduling region; this is one v_cndmask on the address.
template <class T, class S> NM_FN void stsu(T* a, int i, T v, bool m, S* sink) {
  T* p = m ? a + i : reinterpret_cast<T*>(sink) + NM_TID;
  *p = v;
}
// Global-memory accessors. The pointers reach the kernel through an LDS copy of the launch arguments / model struct, so
// the compiler no longer knows their address space: the casts keep these global_load / global_store instead of flat_*.
#define NM_GLOBAL(T) __attribute__((address_space(1))) T
typedef float nm_f4 __attribute__((ext_vector_type(4)));
typedef double nm_d4 __attribute__((ext_vector_type(4)));
template <class T> NM_FN T gldv(const T* p, int i) { return ((const NM_GLOBAL(T)*)p)[i]; }
template <class T> NM_FN void gstv(T* p, int i, T v, bool m) { if (m) ((NM_GLOBAL(T)*)p)[i] = v; }
template <class T> NM_FN T gld1(const T* p, size_t i) { return ((const NM_GLOBAL(T)*)p)[i]; }   // wave-uniform or single-lane
template <class T> NM_FN void gst1(T* p, size_t i, T v) { ((NM_GLOBAL(T)*)p)[i] = v; }
// p[i] += v by the hardware's non-returning float atomic (global_atomic_add_f32): nothing to wait for. The caller is the only writer
// of that word during the launch, so the sum is the same as a load / add / store.
NM_FN void gadd1(float* p, size_t i, float v) { __builtin_amdgcn_global_atomic_fadd_f32((NM_GLOBAL(float)*)p + i, v); }
NM_FN void gld3(const float* p, int i, float* o) { const nm_f4 t = *(const NM_GLOBAL(nm_f4)*)((const NM_GLOBAL(float)*)p + i); o[0] = t.x; o[1] = t.y; o[2] = t.z; }
NM_FN void gld3(const double* p, int i, double* o) { const nm_d4 t = *(const NM_GLOBAL(nm_d4)*)((const NM_GLOBAL(double)*)p + i); o[0] = t.x; o[1] = t.y; o[2] = t.z; }
NM_FN void gld4(const float* p, int i, float* o) { const nm_f4 t = *(const NM_GLOBAL(nm_f4)*)((const NM_GLOBAL(float)*)p + i); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
NM_FN void gld4(const double* p, int i, double* o) { const nm_d4 t = *(const NM_GLOBAL(nm_d4)*)((const NM_GLOBAL(double)*)p + i); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
NM_FN int to_int(float x) { return (int)x; }
NM_FN int to_int(double x) { return (int)x; }
// Orders this wave's LDS traffic: lanes exchange data through LDS only inside their own wavefront, whose DS instructions the
// hardware executes in issue order, so all that is needed is that the compiler keeps the program order (no s_barrier, and no
// draining of outstanding global loads/stores as __syncthreads() would do).
NM_FN void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// keep the scheduler from hoisting a later phase's loads across this point (they would sit in VGPRs and spill)
NM_FN void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
}  // namespace simt

#else
// =====================================================================================  HOST EMULATION (tests only)
#include <cmath>
#include <cstring>
#define NM_FN inline
#define NM_COLD inline
namespace simt {
template <class T> struct V {
  T v[NM_WAVE];
  V() {}
  V(T s) { for (int i = 0; i < NM_WAVE; i++) v[i] = s; }
#define NM_BIN(op)                                                                                   \
  friend V operator op(const V& a, const V& b) { V r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
  NM_BIN(+) NM_BIN(-) NM_BIN(*) NM_BIN(/) NM_BIN(%) NM_BIN(&) NM_BIN(|) NM_BIN(>>) NM_BIN(<<)
#undef NM_BIN
#define NM_CMP(op)                                                                                      \
  friend V<bool> operator op(const V& a, const V& b) { V<bool> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
  NM_CMP(<) NM_CMP(>) NM_CMP(<=) NM_CMP(>=) NM_CMP(==) NM_CMP(!=)
#undef NM_CMP
  friend V operator-(const V& a) { V r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = -a.v[i]; return r; }
  friend V operator!(const V& a) { V r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = !a.v[i]; return r; }
  V& operator+=(const V& b) { for (int i = 0; i < NM_WAVE; i++) v[i] += b.v[i]; return *this; }
  V& operator-=(const V& b) { for (int i = 0; i < NM_WAVE; i++) v[i] -= b.v[i]; return *this; }
  V& operator*=(const V& b) { for (int i = 0; i < NM_WAVE; i++) v[i] *= b.v[i]; return *this; }
};
// float % is not defined: only instantiate % for ints (templates instantiate lazily per use, fine)
using VB = V<bool>;

NM_FN V<int> lane_id() { V<int> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = i; return r; }
#define NM_SEL(T)                                                                                              \
  NM_FN V<T> sel(const VB& c, const V<T>& a, const V<T>& b) { V<T> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; } \
  NM_FN T sel(bool c, T a, T b) { return c ? a : b; }
NM_SEL(float) NM_SEL(double) NM_SEL(int)
#undef NM_SEL
#define NM_MAP1(name, expr)                                                                                   \
  NM_FN V<float> name(const V<float>& a) { V<float> r; for (int i = 0; i < NM_WAVE; i++) { float x = a.v[i]; r.v[i] = expr; } return r; } \
  NM_FN V<double> name(const V<double>& a) { V<double> r; for (int i = 0; i < NM_WAVE; i++) { double x = a.v[i]; r.v[i] = expr; } return r; } \
  NM_FN float name(float x) { return expr; }                                                                  \
  NM_FN double name(double x) { return expr; }
NM_MAP1(vsqrt, std::sqrt(x)) NM_MAP1(vabs, std::fabs(x)) NM_MAP1(vexp, std::exp(x)) NM_MAP1(vacos, std::acos(x))
#undef NM_MAP1
#define NM_MAP2(name, expr, T)                                                                               \
  NM_FN V<T> name(const V<T>& a, const V<T>& b) { V<T> r; for (int i = 0; i < NM_WAVE; i++) { T x = a.v[i], y = b.v[i]; r.v[i] = expr; } return r; } \
  NM_FN T name(T x, T y) { return expr; }
NM_MAP2(vmax, (x > y ? x : y), float) NM_MAP2(vmax, (x > y ? x : y), double) NM_MAP2(vmax, (x > y ? x : y), int)
NM_MAP2(vmin, (x < y ? x : y), float) NM_MAP2(vmin, (x < y ? x : y), double) NM_MAP2(vmin, (x < y ? x : y), int)
NM_MAP2(vpow, std::pow(x, y), float) NM_MAP2(vpow, std::pow(x, y), double)
#undef NM_MAP2
template <class T> NM_FN void vsincos(const V<T>& x, V<T>* s, V<T>* c) { for (int i = 0; i < NM_WAVE; i++) { s->v[i] = std::sin(x.v[i]); c->v[i] = std::cos(x.v[i]); } }
NM_FN void vsincos(float x, float* s, float* c) { *s = std::sin(x); *c = std::cos(x); }
NM_FN void vsincos(double x, double* s, double* c) { *s = std::sin(x); *c = std::cos(x); }
template <class T> NM_FN VB visbad(const V<T>& a) { VB r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = !(std::fabs(a.v[i]) <= (T)1e10); return r; }
NM_FN bool visbad(float x) { return !(std::fabs(x) <= 1e10f); }
NM_FN bool visbad(double x) { return !(std::fabs(x) <= 1e10); }
template <class T> NM_FN V<T> to_real(const V<int>& a) { V<T> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = (T)a.v[i]; return r; }
template <class T> NM_FN T to_real(int a) { return (T)a; }
template <class T> NM_FN V<T> vrcp(const V<T>& a) { V<T> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = (T)1 / a.v[i]; return r; }
NM_FN float vrcp(float x) { return 1.0f / x; }
NM_FN double vrcp(double x) { return 1.0 / x; }
template <class T, class S> NM_FN V<T> vcvt(const V<S>& a) { V<T> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = (T)a.v[i]; return r; }

template <class T> NM_FN V<T> vfma(const V<T>& a, const V<T>& b, const V<T>& c) { return a * b + c; }      // (the host build is compiled with -ffp-contract=off: two roundings, everywhere)
NM_FN float vfma(float a, float b, float c) { return a * b + c; }
NM_FN double vfma(double a, double b, double c) { return a * b + c; }
template <class T> NM_FN V<T> sel_mask(uint64_t m, const V<T>& a, const V<T>& b) {
  V<T> r;
  for (int i = 0; i < NM_WAVE; i++) r.v[i] = ((m >> i) & 1ull) ? a.v[i] : b.v[i];
  return r;
}
NM_FN uint64_t mask_shl(uint64_t m, int k) { return m << k; }
template <int I, class T> NM_FN V<T> fma_half_lane(const V<T>& a, const V<T>& x, const V<T>& g, const VB&) {
  V<T> r;
  for (int i = 0; i < NM_WAVE; i++) r.v[i] = a.v[i] * x.v[(i & 32) + I] + g.v[i];
  return r;
}
template <class T> NM_FN T rdlane(const V<T>& x, int l) { return x.v[l]; }
template <class T> NM_FN T rdlane(T x, int) { return x; }
template <class T> NM_FN V<T> wrlane(V<T> x, T v, int l) { x.v[l] = v; return x; }
NM_FN V<int> opaque_lane() { return lane_id(); }
template <class T> NM_FN T uniform(const V<T>& x) { return x.v[0]; }
template <class T> NM_FN T uniform(T x) { return x; }
template <class T> NM_FN V<T> shfl_xor(const V<T>& x, int m) { V<T> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = x.v[i ^ m]; return r; }
template <int P, class T> NM_FN V<T> quad(const V<T>& x) { V<T> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = x.v[(i & ~3) | ((P >> (2 * (i & 3))) & 3)]; return r; }
template <class T> NM_FN V<T> shfl_xor1(const V<T>& x) { return shfl_xor(x, 1); }
template <class T> NM_FN V<T> shfl_xor2(const V<T>& x) { return shfl_xor(x, 2); }
template <class T> NM_FN V<T> half_mirror(const V<T>& x) { V<T> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = x.v[(i & ~7) | (7 - (i & 7))]; return r; }
template <class T> NM_FN V<T> row_mirror(const V<T>& x) { V<T> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = x.v[(i & ~15) | (15 - (i & 15))]; return r; }
template <class T> NM_FN V<T> gsum8(V<T> x) {
  x = x + shfl_xor(x, 1); x = x + shfl_xor(x, 2); x = x + half_mirror(x);
  return x;
}
template <class T> NM_FN T wsum8(V<T> x) {
  x = x + shfl_xor(x, 1); x = x + shfl_xor(x, 2); x = x + half_mirror(x);
  return x.v[0];
}
template <class T> NM_FN T wsum(V<T> x) {
  x = x + shfl_xor(x, 1); x = x + shfl_xor(x, 2); x = x + half_mirror(x); x = x + row_mirror(x);
  return (x.v[0] + x.v[16]) + (x.v[32] + x.v[48]);
}
template <class T> NM_FN V<T> hsum32(V<T> x) {
  x = x + shfl_xor(x, 1); x = x + shfl_xor(x, 2); x = x + half_mirror(x); x = x + row_mirror(x);
  const T s0 = x.v[0] + x.v[16], s1 = x.v[32] + x.v[48];
  V<T> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = i < 32 ? s0 : s1;
  return r;
}
NM_FN bool wany(const VB& c) { for (int i = 0; i < NM_WAVE; i++) if (c.v[i]) return true; return false; }
NM_FN V<int> lane_rank(uint64_t m) { V<int> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = __builtin_popcountll(m & ((1ull << i) - 1ull)); return r; }
NM_FN int popc64(uint64_t m) { return __builtin_popcountll(m); }
NM_FN uint64_t ballot(const VB& c) { uint64_t m = 0; for (int i = 0; i < NM_WAVE; i++) if (c.v[i]) m |= 1ull << i; return m; }
NM_FN VB in_mask(uint64_t m) { VB r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = (m >> i) & 1; return r; }
template <class T> NM_FN void order_after(V<int>&, const V<T>&) {}
template <class T> NM_FN void wargmax(V<T> val, V<int> idx, T* best, int* ibest) {
  T bv = val.v[0]; int bi = idx.v[0];
  for (int i = 1; i < NM_WAVE; i++)
    if (val.v[i] > bv || (val.v[i] == bv && idx.v[i] < bi)) { bv = val.v[i]; bi = idx.v[i]; }
  *best = bv; *ibest = bi;
}
template <class T> NM_FN void wmaxfirst(const V<T>& val, T* best, int* lane_of_best) {
  T bv = val.v[0]; int bi = 0;
  for (int i = 1; i < NM_WAVE; i++) if (val.v[i] > bv) { bv = val.v[i]; bi = i; }
  *best = bv; *lane_of_best = bi;
}
template <class T> NM_FN V<T> ldsv(const T* a, const V<int>& i) { V<T> r; for (int k = 0; k < NM_WAVE; k++) r.v[k] = a[i.v[k]]; return r; }
template <class T> NM_FN T ldsv(const T* a, int i) { return a[i]; }
template <class T> NM_FN void stsv(T* a, const V<int>& i, const V<T>& v, const VB& m) { for (int k = 0; k < NM_WAVE; k++) if (m.v[k]) a[i.v[k]] = v.v[k]; }
template <class T> NM_FN void stsv(T* a, const V<int>& i, T v, const VB& m) { for (int k = 0; k < NM_WAVE; k++) if (m.v[k]) a[i.v[k]] = v; }
template <class T, class S> NM_FN void stsu(T* a, const V<int>& i, const V<T>& v, const VB& m, S*) { stsv(a, i, v, m); }
template <class T, class S> NM_FN void stsu(T* a, const V<int>& i, T v, const VB& m, S*) { stsv(a, i, v, m); }
template <class T> NM_FN T gld1(const T* p, size_t i) { return p[i]; }
template <class T> NM_FN void gst1(T* p, size_t i, T v) { p[i] = v; }
NM_FN void gadd1(float* p, size_t i, float v) { p[i] += v; }
template <class T> NM_FN V<T> gldv(const T* p, const V<int>& i) { return ldsv(p, i); }
template <class T> NM_FN void gstv(T* p, const V<int>& i, const V<T>& v, const VB& m) { stsv(p, i, v, m); }
template <class T> NM_FN void gstv(T* p, const V<int>& i, T v, const VB& m) { stsv(p, i, v, m); }
template <class T> NM_FN void gld4(const T* p, const V<int>& i, V<T>* o) { for (int k = 0; k < NM_WAVE; k++) for (int c = 0; c < 4; c++) o[c].v[k] = p[i.v[k] + c]; }
template <class T> NM_FN V<int> to_int(const V<T>& a) { V<int> r; for (int i = 0; i < NM_WAVE; i++) r.v[i] = (int)a.v[i]; return r; }
template <class T> NM_FN void gld3(const T* p, const V<int>& i, V<T>* o) { for (int k = 0; k < NM_WAVE; k++) for (int c = 0; c < 3; c++) o[c].v[k] = p[i.v[k] + c]; }
template <class T> NM_FN void gld3(const T* p, int i, T* o) { o[0] = p[i]; o[1] = p[i + 1]; o[2] = p[i + 2]; }
NM_FN void wave_sync() {}
NM_FN void sched_fence() {}
}  // namespace simt
#endif
