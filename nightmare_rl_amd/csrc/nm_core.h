// nm_core.h - NightmareV3Env.step() for ONE env on ONE 64-lane wavefront (CDNA4), written against simt.h.
//
// Path restated (reference file:line): envs/nightmare_v3_env.py:145-311 (step), :321-371 (command resample,
// reset_idx) and, below it, what MuJoCo 3.1.2's mj_step executes for models/nightmare_v3/mjmodel.xml
// (SURVEY.md section 8a rows E1-E8, P1-P10).  Algorithmically equivalent to oracle/ but laid out for the
// hardware instead of following MuJoCo's data structures:
//
//   stage A  "lane = leg"  (6 lanes + redundant uniform base math): kinematics, spatial inertias about the
//            BASE ORIGIN (world axes), composite inertias, joint-space inertia as blocks
//            [Mbb 6x6 | Mlb_l 3x6 | M_l 3x3] (legs only couple through the base), block factorisation
//            (L'DL of M_l, W_l = M_l^-1 Mlb_l, Schur complement LDL'), RNE bias, servo forces, qacc_smooth.
//            Done for M and for M + h*kv*I (implicitfast) in one pass.
//   stage B  "lane = hull vertex": plane-vs-convex-hull support search for the 7 colliding meshes
//            (wave arg-max), up to 3 extra penetrating hull neighbours chosen by ballot.
//   stage C  "lane = constraint row" (4 pyramid rows per contact, <= 64 rows): sparse Jacobian row (6 base +
//            3 own-leg entries), B_i = M^-1 J_i' by the block factor, A = J M^-1 J' held as one ROW PER LANE IN
//            REGISTERS, warm start, PGS x3 and NoSlip x4 as statically unrolled sweeps that broadcast the
//            updated row's delta with v_readlane and rank-1 update every lane's residual.
//   stage D  "lane = leg": implicitfast solve, semi-implicit Euler, quaternion integration.
//   epilogue "lane = observation slot": frame transforms, termination, reset, rewards, obs.
//
// All inter-stage traffic goes through the wave's LDS (18.6 KB per wave of two envs: two env images, the row buffer, the wave's copy
// of the model constants and of the launch arguments); HBM is touched once per env-step (state in, state + obs out).
#pragma once
#include <type_traits>
#include <stddef.h>

#include "simt.h"

// Stage-skipping switches exist only in measurement builds (make measure: -DNM_MEASURE, libnightmare_hip_measure.so); in the shipped
// library the mask is the constant 0 and every branch on it is compiled away.
#ifdef NM_MEASURE
#define NM_ABLATE(mask) (mask)
#else
#define NM_ABLATE(mask) 0
#endif
namespace nm {
using namespace simt;

// ----------------------------------------------------------------------------------------- layout constants
constexpr int kNQ = 25, kNV = 24, kNU = 18, kNLEG = 6, kNCOL = 7, kNSENS = 13, kNOBS = 66, kNREW = 16;
constexpr int kLinkN = 25;             // per-link constants: bpos3 bR9 axis3 ipos3 Ibody6 mass1
constexpr int kLegN = 3 * kLinkN;      // per-leg constants
constexpr int kBaseN = 10;             // ipos3 Ibody6 mass1
constexpr int kColN = 24;              // per colliding mesh: center3 rbound invweight0 nvert vadr pad | obb: center3 axes9 half3 pad
constexpr int kMaxCon = 16;            // contacts the register-resident solver takes (rows = 4*kMaxCon = one per lane)
constexpr int kMaxRow = 4 * kMaxCon;
constexpr int kMaxConBig = 44;         // contact slots per env; the model can produce 7 meshes x 4 floor contacts + 15 tibia pairs = 43
constexpr int kBigSlots = (4 * kMaxConBig + NM_WAVE - 1) / NM_WAVE;   // rows per lane of the matrix-free solver used beyond kMaxCon
constexpr int kJRow = 16;              // LDS row: Jb6 Jl3 leg | Jm3 leg1 (body1 side of a tibia-tibia contact) pad2
// every reward name of the reference config that has a _reward_ function (env.py:399-497), in the order class_to_dict yields them
// (dir() = alphabetical, helpers.py:7); termination last because step() adds it last (env.py:285-288)
enum { R_ACTION_RATE, R_ANG_VEL_XY, R_BASE_HEIGHT, R_BODY_CONTACT, R_DEFAULT_POS, R_DOF_ACC, R_DOF_VEL, R_FEET_AIR_TIME, R_FEET_CONTACT,
       R_LIN_VEL_Z, R_ORIENTATION, R_STAND_STILL, R_TORQUES, R_TRACK_ANG, R_TRACK_LIN, R_TERMINATION };

// model + config constants, converted once to `real` by the host. The device kernel copies the struct (3 KB in fp32) into LDS
// when a wave starts, so every M.x below is an LDS broadcast read, not a scalar load through a pointer.
template <class real> struct Model {
  const real* hullv;    // [nhull][4] xyz0, body frame (HBM/L2: 27 KB)
  const real* hullnv;   // [nhull][maxnbr+1][4]: per vertex, its hull neighbours as (x, y, z, local id or -1) and, last, itself:
                        // one coalesced gather gives a vertex's whole ring (HBM/L2: 0.95 MB)
  real legc[kNLEG * kLegN];    // the small tables travel inside the struct: the kernel keeps one copy per wave in LDS
  real basec[kBaseN];
  real colc[kNCOL * kColN];
  real footc[kNLEG * 4];       // foot site pos (tibia frame) + radius
  real qpos0[kNQ];
  int maxnbr;
  int link_rot_id[3];          // link k of every leg has body_quat = 1: its frame is its parent's, rotated by the joint only
  real total_mass;
  real h, kv, ctrl_max, grav, mu;
  real solref_K, solref_B, si_d0, si_dmax, si_width, si_mid, si_power;
  real pgs_scale, pgs_tol, noslip_tol, tol_planemesh;
  int pgs_iters, noslip_iters, mpr_iters;
  real mpr_tol;
  // env config (reference envs/nightmare_v3_config.py)
  real dt, p_gain, clip_obs, obs_lin, obs_ang, obs_dofpos, obs_dofvel, max_lin_x, max_ang, term_force, sigma, max_ep_len, default_pos[3];
  float action_scale, clip_actions;
  int resample_every;
  real rew_scale[kNREW];   // config scale x dt (env.py:123-128); 0 = not in the reward table, its function never runs
  real ep_len_s;
  // terms the reference config ships with scale 0 (config.py:88-95) and contact modes other than 1 (config.py:17-21)
  int rew_extra;           // any of ang_vel_xy base_height dof_vel feet_air_time feet_contact_forces lin_vel_z stand_still is in the table
  int tibia_mode, body_mode;
  int collide_batch_min;   // floor contacts of an env go out in one batched pass from this many touching meshes on (stage_collide); 3
  real tibia_max, body_max, base_h_target, max_contact_force;
};

// per-launch arguments (device pointers, AoS-by-env rows so one wave reads contiguous bytes)
template <class real> struct Args {
  int N;
  uint64_t seed;
  int64_t env_offset;
  // physics state
  real *qpos, *qvel, *qwarm;
  // env buffers the reference keeps between steps
  real *dofpos, *dofvel, *act, *cmd, *epsum;
  real* feetair;         // [N,6] feet_air_time (env.py:90); touched only while feet_air_time is in the reward table
  int* feetflags;        // [N] bits 0..5 last_contacts, bits 6..11 last_contacts_filt (env.py:92-93)
  int64_t* eplen;
  uint32_t* rngctr;
  int* hullcache;        // [N,8] warm start of the support-vertex search (any vertex index in range is valid); bits 16.. of entry 0: the
                         // env's contact count at the end of the previous step (issue-priority hint, nm_set_priority), entry 7: fallback count
  // inputs
  const float* actions;  // [N,18]
  const real* cmd_u;     // [N,4] or null
  // outputs
  float *obs, *rew, *timeout_now;
  float* ret_acc;        // optional [N]: ret_acc[env] += reward of this step (the caller's running return), null = off
  int64_t* done;
  real* stat_sum;        // [kNREW] sums of episode sums over envs that reset this step
  int* stat_cnt;         // [4]: #resets this step, #contacts dropped (contact cap), #bad-state resets (mj_check*), #hull-search fallbacks
  // end of step, done by whichever wave finishes last (device build): extras refreshed only when >= 1 env reset (env.py:344-371)
  int* wave_done;        // tickets of the waves that have published their results in this launch: [g*32] per group of 64 waves,
                         // [kTicketTop] over the groups (one counter for 2048 waves serialises ~25 us of same-address atomics)
  int *nto, *to_list;    // [1], [N]: envs that timed out in this launch (what the closing wave turns into extras['time_outs'])
  int *nprev, *to_prev;  // [1], [N]: the envs whose time_outs entry the last refresh set to 1 (cleared one by one at the next refresh)
  unsigned long long* to_owner;   // [1] address of the time_outs buffer the last refresh wrote (another buffer is rewritten in full)
  float *ep_stats, *time_outs;   // [kNREW] extras['episode'], [N] extras['time_outs']
  long long* counters;   // [3] running totals of stat_cnt[1..3]
  real* dbg;             // optional [N][kDbgN]
  int nsub;              // decimation
  int nxcd;              // XCDs the dispatcher deals workgroups over (hipDeviceAttributeNumberOfXccs); the block -> wave remap is used only when it is 8
  int physics_only;      // 1: skip env epilogue (BASELINE config 2: dynamics+contact only)
  int ablate;            // read only by -DNM_MEASURE builds (the shipped library has no way to set it and compiles the tests away): bit0 no collision, bit1 no solver sweeps, bit2 no constraint stage, bit3 no smooth stage, bit4 no tibia pairs,
                         // bit5 constraint stage one env at a time, bit7 no env epilogue (E3-E8), bit8 no observation, bit9 empty launch, bit10 load only, bit11 no substeps
  // observation noise (env.py:109-119,304-305): null = off
  const real* noise_vec; // [66] noise_scale_vec
  const real* noise_u;   // [N,66] injected uniforms (parity tests) or null = counter RNG
  uint64_t noise_step;   // step index of this launch (RNG counter = noise_step*66 + k)
  // state log of one env (env.py:261-272): post-physics, pre-reset qpos[25] qvel[24] + bad-state-reset count; null = off
  real* rec;
  int rec_env;
};
constexpr int kTicketGroup = 64, kTicketStride = 32, kTicketTop = 0;   // group g's counter at [(g + 1) * kTicketStride] (own 128 B line)
constexpr uint64_t kNoiseKey = 0x4E4F495345ull;
constexpr int kDbgN = 256;

// ----------------------------------------------------------------------------------------- LDS image of one env
template <class real> struct Sh {
  real qpos[28], qvel[24], warm[24], ctrl[20];
  real Rb[9], wv[6];             // base rotation; base spatial velocity [w_world; v_origin]
  real anc[kNU * 3], axs[kNU * 3];  // hinge anchors (relative to base origin) and axes, world-aligned
  real colR[kNCOL * 9], colp[kNCOL * 3];
  real Minv[kNLEG * 6], W[kNLEG * 18], Lb[15], Dbi[6];      // factor of M
  real MinvH[kNLEG * 6], WH[kNLEG * 18], LbH[15], DbiH[6];  // factor of M + h kv I
  real qas[24], qfs[24], qfc[24];
  real vv[24];                    // M^-1 J' f of the matrix-free solver (envs with more than kMaxCon contacts)
  real sens[16], cvb[6];
  real bh[2];                     // xipos[1][2] of the last forward pass (env.py:223)
  real mbb[36], sc[36];           // base block of M and its Schur complement (upper triangles)
  // per-leg staging between the forward and backward chain passes of stage A: 3 x (S6 I10 f6). Stages B and C reuse it for the
  // contact list (stage A of an env is over before its collision starts): kMaxConBig x (pos3 normal3 dist, leg of body2 (-1 = base),
  // leg of body1 (-1 = world/floor))
  real legtmp[kNLEG * 66];
  NM_FN real* cpos() { return legtmp; }
  NM_FN real* cnrm() { return legtmp + 3 * kMaxConBig; }
  NM_FN real* cdist() { return legtmp + 6 * kMaxConBig; }
  NM_FN int* cleg() { return reinterpret_cast<int*>(legtmp + 7 * kMaxConBig); }
  NM_FN int* cleg1() { return reinterpret_cast<int*>(legtmp + 8 * kMaxConBig); }
  real efc_f[kMaxRow];
#ifdef NM_DEBUG_SOLVER
  real dbg_b[kMaxRow], dbg_a[kMaxRow], dbg_f0[kMaxRow];
#endif
  int ncon, nwarn, it_pgs, it_noslip, anypair;
  int nhop;
  int ntog;                       // debug buffer only. [env 0 of a wave] substeps of this step whose constraint stage took both envs in one pass;
                                  // [env 1] contact counts of the first substep, n0 * 64 + n1
  real eact[kNU], epact[kNU], epdv[kNU];  // this step's clipped actions, last step's actions and joint velocities (epilogue inputs)
  real ecmd[4], eepsum[kNREW];    // env buffers fetched at load time for the epilogue: commands, episode sums
  int eplen_lo, eplen_hi;         // episode_length_buf[env] (int64) as it was before this step
  unsigned ectr;                  // command RNG counter
  int cstart[8], ccnt[8];         // contacts of colliding mesh g (0 = base, 1..6 = tibias): first index and count (floor contacts)
  real sink[NM_WAVE];             // scratch row for branch-free masked stores (stsu): one word per lane, never read
  int hcache[8];                  // [0..6] support vertex of each colliding mesh found last time (warm start of the hull search);
                                  // [7] running count of this env's exhaustive-scan fallbacks (a tuning diagnostic that travels with the
                                  // row: 2048 waves adding to ONE counter serialise at ~12 ns each and hold the kernel's end back)
};

static_assert(kNLEG * 66 >= 9 * kMaxConBig, "contact list must fit the leg staging area");
// LDS of one wavefront: G env images (the leg-lane stages run all G envs at once, lanes 8g..8g+5 = legs of env g; collision,
// constraints and the env epilogue take the envs one after the other on all 64 lanes) + one shared row buffer.
template <class real, int G> struct ShW {
  Sh<real> e[G];
  alignas(16) real jrow[kMaxRow * kJRow];
};
#define NM_OFS(field) ((int)(offsetof(Sh<real>, field) / sizeof(real)))
#define NM_IOFS(field) ((int)(offsetof(Sh<real>, field) / sizeof(int)))

// ----------------------------------------------------------------------------------------- small algebra
template <class A, class B, class C> NM_FN void cross3(A* r, const B* a, const C* b) {
  A x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
template <class A, class B, class C> NM_FN A dot3(const B* a, const C* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class A, class B, class C> NM_FN void matvec3(A* r, const B* m, const C* v) {
  A x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
// the same with the order of the roundings written down (fma(m2, v2, fma(m1, v1, m0 v0))): where two code paths must agree bit for bit on
// the device, the compiler must not be the one who picks which product is rounded first
template <class A, class B, class C> NM_FN void matvec3_fma(A* r, const B* m, const C* v) {
  A x = vfma(A(m[2]), A(v[2]), vfma(A(m[1]), A(v[1]), A(m[0]) * A(v[0])));
  A y = vfma(A(m[5]), A(v[2]), vfma(A(m[4]), A(v[1]), A(m[3]) * A(v[0])));
  A z = vfma(A(m[8]), A(v[2]), vfma(A(m[7]), A(v[1]), A(m[6]) * A(v[0])));
  r[0] = x; r[1] = y; r[2] = z;
}
template <class A, class B, class C> NM_FN void matmul3(A* r, const B* a, const C* b) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
// spatial inertia (Ixx Iyy Izz Ixy Ixz Iyz, m*d(3), m) times motion vector [ang; lin]
template <class A, class B, class C> NM_FN void inert_mul(A* r, const B* i, const C* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
template <class A, class B, class C> NM_FN void cross_motion(A* r, const B* vel, const C* v) {
  A a[3], b[3], c[3];
  cross3(a, vel, v); cross3(b, vel, v + 3); cross3(c, vel + 3, v);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
template <class A, class B, class C> NM_FN void cross_force(A* r, const B* vel, const C* f) {
  A a[3], b[3], c[3];
  cross3(a, vel, f); cross3(b, vel + 3, f + 3); cross3(c, vel, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
template <class A, class B, class C> NM_FN A dot6(const B* a, const C* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
// sum over the six leg lanes (lanes 0..5) -> wave-uniform
template <class real> NM_FN real lanesum6(const V<real>& x) {
  return ((rdlane(x, 0) + rdlane(x, 1)) + (rdlane(x, 2) + rdlane(x, 3))) + (rdlane(x, 4) + rdlane(x, 5));
}
// spatial inertia of a body about the reference point: body-frame tensor Ib(6), rotation R, COM offset d, mass m
template <class X, class real> NM_FN void spatial_inertia(X* I10, const X* R, const X* Ib, const X* d, X m, real) {
  X T[9];  // R * Ib
#pragma unroll
  for (int i = 0; i < 3; i++) {
    T[3 * i + 0] = R[3 * i] * Ib[0] + R[3 * i + 1] * Ib[3] + R[3 * i + 2] * Ib[4];
    T[3 * i + 1] = R[3 * i] * Ib[3] + R[3 * i + 1] * Ib[1] + R[3 * i + 2] * Ib[5];
    T[3 * i + 2] = R[3 * i] * Ib[4] + R[3 * i + 1] * Ib[5] + R[3 * i + 2] * Ib[2];
  }
  X xx = T[0] * R[0] + T[1] * R[1] + T[2] * R[2], yy = T[3] * R[3] + T[4] * R[4] + T[5] * R[5], zz = T[6] * R[6] + T[7] * R[7] + T[8] * R[8];
  X xy = T[0] * R[3] + T[1] * R[4] + T[2] * R[5], xz = T[0] * R[6] + T[1] * R[7] + T[2] * R[8], yz = T[3] * R[6] + T[4] * R[7] + T[5] * R[8];
  I10[0] = xx + m * (d[1] * d[1] + d[2] * d[2]);
  I10[1] = yy + m * (d[0] * d[0] + d[2] * d[2]);
  I10[2] = zz + m * (d[0] * d[0] + d[1] * d[1]);
  I10[3] = xy - m * d[0] * d[1];
  I10[4] = xz - m * d[0] * d[2];
  I10[5] = yz - m * d[1] * d[2];
  I10[6] = m * d[0]; I10[7] = m * d[1]; I10[8] = m * d[2];
  I10[9] = m;
}
// L'DL of a symmetric 3x3 leg block (entries 00 01 02 11 12 22; 0 = coxa ... 2 = tibia), eliminating the
// leaf (tibia) first as mj_factorM does: backward stable on these graded blocks (the cofactor inverse is not
// and costs fp32 three digits). Factor = (l21 l20 l10 1/d2 1/d1 1/d0).
template <class X, class real> NM_FN void ldl3(X* r, const X* a, real one) {
  X i2 = vrcp(a[5]);
  X l21 = a[4] * i2, l20 = a[2] * i2;
  X m11 = a[3] - l21 * a[4], m01 = a[1] - l21 * a[2], m00 = a[0] - l20 * a[2];
  X i1 = vrcp(m11);
  X l10 = m01 * i1;
  X d0 = m00 - l10 * m01;
  r[0] = l21; r[1] = l20; r[2] = l10; r[3] = i2; r[4] = i1; r[5] = vrcp(d0);
}
template <class X, class Y> NM_FN void ldl3_solve(X* x, const X* f, const Y* y) {
  X y2 = y[2];
  X y1 = y[1] - f[0] * y2;
  X y0 = y[0] - f[1] * y2 - f[2] * y1;
  X x0 = y0 * f[5];
  X x1 = y1 * f[4] - f[2] * x0;
  x[2] = y2 * f[3] - f[0] * x1 - f[1] * x0;
  x[1] = x1; x[0] = x0;
}
// LDL' of a symmetric 6x6 given by its upper triangle through the accessor S(row, col), row <= col:
// L strictly lower (15, row-major packed), Dinv(6). X = real (wave-uniform) or V<real> (one matrix per lane group).
template <class X, class real, class Acc> NM_FN void ldl6(Acc S, X* L, X* Dinv, real one) {
  X Lf[36], D[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    X d = S(j, j);
#pragma unroll
    for (int k = 0; k < j; k++) d = d - Lf[6 * j + k] * Lf[6 * j + k] * D[k];
    D[j] = d;
    X di = vrcp(d);
    Dinv[j] = di;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      X s = S(j, i);
#pragma unroll
      for (int k = 0; k < j; k++) s = s - Lf[6 * i + k] * Lf[6 * j + k] * D[k];
      Lf[6 * i + j] = s * di;
    }
  }
  int n = 0;
#pragma unroll
  for (int i = 1; i < 6; i++)
#pragma unroll
    for (int j = 0; j < i; j++) L[n++] = Lf[6 * i + j];
}
// x <- (L D L')^-1 x
template <class LT, class X> NM_FN void ldl6_solve(const LT* L, const LT* Dinv, X* x) {
  int n = 0;
#pragma unroll
  for (int i = 1; i < 6; i++)
#pragma unroll
    for (int j = 0; j < i; j++) x[i] = x[i] - L[n++] * x[j];
#pragma unroll
  for (int i = 0; i < 6; i++) x[i] = x[i] * Dinv[i];
#pragma unroll
  for (int i = 5; i >= 1; i--) {
    n = i * (i - 1) / 2;
#pragma unroll
    for (int j = 0; j < i; j++) x[j] = x[j] - L[n + j] * x[i];
  }
}

// counter-based uniform in [0,1) with 24 random bits (same definition as oracle/nm_oracle_env.c nmo_rand_u24)
NM_FN uint32_t rand_u24_bits(uint64_t seed, uint64_t genv, uint32_t ctr) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * (genv + 1) + 0xD1B54A32D192ED03ull * (uint64_t)ctr;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 40);
}

// Cross-wave results (episode statistics, counters, the time-out list) travel through device-scope atomics only. A wave consumes
// the value every such atomic returns before it takes its end-of-step ticket (k_env_step), so they have been performed at the
// coherence point by then - no release fence, i.e. no L2 write-back per wave (measured: a __threadfence() per wave cost 40 us of 101).
#ifndef NM_EMUL
template <class T> NM_FN void nm_consume(T x) { asm volatile("" ::"v"(x) : "memory"); }
#endif

// Optional in-kernel stage timing (build with -DNM_STAMPS; measurement builds only, never the shipped library):
// lane 0 of every wave adds the s_memtime ticks since its previous stamp to g_stamps[k].
#if defined(NM_STAMPS) && !defined(NM_EMUL)
__device__ unsigned long long g_stamps[16];
NM_FN void nm_stamp(int k) {   // k = -1 starts the clock, k = 10 is the last stamp of a wave and flushes its sums
  __shared__ unsigned long long tl, acc[16];
#ifdef NM_STAMPS_B
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // exact attribution: outstanding loads belong to the segment that issued them
#endif
  unsigned long long n = __builtin_amdgcn_s_memtime();
  if (NM_TID == 0) {
    if (k < 0) for (int i = 0; i < 16; i++) acc[i] = 0;
    else acc[k] += n - tl;
    if (k == 10) for (int i = 0; i < 16; i++) atomicAdd(&g_stamps[i], acc[i]);
    tl = n;
  }
}
#else
NM_FN void nm_stamp(int) {}
#endif
// -DNM_STAMPS_B: slots 11..13 time the sub-phases of the floor collision instead of the epilogue segments
#ifdef NM_STAMPS_B
#define NM_ESTAMP(k)
#define NM_BSTAMP(k) nm_stamp(k)
#else
#define NM_ESTAMP(k) nm_stamp(k)
#define NM_BSTAMP(k)
#endif

// =========================================================================================  stage A
// Everything "smooth": kinematics, inertia blocks + both factorisations, bias, servo forces, qacc_smooth - for all G
// envs of the wave at once, on eight lane groups of 8 (lane = leg, 6 of 8 used): group g works on env g % G in role g / G. What is
// "per env" (base frame, Schur complement ...) is computed redundantly by the 8 lanes of every group of the env, so no cross-lane
// broadcast is needed and sums over legs are three DPP adds (gsum8). The roles split what would otherwise be repeated instruction
// sequences on the same lanes: one body's spatial inertia and force each (links 0..2, base), one of the two factorisations each.
// The forward pass parks per-link results in the leg's LDS slots; the backward pass picks them up again (keeps the live
// register set small).
constexpr int kLinkTmp = 22;  // per link in LDS: S(6) I10(10) f(6)

template <class real> NM_FN V<real> legsum(const V<real>& x, const VB& isleg) {  // sum over the legs of the lane's env
  return gsum8(sel(isleg, x, V<real>(real(0))));
}

template <class real, int G> NM_FN void stage_smooth(ShW<real, G>& w, const Model<real>& M, bool last) {
  typedef V<real> vr;
  real* lds = reinterpret_cast<real*>(&w.e[0]);
  // Lane groups of 8: group g works on env g % G; groups [0, G) factor M, groups [G, 2G) factor M + h*kv*I (implicitfast) - the two
  // factorisations are the same instructions on different lanes, not two passes. Everything before the factorisation does not depend
  // on the pass, so the groups of an env hold exact duplicates there; groups beyond 2G repeat the first 2G.
  static_assert((G & (G - 1)) == 0 && 2 * G <= 8, "lane groups: G envs x 2 factorisations");
  const V<int> lane0 = opaque_lane();
  const V<int> sub = lane0 & 7, grp = lane0 >> 3;
  const V<int> leg = vmin(sub, V<int>(5)), eo = (grp & (G - 1)) * (int)(sizeof(Sh<real>) / sizeof(real));
  const VB pass1 = (grp & G) != 0;
  const VB isleg = sub < V<int>(6);
#define LDG(field, i) ldsv(lds, eo + (NM_OFS(field) + (i)))
  // Stores of this stage are NOT masked: the lanes outside the legs / the first lane of a group hold exact duplicates (sub 6, 7 compute
  // leg 5 again, per-env values are computed by all eight lanes of every group of the env), so they write the same value to
  // the same address. A masked store is an exec-mask branch: 70 of them cut this stage into ~45-instruction scheduling regions.
  const VB st_all = VB(true), st_leg = VB(true);
#define STG(field, i, val) stsv(lds, eo + (NM_OFS(field) + (i)), val, st_all)
#define LDL(field, idx) ldsv(lds, eo + (idx) + NM_OFS(field))
#define STL(field, idx, val) stsv(lds, eo + (idx) + NM_OFS(field), val, st_leg)
  const V<int> slot = leg * (3 * kLinkTmp);

  // ---- base frame (per env)
  vr Rb[9];
  {
    vr qw = LDG(qpos, 3), qx = LDG(qpos, 4), qy = LDG(qpos, 5), qz = LDG(qpos, 6);
    {  // mj_kinematics normalises the free joint's quaternion in qpos (every lane of the env's groups does, and stores, the same)
      const vr in_ = vrcp(vsqrt(qw * qw + qx * qx + qy * qy + qz * qz));
      qw = qw * in_; qx = qx * in_; qy = qy * in_; qz = qz * in_;
      STG(qpos, 3, qw); STG(qpos, 4, qx); STG(qpos, 5, qy); STG(qpos, 6, qz);
    }
    vr q00 = qw * qw, q01 = qw * qx, q02 = qw * qy, q03 = qw * qz, q11 = qx * qx, q12 = qx * qy, q13 = qx * qz, q22 = qy * qy,
       q23 = qy * qz, q33 = qz * qz;
    Rb[0] = q00 + q11 - q22 - q33; Rb[4] = q00 - q11 + q22 - q33; Rb[8] = q00 - q11 - q22 + q33;
    Rb[1] = real(2) * (q12 - q03); Rb[2] = real(2) * (q13 + q02); Rb[3] = real(2) * (q12 + q03);
    Rb[5] = real(2) * (q23 - q01); Rb[6] = real(2) * (q13 - q02); Rb[7] = real(2) * (q23 + q01);
  }
  vr vb[6], ab[6];  // base spatial velocity [w_world; v_origin] and bias acceleration (-gravity + v x w)
  {
    vr wl[3] = {LDG(qvel, 3), LDG(qvel, 4), LDG(qvel, 5)};
    matvec3(vb, Rb, wl);
    vb[3] = LDG(qvel, 0); vb[4] = LDG(qvel, 1); vb[5] = LDG(qvel, 2);
    vr t[3];
    cross3(t, vb + 3, vb);
    ab[0] = ab[1] = ab[2] = vr(real(0));
    ab[3] = t[0]; ab[4] = t[1]; ab[5] = t[2] + M.grav;
  }
#pragma unroll
  for (int j = 0; j < 9; j++) { STG(colR, j, Rb[j]); STG(Rb, j, Rb[j]); }
#pragma unroll
  for (int j = 0; j < 3; j++) STG(colp, j, vr(real(0)));
#pragma unroll
  for (int j = 0; j < 6; j++) STG(wv, j, vb[j]);

  // ---- forward pass down the chain. Serial in the link (every lane walks all three links of its leg): pose, motion vector,
  // velocity and bias acceleration. What then needs only ONE link's pose and velocities - spatial inertia about the base origin,
  // I a + v x* I v - is done once, by the lane group whose role is that link (role = group / G: links 0, 1, 2, and 2 again), instead
  // of three times by everybody: each lane keeps the state of its own link as the walk passes it. The fourth role does the same
  // arithmetic for the BASE body (pose Rb at the origin, velocity vb, bias acceleration ab, constants from M.basec).
  const V<int> mylink = (grp >> (G == 1 ? 0 : (G == 2 ? 1 : 2))) & 3;      // role: link 0, 1, 2 of the lane's leg, or 3 = the base body
  const VB isbase = mylink == 3;
#ifndef NM_NO_ROLE_ROT
  {  // the joint rotations (sine / cosine, Rodrigues about the local axis) do not depend on the parent: each role does its own link's
     // (the base role link 2's again) and leaves it in the leg staging area for the walk below
    const V<int> lk = vmin(mylink, V<int>(2));
    const V<int> cbr = leg * kLegN + lk * kLinkN;
    const vr ax[3] = {ldsv(M.legc, cbr + 12), ldsv(M.legc, cbr + 13), ldsv(M.legc, cbr + 14)};
    vr s_, co;
    vsincos(LDL(qpos, leg * 3 + lk + 7), &s_, &co);
    const vr oc = vr(real(1)) - co;
    vr Rl[9];
    Rl[0] = co + oc * ax[0] * ax[0]; Rl[1] = oc * ax[0] * ax[1] - s_ * ax[2]; Rl[2] = oc * ax[0] * ax[2] + s_ * ax[1];
    Rl[3] = oc * ax[0] * ax[1] + s_ * ax[2]; Rl[4] = co + oc * ax[1] * ax[1]; Rl[5] = oc * ax[1] * ax[2] - s_ * ax[0];
    Rl[6] = oc * ax[0] * ax[2] - s_ * ax[1]; Rl[7] = oc * ax[1] * ax[2] + s_ * ax[0]; Rl[8] = co + oc * ax[2] * ax[2];
#pragma unroll
    for (int j = 0; j < 9; j++) STL(legtmp, leg * 27 + lk * 9 + j, Rl[j]);
    wave_sync();
  }
#endif
  vr mcom[3] = {vr(real(0)), vr(real(0)), vr(real(0))};
  {
    vr Rp[9], pp[3], vpar[6], apar[6];
    vr cR[9], cpos[3], cS[6], cv[6], ca[6];      // this lane's link: pose, motion vector, velocity, bias acceleration
#pragma unroll
    for (int k = 0; k < 9; k++) Rp[k] = Rb[k];
    pp[0] = pp[1] = pp[2] = vr(real(0));
#pragma unroll
    for (int k = 0; k < 6; k++) { vpar[k] = vb[k]; apar[k] = ab[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      sched_fence();
      const V<int> cb = leg * kLegN + k * kLinkN;
      vr q = LDL(qpos, leg * 3 + (7 + k)), qd = LDL(qvel, leg * 3 + (6 + k));
      vr pos[3], t3[3], aw[3], R[9];
      {
        vr bpos[3] = {ldsv(M.legc, cb), ldsv(M.legc, cb + 1), ldsv(M.legc, cb + 2)};
        matvec3(t3, Rp, bpos);
        pos[0] = pp[0] + t3[0]; pos[1] = pp[1] + t3[1]; pos[2] = pp[2] + t3[2];
      }
      {
        vr ax[3] = {ldsv(M.legc, cb + 12), ldsv(M.legc, cb + 13), ldsv(M.legc, cb + 14)};
        vr R0[9];
        {
          if (uniform(M.link_rot_id[k])) {   // the link frame is not rotated against its parent's (femur, tibia): Rp * 1
#pragma unroll
            for (int j = 0; j < 9; j++) R0[j] = Rp[j];
          } else {
            vr bR[9];
#pragma unroll
            for (int j = 0; j < 9; j++) bR[j] = ldsv(M.legc, cb + 3 + j);
            matmul3(R0, Rp, bR);
          }
        }
        matvec3(aw, R0, ax);
        vr Rl[9];  // Rodrigues about the local axis
#ifndef NM_NO_ROLE_ROT
#pragma unroll
        for (int j = 0; j < 9; j++) Rl[j] = LDL(legtmp, leg * 27 + (k * 9 + j));
        (void)q;
#else
        vr s, co;
        vsincos(q, &s, &co);
        vr oc = vr(real(1)) - co;
        Rl[0] = co + oc * ax[0] * ax[0]; Rl[1] = oc * ax[0] * ax[1] - s * ax[2]; Rl[2] = oc * ax[0] * ax[2] + s * ax[1];
        Rl[3] = oc * ax[0] * ax[1] + s * ax[2]; Rl[4] = co + oc * ax[1] * ax[1]; Rl[5] = oc * ax[1] * ax[2] - s * ax[0];
        Rl[6] = oc * ax[0] * ax[2] - s * ax[1]; Rl[7] = oc * ax[1] * ax[2] + s * ax[0]; Rl[8] = co + oc * ax[2] * ax[2];
#endif
        matmul3(R, R0, Rl);
      }
      // publish joint anchor/axis (row stage) and, for the tibia, the collision frame
#pragma unroll
      for (int j = 0; j < 3; j++) {
        STL(anc, leg * 9 + (3 * k + j), pos[j]);
        STL(axs, leg * 9 + (3 * k + j), aw[j]);
      }
      if (k == 2) {
#pragma unroll
        for (int j = 0; j < 9; j++) STL(colR, (leg + 1) * 9 + j, R[j]);
#pragma unroll
        for (int j = 0; j < 3; j++) STL(colp, (leg + 1) * 3 + j, pos[j]);
      }
      sched_fence();
      vr S[6];
      S[0] = aw[0]; S[1] = aw[1]; S[2] = aw[2];
      cross3(S + 3, pos, aw);  // motion vector about the base origin: [a; r x a]
      {  // RNE forward: velocity and bias acceleration of this link
        vr Sd[6];
        cross_motion(Sd, vpar, S);
#pragma unroll
        for (int j = 0; j < 6; j++) {
          vpar[j] = vpar[j] + S[j] * qd;
          apar[j] = apar[j] + Sd[j] * qd;
        }
      }
      if (k == 0) {   // roles 1, 2 overwrite this below; role 3 (base) keeps the base's own state
#pragma unroll
        for (int j = 0; j < 9; j++) cR[j] = sel(isbase, Rb[j], R[j]);
#pragma unroll
        for (int j = 0; j < 3; j++) cpos[j] = sel(isbase, vr(real(0)), pos[j]);
#pragma unroll
        for (int j = 0; j < 6; j++) { cS[j] = S[j]; cv[j] = sel(isbase, vb[j], vpar[j]); ca[j] = sel(isbase, ab[j], apar[j]); }
      } else {
        const VB mine = mylink == k;
#pragma unroll
        for (int j = 0; j < 9; j++) cR[j] = sel(mine, R[j], cR[j]);
#pragma unroll
        for (int j = 0; j < 3; j++) cpos[j] = sel(mine, pos[j], cpos[j]);
#pragma unroll
        for (int j = 0; j < 6; j++) { cS[j] = sel(mine, S[j], cS[j]); cv[j] = sel(mine, vpar[j], cv[j]); ca[j] = sel(mine, apar[j], ca[j]); }
      }
#pragma unroll
      for (int j = 0; j < 9; j++) Rp[j] = R[j];
      pp[0] = pos[0]; pp[1] = pos[1]; pp[2] = pos[2];
    }
    sched_fence();
    {  // this lane's link: spatial inertia, body force; parked for the backward pass
      static_assert(offsetof(Model<real>, basec) == offsetof(Model<real>, legc) + sizeof(real) * kNLEG * kLegN, "basec follows legc");
      const V<int> cbm = sel(isbase, V<int>(kNLEG * kLegN - 15), leg * kLegN + mylink * kLinkN);   // base: ipos Ibody mass = basec[0..9]
      vr I10[10], f[6];
      {
        vr ipos[3] = {ldsv(M.legc, cbm + 15), ldsv(M.legc, cbm + 16), ldsv(M.legc, cbm + 17)};
        vr Ib[6];
#pragma unroll
        for (int j = 0; j < 6; j++) Ib[j] = ldsv(M.legc, cbm + 18 + j);
        vr mass = ldsv(M.legc, cbm + 24);
        vr d[3], t3[3];
        matvec3(t3, cR, ipos);
        d[0] = cpos[0] + t3[0]; d[1] = cpos[1] + t3[1]; d[2] = cpos[2] + t3[2];
        spatial_inertia(I10, cR, Ib, d, mass, real(0));
        mcom[0] = mass * d[0]; mcom[1] = mass * d[1]; mcom[2] = mass * d[2];
      }
      {
        vr t6[6], u6[6];
        inert_mul(t6, I10, ca);
        inert_mul(u6, I10, cv);
        cross_force(f, cv, u6);
#pragma unroll
        for (int j = 0; j < 6; j++) f[j] = f[j] + t6[j];
      }
      // links: the leg's staging slot; base: words 9..30 of the (still unused) Schur-complement buffer
      const V<int> o = sel(isbase, V<int>(NM_OFS(sc) + 9 - NM_OFS(legtmp)), slot + mylink * kLinkTmp);
#pragma unroll
      for (int j = 0; j < 6; j++) STL(legtmp, o + j, cS[j]);
#pragma unroll
      for (int j = 0; j < 10; j++) STL(legtmp, o + (6 + j), I10[j]);
#pragma unroll
      for (int j = 0; j < 6; j++) STL(legtmp, o + (16 + j), f[j]);
      if (last) {   // m d summed over the legs, per link: the subtree COM below adds the three links up
#pragma unroll
        for (int j = 0; j < 3; j++) stsv(lds, eo + mylink * 3 + (NM_OFS(sc) + j), legsum<real>(mcom[j], isleg), st_all);
      }
    }
  }
  wave_sync();

  // ---- backward pass: accumulated body forces -> bias; composite inertias -> M_l, Mlb
  vr Ml[6], Mlb[3][6], cl[3], fs[6], Ic[10];
#pragma unroll
  for (int j = 0; j < 6; j++) fs[j] = vr(real(0));
#pragma unroll
  for (int j = 0; j < 10; j++) Ic[j] = vr(real(0));
#pragma unroll
  for (int k = 2; k >= 0; k--) {
    sched_fence();
    const V<int> o = slot + k * kLinkTmp;
    vr S[6], F[6];
#pragma unroll
    for (int j = 0; j < 6; j++) S[j] = LDL(legtmp, o + j);
#pragma unroll
    for (int j = 0; j < 10; j++) Ic[j] += LDL(legtmp, o + (6 + j));
#pragma unroll
    for (int j = 0; j < 6; j++) fs[j] += LDL(legtmp, o + (16 + j));
    cl[k] = dot6<vr>(S, fs);
    inert_mul(F, Ic, S);
    Mlb[k][0] = F[3]; Mlb[k][1] = F[4]; Mlb[k][2] = F[5];
#pragma unroll
    for (int j = 0; j < 3; j++) Mlb[k][3 + j] = F[0] * Rb[j] + F[1] * Rb[3 + j] + F[2] * Rb[6 + j];
    const int dk = (k == 0) ? 0 : (k == 1 ? 3 : 5);  // packed index of M[k][k] in (00 01 02 11 12 22)
    Ml[dk] = dot6<vr>(S, F);
#pragma unroll
    for (int jj = 0; jj < k; jj++) {  // ancestors within the leg
      vr Sj[6];
#pragma unroll
      for (int j = 0; j < 6; j++) Sj[j] = LDL(legtmp, slot + (jj * kLinkTmp + j));
      Ml[(jj == 0) ? k : 4] = dot6<vr>(Sj, F);  // (0,k) -> index k ; (1,2) -> index 4
    }
  }
  sched_fence();
  // ---- base: own inertia/force + legs
  vr Icb[10], cbias[6];
  {
    vr Ib10[10];     // spatial inertia and body force of the base: computed by the base-role lane groups in the forward pass
#pragma unroll
    for (int j = 0; j < 10; j++) Ib10[j] = LDG(sc, 15 + j);
    if (last) {  // subtree COM (relative to the base origin) -> cvel[1] as MuJoCo reports it (about the COM)
      vr ipos[3] = {vr(M.basec[0]), vr(M.basec[1]), vr(M.basec[2])};
      vr d[3];
      matvec3(d, Rb, ipos);
      vr cr[3], t[3];
#pragma unroll
      for (int j = 0; j < 3; j++) cr[j] = (M.basec[9] * d[j] + (LDG(sc, j) + LDG(sc, 3 + j) + LDG(sc, 6 + j))) / M.total_mass;
      cross3(t, vb, cr);
#pragma unroll
      for (int j = 0; j < 3; j++) { STG(cvb, j, vb[j]); STG(cvb, 3 + j, vb[3 + j] + t[j]); }
      STG(bh, 0, LDG(qpos, 2) + d[2]);   // z of the base body's COM
    }
    vr fb[6];
#pragma unroll
    for (int j = 0; j < 6; j++) fb[j] = LDG(sc, 25 + j) + legsum<real>(fs[j], isleg);
    cbias[0] = fb[3]; cbias[1] = fb[4]; cbias[2] = fb[5];
#pragma unroll
    for (int j = 0; j < 3; j++) cbias[3 + j] = Rb[j] * fb[0] + Rb[3 + j] * fb[1] + Rb[6 + j] * fb[2];
#pragma unroll
    for (int j = 0; j < 10; j++) Icb[j] = Ib10[j] + legsum<real>(Ic[j], isleg);
  }
  // base block of M from the composite inertia (I6, h = m*d, m) about the base origin, dofs (x y z | body axes a_j = Rb[:,j]):
  //   trans-trans m*1 ; trans-rot column j = a_j x h ; rot-rot Rb' I Rb      (upper triangle, parked in LDS)
  {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = i; j < 3; j++) STG(mbb, 6 * i + j, (i == j) ? Icb[9] : vr(real(0)));
    vr T[9];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      vr a[3] = {Rb[j], Rb[3 + j], Rb[6 + j]}, c[3];
      cross3(c, a, Icb + 6);
#pragma unroll
      for (int i = 0; i < 3; i++) STG(mbb, 6 * i + 3 + j, c[i]);
      T[j] = Icb[0] * a[0] + Icb[3] * a[1] + Icb[4] * a[2];       // (I a_j), stored column-wise: T[3*r + j]
      T[3 + j] = Icb[3] * a[0] + Icb[1] * a[1] + Icb[5] * a[2];
      T[6 + j] = Icb[4] * a[0] + Icb[5] * a[1] + Icb[2] * a[2];
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = i; j < 3; j++) STG(mbb, 6 * (3 + i) + 3 + j, Rb[i] * T[j] + Rb[3 + i] * T[3 + j] + Rb[6 + i] * T[6 + j]);
  }
  wave_sync();

  // ---- block factorisations: M (lane groups [0, G)) and M + h*kv*I on the actuated dofs (groups [G, 2G), implicitfast), at once.
  // The second factor lives kFacH words behind the first in the env image; its Schur complement is staged in the (now free) leg
  // staging area, and what only the first factorisation feeds (qfrc_smooth, qacc_smooth) is written by the other groups to a
  // junk row of that area.
  {
    sched_fence();
    constexpr int kFacH = NM_OFS(MinvH) - NM_OFS(Minv);
    static_assert(NM_OFS(WH) - NM_OFS(W) == kFacH && NM_OFS(LbH) - NM_OFS(Lb) == kFacH && NM_OFS(DbiH) - NM_OFS(Dbi) == kFacH, "factor blocks");
    static_assert(kNLEG * 66 >= 36 + 24 + 24, "leg staging area holds the second Schur complement and the junk rows");
    const V<int> eoF = eo + sel(pass1, V<int>(kFacH), V<int>(0));
    const V<int> eoS = eo + sel(pass1, V<int>(NM_OFS(legtmp)), V<int>(NM_OFS(sc)));
    const V<int> eoQ = eo + sel(pass1, V<int>(NM_OFS(legtmp) + 36), V<int>(NM_OFS(qfs)));
    const V<int> eoA = eo + sel(pass1, V<int>(NM_OFS(legtmp) + 60), V<int>(NM_OFS(qas)));
    vr Mh[6];
    const vr dg = sel(pass1, vr(M.h * M.kv), vr(real(0)));
    Mh[0] = Ml[0] + dg; Mh[1] = Ml[1]; Mh[2] = Ml[2]; Mh[3] = Ml[3] + dg; Mh[4] = Ml[4]; Mh[5] = Ml[5] + dg;
    vr Mi[6], W[3][6];
    ldl3(Mi, Mh, real(1));
#pragma unroll
    for (int j = 0; j < 6; j++) {
      vr col[3] = {Mlb[0][j], Mlb[1][j], Mlb[2][j]}, r3[3];
      ldl3_solve(r3, Mi, col);
      W[0][j] = r3[0]; W[1][j] = r3[1]; W[2][j] = r3[2];
    }
#pragma unroll
    for (int j = 0; j < 6; j++) stsv(lds, eoF + leg * 6 + (NM_OFS(Minv) + j), Mi[j], st_leg);
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int j = 0; j < 6; j++) stsv(lds, eoF + leg * 18 + (NM_OFS(W) + 6 * k + j), W[k][j], st_leg);
    // Schur complement of the leg blocks (upper triangle, row-major in LDS)
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = i; j < 6; j++) {
        vr cij = Mlb[0][i] * W[0][j] + Mlb[1][i] * W[1][j] + Mlb[2][i] * W[2][j];
        stsv(lds, eoS + (6 * i + j), LDG(mbb, 6 * i + j) - legsum<real>(cij, isleg), st_all);
      }
    wave_sync();
    vr L[15], Di[6];
    ldl6([&](int r, int c) { return ldsv(lds, eoS + (6 * r + c)); }, L, Di, real(1));
#pragma unroll
    for (int j = 0; j < 15; j++) stsv(lds, eoF + (NM_OFS(Lb) + j), L[j], st_all);
#pragma unroll
    for (int j = 0; j < 6; j++) stsv(lds, eoF + (NM_OFS(Dbi) + j), Di[j], st_all);
    {
      // ---- servo forces, qfrc_smooth, qacc_smooth = M^-1 qfrc_smooth (block solve in the leg layout); the groups that hold the
      // factor of M write the results, the others a junk row
      vr y[3], t[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        vr ctrl = LDL(ctrl, leg * 3 + k), qd = LDL(qvel, leg * 3 + (6 + k));
        ctrl = vmin(vmax(ctrl, vr(-M.ctrl_max)), vr(M.ctrl_max));
        y[k] = M.kv * ctrl - M.kv * qd - cl[k];
        stsv(lds, eoQ + leg * 3 + (6 + k), y[k], st_leg);
      }
      ldl3_solve(t, Mi, y);
      vr xb[6];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        vr wy = W[0][j] * y[0] + W[1][j] * y[1] + W[2][j] * y[2];
        xb[j] = -cbias[j] - legsum<real>(wy, isleg);
        stsv(lds, eoQ + j, -cbias[j], st_all);
      }
      ldl6_solve(L, Di, xb);
#pragma unroll
      for (int j = 0; j < 6; j++) stsv(lds, eoA + j, xb[j], st_all);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        vr x = t[k];
#pragma unroll
        for (int j = 0; j < 6; j++) x = x - W[k][j] * xb[j];
        stsv(lds, eoA + leg * 3 + (6 + k), x, st_leg);
      }
    }
  }
  wave_sync();
#undef LDG
#undef STG
#undef LDL
#undef STL
}

// =========================================================================================  convex-convex (MPR)
// Tibia vs tibia (contype 2 / conaffinity 3, reference mjmodel.xml:47...). Restates libccd's ccdMPRPenetration the way
// MuJoCo 3.1.2's mjc_Convex drives it (centre = mesh COM, support = hull vertex of largest dot product, 50 iterations,
// tolerance 1e-6, one contact per pair). The portal logic is wave-uniform scalar code; every support query is a
// lane-parallel arg-max over the two hulls.
template <class real> struct Sup { real v[3], v1[3], v2[3]; };
template <class real> NM_FN real ccd_eps() { return sizeof(real) == 8 ? real(2.220446049250313e-16) : real(1.1920929e-7); }
template <class real> NM_FN bool ccd_zero(real x) { return vabs(x) < ccd_eps<real>(); }
template <class real> NM_FN bool ccd_eq(real a, real b) {
  real ab = vabs(a - b);
  if (ab < ccd_eps<real>()) return true;
  real fa = vabs(a), fb = vabs(b);
  return ab < ccd_eps<real>() * (fb > fa ? fb : fa);
}
// `real` is the precision of the env state (LDS image, hull table), `acc` the precision MPR computes in. The fp32 kernel runs MPR in
// fp64 (acc = double): the portal refinement is a chain of discrete decisions on cross products of nearly parallel differences, and in
// fp32 5 % of the env-steps with deeply interpenetrating tibias picked another portal than the fp64 reference (errors up to 8 in the
// observation: profiles/r03_mpr_fp32_study.txt). The path is rare, wave-uniform and out of line, so the price is code size only.
template <class acc, class real> NM_FN void hull_support(const Sh<real>& sh, const Model<real>& M, int g, const acc* dir, acc* out) {
  typedef V<acc> vr;
  const V<int> lane = lane_id();
  acc R[9], p[3];
#pragma unroll
  for (int j = 0; j < 9; j++) R[j] = (acc)sh.colR[9 * g + j];
#pragma unroll
  for (int j = 0; j < 3; j++) p[j] = (acc)sh.colp[3 * g + j];
  const real* cc = M.colc + kColN * g;
  const int nvert = (int)cc[5], vadr = (int)cc[6];
  const acc ld[3] = {R[0] * dir[0] + R[3] * dir[1] + R[6] * dir[2], R[1] * dir[0] + R[4] * dir[1] + R[7] * dir[2],
                     R[2] * dir[0] + R[5] * dir[1] + R[8] * dir[2]};
  // two passes: wave max, then the lowest vertex index within kMprTie of it (tie-robust, see oracle hull_support)
  vr best = vr(acc(-1e30));
  for (int it = 0; it * NM_WAVE < nvert; it++) {
    V<int> vi = lane + it * NM_WAVE;
    VB ok = vi < nvert;
    V<int> ad = (sel(ok, vi, V<int>(0)) + vadr) * 4;
    vr val = ld[0] * vcvt<acc>(gldv(M.hullv, ad)) + ld[1] * vcvt<acc>(gldv(M.hullv, ad + 1)) + ld[2] * vcvt<acc>(gldv(M.hullv, ad + 2));
    best = sel(ok & (val > best), val, best);
  }
  acc sv; int si;
  wargmax(best, lane, &sv, &si);
  V<int> ifirst = V<int>(1 << 30);
  for (int it = 0; it * NM_WAVE < nvert; it++) {
    V<int> vi = lane + it * NM_WAVE;
    VB ok = vi < nvert;
    V<int> ad = (sel(ok, vi, V<int>(0)) + vadr) * 4;
    vr val = ld[0] * vcvt<acc>(gldv(M.hullv, ad)) + ld[1] * vcvt<acc>(gldv(M.hullv, ad + 1)) + ld[2] * vcvt<acc>(gldv(M.hullv, ad + 2));
    ifirst = sel(ok & (val >= vr(sv - acc(1e-7))) & (vi < ifirst), vi, ifirst);
  }
  wargmax(to_real<acc>(-ifirst), lane, &sv, &si);  // largest -index = lowest index
  si = (int)(-sv);
  const real* vp = M.hullv + 4 * (vadr + si);
  const acc v[3] = {(acc)vp[0], (acc)vp[1], (acc)vp[2]};
  acc t[3];
  matvec3(t, R, v);
  out[0] = p[0] + t[0]; out[1] = p[1] + t[1]; out[2] = p[2] + t[2];
}
template <class acc, class real> NM_COLD void mpr_support(const Sh<real>& sh, const Model<real>& M, int g1, int g2, const acc* dir, Sup<acc>& s) {
  acc nd[3] = {-dir[0], -dir[1], -dir[2]};
  hull_support<acc>(sh, M, g1, dir, s.v1);
  hull_support<acc>(sh, M, g2, nd, s.v2);
  s.v[0] = s.v1[0] - s.v2[0]; s.v[1] = s.v1[1] - s.v2[1]; s.v[2] = s.v1[2] - s.v2[2];
}
template <class real> NM_FN void sub3(real* r, const real* a, const real* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
template <class real> NM_FN void ccd_normalize(real* v) { real s = real(1) / vsqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] *= s; v[1] *= s; v[2] *= s; }
template <class real> NM_FN void portal_dir(const Sup<real>* p, real* dir) {
  real a[3], b[3];
  sub3(a, p[2].v, p[1].v); sub3(b, p[3].v, p[1].v);
  cross3(dir, a, b);
  ccd_normalize(dir);
}
template <class real> NM_FN bool portal_reach_tol(const Sup<real>* p, const Sup<real>& v4, const real* dir, real tol) {
  real dv4 = dot3<real>(v4.v, dir);
  real d1 = dv4 - dot3<real>(p[1].v, dir), d2 = dv4 - dot3<real>(p[2].v, dir), d3 = dv4 - dot3<real>(p[3].v, dir);
  real m = vmin(vmin(d1, d2), d3);
  return ccd_eq(m, tol) || m < tol;
}
template <class real> NM_FN void expand_portal(Sup<real>* p, const Sup<real>& v4) {
  real v4v0[3];
  cross3(v4v0, v4.v, p[0].v);
  if (dot3<real>(p[1].v, v4v0) > real(0)) {
    if (dot3<real>(p[2].v, v4v0) > real(0)) p[1] = v4; else p[3] = v4;
  } else {
    if (dot3<real>(p[3].v, v4v0) > real(0)) p[2] = v4; else p[1] = v4;
  }
}
template <class real> NM_FN real point_seg_dist2(const real* x0, const real* b, real* wit) {  // point P = origin
  real dd[3];
  sub3(dd, b, x0);
  real t = -dot3<real>(x0, dd) / dot3<real>(dd, dd);
  if (t < real(0) || ccd_zero(t)) { wit[0] = x0[0]; wit[1] = x0[1]; wit[2] = x0[2]; }
  else if (t > real(1) || ccd_eq(t, real(1))) { wit[0] = b[0]; wit[1] = b[1]; wit[2] = b[2]; }
  else { wit[0] = x0[0] + t * dd[0]; wit[1] = x0[1] + t * dd[1]; wit[2] = x0[2] + t * dd[2]; }
  return dot3<real>(wit, wit);
}
template <class real> NM_FN real point_tri_dist2(const real* x0, const real* B, const real* C, real* wit) {  // point P = origin
  real d1[3], d2[3];
  sub3(d1, B, x0); sub3(d2, C, x0);
  real v = dot3<real>(d1, d1), w = dot3<real>(d2, d2), p = dot3<real>(x0, d1), q = dot3<real>(x0, d2), r = dot3<real>(d1, d2);
  real dd = w * v - r * r, s, t;
  if (ccd_zero(dd)) { s = real(-1); t = real(-1); }
  else { s = (q * r - w * p) / dd; t = (-s * r - q) / w; }
  if ((ccd_zero(s) || s > real(0)) && (ccd_eq(s, real(1)) || s < real(1)) && (ccd_zero(t) || t > real(0)) && (ccd_eq(t, real(1)) || t < real(1)) &&
      (ccd_eq(t + s, real(1)) || t + s < real(1))) {
    wit[0] = x0[0] + s * d1[0] + t * d2[0]; wit[1] = x0[1] + s * d1[1] + t * d2[1]; wit[2] = x0[2] + s * d1[2] + t * d2[2];
    return dot3<real>(wit, wit);
  }
  real w2[3];
  real dist = point_seg_dist2(x0, B, wit);
  real dist2 = point_seg_dist2(x0, C, w2);
  if (dist2 < dist) { dist = dist2; wit[0] = w2[0]; wit[1] = w2[1]; wit[2] = w2[2]; }
  dist2 = point_seg_dist2(B, C, w2);
  if (dist2 < dist) { dist = dist2; wit[0] = w2[0]; wit[1] = w2[1]; wit[2] = w2[2]; }
  return dist;
}
// true and (depth, dir from geom1 to geom2, pos) when the two hulls penetrate
template <class real_, class st> NM_COLD bool mpr_penetration(const Sh<st>& sh, const Model<st>& M, int g1, int g2, real_* depth, real_* dir_out, real_* pos) {
  typedef real_ real;   // everything below computes in `real` (= acc); the env state is `st`
  Sup<real> p[4], v4;
  real dir[3], va[3], vb[3], dot;
  const real mpr_tol = (real)M.mpr_tol;
  {  // discoverPortal: v0 = centre1 - centre2
    real R1[9], R2[9], c1[3], c2[3];
#pragma unroll
    for (int k = 0; k < 9; k++) { R1[k] = (real)sh.colR[9 * g1 + k]; R2[k] = (real)sh.colR[9 * g2 + k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { c1[k] = (real)M.colc[kColN * g1 + k]; c2[k] = (real)M.colc[kColN * g2 + k]; }
    real t1[3], t2[3];
    matvec3(t1, R1, c1); matvec3(t2, R2, c2);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      p[0].v1[k] = (real)sh.colp[3 * g1 + k] + t1[k];
      p[0].v2[k] = (real)sh.colp[3 * g2 + k] + t2[k];
      p[0].v[k] = p[0].v1[k] - p[0].v2[k];
    }
  }
  if (ccd_eq(p[0].v[0], real(0)) && ccd_eq(p[0].v[1], real(0)) && ccd_eq(p[0].v[2], real(0))) p[0].v[0] += ccd_eps<real>() * real(10);
  dir[0] = -p[0].v[0]; dir[1] = -p[0].v[1]; dir[2] = -p[0].v[2];
  ccd_normalize(dir);
  mpr_support<real>(sh, M, g1, g2, dir, p[1]);
  dot = dot3<real>(p[1].v, dir);
  if (ccd_zero(dot) || dot < real(0)) return false;
  cross3(dir, p[0].v, p[1].v);
  if (ccd_zero(dot3<real>(dir, dir))) {
    if (ccd_eq(p[1].v[0], real(0)) && ccd_eq(p[1].v[1], real(0)) && ccd_eq(p[1].v[2], real(0))) return false;  // touching: direction undefined
    *depth = vsqrt(dot3<real>(p[1].v, p[1].v));   // origin on the v0-v1 segment
    dir_out[0] = p[1].v[0]; dir_out[1] = p[1].v[1]; dir_out[2] = p[1].v[2];
    ccd_normalize(dir_out);
#pragma unroll
    for (int k = 0; k < 3; k++) pos[k] = real(0.5) * (p[1].v1[k] + p[1].v2[k]);
    return true;
  }
  ccd_normalize(dir);
  mpr_support<real>(sh, M, g1, g2, dir, p[2]);
  dot = dot3<real>(p[2].v, dir);
  if (ccd_zero(dot) || dot < real(0)) return false;
  sub3(va, p[1].v, p[0].v); sub3(vb, p[2].v, p[0].v);
  cross3(dir, va, vb);
  ccd_normalize(dir);
  if (dot3<real>(dir, p[0].v) > real(0)) {
    Sup<real> t = p[1]; p[1] = p[2]; p[2] = t;
    dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2];
  }
  for (int size = 3, guard = 0; size < 4; guard++) {
    if (guard > 64) return false;
    mpr_support<real>(sh, M, g1, g2, dir, p[3]);
    dot = dot3<real>(p[3].v, dir);
    if (ccd_zero(dot) || dot < real(0)) return false;
    bool cont = false;
    cross3(va, p[1].v, p[3].v);
    dot = dot3<real>(va, p[0].v);
    if (dot < real(0) && !ccd_zero(dot)) { p[2] = p[3]; cont = true; }
    if (!cont) {
      cross3(va, p[3].v, p[2].v);
      dot = dot3<real>(va, p[0].v);
      if (dot < real(0) && !ccd_zero(dot)) { p[1] = p[3]; cont = true; }
    }
    if (cont) {
      sub3(va, p[1].v, p[0].v); sub3(vb, p[2].v, p[0].v);
      cross3(dir, va, vb);
      ccd_normalize(dir);
    } else size = 4;
  }
  for (int guard = 0;; guard++) {  // refinePortal
    portal_dir(p, dir);
    dot = dot3<real>(dir, p[1].v);
    if (ccd_zero(dot) || dot > real(0)) break;
    if (guard > 100) return false;
    mpr_support<real>(sh, M, g1, g2, dir, v4);
    dot = dot3<real>(v4.v, dir);
    if (!(ccd_zero(dot) || dot > real(0)) || portal_reach_tol(p, v4, dir, mpr_tol)) return false;
    expand_portal(p, v4);
  }
  for (int it = 0;; it++) {  // findPenetr
    portal_dir(p, dir);
    mpr_support<real>(sh, M, g1, g2, dir, v4);
    if (portal_reach_tol(p, v4, dir, mpr_tol) || it > M.mpr_iters) {
      real pd[3];
      *depth = vsqrt(point_tri_dist2(p[1].v, p[2].v, p[3].v, pd));
      if (ccd_zero(pd[0]) && ccd_zero(pd[1]) && ccd_zero(pd[2])) { pd[0] = dir[0]; pd[1] = dir[1]; pd[2] = dir[2]; }
      ccd_normalize(pd);
      dir_out[0] = pd[0]; dir_out[1] = pd[1]; dir_out[2] = pd[2];
      real b[4], t[3], sum;  // findPos: barycentric coordinates of the origin in the portal tetrahedron
      portal_dir(p, dir);
      cross3(t, p[1].v, p[2].v); b[0] = dot3<real>(t, p[3].v);
      cross3(t, p[3].v, p[2].v); b[1] = dot3<real>(t, p[0].v);
      cross3(t, p[0].v, p[1].v); b[2] = dot3<real>(t, p[3].v);
      cross3(t, p[2].v, p[1].v); b[3] = dot3<real>(t, p[0].v);
      sum = b[0] + b[1] + b[2] + b[3];
      if (ccd_zero(sum) || sum < real(0)) {
        b[0] = real(0);
        cross3(t, p[2].v, p[3].v); b[1] = dot3<real>(t, dir);
        cross3(t, p[3].v, p[1].v); b[2] = dot3<real>(t, dir);
        cross3(t, p[1].v, p[2].v); b[3] = dot3<real>(t, dir);
        sum = b[1] + b[2] + b[3];
      }
      real inv = real(1) / sum;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        real p1 = b[0] * p[0].v1[k] + b[1] * p[1].v1[k] + b[2] * p[2].v1[k] + b[3] * p[3].v1[k];
        real p2 = b[0] * p[0].v2[k] + b[1] * p[1].v2[k] + b[2] * p[2].v2[k] + b[3] * p[3].v2[k];
        pos[k] = real(0.5) * (p1 * inv + p2 * inv);
      }
      return true;
    }
    expand_portal(p, v4);
  }
}

// tibia-tibia pairs: lanes 0..14 run the cull (MuJoCo's bounding-sphere filter AND a conservative separating-axis test of
// the two hull OBBs along the line of centres); surviving pairs go through MPR one at a time.
// the pairs the cull let through (bit p of m = pair p), one at a time through MPR
template <class real> NM_FN void pairs_narrow(Sh<real>& sh, const Model<real>& M, int* dropped, uint64_t m) {
  if (!m) return;
  int ncon = sh.ncon;
  while (m) {
    int pidx = __builtin_ctzll(m);
    m &= m - 1;
    int a = (pidx >= 5) + (pidx >= 9) + (pidx >= 12) + (pidx >= 14);
    int s0 = a == 0 ? 0 : (a == 1 ? 5 : (a == 2 ? 9 : (a == 3 ? 12 : 14)));
    int h1 = a + 1, h2 = h1 + 1 + (pidx - s0);
    {  // mj_filterSphere on the COM-centred bounding spheres
      const real *R1 = sh.colR + 9 * h1, *R2 = sh.colR + 9 * h2, *k1 = M.colc + kColN * h1, *k2 = M.colc + kColN * h2;
      real t1[3], t2[3], dd[3];
      matvec3(t1, R1, k1); matvec3(t2, R2, k2);
#pragma unroll
      for (int k = 0; k < 3; k++) dd[k] = (sh.colp[3 * h1 + k] + t1[k]) - (sh.colp[3 * h2 + k] + t2[k]);
      real bound = k1[3] + k2[3];
      if (dot3<real>(dd, dd) > bound * bound) continue;
    }
    typedef double acc;      // MPR computes in fp64 in both builds (see hull_support)
    acc depth, dir[3], pos[3];
    if (!mpr_penetration<acc>(sh, M, h1, h2, &depth, dir, pos)) continue;
    if (!(depth > acc(0))) continue;
    if (ncon >= kMaxConBig) { *dropped += 1; continue; }   // cannot happen: see kMaxConBig
    sh.cpos()[3 * ncon] = (real)pos[0]; sh.cpos()[3 * ncon + 1] = (real)pos[1]; sh.cpos()[3 * ncon + 2] = (real)pos[2];
    sh.cnrm()[3 * ncon] = (real)dir[0]; sh.cnrm()[3 * ncon + 1] = (real)dir[1]; sh.cnrm()[3 * ncon + 2] = (real)dir[2];
    sh.cdist()[ncon] = (real)(-depth);
    sh.cleg()[ncon] = h2 - 1;
    sh.cleg1()[ncon] = h1 - 1;
    sh.anypair = 1;
    ncon++;
  }
  sh.ncon = ncon;
  wave_sync();
}


template <class real> NM_FN void stage_collide_pairs(Sh<real>& sh, const Model<real>& M, int* dropped) {
  typedef V<real> vr;
  const V<int> lane = lane_id();
  const VB isp = lane < 15;
  const V<int> pl = sel(isp, lane, V<int>(0));
  V<int> i1 = sel(pl >= 5, V<int>(1), V<int>(0)) + sel(pl >= 9, V<int>(1), V<int>(0)) + sel(pl >= 12, V<int>(1), V<int>(0)) +
              sel(pl >= 14, V<int>(1), V<int>(0));                       // 0..4
  V<int> st = sel(i1 == 0, V<int>(0), sel(i1 == 1, V<int>(5), sel(i1 == 2, V<int>(9), sel(i1 == 3, V<int>(12), V<int>(14)))));
  V<int> g1 = i1 + 1, g2 = g1 + 1 + (pl - st);
  vr c1[3], c2[3], e[2];
  V<int> gg[2] = {g1, g2};
  vr u[3];
#pragma unroll
  for (int w = 0; w < 2; w++) {
    vr R[9], oc[3], c[3];
#pragma unroll
    for (int j = 0; j < 9; j++) R[j] = ldsv(sh.colR, gg[w] * 9 + j);
#pragma unroll
    for (int j = 0; j < 3; j++) oc[j] = ldsv(M.colc, gg[w] * kColN + (8 + j));
    matvec3(c, R, oc);
#pragma unroll
    for (int j = 0; j < 3; j++) { c[j] = c[j] + ldsv(sh.colp, gg[w] * 3 + j); if (w == 0) c1[j] = c[j]; else c2[j] = c[j]; }
  }
  u[0] = c2[0] - c1[0]; u[1] = c2[1] - c1[1]; u[2] = c2[2] - c1[2];
  vr dist = vsqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  vr inv = vr(real(1)) / vmax(dist, vr(real(1e-12)));
  u[0] = u[0] * inv; u[1] = u[1] * inv; u[2] = u[2] * inv;
#pragma unroll
  for (int w = 0; w < 2; w++) {  // extent of each OBB along u: sum_i h_i |u . (R a_i)|
    vr R[9], ul[3];
#pragma unroll
    for (int j = 0; j < 9; j++) R[j] = ldsv(sh.colR, gg[w] * 9 + j);
    ul[0] = R[0] * u[0] + R[3] * u[1] + R[6] * u[2]; ul[1] = R[1] * u[0] + R[4] * u[1] + R[7] * u[2]; ul[2] = R[2] * u[0] + R[5] * u[1] + R[8] * u[2];
    vr ext = vr(real(0));
#pragma unroll
    for (int a = 0; a < 3; a++) {
      vr ax[3] = {ldsv(M.colc, gg[w] * kColN + (11 + 3 * a)), ldsv(M.colc, gg[w] * kColN + (12 + 3 * a)), ldsv(M.colc, gg[w] * kColN + (13 + 3 * a))};
      ext += ldsv(M.colc, gg[w] * kColN + (20 + a)) * vabs(ax[0] * ul[0] + ax[1] * ul[1] + ax[2] * ul[2]);
    }
    e[w] = ext;
  }
  // MuJoCo's sphere filter is on the mesh COMs; it is implied by the (tighter) OBB test in practice, but keep both exact-conservative
  uint64_t m = ballot(isp & !(dist > e[0] + e[1]));
  pairs_narrow(sh, M, dropped, m);
}

// the same cull for BOTH envs of a wave at once (lanes 0..14 env 0, lanes 32..46 env 1: same arithmetic per lane)
template <class real> NM_FN void stage_collide_pairs2(ShW<real, 2>& w, const Model<real>& M, int* dropped) {
  typedef V<real> vr;
  const V<int> lane = lane_id();
  const V<int> hl = lane & 31, ho = (lane >> 5) * (int)(sizeof(Sh<real>) / sizeof(real));
  const real* rb = reinterpret_cast<const real*>(&w.e[0]);
  const VB isp = hl < 15;
  const V<int> pl = sel(isp, hl, V<int>(0));
  V<int> i1 = sel(pl >= 5, V<int>(1), V<int>(0)) + sel(pl >= 9, V<int>(1), V<int>(0)) + sel(pl >= 12, V<int>(1), V<int>(0)) +
              sel(pl >= 14, V<int>(1), V<int>(0));                       // 0..4
  V<int> st = sel(i1 == 0, V<int>(0), sel(i1 == 1, V<int>(5), sel(i1 == 2, V<int>(9), sel(i1 == 3, V<int>(12), V<int>(14)))));
  V<int> g1 = i1 + 1, g2 = g1 + 1 + (pl - st);
  vr c1[3], c2[3], e[2];
  V<int> gg[2] = {g1, g2};
  vr u[3];
#pragma unroll
  for (int w = 0; w < 2; w++) {
    vr R[9], oc[3], c[3];
#pragma unroll
    for (int j = 0; j < 9; j++) R[j] = ldsv(rb, ho + (gg[w] * 9 + (j + NM_OFS(colR))));
#pragma unroll
    for (int j = 0; j < 3; j++) oc[j] = ldsv(M.colc, gg[w] * kColN + (8 + j));
    matvec3(c, R, oc);
#pragma unroll
    for (int j = 0; j < 3; j++) { c[j] = c[j] + ldsv(rb, ho + (gg[w] * 3 + (j + NM_OFS(colp)))); if (w == 0) c1[j] = c[j]; else c2[j] = c[j]; }
  }
  u[0] = c2[0] - c1[0]; u[1] = c2[1] - c1[1]; u[2] = c2[2] - c1[2];
  vr dist = vsqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  vr inv = vr(real(1)) / vmax(dist, vr(real(1e-12)));
  u[0] = u[0] * inv; u[1] = u[1] * inv; u[2] = u[2] * inv;
#pragma unroll
  for (int w = 0; w < 2; w++) {  // extent of each OBB along u: sum_i h_i |u . (R a_i)|
    vr R[9], ul[3];
#pragma unroll
    for (int j = 0; j < 9; j++) R[j] = ldsv(rb, ho + (gg[w] * 9 + (j + NM_OFS(colR))));
    ul[0] = R[0] * u[0] + R[3] * u[1] + R[6] * u[2]; ul[1] = R[1] * u[0] + R[4] * u[1] + R[7] * u[2]; ul[2] = R[2] * u[0] + R[5] * u[1] + R[8] * u[2];
    vr ext = vr(real(0));
#pragma unroll
    for (int a = 0; a < 3; a++) {
      vr ax[3] = {ldsv(M.colc, gg[w] * kColN + (11 + 3 * a)), ldsv(M.colc, gg[w] * kColN + (12 + 3 * a)), ldsv(M.colc, gg[w] * kColN + (13 + 3 * a))};
      ext += ldsv(M.colc, gg[w] * kColN + (20 + a)) * vabs(ax[0] * ul[0] + ax[1] * ul[1] + ax[2] * ul[2]);
    }
    e[w] = ext;
  }
  // MuJoCo's sphere filter is on the mesh COMs; it is implied by the (tighter) OBB test in practice, but keep both exact-conservative
  uint64_t m = ballot(isp & !(dist > e[0] + e[1]));
  pairs_narrow(w.e[0], M, dropped, m & 0xffffffffull);
  pairs_narrow(w.e[1], M, dropped, m >> 32);
}

// =========================================================================================  stage B
// Floor (z = 0) against the convex hulls of base_link and the six tibias.
// Support vertex of hull g along ld (mesh frame) by exhaustive scan, lowest index wins ties (the reference rule).
constexpr int kHullIters = 5;   // ceil(largest hull / 64): 284 vertices (host asserts it)
template <class real> NM_FN int support_exhaustive(const Model<real>& M, const real* ld, int nvert, int vadr) {
  typedef V<real> vr;
  const V<int> lane = lane_id();
  vr best = vr(real(-1e30));
  V<int> ibest = lane;
  vr v[kHullIters][3];
#pragma unroll
  for (int it = 0; it < kHullIters; it++) {   // all gathers in flight at once: one L2 round trip, not five
    V<int> vi = lane + it * NM_WAVE;
    gld3(M.hullv, (sel(vi < nvert, vi, V<int>(0)) + vadr) * 4, v[it]);
  }
#pragma unroll
  for (int it = 0; it < kHullIters; it++) {
    V<int> vi = lane + it * NM_WAVE;
    VB ok = vi < nvert;
    vr val = ld[0] * v[it][0] + ld[1] * v[it][1] + ld[2] * v[it][2];
    VB take = ok & (val > best);
    best = sel(take, val, best);
    ibest = sel(take, vi, ibest);
  }
  // one lane holds the maximum in all but exact ties: then its vertex is the answer without carrying indices through the reduction
  real sv; int si;
  wmaxfirst(best, &sv, &si);
  if (popc64(ballot(best == vr(sv))) == 1) return rdlane(ibest, si);
  wargmax(best, ibest, &sv, &si);   // exact tie across lanes: the lowest vertex index wins (the reference rule)
  return si;
}
template <class real> NM_FN real hull_tie_tol() { return sizeof(real) == 8 ? real(1e-13) : real(4e-7); }  // metres; >> rounding of ld.v
constexpr int kMaxHop = 3;
template <class real> constexpr real kObbSlack() { return real(1e-5); }   // metres; >> the float32 rounding of the box and of the hull vertices

// Floor (z = 0) against the convex hulls of base_link and the six tibias. Like mjc_support, the search for the
// support vertex is warm-started: lanes 0..maxnbr-1 evaluate the hull neighbours of last time's support vertex and lane
// 63 the vertex itself, for ALL seven meshes with independent (batched) gathers; only when a neighbour ties or beats it
// does the mesh fall back to the exhaustive scan - so the result is always the exhaustive one (lowest index on ties).
// The same neighbour data then yields the <= 3 extra plane-mesh contacts.
constexpr int kSelfLane = 63;
// A plane-mesh contact brings up to three more: penetrating hull neighbours of the support vertex that lie >= tol_planemesh x rbound from it.
// When even the farthest neighbour of the vertex is nearer than that - every vertex of a foot tip (neighbours within 3 mm, tolerance 37 mm) -
// the contact is a lone one whatever the pose: the table holds, per vertex, slack = 0.999 tol - 1.00001 max |neighbour - vertex| in units of
// 2^-20 m (an integer, exact in float32; < 0 = never), and 0.5 |dist| < slack certifies it (the first contact sits 0.5 dist off the vertex).
constexpr int kSlackScale = 1 << 20;
constexpr int kBatchLane0 = kSelfLane + 1 - kNCOL;   // the batched emission puts mesh g on lane kBatchLane0 + g (those lanes hold the mesh's support vertex)
// lanes 0..maxnbr-1: the hull neighbours of global vertex gv (id -1 = none), lane 63: the vertex itself (local id `self`)
// lanes maxnbr.. of nbg: the vertex's `lone-contact slack` (its table entry's fourth word, see kSlackScale) - neighbour ids are read under
// `lane < maxnbr` only
template <class real> NM_FN void hull_ring(const Model<real>& M, int gv, V<int>& nbg, V<real>* v) {
  const V<int> lane = lane_id();
  const VB nbl = lane < M.maxnbr;
  V<real> o[4];
  gld4(M.hullnv, (sel(nbl, lane, V<int>(M.maxnbr)) + gv * (M.maxnbr + 1)) * 4, o);
  v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
  nbg = to_int(o[3]);
}
// the seven ring gathers of one env (L2 round trips): issued apart from their use, so that a wave can have both of its envs' in flight
template <class real> struct Rings {
  V<int> nb[kNCOL];
  V<real> vv[kNCOL][3];
  int cur[kNCOL];
};
template <class real> NM_FN void collide_rings(const Sh<real>& sh, const Model<real>& M, Rings<real>& r) {
#pragma unroll
  for (int g = 0; g < kNCOL; g++) {
    const int vadr = (int)M.colc[kColN * g + 6];
    r.cur[g] = uniform(sh.hcache[g]);
    hull_ring(M, vadr + r.cur[g], r.nb[g], r.vv[g]);
  }
}
template <class real> NM_FN void stage_collide(Sh<real>& sh, const Model<real>& M, int* dropped, Rings<real>& rg, bool pairs = true) {
  typedef V<real> vr;
  const V<int> lane = lane_id();
  const real bz = sh.qpos[2];
  const VB nbl = lane < M.maxnbr;
  V<int>* nb = rg.nb;
  vr (*vv)[3] = rg.vv;
  int* cur = rg.cur;
  // The seven meshes go through the stage together, phase by phase, so that their (independent) dependency chains overlap:
  // P1 frame scalars + bounding-sphere prefilter, P2 support values and the occasional hill climb, P3 contact emission with
  // contact slots from a running count (the order - mesh by mesh, support vertex first, then neighbours in graph order - is
  // what the Gauss-Seidel sweeps downstream depend on).
  real pz[kNCOL], ldv[kNCOL][3], sv[kNCOL];
  bool near[kNCOL];
  vr val[kNCOL];
#pragma unroll
  for (int g = 0; g < kNCOL; g++) {
    const real* R = sh.colR + 9 * g;
    const real* cc = M.colc + kColN * g;
    pz[g] = sh.colp[3 * g + 2] + bz;
    const real cz = R[6] * cc[0] + R[7] * cc[1] + R[8] * cc[2] + pz[g];
    near[g] = !(cz - cc[3] > real(0));                    // bounding-sphere prefilter
    ldv[g][0] = -R[6]; ldv[g][1] = -R[7]; ldv[g][2] = -R[8];  // -normal in the mesh frame
    val[g] = ldv[g][0] * vv[g][0] + ldv[g][1] * vv[g][1] + ldv[g][2] * vv[g][2];
    sv[g] = rdlane(val[g], kSelfLane);
  }
  NM_BSTAMP(11);
#pragma unroll
  for (int g = 0; g < kNCOL; g++) {
    const real tol = hull_tie_tol<real>();
    if (near[g] && wany(nbl & (nb[g] >= 0) & (val[g] >= vr(sv[g] - tol)))) {
      // Before searching: a mesh whose oriented bounding box is clear of the floor cannot touch it, whichever vertex is its support
      // vertex - no walk, no exhaustive scan, the warm start stays. (The bounding SPHERE of the base reaches the floor whenever the
      // robot stands, and two of its bottom vertices lie 28 um apart: without this test a level base ties and takes the exhaustive
      // scan in every forward pass - 2.0 per env-step in the standing regime, profiles/r05_fallback_study.txt.)
      {
        const real* cc = M.colc + kColN * g;
        real top = ldv[g][0] * cc[8] + ldv[g][1] * cc[9] + ldv[g][2] * cc[10];     // support value of the box along -normal
#pragma unroll
        for (int a = 0; a < 3; a++) top += cc[20 + a] * vabs(ldv[g][0] * cc[11 + 3 * a] + ldv[g][1] * cc[12 + 3 * a] + ldv[g][2] * cc[13 + 3 * a]);
        if (uniform(pz[g] - top > kObbSlack<real>())) { near[g] = false; continue; }
      }
      // Hill climbing on the hull graph (what mjc_support's warm start does): while a neighbour is clearly higher, move to
      // the highest one. A vertex that beats all its neighbours by a margin is the support vertex of a convex hull. Near
      // ties (and walks longer than kMaxHop) go to the exhaustive scan, whose lowest-index rule is the reference behaviour.
      const int nvert = (int)M.colc[kColN * g + 5], vadr = (int)M.colc[kColN * g + 6];
      const real* ld = ldv[g];
      int si = cur[g];
      bool full = false;
      for (int hop = 0;; hop++) {
        const VB nbv = nbl & (nb[g] >= 0);
        if (!wany(nbv & (val[g] >= vr(sv[g] - tol)))) break;
        if (hop == kMaxHop || !wany(nbv & (val[g] > vr(sv[g] + tol)))) { full = true; break; }
        // the best neighbour; among exactly equal ones any will do (the climb only has to arrive: a strict maximum is THE support
        // vertex of a convex hull, anything else goes to the exhaustive scan and its lowest-index rule)
        real bv; int lbest;
        wmaxfirst(sel(nbv, val[g], vr(real(-1e30))), &bv, &lbest);
        si = rdlane(nb[g], lbest);
        NM_BSTAMP(12);
        hull_ring(M, vadr + si, nb[g], vv[g]);
        val[g] = ld[0] * vv[g][0] + ld[1] * vv[g][1] + ld[2] * vv[g][2];
        sv[g] = rdlane(val[g], kSelfLane);
        NM_BSTAMP(14);
        sh.nhop += 1;
      }
      if (full) {
        NM_BSTAMP(12);
        const int held = si;                    // the vertex whose ring the lanes hold
        si = support_exhaustive(M, ld, nvert, vadr);
        sh.hcache[7] += 1;
#ifdef NM_MEASURE
        sh.nhop += g == 0 ? 1000 : 100000;   // which mesh fell back, for scripts/fallback_study.py (debug slot 159)
#endif
        if (si != held) {                       // a tie fallback usually confirms the vertex it started from: no second dependent gather then
          hull_ring(M, vadr + si, nb[g], vv[g]);
          val[g] = ld[0] * vv[g][0] + ld[1] * vv[g][1] + ld[2] * vv[g][2];
          sv[g] = rdlane(val[g], kSelfLane);
        }
        NM_BSTAMP(15);
      }
      if (si != cur[g]) sh.hcache[g] = si;
    }
  }
  NM_BSTAMP(12);
  int total = 0;   // contacts found so far (not capped)
  unsigned hitbits = 0;      // bit g: mesh g touches the floor
#pragma unroll
  for (int g = 0; g < kNCOL; g++) hitbits |= (uniform(near[g] && pz[g] - sv[g] < real(0)) ? 1u : 0u) << g;
  const int nhit = popc64(hitbits);
  // The standing / walking robot: several feet on the floor, each a lone contact at its support vertex (certified from the table, above).
  // Those go out together, mesh g on lane kBatchLane0 + g: one pass of per-lane frames instead of one pass per mesh.
  bool batch = false;
  if (nhit >= M.collide_batch_min) {
    const uint64_t hitm = (uint64_t)hitbits << kBatchLane0;
    vr X[3] = {vv[kNCOL - 1][0], vv[kNCOL - 1][1], vv[kNCOL - 1][2]}, svl = val[kNCOL - 1];
    V<int> slackl = nb[kNCOL - 1];
    V<int> lb = lane;
#pragma unroll
    for (int g = kNCOL - 2; g >= 0; g--) {
      order_after(lb, X[0]);      // one lane mask alive at a time (7 at once cost 14 spilled SGPRs)
      const VB mg = lb == kBatchLane0 + g;
      X[0] = sel(mg, vv[g][0], X[0]); X[1] = sel(mg, vv[g][1], X[1]); X[2] = sel(mg, vv[g][2], X[2]);
      svl = sel(mg, val[g], svl);
      slackl = sel(mg, nb[g], slackl);
    }
    const VB ml = lane >= kBatchLane0;
    const V<int> gl = sel(ml, lane - kBatchLane0, V<int>(0));
    const vr pzl = ldsv(sh.colp, gl * 3 + 2) + bz;
    const vr distl = pzl - svl;
    const VB wr = in_mask(hitm);
    const VB lone = distl * real(-0.5 * kSlackScale) < to_real<real>(slackl);
    if (!wany(wr & !lone)) {
      vr Rl[9], pl[2], pnt[3];
#pragma unroll
      for (int j = 0; j < 9; j++) Rl[j] = ldsv(sh.colR, gl * 9 + j);
#pragma unroll
      for (int j = 0; j < 2; j++) pl[j] = ldsv(sh.colp, gl * 3 + j);
      matvec3_fma(pnt, Rl, X);
      pnt[0] = pnt[0] + pl[0]; pnt[1] = pnt[1] + pl[1]; pnt[2] = pnt[2] + ldsv(sh.colp, gl * 3 + 2);
      const V<int> slot = lane_rank(hitm);
      const V<int> sl = sel(wr, slot, V<int>(0));
      stsv(sh.cpos(), sl * 3, pnt[0], wr);
      stsv(sh.cpos(), sl * 3 + 1, pnt[1], wr);
      stsv(sh.cpos(), sl * 3 + 2, pnt[2] - real(0.5) * distl, wr);
      stsv(sh.cdist(), sl, distl, wr);
      stsv(sh.cleg(), sl, gl - 1, wr);
      stsv(sh.cstart, gl, slot, ml);
      stsv(sh.ccnt, gl, sel(wr, V<int>(1), V<int>(0)), ml);
      total = nhit;
      batch = true;
    }
  }
  if (!batch) {
#pragma unroll
  for (int g = 0; g < kNCOL; g++) {
    const real* R = sh.colR + 9 * g;
    const real* p = sh.colp + 3 * g;
    const real dist = pz[g] - sv[g];
    const bool hitg = (hitbits >> g) & 1;
    int nextra = 0;
    if (hitg) {   // most meshes do not touch the floor in a given substep: their contact emission is skipped, not masked
      vr pnt[3];
      matvec3_fma(pnt, R, vv[g]);
      pnt[0] = pnt[0] + p[0]; pnt[1] = pnt[1] + p[1]; pnt[2] = pnt[2] + p[2];
      const real first[3] = {rdlane(pnt[0], kSelfLane), rdlane(pnt[1], kSelfLane), rdlane(pnt[2], kSelfLane) - real(0.5) * dist};
      // up to three more: penetrating hull neighbours of the support vertex, >= tolerance from the first contact
      const real tol = M.tol_planemesh * M.colc[kColN * g + 3];
      vr dd[3] = {pnt[0] - first[0], pnt[1] - first[1], pnt[2] - first[2]};
      vr d2 = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2];
      const VB ok = nbl & (nb[g] >= 0) & (val[g] > vr(pz[g])) & !(d2 < vr(tol * tol));     // |.| < tol without the square root
      const uint64_t m = ballot(ok);
      if (m == 0) {
        // The support vertex alone (every foot of a standing or walking robot: its penetrating hull neighbours lie within the
        // tolerance of it): its lane stores the contact, nothing is ranked.
        const VB wr = (lane == kSelfLane) & VB(total < kMaxConBig);
        const int sl = vmin(total, kMaxConBig - 1);
        stsv(sh.cpos(), V<int>(sl * 3), pnt[0], wr);
        stsv(sh.cpos(), V<int>(sl * 3 + 1), pnt[1], wr);
        stsv(sh.cpos(), V<int>(sl * 3 + 2), pnt[2] - real(0.5) * dist, wr);
        stsv(sh.cdist(), V<int>(sl), vr(dist), wr);
        stsv(sh.cleg(), V<int>(sl), V<int>(g - 1), wr);
      } else {
      nextra = vmin(popc64(m), 3);
      const V<int> rank = lane_rank(m);
      const VB self = lane == kSelfLane;
      const V<int> slot = sel(self, V<int>(total), rank + (total + 1));
      const VB wr = (self | (ok & (rank < 3))) & (slot < kMaxConBig);
      const V<int> sl = sel(wr, slot, V<int>(0));
      const vr cd = sel(self, vr(dist), pnt[2] + bz);
      stsu(sh.cpos(), sl * 3, pnt[0], wr, sh.sink);
      stsu(sh.cpos(), sl * 3 + 1, pnt[1], wr, sh.sink);
      stsu(sh.cpos(), sl * 3 + 2, pnt[2] - real(0.5) * cd, wr, sh.sink);
      stsu(sh.cdist(), sl, cd, wr, sh.sink);
      stsu(sh.cleg(), sl, V<int>(g - 1), wr, sh.sink);
      }
      // (a floor contact's normal (0,0,1) and "no first body" are not stored here: only the rare constraint paths read them, and
      // those fill them in first - floor_frames())
    }
    const int ng = hitg ? 1 + nextra : 0;
    const int c0 = vmin(total, kMaxConBig), c1 = vmin(total + ng, kMaxConBig);
    sh.cstart[g] = c0;
    sh.ccnt[g] = c1 - c0;
    total += ng;
  }
  }
  const int ncon = vmin(total, kMaxConBig);
  *dropped += total - ncon;
  sh.ncon = ncon;
  sh.cstart[7] = ncon;       // number of floor contacts (the tibia-pair contacts follow them)
  sh.anypair = 0;
  wave_sync();
  NM_BSTAMP(13);
  nm_stamp(3);
  if (pairs) stage_collide_pairs(sh, M, dropped);
  nm_stamp(4);
}


// =========================================================================================  stage C
// Runs f(0), f(1), ... f(ncon-1) with compile-time indices (the A matrix lives in registers, so row numbers must be constants)
// as NESTED ifs: the first unused contact leaves the whole chain with one taken branch, instead of one skipped test per
// remaining contact as a flat unrolled loop of `if (c < ncon)` would cost (a taken scalar branch is ~80 cycles here).
template <int C, int N, class F> NM_FN void for_contacts(int ncon, F&& f) {
  if constexpr (C < N) {
    if (C < ncon) {
      f(std::integral_constant<int, C>{});
      for_contacts<C + 1, N>(ncon, f);
    }
  }
}
// f(0) .. f(3) with compile-time indices (the four pyramid rows of a contact)
template <class F> NM_FN void sfor4(F&& f) {
  f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{}); f(std::integral_constant<int, 3>{});
}
// f(0), f(1), ... over the pyramid pairs of the first ncon contacts (two pairs per contact), compile-time indices, one uniform test per contact
template <int NP, int P = 0, class F> NM_FN void sfor_pairs(int ncon, F&& f) {
  if constexpr (P < NP) {
    if (P / 2 < ncon) {
      f(std::integral_constant<int, P>{});
      f(std::integral_constant<int, P + 1>{});
      sfor_pairs<NP, P + 2>(ncon, f);
    }
  }
}
// ---- The two update rules of the constraint solver, stated ONCE. The three row layouts below (row per lane on 64 lanes, two envs on
// half-waves, matrix-free rows in three slots per lane) differ in how a row's delta reaches the other rows' residuals - a v_readlane
// broadcast into a register-resident A row, two of them under the halves' masks, a block-factor solve through LDS - not in these
// formulas; a change to either rule is made here and nowhere else.
// mj_solPGS (engine_solver.c), one row of a pyramidal contact: residual res = (A f + b)_i + R_i f_i with the CURRENT forces, projected
// coordinate step f_i <- max(f_i - res / (A_ii + R_i), 0), i.e. delta = max(-(g + R f) / AR, -f) with g = (A f + b)_i. A row is updated once
// per sweep and its force only changes at its own step, so everything but g is constant over a sweep: k0 = -1 / AR, k1 = -(R f) / AR,
// nf = -f are prepared per sweep (round 5: the sweeps are VALU-ISSUE bound - scripts/micro/pgs_chain.hip: a row update costs its
// instruction count x 4.6 ticks whatever the dependences - so the rule is written for the fewest instructions per row: one fma + one max).
// A row that must not move any more (two-env pass: its half has met the tolerance) gets k0 = k1 = nf = 0: delta = max(0, 0) = 0.
template <class X> NM_FN void pgs_row_prepare(const X& Rr, const X& f, const X& ARinv, X& k0, X& k1, X& nf) {
  k0 = -ARinv;
  k1 = (Rr * f) * k0;
  nf = -f;
}
template <class X> NM_FN X pgs_row_delta(const X& g, const X& k0, const X& k1, const X& nf) { return vmax(vfma(g, k0, k1), nf); }
// ... and its cost change 0.5 dl^2 AR_ii + dl res (hA = 0.5 AR_ii; res = g + R f with the g the row saw at its own step): what mj_solPGS
// sums for the tolerance exit
template <class X> NM_FN X pgs_row_cost(const X& dl, const X& hA, const X& g, const X& Rr, const X& f) { return dl * vfma(hA, dl, vfma(Rr, f, g)); }
// mj_solNoSlip (engine_solver.c), one opposing pyramid pair seen from ONE of its two rows (f, g = own force and residual without R; fp, gp =
// the partner's): exact 1-D minimisation along (f - fp) - a Newton step on the difference dg = g - gp of the two residuals (K1 = A00 + A11 -
// 2 A01, invK1 = 1 / K1, hK1 = 0.5 K1), clamped to [lo, hi] = [-f, fp] so that both forces stay non-negative (their sum is kept); a
// degenerate pair (K1 < 1e-15) goes to its mean: lo = hi = 0.5 (fp - f). The bounds are constant over a sweep (prepared once per sweep; a
// pair that must not move gets lo = hi = 0). cost change of the pair = d (0.5 K1 d + dg); mj_solNoSlip reverts an update whose cost change
// exceeds 1e-10 (`noslip_pair_bad`): the caller drops d then.
template <class real, class X, class B> NM_FN void noslip_pair_prepare(const X& f, const X& fp, const B& small, X& lo, X& hi) {
  const X mean = real(0.5) * (fp - f);
  lo = sel(small, mean, -f);
  hi = sel(small, mean, fp);
}
template <class X> NM_FN X noslip_pair_delta(const X& dg, const X& invK1, const X& lo, const X& hi) { return vmin(vmax(-dg * invK1, lo), hi); }
template <class X> NM_FN X noslip_pair_cost(const X& d, const X& hK1, const X& dg) { return d * vfma(hK1, d, dg); }
template <class real, class X> NM_FN auto noslip_pair_bad(const X& change) { return change > X(real(1e-10)); }   // costChange: revert an update that does not decrease the cost

// Contact rows on lanes: build, project (A = J M^-1 J'), warm start, PGS, NoSlip, map back, sensors.
template <class real, bool PAIR> NM_FN void stage_constraint_body(Sh<real>& sh, real* jrow, const Model<real>& M, bool last, bool nosweep) {
  typedef V<real> vr;
  const V<int> lane = lane_id();
  const int ncon = uniform(sh.ncon), nefc = 4 * ncon;
  if (nefc == 0) {
#pragma unroll
    for (int j = 0; j < 24; j++) sh.qfc[j] = real(0);
    if (last)
#pragma unroll
      for (int j = 0; j < kNSENS; j++) sh.sens[j] = real(0);
    wave_sync();
    return;
  }
  const VB act = lane < nefc;
  const V<int> c = sel(act, lane >> 2, V<int>(0));
  const V<int> tk = (lane >> 1) & 1, sg = lane & 1;
  vr cp[3] = {ldsv(sh.cpos(), c * 3), ldsv(sh.cpos(), c * 3 + 1), ldsv(sh.cpos(), c * 3 + 2)};
  vr dist = ldsv(sh.cdist(), c);
  V<int> L = ldsv(sh.cleg(), c);
  const VB onleg = L >= 0;
  const V<int> Lc = vmax(L, V<int>(0));
  // tibia-tibia contacts (rare) carry a second leg (body1) and a general frame; floor-only envs skip all of that
  constexpr bool anypair = PAIR;  // floor-only envs (the common case) compile without the second-body terms
  V<int> L1 = V<int>(-1);
  if (anypair) L1 = ldsv(sh.cleg1(), c);
  const VB onleg1 = L1 >= 0;
  const V<int> Lc1 = vmax(L1, V<int>(0));
  // Contact frame (mju_makeFrame): floor contacts have n=(0,0,1), t1=(0,1,0), t2=(-1,0,0). Lane q of a contact's quad first
  // builds the Jacobian row of ONE frame axis (q: n, t1, t2, n) and publishes it for the projection sweep; its own pyramid
  // row is then J(n) +- mu J(t_k) (mj_instantiateContact's formula), fetched from the quad by DPP.
  const V<int> q4 = lane & 3;
  vr smu = sel(sg == 0, vr(M.mu), vr(-M.mu));
  vr nrm[3] = {vr(real(0)), vr(real(0)), vr(real(1))};
  vr e[3] = {sel(q4 == 2, vr(real(-1)), vr(real(0))), sel(q4 == 1, vr(real(1)), vr(real(0))), sel((q4 == 0) | (q4 == 3), vr(real(1)), vr(real(0)))};
  if (anypair) {
    nrm[0] = ldsv(sh.cnrm(), c * 3); nrm[1] = ldsv(sh.cnrm(), c * 3 + 1); nrm[2] = ldsv(sh.cnrm(), c * 3 + 2);
    VB usey = (nrm[1] < vr(real(0.5))) & (nrm[1] > vr(real(-0.5)));
    vr y0[3] = {vr(real(0)), sel(usey, vr(real(1)), vr(real(0))), sel(usey, vr(real(0)), vr(real(1)))};
    vr dt = nrm[1] * y0[1] + nrm[2] * y0[2];
    vr t1[3] = {y0[0] - dt * nrm[0], y0[1] - dt * nrm[1], y0[2] - dt * nrm[2]}, t2[3];
    vr il = vr(real(1)) / vsqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
    t1[0] = t1[0] * il; t1[1] = t1[1] * il; t1[2] = t1[2] * il;
    cross3(t2, nrm, t1);
#pragma unroll
    for (int j = 0; j < 3; j++) e[j] = sel(q4 == 1, t1[j], sel(q4 == 2, t2[j], nrm[j]));
  }
  // frame-axis Jacobian row: base translation, base rotation (body axes), the 3 hinges of the contact's own leg
  vr Fb[6], Fl[3], Fm[3] = {vr(real(0)), vr(real(0)), vr(real(0))};
  {
    vr m[3];
    cross3(m, cp, e);  // (p - O) x e
    Fb[0] = e[0]; Fb[1] = e[1]; Fb[2] = e[2];
#pragma unroll
    for (int j = 0; j < 3; j++) Fb[3 + j] = sh.Rb[j] * m[0] + sh.Rb[3 + j] * m[1] + sh.Rb[6 + j] * m[2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      vr a[3], r[3], rel[3], mm[3];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        a[j] = ldsv(sh.axs, Lc * 9 + (3 * k + j));
        r[j] = ldsv(sh.anc, Lc * 9 + (3 * k + j));
        rel[j] = cp[j] - r[j];
      }
      cross3(mm, rel, e);
      Fl[k] = sel(onleg, dot3<vr>(a, mm), vr(real(0)));
    }
  }
  if (anypair) {  // body1 side: -J(body1); the base columns of J(b2) - J(b1) cancel exactly
#pragma unroll
    for (int k = 0; k < 3; k++) {
      vr a[3], r[3], rel[3], mm[3];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        a[j] = ldsv(sh.axs, Lc1 * 9 + (3 * k + j));
        r[j] = ldsv(sh.anc, Lc1 * 9 + (3 * k + j));
        rel[j] = cp[j] - r[j];
      }
      cross3(mm, rel, e);
      Fm[k] = sel(onleg1 & act, -dot3<vr>(a, mm), vr(real(0)));
    }
#pragma unroll
    for (int j = 0; j < 6; j++) Fb[j] = sel(onleg1, vr(real(0)), Fb[j]);
  }
#pragma unroll
  for (int j = 0; j < 6; j++) Fb[j] = sel(act, Fb[j], vr(real(0)));
#pragma unroll
  for (int k = 0; k < 3; k++) Fl[k] = sel(act, Fl[k], vr(real(0)));
  // A = J M^-1 J' through the block factor: with u_i = Jb_i - W_l' Jl_i (l = row i's leg; the base part of the row after
  // eliminating its leg), xb_j = S^-1 u_j (Schur solve) and t_j = M_l^-1 Jl_j,   A_ij = u_i . xb_j + [l_i == l_j] Jl_i . t_j.
  // So a row publishes (u, Jl) and a lane keeps (xb, t): no 24-vector B per lane, no reads of the other legs' W blocks.
  vr uF[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    uF[j] = Fb[j];
#pragma unroll
    for (int k = 0; k < 3; k++) uF[j] = uF[j] - ldsv(sh.W, Lc * 18 + (6 * k + j)) * Fl[k];
  }
  if (anypair) {
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
      for (int k = 0; k < 3; k++) uF[j] = uF[j] - ldsv(sh.W, Lc1 * 18 + (6 * k + j)) * Fm[k];
  }
  // publish the frame rows (rows 4c, 4c+1, 4c+2 of the buffer = n, t1, t2 of contact c) for the projection sweep
#pragma unroll
  for (int j = 0; j < 6; j++) stsv(jrow, lane * kJRow + j, uF[j], lane < kMaxRow);
#pragma unroll
  for (int k = 0; k < 3; k++) stsv(jrow, lane * kJRow + (6 + k), Fl[k], lane < kMaxRow);
  if (anypair) {
#pragma unroll
    for (int k = 0; k < 3; k++) stsv(jrow, lane * kJRow + (10 + k), Fm[k], lane < kMaxRow);
  }
  // own pyramid row J = J(n) + smu J(t_k): quad lane 0 holds n, lanes 1 / 2 hold t1 / t2
  vr Jb[6], Jl[3], Jm[3] = {vr(real(0)), vr(real(0)), vr(real(0))};
#pragma unroll
  for (int j = 0; j < 6; j++) Jb[j] = quad<0x00>(Fb[j]) + smu * quad<0xA5>(Fb[j]);
#pragma unroll
  for (int k = 0; k < 3; k++) Jl[k] = quad<0x00>(Fl[k]) + smu * quad<0xA5>(Fl[k]);
  if (anypair) {
#pragma unroll
    for (int k = 0; k < 3; k++) Jm[k] = quad<0x00>(Fm[k]) + smu * quad<0xA5>(Fm[k]);
  }
  vr u[6];
#pragma unroll
  for (int j = 0; j < 6; j++) u[j] = quad<0x00>(uF[j]) + smu * quad<0xA5>(uF[j]);

  // impedance, regulariser, reference acceleration (mj_makeImpedance / mj_referenceConstraint)
  vr imp;
  {
    vr x = vabs(dist) * vrcp(M.si_width);
    vr ylo, yhi;
    if (M.si_power == real(2)) {  // the model's solimp (mjmodel.xml defaults): pow(x, 2) = x*x and pow(mid, 1) = mid, both exact
      vr omx = vmax(vr(real(1)) - x, vr(real(0)));
      ylo = (x * x) * vrcp(M.si_mid);
      yhi = vr(real(1)) - (omx * omx) * vrcp(real(1) - M.si_mid);
    } else {
      ylo = vpow(x, vr(M.si_power)) / vpow(vr(M.si_mid), vr(M.si_power - real(1)));
      yhi = vr(real(1)) - vpow(vmax(vr(real(1)) - x, vr(real(0))), vr(M.si_power)) / vpow(vr(real(1) - M.si_mid), vr(M.si_power - real(1)));
    }
    vr y = sel(x <= vr(M.si_mid), ylo, yhi);
    imp = M.si_d0 + y * (M.si_dmax - M.si_d0);
    imp = sel(x >= vr(real(1)), vr(M.si_dmax), imp);
    imp = sel(x <= vr(real(0)), vr(M.si_d0), imp);
  }
  vr invw = ldsv(M.colc, (Lc + sel(onleg, V<int>(1), V<int>(0))) * kColN + 4);
  if (anypair) invw = invw + sel(onleg1, ldsv(M.colc, (Lc1 + 1) * kColN + 4), vr(real(0)));
  vr Rr = vmax((vr(real(1)) - imp) * (invw + M.mu * M.mu * invw) * vrcp(imp), vr(real(1e-15))) * (real(2) * M.mu * M.mu);
  vr Dd = vrcp(Rr);
  // sparse dots with the wave-uniform vectors qvel, qacc_smooth, qacc_warmstart
  vr vel = vr(real(0)), jas = vr(real(0)), jaw = vr(real(0));
#pragma unroll
  for (int j = 0; j < 6; j++) {
    vel += Jb[j] * sh.qvel[j]; jas += Jb[j] * sh.qas[j]; jaw += Jb[j] * sh.warm[j];
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    V<int> di = Lc * 3 + (6 + k);
    vel += Jl[k] * ldsv(sh.qvel, di); jas += Jl[k] * ldsv(sh.qas, di); jaw += Jl[k] * ldsv(sh.warm, di);
  }
  if (anypair) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      V<int> di = Lc1 * 3 + (6 + k);
      vel += Jm[k] * ldsv(sh.qvel, di); jas += Jm[k] * ldsv(sh.qas, di); jaw += Jm[k] * ldsv(sh.warm, di);
    }
  }
  vr aref = -M.solref_B * vel - M.solref_K * imp * dist;
  vr bb = jas - aref;

  // this lane's side of the projection: t = M_l^-1 Jl (own leg), xb = S^-1 u
  vr t[3], t1m[3] = {vr(real(0)), vr(real(0)), vr(real(0))}, xb[6];
  {
    vr Mi[6];
#pragma unroll
    for (int j = 0; j < 6; j++) Mi[j] = ldsv(sh.Minv, Lc * 6 + j);
    ldl3_solve(t, Mi, Jl);
    if (anypair) {
      vr Mi1[6];
#pragma unroll
      for (int j = 0; j < 6; j++) Mi1[j] = ldsv(sh.Minv, Lc1 * 6 + j);
      ldl3_solve(t1m, Mi1, Jm);
    }
#pragma unroll
    for (int j = 0; j < 6; j++) xb[j] = u[j];
    ldl6_solve(sh.Lb, sh.Dbi, xb);
  }
  wave_sync();
  // A[i] (in lane j) -> lane j holds row j of the symmetric A = J M^-1 J'. Per contact: the three frame rows are read once
  // (wave-uniform LDS reads), contracted with this lane's (xb, t), and combined into the four pyramid rows (n +- mu t1,
  // n +- mu t2); no per-row branches.
  vr A[kMaxRow];
#pragma unroll
  for (int i = 0; i < kMaxRow; i++) A[i] = vr(real(0));
  auto build_contact = [&](auto ccT) {
    constexpr int cc = decltype(ccT)::value;
    if (cc < ncon) {
      const int Lcc = uniform(sh.cleg()[cc]);
      const VB same = onleg & (L == Lcc);
      VB s01 = VB(false), s10 = VB(false), s11 = VB(false);
      if (anypair) {
        const int L1cc = uniform(sh.cleg1()[cc]);
        s01 = onleg1 & (L1 == Lcc);      // row's body2 leg meets this lane's body1 leg
        s10 = onleg & (L == L1cc);       // row's body1 leg meets this lane's body2 leg
        s11 = onleg1 & (L1 == L1cc);
      }
      vr a3[3];
  #pragma unroll
      for (int r = 0; r < 3; r++) {
        const real* jr = jrow + (4 * cc + r) * kJRow;
        vr a = jr[0] * xb[0] + jr[1] * xb[1] + jr[2] * xb[2] + jr[3] * xb[3] + jr[4] * xb[4] + jr[5] * xb[5];
        a += sel(same, jr[6] * t[0] + jr[7] * t[1] + jr[8] * t[2], vr(real(0)));
        if (anypair) {
          a += sel(s01, jr[6] * t1m[0] + jr[7] * t1m[1] + jr[8] * t1m[2], vr(real(0)));
          a += sel(s10, jr[10] * t[0] + jr[11] * t[1] + jr[12] * t[2], vr(real(0)));
          a += sel(s11, jr[10] * t1m[0] + jr[11] * t1m[1] + jr[12] * t1m[2], vr(real(0)));
        }
        a3[r] = a;
      }
      A[4 * cc] = a3[0] + M.mu * a3[1];
      A[4 * cc + 1] = a3[0] - M.mu * a3[1];
      A[4 * cc + 2] = a3[0] + M.mu * a3[2];
      A[4 * cc + 3] = a3[0] - M.mu * a3[2];
    }
  };
  // a flat chain of 16 guarded bodies: nesting them (one exit branch instead of one skipped test per unused contact) makes the
  // register allocator spill 20-300 VGPRs and measured no faster
#define NM_BUILD4(c0) build_contact(std::integral_constant<int, c0>{}); build_contact(std::integral_constant<int, c0 + 1>{}); \
                      build_contact(std::integral_constant<int, c0 + 2>{}); build_contact(std::integral_constant<int, c0 + 3>{});
  NM_BUILD4(0) NM_BUILD4(4) NM_BUILD4(8) NM_BUILD4(12)
#undef NM_BUILD4
  // own diagonal entry (needed as 1/AR_ii)
  vr Ajj = u[0] * xb[0] + u[1] * xb[1] + u[2] * xb[2] + u[3] * xb[3] + u[4] * xb[4] + u[5] * xb[5] + (Jl[0] * t[0] + Jl[1] * t[1] + Jl[2] * t[2]);
  if (anypair) Ajj += Jm[0] * t1m[0] + Jm[1] * t1m[1] + Jm[2] * t1m[2];   // the two bodies of a pair are different legs
  vr ARjj = Ajj + Rr;
  vr ARinv = sel(act, vrcp(ARjj), vr(real(0)));

  nm_stamp(5);
  // ---- warm start (PGS branch of mj_fwdConstraint): f from qacc_warmstart, kept only if its dual cost < 0
  vr jar = jaw - aref;
  vr f = sel(act & (jar < vr(real(0))), -Dd * jar, vr(real(0)));
  vr g = bb;  // g_j = (A f)_j + b_j  (residual without the R term)
  for_contacts<0, kMaxCon>(ncon, [&](auto ccT) {
    constexpr int cc = decltype(ccT)::value;
#pragma unroll
    for (int r = 0; r < 4; r++) g += A[4 * cc + r] * rdlane(f, 4 * cc + r);
  });
  {
    real cost = wsum<real>(sel(act, f * (bb + real(0.5) * (g - bb + Rr * f)), vr(real(0))));
    if (cost > real(0)) { f = vr(real(0)); g = bb; }
  }
#ifdef NM_DEBUG_SOLVER
  stsv(sh.dbg_b, lane, bb, lane < kMaxRow); stsv(sh.dbg_a, lane, ARjj, lane < kMaxRow); stsv(sh.dbg_f0, lane, f, lane < kMaxRow);
#endif
  // ---- mj_solPGS: Gauss-Seidel over rows; every lane keeps its residual current by a rank-1 update.
  // Per row: all lanes evaluate their own candidate, lane i's delta is broadcast (v_readlane) and applied.
  const vr hA = real(0.5) * ARjj;
  for (int iter = 0; iter < (nosweep ? 0 : M.pgs_iters); iter++) {
    // A row's update is decided by its own lane at its own step: the lane keeps the residual it saw there (one select under a literal lane
    // mask) and forms its delta and cost change from it after the sweep; the other lanes only need the delta, which reaches them through g.
    vr k0, k1, nf;
    pgs_row_prepare(Rr, f, ARinv, k0, k1, nf);
    vr gcap = g;
    uint64_t rowm = mask_shl(1ull, 0);                        // lane mask of the row whose turn it is: shifted along with the sweep
    for_contacts<0, kMaxCon>(ncon, [&](auto ccT) {
      constexpr int cc = decltype(ccT)::value;
      sfor4([&](auto rT) {
        constexpr int i = 4 * cc + decltype(rT)::value;
        gcap = sel_mask(rowm, g, gcap);
        rowm = mask_shl(rowm, 1);
        const vr dl = pgs_row_delta(g, k0, k1, nf);
        g = vfma(A[i], vr(rdlane(dl, i)), g);
      });
    });
    // cost change of a row, 0.5 dl^2 AR_ii + dl res, formed once after the sweep from what the lane kept. mj_solPGS reverts an update
    // whose cost change exceeds +1e-10; for this projected coordinate step that cannot happen: unclamped, 0.5 AR dl + res =
    // res (1 - 0.5 AR/AR~) has the sign of res = -sign(dl); clamped at zero, res >= AR f makes it -f (res - 0.5 AR f) <= 0 - no
    // cancellation in either case, so no test.
    const vr dcap = pgs_row_delta(gcap, k0, k1, nf);
    const vr ccap = pgs_row_cost(dcap, hA, gcap, Rr, f);
    f = f + dcap;
    sh.it_pgs = iter + 1;
    if (-wsum<real>(ccap) * M.pgs_scale < M.pgs_tol) break;
  }
  nm_stamp(6);
  // ---- mj_solNoSlip: per opposing pyramid pair (lanes 2p, 2p+1), exact 1-D minimisation along (f0 - f1) without R.
  // mj_solNoSlip writes the pair's quadratic as 0.5 K1 y^2 + K0 y around mid = (f0 + f1)/2 with K1 = A00 + A11 - 2 A01 and
  // K0 = mid (A00 - A11) + bc0 - bc1 (bc = residual without the pair's own forces). Substituting bc gives
  // K0 = (g0 - g1) - 0.5 K1 (f0 - f1), so the new value of a lane is f - (g_mine - g_partner)/K1: a Newton step on the
  // difference of the two residuals. The sum f0 + f1 is kept, hence d_partner = -d, the three-way case split of the
  // reference is the clamp d in [-f_mine, f_partner], and the cost change is d (0.5 K1 d + (g_mine - g_partner)). Each lane
  // evaluates its side with the partner's (f, g) from DPP; 1/K1 and 0.5 K1 are prepared once.
  {
    const V<int> lv = opaque_lane();
    const VB even = (lv & 1) == 0;
    vr Amq = vr(real(0));  // A[2p][2p+1], the even lane's copy in both lanes
    for_contacts<0, kMaxCon>(ncon, [&](auto ccT) {
      constexpr int cc = decltype(ccT)::value;
      Amq = sel(lv == 4 * cc, A[4 * cc + 1], Amq);
      Amq = sel(lv == 4 * cc + 2, A[4 * cc + 3], Amq);
    });
    Amq = sel(even, Amq, shfl_xor1(Amq));
    const vr K1 = Ajj + shfl_xor1(Ajj) - Amq - Amq;
    const VB small = K1 < vr(real(1e-15));       // degenerate pair: both forces go to their mean
    const vr invK1 = vrcp(K1), hK1 = real(0.5) * K1;
    // d_partner = -d exactly (the subtraction, the clamp and the halved difference are antisymmetric in the pair), so a pair moves every
    // residual by (A[.][2p] - A[.][2p+1]) d(2p): one broadcast and one fma per pair. The sweeps of mj_solPGS are over: the column
    // difference takes the even column's registers.
    for_contacts<0, kMaxCon>(ncon, [&](auto ccT) {
      constexpr int cc = decltype(ccT)::value;
      A[4 * cc] = A[4 * cc] - A[4 * cc + 1];
      A[4 * cc + 2] = A[4 * cc + 2] - A[4 * cc + 3];
    });
    for (int iter = 0; iter < (nosweep ? 0 : M.noslip_iters); iter++) {
      real improvement = real(0);
      if (iter == 0) improvement = wsum<real>(sel(act, real(0.5) * f * f * Rr, vr(real(0))));
      vr lo, hi;
      noslip_pair_prepare<real>(f, shfl_xor1(f), small, lo, hi);
      vr dcap = vr(real(0)), dgcap = vr(real(0));      // a pair keeps its step and the residual difference it saw: the cost change is formed after the sweep
      uint64_t pairm = mask_shl(3ull, 0);
      sfor_pairs<kMaxRow / 2>(ncon, [&](auto pT) {
        constexpr int p = decltype(pT)::value;
        const vr dg = g - shfl_xor1(g);
        vr d = noslip_pair_delta(dg, invK1, lo, hi);
        d = sel(noslip_pair_bad<real>(noslip_pair_cost(d, hK1, dg)), vr(real(0)), d);
        g = vfma(A[2 * p], vr(rdlane(d, 2 * p)), g);
        dcap = sel_mask(pairm, d, dcap);
        dgcap = sel_mask(pairm, dg, dgcap);
        pairm = mask_shl(pairm, 2);
      });
      f = f + dcap;
      improvement = improvement - wsum<real>(sel(even, noslip_pair_cost(dcap, hK1, dgcap), vr(real(0))));     // a reverted pair kept d = 0: no change
      sh.it_noslip = iter + 1;
      if (improvement * M.pgs_scale < M.noslip_tol) break;
    }
  }
  nm_stamp(7);
  stsv(sh.efc_f, lane, f, lane < kMaxRow);
  // ---- qfrc_constraint = J' f and the touch sensors (only the last forward pass is observable after mj_step)
  vr nf = vr(real(0));
  VB hit = VB(false), hit1 = VB(false);
  if (last) {
    nf = f + shfl_xor1(f);
    nf = nf + shfl_xor2(nf);  // contact normal force = sum of its 4 pyramid edge forces
    // foot site sphere (mjmodel.xml:49...): the ray from the contact point along -normal (sensor on body2) or +normal
    // (sensor on body1) must hit it (mju_rayGeom, sphere)
    auto foot_hit = [&](const V<int>& Lx, real sgn) {
      vr ft[3], fr, s[3], Rt[9];
#pragma unroll
      for (int j = 0; j < 3; j++) s[j] = ldsv(M.footc, Lx * 4 + j);
      fr = ldsv(M.footc, Lx * 4 + 3);
#pragma unroll
      for (int j = 0; j < 9; j++) Rt[j] = ldsv(sh.colR, (Lx + 1) * 9 + j);
      matvec3(ft, Rt, s);
#pragma unroll
      for (int j = 0; j < 3; j++) ft[j] = ft[j] + ldsv(sh.colp, (Lx + 1) * 3 + j);
      vr dif[3] = {cp[0] - ft[0], cp[1] - ft[1], cp[2] - ft[2]};
      vr b2 = sgn * (nrm[0] * dif[0] + nrm[1] * dif[1] + nrm[2] * dif[2]);
      vr cc = dif[0] * dif[0] + dif[1] * dif[1] + dif[2] * dif[2] - fr * fr;
      vr det = b2 * b2 - cc;
      vr sq = vsqrt(vmax(det, vr(real(0))));
      return !(det < vr(real(1e-15))) & (((-b2 - sq) >= vr(real(0))) | ((-b2 + sq) >= vr(real(0))));
    };
    hit = foot_hit(Lc, real(-1));
    if (anypair) hit1 = foot_hit(Lc1, real(1));
  }
#pragma unroll
  for (int j = 0; j < 6; j++) sh.qfc[j] = wsum<real>(Jb[j] * f);
  if (!anypair) {
    // Floor-only env: the contacts of a leg are consecutive (stage B emits them mesh by mesh). Sum inside each contact's
    // quad, park the per-contact sums in the row buffer, then lane (leg l, quantity q) adds up its leg's <= 4 contacts.
    vr s3[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      vr x = Jl[k] * f;
      x = x + shfl_xor1(x);
      s3[k] = x + shfl_xor2(x);
    }
    const vr tn = sel(act & (nf > vr(real(0))), nf, vr(real(0))), tf = sel(hit, tn, vr(real(0)));
    const VB lead = act & ((lane & 3) == 0);
    wave_sync();
#pragma unroll
    for (int k = 0; k < 3; k++) stsv(jrow, c * 8 + k, s3[k], lead);
    if (last) { stsv(jrow, c * 8 + 3, tn, lead); stsv(jrow, c * 8 + 4, tf, lead); }
    wave_sync();
    const VB on = lane < 30;
    const V<int> lg = sel(on, (lane * 13) >> 6, V<int>(0)), qq = lane - lg * 5;
    const V<int> c0 = ldsv(sh.cstart, lg + 1), cn = ldsv(sh.ccnt, lg + 1);
    vr acc = vr(real(0));
#pragma unroll
    for (int k = 0; k < 4; k++) {
      VB in = on & (cn > k);
      acc = acc + sel(in, ldsv(jrow, sel(in, (c0 + k) * 8 + qq, V<int>(0))), vr(real(0)));
    }
    stsv(sh.qfc, 6 + lg * 3 + qq, acc, on & (qq < 3));
    if (last) {
      stsv(sh.sens, lg, acc, on & (qq == 3));          // tibia sites: 10 m spheres see every contact of the body
      stsv(sh.sens, lg + 6, acc, on & (qq == 4));      // foot sites
      sh.sens[12] = wsum<real>(sel(act & ((lane & 3) == 0) & (L < 0), tn, vr(real(0))));
    }
  } else {
#pragma unroll
    for (int l = 0; l < 6; l++)
#pragma unroll
      for (int k = 0; k < 3; k++) sh.qfc[6 + 3 * l + k] = wsum<real>(sel(L == l, Jl[k] * f, vr(real(0))) + sel(L1 == l, Jm[k] * f, vr(real(0))));
    if (last) {
      VB head = act & ((lane & 3) == 0) & (nf > vr(real(0)));
#pragma unroll
      for (int l = 0; l < 6; l++) {
        VB onl = (L == l) | (L1 == l);
        sh.sens[l] = wsum<real>(sel(head & onl, nf, vr(real(0))));
        sh.sens[6 + l] = wsum<real>(sel(head & (((L == l) & hit) | ((L1 == l) & hit1)), nf, vr(real(0))));
      }
      sh.sens[12] = wsum<real>(sel(head & (L < 0), nf, vr(real(0))));
    }
  }
  wave_sync();
}

template <class real> NM_COLD void stage_constraint_pairs(Sh<real>& sh, real* jrow, const Model<real>& M, bool last, bool nosweep) {
  stage_constraint_body<real, true>(sh, jrow, M, last, nosweep);
}

// =========================================================================================  stage C, two envs at once
// Floor-only envs with 1..kMaxCon2 contacts each (the common case once a robot stands): the wave's two envs go through the
// constraint stage TOGETHER, lanes 0..31 = rows of env 0, lanes 32..63 = rows of env 1. Same arithmetic in the same order as
// stage_constraint_body<real, false> per env (half-wave sums keep wsum's association), so the results are identical; what changes is
// the cost: one pass whose length is max(ncon0, ncon1) instead of two passes of ncon0 + ncon1. Per-env scalars become per-lane
// values (equal inside a half), LDS addresses carry the half's image offset, a row's broadcast is two v_readlane + a select, the
// solvers' early exits become per-half run masks. The A matrix is 32 VGPRs here.
constexpr int kMaxCon2 = 8, kMaxRow2 = 4 * kMaxCon2;
#ifdef NM_EMUL
inline long& nm_emul_together() { static long n = 0; return n; }   // host emulation only: how often the two-env pass ran (tests assert it did)
#endif
template <class real> NM_FN void stage_constraint2(ShW<real, 2>& w, const Model<real>& M, bool last, bool nosweep) {
  typedef V<real> vr;
  constexpr int kSR = (int)(sizeof(Sh<real>) / sizeof(real)), kSI = (int)(sizeof(Sh<real>) / sizeof(int));
  const V<int> lane = lane_id();
  const V<int> h = lane >> 5, hl = lane & 31;
  const VB h1 = h != 0;
  real* rbw = reinterpret_cast<real*>(&w.e[0]);
  int* ibw = reinterpret_cast<int*>(&w.e[0]);
  const real* rb = rbw;
  const int* ib = ibw;
  real* jrow = w.jrow;
  const V<int> ho = h * kSR, hoi = h * kSI, jh = h * (kMaxRow2 * kJRow);
  const int oLeg = NM_OFS(legtmp);
  const int oCleg = (int)((offsetof(Sh<real>, legtmp) + 7 * kMaxConBig * sizeof(real)) / sizeof(int));
#define SHR(field, idx) ldsv(rb, ho + ((idx) + NM_OFS(field)))
#define SHI(field, idx) ldsv(ib, hoi + ((idx) + NM_IOFS(field)))
  const int n0 = uniform(w.e[0].ncon), n1 = uniform(w.e[1].ncon), nmax = vmax(n0, n1);
  const V<int> nconv = sel(h1, V<int>(n1), V<int>(n0)), nefc = nconv * 4;
  const VB act = hl < nefc;
  const V<int> c = sel(act, hl >> 2, V<int>(0));
  const V<int> sg = hl & 1;
  vr cp[3] = {ldsv(rb, ho + (c * 3 + oLeg)), ldsv(rb, ho + (c * 3 + (oLeg + 1))), ldsv(rb, ho + (c * 3 + (oLeg + 2)))};
  vr dist = ldsv(rb, ho + (c + (oLeg + 6 * kMaxConBig)));
  V<int> L = ldsv(ib, hoi + (c + oCleg));
  {  // a half whose env has no contact at all (the pass then runs for the other env alone) reads a stale record as contact 0: every
     // lane of it is inactive, but what it computes must stay finite (0 * NaN would switch off that half's solver exits)
    const VB any = nconv > 0;
    cp[0] = sel(any, cp[0], vr(real(0))); cp[1] = sel(any, cp[1], vr(real(0))); cp[2] = sel(any, cp[2], vr(real(0)));
    dist = sel(any, dist, vr(real(0)));
    L = sel(any, L, V<int>(-1));
  }
  const VB onleg = L >= 0;
  const V<int> Lc = vmax(L, V<int>(0));
  const V<int> q4 = hl & 3;
  vr smu = sel(sg == 0, vr(M.mu), vr(-M.mu));
  vr nrm[3] = {vr(real(0)), vr(real(0)), vr(real(1))};
  vr e[3] = {sel(q4 == 2, vr(real(-1)), vr(real(0))), sel(q4 == 1, vr(real(1)), vr(real(0))), sel((q4 == 0) | (q4 == 3), vr(real(1)), vr(real(0)))};
  // frame-axis Jacobian row: base translation, base rotation (body axes), the 3 hinges of the contact's own leg
  vr Fb[6], Fl[3];
  {
    vr m[3];
    cross3(m, cp, e);  // (p - O) x e
    Fb[0] = e[0]; Fb[1] = e[1]; Fb[2] = e[2];
#pragma unroll
    for (int j = 0; j < 3; j++) Fb[3 + j] = SHR(Rb, V<int>(j)) * m[0] + SHR(Rb, V<int>(3 + j)) * m[1] + SHR(Rb, V<int>(6 + j)) * m[2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      vr a[3], r[3], rel[3], mm[3];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        a[j] = SHR(axs, Lc * 9 + (3 * k + j));
        r[j] = SHR(anc, Lc * 9 + (3 * k + j));
        rel[j] = cp[j] - r[j];
      }
      cross3(mm, rel, e);
      Fl[k] = sel(onleg, dot3<vr>(a, mm), vr(real(0)));
    }
  }
#pragma unroll
  for (int j = 0; j < 6; j++) Fb[j] = sel(act, Fb[j], vr(real(0)));
#pragma unroll
  for (int k = 0; k < 3; k++) Fl[k] = sel(act, Fl[k], vr(real(0)));
  vr uF[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    uF[j] = Fb[j];
#pragma unroll
    for (int k = 0; k < 3; k++) uF[j] = uF[j] - SHR(W, Lc * 18 + (6 * k + j)) * Fl[k];
  }
  // publish the frame rows (rows 4c, 4c+1, 4c+2 of the half's part of the buffer = n, t1, t2 of contact c)
#pragma unroll
  for (int j = 0; j < 6; j++) stsv(jrow, lane * kJRow + j, uF[j], VB(true));
#pragma unroll
  for (int k = 0; k < 3; k++) stsv(jrow, lane * kJRow + (6 + k), Fl[k], VB(true));
  // own pyramid row J = J(n) + smu J(t_k): quad lane 0 holds n, lanes 1 / 2 hold t1 / t2
  vr Jb[6], Jl[3];
#pragma unroll
  for (int j = 0; j < 6; j++) Jb[j] = quad<0x00>(Fb[j]) + smu * quad<0xA5>(Fb[j]);
#pragma unroll
  for (int k = 0; k < 3; k++) Jl[k] = quad<0x00>(Fl[k]) + smu * quad<0xA5>(Fl[k]);
  vr u[6];
#pragma unroll
  for (int j = 0; j < 6; j++) u[j] = quad<0x00>(uF[j]) + smu * quad<0xA5>(uF[j]);

  // impedance, regulariser, reference acceleration (mj_makeImpedance / mj_referenceConstraint)
  vr imp;
  {
    vr x = vabs(dist) * vrcp(M.si_width);
    vr ylo, yhi;
    if (M.si_power == real(2)) {
      vr omx = vmax(vr(real(1)) - x, vr(real(0)));
      ylo = (x * x) * vrcp(M.si_mid);
      yhi = vr(real(1)) - (omx * omx) * vrcp(real(1) - M.si_mid);
    } else {
      ylo = vpow(x, vr(M.si_power)) / vpow(vr(M.si_mid), vr(M.si_power - real(1)));
      yhi = vr(real(1)) - vpow(vmax(vr(real(1)) - x, vr(real(0))), vr(M.si_power)) / vpow(vr(real(1) - M.si_mid), vr(M.si_power - real(1)));
    }
    vr y = sel(x <= vr(M.si_mid), ylo, yhi);
    imp = M.si_d0 + y * (M.si_dmax - M.si_d0);
    imp = sel(x >= vr(real(1)), vr(M.si_dmax), imp);
    imp = sel(x <= vr(real(0)), vr(M.si_d0), imp);
  }
  vr invw = ldsv(M.colc, (Lc + sel(onleg, V<int>(1), V<int>(0))) * kColN + 4);
  vr Rr = vmax((vr(real(1)) - imp) * (invw + M.mu * M.mu * invw) * vrcp(imp), vr(real(1e-15))) * (real(2) * M.mu * M.mu);
  vr Dd = vrcp(Rr);
  vr vel = vr(real(0)), jas = vr(real(0)), jaw = vr(real(0));
#pragma unroll
  for (int j = 0; j < 6; j++) {
    vel += Jb[j] * SHR(qvel, V<int>(j)); jas += Jb[j] * SHR(qas, V<int>(j)); jaw += Jb[j] * SHR(warm, V<int>(j));
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    V<int> di = Lc * 3 + (6 + k);
    vel += Jl[k] * SHR(qvel, di); jas += Jl[k] * SHR(qas, di); jaw += Jl[k] * SHR(warm, di);
  }
  vr aref = -M.solref_B * vel - M.solref_K * imp * dist;
  vr bb = jas - aref;

  // this lane's side of the projection: t = M_l^-1 Jl (own leg), xb = S^-1 u
  vr t[3], xb[6];
  {
    vr Mi[6];
#pragma unroll
    for (int j = 0; j < 6; j++) Mi[j] = SHR(Minv, Lc * 6 + j);
    ldl3_solve(t, Mi, Jl);
    vr Lbv[15], Dbv[6];
#pragma unroll
    for (int j = 0; j < 15; j++) Lbv[j] = SHR(Lb, V<int>(j));
#pragma unroll
    for (int j = 0; j < 6; j++) Dbv[j] = SHR(Dbi, V<int>(j));
#pragma unroll
    for (int j = 0; j < 6; j++) xb[j] = u[j];
    ldl6_solve(Lbv, Dbv, xb);
  }
  wave_sync();
  vr A[kMaxRow2];
#pragma unroll
  for (int i = 0; i < kMaxRow2; i++) A[i] = vr(real(0));
  auto build_contact = [&](auto ccT) {
    constexpr int cc = decltype(ccT)::value;
    if (cc < nmax) {
      const V<int> Lcc = ldsv(ib, hoi + (cc + oCleg));
      const VB same = onleg & (L == Lcc);
      const VB has = V<int>(cc) < nconv;            // this half's env has contact cc
      vr a3[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const V<int> jr = jh + (4 * cc + r) * kJRow;
        vr a = ldsv(jrow, jr) * xb[0] + ldsv(jrow, jr + 1) * xb[1] + ldsv(jrow, jr + 2) * xb[2] + ldsv(jrow, jr + 3) * xb[3] + ldsv(jrow, jr + 4) * xb[4] +
               ldsv(jrow, jr + 5) * xb[5];
        a += sel(same, ldsv(jrow, jr + 6) * t[0] + ldsv(jrow, jr + 7) * t[1] + ldsv(jrow, jr + 8) * t[2], vr(real(0)));
        a3[r] = a;
      }
      A[4 * cc] = sel(has, a3[0] + M.mu * a3[1], vr(real(0)));
      A[4 * cc + 1] = sel(has, a3[0] - M.mu * a3[1], vr(real(0)));
      A[4 * cc + 2] = sel(has, a3[0] + M.mu * a3[2], vr(real(0)));
      A[4 * cc + 3] = sel(has, a3[0] - M.mu * a3[2], vr(real(0)));
    }
  };
#define NM_BUILD4(c0) build_contact(std::integral_constant<int, c0>{}); build_contact(std::integral_constant<int, c0 + 1>{}); \
                      build_contact(std::integral_constant<int, c0 + 2>{}); build_contact(std::integral_constant<int, c0 + 3>{});
  NM_BUILD4(0) NM_BUILD4(4)
#undef NM_BUILD4
  vr Ajj = u[0] * xb[0] + u[1] * xb[1] + u[2] * xb[2] + u[3] * xb[3] + u[4] * xb[4] + u[5] * xb[5] + (Jl[0] * t[0] + Jl[1] * t[1] + Jl[2] * t[2]);
  vr ARjj = Ajj + Rr;
  vr ARinv = sel(act, vrcp(ARjj), vr(real(0)));

  nm_stamp(5);
  // ---- warm start
  vr jar = jaw - aref;
  vr f = sel(act & (jar < vr(real(0))), -Dd * jar, vr(real(0)));
  vr g = bb;
  for_contacts<0, kMaxCon2>(nmax, [&](auto ccT) {
    constexpr int cc = decltype(ccT)::value;
    sfor4([&](auto rT) { constexpr int i = 4 * cc + decltype(rT)::value; g = fma_half_lane<i>(A[i], f, g, h1); });
  });
  {
    const vr cost = hsum32(sel(act, f * (bb + real(0.5) * (g - bb + Rr * f)), vr(real(0))));
    const VB worse = cost > vr(real(0));
    f = sel(worse, vr(real(0)), f);
    g = sel(worse, bb, g);
  }
  // ---- mj_solPGS, each half until ITS tolerance exit
  const vr hA = real(0.5) * ARjj;
  V<int> itp = V<int>(0), itn = V<int>(0);
  {
    VB run = VB(true);
    for (int iter = 0; iter < (nosweep ? 0 : M.pgs_iters); iter++) {
      vr k0, k1, nf;
      pgs_row_prepare(Rr, f, ARinv, k0, k1, nf);
      k0 = sel(run, k0, vr(real(0))); k1 = sel(run, k1, vr(real(0))); nf = sel(run, nf, vr(real(0)));      // a half that has met its tolerance stands still
      vr gcap = g;
      uint64_t rowm = mask_shl(0x0000000100000001ull, 0);     // lanes i and 32 + i of the row whose turn it is
      for_contacts<0, kMaxCon2>(nmax, [&](auto ccT) {
        constexpr int cc = decltype(ccT)::value;
        sfor4([&](auto rT) {
          constexpr int i = 4 * cc + decltype(rT)::value;
          gcap = sel_mask(rowm, g, gcap);        // the residual this row sees at its own step
          rowm = mask_shl(rowm, 1);
          const vr dl = pgs_row_delta(g, k0, k1, nf);
          g = fma_half_lane<i>(A[i], dl, g, h1);
        });
      });
      const vr dcap = pgs_row_delta(gcap, k0, k1, nf);
      const vr ccap = pgs_row_cost(dcap, hA, gcap, Rr, f);
      f = f + dcap;
      itp = itp + sel(run, V<int>(1), V<int>(0));
      run = run & !((-hsum32(ccap)) * M.pgs_scale < vr(M.pgs_tol));
      if (!wany(run)) break;
    }
  }
  nm_stamp(6);
  // ---- mj_solNoSlip
  {
    const V<int> lv = opaque_lane() & 31;
    const VB even = (lv & 1) == 0;
    vr Amq = vr(real(0));
    for_contacts<0, kMaxCon2>(nmax, [&](auto ccT) {
      constexpr int cc = decltype(ccT)::value;
      Amq = sel(lv == 4 * cc, A[4 * cc + 1], Amq);
      Amq = sel(lv == 4 * cc + 2, A[4 * cc + 3], Amq);
    });
    Amq = sel(even, Amq, shfl_xor1(Amq));
    const vr K1 = Ajj + shfl_xor1(Ajj) - Amq - Amq;
    const VB small = K1 < vr(real(1e-15));
    const vr invK1 = vrcp(K1), hK1 = real(0.5) * K1;
    for_contacts<0, kMaxCon2>(nmax, [&](auto ccT) {     // column differences, as in stage_constraint_body
      constexpr int cc = decltype(ccT)::value;
      A[4 * cc] = A[4 * cc] - A[4 * cc + 1];
      A[4 * cc + 2] = A[4 * cc + 2] - A[4 * cc + 3];
    });
    VB run = VB(true);
    const bool anysmall = wany(small & act);
    for (int iter = 0; iter < (nosweep ? 0 : M.noslip_iters); iter++) {
      vr improvement = vr(real(0));
      if (iter == 0) improvement = hsum32(sel(act, real(0.5) * f * f * Rr, vr(real(0))));
      vr lo, hi;
      noslip_pair_prepare<real>(f, shfl_xor1(f), small, lo, hi);
      lo = sel(run, lo, vr(real(0))); hi = sel(run, hi, vr(real(0)));
      vr dcap = vr(real(0)), dgcap = vr(real(0));
      uint64_t pairm = mask_shl(0x0000000300000003ull, 0);
      // mj_solNoSlip's revert test (cost change > 1e-10) can only fire on a DEGENERATE pair: with K1 > 0 the step is d = clamp(-dg / K1) to
      // [-f, fp] (-f <= 0 <= fp), so d and 0.5 K1 d + dg have opposite signs or one of them is zero in every case of the clamp - unclamped the
      // second factor is ~ dg / 2, clamped at -f (then dg > 0) it lies in [dg / 2, dg], at fp (dg < 0) in [dg, dg / 2] - and the change
      // d (0.5 K1 d + dg) <= 0 in floating point as well (the signs of the factors are exact, the magnitudes far from cancelling). So the
      // test (fma, mul, compare, select: 4 of a pair's 16 VALU instructions) runs only in sweeps of a wave that HAS a degenerate pair -
      // one uniform branch per sweep; same results either way.
      auto pair_sweep = [&](auto checkT) {
        sfor_pairs<kMaxRow2 / 2>(nmax, [&](auto pT) {
          constexpr int p = decltype(pT)::value;
          const vr dg = g - shfl_xor1(g);
          vr d = noslip_pair_delta(dg, invK1, lo, hi);
          if constexpr (decltype(checkT)::value) d = sel(noslip_pair_bad<real>(noslip_pair_cost(d, hK1, dg)), vr(real(0)), d);
          g = fma_half_lane<2 * p>(A[2 * p], d, g, h1);
          dcap = sel_mask(pairm, d, dcap);
          dgcap = sel_mask(pairm, dg, dgcap);
          pairm = mask_shl(pairm, 2);
        });
      };
      if (anysmall) pair_sweep(std::true_type{}); else pair_sweep(std::false_type{});
      f = f + dcap;
      improvement = improvement - hsum32(sel(even, noslip_pair_cost(dcap, hK1, dgcap), vr(real(0))));
      itn = itn + sel(run, V<int>(1), V<int>(0));
      run = run & !(improvement * M.pgs_scale < vr(M.noslip_tol));
      if (!wany(run)) break;
    }
  }
  nm_stamp(7);
  // values that are equal in all 32 lanes of a half are stored unmasked (same value, same address): a masked store is an exec-mask
  // branch, i.e. the end of a scheduling region
  stsv(ibw, hoi + NM_IOFS(it_pgs), itp, VB(true));
  stsv(ibw, hoi + NM_IOFS(it_noslip), itn, VB(true));
  stsv(rbw, ho + (hl + NM_OFS(efc_f)), f, VB(true));
  stsv(rbw, ho + (hl + (kMaxRow2 + NM_OFS(efc_f))), vr(real(0)), VB(true));
  // ---- qfrc_constraint = J' f and the touch sensors
  vr nf = vr(real(0));
  VB hit = VB(false);
  if (last) {
    nf = f + shfl_xor1(f);
    nf = nf + shfl_xor2(nf);
    vr ft[3], fr, sv3[3], Rt[9];
#pragma unroll
    for (int j = 0; j < 3; j++) sv3[j] = ldsv(M.footc, Lc * 4 + j);
    fr = ldsv(M.footc, Lc * 4 + 3);
#pragma unroll
    for (int j = 0; j < 9; j++) Rt[j] = SHR(colR, (Lc + 1) * 9 + j);
    matvec3(ft, Rt, sv3);
#pragma unroll
    for (int j = 0; j < 3; j++) ft[j] = ft[j] + SHR(colp, (Lc + 1) * 3 + j);
    vr dif[3] = {cp[0] - ft[0], cp[1] - ft[1], cp[2] - ft[2]};
    vr b2 = real(-1) * (nrm[0] * dif[0] + nrm[1] * dif[1] + nrm[2] * dif[2]);
    vr cq = dif[0] * dif[0] + dif[1] * dif[1] + dif[2] * dif[2] - fr * fr;
    vr det = b2 * b2 - cq;
    vr sq = vsqrt(vmax(det, vr(real(0))));
    hit = !(det < vr(real(1e-15))) & (((-b2 - sq) >= vr(real(0))) | ((-b2 + sq) >= vr(real(0))));
  }
#pragma unroll
  for (int j = 0; j < 6; j++) stsv(rbw, ho + (j + NM_OFS(qfc)), hsum32(Jb[j] * f), VB(true));
  {
    vr s3[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      vr x = Jl[k] * f;
      x = x + shfl_xor1(x);
      s3[k] = x + shfl_xor2(x);
    }
    const vr tn = sel(act & (nf > vr(real(0))), nf, vr(real(0))), tf = sel(hit, tn, vr(real(0)));
    const VB lead = act & ((hl & 3) == 0);
    wave_sync();
#pragma unroll
    for (int k = 0; k < 3; k++) stsu(jrow, jh + (c * 8 + k), s3[k], lead, w.e[0].sink);
    if (last) { stsu(jrow, jh + (c * 8 + 3), tn, lead, w.e[0].sink); stsu(jrow, jh + (c * 8 + 4), tf, lead, w.e[0].sink); }
    wave_sync();
    const VB on = hl < 30;
    const V<int> lg = sel(on, (hl * 13) >> 6, V<int>(0)), qq = hl - lg * 5;
    const V<int> c0 = SHI(cstart, lg + 1), cn = SHI(ccnt, lg + 1);
    vr acc = vr(real(0));
#pragma unroll
    for (int k = 0; k < 4; k++) {
      VB in = on & (cn > k);
      acc = acc + sel(in, ldsv(jrow, jh + sel(in, (c0 + k) * 8 + qq, V<int>(0))), vr(real(0)));
    }
    stsu(rbw, ho + (lg * 3 + qq + (6 + NM_OFS(qfc))), acc, on & (qq < 3), w.e[0].sink);
    if (last) {
      stsu(rbw, ho + (lg + NM_OFS(sens)), acc, on & (qq == 3), w.e[0].sink);          // tibia sites: 10 m spheres see every contact of the body
      stsu(rbw, ho + (lg + (6 + NM_OFS(sens))), acc, on & (qq == 4), w.e[0].sink);    // foot sites
      stsv(rbw, ho + V<int>(12 + NM_OFS(sens)), hsum32(sel(act & ((hl & 3) == 0) & (L < 0), tn, vr(real(0)))), VB(true));
    }
  }
  wave_sync();
#undef SHR
#undef SHI
}

// =========================================================================================  stage C beyond kMaxCon contacts
// Rare: a robot lying on the floor with folded legs can touch it with up to 28 hull vertices, and 15 tibia pairs can collide on
// top of that; upstream keeps every contact (MuJoCo has no cap), so this path does too. It solves the SAME constraint set with the
// SAME Gauss-Seidel / NoSlip updates in the SAME row order as the register-resident solver above, but matrix-free: row i lives in
// slot i/64 of lane i%64 (kBigSlots rows per lane), and instead of a row of A = J M^-1 J' per lane the wave keeps the 24-vector
// v = M^-1 J' f in LDS: residual_i = J_i.v + b_i, and a change d of f_i updates v by M^-1 J_i' d through the block factor
// (legs on lanes 0..5 as in stage D). O(rows) state, ~0.1 us per row update; speed does not matter here, equality with upstream does.
template <class real> struct BigRows {
  V<real> Jb[kBigSlots][6], Jl[kBigSlots][3], Jm[kBigSlots][3], Rr[kBigSlots], bb[kBigSlots], f[kBigSlots], ARinv[kBigSlots], hA[kBigSlots];
  V<real> invK1[kBigSlots], hK1[kBigSlots];
  V<int> L[kBigSlots], L1[kBigSlots];
  VB act[kBigSlots], small[kBigSlots];
};
// v (+)= M^-1 y for y = (yb[6] uniform base part, yl[3] per leg on lanes 0..5) through the block factor of M
template <class real> NM_FN void big_minv(Sh<real>& sh, const real* yb, const V<real>* yl, bool accumulate) {
  typedef V<real> vr;
  const V<int> lane = opaque_lane();
  const VB isleg = lane < kNLEG;
  const V<int> leg = vmin(lane, V<int>(kNLEG - 1));
  vr t[3], Mi[6], xb[6];
#pragma unroll
  for (int j = 0; j < 6; j++) Mi[j] = ldsv(sh.Minv, leg * 6 + j);
  ldl3_solve(t, Mi, yl);
#pragma unroll
  for (int j = 0; j < 6; j++) {
    vr wy = ldsv(sh.W, leg * 18 + j) * yl[0] + ldsv(sh.W, leg * 18 + (6 + j)) * yl[1] + ldsv(sh.W, leg * 18 + (12 + j)) * yl[2];
    xb[j] = vr(yb[j] - lanesum6<real>(sel(isleg, wy, vr(real(0)))));
  }
  {
    vr Lf[15], Di[6];
#pragma unroll
    for (int j = 0; j < 15; j++) Lf[j] = vr(sh.Lb[j]);
#pragma unroll
    for (int j = 0; j < 6; j++) Di[j] = vr(sh.Dbi[j]);
    ldl6_solve(Lf, Di, xb);
  }
  vr x[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    x[k] = t[k];
#pragma unroll
    for (int j = 0; j < 6; j++) x[k] = x[k] - ldsv(sh.W, leg * 18 + (6 * k + j)) * xb[j];
  }
  wave_sync();
#pragma unroll
  for (int j = 0; j < 6; j++) {
    vr nv = accumulate ? ldsv(sh.vv, V<int>(j)) + xb[j] : xb[j];
    stsv(sh.vv, V<int>(j), nv, lane == 0);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    vr nv = accumulate ? ldsv(sh.vv, leg * 3 + (6 + k)) + x[k] : x[k];
    stsv(sh.vv, leg * 3 + (6 + k), nv, isleg);
  }
  wave_sync();
}
// v += M^-1 J' d for ONE row given by wave-uniform copies of its Jacobian (times d) and legs
template <class real> NM_FN void big_update_v(Sh<real>& sh, const real* jb, const real* jl, const real* jm, int Li, int L1i) {
  typedef V<real> vr;
  const V<int> leg = vmin(opaque_lane(), V<int>(kNLEG - 1));
  vr yl[3];
#pragma unroll
  for (int k = 0; k < 3; k++) yl[k] = sel(leg == Li, vr(jl[k]), vr(real(0))) + sel(leg == L1i, vr(jm[k]), vr(real(0)));
  big_minv(sh, jb, yl, true);
}
template <class real> NM_FN V<real> big_residual(const Sh<real>& sh, const BigRows<real>& r, int k) {   // J_i . v + b_i (no R term)
  V<real> g = r.bb[k];
  const V<int> Lc = vmax(r.L[k], V<int>(0)), Lc1 = vmax(r.L1[k], V<int>(0));
#pragma unroll
  for (int j = 0; j < 6; j++) g += r.Jb[k][j] * sh.vv[j];
#pragma unroll
  for (int j = 0; j < 3; j++) g += r.Jl[k][j] * ldsv(sh.vv, Lc * 3 + (6 + j)) + r.Jm[k][j] * ldsv(sh.vv, Lc1 * 3 + (6 + j));
  return g;
}
template <class real> NM_FN void big_jtf(Sh<real>& sh, const BigRows<real>& r) {   // qfc = J' f
  typedef V<real> vr;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    vr a = vr(real(0));
#pragma unroll
    for (int k = 0; k < kBigSlots; k++) a += r.Jb[k][j] * r.f[k];
    sh.qfc[j] = wsum<real>(a);
  }
  for (int l = 0; l < kNLEG; l++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      vr a = vr(real(0));
#pragma unroll
      for (int k = 0; k < kBigSlots; k++)
        a += sel(r.L[k] == l, r.Jl[k][j] * r.f[k], vr(real(0))) + sel(r.L1[k] == l, r.Jm[k][j] * r.f[k], vr(real(0)));
      sh.qfc[6 + 3 * l + j] = wsum<real>(a);
    }
  wave_sync();
}
template <class real> NM_COLD void stage_constraint_big(Sh<real>& sh, const Model<real>& M, bool last, bool nosweep) {
  typedef V<real> vr;
  const V<int> lane = opaque_lane();
  const int ncon = uniform(sh.ncon), nefc = 4 * ncon;
  BigRows<real> r;
  const V<int> q4 = lane & 3;
  const vr smu = sel((lane & 1) == 0, vr(M.mu), vr(-M.mu));
  const VB even = (lane & 1) == 0;
  // ---- rows: frame-axis Jacobians, pyramid rows, impedance / R / aref, diagonal of A and the pair coupling for NoSlip
#pragma unroll
  for (int k = 0; k < kBigSlots; k++) {
    const V<int> row = lane + NM_WAVE * k;
    const VB act = row < nefc;
    const V<int> c = sel(act, row >> 2, V<int>(0));
    vr cp[3] = {ldsv(sh.cpos(), c * 3), ldsv(sh.cpos(), c * 3 + 1), ldsv(sh.cpos(), c * 3 + 2)};
    vr nrm[3] = {ldsv(sh.cnrm(), c * 3), ldsv(sh.cnrm(), c * 3 + 1), ldsv(sh.cnrm(), c * 3 + 2)};
    const vr dist = ldsv(sh.cdist(), c);
    const V<int> L = sel(act, ldsv(sh.cleg(), c), V<int>(-1)), L1 = sel(act, ldsv(sh.cleg1(), c), V<int>(-1));
    const VB onleg = L >= 0, onleg1 = L1 >= 0;
    const V<int> Lc = vmax(L, V<int>(0)), Lc1 = vmax(L1, V<int>(0));
    vr e[3];
    {  // mju_makeFrame
      VB usey = (nrm[1] < vr(real(0.5))) & (nrm[1] > vr(real(-0.5)));
      vr y0[3] = {vr(real(0)), sel(usey, vr(real(1)), vr(real(0))), sel(usey, vr(real(0)), vr(real(1)))};
      vr dt = nrm[1] * y0[1] + nrm[2] * y0[2];
      vr t1[3] = {y0[0] - dt * nrm[0], y0[1] - dt * nrm[1], y0[2] - dt * nrm[2]}, t2[3];
      vr il = vr(real(1)) / vsqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
      t1[0] = t1[0] * il; t1[1] = t1[1] * il; t1[2] = t1[2] * il;
      cross3(t2, nrm, t1);
#pragma unroll
      for (int j = 0; j < 3; j++) e[j] = sel(q4 == 1, t1[j], sel(q4 == 2, t2[j], nrm[j]));
    }
    vr Fb[6], Fl[3], Fm[3];
    {
      vr m[3];
      cross3(m, cp, e);
      Fb[0] = e[0]; Fb[1] = e[1]; Fb[2] = e[2];
#pragma unroll
      for (int j = 0; j < 3; j++) Fb[3 + j] = sh.Rb[j] * m[0] + sh.Rb[3 + j] * m[1] + sh.Rb[6 + j] * m[2];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        vr a[3], rel[3], mm[3], a1[3], rel1[3], mm1[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
          a[j] = ldsv(sh.axs, Lc * 9 + (3 * kk + j));
          rel[j] = cp[j] - ldsv(sh.anc, Lc * 9 + (3 * kk + j));
          a1[j] = ldsv(sh.axs, Lc1 * 9 + (3 * kk + j));
          rel1[j] = cp[j] - ldsv(sh.anc, Lc1 * 9 + (3 * kk + j));
        }
        cross3(mm, rel, e);
        cross3(mm1, rel1, e);
        Fl[kk] = sel(onleg & act, dot3<vr>(a, mm), vr(real(0)));
        Fm[kk] = sel(onleg1 & act, -dot3<vr>(a1, mm1), vr(real(0)));   // body1 side: -J(body1)
      }
#pragma unroll
      for (int j = 0; j < 6; j++) Fb[j] = sel(act & !onleg1, Fb[j], vr(real(0)));   // base columns of J(b2) - J(b1) cancel
    }
    vr uF[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      uF[j] = Fb[j];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) uF[j] = uF[j] - ldsv(sh.W, Lc * 18 + (6 * kk + j)) * Fl[kk] - ldsv(sh.W, Lc1 * 18 + (6 * kk + j)) * Fm[kk];
    }
    vr u[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      r.Jb[k][j] = quad<0x00>(Fb[j]) + smu * quad<0xA5>(Fb[j]);
      u[j] = quad<0x00>(uF[j]) + smu * quad<0xA5>(uF[j]);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      r.Jl[k][j] = quad<0x00>(Fl[j]) + smu * quad<0xA5>(Fl[j]);
      r.Jm[k][j] = quad<0x00>(Fm[j]) + smu * quad<0xA5>(Fm[j]);
    }
    vr imp;
    {
      vr x = vabs(dist) * vrcp(M.si_width);
      vr ylo, yhi;
      if (M.si_power == real(2)) {
        vr omx = vmax(vr(real(1)) - x, vr(real(0)));
        ylo = (x * x) * vrcp(M.si_mid);
        yhi = vr(real(1)) - (omx * omx) * vrcp(real(1) - M.si_mid);
      } else {
        ylo = vpow(x, vr(M.si_power)) / vpow(vr(M.si_mid), vr(M.si_power - real(1)));
        yhi = vr(real(1)) - vpow(vmax(vr(real(1)) - x, vr(real(0))), vr(M.si_power)) / vpow(vr(real(1) - M.si_mid), vr(M.si_power - real(1)));
      }
      vr y = sel(x <= vr(M.si_mid), ylo, yhi);
      imp = M.si_d0 + y * (M.si_dmax - M.si_d0);
      imp = sel(x >= vr(real(1)), vr(M.si_dmax), imp);
      imp = sel(x <= vr(real(0)), vr(M.si_d0), imp);
    }
    vr invw = ldsv(M.colc, (Lc + sel(onleg, V<int>(1), V<int>(0))) * kColN + 4);
    invw = invw + sel(onleg1, ldsv(M.colc, (Lc1 + 1) * kColN + 4), vr(real(0)));
    const vr Rr = vmax((vr(real(1)) - imp) * (invw + M.mu * M.mu * invw) * vrcp(imp), vr(real(1e-15))) * (real(2) * M.mu * M.mu);
    vr vel = vr(real(0)), jas = vr(real(0)), jaw = vr(real(0));
#pragma unroll
    for (int j = 0; j < 6; j++) {
      vel += r.Jb[k][j] * sh.qvel[j]; jas += r.Jb[k][j] * sh.qas[j]; jaw += r.Jb[k][j] * sh.warm[j];
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      V<int> di = Lc * 3 + (6 + j), di1 = Lc1 * 3 + (6 + j);
      vel += r.Jl[k][j] * ldsv(sh.qvel, di) + r.Jm[k][j] * ldsv(sh.qvel, di1);
      jas += r.Jl[k][j] * ldsv(sh.qas, di) + r.Jm[k][j] * ldsv(sh.qas, di1);
      jaw += r.Jl[k][j] * ldsv(sh.warm, di) + r.Jm[k][j] * ldsv(sh.warm, di1);
    }
    const vr aref = -M.solref_B * vel - M.solref_K * imp * dist;
    vr t[3], t1m[3], xb[6];
    {
      vr Mi[6], Mi1[6];
#pragma unroll
      for (int j = 0; j < 6; j++) { Mi[j] = ldsv(sh.Minv, Lc * 6 + j); Mi1[j] = ldsv(sh.Minv, Lc1 * 6 + j); }
      ldl3_solve(t, Mi, r.Jl[k]);
      ldl3_solve(t1m, Mi1, r.Jm[k]);
#pragma unroll
      for (int j = 0; j < 6; j++) xb[j] = u[j];
      ldl6_solve(sh.Lb, sh.Dbi, xb);
    }
    vr Ajj = r.Jl[k][0] * t[0] + r.Jl[k][1] * t[1] + r.Jl[k][2] * t[2] + (r.Jm[k][0] * t1m[0] + r.Jm[k][1] * t1m[1] + r.Jm[k][2] * t1m[2]);
    vr Apq = r.Jl[k][0] * shfl_xor1(t[0]) + r.Jl[k][1] * shfl_xor1(t[1]) + r.Jl[k][2] * shfl_xor1(t[2]) +
             (r.Jm[k][0] * shfl_xor1(t1m[0]) + r.Jm[k][1] * shfl_xor1(t1m[1]) + r.Jm[k][2] * shfl_xor1(t1m[2]));
#pragma unroll
    for (int j = 0; j < 6; j++) { Ajj += u[j] * xb[j]; Apq += u[j] * shfl_xor1(xb[j]); }
    Apq = sel(even, Apq, shfl_xor1(Apq));   // the even row's copy of A[2p][2p+1] in both lanes of the pair
    const vr K1 = Ajj + shfl_xor1(Ajj) - Apq - Apq;
    r.small[k] = K1 < vr(real(1e-15));
    r.invK1[k] = vrcp(K1);
    r.hK1[k] = real(0.5) * K1;
    const vr ARjj = Ajj + Rr;
    r.ARinv[k] = sel(act, vrcp(ARjj), vr(real(0)));
    r.hA[k] = real(0.5) * ARjj;
    r.Rr[k] = Rr;
    r.bb[k] = jas - aref;
    r.L[k] = L; r.L1[k] = L1; r.act[k] = act;
    const vr jar = jaw - aref;   // warm start (PGS branch of mj_fwdConstraint)
    r.f[k] = sel(act & (jar < vr(real(0))), -vrcp(Rr) * jar, vr(real(0)));
  }
  // ---- warm start: keep f only if its dual cost is below that of zero
  {
    big_jtf(sh, r);
    const V<int> leg = vmin(lane, V<int>(kNLEG - 1));
    vr yl[3] = {ldsv(sh.qfc, leg * 3 + 6), ldsv(sh.qfc, leg * 3 + 7), ldsv(sh.qfc, leg * 3 + 8)};
    real yb[6];
#pragma unroll
    for (int j = 0; j < 6; j++) yb[j] = sh.qfc[j];
    big_minv(sh, yb, yl, false);
    vr cterm = vr(real(0));
#pragma unroll
    for (int k = 0; k < kBigSlots; k++) {
      const vr g = big_residual(sh, r, k);
      cterm += sel(r.act[k], r.f[k] * (r.bb[k] + real(0.5) * (g - r.bb[k] + r.Rr[k] * r.f[k])), vr(real(0)));
    }
    if (wsum<real>(cterm) > real(0)) {
#pragma unroll
      for (int k = 0; k < kBigSlots; k++) r.f[k] = vr(real(0));
      wave_sync();
      stsv(sh.vv, sel(lane < 24, lane, V<int>(0)), real(0), lane < 24);
      wave_sync();
    }
  }
  // ---- mj_solPGS, row by row
  for (int iter = 0; iter < (nosweep ? 0 : M.pgs_iters); iter++) {
    real improvement = real(0);
#pragma unroll
    for (int k = 0; k < kBigSlots; k++) {
      for (int ln = 0; ln < NM_WAVE; ln++) {
        if (NM_WAVE * k + ln >= nefc) break;
        vr k0, k1, nf;
        pgs_row_prepare(r.Rr[k], r.f[k], r.ARinv[k], k0, k1, nf);
        const vr gk = big_residual(sh, r, k);
        const vr dl = pgs_row_delta(gk, k0, k1, nf);
        const vr change = pgs_row_cost(dl, r.hA[k], gk, r.Rr[k], r.f[k]);
        const real d = rdlane(dl, ln);
        improvement -= rdlane(change, ln);
        r.f[k] = sel(lane == ln, r.f[k] + dl, r.f[k]);
        real jb[6], jl[3], jm[3];
#pragma unroll
        for (int j = 0; j < 6; j++) jb[j] = rdlane(r.Jb[k][j], ln) * d;
#pragma unroll
        for (int j = 0; j < 3; j++) { jl[j] = rdlane(r.Jl[k][j], ln) * d; jm[j] = rdlane(r.Jm[k][j], ln) * d; }
        big_update_v(sh, jb, jl, jm, rdlane(r.L[k], ln), rdlane(r.L1[k], ln));
      }
    }
    sh.it_pgs = iter + 1;
    if (improvement * M.pgs_scale < M.pgs_tol) break;
  }
  // ---- mj_solNoSlip, pair by pair (the Newton form derived above)
  for (int iter = 0; iter < (nosweep ? 0 : M.noslip_iters); iter++) {
    real improvement = real(0);
    if (iter == 0) {
      vr a = vr(real(0));
#pragma unroll
      for (int k = 0; k < kBigSlots; k++) a += sel(r.act[k], real(0.5) * r.f[k] * r.f[k] * r.Rr[k], vr(real(0)));
      improvement = wsum<real>(a);
    }
#pragma unroll
    for (int k = 0; k < kBigSlots; k++) {
      for (int pp = 0; pp < NM_WAVE / 2; pp++) {
        if (NM_WAVE * k + 2 * pp >= nefc) break;
        const vr g = big_residual(sh, r, k);
        vr lo, hi;
        noslip_pair_prepare<real>(r.f[k], shfl_xor1(r.f[k]), r.small[k], lo, hi);
        const vr dg = g - shfl_xor1(g);
        vr d = noslip_pair_delta(dg, r.invK1[k], lo, hi);
        const vr change = noslip_pair_cost(d, r.hK1[k], dg);
        const VB bad = noslip_pair_bad<real>(change);
        d = sel(bad, vr(real(0)), d);
        const real d0 = rdlane(d, 2 * pp), d1 = rdlane(d, 2 * pp + 1);
        improvement -= rdlane(sel(bad, vr(real(0)), change), 2 * pp);
        r.f[k] = sel((lane >> 1) == pp, r.f[k] + d, r.f[k]);
        real jb[6], jl[3], jm[3];
#pragma unroll
        for (int j = 0; j < 6; j++) jb[j] = rdlane(r.Jb[k][j], 2 * pp) * d0 + rdlane(r.Jb[k][j], 2 * pp + 1) * d1;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          jl[j] = rdlane(r.Jl[k][j], 2 * pp) * d0 + rdlane(r.Jl[k][j], 2 * pp + 1) * d1;
          jm[j] = rdlane(r.Jm[k][j], 2 * pp) * d0 + rdlane(r.Jm[k][j], 2 * pp + 1) * d1;
        }
        big_update_v(sh, jb, jl, jm, rdlane(r.L[k], 2 * pp), rdlane(r.L1[k], 2 * pp));
      }
    }
    sh.it_noslip = iter + 1;
    if (improvement * M.pgs_scale < M.noslip_tol) break;
  }
  // ---- qfrc_constraint = J' f and the touch sensors
  big_jtf(sh, r);
  stsv(sh.efc_f, lane, r.f[0], lane < kMaxRow);
  if (last) {
    vr s_all[kNLEG], s_foot[kNLEG], s_base = vr(real(0));
#pragma unroll
    for (int l = 0; l < kNLEG; l++) { s_all[l] = vr(real(0)); s_foot[l] = vr(real(0)); }
#pragma unroll
    for (int k = 0; k < kBigSlots; k++) {
      const V<int> row = lane + NM_WAVE * k;
      const V<int> c = sel(r.act[k], row >> 2, V<int>(0));
      vr cp[3] = {ldsv(sh.cpos(), c * 3), ldsv(sh.cpos(), c * 3 + 1), ldsv(sh.cpos(), c * 3 + 2)};
      vr nrm[3] = {ldsv(sh.cnrm(), c * 3), ldsv(sh.cnrm(), c * 3 + 1), ldsv(sh.cnrm(), c * 3 + 2)};
      vr nf = r.f[k] + shfl_xor1(r.f[k]);
      nf = nf + shfl_xor2(nf);
      auto foot_hit = [&](const V<int>& Lx, real sgn) {   // mju_rayGeom against the foot site sphere, as in the resident solver
        vr ft[3], fr, sp[3], Rt[9];
#pragma unroll
        for (int j = 0; j < 3; j++) sp[j] = ldsv(M.footc, Lx * 4 + j);
        fr = ldsv(M.footc, Lx * 4 + 3);
#pragma unroll
        for (int j = 0; j < 9; j++) Rt[j] = ldsv(sh.colR, (Lx + 1) * 9 + j);
        matvec3(ft, Rt, sp);
#pragma unroll
        for (int j = 0; j < 3; j++) ft[j] = ft[j] + ldsv(sh.colp, (Lx + 1) * 3 + j);
        vr dif[3] = {cp[0] - ft[0], cp[1] - ft[1], cp[2] - ft[2]};
        vr b2 = sgn * (nrm[0] * dif[0] + nrm[1] * dif[1] + nrm[2] * dif[2]);
        vr cc = dif[0] * dif[0] + dif[1] * dif[1] + dif[2] * dif[2] - fr * fr;
        vr det = b2 * b2 - cc;
        vr sq = vsqrt(vmax(det, vr(real(0))));
        return !(det < vr(real(1e-15))) & (((-b2 - sq) >= vr(real(0))) | ((-b2 + sq) >= vr(real(0))));
      };
      const VB hit = foot_hit(vmax(r.L[k], V<int>(0)), real(-1)), hit1 = foot_hit(vmax(r.L1[k], V<int>(0)), real(1));
      const VB head = r.act[k] & ((lane & 3) == 0) & (nf > vr(real(0)));
#pragma unroll
      for (int l = 0; l < kNLEG; l++) {
        s_all[l] += sel(head & ((r.L[k] == l) | (r.L1[k] == l)), nf, vr(real(0)));
        s_foot[l] += sel(head & (((r.L[k] == l) & hit) | ((r.L1[k] == l) & hit1)), nf, vr(real(0)));
      }
      s_base += sel(head & (r.L[k] < 0), nf, vr(real(0)));
    }
#pragma unroll
    for (int l = 0; l < kNLEG; l++) { sh.sens[l] = wsum<real>(s_all[l]); sh.sens[6 + l] = wsum<real>(s_foot[l]); }
    sh.sens[12] = wsum<real>(s_base);
  }
  wave_sync();
}
// contact frame and first body of the floor contacts: constants that stage B does not store; the two rare paths that read the
// general contact record (tibia pairs present, more than kMaxCon contacts) get them here
template <class real> NM_COLD void floor_frames(Sh<real>& sh) {
  const V<int> lane = lane_id();
  const int nfloor = uniform(sh.cstart[7]);
  const VB on = lane < nfloor;                  // nfloor <= 28 < 64
  const V<int> c = sel(on, lane, V<int>(0));
  stsv(sh.cleg1(), c, V<int>(-1), on);
  stsv(sh.cnrm(), c * 3, V<real>(real(0)), on);
  stsv(sh.cnrm(), c * 3 + 1, V<real>(real(0)), on);
  stsv(sh.cnrm(), c * 3 + 2, V<real>(real(1)), on);
  wave_sync();
}
template <class real> NM_FN void stage_constraint(Sh<real>& sh, real* jrow, const Model<real>& M, bool last, bool nosweep = false) {
  if (uniform(sh.ncon) > kMaxCon) { floor_frames(sh); stage_constraint_big<real>(sh, M, last, nosweep); }
  else if (uniform(sh.anypair) != 0) { floor_frames(sh); stage_constraint_pairs<real>(sh, jrow, M, last, nosweep); }
  else stage_constraint_body<real, false>(sh, jrow, M, last, nosweep);
}

// =========================================================================================  stage D
// qacc (for the warm start), implicitfast velocity update, position integration - all G envs of the wave at once
// (lane groups as in stage A). An env whose qacc is bad (mj_checkAcc) is reset and advanced by the closed form of
// "mj_resetData; mj_forward; integrate": free fall from qpos0 with zero ctrl.
template <class real, int G> NM_FN void stage_integrate(ShW<real, G>& w, const Model<real>& M) {
  typedef V<real> vr;
  real* lds = reinterpret_cast<real*>(&w.e[0]);
  // lane groups as in stage A: group g works on env g % G; groups [0, G) solve with the factor of M (qacc, for the warm start and
  // mj_checkAcc), groups [G, 2G) with the factor of M + h kv I (the implicitfast velocity update) - one pass on different lanes. What
  // a group's solve does not feed is stored to a junk row (stage A's mbb / sc, the matrix-free solver's vv: all spent here; the contact
  // list in the leg staging area stays intact for the debug dump).
  const V<int> lane0 = opaque_lane();
  const V<int> sub = lane0 & 7, grp = lane0 >> 3;
  const V<int> leg = vmin(sub, V<int>(5)), eo = (grp & (G - 1)) * (int)(sizeof(Sh<real>) / sizeof(real));
  const VB pass1 = (grp & G) != 0;
  const VB isleg = sub < V<int>(6);   // as in stage A: every lane group computes (and stores) its env's values, duplicates included
#define LDG(field, i) ldsv(lds, eo + (NM_OFS(field) + (i)))
#define LDL(field, idx) ldsv(lds, eo + (idx) + NM_OFS(field))
  constexpr int kFacH = NM_OFS(MinvH) - NM_OFS(Minv);
  static_assert(NM_OFS(WH) - NM_OFS(W) == kFacH && NM_OFS(LbH) - NM_OFS(Lb) == kFacH && NM_OFS(DbiH) - NM_OFS(Dbi) == kFacH, "factor blocks");
  static_assert(sizeof(Sh<real>::mbb) / sizeof(real) >= 24 && sizeof(Sh<real>::sc) / sizeof(real) >= 28 && sizeof(Sh<real>::vv) / sizeof(real) >= 24, "junk rows");
  const V<int> eoF = eo + sel(pass1, V<int>(kFacH), V<int>(0));
  vr xl[3], xb[6];     // groups [0, G): M^-1 qfrc_constraint ; groups [G, 2G): (M + h kv I)^-1 (qfrc_smooth + qfrc_constraint)
  {
    vr y[3], t[3], Mi[6];
#pragma unroll
    for (int k = 0; k < 3; k++) y[k] = LDL(qfc, leg * 3 + (6 + k)) + sel(pass1, LDL(qfs, leg * 3 + (6 + k)), vr(real(0)));
#pragma unroll
    for (int j = 0; j < 6; j++) Mi[j] = ldsv(lds, eoF + leg * 6 + (NM_OFS(Minv) + j));
    ldl3_solve(t, Mi, y);
#pragma unroll
    for (int j = 0; j < 6; j++) {
      vr wy = ldsv(lds, eoF + leg * 18 + (NM_OFS(W) + j)) * y[0] + ldsv(lds, eoF + leg * 18 + (NM_OFS(W) + 6 + j)) * y[1] +
              ldsv(lds, eoF + leg * 18 + (NM_OFS(W) + 12 + j)) * y[2];
      xb[j] = LDG(qfc, j) - legsum<real>(wy, isleg) + sel(pass1, LDG(qfs, j), vr(real(0)));
    }
    {
      vr L[15], Di[6];
#pragma unroll
      for (int j = 0; j < 15; j++) L[j] = ldsv(lds, eoF + (NM_OFS(Lb) + j));
#pragma unroll
      for (int j = 0; j < 6; j++) Di[j] = ldsv(lds, eoF + (NM_OFS(Dbi) + j));
      ldl6_solve(L, Di, xb);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      vr x = t[k];
#pragma unroll
      for (int j = 0; j < 6; j++) x = x - ldsv(lds, eoF + leg * 18 + (NM_OFS(W) + 6 * k + j)) * xb[j];
      xl[k] = x;
    }
  }
  vr qacc_l[3], qacc_b[6];       // meaningful in groups [0, G)
#pragma unroll
  for (int k = 0; k < 3; k++) qacc_l[k] = xl[k] + LDL(qas, leg * 3 + (6 + k));
#pragma unroll
  for (int j = 0; j < 6; j++) qacc_b[j] = xb[j] + LDG(qas, j);
  const vr* qint_l = xl; const vr* qint_b = xb;   // meaningful in groups [G, 2G)
  // mj_checkAcc, per env
  VB badl = (visbad(qacc_l[0]) | visbad(qacc_l[1]) | visbad(qacc_l[2])) & isleg;
#pragma unroll
  for (int j = 0; j < 6; j++) badl = badl | visbad(qacc_b[j]);
  const VB gbad = gsum8(sel(badl, V<int>(1), V<int>(0))) > 0;
  // unmasked stores (stage A's argument); an env with a bad qacc stores garbage here and is rewritten completely by the cold path below
  const VB okl = VB(true), okb = VB(true);
  wave_sync();
  // mj_advance: qacc_warmstart <- qacc ; qvel += h qacc_int ; qpos integrates the NEW velocity
  vr nv[6];
#pragma unroll
  for (int j = 0; j < 6; j++) nv[j] = LDG(qvel, j) + M.h * qint_b[j];
  vr np_[3], nq[4];
#pragma unroll
  for (int j = 0; j < 3; j++) np_[j] = LDG(qpos, j) + M.h * nv[j];
  {  // mju_quatIntegrate: q <- q * exp(h w), w in the body frame
    vr n = vsqrt(nv[3] * nv[3] + nv[4] * nv[4] + nv[5] * nv[5]);
    VB rot = !(n < vr(real(1e-15)));
    vr ns = sel(rot, n, vr(real(1)));
    vr ang = M.h * n, sn, cs;
    vsincos(real(0.5) * ang, &sn, &cs);
    rot = rot & (ang != vr(real(0)));
    const vr isn = vrcp(ns) * sn;
    vr qr[4] = {sel(rot, cs, vr(real(1))), sel(rot, nv[3] * isn, vr(real(0))), sel(rot, nv[4] * isn, vr(real(0))),
                sel(rot, nv[5] * isn, vr(real(0)))};
    vr a[4] = {LDG(qpos, 3), LDG(qpos, 4), LDG(qpos, 5), LDG(qpos, 6)};
    vr t[4] = {a[0] * qr[0] - a[1] * qr[1] - a[2] * qr[2] - a[3] * qr[3], a[0] * qr[1] + a[1] * qr[0] + a[2] * qr[3] - a[3] * qr[2],
               a[0] * qr[2] - a[1] * qr[3] + a[2] * qr[0] + a[3] * qr[1], a[0] * qr[3] + a[1] * qr[2] - a[2] * qr[1] + a[3] * qr[0]};
    vr nn = vsqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3]);  // mj_kinematics normalises before use
#pragma unroll
    for (int j = 0; j < 4; j++) nq[j] = t[j] * vrcp(nn);
  }
  vr jq[3], jv[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    jv[k] = LDL(qvel, leg * 3 + (6 + k)) + M.h * qint_l[k];
    jq[k] = LDL(qpos, leg * 3 + (7 + k)) + M.h * jv[k];
  }
  wave_sync();
  const V<int> eoV = eo + sel(pass1, V<int>(NM_OFS(qvel)), V<int>(NM_OFS(mbb)));
  const V<int> eoP = eo + sel(pass1, V<int>(NM_OFS(qpos)), V<int>(NM_OFS(sc)));
  const V<int> eoW = eo + sel(pass1, V<int>(NM_OFS(vv)), V<int>(NM_OFS(warm)));
#pragma unroll
  for (int j = 0; j < 6; j++) {
    stsv(lds, eoV + j, nv[j], okb);
    stsv(lds, eoW + j, qacc_b[j], okb);
  }
#pragma unroll
  for (int j = 0; j < 3; j++) stsv(lds, eoP + j, np_[j], okb);
#pragma unroll
  for (int j = 0; j < 4; j++) stsv(lds, eoP + (3 + j), nq[j], okb);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    stsv(lds, eoV + leg * 3 + (6 + k), jv[k], okl);
    stsv(lds, eoP + leg * 3 + (7 + k), jq[k], okl);
    stsv(lds, eoW + leg * 3 + (6 + k), qacc_l[k], okl);
  }
  wave_sync();
  // bad envs (rare): mj_resetData, then one free-fall step from qpos0 (no contacts at z0, ctrl = 0 -> qacc = (0,0,-g,0...))
  for (int e = 0; e < G; e++) {
    if (rdlane(sel(gbad, V<int>(1), V<int>(0)), 8 * e) != 0) {
      Sh<real>& sh = w.e[e];
      const V<int> lane = opaque_lane();   // cold path: keep its address math out of the hot loop's live ranges
      sh.nwarn += 1;
      vr q0 = ldsv(M.qpos0, sel(lane < kNQ, lane, V<int>(0)));
      real vz = -M.grav * M.h;
      q0 = sel(lane == 2, q0 + vz * M.h, q0);
      stsv(sh.qpos, lane, q0, lane < kNQ);
      stsv(sh.qvel, lane, sel(lane == 2, vr(vz), vr(real(0))), lane < kNV);
      stsv(sh.warm, lane, sel(lane == 2, vr(-M.grav), vr(real(0))), lane < kNV);
      stsv(sh.ctrl, lane, real(0), lane < kNU);
      wave_sync();
    }
  }
#undef LDG
#undef LDL
}

template <class real> NM_FN void reset_data(Sh<real>& sh, const Model<real>& M) {  // mj_resetData
  const V<int> lane = opaque_lane();
  stsv(sh.qpos, lane, ldsv(M.qpos0, sel(lane < kNQ, lane, V<int>(0))), lane < kNQ);
  stsv(sh.qvel, lane, real(0), lane < kNV);
  stsv(sh.warm, lane, real(0), lane < kNV);
  stsv(sh.ctrl, lane, real(0), lane < kNU);
  wave_sync();
}

// Critical-path scheduling between the two waves that share a SIMD for the whole launch (4096 envs = exactly 2 waves per SIMD, one
// round): the launch ends with its slowest wave, and what makes a wave slow is known as soon as its collision stage has run - the
// contact count (3.9 k cycles per contact, profiles/r02_wavetimes.txt). A wave with many contacts raises its issue priority, so that
// where both waves of the SIMD have an instruction ready the one on the critical path goes first; its lighter partner has the slack.
// Measured (MI355X, 4096 envs): 75.0 -> 72.5 us; the wave-lifetime regression's per-contact cost 3.8 k -> 2.2 k cycles, p99 158 k -> 150 k.
#ifndef NM_PRIO
#define NM_PRIO 3      /* contacts (both envs of the wave) from which the priority starts to rise; a large value switches it off */
#endif
#if !defined(NM_EMUL)
NM_FN void nm_set_priority(int load) {
  if (load >= NM_PRIO + 4) __builtin_amdgcn_s_setprio(3);
  else if (load >= NM_PRIO + 2) __builtin_amdgcn_s_setprio(2);
  else if (load >= NM_PRIO) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
#else
inline void nm_set_priority(int) {}
#endif

// mj_step(model, data, 1) for the G envs of the wave
template <class real, int G> NM_FN void substep(ShW<real, G>& w, const Model<real>& M, bool last, int* dropped, int ablate) {
  const V<int> lane = opaque_lane();
  if constexpr (G == 2) {   // both envs at once on half-waves (lanes 0..31 | 32..63), as in env_load2
    constexpr int kSR = (int)(sizeof(Sh<real>) / sizeof(real));
    real* rbw = reinterpret_cast<real*>(&w.e[0]);
    const V<int> hl = lane & 31, ho = (lane >> 5) * kSR;
    {  // mj_checkPos / mj_checkVel
      const VB bad = (visbad(ldsv(rbw, ho + (sel(hl < kNQ, hl, V<int>(0)) + NM_OFS(qpos)))) & (hl < kNQ)) |
                     (visbad(ldsv(rbw, ho + (sel(hl < kNV, hl, V<int>(0)) + NM_OFS(qvel)))) & (hl < kNV));
      const uint64_t bm = ballot(bad);
      if (bm) {
#pragma unroll
        for (int e = 0; e < 2; e++)
          if ((bm >> (32 * e)) & 0xffffffffull) { w.e[e].nwarn += 1; reset_data(w.e[e], M); }
      }
    }
    // (mj_kinematics' normalisation of the free joint's quaternion happens at the head of stage A, in registers)
  } else {
    for (int e = 0; e < G; e++) {
      Sh<real>& sh = w.e[e];
      {  // mj_checkPos / mj_checkVel
        VB bad = (visbad(ldsv(sh.qpos, sel(lane < kNQ, lane, V<int>(0)))) & (lane < kNQ)) |
                 (visbad(ldsv(sh.qvel, sel(lane < kNV, lane, V<int>(0)))) & (lane < kNV));
        if (wany(bad)) { sh.nwarn += 1; reset_data(sh, M); }
      }
    }
  }
  wave_sync();
  nm_stamp(1);
  if (!(ablate & 8)) stage_smooth(w, M, last);
  nm_stamp(2);
  // collision of every env of the wave, then the constraint stage: both envs in one pass when each has 1..kMaxCon2 floor contacts
  // (stage_constraint2), else one after the other
  for (int e = 0; e < G; e++) {
    Sh<real>& sh = w.e[e];
    if (!(ablate & 1)) { Rings<real> rg; collide_rings(sh, M, rg); stage_collide(sh, M, dropped, rg, G != 2 && !(ablate & 16)); }
    else { sh.ncon = 0; sh.anypair = 0; wave_sync(); }
  }
  if constexpr (G == 2) {
    if (!(ablate & 1) && !(ablate & 16)) stage_collide_pairs2(w, M, dropped);   // tibia pairs: the cull of both envs in one pass
  }
  for (int e = 0; e < G; e++)
    if (ablate & 4) { w.e[e].ncon = 0; w.e[e].anypair = 0; wave_sync(); }
  bool together = false;
  if constexpr (G == 2) {
    const int n0 = uniform(w.e[0].ncon), n1 = uniform(w.e[1].ncon);
    nm_set_priority(n0 + n1);      // (counting the support search's hops / fallbacks as well measured no better: 68.7 vs 68.8-69.4 us)
    together = n0 + n1 >= 1 && n0 <= kMaxCon2 && n1 <= kMaxCon2 && uniform(w.e[0].anypair) == 0 && uniform(w.e[1].anypair) == 0 && !(ablate & 32);
    if (!last) w.e[1].ntog = n0 * 64 + n1;   // debug buffer only: the first substep's contact counts
    if (together) {
#ifdef NM_EMUL
      nm_emul_together() += 1;
#endif
      w.e[0].ntog += 1;
      stage_constraint2(w, M, last, (ablate & 2) != 0);
      nm_stamp(8);
    }
  }
  if (!together)
    for (int e = 0; e < G; e++) {
      stage_constraint(w.e[e], w.jrow, M, last, (ablate & 2) != 0);
      nm_stamp(8);
    }
  stage_integrate(w, M);
  nm_stamp(9);
}

// =========================================================================================  env step

// load one env's state into its LDS image, action -> servo command (E1)
template <class real> NM_FN void env_load(Sh<real>& sh, const Model<real>& M, const Args<real>& A, int env) {
  typedef V<real> vr;
  const V<int> lane = opaque_lane();   // index math stays local to this function (not kept live across the physics)
  const VB l18 = lane < kNU;
  const V<int> l18c = sel(l18, lane, V<int>(0));
  // ---- every HBM read of this env-step is issued here, back to back, before the first LDS store (a masked store is a
  // branch: loads placed after one would each wait for their own round trip)
  const vr q_in = gldv(A.qpos, sel(lane < kNQ, lane, V<int>(0)) + env * kNQ);
  const vr v_in = gldv(A.qvel, sel(lane < kNV, lane, V<int>(0)) + env * kNV);
  const vr w_in = gldv(A.qwarm, sel(lane < kNV, lane, V<int>(0)) + env * kNV);
  const V<int> hc_in = gldv(A.hullcache, sel(lane < 8, lane, V<int>(0)) + env * 8);
  const V<float> a_in = gldv(A.actions, l18c + env * kNU);
  vr cmd_in = vr(real(0)), eps_in = vr(real(0)), prev_act = vr(real(0)), prev_dofvel = vr(real(0)), dofpos_old = vr(real(0));
  int64_t ep = 0;
  uint32_t ctr_in = 0;
  if (!A.physics_only) {   // what the epilogue needs from HBM is fetched now, under the physics, not when it is needed
    cmd_in = gldv(A.cmd, sel(lane < 3, lane, V<int>(0)) + env * 3);
    eps_in = gldv(A.epsum, sel(lane < kNREW, lane, V<int>(0)) + env * kNREW);
    ep = gld1(A.eplen, env);
    ctr_in = gld1(A.rngctr, env);
    prev_act = gldv(A.act, l18c + env * kNU);
    prev_dofvel = gldv(A.dofvel, l18c + env * kNU);
    dofpos_old = gldv(A.dofpos, l18c + env * kNU);
  }
  stsv(sh.qpos, lane, q_in, lane < kNQ);
  stsv(sh.qvel, lane, v_in, lane < kNV);
  stsv(sh.warm, lane, w_in, lane < kNV);
  stsv(sh.hcache, lane, sel(lane == 0, hc_in & V<int>(0xffff), hc_in), lane < 8);     // bits 16.. of [0]: load hint of the two-env build
  sh.nwarn = 0;
  sh.nhop = 0;
  sh.ntog = 0;
  // ---- E1 (env.py:152-156,181-192): float32 scale + clip; PD -> velocity command from the env's own dof_pos buffer
  V<float> af = a_in * M.action_scale;
  af = vmin(vmax(af, V<float>(-M.clip_actions)), V<float>(M.clip_actions));
  vr act;
  vr defp;
  {
    V<int> m3 = lane % 3;
    defp = sel(m3 == 1, vr(M.default_pos[1]), sel(m3 == 0, vr(M.default_pos[0]), vr(M.default_pos[2])));
  }
#ifdef NM_EMUL
  for (int i = 0; i < NM_WAVE; i++) act.v[i] = (real)af.v[i];
#else
  act = (real)af;
#endif
  if (!A.physics_only) {
    stsv(sh.ecmd, lane, cmd_in, lane < 3);
    stsv(sh.eepsum, lane, eps_in, lane < kNREW);
    sh.eplen_lo = (int)(uint32_t)(ep & 0xffffffffll);
    sh.eplen_hi = (int)(ep >> 32);
    sh.ectr = ctr_in;
    stsv(sh.ctrl, lane, ((act - defp) - dofpos_old) * M.p_gain, l18);
  } else {  // dynamics-only mode (BASELINE config 2): same PD law on the current joint angles, no env buffers
    wave_sync();
    stsv(sh.ctrl, lane, ((act - defp) - ldsv(sh.qpos, l18c + 7)) * M.p_gain, l18);
  }
  // carried to the epilogue through LDS, not in registers: nothing stays live across the physics
  stsv(sh.eact, lane, act, l18); stsv(sh.epact, lane, prev_act, l18); stsv(sh.epdv, lane, prev_dofvel, l18);
  wave_sync();
}

// debug buffer of one env (measurement / tests only)
template <class real> NM_FN void env_debug(Sh<real>& sh, const Args<real>& A, int env, int dropped) {
  const V<int> lane = opaque_lane();
  {
    real* dbg = A.dbg + (size_t)env * kDbgN;
    gstv(dbg, lane, ldsv(sh.qas, sel(lane < 24, lane, V<int>(0))), lane < 24);
    gstv(dbg, lane + 24, ldsv(sh.qfs, sel(lane < 24, lane, V<int>(0))), lane < 24);
    gstv(dbg, lane + 48, ldsv(sh.qfc, sel(lane < 24, lane, V<int>(0))), lane < 24);
    gstv(dbg, lane + 72, ldsv(sh.sens, sel(lane < 13, lane, V<int>(0))), lane < 13);
    gstv(dbg, lane + 88, ldsv(sh.cvb, sel(lane < 6, lane, V<int>(0))), lane < 6);
    gstv(dbg, lane + 96, ldsv(sh.cdist(), sel(lane < kMaxCon, lane, V<int>(0))), lane < kMaxCon);
    gstv(dbg, lane + 112, ldsv(sh.cpos(), sel(lane < 3 * kMaxCon, lane, V<int>(0))), lane < 3 * kMaxCon);
    gstv(dbg, V<int>(160), to_real<real>(sh.ncon), lane == 0);
    gstv(dbg, V<int>(161), to_real<real>(sh.nwarn), lane == 0);
    gstv(dbg, V<int>(162), to_real<real>(dropped), lane == 0);
    gstv(dbg, lane + 165, ldsv(sh.cnrm(), sel(lane < 6, lane, V<int>(0))), lane < 6);
    gstv(dbg, lane + 150, to_real<real>(ldsv(sh.hcache, sel(lane < 7, lane, V<int>(0)))), lane < 7);
    gstv(dbg, V<int>(158), to_real<real>(sh.hcache[7]), lane == 0);
    gstv(dbg, V<int>(159), to_real<real>(sh.nhop), lane == 0);
    gstv(dbg, V<int>(157), to_real<real>(sh.anypair), lane == 0);
    gstv(dbg, V<int>(156), to_real<real>(sh.ntog), lane == 0);
    gstv(dbg, V<int>(163), to_real<real>(sh.it_pgs), lane == 0);
    gstv(dbg, V<int>(164), to_real<real>(sh.it_noslip), lane == 0);
    gstv(dbg, lane + 176, ldsv(sh.efc_f, lane), lane < kMaxRow);
#ifdef NM_DEBUG_SOLVER
    gstv(dbg, lane + 240, ldsv(sh.dbg_b, sel(lane < 4, lane, V<int>(0))), lane < 4);
    gstv(dbg, lane + 244, ldsv(sh.dbg_a, sel(lane < 4, lane, V<int>(0))), lane < 4);
    gstv(dbg, lane + 248, ldsv(sh.dbg_f0, sel(lane < 4, lane, V<int>(0))), lane < 4);
#endif
  }
}

// store one env's state, run the env epilogue (E3-E8). `live` = the slot holds a real env (last wave may be padded)
template <class real> NM_FN void env_finish(Sh<real>& sh, const Model<real>& M, const Args<real>& A, int env,
                                            int dropped, bool live) {
  typedef V<real> vr;
  if (!live) return;
  const V<int> lane = opaque_lane();
  const VB l18 = lane < kNU;
  const V<int> l18c = sel(l18, lane, V<int>(0));
  const vr act = ldsv(sh.eact, l18c), prev_act = ldsv(sh.epact, l18c), prev_dofvel = ldsv(sh.epdv, l18c);
  vr defp;
  {
    V<int> m3 = lane % 3;
    defp = sel(m3 == 1, vr(M.default_pos[1]), sel(m3 == 0, vr(M.default_pos[0]), vr(M.default_pos[2])));
  }
  // ---- store physics state
  {  // LDS reads first, then the (branchy, masked) global stores: one LDS round trip for the four of them
    const V<int> hc = ldsv(sh.hcache, sel(lane < 8, lane, V<int>(0)));
    const vr q = ldsv(sh.qpos, sel(lane < kNQ, lane, V<int>(0))), v = ldsv(sh.qvel, sel(lane < kNV, lane, V<int>(0))),
             w = ldsv(sh.warm, sel(lane < kNV, lane, V<int>(0)));
    gstv(A.hullcache, lane + env * 8, hc, lane < 8);
    gstv(A.qpos, lane + env * kNQ, q, lane < kNQ);
    gstv(A.qvel, lane + env * kNV, v, lane < kNV);
    gstv(A.qwarm, lane + env * kNV, w, lane < kNV);
  }
  if (A.rec && env == A.rec_env) {
    gstv(A.rec, lane, ldsv(sh.qpos, sel(lane < kNQ, lane, V<int>(0))), lane < kNQ);
    gstv(A.rec, lane + kNQ, ldsv(sh.qvel, sel(lane < kNV, lane, V<int>(0))), lane < kNV);
    gstv(A.rec, V<int>(kNQ + kNV), to_real<real>(sh.nwarn), lane == 0);
  }
  if (A.dbg) env_debug(sh, A, env, dropped);
  if ((dropped | sh.nwarn) && A.stat_cnt) {
#ifdef NM_EMUL
    A.stat_cnt[1] += dropped; A.stat_cnt[2] += sh.nwarn;
#else
    if (NM_TID == 0) { nm_consume(atomicAdd(A.stat_cnt + 1, dropped)); nm_consume(atomicAdd(A.stat_cnt + 2, sh.nwarn)); }
#endif
  }
  if (A.physics_only) return;
  NM_ESTAMP(11);

  // ---- E3 (env.py:212-232): frame transforms with the POST-integration quaternion, stale cvel/sensors
  int64_t eplen = (((int64_t)sh.eplen_hi << 32) | (int64_t)(uint32_t)sh.eplen_lo) + 1;
  real bq[4] = {sh.qpos[3], -sh.qpos[4], -sh.qpos[5], -sh.qpos[6]};  // mju_negQuat
  auto rot = [&](real* r, const real* v) {  // mju_rotVecQuat
    real tx = real(2) * (bq[2] * v[2] - bq[3] * v[1]), ty = real(2) * (bq[3] * v[0] - bq[1] * v[2]), tz = real(2) * (bq[1] * v[1] - bq[2] * v[0]);
    r[0] = v[0] + bq[0] * tx + (bq[2] * tz - bq[3] * ty);
    r[1] = v[1] + bq[0] * ty + (bq[3] * tx - bq[1] * tz);
    r[2] = v[2] + bq[0] * tz + (bq[1] * ty - bq[2] * tx);
  };
  real blv[3], bav[3], pg[3];
  {
    real lin[3] = {sh.cvb[3], sh.cvb[4], sh.cvb[5]}, ang[3] = {sh.cvb[0], sh.cvb[1], sh.cvb[2]}, gv[3] = {real(0), real(0), -M.grav};
    rot(blv, lin); rot(bav, ang); rot(pg, gv);
  }
  vr dofpos = ldsv(sh.qpos, l18c + 7), dofvel = ldsv(sh.qvel, l18c + 6);
  real tib[6], feet[6], body = sh.sens[12];
#pragma unroll
  for (int l = 0; l < 6; l++) {
    feet[l] = sh.sens[6 + l];
    tib[l] = feet[l] == real(0) ? sh.sens[l] : real(0);  // env.py:232
  }
  // ---- E4 (env.py:235-236, 321-333): periodic command resample
  real cmd[3] = {sh.ecmd[0], sh.ecmd[1], sh.ecmd[2]};
  uint32_t ctr = sh.ectr;
  auto resample = [&](int which) {
    real ux, uy;
    if (A.cmd_u) { ux = gld1(A.cmd_u, env * 4 + 2 * which); uy = gld1(A.cmd_u, env * 4 + 2 * which + 1); }
    else {
      ux = (real)rand_u24_bits(A.seed, (uint64_t)(A.env_offset + env), ctr) * real(1.0 / 16777216.0);
      uy = (real)rand_u24_bits(A.seed, (uint64_t)(A.env_offset + env), ctr + 1) * real(1.0 / 16777216.0);
      ctr += 2;
    }
    cmd[0] = ux * real(2) * M.max_lin_x - M.max_lin_x;
    cmd[1] = real(0);
    cmd[2] = uy * real(2) * M.max_ang - M.max_ang;
    real keep = vsqrt(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > real(0.02) ? real(1) : real(0);
    cmd[0] = cmd[0] * keep; cmd[1] = cmd[1] * keep;
  };
  {
    bool due;
    if (eplen >= 0 && eplen < (1 << 22)) {   // episode lengths are a few thousand steps: float quotient + exact integer remainder
      const int e32 = (int)eplen, R = M.resample_every;
      const int q = (int)((float)e32 / (float)R);
      const int r = e32 - q * R;             // q is off by at most one
      due = r == 0 || r == R || r == -R;
    } else due = eplen % M.resample_every == 0;
    if (due) resample(0);
  }
  // ---- E5 (env.py:239-258): termination
  bool time_out = (real)eplen > M.max_ep_len;
  bool reset = time_out;
  {
    real fm = feet[0];
#pragma unroll
    for (int l = 1; l < 6; l++) fm = vmax(fm, feet[l]);
    reset = reset | (fm > M.term_force);
    if ((M.tibia_mode == 2) | (M.body_mode == 2)) {   // env.py:248-251: terminate on tibia / body contact (off in the default config)
      real tm = tib[0];
#pragma unroll
      for (int l = 1; l < 6; l++) tm = vmax(tm, tib[l]);
      if (M.tibia_mode == 2) reset = reset | (tm > M.tibia_max);
      if (M.body_mode == 2) reset = reset | (body > M.body_max);
    }
    real nrm = vsqrt(pg[0] * pg[0] + pg[1] * pg[1] + pg[2] * pg[2]);
    // angle(projected gravity, straight down) > 60 deg. The fp32 build compares cosines instead of calling acosf (equivalent
    // except within an ulp of the threshold); the fp64 verification build keeps the reference's expression.
    if (sizeof(real) == 8) reset = reset | (vacos(-pg[2] / nrm) > real(1.0471975511965976));
    else reset = reset | (-pg[2] / nrm <= real(0.5));
  }
  NM_ESTAMP(12);
  // ---- E6 (env.py:274, 335-371): reset BEFORE rewards/obs: qpos0, zero velocity, new command, episode stats
  vr epsum[1];
  V<int> l8c = sel(lane < kNREW, lane, V<int>(0));
  epsum[0] = ldsv(sh.eepsum, l8c);
  if (reset) {
    gstv(A.qpos, lane + env * kNQ, ldsv(M.qpos0, sel(lane < kNQ, lane, V<int>(0))), lane < kNQ);
    gstv(A.qvel, lane + env * kNV, real(0), lane < kNV);
    resample(1);
    eplen = 0;
#ifdef NM_EMUL
    for (int k = 0; k < kNREW; k++) A.stat_sum[k] += epsum[0].v[k];
    A.stat_cnt[0] += 1;
#else
    if (NM_TID < kNREW) nm_consume(atomicAdd(A.stat_sum + NM_TID, epsum[0]));
    if (NM_TID == 0) {
      nm_consume(atomicAdd(A.stat_cnt, 1));
      if (time_out && A.to_list) nm_consume(atomicExch(A.to_list + atomicAdd(A.nto, 1), env));
    }
#endif
    epsum[0] = vr(real(0));
  }
  NM_ESTAMP(13);
  // ---- E7 (env.py:277-288, 399-497): rewards (alphabetical, termination last)
  real rt[kNREW];
#pragma unroll
  for (int k = 0; k < kNREW; k++) rt[k] = real(0);
  {
    vr da = prev_act - act;
    rt[R_ACTION_RATE] = wsum<real>(sel(l18, da * da, vr(real(0)))) * M.rew_scale[R_ACTION_RATE];
    real sumt = ((tib[0] + tib[1]) + (tib[2] + tib[3])) + (tib[4] + tib[5]);
    if ((M.tibia_mode != 1) | (M.body_mode != 1)) {   // env.py:479-485: each part counts only in mode 1
      sumt = M.tibia_mode == 1 ? sumt : real(0);
      body = M.body_mode == 1 ? body : real(0);
    }
    rt[R_BODY_CONTACT] = (sumt + body) * M.rew_scale[R_BODY_CONTACT];
    vr dp = dofpos - defp;
    rt[R_DEFAULT_POS] = wsum<real>(sel(l18, dp * dp, vr(real(0)))) * M.rew_scale[R_DEFAULT_POS];
    vr acc = (dofvel - prev_dofvel) / M.dt;
    rt[R_DOF_ACC] = wsum<real>(sel(l18, acc * acc, vr(real(0)))) * M.rew_scale[R_DOF_ACC];
    rt[R_ORIENTATION] = (pg[0] * pg[0] + pg[1] * pg[1]) * M.rew_scale[R_ORIENTATION];
    real ea = (cmd[2] - bav[2]) * (cmd[2] - bav[2]);
    rt[R_TRACK_ANG] = vexp(-ea / M.sigma) * M.rew_scale[R_TRACK_ANG];
    real el = (cmd[0] - blv[0]) * (cmd[0] - blv[0]) + (cmd[1] - blv[1]) * (cmd[1] - blv[1]);
    rt[R_TRACK_LIN] = vexp(-el / M.sigma) * M.rew_scale[R_TRACK_LIN];
    rt[R_TERMINATION] = (reset && !time_out) ? M.rew_scale[R_TERMINATION] : real(0);
    // rt[R_TORQUES] stays 0: env.py:415-417 squares qfrc_applied[-18:] (env.py:222), which nothing ever writes
    if (M.rew_extra) {   // the terms config.py:88-95 ships with scale 0: one uniform branch in the default configuration
      rt[R_ANG_VEL_XY] = (bav[0] * bav[0] + bav[1] * bav[1]) * M.rew_scale[R_ANG_VEL_XY];                  // env.py:403-405
      real dh = sh.bh[0] - M.base_h_target;
      rt[R_BASE_HEIGHT] = dh * dh * M.rew_scale[R_BASE_HEIGHT];                                            // env.py:411-413
      rt[R_DOF_VEL] = wsum<real>(sel(l18, dofvel * dofvel, vr(real(0)))) * M.rew_scale[R_DOF_VEL];           // env.py:419-421
      rt[R_LIN_VEL_Z] = blv[2] * blv[2] * M.rew_scale[R_LIN_VEL_Z];                                        // env.py:399-401
      real still = vsqrt(cmd[0] * cmd[0] + cmd[1] * cmd[1]) < real(0.01) ? real(1) : real(0);
      rt[R_STAND_STILL] = wsum<real>(sel(l18, vabs(dp), vr(real(0)))) * still * M.rew_scale[R_STAND_STILL];  // env.py:487-489
      const VB l6 = lane < kNLEG;
      const V<int> l6c = sel(l6, lane, V<int>(0));
      const vr ff = ldsv(sh.sens, l6c + 6);
      vr over = sel(ff > vr(M.max_contact_force), ff - M.max_contact_force, vr(real(0)));
      rt[R_FEET_CONTACT] = wsum<real>(sel(l6, over * over, vr(real(0)))) * M.rew_scale[R_FEET_CONTACT];      // env.py:491-493
      if (M.rew_scale[R_FEET_AIR_TIME] != real(0)) {   // env.py:458-477; stateful: runs only while it is in the reward table
        vr air = gldv(A.feetair, l6c + env * kNLEG);
        const V<int> fl = V<int>(gld1(A.feetflags, env));
        if (reset) air = vr(real(0));                  // reset_idx zeroes feet_air_time (env.py:359) before the rewards run
        const VB contact = ff > vr(real(1));
        const VB filt = contact | (((fl >> l6c) & 1) != 0);
        const VB lastf = ((fl >> (l6c + kNLEG)) & 1) != 0;
        air = air + M.dt;
        air = sel(filt == lastf, air, vr(real(0)));    // reset air time if the filtered contact changes
        const vr single = sel(air > vr(real(1)), air - real(1), vr(real(0))) + sel(air < vr(real(0.5)), real(0.5) - air, vr(real(0)));
        rt[R_FEET_AIR_TIME] = wsum<real>(sel(l6, single * single, vr(real(0)))) * M.rew_scale[R_FEET_AIR_TIME];
        gstv(A.feetair, l6c + env * kNLEG, air, l6);
        const int nf = (int)(ballot(l6 & contact) | (ballot(l6 & filt) << kNLEG));
#ifdef NM_EMUL
        A.feetflags[env] = nf;
#else
        if (NM_TID == 0) gst1(A.feetflags, env, nf);
#endif
      }
    }
  }
  real rew = real(0);
#pragma unroll
  for (int k = 0; k < kNREW; k++) rew = rew + rt[k];
  {
    vr add = vr(real(0));
#pragma unroll
    for (int k = 0; k < kNREW; k++) add = wrlane(add, rt[k], k);
    gstv(A.epsum, lane + env * kNREW, epsum[0] + add, lane < kNREW);
  }
  NM_ESTAMP(14);
  // ---- E8 (env.py:291-311): observation (66), clipped, float32
  {
    real head[12] = {blv[0] * M.obs_lin, blv[1] * M.obs_lin, blv[2] * M.obs_lin, bav[0] * M.obs_ang, bav[1] * M.obs_ang, bav[2] * M.obs_ang,
                     pg[0], pg[1], pg[2], cmd[0] * M.obs_lin, cmd[1] * M.obs_lin, cmd[2] * M.obs_ang};
    vr o = vr(real(0));
#pragma unroll
    for (int k = 0; k < 12; k++) o = wrlane(o, head[k], k);
    auto put = [&](const vr& x0, const V<int>& idx, const VB& m) {
      vr x = x0;
      if (A.noise_vec) {  // env.py:304-305: + (2 U[0,1) - 1) * noise_scale_vec before the clip
        auto nz = [&](int k) {
          real u = A.noise_u ? gld1(A.noise_u, (size_t)env * kNOBS + k)
                             : (real)rand_u24_bits(A.seed + kNoiseKey, (uint64_t)(A.env_offset + env), (uint32_t)(A.noise_step * kNOBS + k)) * real(1.0 / 16777216.0);
          return (real(2) * u - real(1)) * gld1(A.noise_vec, k);
        };
#ifdef NM_EMUL
        for (int i = 0; i < NM_WAVE; i++) if (m.v[i]) x.v[i] = x.v[i] + nz(idx.v[i]);
#else
        if (m) x = x + nz(idx);
#endif
      }
      vr cx = vmin(vmax(x, vr(-M.clip_obs)), vr(M.clip_obs));
#ifdef NM_EMUL
      for (int i = 0; i < NM_WAVE; i++) if (m.v[i]) A.obs[(size_t)env * kNOBS + idx.v[i]] = (float)cx.v[i];
#else
      if (m) gst1(A.obs, (size_t)env * kNOBS + idx, (float)cx);
#endif
    };
    put(o, lane, lane < 12);
    put((dofpos - defp) * M.obs_dofpos, lane + 12, l18);
    put(dofvel * M.obs_dofvel, lane + 30, l18);
    put(act, lane + 48, l18);
  }
  NM_ESTAMP(15);
  // ---- buffers the reference keeps between steps
  gstv(A.dofpos, lane + env * kNU, dofpos, l18);
  gstv(A.dofvel, lane + env * kNU, dofvel, l18);
  gstv(A.act, lane + env * kNU, act, l18);
  {
    vr cv = sel(lane == 0, vr(cmd[0]), sel(lane == 1, vr(cmd[1]), vr(cmd[2])));
    gstv(A.cmd, lane + env * 3, cv, lane < 3);
  }
#ifdef NM_EMUL
  const bool lane0 = true;
#else
  const bool lane0 = NM_TID == 0;
#endif
  if (lane0) {
    gst1(A.eplen, env, eplen);
    gst1(A.rngctr, env, ctr);
    gst1(A.rew, env, (float)rew);
    if (A.ret_acc) gadd1(A.ret_acc, env, (float)rew);
    gst1(A.done, env, (int64_t)(reset ? 1 : 0));
    gst1(A.timeout_now, env, time_out ? 1.0f : 0.0f);
  }
}

// ---- G = 2: both envs of the wave at once, lanes 0..31 for env 0 and lanes 32..63 for env 1. The same instruction stream as
// env_load / env_finish, executed once instead of twice: per-env scalars become per-lane values that are equal inside a half,
// LDS addresses carry the half's image offset, sums over joints are half-wave sums (hsum32). What is genuinely scalar (episode
// length as int64, the command RNG) stays scalar, once per half. Every HBM read of BOTH envs is in flight before the first wait.
// `mid()` runs after every HBM read has been issued and before the first of them is waited for: the kernel copies the model constants
// (L2 -> LDS, `M` is that copy) there, so the two round trips of a wave's start-up overlap instead of following each other.
struct NoMid { NM_FN void operator()() const {} NM_FN void operator()(int) const {} };
template <class real, class Mid> NM_FN void env_load2(ShW<real, 2>& w, const Model<real>& M, const Args<real>& A, int wave, Mid&& mid) {
  typedef V<real> vr;
  constexpr int kSR = (int)(sizeof(Sh<real>) / sizeof(real)), kSI = (int)(sizeof(Sh<real>) / sizeof(int));
  const V<int> lane = opaque_lane();
  const V<int> h = lane >> 5, hl = lane & 31;
  const V<int> env0 = h + wave * 2;
  const V<int> env = sel(env0 < V<int>(A.N), env0, V<int>(A.N - 1));   // a padded slot recomputes the last env; nothing of it is stored
  real* rb = reinterpret_cast<real*>(&w.e[0]);
  int* ib = reinterpret_cast<int*>(&w.e[0]);
  const V<int> ho = h * kSR, hoi = h * kSI;
  const VB l18 = hl < kNU;
  const V<int> l18c = sel(l18, hl, V<int>(0));
  const vr q_in = gldv(A.qpos, sel(hl < kNQ, hl, V<int>(0)) + env * kNQ);
  const vr v_in = gldv(A.qvel, sel(hl < kNV, hl, V<int>(0)) + env * kNV);
  const vr w_in = gldv(A.qwarm, sel(hl < kNV, hl, V<int>(0)) + env * kNV);
  const V<int> hc_in = gldv(A.hullcache, sel(hl < 8, hl, V<int>(0)) + env * 8);
  const V<float> a_in = gldv(A.actions, l18c + env * kNU);
  vr cmd_in = vr(real(0)), eps_in = vr(real(0)), prev_act = vr(real(0)), prev_dofvel = vr(real(0)), dofpos_old = vr(real(0));
#ifndef NM_EMUL
  int64_t ep = 0;
  uint32_t ctr_in = 0;
#endif
  if (!A.physics_only) {
    cmd_in = gldv(A.cmd, sel(hl < 3, hl, V<int>(0)) + env * 3);
    eps_in = gldv(A.epsum, sel(hl < kNREW, hl, V<int>(0)) + env * kNREW);
#ifndef NM_EMUL
    ep = gld1(A.eplen, (size_t)env);
    ctr_in = gld1(A.rngctr, (size_t)env);
#endif
    prev_act = gldv(A.act, l18c + env * kNU);
    prev_dofvel = gldv(A.dofvel, l18c + env * kNU);
    dofpos_old = gldv(A.dofpos, l18c + env * kNU);
  }
  mid();
  stsv(rb, ho + (hl + NM_OFS(qpos)), q_in, hl < kNQ);
  stsv(rb, ho + (hl + NM_OFS(qvel)), v_in, hl < kNV);
  stsv(rb, ho + (hl + NM_OFS(warm)), w_in, hl < kNV);
  // hullcache[0] carries, above bit 16, the env's contact count at the end of the previous step (see env_finish2): the wave's issue
  // priority from its first instruction on (contact counts of consecutive steps correlate), refined after every collision stage
  stsv(ib, hoi + (hl + NM_IOFS(hcache)), sel(hl == 0, hc_in & V<int>(0xffff), hc_in), hl < 8);
  nm_set_priority((rdlane(hc_in, 0) >> 16) + (rdlane(hc_in, 32) >> 16));
  w.e[0].nwarn = 0; w.e[1].nwarn = 0;
  w.e[0].nhop = 0; w.e[1].nhop = 0;
  w.e[0].ntog = 0; w.e[1].ntog = 0;
  // ---- E1 (env.py:152-156,181-192): float32 scale + clip; PD -> velocity command from the env's own dof_pos buffer
  V<float> af = a_in * M.action_scale;
  af = vmin(vmax(af, V<float>(-M.clip_actions)), V<float>(M.clip_actions));
  vr act;
  vr defp;
  {
    V<int> m3 = hl % 3;
    defp = sel(m3 == 1, vr(M.default_pos[1]), sel(m3 == 0, vr(M.default_pos[0]), vr(M.default_pos[2])));
  }
#ifdef NM_EMUL
  for (int i = 0; i < NM_WAVE; i++) act.v[i] = (real)af.v[i];
#else
  act = (real)af;
#endif
  if (!A.physics_only) {
    stsv(rb, ho + (hl + NM_OFS(ecmd)), cmd_in, hl < 3);
    stsv(rb, ho + (hl + NM_OFS(eepsum)), eps_in, hl < kNREW);
#ifdef NM_EMUL
    for (int hh = 0; hh < 2; hh++) {
      const int e = env.v[32 * hh];
      const int64_t ep = A.eplen[e];
      w.e[hh].eplen_lo = (int)(uint32_t)(ep & 0xffffffffll);
      w.e[hh].eplen_hi = (int)(ep >> 32);
      w.e[hh].ectr = A.rngctr[e];
    }
#else
    stsv(ib, hoi + NM_IOFS(eplen_lo), (int)(uint32_t)(ep & 0xffffffffll), hl == 0);
    stsv(ib, hoi + NM_IOFS(eplen_hi), (int)(ep >> 32), hl == 0);
    stsv(ib, hoi + NM_IOFS(ectr), (int)ctr_in, hl == 0);
#endif
    stsv(rb, ho + (hl + NM_OFS(ctrl)), ((act - defp) - dofpos_old) * M.p_gain, l18);
  } else {
    wave_sync();
    stsv(rb, ho + (hl + NM_OFS(ctrl)), ((act - defp) - ldsv(rb, ho + (l18c + (7 + NM_OFS(qpos))))) * M.p_gain, l18);
  }
  stsv(rb, ho + (hl + NM_OFS(eact)), act, l18); stsv(rb, ho + (hl + NM_OFS(epact)), prev_act, l18); stsv(rb, ho + (hl + NM_OFS(epdv)), prev_dofvel, l18);
  wave_sync();
}

// `published()` runs as soon as everything this wave contributes to the end-of-step bookkeeping (the consumed atomics of E6 and of
// the counters) is out: the kernel takes the wave's end-of-step ticket there, so the ticket's round trip is hidden under E7 / E8.
// `published(1)` comes again after E7, a microsecond later: the group ticket is back by then, and the wave that turns out to be the last of
// its group draws the top-level ticket there - under E8 and the buffer stores.
template <class real, class Pub> NM_FN void env_finish2(ShW<real, 2>& w, const Model<real>& M, const Args<real>& A, int wave, int dropped, Pub&& published) {
  typedef V<real> vr;
  constexpr int kSR = (int)(sizeof(Sh<real>) / sizeof(real)), kSI = (int)(sizeof(Sh<real>) / sizeof(int));
  const V<int> lane = opaque_lane();
  const V<int> h = lane >> 5, hl = lane & 31;
  const VB h1 = h != 0;
  const V<int> env = h + wave * 2;
  const VB live = env < V<int>(A.N);
  const V<int> envc = sel(live, env, V<int>(A.N - 1));
  const bool lives[2] = {wave * 2 < A.N, wave * 2 + 1 < A.N};
  const real* rb = reinterpret_cast<const real*>(&w.e[0]);
  const int* ib = reinterpret_cast<const int*>(&w.e[0]);
  const V<int> ho = h * kSR, hoi = h * kSI;
#define SHR(field, idx) ldsv(rb, ho + ((idx) + NM_OFS(field)))
#define SHI(field, idx) ldsv(ib, hoi + ((idx) + NM_IOFS(field)))
  const VB l18 = hl < kNU;
  const V<int> l18c = sel(l18, hl, V<int>(0));
  const vr act = SHR(eact, l18c), prev_act = SHR(epact, l18c), prev_dofvel = SHR(epdv, l18c);
  vr defp;
  {
    V<int> m3 = hl % 3;
    defp = sel(m3 == 1, vr(M.default_pos[1]), sel(m3 == 0, vr(M.default_pos[0]), vr(M.default_pos[2])));
  }
  // ---- store physics state
  const V<int> nwarn = SHI(nwarn, V<int>(0));
  {
    V<int> hc = SHI(hcache, sel(hl < 8, hl, V<int>(0)));
    hc = sel(hl == 0, hc | (SHI(ncon, V<int>(0)) << 16), hc);     // the load hint for the next step
    const vr q = SHR(qpos, sel(hl < kNQ, hl, V<int>(0))), v = SHR(qvel, sel(hl < kNV, hl, V<int>(0))), wq = SHR(warm, sel(hl < kNV, hl, V<int>(0)));
    gstv(A.hullcache, hl + env * 8, hc, live & (hl < 8));
    gstv(A.qpos, hl + env * kNQ, q, live & (hl < kNQ));
    gstv(A.qvel, hl + env * kNV, v, live & (hl < kNV));
    gstv(A.qwarm, hl + env * kNV, wq, live & (hl < kNV));
    if (A.rec) {
      const VB r = live & (env == V<int>(A.rec_env));
      gstv(A.rec, hl, q, r & (hl < kNQ));
      gstv(A.rec, hl + kNQ, v, r & (hl < kNV));
      gstv(A.rec, V<int>(kNQ + kNV), to_real<real>(nwarn), r & (hl == 0));
    }
  }
  if (A.dbg) {
    if (lives[0]) env_debug(w.e[0], A, wave * 2, dropped);
    if (lives[1]) env_debug(w.e[1], A, wave * 2 + 1, 0);
  }
  if (A.stat_cnt && (dropped != 0 || wany(live & (nwarn != 0)))) {
#ifdef NM_EMUL
    A.stat_cnt[1] += dropped;
    for (int hh = 0; hh < 2; hh++) if (lives[hh]) A.stat_cnt[2] += nwarn.v[32 * hh];
#else
    if (NM_TID == 0 && dropped) nm_consume(atomicAdd(A.stat_cnt + 1, dropped));
    if (hl == 0 && live && nwarn) nm_consume(atomicAdd(A.stat_cnt + 2, nwarn));
#endif
  }
  if (A.physics_only) return;
  if (NM_ABLATE(A.ablate) & 128) return;   // measurement only: epilogue inputs were loaded, nothing of E3-E8 runs
  NM_ESTAMP(11);

  // ---- E3 (env.py:212-232): frame transforms with the POST-integration quaternion, stale cvel/sensors
  int64_t ep[2];
  uint32_t ctr[2];
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {
    ep[hh] = (((int64_t)w.e[hh].eplen_hi << 32) | (int64_t)(uint32_t)w.e[hh].eplen_lo) + 1;
    ctr[hh] = w.e[hh].ectr;
  }
  vr bq[4] = {SHR(qpos, V<int>(3)), -SHR(qpos, V<int>(4)), -SHR(qpos, V<int>(5)), -SHR(qpos, V<int>(6))};  // mju_negQuat
  auto rot = [&](vr* r, const vr* v) {  // mju_rotVecQuat
    vr tx = real(2) * (bq[2] * v[2] - bq[3] * v[1]), ty = real(2) * (bq[3] * v[0] - bq[1] * v[2]), tz = real(2) * (bq[1] * v[1] - bq[2] * v[0]);
    r[0] = v[0] + bq[0] * tx + (bq[2] * tz - bq[3] * ty);
    r[1] = v[1] + bq[0] * ty + (bq[3] * tx - bq[1] * tz);
    r[2] = v[2] + bq[0] * tz + (bq[1] * ty - bq[2] * tx);
  };
  vr blv[3], bav[3], pg[3];
  {
    vr lin[3] = {SHR(cvb, V<int>(3)), SHR(cvb, V<int>(4)), SHR(cvb, V<int>(5))}, ang[3] = {SHR(cvb, V<int>(0)), SHR(cvb, V<int>(1)), SHR(cvb, V<int>(2))};
    vr gv[3] = {vr(real(0)), vr(real(0)), vr(-M.grav)};
    rot(blv, lin); rot(bav, ang); rot(pg, gv);
  }
  vr dofpos = SHR(qpos, l18c + 7), dofvel = SHR(qvel, l18c + 6);
  vr tib[6], feet[6], body = SHR(sens, V<int>(12));
#pragma unroll
  for (int l = 0; l < 6; l++) {
    feet[l] = SHR(sens, V<int>(6 + l));
    tib[l] = sel(feet[l] == vr(real(0)), SHR(sens, V<int>(l)), vr(real(0)));  // env.py:232
  }
  // ---- E4 (env.py:235-236, 321-333): periodic command resample - scalar, per half
  vr cmd[3] = {SHR(ecmd, V<int>(0)), SHR(ecmd, V<int>(1)), SHR(ecmd, V<int>(2))};
  auto resample = [&](int hh, int which) {
    const int e = wave * 2 + hh;
    real ux, uy;
    if (A.cmd_u) { ux = gld1(A.cmd_u, (size_t)(e * 4 + 2 * which)); uy = gld1(A.cmd_u, (size_t)(e * 4 + 2 * which + 1)); }
    else {
      ux = (real)rand_u24_bits(A.seed, (uint64_t)(A.env_offset + e), ctr[hh]) * real(1.0 / 16777216.0);
      uy = (real)rand_u24_bits(A.seed, (uint64_t)(A.env_offset + e), ctr[hh] + 1) * real(1.0 / 16777216.0);
      ctr[hh] += 2;
    }
    real c0 = ux * real(2) * M.max_lin_x - M.max_lin_x, c1 = real(0), c2 = uy * real(2) * M.max_ang - M.max_ang;
    real keep = vsqrt(c0 * c0 + c1 * c1) > real(0.02) ? real(1) : real(0);
    c0 = c0 * keep; c1 = c1 * keep;
    const VB mine = hh ? h1 : !h1;
    cmd[0] = sel(mine, vr(c0), cmd[0]); cmd[1] = sel(mine, vr(c1), cmd[1]); cmd[2] = sel(mine, vr(c2), cmd[2]);
  };
  bool tos[2];
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {
    bool due;
    if (ep[hh] >= 0 && ep[hh] < (1 << 22)) {   // episode lengths are a few thousand steps: float quotient + exact integer remainder
      const int e32 = (int)ep[hh], R = M.resample_every;
      const int q = (int)((float)e32 / (float)R);
      const int r = e32 - q * R;             // q is off by at most one
      due = r == 0 || r == R || r == -R;
    } else due = ep[hh] % M.resample_every == 0;
    if (due && lives[hh]) resample(hh, 0);
    tos[hh] = (real)ep[hh] > M.max_ep_len;
  }
  // ---- E5 (env.py:239-258): termination
  const VB time_out = (h1 & VB(tos[1])) | (!h1 & VB(tos[0]));
  VB reset = time_out;
  {
    vr fm = feet[0];
#pragma unroll
    for (int l = 1; l < 6; l++) fm = vmax(fm, feet[l]);
    reset = reset | (fm > vr(M.term_force));
    if ((M.tibia_mode == 2) | (M.body_mode == 2)) {   // env.py:248-251
      vr tm = tib[0];
#pragma unroll
      for (int l = 1; l < 6; l++) tm = vmax(tm, tib[l]);
      if (M.tibia_mode == 2) reset = reset | (tm > vr(M.tibia_max));
      if (M.body_mode == 2) reset = reset | (body > vr(M.body_max));
    }
    vr nrm = vsqrt(pg[0] * pg[0] + pg[1] * pg[1] + pg[2] * pg[2]);
    if (sizeof(real) == 8) reset = reset | (vacos(-pg[2] / nrm) > vr(real(1.0471975511965976)));
    else reset = reset | ((-pg[2] / nrm) <= vr(real(0.5)));
  }
  NM_ESTAMP(12);
  // ---- E6 (env.py:274, 335-371): reset BEFORE rewards/obs: qpos0, zero velocity, new command, episode stats
  vr epsum = SHR(eepsum, sel(hl < kNREW, hl, V<int>(0)));
  const VB resetL = reset & live;
  const uint64_t rmask = ballot(resetL);
  if (rmask) {
    gstv(A.qpos, hl + env * kNQ, ldsv(M.qpos0, sel(hl < kNQ, hl, V<int>(0))), resetL & (hl < kNQ));
    gstv(A.qvel, hl + env * kNV, real(0), resetL & (hl < kNV));
#pragma unroll
    for (int hh = 0; hh < 2; hh++)
      if ((rmask >> (32 * hh)) & 1) { resample(hh, 1); ep[hh] = 0; }
#ifdef NM_EMUL
    for (int hh = 0; hh < 2; hh++)
      if ((rmask >> (32 * hh)) & 1) {
        for (int k = 0; k < kNREW; k++) A.stat_sum[k] += epsum.v[32 * hh + k];
        A.stat_cnt[0] += 1;
      }
#else
    if (resetL && hl < kNREW) nm_consume(atomicAdd(A.stat_sum + hl, epsum));
    if (resetL && hl == 0) {
      nm_consume(atomicAdd(A.stat_cnt, 1));
      if (time_out && A.to_list) nm_consume(atomicExch(A.to_list + atomicAdd(A.nto, 1), env));
    }
#endif
    epsum = sel(resetL, vr(real(0)), epsum);
  }
  published(0);
  NM_ESTAMP(13);
  // ---- E7 (env.py:277-288, 399-497): rewards (alphabetical, termination last)
  vr rt[kNREW];
#pragma unroll
  for (int k = 0; k < kNREW; k++) rt[k] = vr(real(0));
  {
    vr da = prev_act - act;
    rt[R_ACTION_RATE] = hsum32(sel(l18, da * da, vr(real(0)))) * M.rew_scale[R_ACTION_RATE];
    vr sumt = ((tib[0] + tib[1]) + (tib[2] + tib[3])) + (tib[4] + tib[5]);
    if ((M.tibia_mode != 1) | (M.body_mode != 1)) {   // env.py:479-485: each part counts only in mode 1
      sumt = M.tibia_mode == 1 ? sumt : vr(real(0));
      body = M.body_mode == 1 ? body : vr(real(0));
    }
    rt[R_BODY_CONTACT] = (sumt + body) * M.rew_scale[R_BODY_CONTACT];
    vr dp = dofpos - defp;
    rt[R_DEFAULT_POS] = hsum32(sel(l18, dp * dp, vr(real(0)))) * M.rew_scale[R_DEFAULT_POS];
    vr acc = (dofvel - prev_dofvel) / M.dt;
    rt[R_DOF_ACC] = hsum32(sel(l18, acc * acc, vr(real(0)))) * M.rew_scale[R_DOF_ACC];
    rt[R_ORIENTATION] = (pg[0] * pg[0] + pg[1] * pg[1]) * M.rew_scale[R_ORIENTATION];
    vr ea = (cmd[2] - bav[2]) * (cmd[2] - bav[2]);
    rt[R_TRACK_ANG] = vexp(-ea / M.sigma) * M.rew_scale[R_TRACK_ANG];
    vr el = (cmd[0] - blv[0]) * (cmd[0] - blv[0]) + (cmd[1] - blv[1]) * (cmd[1] - blv[1]);
    rt[R_TRACK_LIN] = vexp(-el / M.sigma) * M.rew_scale[R_TRACK_LIN];
    rt[R_TERMINATION] = sel(reset & !time_out, vr(M.rew_scale[R_TERMINATION]), vr(real(0)));
    if (M.rew_extra) {   // the terms config.py:88-95 ships with scale 0: one uniform branch in the default configuration
      rt[R_ANG_VEL_XY] = (bav[0] * bav[0] + bav[1] * bav[1]) * M.rew_scale[R_ANG_VEL_XY];                  // env.py:403-405
      vr dh = SHR(bh, V<int>(0)) - M.base_h_target;
      rt[R_BASE_HEIGHT] = dh * dh * M.rew_scale[R_BASE_HEIGHT];                                            // env.py:411-413
      rt[R_DOF_VEL] = hsum32(sel(l18, dofvel * dofvel, vr(real(0)))) * M.rew_scale[R_DOF_VEL];              // env.py:419-421
      rt[R_LIN_VEL_Z] = blv[2] * blv[2] * M.rew_scale[R_LIN_VEL_Z];                                        // env.py:399-401
      vr still = sel(vsqrt(cmd[0] * cmd[0] + cmd[1] * cmd[1]) < vr(real(0.01)), vr(real(1)), vr(real(0)));
      rt[R_STAND_STILL] = hsum32(sel(l18, vabs(dp), vr(real(0)))) * still * M.rew_scale[R_STAND_STILL];     // env.py:487-489
      const VB l6 = hl < kNLEG;
      const V<int> l6c = sel(l6, hl, V<int>(0));
      const vr ff = SHR(sens, l6c + 6);
      vr over = sel(ff > vr(M.max_contact_force), ff - M.max_contact_force, vr(real(0)));
      rt[R_FEET_CONTACT] = hsum32(sel(l6, over * over, vr(real(0)))) * M.rew_scale[R_FEET_CONTACT];         // env.py:491-493
      if (M.rew_scale[R_FEET_AIR_TIME] != real(0)) {   // env.py:458-477; stateful: runs only while it is in the reward table
        vr air = gldv(A.feetair, l6c + envc * kNLEG);
        const V<int> fl = gldv(A.feetflags, envc);
        air = sel(reset, vr(real(0)), air);            // reset_idx zeroes feet_air_time (env.py:359) before the rewards run
        const VB contact = ff > vr(real(1));
        const VB filt = contact | (((fl >> l6c) & 1) != 0);
        const VB lastf = ((fl >> (l6c + kNLEG)) & 1) != 0;
        air = air + M.dt;
        air = sel(filt == lastf, air, vr(real(0)));    // reset air time if the filtered contact changes
        const vr single = sel(air > vr(real(1)), air - real(1), vr(real(0))) + sel(air < vr(real(0.5)), real(0.5) - air, vr(real(0)));
        rt[R_FEET_AIR_TIME] = hsum32(sel(l6, single * single, vr(real(0)))) * M.rew_scale[R_FEET_AIR_TIME];
        gstv(A.feetair, l6c + env * kNLEG, air, live & l6);
        const uint64_t bc = ballot(l6 & contact), bf = ballot(l6 & filt);
        const int nf0 = (int)((bc & 63) | ((bf & 63) << kNLEG)), nf1 = (int)(((bc >> 32) & 63) | (((bf >> 32) & 63) << kNLEG));
        gstv(A.feetflags, env, sel(h1, V<int>(nf1), V<int>(nf0)), live & (hl == 0));
      }
    }
  }
  vr rew = vr(real(0));
#pragma unroll
  for (int k = 0; k < kNREW; k++) rew = rew + rt[k];
  {
    vr add = vr(real(0));
#pragma unroll
    for (int k = 0; k < kNREW; k++) add = sel(hl == k, rt[k], add);
    gstv(A.epsum, hl + env * kNREW, epsum + add, live & (hl < kNREW));
  }
  published(1);
  NM_ESTAMP(14);
  // ---- E8 (env.py:291-311): observation (66), clipped, float32
  if (!(NM_ABLATE(A.ablate) & 256)) {
    vr head[12] = {blv[0] * M.obs_lin, blv[1] * M.obs_lin, blv[2] * M.obs_lin, bav[0] * M.obs_ang, bav[1] * M.obs_ang, bav[2] * M.obs_ang,
                   pg[0], pg[1], pg[2], cmd[0] * M.obs_lin, cmd[1] * M.obs_lin, cmd[2] * M.obs_ang};
    vr o = vr(real(0));
#pragma unroll
    for (int k = 0; k < 12; k++) o = sel(hl == k, head[k], o);
    auto put = [&](const vr& x0, const V<int>& idx, const VB& m0) {
      const VB m = m0 & live;
      vr x = x0;
      if (A.noise_vec) {  // env.py:304-305: + (2 U[0,1) - 1) * noise_scale_vec before the clip
        auto nz = [&](int k, int e) {
          real u = A.noise_u ? gld1(A.noise_u, (size_t)e * kNOBS + k)
                             : (real)rand_u24_bits(A.seed + kNoiseKey, (uint64_t)(A.env_offset + e), (uint32_t)(A.noise_step * kNOBS + k)) * real(1.0 / 16777216.0);
          return (real(2) * u - real(1)) * gld1(A.noise_vec, (size_t)k);
        };
#ifdef NM_EMUL
        for (int i = 0; i < NM_WAVE; i++) if (m.v[i]) x.v[i] = x.v[i] + nz(idx.v[i], env.v[i]);
#else
        if (m) x = x + nz(idx, env);
#endif
      }
      vr cx = vmin(vmax(x, vr(-M.clip_obs)), vr(M.clip_obs));
#ifdef NM_EMUL
      for (int i = 0; i < NM_WAVE; i++) if (m.v[i]) A.obs[(size_t)env.v[i] * kNOBS + idx.v[i]] = (float)cx.v[i];
#else
      if (m) gst1(A.obs, (size_t)env * kNOBS + idx, (float)cx);
#endif
    };
    put(o, hl, hl < 12);
    put((dofpos - defp) * M.obs_dofpos, hl + 12, l18);
    put(dofvel * M.obs_dofvel, hl + 30, l18);
    put(act, hl + 48, l18);
  }
  NM_ESTAMP(15);
  // ---- buffers the reference keeps between steps
  gstv(A.dofpos, hl + env * kNU, dofpos, live & l18);
  gstv(A.dofvel, hl + env * kNU, dofvel, live & l18);
  gstv(A.act, hl + env * kNU, act, live & l18);
  {
    vr cv = sel(hl == 0, cmd[0], sel(hl == 1, cmd[1], cmd[2]));
    gstv(A.cmd, hl + env * 3, cv, live & (hl < 3));
  }
#ifdef NM_EMUL
  for (int hh = 0; hh < 2; hh++)
    if (lives[hh]) {
      const int e = wave * 2 + hh, l0 = 32 * hh;
      A.eplen[e] = ep[hh]; A.rngctr[e] = ctr[hh]; A.rew[e] = (float)rew.v[l0];
      if (A.ret_acc) A.ret_acc[e] += (float)rew.v[l0];
      A.done[e] = (int64_t)(reset.v[l0] ? 1 : 0); A.timeout_now[e] = time_out.v[l0] ? 1.0f : 0.0f;
    }
#else
  if (hl == 0 && live) {
    gst1(A.eplen, (size_t)env, h1 ? ep[1] : ep[0]);
    gst1(A.rngctr, (size_t)env, h1 ? ctr[1] : ctr[0]);
    gst1(A.rew, (size_t)env, (float)rew);
    if (A.ret_acc) gadd1(A.ret_acc, (size_t)env, (float)rew);
    gst1(A.done, (size_t)env, (int64_t)(reset ? 1 : 0));
    gst1(A.timeout_now, (size_t)env, time_out ? 1.0f : 0.0f);
  }
#endif
#undef SHR
#undef SHI
}

// one wavefront = G consecutive envs (E2, env.py:200: mj_step(model, data, decimation) between load and epilogue)
template <class real, int G, class Mid = NoMid, class Pub = NoMid>
NM_FN void wave_step(ShW<real, G>& w, const Model<real>& M, const Args<real>& A, int wave, Mid&& mid = Mid(), Pub&& published = Pub()) {
  nm_stamp(-1);
  if constexpr (G == 2) {
    env_load2(w, M, A, wave, mid);
  } else {
    mid();
#pragma unroll
    for (int e = 0; e < G; e++) {
      int env = wave * G + e;
      env_load(w.e[e], M, A, env < A.N ? env : A.N - 1);
    }
  }
  nm_stamp(0);
  int dropped = 0;
  if (NM_ABLATE(A.ablate) & 1024) return;   // measurement only: load stage alone
  if (!(NM_ABLATE(A.ablate) & 2048))
  for (int s = 0; s < A.nsub; s++) substep(w, M, s == A.nsub - 1, &dropped, NM_ABLATE(A.ablate));
  if constexpr (G == 2) {
    env_finish2(w, M, A, wave, dropped, published);
  } else {
#pragma unroll
    for (int e = 0; e < G; e++) {
      int env = wave * G + e;
      env_finish(w.e[e], M, A, env, e == 0 ? dropped : 0, env < A.N);
    }
  }
  nm_stamp(10);
}

}  // namespace nm
