// nm_nik.hip - the scripted gait / inverse-kinematics engine (reference nikengine/engine.py EngineNode.update :700-715,
// FSM states :414-638, relative_ik :679-698; caller custom_play.py:49-76), batched over environments.
//
// 16 lanes own one env (4 envs per wavefront): lanes 0..14 are the 15 unordered leg pairs of the keep-out line search
// (the reference scans the 30 ordered pairs, the distance is symmetric), lanes 0..5 are the legs for the pose update
// and the IK. Everything is f64 like the reference's numpy: the search compares a distance with a threshold and a
// rounding flip there moves a foot target by a tenth of a step. State per env lives in HBM as rows of 18 doubles.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "../../include/nightmare_hip.h"

extern "C" int nm_policy_set_error(const char* m);

namespace {

constexpr double kPi = 3.14159265358979323846, kEps = 1e-6;
constexpr double kStandH = 0.10, kSitH = 0.0;                       // engine.py MyConfig STAND_HEIGHT / SIT_HEIGHT
constexpr double kKeepout = 0.03, kStepTime = 1.0, kStepHeight = 0.05;
constexpr double kTAdj = 1.0, kTGetUp = 2.5, kTSit = 2.5;
constexpr double kCX = 0.065, kFM = 0.13, kTB = 0.17;               // leg segment lengths
// gaits (engine.py:214-225): swing masks per step, bit l = leg l swings. 0 tripod, 1 ripple, 2 wave
__constant__ unsigned char c_gait_mask[3][6] = {{0x15, 0x2A, 0, 0, 0, 0}, {0x11, 0x0A, 0x24, 0, 0, 0}, {0x01, 0x02, 0x04, 0x08, 0x10, 0x20}};
__constant__ int c_gait_len[3] = {2, 3, 6};
enum { IDLE, ADJ_GETUP, GETUP, SIT, ADJ_SIT, STAND, WALK };

__constant__ double c_default_xy[6][2] = {{0.20, 0.20}, {0.26, 0.0}, {0.20, -0.20}, {-0.20, -0.20}, {-0.26, 0.0}, {-0.20, 0.20}};
__constant__ double c_offset_xy[6][2] = {{0.0685, 0.0775}, {0.093, 0}, {0.0685, -0.0775}, {-0.0685, -0.0775}, {-0.093, 0}, {-0.0685, 0.0775}};
__constant__ double c_servo[6] = {kPi / 4, 0, -kPi / 4, kPi / 4, 0, -kPi / 4};   // SERVO_OFFSET, coxa entries
__constant__ double c_urdf[3] = {0, -1.2734, -0.7854};                           // URDF_JOINT_OFFSETS per leg
__constant__ unsigned char c_pair[16][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {1, 2}, {1, 3}, {1, 4},
                                            {1, 5}, {2, 3}, {2, 4}, {2, 5}, {3, 4}, {3, 5}, {4, 5}, {0, 1}};

struct NikArgs {
  int N;
  int* fsm; int* step;
  int* gait_cmd; int* gait_cur;      // state.cmd.gait (what the caller asked for) and WalkState._gait (taken over at step boundaries)
  double* t0; double* gss;
  double* pose; double* start; double* last;   // [N,18] each
  const double* lin; const double* ang;
  const unsigned char* awake; const unsigned char* walk;
  double now, fps;
  float* out32; double* out64;
};

__device__ inline double dist2(double ax, double ay, double bx, double by) { return sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by)); }

// modules/math.py:107-126
__device__ inline double pt_seg(double x1, double y1, double x2, double y2, double px, double py) {
  const double dx = x2 - x1, dy = y2 - y1;
  if (dx == 0 && dy == 0) return dist2(px, py, x1, y1);
  const double t = ((px - x1) * dx + (py - y1) * dy) / (dx * dx + dy * dy);
  if (t < 0) return dist2(px, py, x1, y1);
  if (t > 1) return dist2(px, py, x2, y2);
  return dist2(px, py, x1 + t * dx, y1 + t * dy);
}
__device__ inline bool ccw(double ax, double ay, double bx, double by, double cx, double cy) { return (cy - ay) * (bx - ax) > (by - ay) * (cx - ax); }
// modules/math.py:100-104,129-146
__device__ inline double seg_seg(double a1x, double a1y, double a2x, double a2y, double b1x, double b1y, double b2x, double b2y) {
  if (ccw(a1x, a1y, b1x, b1y, b2x, b2y) != ccw(a2x, a2y, b1x, b1y, b2x, b2y) && ccw(a1x, a1y, a2x, a2y, b1x, b1y) != ccw(a1x, a1y, a2x, a2y, b2x, b2y))
    return 0.0;
  return fmin(fmin(pt_seg(a1x, a1y, a2x, a2y, b1x, b1y), pt_seg(a1x, a1y, a2x, a2y, b2x, b2y)),
              fmin(pt_seg(b1x, b1y, b2x, b2y, a1x, a1y), pt_seg(b1x, b1y, b2x, b2y, a2x, a2y)));
}
__device__ inline void rotz(double& x, double& y, double a) {
  double s, c;
  sincos(a, &s, &c);
  const double nx = c * x - s * y, ny = s * x + c * y;
  x = nx; y = ny;
}
__device__ inline double sigmoid(double v) { return 1.0 / (1.0 + exp(-13.0 * (v - 0.5))); }   // modules/math.py:20
__device__ inline double min16(double v) {
  for (int m = 8; m; m >>= 1) v = fmin(v, __shfl_xor(v, m, 16));
  return v;
}
// where leg `l` stands in the line search for reduction x (engine.py:560-565): grounded legs are dragged back by the
// rest of the half step, stepping legs sit at their ahead-of-time target
__device__ inline void probe_xy(int l, bool stepping, double px, double py, double lin, double ang, double x, double gss, int kNGait, double& ox, double& oy) {
  if (stepping) {
    const double f = x * kStepTime;
    ox = c_default_xy[l][0]; oy = c_default_xy[l][1] + lin * f;
    rotz(ox, oy, ang * f);
  } else {
    const double f = x * 2 * kNGait * (1 - gss);
    ox = px; oy = py - lin * f;
    rotz(ox, oy, -ang * f);
  }
}

// engine.py:679-698 for one leg; rel = (body_pos - POSE_OFFSET) * REL_CONVERT
__device__ inline void relative_ik(double x, double y, double z, double* out) {
  {
    const double n = sqrt(x * x + y * y);
    const double tx = x / n * kCX, ty = y / n * kCX;
    const double dx = x - tx, dy = y - ty;
    const double dist = sqrt(dx * dx + dy * dy + z * z);
    const double lo = fabs(kFM - kTB);
    const double ux = dx / dist, uy = dy / dist, uz = z / dist;
    if (dist > kFM + kTB) {
      const double L = kFM + kTB - kEps;
      x = tx + L * ux; y = ty + L * uy; z = L * uz;
    } else if (dist < lo) {
      const double L = lo + kEps;
      x = tx + L * ux; y = ty + L * uy; z = L * uz;
    }
  }
  const double d1 = sqrt(y * y + x * x) - kCX;
  const double d = sqrt(z * z + d1 * d1);
  const double zz = fabs(z) < kEps ? kEps : z;
  const double alpha = -atan2(y, x);
  const double beta = acos((z * z + d * d - d1 * d1) / (2 * (-zz) * d)) + acos((kFM * kFM + d * d - kTB * kTB) / (2 * kFM * d));
  const double gamma = -acos((kFM * kFM + kTB * kTB - d * d) / (2 * kFM * kTB)) + 2 * kPi;
  out[0] = alpha; out[1] = beta - kPi / 2; out[2] = gamma - 1.5 * kPi;
}

__global__ void __launch_bounds__(64) k_nik_update(NikArgs a) {
  const int gid = blockIdx.x * 4 + (threadIdx.x >> 4), s = threadIdx.x & 15;
  const bool live = gid < a.N;
  const int e = live ? gid : a.N - 1;                     // idle groups shadow the last env and store nothing
  const int leg = s < 6 ? s : 5;
  const bool isleg = live && s < 6;
  const int fsm = a.fsm[e], step = a.step[e];
  const int gcur = a.gait_cur[e], gcmd = a.gait_cmd[e];
  const int kNGait = c_gait_len[gcur];
  const unsigned swingm = c_gait_mask[gcur][step];
  const double t0 = a.t0[e], gss = a.gss[e];
  const double lin = a.lin[e], ang = a.ang[e];
  const bool awake = a.awake ? a.awake[e] != 0 : true, walk = a.walk ? a.walk[e] != 0 : true;
  const double* prow = a.pose + (size_t)e * 18;
  const double P[3] = {prow[3 * leg], prow[3 * leg + 1], prow[3 * leg + 2]};   // the FSM's previous output
  const double D[3] = {c_default_xy[leg][0], c_default_xy[leg][1], -kStandH};
  const double S[3] = {D[0], D[1], kSitH};
  double out[3] = {P[0], P[1], P[2]};
  int nfsm = fsm;                                          // state entered this tick (engine.py:649-653)
  double ngss = gss;
  int nstep = step, ngait = gcur;
  bool set_last = false;
  const double adv_t = a.now - t0;
  if (fsm == IDLE) {                                       // :414-431 (RobotState.pose is the default pose, never updated)
    for (int k = 0; k < 3; k++) out[k] = D[k];
    if (awake) nfsm = ADJ_GETUP;
  } else if (fsm == ADJ_GETUP) {                           // :434-458 start pose = RobotState.pose
    const double adv = adv_t / kTAdj;
    for (int k = 0; k < 3; k++) out[k] = adv < 1 ? D[k] + (S[k] - D[k]) * adv : S[k];
    if (adv >= 2) nfsm = GETUP;
  } else if (fsm == GETUP) {                               // :461-485
    const double adv = adv_t / kTGetUp;
    if (adv < 1) {
      const double w = sigmoid(adv);
      for (int k = 0; k < 3; k++) out[k] = S[k] + (D[k] - S[k]) * w;
    } else {
      for (int k = 0; k < 3; k++) out[k] = D[k];
      nfsm = !awake ? ADJ_SIT : (walk ? WALK : STAND);
    }
  } else if (fsm == ADJ_SIT) {                             // :509-520
    for (int k = 0; k < 3; k++) out[k] = D[k];
    nfsm = SIT;
  } else if (fsm == SIT) {                                 // :488-506
    const double adv = adv_t / kTSit;
    if (adv < 1) {
      const double w = sigmoid(adv);
      for (int k = 0; k < 3; k++) out[k] = D[k] + (S[k] - D[k]) * w;
    } else {
      for (int k = 0; k < 3; k++) out[k] = S[k];
      nfsm = IDLE;
    }
  } else if (fsm == STAND) {                               // :523-540 body commands are zero through EngineNode.update
    if (awake && !walk) {
      const double* srow = a.start + (size_t)e * 18;
      for (int k = 0; k < 3; k++) out[k] = srow[3 * leg + k];
    } else {
      nfsm = awake ? WALK : ADJ_SIT;
    }
  }
  const bool walking = fsm == WALK && ((awake && walk) || gss != 0);
  if (fsm == WALK && !walking) {                           // :629-636
    if (awake) nfsm = STAND;
    else { nfsm = IDLE; for (int k = 0; k < 3; k++) out[k] = D[k]; }
  }
  // ---- WalkState (:543-628). Executed by every lane so the 16-lane exchanges stay convergent; results are used
  // only by walking envs.
  {
    const int pi = c_pair[s][0], pj = c_pair[s][1];
    const double pix = prow[3 * pi], piy = prow[3 * pi + 1], pjx = prow[3 * pj], pjy = prow[3 * pj + 1];
    const bool wi = (swingm >> pi) & 1, wj = (swingm >> pj) & 1;
    double red = 1.0;
    bool done = !walking;
    for (int it = 0; it < 10; it++) {
      if (__all(done)) break;
      double ix, iy, jx, jy;
      probe_xy(pi, wi, pix, piy, lin, ang, red, gss, kNGait, ix, iy);
      probe_xy(pj, wj, pjx, pjy, lin, ang, red, gss, kNGait, jx, jy);
      double dmin = seg_seg(ix, iy, c_offset_xy[pi][0], c_offset_xy[pi][1], jx, jy, c_offset_xy[pj][0], c_offset_xy[pj][1]);
      dmin = min16(s < 15 ? dmin : 1e30);
      double cost = kKeepout - dmin;
      cost = cost < 0 ? 0.0 : cost;
      if (!done) {
        if (cost < 0.01 || red < 0) done = true;
        else red -= 0.1;
      }
    }
    if (walking) {
      const bool swing = (swingm >> leg) & 1;
      if (!swing) {
        const double m = red * (1.0 / a.fps) * 2 * kNGait;
        out[0] = P[0]; out[1] = P[1] - lin * m; out[2] = P[2];
        rotz(out[0], out[1], -ang * m);
      } else {
        const double f = red * kStepTime;
        double T[3] = {D[0], D[1] + lin * f, D[2]};
        rotz(T[0], T[1], ang * f);
        const double* lrow = a.last + (size_t)e * 18;
        const double t = gss, u = 1 - gss;
        for (int k = 0; k < 3; k++) {                      // modules/bezier.py:49-66 (de Casteljau on 4 points)
          const double up = k == 2 ? kStepHeight : 0.0;
          const double b0 = lrow[3 * leg + k], b1 = b0 + up, b2 = T[k] + up, b3 = T[k];
          const double c0 = u * b0 + t * b1, c1 = u * b1 + t * b2, c2 = u * b2 + t * b3;
          const double d0 = u * c0 + t * c1, d1 = u * c1 + t * c2;
          out[k] = u * d0 + t * d1;
        }
      }
      ngss = gss + kNGait / (kStepTime * a.fps);
      if (ngss > 1) { ngait = gcmd; ngss = 0; nstep = (step + 1) % c_gait_len[ngait]; set_last = true; }   // engine.py:626-629
    }
  }
  // ---- state entry (constructors of the FSM states)
  if (nfsm != fsm) {
    if (isleg && nfsm == STAND) { double* srow = a.start + (size_t)e * 18; for (int k = 0; k < 3; k++) srow[3 * leg + k] = P[k]; }
    if (nfsm == WALK) { ngss = 0; nstep = 0; ngait = gcmd; set_last = true; }   // WalkState.__init__ (:539-543)
    if (live && s == 0) { a.fsm[e] = nfsm; a.t0[e] = a.now; }
  }
  if (isleg && set_last) { double* lrow = a.last + (size_t)e * 18; for (int k = 0; k < 3; k++) lrow[3 * leg + k] = P[k]; }
  if (live && s == 0) { a.gss[e] = ngss; a.step[e] = nstep; a.gait_cur[e] = ngait; }
  if (isleg) {
    double* wrow = a.pose + (size_t)e * 18;
    for (int k = 0; k < 3; k++) wrow[3 * leg + k] = out[k];
    // EngineNode.set_hardware_pose (:700-708)
    const double sgn = leg < 3 ? 1.0 : -1.0;
    double q[3];
    relative_ik((out[0] - c_offset_xy[leg][0]) * sgn, (out[1] - c_offset_xy[leg][1]) * sgn, out[2], q);
    q[0] += c_servo[leg];
    for (int k = 0; k < 3; k++) {
      const double v = q[k] + c_urdf[k];
      if (a.out64) a.out64[(size_t)e * 18 + 3 * leg + k] = v;
      if (a.out32) a.out32[(size_t)e * 18 + 3 * leg + k] = (float)v;
    }
  }
}

__global__ void k_nik_set_gait(NikArgs a, const int* ids, int n, int gait) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a.gait_cmd[ids ? ids[i] : i] = gait;
}
__global__ void k_nik_reset(NikArgs a, const int* ids, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int e = ids ? ids[i] : i;
  a.fsm[e] = IDLE; a.step[e] = 0; a.t0[e] = 0; a.gss[e] = 0; a.gait_cur[e] = a.gait_cmd[e];
  for (int l = 0; l < 6; l++) {
    const double d[3] = {c_default_xy[l][0], c_default_xy[l][1], -kStandH};
    for (int k = 0; k < 3; k++) { a.pose[(size_t)e * 18 + 3 * l + k] = d[k]; a.start[(size_t)e * 18 + 3 * l + k] = d[k]; a.last[(size_t)e * 18 + 3 * l + k] = d[k]; }
  }
}

}  // namespace

struct nm_nik {
  int N = 0, device = 0;
  NikArgs a{};
  int* ids = nullptr;
};

#define NIK_CHECK(x, msg) do { if ((x) != hipSuccess) return nm_policy_set_error(msg); } while (0)

extern "C" nm_nik* nm_nik_create(int32_t num_envs, int32_t device) {
  if (num_envs <= 0) { nm_policy_set_error("nm_nik_create: num_envs must be positive"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { nm_policy_set_error("nm_nik_create: no such HIP device"); return nullptr; }
  nm_nik* h = new nm_nik;
  h->N = num_envs; h->device = device; h->a.N = num_envs;
  const size_t N = (size_t)num_envs;
  bool ok = hipMalloc((void**)&h->a.fsm, N * sizeof(int)) == hipSuccess && hipMalloc((void**)&h->a.step, N * sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&h->a.t0, N * sizeof(double)) == hipSuccess && hipMalloc((void**)&h->a.gss, N * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&h->a.pose, N * 18 * sizeof(double)) == hipSuccess && hipMalloc((void**)&h->a.start, N * 18 * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&h->a.last, N * 18 * sizeof(double)) == hipSuccess && hipMalloc((void**)&h->ids, N * sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&h->a.gait_cmd, N * sizeof(int)) == hipSuccess && hipMalloc((void**)&h->a.gait_cur, N * sizeof(int)) == hipSuccess;
  if (ok) ok = hipMemset(h->a.gait_cmd, 0, N * sizeof(int)) == hipSuccess;
  if (!ok) { nm_policy_set_error("nm_nik_create: hipMalloc failed"); delete h; return nullptr; }
  hipLaunchKernelGGL(k_nik_reset, dim3((num_envs + 255) / 256), dim3(256), 0, 0, h->a, (const int*)nullptr, num_envs);
  if (hipDeviceSynchronize() != hipSuccess) { nm_policy_set_error("nm_nik_create: reset kernel failed"); delete h; return nullptr; }
  return h;
}

extern "C" void nm_nik_destroy(nm_nik* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  void* p[] = {h->a.fsm, h->a.step, h->a.t0, h->a.gss, h->a.pose, h->a.start, h->a.last, h->ids, h->a.gait_cmd, h->a.gait_cur};
  for (void* q : p) if (q) (void)hipFree(q);
  delete h;
}

extern "C" int nm_nik_reset(nm_nik* h, const int32_t* ids_host, int32_t n, void* stream) {
  if (!h) return nm_policy_set_error("nm_nik_reset: null handle");
  NIK_CHECK(hipSetDevice(h->device), "nm_nik_reset: hipSetDevice failed");
  if (!ids_host) n = h->N;
  if (n <= 0) return 0;
  if (ids_host) {
    if (n > h->N) return nm_policy_set_error("nm_nik_reset: more ids than envs");
    for (int i = 0; i < n; i++) if (ids_host[i] < 0 || ids_host[i] >= h->N) return nm_policy_set_error("nm_nik_reset: env id out of range");
    NIK_CHECK(hipMemcpyAsync(h->ids, ids_host, (size_t)n * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream), "nm_nik_reset: copy failed");
  }
  hipLaunchKernelGGL(k_nik_reset, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->a, ids_host ? (const int*)h->ids : (const int*)nullptr, n);
  NIK_CHECK(hipGetLastError(), "nm_nik_reset: launch failed");
  if (ids_host) NIK_CHECK(hipStreamSynchronize((hipStream_t)stream), "nm_nik_reset: sync failed");   // ids_host may be reused by the caller
  return 0;
}

// state.cmd.gait of the listed engines (engine.py:297): 0 'tripod' (the default), 1 'ripple', 2 'wave' (engine.py:214-225). Like upstream
// a walking engine takes the new gait over when its current step completes (:627), a starting one when WalkState is built (:543).
extern "C" int nm_nik_set_gait(nm_nik* h, const int32_t* ids_host, int32_t n, int32_t gait, void* stream) {
  if (!h) return nm_policy_set_error("nm_nik_set_gait: null handle");
  if (gait < 0 || gait > 2) return nm_policy_set_error("nm_nik_set_gait: gait is 0 (tripod), 1 (ripple) or 2 (wave)");
  NIK_CHECK(hipSetDevice(h->device), "nm_nik_set_gait: hipSetDevice failed");
  if (!ids_host) n = h->N;
  if (n <= 0) return 0;
  if (ids_host) {
    if (n > h->N) return nm_policy_set_error("nm_nik_set_gait: more ids than envs");
    for (int i = 0; i < n; i++) if (ids_host[i] < 0 || ids_host[i] >= h->N) return nm_policy_set_error("nm_nik_set_gait: env id out of range");
    NIK_CHECK(hipMemcpyAsync(h->ids, ids_host, (size_t)n * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream), "nm_nik_set_gait: copy failed");
  }
  hipLaunchKernelGGL(k_nik_set_gait, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->a, ids_host ? (const int*)h->ids : (const int*)nullptr, n, gait);
  NIK_CHECK(hipGetLastError(), "nm_nik_set_gait: launch failed");
  if (ids_host) NIK_CHECK(hipStreamSynchronize((hipStream_t)stream), "nm_nik_set_gait: sync failed");
  return 0;
}

extern "C" int nm_nik_update(nm_nik* h, const double* lin_dev, const double* ang_dev, const unsigned char* awake_dev, const unsigned char* walk_dev,
                             double now_s, double engine_fps, float* angles_f32_dev, double* angles_f64_dev, void* stream) {
  if (!h || !lin_dev || !ang_dev) return nm_policy_set_error("nm_nik_update: bad argument");
  if (!(engine_fps > 0)) return nm_policy_set_error("nm_nik_update: engine_fps must be positive");
  NIK_CHECK(hipSetDevice(h->device), "nm_nik_update: hipSetDevice failed");
  NikArgs a = h->a;
  a.lin = lin_dev; a.ang = ang_dev; a.awake = awake_dev; a.walk = walk_dev; a.now = now_s; a.fps = engine_fps;
  a.out32 = angles_f32_dev; a.out64 = angles_f64_dev;
  hipLaunchKernelGGL(k_nik_update, dim3((h->N + 3) / 4), dim3(64), 0, (hipStream_t)stream, a);
  NIK_CHECK(hipGetLastError(), "nm_nik_update: launch failed");
  return 0;
}

extern "C" int nm_nik_get_state(nm_nik* h, double* pose_host, int32_t* fsm_host, double* gait_step_state_host) {
  if (!h) return nm_policy_set_error("nm_nik_get_state: null handle");
  NIK_CHECK(hipSetDevice(h->device), "nm_nik_get_state: hipSetDevice failed");
  NIK_CHECK(hipDeviceSynchronize(), "nm_nik_get_state: sync failed");
  const size_t N = (size_t)h->N;
  if (pose_host) NIK_CHECK(hipMemcpy(pose_host, h->a.pose, N * 18 * sizeof(double), hipMemcpyDeviceToHost), "nm_nik_get_state: copy failed");
  if (fsm_host) NIK_CHECK(hipMemcpy(fsm_host, h->a.fsm, N * sizeof(int), hipMemcpyDeviceToHost), "nm_nik_get_state: copy failed");
  if (gait_step_state_host) NIK_CHECK(hipMemcpy(gait_step_state_host, h->a.gss, N * sizeof(double), hipMemcpyDeviceToHost), "nm_nik_get_state: copy failed");
  return 0;
}
