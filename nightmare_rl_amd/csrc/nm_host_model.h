// nm_host_model.h - host side: turn the generated model tables (model/nm_model_data.h, doubles) and the
// env configuration (reference envs/nightmare_v3_config.py) into the flat `real` arrays nm::Model<real> points at.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "../model/nm_model_data.h"
#include "nm_core.h"

namespace nmhost {

// env configuration with the reference defaults (file:line into reference envs/nightmare_v3_config.py)
struct EnvConfig {
  int decimation = 2;               // :45
  double p_gain = 20.0;             // :36
  double action_scale = 0.2;        // :46
  double default_pos[3] = {0.0, 3.14159265358979323846 / 5, 0.0};  // :39-44
  double clip_actions = 1.0;        // :74
  double clip_observations = 100.0; // :73
  double obs_lin_vel = 2.0, obs_ang_vel = 0.25, obs_dof_pos = 1.0, obs_dof_vel = 0.05;  // :68-71
  double episode_length_s = 20.0;   // :14
  double resampling_time = 10.0;    // :60
  double max_lin_vel_x = 0.5, max_ang_vel = 0.8;  // :62,:64
  double termination_contact_force = 160.0;       // :22
  double tracking_sigma = 0.008;    // :98
  // reward scales (:78-95), order = nm::R_* (alphabetical, termination last); the reference ships eight of them as 0
  double rew_scales[nm::kNREW] = {-0.02, 0, 0, -5.0, -0.01, -2.5e-5, 0, 0, 0, 0, -5.0, 0, 0, 6.0, 8.0, -200.0};
  int tibia_contact_mode = 1, body_contact_mode = 1;                    // :18,:20
  double tibia_max_contact_force = 2.0, body_max_contact_force = 2.0;   // :19,:21
  double base_height_target = 0.1, max_contact_force = 10.0;            // :99,:100
};

inline void quat2mat(const double* q, double* m) {
  double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q11 = q[1] * q[1], q12 = q[1] * q[2],
         q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02); m[3] = 2 * (q12 + q03);
  m[5] = 2 * (q23 - q01); m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
// body-frame inertia tensor (xx yy zz xy xz yz) from principal moments + orientation
inline void body_inertia6(const double* iquat, const double* diag, double* out) {
  double R[9], T[9];
  quat2mat(iquat, R);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[3 * i + j] = R[3 * i + j] * diag[j];
  double I[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) I[3 * i + j] = T[3 * i] * R[3 * j] + T[3 * i + 1] * R[3 * j + 1] + T[3 * i + 2] * R[3 * j + 2];
  out[0] = I[0]; out[1] = I[4]; out[2] = I[8]; out[3] = I[1]; out[4] = I[2]; out[5] = I[5];
}

constexpr double kTolPlaneMesh = 0.3;   // MuJoCo's mjTOLPLANEMESH [3P]: extra plane-mesh contacts keep this x rbound from the first one
template <class real> struct Tables {
  std::vector<real> legc, basec, colc, hullv, footc, qpos0;
  std::vector<real> hullnv;
  double total_mass = 0;

  void build() {
    legc.assign(nm::kNLEG * nm::kLegN, real(0));
    for (int l = 0; l < nm::kNLEG; l++)
      for (int k = 0; k < 3; k++) {
        int b = 2 + 3 * l + k;
        real* c = &legc[l * nm::kLegN + k * nm::kLinkN];
        double R[9], I6[6];
        quat2mat(nm_body_quat[b], R);
        body_inertia6(nm_body_iquat[b], nm_body_inertia[b], I6);
        for (int j = 0; j < 3; j++) c[j] = (real)nm_body_pos[b][j];
        for (int j = 0; j < 9; j++) c[3 + j] = (real)R[j];
        for (int j = 0; j < 3; j++) c[12 + j] = (real)nm_jnt_axis[b - 2][j];
        for (int j = 0; j < 3; j++) c[15 + j] = (real)nm_body_ipos[b][j];
        for (int j = 0; j < 6; j++) c[18 + j] = (real)I6[j];
        c[24] = (real)nm_body_mass[b];
      }
    basec.assign(nm::kBaseN, real(0));
    {
      double I6[6];
      body_inertia6(nm_body_iquat[1], nm_body_inertia[1], I6);
      for (int j = 0; j < 3; j++) basec[j] = (real)nm_body_ipos[1][j];
      for (int j = 0; j < 6; j++) basec[3 + j] = (real)I6[j];
      basec[9] = (real)nm_body_mass[1];
    }
    total_mass = 0;
    for (int b = 1; b < NM_NBODY; b++) total_mass += nm_body_mass[b];
    colc.assign(nm::kNCOL * nm::kColN, real(0));
    for (int g = 0; g < nm::kNCOL; g++) {
      real* c = &colc[g * nm::kColN];
      for (int j = 0; j < 3; j++) c[j] = (real)nm_col_center[g][j];
      c[3] = (real)nm_col_rbound[g];
      c[4] = (real)nm_body_invweight0[nm_col_body[g]][0];
      if (nm_col_nvert[g] > nm::kHullIters * NM_WAVE) std::abort();  // support_exhaustive covers kHullIters x 64 vertices
      c[5] = (real)nm_col_nvert[g];
      c[6] = (real)nm_col_vadr[g];
      for (int j = 0; j < 3; j++) c[8 + j] = (real)nm_col_obb_center[g][j];
      for (int j = 0; j < 9; j++) c[11 + j] = (real)nm_col_obb_axes[g][j];
      for (int j = 0; j < 3; j++) c[20 + j] = (real)nm_col_obb_half[g][j];
    }
    hullv.assign((size_t)NM_NHULLVERT * 4, real(0));
    for (int i = 0; i < NM_NHULLVERT; i++)
      for (int j = 0; j < 3; j++) hullv[4 * i + j] = (real)nm_hull_vert[i][j];
    {  // ring table: vertex -> (neighbour xyz, neighbour local id) x maxnbr, then the vertex itself
      const int S = NM_HULL_MAXNBR + 1;
      static_assert(NM_HULL_MAXNBR <= nm::kBatchLane0, "lanes kBatchLane0.. of a ring must hold the vertex itself (stage_collide's batched emission)");
      hullnv.assign((size_t)NM_NHULLVERT * S * 4, real(0));
      for (int g = 0; g < nm::kNCOL; g++) {
        const int vadr = nm_col_vadr[g], nv = nm_col_nvert[g];
        for (int i = 0; i < nv; i++) {
          real* row = &hullnv[(size_t)(vadr + i) * S * 4];
          for (int j = 0; j < NM_HULL_MAXNBR; j++) {
            const int n = nm_hull_nbr[vadr + i][j];
            for (int k = 0; k < 3; k++) row[4 * j + k] = n >= 0 ? (real)nm_hull_vert[vadr + n][k] : real(0);
            row[4 * j + 3] = (real)n;
          }
          for (int k = 0; k < 3; k++) row[4 * NM_HULL_MAXNBR + k] = (real)nm_hull_vert[vadr + i][k];
          // the vertex's lone-contact slack (nm_core.h kSlackScale): 0.999 tol_planemesh rbound - 1.00001 max |neighbour - vertex|, in 2^-20 m
          double far = 0;
          for (int j = 0; j < NM_HULL_MAXNBR; j++) {
            const int n = nm_hull_nbr[vadr + i][j];
            if (n < 0) continue;
            double d2 = 0;
            for (int k = 0; k < 3; k++) { const double d = (double)nm_hull_vert[vadr + n][k] - (double)nm_hull_vert[vadr + i][k]; d2 += d * d; }
            far = std::max(far, std::sqrt(d2));
          }
          const double slack = (0.999 * kTolPlaneMesh * (double)nm_col_rbound[g] - 1.00001 * far) * (double)nm::kSlackScale;
          row[4 * NM_HULL_MAXNBR + 3] = (real)std::floor(std::min(std::max(slack, -1.0), 8388607.0));
        }
      }
    }
    footc.assign(nm::kNLEG * 4, real(0));
    for (int l = 0; l < nm::kNLEG; l++) {
      for (int j = 0; j < 3; j++) footc[4 * l + j] = (real)nm_sens_pos[6 + l][j];
      footc[4 * l + 3] = (real)nm_sens_radius[6 + l];
    }
    qpos0.assign(nm::kNQ, real(0));
    for (int j = 0; j < nm::kNQ; j++) qpos0[j] = (real)nm_qpos0[j];
  }

  // scalars of nm::Model (pointers are set by the caller: host pointers for the emulation, device pointers for HIP)
  void fill_scalars(nm::Model<real>& M, const EnvConfig& cfg) const {
    std::copy(legc.begin(), legc.end(), M.legc); std::copy(basec.begin(), basec.end(), M.basec); std::copy(colc.begin(), colc.end(), M.colc);
    std::copy(footc.begin(), footc.end(), M.footc); std::copy(qpos0.begin(), qpos0.end(), M.qpos0);
    M.maxnbr = NM_HULL_MAXNBR;
    for (int k = 0; k < 3; k++) {
      bool id = true;
      for (int l = 0; l < nm::kNLEG; l++)
        for (int j = 0; j < 9; j++) id = id && legc[l * nm::kLegN + k * nm::kLinkN + 3 + j] == real(j % 4 == 0 ? 1 : 0);
      M.link_rot_id[k] = id ? 1 : 0;
    }
    M.total_mass = (real)total_mass;
    M.h = (real)NM_TIMESTEP; M.kv = (real)NM_KV; M.ctrl_max = (real)NM_CTRL_MAX; M.grav = (real)(-nm_gravity[2]); M.mu = (real)NM_FRICTION;
    double dmax = nm_solimp[1], tc = nm_solref[0], dr = nm_solref[1];
    M.solref_K = (real)(1.0 / std::fmax(1e-15, dmax * dmax * tc * tc * dr * dr));
    M.solref_B = (real)(2.0 / std::fmax(1e-15, dmax * tc));
    M.si_d0 = (real)nm_solimp[0]; M.si_dmax = (real)nm_solimp[1]; M.si_width = (real)nm_solimp[2]; M.si_mid = (real)nm_solimp[3]; M.si_power = (real)nm_solimp[4];
    M.pgs_scale = (real)(1.0 / (NM_MEANINERTIA * NM_NV));
    M.pgs_tol = (real)NM_TOLERANCE; M.noslip_tol = (real)NM_NOSLIP_TOLERANCE; M.tol_planemesh = (real)kTolPlaneMesh;
    M.pgs_iters = NM_ITERATIONS; M.noslip_iters = NM_NOSLIP_ITERATIONS;
    M.mpr_iters = 50; M.mpr_tol = (real)1e-6;  // MuJoCo 3.1.2 defaults opt.mpr_iterations / opt.mpr_tolerance
    double dt = NM_TIMESTEP * cfg.decimation;                       // env.py:99
    M.dt = (real)dt; M.p_gain = (real)cfg.p_gain; M.clip_obs = (real)cfg.clip_observations;
    M.obs_lin = (real)cfg.obs_lin_vel; M.obs_ang = (real)cfg.obs_ang_vel; M.obs_dofpos = (real)cfg.obs_dof_pos; M.obs_dofvel = (real)cfg.obs_dof_vel;
    M.max_lin_x = (real)cfg.max_lin_vel_x; M.max_ang = (real)cfg.max_ang_vel; M.term_force = (real)cfg.termination_contact_force;
    M.sigma = (real)cfg.tracking_sigma;
    M.max_ep_len = (real)std::ceil(cfg.episode_length_s / dt);      // env.py:101
    for (int j = 0; j < 3; j++) M.default_pos[j] = (real)cfg.default_pos[j];
    M.action_scale = (float)cfg.action_scale; M.clip_actions = (float)cfg.clip_actions;
    M.resample_every = (int)(cfg.resampling_time / dt);             // env.py:235
    for (int k = 0; k < nm::kNREW; k++) M.rew_scale[k] = (real)(cfg.rew_scales[k] * dt);  // env.py:123-128
    M.ep_len_s = (real)cfg.episode_length_s;
    M.rew_extra = 0;
    for (int k : {nm::R_ANG_VEL_XY, nm::R_BASE_HEIGHT, nm::R_DOF_VEL, nm::R_FEET_AIR_TIME, nm::R_FEET_CONTACT, nm::R_LIN_VEL_Z, nm::R_STAND_STILL})
      if (cfg.rew_scales[k] != 0) M.rew_extra = 1;
    M.tibia_mode = cfg.tibia_contact_mode; M.body_mode = cfg.body_contact_mode;
    M.collide_batch_min = 3;
    if (const char* e = std::getenv("NM_COLLIDE_BATCH_MIN")) M.collide_batch_min = std::atoi(e);   // tests: 99 = never batch, 1 = whenever certified
    M.tibia_max = (real)cfg.tibia_max_contact_force; M.body_max = (real)cfg.body_max_contact_force;
    M.base_h_target = (real)cfg.base_height_target; M.max_contact_force = (real)cfg.max_contact_force;
  }
};

}  // namespace nmhost
