/* nm_oracle.h - CPU (fp64) restatement of the NightmareV3Env.step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under nightmare_rl_amd/ (the product) may include, link or
 * call this. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / reported CPU baseline.
 *
 * PARITY STATUS: the env logic (E1-E9, reference envs/nightmare_v3_env.py:145-371) is pinned by
 * golden vectors generated from the reference's own Python class (tests/golden/). The rigid-body
 * physics restates MuJoCo 3.1.2 (`mujoco<=3.1.2`, requirements.txt:2; tag 3.1.2,
 * build_custom_mujoco.sh:8-10), which is a third-party dependency absent from /root/reference and
 * not installable here: the physics part is "PARITY UNPINNED" against MuJoCo itself and is pinned
 * only by the known-answer tests in tests/test_oracle_physics.py.
 */
#ifndef NM_ORACLE_H
#define NM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMO_NBODY 20
#define NMO_NV 24
#define NMO_NQ 25
#define NMO_NU 18
#define NMO_NSENS 13
#define NMO_MAXCON 48
#define NMO_MAXEFC (4 * NMO_MAXCON)

/* per-env MjData equivalent: persistent state + the outputs the env reads after mj_step */
typedef struct {
  /* state (what mj_step advances) */
  double qpos[NMO_NQ], qvel[NMO_NV], qacc_warmstart[NMO_NV], ctrl[NMO_NU], time;
  /* mj_kinematics / mj_comPos */
  double xpos[NMO_NBODY][3], xquat[NMO_NBODY][4], xmat[NMO_NBODY][9];
  double xipos[NMO_NBODY][3], ximat[NMO_NBODY][9];
  double xanchor[NMO_NU][3], xaxis[NMO_NU][3]; /* hinge j = body 2+j = dof 6+j */
  double subtree_com[3];
  double cinert[NMO_NBODY][10], cdof[NMO_NV][6];
  /* mj_crb / mj_factorM (dense storage, tree-sparse content) */
  double qM[NMO_NV][NMO_NV], qLD[NMO_NV][NMO_NV], qLDiagInv[NMO_NV];
  /* mj_comVel / mj_rne */
  double cvel[NMO_NBODY][6], cdof_dot[NMO_NV][6], qfrc_bias[NMO_NV];
  /* forces / accelerations */
  double qfrc_actuator[NMO_NV], qfrc_smooth[NMO_NV], qacc_smooth[NMO_NV], qfrc_constraint[NMO_NV], qacc[NMO_NV];
  /* contacts */
  int32_t ncon, nefc;
  double con_pos[NMO_MAXCON][3], con_frame[NMO_MAXCON][9], con_dist[NMO_MAXCON];
  int32_t con_body[NMO_MAXCON], con_body1[NMO_MAXCON], con_geom[NMO_MAXCON]; /* body2, body1 (0 = world), colliding-mesh index of geom2 */
  double efc_force[NMO_MAXEFC];
  double sensordata[NMO_NSENS];
  int32_t solver_niter, noslip_niter, nwarning, pad;
} nmo_data;

/* per-thread scratch for the constraint rows */
typedef struct {
  double J[NMO_MAXEFC][NMO_NV], JM2[NMO_MAXEFC][NMO_NV];
  double pos[NMO_MAXEFC], diagApprox[NMO_MAXEFC], R[NMO_MAXEFC], D[NMO_MAXEFC], K[NMO_MAXEFC], B[NMO_MAXEFC], imp[NMO_MAXEFC];
  double vel[NMO_MAXEFC], aref[NMO_MAXEFC], b[NMO_MAXEFC], jar[NMO_MAXEFC];
  double AR[NMO_MAXEFC * NMO_MAXEFC];
} nmo_scratch;

int nmo_sizeof_data(void);
int nmo_sizeof_scratch(void);
void nmo_reset_data(nmo_data* d);                         /* mj_resetData */
void nmo_forward(nmo_data* d, nmo_scratch* s);            /* mj_forward */
void nmo_step(nmo_data* d, nmo_scratch* s, int nstep);    /* mj_step(model, data, nstep) */
void nmo_set_collide_self(int on);                        /* tibia-tibia pairs (P5); 0 = floor only */

/* ---- env layer: NightmareV3Env restated (reference envs/nightmare_v3_env.py) ---- */
typedef struct nmo_env nmo_env;
/* every reward name of the reference config that has a _reward_ function (env.py:399-497), alphabetical, termination last:
 * action_rate ang_vel_xy base_height body_contact_forces default_position dof_acc dof_vel feet_air_time feet_contact_forces lin_vel_z
 * orientation stand_still torques tracking_ang_vel tracking_lin_vel termination */
#define NMO_NREW 16

nmo_env* nmo_env_create(int num_envs, uint64_t seed, int64_t env_id_offset, int num_threads);
void nmo_env_destroy(nmo_env* e);
/* reset_idx(ids) (env.py:335-371); ids==NULL -> all. cmd_u: optional [n,2] uniforms for the command resample */
void nmo_env_reset_idx(nmo_env* e, const int32_t* ids, int n, const double* cmd_u);
/* step (env.py:145-311). actions [N,18] f32. cmd_u optional [N,4] uniforms: (x,yaw) for the periodic
 * resample, (x,yaw) for the reset resample; NULL -> internal counter RNG. Outputs may be NULL. */
void nmo_env_step(nmo_env* e, const float* actions, const double* cmd_u, float* obs, float* rew, int64_t* done,
                  float* time_outs, double* obs64, double* rew64);
/* dynamics + contact only (BASELINE config 2; reference simple_test.py:25-45): servo command from the live joint angles, then
 * mj_step(model, data[i], decimation) over the threads. No epilogue; env-level buffers untouched. */
void nmo_env_step_physics(nmo_env* e, const float* actions);
/* non-default reward table / contact modes (config.py:17-21, 77-100). reward_scales[NMO_NREW] are the RAW config scales (x dt is
 * applied here, env.py:123-128; 0 drops the term); NULL keeps the current ones */
void nmo_env_configure(nmo_env* e, const double* reward_scales, int tibia_mode, double tibia_max_force, int body_mode, double body_max_force,
                       double base_height_target, double max_contact_force);
/* state of _reward_feet_air_time: feet_air_time[N,6], last_contacts[N,6], last_contacts_filt[N,6] (env.py:90-93) */
void nmo_env_get_feet_state(nmo_env* e, double* air6, uint8_t* last6, uint8_t* filt6);
void nmo_env_set_feet_state(nmo_env* e, const double* air6, const uint8_t* last6, const uint8_t* filt6);
/* observation noise: noise_scale_vec66 NULL = off; u = injected [N,66] uniforms for the next steps or NULL = counter RNG
 * keyed (seed + NMO_NOISE_KEY, global env id, step*66 + k) */
#define NMO_NOISE_KEY 0x4E4F495345ull
void nmo_env_set_noise(nmo_env* e, const double* noise_scale_vec66, const double* u);
void nmo_env_get_state(nmo_env* e, double* qpos, double* qvel, double* qacc_warmstart);
void nmo_env_set_state(nmo_env* e, const double* qpos, const double* qvel, const double* qacc_warmstart);
/* host-side buffers the reference keeps between steps: dof_pos, dof_vel (stale across resets), actions, commands,
 * episode length, episode sums */
void nmo_env_get_buffers(nmo_env* e, double* dof_pos, double* dof_vel, double* actions, double* commands,
                         int64_t* ep_len, double* episode_sums /* [NMO_NREW][N] */);
void nmo_env_set_buffers(nmo_env* e, const double* dof_pos, const double* dof_vel, const double* actions,
                         const double* commands, const int64_t* ep_len, const double* episode_sums);
/* extras['episode'] of the last step in which >=1 env reset: mean episode sum / 20 per reward; returns #resets of last step */
int nmo_env_episode_stats(nmo_env* e, double* out /* [NMO_NREW] */);
nmo_data* nmo_env_data(nmo_env* e, int i);
/* intermediate per-env buffers of the last step, for golden comparison (each may be NULL):
 * base_lin_vel[N,3] base_ang_vel[N,3] projected_gravity[N,3] tibia[N,6] feet[N,6] body[N] rew_terms[NMO_NREW,N] */
void nmo_env_get_debug(nmo_env* e, double* blv, double* bav, double* pg, double* tibia, double* feet, double* body,
                       double* rew_terms);
/* counter-based uniform in [0,1) with 24 random bits (exact in fp32): shared definition with the HIP path */
double nmo_rand_u24(uint64_t seed, uint64_t global_env, uint32_t counter);

#ifdef __cplusplus
}
#endif
#endif
