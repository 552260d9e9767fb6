/* nightmare_hip.h - C ABI of the MI355X-native NightmareV3Env.step() backend (libnightmare_hip.so).
 *
 * Drop-in boundary for the reference's env hot path. Every entry point names the reference interface it
 * replaces (file:line into the reference tree). Plain pointers and sizes only; device pointers are HIP device
 * memory on the env's device, `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All functions return 0 on success, nonzero on error (text via nm_last_error()). There is NO CPU path:
 * creation fails if no HIP device is usable.
 */
#ifndef NIGHTMARE_HIP_H
#define NIGHTMARE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NM_NUM_OBS 66      /* envs/nightmare_v3_config.py:11 */
#define NM_NUM_ACTIONS 18  /* envs/nightmare_v3_config.py:13 */
#define NM_NUM_REWARDS 16  /* reward names of envs/nightmare_v3_config.py:78-96 that have a _reward_ function (envs/nightmare_v3_env.py:399-497) */
#define NM_DTYPE_F32 0
#define NM_DTYPE_F64 1     /* verification build of the same kernels in double */

typedef struct nm_env nm_env;

/* The NightmareV3Config fields the step path reads (envs/nightmare_v3_config.py:4-100). Zero-initialised
 * fields are NOT defaults: start from nm_default_config(). Reward scales are the raw config values (before
 * the x dt of envs/nightmare_v3_env.py:128), order = nm_reward_name(i). */
typedef struct {
  int32_t decimation;               /* control.decimation :45 */
  double p_gain;                    /* control.p_gain :36 */
  double action_scale;              /* control.action_scale :46 */
  double default_pos[3];            /* control.default_pos (coxa, femur, tibia) :39-44 */
  double clip_actions;              /* normalization.clip_actions :74 */
  double clip_observations;         /* normalization.clip_observations :73 */
  double obs_lin_vel, obs_ang_vel, obs_dof_pos, obs_dof_vel; /* normalization.obs_scales :68-71 */
  double episode_length_s;          /* env.episode_length_s :14 */
  double resampling_time;           /* commands.resampling_time :60 */
  double max_lin_vel_x, max_ang_vel;/* commands.ranges :62,:64 */
  double termination_contact_force; /* env.termination_contact_force :22 */
  double tracking_sigma;            /* rewards.tracking_sigma :98 */
  double reward_scales[NM_NUM_REWARDS]; /* rewards.scales :78-95; 0 drops the term from the reward table (env.py:123-128) */
  int32_t tibia_contact_mode;       /* env.tibia_contact_mode :18  0 ignore, 1 penalise, 2 terminate (env.py:248-251,479-485) */
  double tibia_max_contact_force;   /* env.tibia_max_contact_force :19 */
  int32_t body_contact_mode;        /* env.body_contact_mode :20 */
  double body_max_contact_force;    /* env.body_max_contact_force :21 */
  double base_height_target;        /* rewards.base_height_target :99 */
  double max_contact_force;         /* rewards.max_contact_force :100 */
} nm_config;

void nm_default_config(nm_config* cfg);
/* "action_rate", "ang_vel_xy", "base_height", "body_contact_forces", "default_position", "dof_acc", "dof_vel", "feet_air_time",
 * "feet_contact_forces", "lin_vel_z", "orientation", "stand_still", "torques", "tracking_ang_vel", "tracking_lin_vel", "termination":
 * the order rewards are evaluated in (class_to_dict iterates dir(), envs/helpers.py:7; termination is added last,
 * envs/nightmare_v3_env.py:132-137, 285). The config names `collision` and `feet_stumble` (:95-96) have no function upstream. */
const char* nm_reward_name(int i);
const char* nm_last_error(void);

/* NightmareV3Env.__init__ (envs/nightmare_v3_env.py:27-140): model tables to the device, N env states at qpos0.
 * env_id_offset = global id of env 0 (multi-GPU sharding: counter-based RNG is keyed by the global id). */
int nm_create(const nm_config* cfg, int32_t num_envs, int32_t device, uint64_t seed, int64_t env_id_offset,
              int32_t dtype, nm_env** out);
int nm_destroy(nm_env* env);
int32_t nm_num_envs(const nm_env* env);
int32_t nm_dtype(const nm_env* env);

/* reset_idx(env_ids) (envs/nightmare_v3_env.py:335-371). ids: HOST int32 array, NULL = all envs.
 * ep_stats_dev [NM_NUM_REWARDS] f32 receives extras['episode'] (mean episode sums / episode_length_s);
 * episode_length_dev [N] i64 is the caller-owned episode_length_buf (envs/nightmare_v3_env.py:88). */
int nm_reset(nm_env* env, const int32_t* ids_host, int32_t n, int64_t* episode_length_dev, float* ep_stats_dev,
             void* stream);

/* step(actions) (envs/nightmare_v3_env.py:145-311) for all N envs, one launch.
 *   actions_dev        [N,18] f32 (read only)
 *   episode_length_dev [N] i64   in/out (episode_length_buf)
 *   obs_dev [N,66] f32, rew_dev [N] f32, done_dev [N] i64: the returned tuple (:311)
 *   time_outs_dev [N] f32, ep_stats_dev [NM_NUM_REWARDS] f32: extras; like the reference (:344-371) they are only
 *   refreshed by a step in which at least one env reset. A refresh of the buffer the previous refresh wrote is incremental (its
 *   ones are cleared, the new ones set), so the caller must not write into time_outs_dev between steps; a buffer at ANOTHER ADDRESS
 *   (a first call, a different allocation) is rewritten in full - decided on the device, also for launches replayed from a graph.
 *   Only the address is compared: a caller that frees the buffer and gets the same address back from its allocator, or overwrites the
 *   buffer's contents, must call nm_invalidate_time_outs() before the next step. */
int nm_step(nm_env* env, const float* actions_dev, int64_t* episode_length_dev, float* obs_dev, float* rew_dev,
            int64_t* done_dev, float* time_outs_dev, float* ep_stats_dev, void* stream);

/* Forget which time_outs buffer the last refresh wrote: the next refresh rewrites the whole buffer it is given (stream-ordered).
 * For callers that re-allocate or overwrite their extras['time_outs'] tensor (envs/nightmare_v3_env.py:369-371 builds a new one each time). */
int nm_invalidate_time_outs(nm_env* env, void* stream);

/* Physics only: action -> PD velocity command -> mj_step x decimation (envs/nightmare_v3_env.py:152-210),
 * no rewards/obs/reset (BASELINE config "dynamics+contact kernel only"). */
int nm_step_physics(nm_env* env, const float* actions_dev, void* stream);

/* MjData state access (data[i].qpos / qvel / qacc_warmstart, envs/nightmare_v3_env.py:217-221,349-350).
 * HOST double arrays [N,25] [N,24] [N,24]; NULL entries are skipped. Synchronous. */
int nm_get_state(nm_env* env, double* qpos, double* qvel, double* qacc_warmstart);
int nm_set_state(nm_env* env, const double* qpos, const double* qvel, const double* qacc_warmstart);
/* The host-side buffers the reference keeps between steps (self.dof_pos, self.dof_vel, self.actions,
 * self.commands, self.episode_sums; envs/nightmare_v3_env.py:60-61,94-96,140). HOST doubles, NULL = skip.
 * episode_sums is [N,NM_NUM_REWARDS] in nm_reward_name order. */
int nm_get_buffers(nm_env* env, double* dof_pos, double* dof_vel, double* actions, double* commands, double* episode_sums);
int nm_set_buffers(nm_env* env, const double* dof_pos, const double* dof_vel, const double* actions,
                   const double* commands, const double* episode_sums);
/* State of _reward_feet_air_time (self.feet_air_time, self.last_contacts, self.last_contacts_filt; envs/nightmare_v3_env.py:90-93,
 * 447-477). HOST arrays: feet_air_time [N,6] double, last_contacts / last_contacts_filt [N,6] uint8. NULL = skip. Synchronous. */
int nm_get_feet_state(nm_env* env, double* feet_air_time, unsigned char* last_contacts, unsigned char* last_contacts_filt);
int nm_set_feet_state(nm_env* env, const double* feet_air_time, const unsigned char* last_contacts, const unsigned char* last_contacts_filt);
/* RNG-free command resampling for parity tests: HOST [N,4] uniforms in [0,1) used by the next steps instead
 * of the counter RNG ((x,yaw) for the periodic resample :235, (x,yaw) for the reset resample :356). NULL = RNG. */
int nm_set_command_uniforms(nm_env* env, const double* u_host);
/* counters: [0] contacts dropped (always 0 since every contact is kept: the matrix-free solver takes what the register-resident one cannot), [1] MuJoCo-style bad-state resets,
 * [2] support-vertex searches that fell back from the warm start to the exhaustive scan */
int nm_get_counters(nm_env* env, int64_t* out3);
/* optional debug dump [N,256] reals (dtype of the env) written by nm_step; NULL disables */
int nm_set_debug_buffer(nm_env* env, void* dbg_dev);
/* Running return kept by the step kernel itself: with a DEVICE float [N] buffer set, every nm_step adds the step's reward
 * (envs/nightmare_v3_env.py:277-288, what step() returns as reward_buffer) to acc[env] - the caller's
 * `cur_reward_sum += rewards` (rsl_rl OnPolicyRunner.learn, the loop train.py:54 drives) without a second launch per step.
 * The caller owns, zeroes and reads the buffer (stream-ordered with nm_step); NULL switches it off (the default). Not touched
 * by nm_step_physics. */
int nm_set_return_accumulator(nm_env* env, float* acc_dev);

/* Observation noise (envs/nightmare_v3_env.py:109-119 builds noise_scale_vec, :304-305 applies it): with a HOST [66] vector
 * set, every nm_step adds (2u-1)*noise_scale_vec[k] to observation k before the clip, u ~ U[0,1) from the counter RNG
 * keyed by (seed, global env id, step, k). NULL switches it off (the default, config add_noise=False :49). */
int nm_set_observation_noise(nm_env* env, const double* noise_scale_vec_host);
/* RNG-free noise for parity tests: HOST [N,66] uniforms used by the following steps instead of the counter RNG
 * (what np.random.rand(N,66) returned at :305). NULL = RNG. */
int nm_set_noise_uniforms(nm_env* env, const double* u_host);
/* State log (envs/nightmare_v3_env.py:261-272 records data[0]): env_index >= 0 makes every nm_step keep that env's
 * post-physics, pre-reset qpos/qvel; -1 = off. nm_get_state_record copies the last record to HOST qpos[25], qvel[24] and
 * the number of MuJoCo bad-state resets inside that step (data.time restarts there). Synchronous. */
int nm_set_state_record(nm_env* env, int32_t env_index);
int nm_get_state_record(nm_env* env, double* qpos, double* qvel, int32_t* bad_state_resets);

/* Measurement hook (no reference counterpart): when enabled, every nm_step / nm_step_physics brackets its step
 * kernel with HIP events on the launch stream. Each call synchronises, returns the summed kernel time and the
 * launch count since the previous call, clears them, and sets the new enable state. */
int nm_profile(nm_env* env, int32_t enable, double* sum_ms, int64_t* count);
/* Stage-skipping measurement switches are NOT part of this ABI: they exist only in the -DNM_MEASURE build
 * (libnightmare_hip_measure.so, include/nightmare_hip_measure.h). */

/* ---- ActorCritic MLP forward on the matrix cores (rsl_rl v1.0.2 ActorCritic: Linear -> ELU x n_hidden -> Linear; reference call
 * sites play.py:122 `nn.act(obs)`, train.py:40), batched over envs, exact-f32 MFMA, all layers in one launch.
 * One handle = one network: it owns a packed copy of the parameters, so nothing is shared between networks, streams or devices.
 * dims = {n_in, h1, ..., n_out}, n_layers = len(dims) - 1. Networks of <= 4 layers and <= 256 units run fused; others per layer. */
typedef struct nm_policy nm_policy;
int nm_policy_create(const int32_t* dims, int32_t n_layers, int32_t device, nm_policy** out);
int nm_policy_destroy(nm_policy* h);
/* Copy + repack the parameters (device f32, torch.nn.Linear layout: weights[l] is [out_l, in_l] row-major, bias[l] is [out_l]).
 * Stream-ordered; call again after every optimiser step / load_state_dict - the handle never reads the caller's tensors later. */
int nm_policy_load(nm_policy* h, const float* const* weights_dev, const float* const* bias_dev, void* stream);
/* out_dev[N, n_out] = MLP(obs_dev[N, n_in]) with the parameters of the last nm_policy_load. One launch on the fused path. */
int nm_policy_forward(nm_policy* h, const float* obs_dev, int32_t num_envs, float* out_dev, void* stream);

/* GAE(lambda) returns of one rollout (rsl_rl v1.0.2 RolloutStorage.compute_returns; caller reference train.py:54).
 * rewards/values/returns [T,N] f32, dones [T,N] u8, last_values [N] f32, all device memory. */
int nm_gae(const float* rewards_dev, const float* values_dev, const unsigned char* dones_dev, const float* last_values_dev,
           int32_t T, int32_t N, float gamma, float lam, float* returns_dev, void* stream);
/* The same plus the rest of RolloutStorage.compute_returns (rsl_rl v1.0.2 `storage/rollout_storage.py`; caller reference train.py:54):
 * advantages[T,N] = returns - values and, normalize != 0, (advantages - mean) / (std + 1e-8) over all T x N entries (unbiased std) - two
 * launches, reproducible summation order. scratch_dev: 2 * ceil(N / 256) floats. One GPU's envs only: a multi-rank job normalises over
 * all ranks (normalize = 0 here, then the caller's all-reduced statistics). */
int nm_gae_advantages(const float* rewards_dev, const float* values_dev, const unsigned char* dones_dev, const float* last_values_dev, int32_t T, int32_t N,
                      float gamma, float lam, float* returns_dev, float* advantages_dev, float* scratch_dev, int32_t normalize, void* stream);

/* Rollout collection of the on-policy loop (rsl_rl v1.0.2 PPO.act / PPO.process_env_step; caller reference train.py:54), one launch each.
 * nm_ppo_sample: net_out_dev [N, A+1] = action means | critic value (nm_policy_forward on the merged actor+critic network), std_dev [A];
 *   draws a = mean + std * N(0,1) from the counter generator keyed by (seed, *iter_dev, step, env, j) and writes step `step` of the rollout
 *   storage: actions/mu/sigma [N,A], log-probability and value [N], and (if obs_store_dev != NULL) a copy of obs_dev [N, n_obs].
 * nm_ppo_record: rewards_store = rew + gamma * value * time_out (time_outs_dev may be NULL), dones_store (u8), running episode return /
 *   length per env, and fin3_dev += (sum of returns, sum of lengths, count) over the episodes that ended in this step; if n_ep > 0 also the
 *   runner's sum of extras['episode'] over the rollout: ep_acc_dev[i] += ep_stats_dev[ep_idx_dev[i]], i < n_ep <= 256. */
int nm_ppo_sample(const float* net_out_dev, const float* std_dev, const float* obs_dev, int32_t N, int32_t A, int32_t n_obs, uint64_t seed,
                  const int64_t* iter_dev, int32_t step, float* actions_dev, float* logp_dev, float* values_dev, float* mu_dev, float* sigma_dev,
                  float* obs_store_dev, void* stream);
int nm_ppo_record(const float* rew_dev, const int64_t* done_dev, const float* time_outs_dev, const float* values_dev, float gamma, int32_t N,
                  float* rewards_store_dev, unsigned char* dones_store_dev, float* cur_ret_dev, float* cur_len_dev, float* fin3_dev,
                  const float* ep_stats_dev, const int32_t* ep_idx_dev, int32_t n_ep, float* ep_acc_dev, void* stream);

/* ---- PPO mini-batch update (rsl_rl v1.0.2 `algorithms/ppo.py` PPO.update: clipped surrogate + clipped value loss + entropy bonus,
 * adaptive-KL learning rate, gradient-norm clipping, Adam; caller reference train.py:54; hyper-parameters envs/nightmare_v3_config.py:111-128)
 * as two launches per mini-batch (forward/backward; then partial-gradient reduction, gradient-norm clip, KL-adaptive learning rate, Adam and
 * the repacking of the weights in one launch with one grid barrier) with no host synchronisation. actor_dims / critic_dims = {n_obs, h1, ..., n_out} (same depth, same
 * observation, critic output 1, ELU). The parameters live in ONE caller-owned flat device vector in the order
 * actor W0 b0 W1 b1 ..., critic W0 b0 ..., std[A] (W row-major [out, in] as torch.nn.Linear); Adam's moments in two more such vectors. */
typedef struct nm_ppo nm_ppo;
int nm_ppo_create(const int32_t* actor_dims, const int32_t* critic_dims, int32_t n_layers, int32_t device, nm_ppo** out);
int nm_ppo_destroy(nm_ppo* h);
int32_t nm_ppo_num_params(const nm_ppo* h);
/* (re)read the flat parameters (after load_state_dict or a step taken elsewhere) and set the learning rate / Adam step count */
int nm_ppo_sync_params(nm_ppo* h, const float* flat_dev, float lr, int64_t step, void* stream);
/* one mini-batch: rows of obs [B,n_obs], actions / old_mu / old_sigma [B,A], old_logp / adv / ret / target_values [B] (device f32).
 * phase 0 = everything; 1 = gradient only (fetch it with nm_ppo_copy_grad, e.g. for an all-reduce); 2 = the step from the current gradient, the KL of the
 * learning-rate rule read from the slot behind it (kl_override >= 0 replaces the KL mean in either phase, < 0 keeps it).
 * Data-parallel update: phase 1, nm_ppo_copy_grad(0), all-reduce + divide by the ranks, nm_ppo_copy_grad(1), phase 2 - one collective
 * per mini-batch, no host synchronisation. */
int nm_ppo_minibatch(nm_ppo* h, float* flat_dev, float* exp_avg_dev, float* exp_avg_sq_dev, const float* obs, const float* actions,
                     const float* old_mu, const float* old_sigma, const float* old_logp, const float* adv, const float* ret, const float* target_values,
                     int32_t B, int32_t n_obs, float clip, float value_coef, float entropy_coef, int32_t clip_value, float desired_kl,
                     int32_t adaptive, float max_grad_norm, float beta1, float beta2, float eps, int32_t phase, float kl_override, void* stream);
/* nm_ppo_minibatch with the mini-batch given as ROW NUMBERS into the whole rollout (rsl_rl's mini_batch_generator gathers obs[batch_idx], ...;
 * here the forward / backward kernel gathers the rows itself, nothing is copied): row i of the mini-batch is row rows_dev[i] (int32, device)
 * of the [T*N, .] arrays. rows_dev == NULL = rows 0..B-1 (nm_ppo_minibatch). */
int nm_ppo_minibatch_rows(nm_ppo* h, float* flat_dev, float* exp_avg_dev, float* exp_avg_sq_dev, const float* obs, const float* actions,
                          const float* old_mu, const float* old_sigma, const float* old_logp, const float* adv, const float* ret, const float* target_values,
                          const int32_t* rows_dev, int32_t B, int32_t n_obs, float clip, float value_coef, float entropy_coef, int32_t clip_value, float desired_kl,
                          int32_t adaptive, float max_grad_norm, float beta1, float beta2, float eps, int32_t phase, float kl_override, void* stream);
/* Declares how many rows the [T*N, .] arrays behind a row list hold (row numbers are in [0, n_rows)); must precede nm_ppo_minibatch_rows with
 * rows_dev != NULL. LIMIT: the kernels address a row as base + a 32-bit byte offset, so n_rows * max(n_obs, n_actions) * 4 must stay below
 * 2^32 (16.2 M rows of 66 floats; BASELINE config 5 has 32768 x 80 = 2.6 M); beyond it this call - and nm_ppo_minibatch for B - fails. */
int nm_ppo_set_storage_rows(nm_ppo* h, int64_t n_rows);
/* 1 if the mini-batch step of this handle is the one-launch k_ppo_step (one grid barrier), 0 if it is the four-launch chain: asked for with
 * NM_PPO_UNFUSED_STEP=1, chosen at creation because the device cannot hold the step's grid at once (occupancy query), or after a barrier
 * time-out. A barrier that times out makes that step and every later fused step a NO-OP (no parameter, moment or packed weight is written)
 * until nm_ppo_get_state has reported it. */
int32_t nm_ppo_step_is_fused(const nm_ppo* h);
/* TEST HOOK: the next fused step's grid barrier will time out (about one second of spinning), to exercise the no-op path. */
int nm_ppo_debug_break_barrier(nm_ppo* h, void* stream);
/* The mini-batch order of one PPO.update (rsl_rl v1.0.2 mini_batch_generator: indices = torch.randperm(num_mini_batches * mini_batch_size)):
 * out_dev[i] (int32) = image of i under a pseudo-random permutation of 0..n-1 keyed by (seed, counter) - a cycle-walked Feistel network,
 * one launch, no sort. */
int nm_ppo_permutation(int32_t* out_dev, int32_t n, uint64_t seed, uint64_t counter, void* stream);
/* 1 if the network runs on the compiled register-resident kernels (the reference's 66 -> 54 -> 42 -> 30 -> 18 | 1 shape), else 0 */
int32_t nm_ppo_has_fast_path(const nm_ppo* h);
/* rsl_rl v1.0.2 PPO.act (caller reference train.py:54) in ONE launch, fast-path networks only: merged actor+critic forward from the
 * update's packed weights (always current: no repack between update and collection), then exactly what nm_ppo_sample does */
int nm_ppo_act(nm_ppo* h, const float* flat_dev, const float* obs_dev, int32_t N, uint64_t seed, const int64_t* iter_dev, int32_t step,
               float* actions_dev, float* logp_dev, float* values_dev, float* mu_dev, float* sigma_dev, float* obs_store_dev, void* stream);
/* nm_ppo_record of the PREVIOUS step and nm_ppo_act of this one in one launch (the record part runs first; arguments as in the two
 * calls, prev_values_dev = the values nm_ppo_act filed for the previous step): PPO.process_env_step(s - 1) + PPO.act(s) of rsl_rl
 * v1.0.2 (caller reference train.py:54). A rollout of T steps needs T + 1 of these launches besides the env's instead of 2 T. */
int nm_ppo_record_act(nm_ppo* h, const float* rew_dev, const int64_t* done_dev, const float* time_outs_dev, const float* prev_values_dev, float gamma,
                      float* rewards_store_dev, unsigned char* dones_store_dev, float* cur_ret_dev, float* cur_len_dev, float* fin3_dev,
                      const float* ep_stats_dev, const int32_t* ep_idx_dev, int32_t n_ep, float* ep_acc_dev,
                      const float* flat_dev, const float* obs_dev, int32_t N, uint64_t seed, const int64_t* iter_dev, int32_t step,
                      float* actions_dev, float* logp_dev, float* values_dev, float* mu_dev, float* sigma_dev, float* obs_store_dev, void* stream);
/* gradient of the last mini-batch in flat order followed by the mini-batch's mean KL to the behaviour policy, [num_params + 1] floats:
 * direction 0 copies them to grad_dev, 1 replaces them by grad_dev */
int nm_ppo_copy_grad(nm_ppo* h, float* grad_dev, int32_t direction, void* stream);
/* The same vector in a buffer the caller owns ([num_params + 1] floats on the device): phase 1 writes gradient | KL there, phase 2 reads it
 * from there, and a data-parallel update all-reduces it IN PLACE between the two - one collective per mini-batch and nothing else (the 60 KB
 * gradient all-reduce of SURVEY 8(e); rsl_rl v1.0.2 itself is single-process, caller reference train.py:54). NULL = the handle's own buffer. */
int nm_ppo_set_grad_buffer(nm_ppo* h, float* grad_kl_dev);
/* HOST out[8]: lr, Adam steps, last KL, sum of value losses, sum of surrogate losses, mini-batches, clip coefficient, grad norm;
 * reset_sums != 0 clears the two loss sums and the count afterwards. Synchronises the stream. */
int nm_ppo_get_state(nm_ppo* h, float* out8_host, int32_t reset_sums, void* stream);
/* DEVICE out[9]: the same eight values and, [8], != 0 if the fused step's grid barrier timed out in the launches since the last read.
 * Stream-ordered, no host synchronisation (rsl_rl's OnPolicyRunner.log reads its statistics after every update, caller reference
 * train.py:54; the runner here reads this snapshot one iteration later, while the next rollout is already running). */
int nm_ppo_snapshot_state(nm_ppo* h, float* out9_dev, int32_t reset_sums, void* stream);

/* ---- The collection loop of rsl_rl v1.0.2 OnPolicyRunner.learn (`for i in range(num_steps_per_env): actions = alg.act(obs, critic_obs);
 * obs, _, rewards, dones, infos = env.step(actions); alg.process_env_step(rewards, dones, infos)`; caller reference train.py:54, horizon
 * envs/nightmare_v3_config.py:135) as ONE launch of `steps` steps: every wavefront keeps its two envs for the whole rollout and evaluates
 * the policy itself between two physics steps (no launch per step, no wait for the step's slowest env). Networks of the reference's shape
 * (envs/nightmare_v3_config.py:105-109; nm_rollout_supported) on the fp32 env only. Results per step are those of the step-by-step path
 * nm_rollout_act + nm_step + nm_ppo_record: bit-identical observations / actions / values / log-probabilities / rewards / dones in the storage
 * (sums that go through float atomics - fin3, ep_acc, ep_stats - agree to rounding). All pointers are device memory except the dims. */
typedef struct {
  int32_t steps;                       /* K, at most the episode length in steps */
  const float* params_flat_dev;        /* actor W0 b0 W1 b1 ..., critic W0 b0 ..., std[18] (the flat vector of nm_ppo_*) */
  uint64_t seed;                       /* action noise: counter generator keyed by (seed, *iter_dev, step, env, action pair) like nm_ppo_sample */
  const int64_t* iter_dev;
  const float* obs0_dev;               /* [N,66] the observation the first act sees (the env's current observation) */
  float* obs_final_dev;                /* [N,66] receives the observation after the last step (may alias obs0_dev) */
  int64_t* episode_length_dev;         /* [N] episode_length_buf, in/out */
  float* rew_dev; int64_t* done_dev;   /* [N] the env's reward / reset buffers: hold the last step's values afterwards */
  float* time_outs_dev;                /* [N] extras['time_outs'] in/out (NULL: not kept), ep_stats_dev [NM_NUM_REWARDS] extras['episode'] in/out */
  float* ep_stats_dev;
  int32_t bootstrap_time_outs;         /* 1: rewards += gamma * value * extras['time_outs'] (what PPO.process_env_step does when the env sends time_outs,
                                          cfg.env.send_timeouts, envs/nightmare_v3_env.py:369); 0: the buffer is kept up to date, the rewards stay raw */
  float *s_obs, *s_actions, *s_logp, *s_values, *s_mu, *s_sigma, *s_rewards;   /* rollout storage rows [K,N,66] [K,N,18] [K,N] [K,N] [K,N,18] [K,N,18] [K,N] */
  unsigned char* s_dones;              /* [K,N] */
  float gamma;                         /* time-out bootstrap: rewards += gamma * value * extras['time_outs'] (PPO.process_env_step) */
  float *cur_ret, *cur_len, *fin3;     /* [N] [N] [3]: as nm_ppo_record */
  const int32_t* ep_idx_dev; int32_t n_ep; float* ep_acc_dev;   /* ep_acc[i] += extras['episode'][ep_idx[i]] after every step, i < n_ep <= NM_NUM_REWARDS */
  float* last_values_dev;              /* [N] or NULL: the critic's value of the observation after the last step - rsl_rl PPO.compute_returns'
                                          `last_values = actor_critic.evaluate(last_critic_obs)` - evaluated by the env's wave at the end of the launch */
} nm_rollout_args;
/* 1 if nm_rollout / nm_rollout_act are compiled for these networks (dims = {n_obs, h1, h2, h3, n_out}, HOST arrays) */
int nm_rollout_supported(const int32_t* actor_dims, const int32_t* critic_dims, int32_t n_layers);
int nm_rollout(nm_env* env, const nm_rollout_args* args, void* stream);
/* PPO.act alone, on the code the rollout's waves run (one wave = two envs): the per-step counterpart of nm_rollout. Arguments as nm_ppo_act. */
int nm_rollout_act(nm_env* env, const float* params_flat_dev, const float* obs_dev, uint64_t seed, const int64_t* iter_dev, int32_t step,
                   float* actions_dev, float* logp_dev, float* values_dev, float* mu_dev, float* sigma_dev, float* obs_store_dev, void* stream);

/* ---- scripted gait / IK engine (reference nikengine/engine.py; caller custom_play.py:49-76), batched over envs ----
 * One handle = num_envs independent EngineNode objects (engine.py:660-677), all in IdleState. */
typedef struct nm_nik nm_nik;
nm_nik* nm_nik_create(int32_t num_envs, int32_t device);
void nm_nik_destroy(nm_nik* h);
/* back to IdleState with the default pose (a fresh EngineNode). ids_host NULL = all envs. */
int nm_nik_reset(nm_nik* h, const int32_t* ids_host, int32_t n, void* stream);
/* state.cmd.gait (engine.py:297; the gait table :214-225): 0 'tripod' (default), 1 'ripple', 2 'wave'; ids_host NULL = all envs. Upstream the
 * field sits on a Command object shared by every EngineNode (RobotState.cmd is a class attribute, :402-406) and EngineNode.update does not
 * touch it; here it is per env. Takes effect like upstream: when WalkState is entered (:543) or the running step completes (:627). */
int nm_nik_set_gait(nm_nik* h, const int32_t* ids_host, int32_t n, int32_t gait, void* stream);
/* EngineNode.update(lin_speed, ang_speed, state, mode) (engine.py:710-715) for every env, one launch.
 *   lin_dev, ang_dev [N] f64 device: walk translation along +y (m/s) and yaw rate (rad/s) commands
 *   awake_dev, walk_dev [N] u8 device or NULL: state == 'awake' (else 'idle'), mode == 'walk' (else 'stand'); NULL = 1
 *   now_s: the engine clock (set_time_s, engine.py:12-19); engine_fps: config.ENGINE_FPS, read every tick like upstream
 *   angles_f32_dev / angles_f64_dev [N,18] device, either may be NULL: the 18 joint targets set_hardware_pose returns
 *   (IK + SERVO_OFFSET + URDF_JOINT_OFFSETS, engine.py:700-708). Arithmetic is f64. */
int nm_nik_update(nm_nik* h, const double* lin_dev, const double* ang_dev, const unsigned char* awake_dev,
                  const unsigned char* walk_dev, double now_s, double engine_fps, float* angles_f32_dev,
                  double* angles_f64_dev, void* stream);
/* FSM inspection for tests: HOST pose [N,18] (foot positions in the body frame), fsm id [N]
 * (0 idle 1 adjust-get-up 2 get-up 3 sit 4 adjust-sit 5 stand 6 walk), gait_step_state [N]. NULL = skip. Synchronous. */
int nm_nik_get_state(nm_nik* h, double* pose_host, int32_t* fsm_host, double* gait_step_state_host);

#ifdef __cplusplus
}
#endif
#endif
