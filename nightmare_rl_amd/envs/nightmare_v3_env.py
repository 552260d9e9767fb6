"""NightmareV3Env: the reference's vectorised env surface (reference envs/nightmare_v3_env.py:26-396) over the
MI355X-native step kernel. Same constructor, attributes and return tuples; state lives in HBM and every call is
asynchronous on the current torch stream.

Differences a caller can observe (all documented in DESIGN.md):
  * returned tensors are views of persistent device buffers (the reference returns fresh CPU tensors, :311). The observation
    returned by step t stays untouched through step t+1 (two buffers, alternating), so a caller that keeps it across the next
    env.step and copies it afterwards - rsl_rl's PPO.act -> env.step -> storage.add_transitions - stores what the policy saw;
    rewards / dones / time_outs are valid until the next step()
  * commands come from a counter-based generator keyed by (seed, global env id) instead of numpy's global RNG (:327-330)
  * no viewer (`cfg.viewer.render` must be False); `cfg.viewer.record_states` writes the reference's pickle log (:261-272)
    at the price of one device sync per step
  * observation noise (`cfg.noise.add_noise`, :304-305) draws from the same counter generator; `cfg.noise.layout`
    chooses between the reference's noise_scale_vec (written for a 12-dof robot, :113-119) and the 18-dof one
"""
import ctypes as C
import os
import pickle
import time

import numpy as np
import torch

from .. import _lib
from .helpers import class_to_dict
from .nightmare_v3_config import NightmareV3Config


class NightmareV3Env:
    def __init__(self, cfg: NightmareV3Config, log_dir="/tmp/nightmare_v3/logs", num_threads=1, *, device=None, seed=0,
                 env_id_offset=0, dtype=torch.float32, lib=None):
        """lib: a loaded library object other than the shipped one (_lib.load_measure(): measurement scripts and one test only)."""
        self.cfg = cfg
        self.log_dir = log_dir
        self.thread_num = num_threads  # accepted for API parity; the GPU path has no host threads
        self.num_envs = int(cfg.env.num_envs)
        self.num_obs = int(cfg.env.num_obs)
        self.num_privileged_obs = self.num_obs  # reference :34
        self.num_actions = int(cfg.env.num_actions)
        if self.num_obs != _lib.NUM_OBS or self.num_actions != _lib.NUM_ACTIONS:
            raise ValueError("the compiled path is specialised for num_obs=66, num_actions=18")
        if cfg.viewer.render:
            raise ValueError("cfg.viewer.render is not available on the GPU path (set it to False)")
        if cfg.env.tibia_contact_mode not in (0, 1, 2) or cfg.env.body_contact_mode not in (0, 1, 2):
            raise ValueError("tibia/body_contact_mode: 0 do nothing, 1 penalize on contact, 2 terminate on contact")
        if not torch.cuda.is_available():
            raise _lib.NightmareHipError("NightmareV3Env needs a HIP device: there is no CPU path")
        L = lib if lib is not None else _lib.load()
        self._L = L
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dev.type != "cuda":
            raise _lib.NightmareHipError(f"device must be a HIP device, got {dev}")
        self.device = dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())
        self.num_dof = 18
        self.gravity_vec = np.array([0., 0., -9.81])
        self.reward_scales = class_to_dict(cfg.rewards.scales)
        self.command_ranges = cfg.commands.ranges
        self.obs_scales = cfg.normalization.obs_scales
        # timestep 0.008 (reference models/nightmare_v3/mjmodel.xml:3) x decimation
        self.dt = 0.008 * cfg.control.decimation
        self.max_episode_length_s = cfg.env.episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.dt)
        self.default_dof_pos = np.array(cfg.control.default_pos, dtype=np.float64)
        if int(cfg.commands.resampling_time / self.dt) < 1:
            raise ValueError("cfg.commands.resampling_time must be at least one env step (reference :235 takes a modulo by it)")
        # reward table: zero scales dropped, the rest x dt (reference :123-128). Every name the reference has a _reward_ function
        # for (:399-497) is compiled; a name without one (`collision`, `feet_stumble`, config :95-96) fails like upstream's getattr.
        names = [L.nm_reward_name(i).decode() for i in range(_lib.NUM_REWARDS)]
        for key in list(self.reward_scales.keys()):
            if self.reward_scales[key] == 0:
                self.reward_scales.pop(key)
            else:
                if key not in names:
                    raise AttributeError(f"'NightmareV3Env' object has no attribute '_reward_{key}'")
                self.reward_scales[key] *= self.dt
        self.reward_names = [n for n in self.reward_scales if n != "termination"]
        dp = list(cfg.control.default_pos)
        if any(abs(dp[i] - dp[i % 3]) > 0 for i in range(18)):
            raise NotImplementedError("default_pos must repeat per leg (coxa, femur, tibia)")
        c = _lib.NmConfig()
        L.nm_default_config(C.byref(c))
        c.decimation = int(cfg.control.decimation)
        c.p_gain = float(cfg.control.p_gain)
        c.action_scale = float(cfg.control.action_scale)
        for i in range(3):
            c.default_pos[i] = float(dp[i])
        c.clip_actions = float(cfg.normalization.clip_actions)
        c.clip_observations = float(cfg.normalization.clip_observations)
        c.obs_lin_vel, c.obs_ang_vel = float(self.obs_scales.lin_vel), float(self.obs_scales.ang_vel)
        c.obs_dof_pos, c.obs_dof_vel = float(self.obs_scales.dof_pos), float(self.obs_scales.dof_vel)
        c.episode_length_s = float(cfg.env.episode_length_s)
        c.resampling_time = float(cfg.commands.resampling_time)
        c.max_lin_vel_x, c.max_ang_vel = float(self.command_ranges.max_lin_vel_x), float(self.command_ranges.max_ang_vel)
        c.termination_contact_force = float(cfg.env.termination_contact_force)
        c.tracking_sigma = float(cfg.rewards.tracking_sigma)
        raw = class_to_dict(cfg.rewards.scales)
        for i, n in enumerate(names):
            c.reward_scales[i] = float(raw.get(n, 0.0))
        c.tibia_contact_mode, c.tibia_max_contact_force = int(cfg.env.tibia_contact_mode), float(cfg.env.tibia_max_contact_force)
        c.body_contact_mode, c.body_max_contact_force = int(cfg.env.body_contact_mode), float(cfg.env.body_max_contact_force)
        c.base_height_target, c.max_contact_force = float(cfg.rewards.base_height_target), float(cfg.rewards.max_contact_force)
        self._dtype = _lib.DTYPE_F64 if dtype == torch.float64 else _lib.DTYPE_F32
        h = C.c_void_p()
        self._ck(L.nm_create(C.byref(c), self.num_envs, self.device.index or 0, int(seed), int(env_id_offset), self._dtype, C.byref(h)))
        self._h = h
        N, dev = self.num_envs, self.device
        # two observation buffers, written alternately: the tensor handed out by step t is not overwritten by step t+1
        self._obs_pair = torch.zeros((2, N, self.num_obs), dtype=torch.float32, device=dev)
        self._obs_idx = 0
        self.obs_buf = self._obs_pair[0]
        self.privileged_obs_buf = None
        self.rew_buf = torch.zeros(N, dtype=torch.float32, device=dev)
        self.reset_buf = torch.ones(N, dtype=torch.int64, device=dev)
        self.episode_length_buf = torch.zeros(N, dtype=torch.int64, device=dev)  # assignable, like the reference (:88)
        self.time_out_buf = torch.zeros(N, dtype=torch.float32, device=dev)
        self._to_bound = self.time_out_buf       # the tensor object the kernel's incremental time-out refresh is bound to
        self._ep_stats = torch.zeros(_lib.NUM_REWARDS, dtype=torch.float32, device=dev)
        self._stat_names = names
        self.extras = {}
        self.common_step_counter = 0
        # observation noise (reference :109-119, :304-305)
        self.noise_scale_vec = self._noise_scale_vec(cfg)
        self.add_noise = bool(cfg.noise.add_noise)
        if self.add_noise:
            self._ck(L.nm_set_observation_noise(h, self.noise_scale_vec.ctypes.data_as(C.c_void_p)))
        # state log of env 0 (reference :261-272; reader open_custom_play.py:50-66)
        self.recorded_states = []
        self._rec_time = 0.0
        if cfg.viewer.record_states:
            self._ck(L.nm_set_state_record(h, 0))

    def _noise_scale_vec(self, cfg):
        ns, lvl, osc = cfg.noise.noise_scales, cfg.noise.noise_level, self.obs_scales
        v = np.zeros(self.num_obs)
        v[:3] = ns.lin_vel * lvl * osc.lin_vel
        v[3:6] = ns.ang_vel * lvl * osc.ang_vel
        v[6:9] = ns.gravity * lvl
        layout = getattr(cfg.noise, "layout", "reference")
        if layout == "reference":       # the upstream index ranges, kept verbatim (they assume 12 dofs)
            v[12:24] = ns.dof_pos * lvl * osc.dof_pos
            v[24:36] = ns.dof_vel * lvl * osc.dof_vel
        elif layout == "dof18":         # the ranges of this robot's observation (E8: dof_pos 12:30, dof_vel 30:48)
            v[12:30] = ns.dof_pos * lvl * osc.dof_pos
            v[30:48] = ns.dof_vel * lvl * osc.dof_vel
        else:
            raise ValueError("cfg.noise.layout must be 'reference' or 'dof18'")
        return v

    def set_noise_uniforms(self, u=None):
        """RNG-free noise for parity tests: [N,66] uniforms in [0,1) used instead of the generator (None = generator)."""
        u = None if u is None else np.ascontiguousarray(u, np.float64).reshape(self.num_envs, self.num_obs)
        self._ck(self._L.nm_set_noise_uniforms(self._h, None if u is None else u.ctypes.data_as(C.c_void_p)))

    def _record_state(self):
        # reference :261-272: when env 0 resets, dump what was logged so far, then log (time, qpos, qvel, act) of env 0
        # as it is after the physics and before reset_idx. This model has no actuator state: act is empty.
        qpos, qvel, nbad = np.empty(25), np.empty(24), C.c_int32(0)
        self._ck(self._L.nm_get_state_record(self._h, qpos.ctypes.data_as(C.c_void_p), qvel.ctypes.data_as(C.c_void_p), C.byref(nbad)))
        if bool(self.reset_buf[0].item()):
            os.makedirs(self.log_dir, exist_ok=True)
            with open(f"{self.log_dir}/{int(time.time())}.pkl", "wb") as f:
                pickle.dump(self.recorded_states, f)
            self.recorded_states = []
        sim_dt = 0.008 * self.cfg.control.decimation
        self._rec_time = sim_dt if nbad.value else self._rec_time + sim_dt   # mj_resetData restarts data.time
        self.recorded_states.append((self._rec_time, qpos, qvel, np.zeros(0)))

    def _ck(self, rc):
        _lib.check(rc, self._L)

    # ------------------------------------------------------------------ reference surface
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _eplen(self):
        b = self.episode_length_buf
        if b.device != self.device or b.dtype != torch.int64 or not b.is_contiguous():
            b = b.to(device=self.device, dtype=torch.int64).contiguous()
            self.episode_length_buf = b
        return b

    def _fill_extras(self):
        # reference :363-371: one 0-d float32 tensor per reward + time_outs; views of buffers the kernel refreshes
        # keys = the reward table (non-zero scales), in its order - what upstream's episode_sums holds (:140, :363-367)
        self.extras["episode"] = {"rew_" + n: self._ep_stats[self._stat_names.index(n)] for n in self.reward_scales}
        if self.cfg.env.send_timeouts:
            self.extras["time_outs"] = self.time_out_buf

    def step(self, actions):
        a = actions
        if a.device != self.device or a.dtype != torch.float32:
            a = a.to(device=self.device, dtype=torch.float32)
        if a.dim() != 2 or a.shape[0] != self.num_envs or a.shape[1] < 18:
            raise ValueError(f"actions must be [{self.num_envs}, 18], got {tuple(a.shape)}")
        if a.shape[1] != 18 or not a.is_contiguous():
            a = a[:, :18].contiguous()
        ep = self._eplen()
        self._obs_idx ^= 1
        self.obs_buf = self._obs_pair[self._obs_idx]
        if self.time_out_buf is not self._to_bound:
            # the caller replaced extras' time-out tensor: a caching allocator may hand out the old address again, which the kernel's
            # address check cannot tell from "the buffer I refreshed last" - make the next refresh a full rewrite
            self._ck(self._L.nm_invalidate_time_outs(self._h, self._stream()))
            self._to_bound = self.time_out_buf
            if "time_outs" in self.extras:
                self.extras["time_outs"] = self.time_out_buf
        self._ck(self._L.nm_step(self._h, a.data_ptr(), ep.data_ptr(), self.obs_buf.data_ptr(), self.rew_buf.data_ptr(),
                                   self.reset_buf.data_ptr(), self.time_out_buf.data_ptr(), self._ep_stats.data_ptr(), self._stream()))
        self._last_actions = a  # keep the input alive until the kernel has read it
        self.common_step_counter += 1
        if self.cfg.viewer.record_states:
            self._record_state()
        if "episode" not in self.extras:
            self._fill_extras()
        return self.obs_buf, None, self.rew_buf, self.reset_buf, self.extras

    # ------------------------------------------------------------------ K steps per launch with the policy in the env's wave
    def policy_rollout(self, steps, params_flat, seed, iter_dev, storage, gamma, cur_ret, cur_len, fin, ep=None, last_values=None):
        """`steps` iterations of rsl_rl's collection loop `act -> env.step -> process_env_step` (OnPolicyRunner.learn; reference
        train.py:54) as ONE launch (nm_rollout): starts from the current observation, files every transition into `storage` (a
        RolloutStorage with `steps` rows: observations, actions, values, log-probabilities, mu, sigma, rewards incl. the time-out bootstrap,
        dones), updates the runner's bookkeeping tensors (cur_ret / cur_len [N], fin [3], and ep = (ep_idx int32, ep_acc) for the running
        sum of extras['episode']) and leaves the env as `steps` calls of step() would: obs_buf / rew_buf / reset_buf / extras of the last step.
        params_flat: the flat parameter vector of FusedUpdate (actor W0 b0 ..., critic ..., std). last_values ([N] float32 on the device,
        optional) receives the critic's value of the last observation - what PPO.compute_returns evaluates next."""
        if self.cfg.viewer.record_states:
            raise ValueError("policy_rollout: cfg.viewer.record_states needs one launch per step")
        T = int(steps)
        if storage.num_transitions_per_env < T or storage.num_envs != self.num_envs or storage.privileged_observations is not None:
            raise ValueError("policy_rollout: storage must hold `steps` rows of this env's transitions (no privileged observations)")
        if self.time_out_buf is not self._to_bound:
            self._ck(self._L.nm_invalidate_time_outs(self._h, self._stream()))
            self._to_bound = self.time_out_buf
        ep_idx, ep_acc = ep if ep is not None else (None, None)
        a = _lib.NmRolloutArgs()
        a.steps, a.params_flat_dev, a.seed, a.iter_dev = T, params_flat.data_ptr(), int(seed), iter_dev.data_ptr()
        a.obs0_dev = self.obs_buf.data_ptr()
        self._obs_idx ^= 1
        self.obs_buf = self._obs_pair[self._obs_idx]           # the tensor handed out before the rollout stays what it was
        a.obs_final_dev = self.obs_buf.data_ptr()
        a.episode_length_dev = self._eplen().data_ptr()
        a.rew_dev, a.done_dev = self.rew_buf.data_ptr(), self.reset_buf.data_ptr()
        a.time_outs_dev = self.time_out_buf.data_ptr()
        a.bootstrap_time_outs = 1 if self.cfg.env.send_timeouts else 0
        a.ep_stats_dev = self._ep_stats.data_ptr()
        a.s_obs, a.s_actions, a.s_logp, a.s_values = (storage.observations.data_ptr(), storage.actions.data_ptr(), storage.actions_log_prob.data_ptr(),
                                                       storage.values.data_ptr())
        a.s_mu, a.s_sigma, a.s_rewards, a.s_dones = storage.mu.data_ptr(), storage.sigma.data_ptr(), storage.rewards.data_ptr(), storage.dones.data_ptr()
        a.gamma = float(gamma)
        a.cur_ret, a.cur_len, a.fin3 = cur_ret.data_ptr(), cur_len.data_ptr(), fin.data_ptr()
        a.ep_idx_dev, a.n_ep, a.ep_acc_dev = (ep_idx.data_ptr(), int(ep_idx.numel()), ep_acc.data_ptr()) if ep_idx is not None else (None, 0, None)
        a.last_values_dev = last_values.data_ptr() if last_values is not None else None
        self._ck(self._L.nm_rollout(self._h, C.byref(a), self._stream()))
        self._keep_rollout = (params_flat, iter_dev, storage, cur_ret, cur_len, fin, ep_idx, ep_acc, last_values)
        self.common_step_counter += T
        storage.step = T
        if "episode" not in self.extras:
            self._fill_extras()
        return self.obs_buf

    def policy_act(self, params_flat, obs, seed, iter_dev, step, storage):
        """PPO.act as one launch of the rollout's wave code (nm_rollout_act): the per-step counterpart of policy_rollout."""
        s = int(step)
        self._ck(self._L.nm_rollout_act(self._h, params_flat.data_ptr(), obs.data_ptr(), int(seed), iter_dev.data_ptr(), s, storage.actions[s].data_ptr(),
                                        storage.actions_log_prob[s].data_ptr(), storage.values[s].data_ptr(), storage.mu[s].data_ptr(),
                                        storage.sigma[s].data_ptr(), storage.observations[s].data_ptr(), self._stream()))
        return storage.actions[s]

    def actions_from_joint_targets(self, targets):
        """Policy-space actions that make the servo track absolute joint targets (the hook at reference :186):
        step() commands (action_scale*a - default_pos - dof_pos)*p_gain (:152-156,:183-188), so a = (q* + default_pos)/action_scale.
        Targets beyond clip_actions - default_pos saturate, as any action does."""
        d = getattr(self, "_default_t", None)
        if d is None:
            d = self._default_t = torch.as_tensor(self.default_dof_pos, dtype=torch.float32, device=self.device)
        return (targets.to(device=self.device, dtype=torch.float32) + d) / float(self.cfg.control.action_scale)

    def step_physics(self, actions):
        """mj_step x decimation only (no rewards/obs/reset): the 'dynamics+contact kernel' configuration."""
        a = actions.to(device=self.device, dtype=torch.float32)[:, :18].contiguous()
        self._ck(self._L.nm_step_physics(self._h, a.data_ptr(), self._stream()))
        self._last_actions = a

    def reset_idx(self, env_ids):
        if env_ids is None:
            ids, n = None, 0
        else:
            idsa = np.ascontiguousarray(torch.as_tensor(env_ids).detach().cpu().numpy().astype(np.int32))
            if idsa.size == 0:
                return
            ids, n = idsa.ctypes.data_as(C.c_void_p), int(idsa.size)
        ep = self._eplen()
        self._ck(self._L.nm_reset(self._h, ids, n, ep.data_ptr(), self._ep_stats.data_ptr(), self._stream()))
        self._fill_extras()

    def reset(self):
        self.reset_idx(None)
        obs, priv, _, _, _ = self.step(torch.zeros((self.num_envs, self.num_actions), device=self.device))
        return obs, priv

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return None

    def render(self):
        pass

    # ------------------------------------------------------------------ state access (parity tests, checkpoints)
    def get_state(self):
        N = self.num_envs
        qpos, qvel, qw = np.empty((N, 25)), np.empty((N, 24)), np.empty((N, 24))
        self._ck(self._L.nm_get_state(self._h, qpos.ctypes.data, qvel.ctypes.data, qw.ctypes.data))
        return qpos, qvel, qw

    def set_state(self, qpos=None, qvel=None, qacc_warmstart=None):
        arrs = [None if a is None else np.ascontiguousarray(a, np.float64) for a in (qpos, qvel, qacc_warmstart)]
        self._ck(self._L.nm_set_state(self._h, *[None if a is None else a.ctypes.data for a in arrs]))

    def get_buffers(self):
        N = self.num_envs
        out = dict(dof_pos=np.empty((N, 18)), dof_vel=np.empty((N, 18)), actions=np.empty((N, 18)), commands=np.empty((N, 3)),
                   episode_sums=np.empty((N, _lib.NUM_REWARDS)))
        self._ck(self._L.nm_get_buffers(self._h, *[out[k].ctypes.data for k in ("dof_pos", "dof_vel", "actions", "commands", "episode_sums")]))
        return out

    def set_buffers(self, dof_pos=None, dof_vel=None, actions=None, commands=None, episode_sums=None):
        arrs = [None if a is None else np.ascontiguousarray(a, np.float64) for a in (dof_pos, dof_vel, actions, commands, episode_sums)]
        self._ck(self._L.nm_set_buffers(self._h, *[None if a is None else a.ctypes.data for a in arrs]))

    def get_feet_state(self):
        """(feet_air_time [N,6] f64, last_contacts [N,6] u8, last_contacts_filt [N,6] u8): state of _reward_feet_air_time (:90-93)."""
        N = self.num_envs
        air, last, filt = np.empty((N, 6)), np.empty((N, 6), np.uint8), np.empty((N, 6), np.uint8)
        self._ck(self._L.nm_get_feet_state(self._h, air.ctypes.data, last.ctypes.data, filt.ctypes.data))
        return air, last, filt

    def set_feet_state(self, air=None, last=None, filt=None):
        f = lambda a, t: None if a is None else np.ascontiguousarray(a, t)
        arrs = [f(air, np.float64), f(last, np.uint8), f(filt, np.uint8)]
        self._ck(self._L.nm_set_feet_state(self._h, *[None if a is None else a.ctypes.data for a in arrs]))

    def set_command_uniforms(self, u):
        a = None if u is None else np.ascontiguousarray(u, np.float64).reshape(self.num_envs, 4)
        self._ck(self._L.nm_set_command_uniforms(self._h, None if a is None else a.ctypes.data))

    def counters(self):
        out = np.zeros(3, np.int64)
        self._ck(self._L.nm_get_counters(self._h, out.ctypes.data))
        return dict(contacts_dropped=int(out[0]), bad_state_resets=int(out[1]), hull_search_fallbacks=int(out[2]))

    def set_debug_buffer(self, t):
        self._dbg = t
        self._ck(self._L.nm_set_debug_buffer(self._h, None if t is None else t.data_ptr()))

    def accumulate_rewards_into(self, t):
        """t: float32 [num_envs] device tensor (or None = off). Every step() then adds its rewards to t inside the step kernel - the
        running return a runner keeps (`cur_reward_sum += rewards`) without a launch of its own. The caller zeroes / reads t."""
        if t is not None and (t.dtype != torch.float32 or t.numel() != self.num_envs or not t.is_contiguous() or t.device != self.obs_buf.device):
            raise ValueError("accumulate_rewards_into: contiguous float32 [num_envs] tensor on the env's device")
        self._ret_acc = t
        self._ck(self._L.nm_set_return_accumulator(self._h, None if t is None else t.data_ptr()))

    def profile(self, enable):
        """(sum_ms, count) of step-kernel HIP-event times since the last call; sets event recording on/off."""
        ms, cnt = C.c_double(0), C.c_int64(0)
        self._ck(self._L.nm_profile(self._h, int(bool(enable)), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    @property
    def commands(self):
        return self.get_buffers()["commands"]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            torch.cuda.synchronize(self.device)
            self._L.nm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
