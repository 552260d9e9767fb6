"""ctypes binding of libnightmare_hip.so (C ABI: include/nightmare_hip.h). No fallback: if the HIP library is
missing or no GPU is usable, this raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NM_HIP_LIB") or os.path.join(HERE, "csrc", "libnightmare_hip.so")   # NM_HIP_LIB: measurement builds
NUM_OBS, NUM_ACTIONS, NUM_REWARDS = 66, 18, 16
DTYPE_F32, DTYPE_F64 = 0, 1

EXPORTS = ["nm_default_config", "nm_reward_name", "nm_last_error", "nm_create", "nm_destroy", "nm_num_envs", "nm_dtype", "nm_reset",
           "nm_step", "nm_step_physics", "nm_get_state", "nm_set_state", "nm_get_buffers", "nm_set_buffers",
           "nm_set_command_uniforms", "nm_get_feet_state", "nm_set_feet_state", "nm_get_counters", "nm_set_debug_buffer", "nm_set_return_accumulator", "nm_invalidate_time_outs", "nm_rollout", "nm_rollout_act", "nm_rollout_supported", "nm_policy_create", "nm_policy_destroy", "nm_policy_load", "nm_policy_forward", "nm_profile", "nm_gae", "nm_gae_advantages", "nm_ppo_sample", "nm_ppo_record", "nm_ppo_create", "nm_ppo_destroy", "nm_ppo_num_params", "nm_ppo_sync_params", "nm_ppo_minibatch", "nm_ppo_minibatch_rows", "nm_ppo_set_storage_rows", "nm_ppo_step_is_fused", "nm_ppo_debug_break_barrier", "nm_ppo_permutation", "nm_ppo_copy_grad", "nm_ppo_set_grad_buffer", "nm_ppo_get_state", "nm_ppo_snapshot_state", "nm_ppo_has_fast_path", "nm_ppo_act", "nm_ppo_record_act",
           "nm_set_observation_noise", "nm_set_noise_uniforms", "nm_set_state_record", "nm_get_state_record",
           "nm_nik_create", "nm_nik_destroy", "nm_nik_reset", "nm_nik_set_gait", "nm_nik_update", "nm_nik_get_state"]


class NmConfig(C.Structure):
    _fields_ = [("decimation", C.c_int32), ("p_gain", C.c_double), ("action_scale", C.c_double), ("default_pos", C.c_double * 3),
                ("clip_actions", C.c_double), ("clip_observations", C.c_double),
                ("obs_lin_vel", C.c_double), ("obs_ang_vel", C.c_double), ("obs_dof_pos", C.c_double), ("obs_dof_vel", C.c_double),
                ("episode_length_s", C.c_double), ("resampling_time", C.c_double), ("max_lin_vel_x", C.c_double), ("max_ang_vel", C.c_double),
                ("termination_contact_force", C.c_double), ("tracking_sigma", C.c_double), ("reward_scales", C.c_double * NUM_REWARDS),
                ("tibia_contact_mode", C.c_int32), ("tibia_max_contact_force", C.c_double), ("body_contact_mode", C.c_int32),
                ("body_max_contact_force", C.c_double), ("base_height_target", C.c_double), ("max_contact_force", C.c_double)]


class NmRolloutArgs(C.Structure):     # nm_rollout_args of include/nightmare_hip.h
    _fields_ = [("steps", C.c_int32), ("params_flat_dev", C.c_void_p), ("seed", C.c_uint64), ("iter_dev", C.c_void_p),
                ("obs0_dev", C.c_void_p), ("obs_final_dev", C.c_void_p), ("episode_length_dev", C.c_void_p),
                ("rew_dev", C.c_void_p), ("done_dev", C.c_void_p), ("time_outs_dev", C.c_void_p), ("ep_stats_dev", C.c_void_p), ("bootstrap_time_outs", C.c_int32),
                ("s_obs", C.c_void_p), ("s_actions", C.c_void_p), ("s_logp", C.c_void_p), ("s_values", C.c_void_p), ("s_mu", C.c_void_p),
                ("s_sigma", C.c_void_p), ("s_rewards", C.c_void_p), ("s_dones", C.c_void_p), ("gamma", C.c_float),
                ("cur_ret", C.c_void_p), ("cur_len", C.c_void_p), ("fin3", C.c_void_p),
                ("ep_idx_dev", C.c_void_p), ("n_ep", C.c_int32), ("ep_acc_dev", C.c_void_p), ("last_values_dev", C.c_void_p)]


class NightmareHipError(RuntimeError):
    pass


_lib = None
_measure = None
MEASURE_LIB_PATH = os.path.join(HERE, "csrc", "libnightmare_hip_measure.so")


def _import_torch_first():
    """PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so). If this library is dlopen'ed before torch has been imported,
    the system runtime it links against comes first and the process ends up with two HIP runtimes - the later one sees no device
    ("nm_create: no HIP device available" although torch.cuda.is_available()). Importing torch first makes its runtime the process's."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def load():
    """Load the HIP extension. Raises if it has not been built (python __graft_entry__.py / make -C nightmare_rl_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NightmareHipError(f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                                "There is no CPU fallback.")
    _import_torch_first()
    _lib = _bind(C.CDLL(LIB_PATH), full=True)
    return _lib


def load_measure():
    """The -DNM_MEASURE build of the env entry points (include/nightmare_hip_measure.h): stage-skipping switches for scripts/ and
    for the device bit-equality test. Never used by the product path; pass it as NightmareV3Env(..., lib=load_measure())."""
    global _measure
    if _measure is None:
        if not os.path.exists(MEASURE_LIB_PATH):
            raise NightmareHipError(f"{MEASURE_LIB_PATH} not found: make -C nightmare_rl_amd/csrc measure")
        _import_torch_first()
        _measure = _bind(C.CDLL(MEASURE_LIB_PATH), full=False)
        _measure.nm_set_ablation.argtypes = [C.c_void_p, C.c_int32]
    return _measure


def _bind(L, full):
    vp = C.c_void_p
    L.nm_default_config.argtypes = [C.POINTER(NmConfig)]
    L.nm_default_config.restype = None
    L.nm_reward_name.argtypes = [C.c_int]
    L.nm_reward_name.restype = C.c_char_p
    L.nm_last_error.restype = C.c_char_p
    L.nm_create.argtypes = [C.POINTER(NmConfig), C.c_int32, C.c_int32, C.c_uint64, C.c_int64, C.c_int32, C.POINTER(vp)]
    L.nm_destroy.argtypes = [vp]
    L.nm_num_envs.argtypes = [vp]
    L.nm_dtype.argtypes = [vp]
    L.nm_reset.argtypes = [vp, vp, C.c_int32, vp, vp, vp]
    L.nm_step.argtypes = [vp] * 9
    L.nm_step_physics.argtypes = [vp, vp, vp]
    L.nm_get_state.argtypes = [vp] * 4
    L.nm_set_state.argtypes = [vp] * 4
    L.nm_get_buffers.argtypes = [vp] * 6
    L.nm_set_buffers.argtypes = [vp] * 6
    L.nm_set_command_uniforms.argtypes = [vp, vp]
    L.nm_get_feet_state.argtypes = [vp] * 4
    L.nm_set_feet_state.argtypes = [vp] * 4
    L.nm_get_counters.argtypes = [vp, vp]
    L.nm_set_debug_buffer.argtypes = [vp, vp]
    L.nm_set_return_accumulator.argtypes = [vp, vp]
    L.nm_invalidate_time_outs.argtypes = [vp, vp]
    if hasattr(L, "nm_rollout"):      # the measurement build of nm_hip.hip exports them too
        L.nm_rollout.argtypes = [vp, C.POINTER(NmRolloutArgs), vp]
        L.nm_rollout_act.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
        L.nm_rollout_supported.argtypes = [vp, vp, C.c_int32]
    L.nm_profile.argtypes = [vp, C.c_int32, vp, vp]
    L.nm_set_observation_noise.argtypes = [vp, vp]
    L.nm_set_noise_uniforms.argtypes = [vp, vp]
    L.nm_set_state_record.argtypes = [vp, C.c_int32]
    L.nm_get_state_record.argtypes = [vp, vp, vp, vp]
    if not full:        # the measurement build holds the env entry points only
        return L
    L.nm_gae.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_float, C.c_float, vp, vp]
    L.nm_gae_advantages.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_float, C.c_float, vp, vp, vp, C.c_int32, vp]
    L.nm_ppo_sample.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.nm_ppo_record.argtypes = [vp, vp, vp, vp, C.c_float, C.c_int32, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp]
    L.nm_ppo_create.argtypes = [vp, vp, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.nm_ppo_destroy.argtypes = [vp]
    L.nm_ppo_num_params.argtypes = [vp]
    L.nm_ppo_sync_params.argtypes = [vp, vp, C.c_float, C.c_int64, vp]
    L.nm_ppo_minibatch.argtypes = [vp] * 12 + [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_float, C.c_int32, C.c_float,
                                               C.c_float, C.c_float, C.c_float, C.c_int32, C.c_float, vp]
    L.nm_ppo_minibatch_rows.argtypes = [vp] * 13 + [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_float, C.c_int32, C.c_float,
                                                    C.c_float, C.c_float, C.c_float, C.c_int32, C.c_float, vp]
    L.nm_ppo_permutation.argtypes = [vp, C.c_int32, C.c_uint64, C.c_uint64, vp]
    L.nm_ppo_copy_grad.argtypes = [vp, vp, C.c_int32, vp]
    L.nm_ppo_set_grad_buffer.argtypes = [vp, vp]
    L.nm_ppo_has_fast_path.argtypes = [vp]
    L.nm_ppo_act.argtypes = [vp, vp, vp, C.c_int32, C.c_uint64, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.nm_ppo_record_act.argtypes = [vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp,
                                    vp, vp, C.c_int32, C.c_uint64, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.nm_ppo_get_state.argtypes = [vp, vp, C.c_int32, vp]
    L.nm_ppo_set_storage_rows.argtypes = [vp, C.c_int64]
    L.nm_ppo_step_is_fused.argtypes = [vp]
    L.nm_ppo_step_is_fused.restype = C.c_int32
    L.nm_ppo_debug_break_barrier.argtypes = [vp, vp]
    L.nm_ppo_snapshot_state.argtypes = [vp, vp, C.c_int32, vp]
    L.nm_policy_create.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.nm_policy_destroy.argtypes = [vp]
    L.nm_policy_load.argtypes = [vp, vp, vp, vp]
    L.nm_policy_forward.argtypes = [vp, vp, C.c_int32, vp, vp]
    L.nm_nik_create.argtypes = [C.c_int32, C.c_int32]
    L.nm_nik_create.restype = vp
    L.nm_nik_destroy.argtypes = [vp]
    L.nm_nik_destroy.restype = None
    L.nm_nik_reset.argtypes = [vp, vp, C.c_int32, vp]
    L.nm_nik_set_gait.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
    L.nm_nik_update.argtypes = [vp, vp, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp]
    L.nm_nik_get_state.argtypes = [vp, vp, vp, vp]
    return L


def check(rc, L=None):
    if rc != 0:
        raise NightmareHipError((L or load()).nm_last_error().decode())


def reward_names():
    L = load()
    return [L.nm_reward_name(i).decode() for i in range(NUM_REWARDS)]
