"""NightmareV3Config / NightmareV3ConfigPPO with the attribute tree and defaults the reference exposes
(reference envs/nightmare_v3_config.py:4-146), so `cfg.env.num_envs = n`, `class_to_dict(train_cfg)` and
train.py / play.py-shaped scripts work unchanged.

The tree is declared as data (`_ENV_SPEC`, `_PPO_SPEC`) and turned into nested classes by `_build`; after
`BaseConfig.__init__` the nested classes are instances, exactly as upstream. Two defaults differ on purpose and are
called out: `device` ('cuda': the env state lives in HBM) and `viewer.render/record_states` (False: no viewer here).
"""
import math

from .base_config import BaseConfig

_PI_5 = math.pi / 5

_ENV_SPEC = {
    "device": "cuda",      # upstream 'cpu' (numpy on the host)
    "rl_device": "cuda",
    "env": {
        "model_path": "models/nightmare_v3/mjmodel.xml",  # kept for API parity; the model is compiled into the kernels
        "num_envs": 8192, "num_obs": 66, "num_privileged_obs": 0, "num_actions": 18,
        "episode_length_s": 20, "send_timeouts": True, "body_name": "base_link",
        # contact modes: 0 ignore, 1 penalise (the upstream default), 2 terminate - all three are compiled and tested (env.py:248-251,479-485)
        "tibia_contact_mode": 1, "tibia_max_contact_force": 2.0,
        "body_contact_mode": 1, "body_max_contact_force": 2.0,
        "termination_contact_force": 160.0,
    },
    "viewer": {"render": False, "record_states": False},  # upstream True/True
    "control": {"p_gain": 20, "default_pos": [0, _PI_5, 0] * 6, "decimation": 2, "action_scale": 0.2},
    "noise": {
        "add_noise": False, "noise_level": 0.1,
        "noise_scales": {"lin_vel": 1.0, "ang_vel": 1.0, "gravity": 1.0, "dof_pos": 1.0, "dof_vel": 1.0, "height_measurements": 1.0},
    },
    "commands": {"resampling_time": 10, "ranges": {"max_lin_vel_x": 0.5, "max_lin_vel_y": 0.5, "max_ang_vel": 0.8}},
    "normalization": {
        "obs_scales": {"lin_vel": 2.0, "ang_vel": 0.25, "dof_pos": 1.0, "dof_vel": 0.05, "height_measurements": 5.0},
        "clip_observations": 100.0, "clip_actions": 1.0,
    },
    "rewards": {
        "scales": {
            "termination": -200.0, "tracking_lin_vel": 8.0, "tracking_ang_vel": 6.0, "dof_acc": -2.5e-5, "action_rate": -0.02,
            "body_contact_forces": -5, "default_position": -0.01, "orientation": -5,
            # present upstream with scale 0 (their reward functions never run)
            "lin_vel_z": 0, "ang_vel_xy": 0, "feet_air_time": 0, "torques": 0, "base_height": 0, "feet_contact_forces": 0,
            "dof_vel": 0, "stand_still": 0, "collision": 0, "feet_stumble": 0,
        },
        "tracking_sigma": 0.008, "base_height_target": 0.1, "max_contact_force": 10.0,
    },
}

_PPO_SPEC = {
    "seed": 1,
    "runner_class_name": "OnPolicyRunner",
    "policy": {"init_noise_std": 1.0, "actor_hidden_dims": [54, 42, 30], "critic_hidden_dims": [54, 42, 30], "activation": "elu"},
    "algorithm": {
        "value_loss_coef": 1.0, "use_clipped_value_loss": True, "clip_param": 0.2, "entropy_coef": 0.0015,
        "num_learning_epochs": 5, "num_mini_batches": 4, "learning_rate": 1.0e-3, "schedule": "adaptive",
        "gamma": 0.99, "lam": 0.95, "desired_kl": 0.01, "max_grad_norm": 1.0,
    },
    "runner": {
        "policy_class_name": "ActorCritic", "algorithm_class_name": "PPO", "num_steps_per_env": 80, "max_iterations": 1000000000,
        "save_interval": 50, "experiment_name": "test", "run_name": "", "resume": False, "load_run": -1, "checkpoint": -1,
        "resume_path": None,
    },
}


def _build(name, spec, bases=()):
    ns = {}
    for key, val in spec.items():
        ns[key] = _build(key, val) if isinstance(val, dict) else (list(val) if isinstance(val, list) else val)
    return type(name, bases, ns)


NightmareV3Config = _build("NightmareV3Config", _ENV_SPEC, (BaseConfig,))
NightmareV3ConfigPPO = _build("NightmareV3ConfigPPO", _PPO_SPEC, (BaseConfig,))
