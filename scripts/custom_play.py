"""Headless counterpart of the reference's custom_play.py:49-76: the scripted gait engine drives the simulated robot.

Every env gets its own (lin, ang) command; the gait kernel turns it into 18 joint targets, a per-step rate limit
(custom_play.py:16,73) smooths them, and the env's PD->velocity servo (env.py:181-188) tracks them.
    python scripts/custom_play.py [num_envs] [seconds]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd import nikengine as nk
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env


def play(num_envs=64, seconds=12.0, lin=0.05, ang=0.0, action_rate=0.08, seed=0, device="cuda:0"):
    cfg = NightmareV3Config()
    cfg.env.num_envs = num_envs
    cfg.env.episode_length_s = 1e6                      # no timeouts while playing
    env = NightmareV3Env(cfg, device=device, seed=seed)
    env.reset()
    nk.config.ENGINE_FPS = 1.0 / env.dt                 # custom_play.py:51
    eng = nk.EngineNode(num_envs, device=device)
    lin_t = torch.full((num_envs,), float(lin), device=device, dtype=torch.float64) if np.ndim(lin) == 0 else torch.as_tensor(lin, device=device)
    ang_t = torch.full((num_envs,), float(ang), device=device, dtype=torch.float64) if np.ndim(ang) == 0 else torch.as_tensor(ang, device=device)
    targets = torch.zeros(num_envs, 18, device=device)
    start = env.get_state()[0][:, :3].copy()
    falls = 0
    for i in range(int(seconds / env.dt)):
        nk.set_time_s(i * env.dt)
        goal = eng.update(lin_t, ang_t, "awake", "walk")
        targets += torch.clamp(goal - targets, -action_rate, action_rate)     # custom_play.py:73
        _, _, _, done, _ = env.step(env.actions_from_joint_targets(targets))
        falls += int(done.sum())
    qpos = env.get_state()[0]
    return dict(displacement=qpos[:, :3] - start, height=qpos[:, 2], falls=falls, fsm=eng.get_state()["fsm"])


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
    r = play(n, secs)
    d = r["displacement"]
    print(f"{n} envs, {secs:.1f} s: mean displacement x {d[:, 0].mean():+.3f} y {d[:, 1].mean():+.3f} m, mean base height {r['height'].mean():.3f} m, "
          f"terminations {r['falls']}, gait state {nk.FSM_NAMES[int(r['fsm'][0])]}")
