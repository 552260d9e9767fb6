#!/usr/bin/env python
"""play.py - headless analogue of the reference's play.py (reference play.py:60-72 loads `model_<it>.pt['model_state_dict']` into
an ActorCritic, :118-132 runs `nn.act(obs)` -> scale/clip -> PD servo command -> mj_step in a viewer loop).

Here: the checkpoint's actor runs on the matrix cores (nm_policy_* handle), the env on the step kernel, any number of robots at
once, no viewer. Like upstream the actions are SAMPLED (`nn.act`, not `act_inference`) unless --deterministic.

  python scripts/play.py [checkpoint.pt | --log-root logs/nightmare_v3] [-e 64] [--steps 1300] [--decimation 2] [--cmd 0.3 0.0 0.2]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nightmare_rl_amd.envs.helpers import get_load_path  # noqa: E402
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config  # noqa: E402
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env  # noqa: E402
from nightmare_rl_amd.policy import ActorMLP  # noqa: E402


def actor_from_checkpoint(path, device):
    """ActorMLP with the weights of `actor.<2i>.weight/bias` (rsl_rl ActorCritic layout) + the learned action std."""
    sd = torch.load(path, map_location="cpu")["model_state_dict"]
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("actor.") and k.endswith(".weight")})
    dims = [sd[f"actor.{idx[0]}.weight"].shape[1]] + [sd[f"actor.{i}.weight"].shape[0] for i in idx]
    net = ActorMLP(dims)
    for layer, i in zip(net.layers, idx):
        layer.weight.data.copy_(sd[f"actor.{i}.weight"])
        layer.bias.data.copy_(sd[f"actor.{i}.bias"])
    net = net.to(device)
    net.mark_dirty()
    return net, sd["std"].to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint", nargs="?", default=None)
    ap.add_argument("--log-root", default="logs/nightmare_v3")
    ap.add_argument("-e", "--envs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=1300)
    ap.add_argument("--decimation", type=int, default=None, help="physics substeps per policy step (reference play.py:23 uses 4, the env 2)")
    ap.add_argument("--deterministic", action="store_true", help="act_inference (mean action) instead of upstream's sampled nn.act")
    ap.add_argument("--cmd", type=float, nargs=3, default=None, metavar=("VX", "VY", "YAW"), help="fixed velocity command (default: the env's own resampling)")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    path = a.checkpoint or get_load_path(a.log_root)
    dev = torch.device("cuda", 0)
    net, std = actor_from_checkpoint(path, dev)
    cfg = NightmareV3Config()
    cfg.env.num_envs = a.envs
    if a.decimation is not None:
        cfg.control.decimation = a.decimation
    env = NightmareV3Env(cfg, device=dev, seed=a.seed)
    torch.manual_seed(a.seed)
    obs, _ = env.reset()
    ret = torch.zeros(a.envs, device=dev)
    done_returns, ndone, track = [], 0, []
    for t in range(a.steps):
        if a.cmd is not None:
            env.set_buffers(commands=np.tile(np.array(a.cmd, np.float64), (a.envs, 1)))
        mean = net(obs)
        act = mean if a.deterministic else mean + std * torch.randn_like(mean)
        obs, _, rew, done, extras = env.step(act)
        ret += rew
        d = done > 0
        if bool(d.any()):
            done_returns += ret[d].tolist()
            ndone += int(d.sum())
            ret[d] = 0
        track.append(float(rew.mean()))
    print(f"checkpoint {path}: {a.envs} robots x {a.steps} steps, decimation {cfg.control.decimation}, "
          f"{'mean' if a.deterministic else 'sampled'} actions")
    print(f"  mean reward per step {np.mean(track):.4f} (last 200 steps {np.mean(track[-200:]):.4f}); episodes finished {ndone}"
          + (f", mean return {np.mean(done_returns):.2f}" if done_returns else ""))
    if "episode" in extras:
        print("  last episode statistics:", {k: round(float(v), 4) for k, v in extras["episode"].items()})
    print("  env counters:", env.counters())


if __name__ == "__main__":
    main()
