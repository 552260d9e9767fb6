"""Attribute step-kernel time to stages by skipping them (measurement only). Usage on a GPU box: python scripts/ablate.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
from nightmare_rl_amd import _lib
MEASURE = _lib.load_measure()       # the -DNM_MEASURE build: the shipped library has no stage-skipping switches

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
names = {0: "full", 1: "no collision (=> no contacts)", 2: "no solver sweeps", 4: "no constraint stage (collision kept)", 5: "no collision+constraint",
         8: "no smooth stage", 13: "nothing but load/integrate/epilogue"}
for mask, name in names.items():
    env = NightmareV3Env(cfg, seed=0, lib=MEASURE)
    env.reset()
    for i in range(150):                      # settle on the ground with the real kernel first
        env.step(acts[i % 16])
    env._L.nm_set_ablation(env._h, mask)
    for i in range(20):
        env.step(acts[i % 16])
    env.profile(True)
    for i in range(100):
        env.step(acts[i % 16])
    ms, n = env.profile(False)
    print(f"mask {mask:2d} {name:45s} kernel avg {ms / n * 1e3:8.1f} us")
    env.close()
