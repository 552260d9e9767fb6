import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
env = NightmareV3Env(cfg, seed=0); env.reset()
g = torch.Generator().manual_seed(0)
tot_done = 0
for scale in (1.0, 5.0, 20.0):
    for i in range(700):
        a = ((torch.rand(N, 18, generator=g) * 2 - 1) * scale).cuda()
        _, _, _, done, _ = env.step(a)
        tot_done += int(done.sum())
    print(scale, env.counters(), "resets", tot_done, flush=True)
