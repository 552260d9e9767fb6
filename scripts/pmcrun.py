"""Short fixed workload for rocprofv3 --pmc passes: settle 300 steps, then 60 profiled-by-counter steps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0      # 0.12: the standing regime (bench.py contact_regime)
acts = ((torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1) * scale).cuda()
env = NightmareV3Env(cfg, seed=0); env.reset()
for i in range(360): env.step(acts[i % 16])
torch.cuda.synchronize()
