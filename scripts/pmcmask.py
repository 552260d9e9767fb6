"""Workload for per-stage dynamic instruction counts (rocprofv3 --pmc SQ_INSTS_*): settle 300 steps with the full kernel of the
measurement build, then 10 single steps with the ablation mask of argv[1], each from the same settled state. The LAST 10 step-kernel
dispatches of the trace are the masked ones."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
from nightmare_rl_amd import _lib
mask = int(sys.argv[1]); N = 4096
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0      # 0.12 = the standing regime (bench.py contact_regime)
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = ((torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1) * scale).cuda()
env = NightmareV3Env(cfg, seed=0, lib=_lib.load_measure()); env.reset()
for i in range(300 if scale == 1.0 else 1500): env.step(acts[i % 16])
q0 = env.get_state(); b0 = env.get_buffers()
env._L.nm_set_ablation(env._h, mask)
for i in range(10):
    env.set_state(*q0); env.set_buffers(**b0)
    env.step(acts[3])
torch.cuda.synchronize()
