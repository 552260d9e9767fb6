"""Contacts per env (last forward pass of the step) in the bench workload: what the constraint stage's loops see."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
env = NightmareV3Env(cfg, seed=0); env.reset()
dbg = torch.zeros(N, 256, device="cuda")
for i in range(300): env.step(acts[i % 16])
env.set_debug_buffer(dbg)
h = np.zeros(48, int); pg = np.zeros(8, int); ns = np.zeros(8, int)
for i in range(50):
    env.step(acts[i % 16])
    d = dbg.cpu().numpy()
    h += np.bincount(d[:, 160].astype(int), minlength=48)[:48]
    pg += np.bincount(d[:, 163].astype(int), minlength=8)[:8]
    ns += np.bincount(d[:, 164].astype(int), minlength=8)[:8]
print("ncon histogram:", {k: int(v) for k, v in enumerate(h) if v})
print("mean", (h * np.arange(48)).sum() / h.sum(), "PGS iterations:", pg.tolist(), "NoSlip iterations:", ns.tolist())
