"""Degree of the hull vertices that actually are support vertices in the bench workload (hull cache rows after settling)."""
import os, sys, numpy as np, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
env = NightmareV3Env(cfg, seed=0); env.reset()
m = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nightmare_rl_amd/model/nm_model.npz"))
deg = (m["hull_nbr"] >= 0).sum(1)
dbg = torch.zeros(N, 256, device="cuda")
env.set_debug_buffer(dbg)
hist = np.zeros(40, int)
for i in range(400):
    env.step(acts[i % 16])
    if i >= 300 and i % 10 == 0:
        d = dbg.cpu().numpy()
        for g in range(1, 7):
            v = d[:, 150 + g].astype(int) + int(m["col_vadr"][g])
            hist += np.bincount(deg[v], minlength=40)[:40]
print("degree histogram of tibia support vertices:", {k: int(v) for k, v in enumerate(hist) if v})
print("share with degree > 15: %.3f   > 31: %.4f" % (hist[16:].sum() / hist.sum(), hist[32:].sum() / hist.sum()))
