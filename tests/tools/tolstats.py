"""Teacher-forced single-step error of the fp32 kernels against the fp64 oracle over many env-steps (diagnostic)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
from oracle import oracle as orc
N, T = 512, 150
cfg = NightmareV3Config(); cfg.env.num_envs = N
env = NightmareV3Env(cfg, device="cuda:0", seed=5)
ora = orc.OracleEnv(N, seed=5, num_threads=16)
env.reset(); ora.reset()
rng = np.random.default_rng(1)
errs = []
for t in range(T):
    a = rng.uniform(-1, 1, (N, 18)).astype(np.float32)
    obs, _, rew, done, _ = env.step(torch.from_numpy(a))
    oobs, orew, odone, _ = ora.step(a)
    errs.append(np.maximum(np.abs(obs.cpu().numpy() - oobs).max(axis=1), np.abs(rew.cpu().numpy() - orew)))
    qpos, qvel, qw = ora.get_state()
    env.set_state(qpos, qvel, qw)
    b = ora.get_buffers()
    env.set_buffers(dof_pos=b["dof_pos"], dof_vel=b["dof_vel"], actions=b["actions"], commands=b["commands"])
e = np.concatenate(errs)
print(f"{len(e)} env-steps: median {np.median(e):.2e} p99 {np.percentile(e,99):.2e} p99.9 {np.percentile(e,99.9):.2e} max {e.max():.2e}; "
      f">1e-4: {int((e>1e-4).sum())} ({100.0*(e>1e-4).mean():.3f} %), >1e-3: {int((e>1e-3).sum())}")
per_t = np.array([(x > 1e-4).sum() for x in errs]); print("outliers by step (first 20):", per_t[:20].tolist(), "rest:", int(per_t[20:].sum()))
