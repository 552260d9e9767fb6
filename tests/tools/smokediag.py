import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
from oracle import oracle as orc
N = 64
for dt in (torch.float32, torch.float64):
    cfg = NightmareV3Config(); cfg.env.num_envs = N
    env = NightmareV3Env(cfg, device="cuda:0", seed=3, dtype=dt)
    ora = orc.OracleEnv(N, seed=3)
    env.reset(); ora.reset()
    rng = np.random.default_rng(0)
    for t in range(30):
        a = rng.uniform(-1, 1, (N, 18)).astype(np.float32)
        obs, _, rew, done, _ = env.step(torch.from_numpy(a))
        oobs, orew, odone, _ = ora.step(a)
        err = np.abs(obs.cpu().numpy() - oobs).max(axis=1)
        if err.max() > 1e-4:
            i = int(err.argmax()); k = int(np.abs(obs.cpu().numpy()[i] - oobs[i]).argmax())
            d = ora.data(i)
            print(dt, "t", t, "max err", err.max(), "env", i, "obs slot", k, "n>1e-4:", int((err > 1e-4).sum()), "ncon", d.ncon, flush=True)
        qpos, qvel, qw = ora.get_state()
        env.set_state(qpos, qvel, qw)
        b = ora.get_buffers()
        env.set_buffers(dof_pos=b["dof_pos"], dof_vel=b["dof_vel"], actions=b["actions"], commands=b["commands"])
    print(dt, "done", env.counters())
