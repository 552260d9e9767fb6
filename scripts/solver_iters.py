"""Solver sweeps per forward pass (debug slots 163 / 164: PGS and NoSlip iterations of the step's last forward pass) in the three regimes.
usage: python scripts/solver_iters.py [train iters=150]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.helpers import class_to_dict
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config, NightmareV3ConfigPPO
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
from nightmare_rl_amd.rl import OnPolicyRunner
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
N = 4096

def study(env, act_fn, label, steps=60):
    dbg = torch.zeros(N, 256, device="cuda"); env.set_debug_buffer(dbg)
    obs = env.get_observations(); hp = np.zeros(64, int); hn = np.zeros(64, int); nc = []
    with torch.no_grad():
        for i in range(steps):
            obs = env.step(act_fn(obs, i))[0]
            if i >= 10:
                d = dbg.cpu().numpy()
                hp += np.bincount(d[:, 163].astype(int), minlength=64)[:64]; hn += np.bincount(d[:, 164].astype(int), minlength=64)[:64]; nc.append(d[:, 160].mean())
    env.set_debug_buffer(None)
    f = lambda h: {k: round(v / h.sum(), 3) for k, v in enumerate(h) if v / h.sum() > 0.002}
    print(f"{label}: contacts {np.mean(nc):.2f}  PGS sweeps mean {(hp * np.arange(64)).sum() / hp.sum():.2f} {f(hp)}  NoSlip sweeps mean {(hn * np.arange(64)).sum() / hn.sum():.2f} {f(hn)}")

cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
env = NightmareV3Env(cfg, seed=0); env.reset()
for i in range(300): env.step(acts[i % 16])
study(env, lambda o, i: acts[i % 16], "random actions      ")
for i in range(1500): env.step(acts[i % 16] * 0.12)
study(env, lambda o, i: acts[i % 16] * 0.12, "0.12 x random       ")
env.close()
torch.manual_seed(0)
env = NightmareV3Env(cfg, seed=0)
runner = OnPolicyRunner(env, class_to_dict(NightmareV3ConfigPPO()), log_dir=None, device="cuda:0")
runner.learn(iters, init_at_random_ep_len=True)
pol = runner.get_inference_policy(device="cuda:0")
study(env, lambda o, i: pol(o), f"policy after {iters} its")
