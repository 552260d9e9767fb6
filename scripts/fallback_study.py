"""Which meshes send the hull search to the exhaustive scan, and in which regime (measurement build: debug slot 159 = hops + 1000 x base-mesh
fallbacks + 100000 x tibia fallbacks of the step). usage: python scripts/fallback_study.py [iters=150]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.helpers import class_to_dict
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config, NightmareV3ConfigPPO
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
from nightmare_rl_amd.rl import OnPolicyRunner
from nightmare_rl_amd import _lib
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
N = 4096

def study(env, act_fn, label, steps=60):
    dbg = torch.zeros(N, 256, device="cuda"); env.set_debug_buffer(dbg)
    obs = env.get_observations(); tot = np.zeros(3); z = []; ncon = []
    with torch.no_grad():
        for i in range(steps):
            obs = env.step(act_fn(obs, i))[0]
            if i >= 10:
                c = dbg[:, 159].cpu().numpy().astype(np.int64)
                tot += [(c % 1000).sum(), ((c // 1000) % 100).sum(), (c // 100000).sum()]
                z.append(env.get_state()[0][:, 2].mean().item()); ncon.append(dbg[:, 160].mean().item())
    env.set_debug_buffer(None)
    n = (steps - 10) * N
    print(f"{label}: per env-step hops {tot[0] / n:.2f}  base-mesh fallbacks {tot[1] / n:.3f}  tibia fallbacks {tot[2] / n:.3f} (of 4 + 24 hull searches)  "
          f"mean base height {np.mean(z):.3f} m  contacts {np.mean(ncon):.2f}")

cfg = NightmareV3Config(); cfg.env.num_envs = N
g = torch.Generator().manual_seed(0)
acts = (torch.rand(16, N, 18, generator=g) * 2 - 1).cuda()
env = NightmareV3Env(cfg, seed=0, lib=_lib.load_measure()); env.reset()
study(env, lambda o, i: acts[i % 16], "random actions      ", 300)
for i in range(1500): env.step(acts[i % 16] * 0.12)
study(env, lambda o, i: acts[i % 16] * 0.12, "0.12 x random       ")
env.close()
torch.manual_seed(0)
env = NightmareV3Env(cfg, seed=0, lib=_lib.load_measure())
runner = OnPolicyRunner(env, class_to_dict(NightmareV3ConfigPPO()), log_dir=None, device="cuda:0")
runner.learn(iters, init_at_random_ep_len=True)
pol = runner.get_inference_policy(device="cuda:0")
study(env, lambda o, i: pol(o), f"policy after {iters} its")
