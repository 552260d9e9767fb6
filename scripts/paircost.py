"""Would pairing heavy envs with light ones shorten the step kernel? Measurement build (-DNM_ENVCOST: the debug buffer's hop slot
carries each env's collision+constraint cycles): per-env cost of consecutive steps, the slowest wave under the fixed pairing
(env 2w, 2w+1) against pairings sorted by the PREVIOUS step's costs."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
env = NightmareV3Env(cfg, seed=0); env.reset()
dbg = torch.zeros(N, 256, device="cuda")
for i in range(300): env.step(acts[i % 16])
env.set_debug_buffer(dbg)
costs, lifes = [], []
for i in range(12):
    env.step(acts[i % 16]); torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)
    costs.append(d[:, 159] * 16)
    w = d[0::2]
    lifes.append((w[:, 252] + w[:, 253] * 2**24) - (w[:, 250] + w[:, 251] * 2**24))
costs, lifes = np.array(costs), np.array(lifes)
print("per-env B+C cycles: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (costs.mean(), *np.percentile(costs, [50, 90, 99, 100])))
print("step-to-step correlation of an env's cost: %.3f" % np.mean([np.corrcoef(costs[t], costs[t + 1])[0, 1] for t in range(len(costs) - 1)]))
pair = costs[:, 0::2] + costs[:, 1::2]
base = (lifes - pair)
print("wave lifetime - (cost a + cost b): mean %.0f std %.0f; max lifetime %.0f, max pair cost %.0f (mean %.0f)" %
      (base.mean(), base.std(), lifes.max(1).mean(), pair.max(1).mean(), pair.mean()))
for lag in (1, 4, 8):
    gains = []
    for t in range(lag, len(costs)):
        order = np.argsort(costs[t - lag], kind="stable")
        a, b = order[: N // 2], order[::-1][: N // 2]          # lightest with heaviest by the older step's costs
        newpair = costs[t][a] + costs[t][b]
        gains.append((pair[t].max(), newpair.max(), np.percentile(newpair, 99)))
    g = np.array(gains)
    print(f"pairing by costs {lag} step(s) old: slowest pair now {g[:, 0].mean():.0f} -> {g[:, 1].mean():.0f} cycles (p99 {g[:, 2].mean():.0f})")
