"""How evenly do the waves of the one-launch rollout (nm_rollout) finish? Every wave keeps its two envs for all 80 steps and never waits
for another wave, so the launch ends with the slowest SIMD. Records each wave's start / end clock (s_memtime through the debug buffer)
and prints the distribution of the waves' total times.   python scripts/rolloutwaves.py [N] [T]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
from nightmare_rl_amd.rl import ActorCritic, RolloutStorage
from nightmare_rl_amd.rl.fused import FusedCollector, FusedUpdate

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = "cuda:0"
cfg = NightmareV3Config()
cfg.env.num_envs = N
env = NightmareV3Env(cfg, device=dev, seed=0)
env.reset()
env.episode_length_buf = torch.randint(0, 1250, (N,), device=dev, dtype=torch.int64)
torch.manual_seed(0)
ac = ActorCritic(66, 66, 18, actor_hidden_dims=[54, 42, 30], critic_hidden_dims=[54, 42, 30], activation="elu", init_noise_std=1.0).to(dev)
fu = FusedUpdate(ac, torch.optim.Adam(ac.parameters(), lr=1e-3), dev, lr=1e-3)
col = FusedCollector(ac, N, dev, seed=1, update=fu)
st = RolloutStorage(N, T, [66], [None], [18], dev)
z = lambda *s: torch.zeros(*s, device=dev)
cur_ret, cur_len, fin = z(N), z(N), z(3)
ep_idx = torch.tensor([env._stat_names.index(k[4:]) for k in sorted(env.extras["episode"])], dtype=torch.int32, device=dev)
ep_acc = z(ep_idx.numel())
for _ in range(4):
    col.rollout(env, st, T, 0.99, cur_ret, cur_len, fin, ep=(ep_idx, ep_acc))
dbg = torch.zeros(N, 256, device=dev)
env.set_debug_buffer(dbg)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
col.rollout(env, st, T, 0.99, cur_ret, cur_len, fin, ep=(ep_idx, ep_acc))
e1.record()
torch.cuda.synchronize()
env.set_debug_buffer(None)
clk = dbg.view(torch.int64).reshape(-1)[: N].cpu().numpy().astype(np.int64).reshape(-1, 2)   # [waves][start, end], waves = N / 2
clk = clk[: N // 2]
# Only differences INSIDE a wave are taken: s_memtime has no common base across the chip (round 4's report subtracted start clocks of
# different waves and printed a span of 3e11 ticks; grouping the waves by XCD still leaves start values tens of millions of ticks apart
# inside a group - more than five launches' worth). What the per-wave times do give: the tick rate (the slowest wave spans all but the
# launch ramp of the kernel, whose duration the event pair measures) and how evenly the waves finish.
life = (clk[:, 1] - clk[:, 0]).astype(np.float64)
ms = e0.elapsed_time(e1)
tick_ns = ms * 1e6 / life.max()
print(f"{N} envs x {T} steps: launches {ms:.3f} ms (event pair around nm_rollout: k_env_rollout + k_rollout_tail); slowest wave {life.max():.0f} ticks -> "
      f">= {1.0 / tick_ns:.2f} ticks per ns")
print(f"wave total time (ticks): mean {life.mean():.0f}  p50 {np.median(life):.0f}  p90 {np.percentile(life, 90):.0f}  p99 {np.percentile(life, 99):.0f}  max {life.max():.0f}"
      f"   max / mean {life.max() / life.mean():.3f}   per step: mean {life.mean() / T:.0f}  max {life.max() / T:.0f}")
print(f"mean wave time / slowest wave = {life.mean() / life.max():.3f}: the share of the launch the average wave slot is occupied (no wave waits for another inside the launch)")
