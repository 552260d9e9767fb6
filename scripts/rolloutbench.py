"""Collection of one PPO iteration (80 steps, reference nets 66->54->42->30->18|1) at N envs: the one-launch rollout (nm_rollout) against
the per-step paths. Prints microseconds per env step of the rollout and env-steps/s.   python scripts/rolloutbench.py [N] [T]"""
import os
import sys
import time

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
assert col.can_rollout(env)
for _ in range(3):
    col.rollout(env, st, T, 0.99, cur_ret, cur_len, fin, ep=(ep_idx, ep_acc))
torch.cuda.synchronize()
reps = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    col.rollout(env, st, T, 0.99, cur_ret, cur_len, fin, ep=(ep_idx, ep_acc))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"one-launch rollout: {N} envs x {T} steps in {ms:.3f} ms = {ms * 1e3 / T:.2f} us per step = {N * T / ms / 1e3:.2f} M env-steps/s (policy + step + record)")
# per-step launches of the same pieces: nm_rollout_act + nm_step (eager), and nm_step alone
it = col.iter_dev
for name, with_act in (("policy_act + step, eager launches", True), ("step alone (random actions resident), eager launches", False)):
    acts = torch.rand(N, 18, device=dev) * 2 - 1
    o = env.get_observations()
    for s in range(10):
        o = env.step(env.policy_act(fu.flat, o, 1, it, s, st) if with_act else acts)[0]
    torch.cuda.synchronize()
    e0.record()
    for rep in range(reps):
        for s in range(T):
            o = env.step(env.policy_act(fu.flat, o, 1, it, s, st) if with_act else acts)[0]
    e1.record()
    torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / reps
    print(f"{name}: {ms2 * 1e3 / T:.2f} us per step = {N * T / ms2 / 1e3:.2f} M env-steps/s")
print("counters", env.counters())
