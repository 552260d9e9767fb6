"""Short fixed workload for rocprofv3 --pmc passes over the one-launch rollout kernel (k_env_rollout): 4096 envs x 80 steps, reference nets,
2 warm-up rollouts + 4 counted ones.   python scripts/pmcrollout.py [N] [T]"""
import os
import sys

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
for _ in range(6):
    col.rollout(env, st, T, 0.99, cur_ret, cur_len, fin, ep=(ep_idx, ep_acc))
torch.cuda.synchronize()
