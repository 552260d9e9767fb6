"""Contacts per env under a TRAINED policy (what the constraint stage sees late in training, when the robot stands and walks), and the
step kernel's time in that regime: train `iters` PPO iterations, then step the env with the policy's mean actions.
usage: python scripts/nconhist_policy.py [iters=150]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.helpers import class_to_dict
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config, NightmareV3ConfigPPO
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
from nightmare_rl_amd.rl import OnPolicyRunner
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
N = 4096
torch.manual_seed(0)
cfg = NightmareV3Config(); cfg.env.num_envs = N
env = NightmareV3Env(cfg, seed=0)
runner = OnPolicyRunner(env, class_to_dict(NightmareV3ConfigPPO()), log_dir=None, device="cuda:0")
runner.learn(iters, init_at_random_ep_len=True)
print("trained", iters, "iterations: mean reward", runner.history[-1]["mean_reward"], "fps first / last", round(runner.history[5]["fps"] / 1e6, 1), round(runner.history[-1]["fps"] / 1e6, 1), "M")
pol = runner.get_inference_policy(device="cuda:0")
obs = env.get_observations()
dbg = torch.zeros(N, 256, device="cuda")
env.set_debug_buffer(dbg)
h = np.zeros(48, int)
with torch.no_grad():
    for i in range(100):
        obs, _, _, _, _ = env.step(pol(obs))
        if i >= 50:
            h += np.bincount(dbg[:, 160].cpu().numpy().astype(int), minlength=48)[:48]
env.set_debug_buffer(None)
print("ncon histogram (last forward pass of a step, trained policy):", {k: int(v) for k, v in enumerate(h) if v})
print("mean contacts per env", (h * np.arange(48)).sum() / h.sum(), " share with > 8:", h[9:].sum() / h.sum(), " > 16:", h[17:].sum() / h.sum())
acts = []
with torch.no_grad():
    for i in range(16):
        a = pol(obs); acts.append(a.clone()); obs, _, _, _, _ = env.step(a)
env.profile(True)
with torch.no_grad():
    for i in range(200): env.step(acts[i % 16])
ms, n = env.profile(False)
print(f"step kernel under the trained policy's actions: {ms / n * 1e3:.1f} us")
