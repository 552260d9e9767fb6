"""BASELINE config 5's 'return curve vs CPU ref': the SAME on-policy runner (nightmare_rl_amd.rl, the reference's hyper-parameters,
envs/nightmare_v3_config.py:102-146) trained on (a) the HIP env and (b) the CPU oracle behind the same env surface, same number of
envs, several seeds each. Test infrastructure: the oracle never enters the product; this script is run by hand on the GPU box.

  python tests/tools/curve_vs_cpu.py [--envs 256] [--iters 150] [--seeds 3] [--out profiles/r02_curve_vs_cpu.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nightmare_rl_amd.envs.helpers import class_to_dict  # noqa: E402
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config, NightmareV3ConfigPPO  # noqa: E402
from nightmare_rl_amd.rl import OnPolicyRunner  # noqa: E402


class OracleAsEnv:
    """The CPU oracle (fp64 restatement of the reference env) behind the reference's env surface, tensors on the host."""

    def __init__(self, num_envs, seed, threads):
        from oracle import oracle as orc
        self.o = orc.OracleEnv(num_envs, seed=seed, num_threads=threads)
        self.orc = orc
        self.cfg = NightmareV3Config()
        self.num_envs, self.num_obs, self.num_privileged_obs, self.num_actions = num_envs, 66, 66, 18
        self.max_episode_length = 1250.0
        self.episode_length_buf = torch.zeros(num_envs, dtype=torch.int64)
        self._last_ep = self.episode_length_buf.clone()
        self.extras = {}
        self.obs_buf = torch.zeros(num_envs, 66)
        self.names = [n for n in orc.REW_NAMES if n in orc.DEFAULT_SCALES]

    def reset(self):
        self.o.reset_idx()
        return self.step(torch.zeros(self.num_envs, 18))[0], None

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return None

    def step(self, actions):
        if not torch.equal(self.episode_length_buf, self._last_ep):        # the runner overwrote it (init_at_random_ep_len)
            self.o.set_buffers(ep_len=self.episode_length_buf.numpy().astype(np.int64))
        obs, rew, done, to = self.o.step(actions.detach().cpu().numpy().astype(np.float32))
        self.episode_length_buf = torch.from_numpy(self.o.get_buffers()["ep_len"].copy())
        self._last_ep = self.episode_length_buf.clone()
        self.obs_buf = torch.from_numpy(obs)
        if done.any():
            n, stats = self.o.episode_stats()
            self.extras["episode"] = {"rew_" + k: torch.tensor(stats[self.orc.REW_NAMES.index(k)], dtype=torch.float32) for k in self.names}
            self.extras["time_outs"] = torch.from_numpy(to)
        return self.obs_buf, None, torch.from_numpy(rew), torch.from_numpy(done), self.extras


def train(kind, envs, iters, seed, threads):
    torch.manual_seed(seed)
    tc = class_to_dict(NightmareV3ConfigPPO())
    tc["runner"]["save_interval"] = 10 ** 9
    if kind == "hip":
        from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
        cfg = NightmareV3Config()
        cfg.env.num_envs = envs
        env, dev = NightmareV3Env(cfg, device="cuda:0", seed=seed), "cuda:0"
    else:
        env, dev = OracleAsEnv(envs, seed, threads), "cpu"
    r = OnPolicyRunner(env, tc, log_dir=None, device=dev)
    t0 = time.time()
    for done in range(0, iters, 10):          # in chunks, so that a long run keeps reporting
        r.learn(min(10, iters - done), init_at_random_ep_len=done == 0)
        print(f"   {kind} seed {seed}: iteration {done + 10}, mean step reward {r.history[-1]['mean_step_reward']:.4f}, {time.time() - t0:.0f} s", flush=True)
    h = r.history
    return dict(kind=kind, seed=seed, seconds=time.time() - t0, mean_step_reward=[x["mean_step_reward"] for x in h],
                mean_reward=[x["mean_reward"] for x in h], mean_episode_length=[x["mean_episode_length"] for x in h])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--iters", type=int, default=150)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--kinds", default="hip,cpu", help="which sides to train here; the CPU side needs no GPU and can run elsewhere")
    ap.add_argument("--merge", default=None, help="JSON of an earlier run (the other side) to merge into the table")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_curve_vs_cpu.json"))
    a = ap.parse_args()
    runs = json.load(open(a.merge))["runs"] if a.merge else []
    torch.set_num_threads(a.threads)
    for kind in a.kinds.split(","):
        for s in range(a.seeds):
            runs.append(train(kind, a.envs, a.iters, 100 + s, a.threads))
            print(kind, "seed", 100 + s, "%.0f s" % runs[-1]["seconds"], "final mean step reward %.4f" % np.mean(runs[-1]["mean_step_reward"][-10:]), flush=True)
    json.dump(dict(envs=a.envs, iters=a.iters, seeds=a.seeds, table=[], runs=runs), open(a.out, "w"))
    marks = [m for m in (10, 25, 50, 75, 100, 150, 200, 300) if m <= a.iters]
    table = []
    for m in marks:
        row = {"iteration": m}
        txt = "it %4d  mean reward per env-step:" % m
        for kind in ("hip", "cpu"):
            v = np.array([np.mean(r["mean_step_reward"][max(0, m - 5):m]) for r in runs if r["kind"] == kind])
            if len(v):
                row[kind + "_mean"], row[kind + "_min"], row[kind + "_max"], row[kind + "_seeds"] = float(v.mean()), float(v.min()), float(v.max()), len(v)
                txt += "   %s %7.4f [%7.4f, %7.4f]" % ("HIP env" if kind == "hip" else "CPU oracle env", v.mean(), v.min(), v.max())
        table.append(row)
        print(txt)
    json.dump(dict(envs=a.envs, iters=a.iters, seeds=a.seeds, table=table, runs=runs), open(a.out, "w"))


if __name__ == "__main__":
    main()
