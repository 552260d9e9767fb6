"""What the two synchronisations around a SHORT timed region cost (the driver runs bench.py with --steps 20 --warmup 5):
torch.cuda.synchronize() alone vs. spinning on an event first, for 20 / 100 / 1000 steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(64, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
env = NightmareV3Env(cfg, seed=0); env.reset()
ret = torch.zeros(N, device="cuda")


def spin():
    e = torch.cuda.Event(); e.record()
    while not e.query(): pass
    torch.cuda.synchronize()


def run(k, sync, accumulate=True):
    sync()
    t0 = time.perf_counter()
    for i in range(k):
        r = env.step(acts[i % 64])[2]
        if accumulate: ret.add_(r)
    sync()
    return (time.perf_counter() - t0) / k * 1e6


for i in range(5): env.step(acts[i])
for k in (20, 100, 1000):
    for name, s in (("synchronize", torch.cuda.synchronize), ("event spin", spin)):
        for acc in (True, False):
            v = sorted(run(k, s, acc) for _ in range(9))
            print(f"steps {k:5d} {name:12s} returns+=rew {acc!s:5s}: us/step min {v[0]:.2f} median {v[4]:.2f} max {v[-1]:.2f}", flush=True)
# host time of one step() call (no GPU wait)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): env.step(acts[i])
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host time per step() call: {(t1 - t0) / 20 * 1e6:.1f} us")
