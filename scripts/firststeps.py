"""Kernel time of the first launches of a fresh process (event pair per launch): what a short timed region after 5 warm-up steps sees."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(64, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
env = NightmareV3Env(cfg, seed=0); env.reset()
torch.cuda.synchronize()
ts = []
for i in range(60):
    env.profile(True)
    env.step(acts[i % 64])
    ms, n = env.profile(False)
    ts.append(ms * 1e3)
print("kernel us, launches 0..59:", " ".join(f"{t:.0f}" for t in ts))
# the same 25-step pattern as the driver's bench, timed on the host, repeated: does a later repetition run faster?
def spin():
    e = torch.cuda.Event(); e.record()
    while not e.query(): pass
    torch.cuda.synchronize()
for rep in range(4):
    for i in range(5): env.step(acts[i])
    spin(); t0 = time.perf_counter()
    for i in range(20): env.step(acts[5 + i])
    spin(); print(f"rep {rep}: 20 steps, {1e6 * (time.perf_counter() - t0) / 20:.1f} us/step")
    time.sleep(0.5)
