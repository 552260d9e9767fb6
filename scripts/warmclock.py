"""Does a 20-step timed region (reset, 5 warm-up steps, 20 timed) run faster right after sustained GPU load? (clock ramp)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(64, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
def spin():
    e = torch.cuda.Event(); e.record()
    while not e.query(): pass
    torch.cuda.synchronize()
def region(tag):
    env = NightmareV3Env(cfg, seed=0); env.reset()
    for i in range(5): env.step(acts[i])
    spin(); t0 = time.perf_counter()
    for i in range(20): env.step(acts[5 + i])
    spin(); dt = time.perf_counter() - t0
    print(f"{tag}: {1e6 * dt / 20:.1f} us/step", flush=True)
    env.close()
region("cold process")
time.sleep(1.0)
region("after 1 s idle")
busy = NightmareV3Env(cfg, seed=1); busy.reset()
for rep in range(3):
    for i in range(4000): busy.step(acts[i % 64])       # ~0.25 s of sustained load
    region("right after 4000 back-to-back steps")
