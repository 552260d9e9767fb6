"""Kernel-time microbench of the step kernel (HIP events), for A/B-ing builds on a GPU box.
usage: [NM_HIP_LIB=path/to/variant.so] python scripts/quickbench.py [envs=4096] [ablation mask=0] [action scale=1.0]
action scale 0.12 = the standing regime of bench.py's `contact_regime` (about 5.3 floor contacts per env instead of 1.4)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
mask = int(sys.argv[2]) if len(sys.argv) > 2 else 0
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
acts = acts * scale
from nightmare_rl_amd import _lib
env = NightmareV3Env(cfg, seed=0, lib=_lib.load_measure() if mask else None); env.reset()   # stage skipping needs the -DNM_MEASURE build
for i in range(300): env.step(acts[i % 16])          # settle with the real kernel
q0 = env.get_state()
if mask: env._L.nm_set_ablation(env._h, mask)
res = []
for rep in range(3):
    if mask: env.set_state(*q0)                       # ablated physics drifts: restart every rep from the settled state
    env.profile(True)
    for i in range(40 if mask else 200): env.step(acts[i % 16])
    ms, n = env.profile(False)
    res.append(ms / n * 1e3)
print(f"lib={os.path.basename(_lib.LIB_PATH)} action scale={scale} mask={mask} N={N} step kernel avg us: " + " ".join(f"{r:.1f}" for r in res) + f"  -> {N / min(res):.2f} M env-steps/s (kernel only)")

print(env.counters(), "per geom-test fallback rate", env.counters()["hull_search_fallbacks"] / (env.common_step_counter * N * 2 * 7))
