import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
from nightmare_rl_amd import _lib
MEASURE = _lib.load_measure()       # the -DNM_MEASURE build: the shipped library has no stage-skipping switches
N = 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
for mask, what in ((13, "load+D+E"), (13 + 128, "load(all)+D, no E math/stores"), (13 + 256, "no obs"), (0, "full"), (128, "full, no E"), (256, "full, no obs")):
    env = NightmareV3Env(cfg, seed=0, lib=MEASURE); env.reset()
    for i in range(100): env.step(acts[i % 16])
    q0 = env.get_state()
    env._L.nm_set_ablation(env._h, mask)
    res = []
    for rep in range(3):
        env.set_state(*q0)
        env.profile(True)
        for i in range(60): env.step(acts[i % 16])
        ms, n = env.profile(False)
        res.append(ms / n * 1e3)
    print(f"mask {mask:4d} {what:34s}: kernel {min(res):.1f} us")
    env.close()
