"""Which workgroups of a step-kernel launch share a SIMD? (HW_ID from the debug buffer.) Prints, per XCD, the dispatch order j of a
block within its XCD against (SE, CU, SIMD, wave slot), the partner's j, and whether the pattern repeats from launch to launch."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
env = NightmareV3Env(cfg, seed=0); env.reset()
dbg = torch.zeros(N, 256, device="cuda")
for i in range(50): env.step(acts[i % 16])
env.set_debug_buffer(dbg)
prev = None
nw = N // 2; nwx = nw // 8
for rep in range(4):
    if rep == 2: torch.cuda.synchronize(); import time; time.sleep(0.01)      # an idle gap before the launch
    if rep == 3: x = torch.zeros(1 << 24, device="cuda"); x += 1                 # another kernel right before the launch
    env.step(acts[rep]); torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)[0::2]
    hw = d[:, 254].astype(np.int64); xcc = d[:, 255].astype(np.int64)
    wave = np.arange(nw)
    blk = (wave % nwx) * 8 + wave // nwx            # inverse of the kernel's XCD-aware mapping
    j = blk >> 3
    print(f"launch {rep}: block % 8 == XCC_ID for {np.mean((blk & 7) == xcc) * 100:.1f} % of the blocks")
    key = (((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)) * 4 + ((hw >> 4) & 3)
    order = {}
    for w in range(nw): order.setdefault(int(key[w]), []).append(w)
    partner_j = np.full(nw, -1)
    for g in order.values():
        if len(g) == 2: partner_j[g[0]], partner_j[g[1]] = j[g[1]], j[g[0]]
    diff = np.abs(partner_j - j)
    vals, cnts = np.unique(diff, return_counts=True)
    print("   |j - partner's j| histogram:", dict(zip(vals.tolist()[:12], cnts.tolist()[:12])), "..." if len(vals) > 12 else "")
    same = None if prev is None else float(np.mean(prev == key))
    print("   same SIMD as in the previous launch:", same)
    prev = key
    if rep == 0:
        m = xcc == 0
        o = np.argsort(j[m])
        print("   XCC 0, in dispatch order j: (se, cu, simd, slot)")
        rows = [(int((hw[m][i] >> 13) & 7), int((hw[m][i] >> 8) & 15), int((hw[m][i] >> 4) & 3), int(hw[m][i] & 15)) for i in o]
        for a in range(0, 256, 16): print("    j %3d.." % a, rows[a:a + 16])
