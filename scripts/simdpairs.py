"""Where the waves of one step-kernel launch ran (HW_ID / XCC_ID from the debug buffer) and how the two waves that share a SIMD
shape each other's lifetime: waves per SIMD / CU / XCD, start ramp, finish time per SIMD and per XCD, partner correlation."""
import os, sys, collections, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
env = NightmareV3Env(cfg, seed=0); env.reset()
dbg = torch.zeros(N, 256, device="cuda")
for i in range(300): env.step(acts[i % 16])
env.set_debug_buffer(dbg)
env.step(acts[15]); torch.cuda.synchronize()
for rep in range(3):
    env.step(acts[rep]); torch.cuda.synchronize()
    raw = dbg.cpu().numpy().astype(np.float64)
    d = raw[0::2]
    t0 = d[:, 250] + d[:, 251] * 2**24
    t1 = d[:, 252] + d[:, 253] * 2**24
    hw = d[:, 254].astype(np.int64); xcc = d[:, 255].astype(np.int64)
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    base = t0.min()
    life = t1 - t0
    end = t1 - base
    print(f"launch {rep}: waves {len(d)}  start p50/p90/p99/max {np.percentile(t0 - base, [50, 90, 99, 100]).round(0)}  "
          f"end p10/50/90/99/max {np.percentile(end, [10, 50, 90, 99, 100]).round(0)}  life p50/p99/max {np.percentile(life, [50, 99, 100]).round(0)}")
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    skey = key * 4 + simd
    cnt = collections.Counter(skey.tolist())
    print("   distinct XCC", len(set(xcc.tolist())), "CUs", len(set(key.tolist())), "SIMDs", len(cnt), " waves/SIMD histogram", sorted(collections.Counter(cnt.values()).items()))
    ccnt = collections.Counter(key.tolist())
    print("   waves/CU histogram", sorted(collections.Counter(ccnt.values()).items()))
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print(f"   XCC {x}: waves {m.sum():4d} CUs {len(set(key[m].tolist())):3d} start p50 {np.median(t0[m] - base):8.0f} max {np.max(t0[m] - base):8.0f}  end p50 {np.median(end[m]):8.0f} max {np.max(end[m]):8.0f}  life p50 {np.median(life[m]):8.0f}")
    # partner effect: per SIMD, sort the waves
    groups = collections.defaultdict(list)
    for i, k in enumerate(skey.tolist()): groups[k].append(i)
    pairs = np.array([g for g in groups.values() if len(g) == 2])
    if len(pairs):
        la, lb = life[pairs[:, 0]], life[pairs[:, 1]]
        print("   SIMDs with exactly 2 waves:", len(pairs), " corr(life a, life b) = %.3f" % np.corrcoef(la, lb)[0, 1])
        fin = np.maximum(end[pairs[:, 0]], end[pairs[:, 1]])
        first = np.minimum(end[pairs[:, 0]], end[pairs[:, 1]])
        print("   SIMD finish p10/50/90/99/max", np.percentile(fin, [10, 50, 90, 99, 100]).round(0), " first-wave end p50/max", np.percentile(first, [50, 100]).round(0),
              " mean gap", (fin - first).mean().round(0))
        ncon = raw[:, 160].reshape(-1, 2).sum(1); hop = raw[:, 159].reshape(-1, 2).sum(1)
        w = ncon * 1700 + hop * 650
        ws = w[pairs[:, 0]] + w[pairs[:, 1]]
        print("   corr(SIMD finish, sum of load proxy) = %.3f ; corr(wave life, own proxy) = %.3f ; corr(wave life, partner proxy) = %.3f" % (
            np.corrcoef(fin, ws)[0, 1], np.corrcoef(np.r_[la, lb], np.r_[w[pairs[:, 0]], w[pairs[:, 1]]])[0, 1],
            np.corrcoef(np.r_[la, lb], np.r_[w[pairs[:, 1]], w[pairs[:, 0]]])[0, 1]))
    # CU level: finish time of a CU vs its 8 waves' summed proxy
    cg = collections.defaultdict(list)
    for i, k in enumerate(key.tolist()): cg[k].append(i)
    cfin = np.array([end[g].max() for g in cg.values()]); cmean = np.array([life[g].mean() for g in cg.values()])
    print("   CU finish p50/p99/max", np.percentile(cfin, [50, 99, 100]).round(0), " CU mean life p50/max", np.percentile(cmean, [50, 100]).round(0))
    slow = np.argsort(-end)[:12]
    print("   last 12 waves to end: (end, life, start, xcc, se, cu, simd, ncon, hops):",
          [(int(end[i]), int(life[i]), int(t0[i] - base), int(xcc[i]), int(se[i]), int(cu[i]), int(simd[i]), int(raw[2 * i:2 * i + 2, 160].sum()), int(raw[2 * i:2 * i + 2, 159].sum())) for i in slow])
