"""Start / end clock of every wave of one step-kernel launch (debug buffer): dispatch ramp, lifetime distribution, and what the
slowest waves have in common."""
import os, sys, numpy as np, torch
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
fb_prev = dbg.cpu().numpy()[:, 158].copy()
for rep in range(3):
    env.step(acts[rep]); torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)[0::2]
    t0 = d[:, 250] + d[:, 251] * 2**24
    t1 = d[:, 252] + d[:, 253] * 2**24
    base = t0.min()
    life = t1 - t0
    print(f"launch {rep}: start spread {np.percentile(t0 - base, [50, 90, 100]).round(0)}  end {np.percentile(t1 - base, [10, 50, 90, 99, 100]).round(0)}  "
          f"lifetime p10/50/90/99/max {np.percentile(life, [10, 50, 90, 99, 100]).round(0)} ticks")
    ncon = dbg.cpu().numpy()[:, 160].reshape(-1, 2)
    slow = np.argsort(-life)[:20]
    print("   slowest waves: lifetime", life[slow].round(0).tolist())
    print("   their ncon (env0, env1):", ncon[slow].astype(int).tolist())
    print("   corr(lifetime, ncon sum) =", np.corrcoef(life, ncon.sum(1))[0, 1].round(3))
    raw = dbg.cpu().numpy()
    fb = (raw[:, 158] - fb_prev).reshape(-1, 2).sum(1); fb_prev = raw[:, 158].copy()
    hop = raw[:, 159].reshape(-1, 2).sum(1)
    pgs = raw[:, 163].reshape(-1, 2).sum(1); nos = raw[:, 164].reshape(-1, 2).sum(1); pair = raw[:, 157].reshape(-1, 2).sum(1)
    X = np.c_[np.ones(len(life)), ncon.sum(1), fb, hop, nos, pair]
    coef, *_ = np.linalg.lstsq(X, life, rcond=None)
    res = life - X @ coef
    print("   lifetime ~ %.0f + %.0f*ncon(last substep, both envs) + %.0f*fallbacks + %.0f*hops + %.0f*noslip_iters + %.0f*pairflag ; residual std %.0f" % (*coef, res.std()))
    print("   means: ncon %.2f fallbacks %.2f hops %.2f noslip iters %.2f pair %.3f" % (ncon.sum(1).mean(), fb.mean(), hop.mean(), nos.mean(), pair.mean()))
    print("   slowest 12 waves: (lifetime, ncon e0+e1, hops, fallbacks, residual):",
          [(int(life[i]), int(ncon[i].sum()), int(hop[i]), int(fb[i]), int(res[i])) for i in slow[:12]])
    print("   hops p50/p90/p99/max %s ; ncon sum p50/p90/p99/max %s" % (np.percentile(hop, [50, 90, 99, 100]).tolist(), np.percentile(ncon.sum(1), [50, 90, 99, 100]).tolist()))
