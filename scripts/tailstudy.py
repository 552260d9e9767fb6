"""Why is the slowest wave of a step-kernel launch 1.29x the median? (a) is a wave's lifetime a function of its data (two identical env
objects, same step: correlation of per-wave lifetimes) or of where / next to whom it ran; (b) lifetime against own and SIMD-partner
features (contacts, two-env constraint pass taken, exactly-one-env-in-contact, hops, fallbacks, solver iterations)."""
import os, sys, collections, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
N = 4096
cfg = NightmareV3Config(); cfg.env.num_envs = N
acts = (torch.rand(16, N, 18, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
envs = [NightmareV3Env(cfg, seed=0) for _ in range(2)]
dbgs = [torch.zeros(N, 256, device="cuda") for _ in range(2)]
for e in envs: e.reset()
for i in range(300):
    for e in envs: e.step(acts[i % 16])
for e, d in zip(envs, dbgs): e.set_debug_buffer(d)
for e in envs: e.step(acts[15])
torch.cuda.synchronize()
fbp = [d.cpu().numpy()[:, 158].copy() for d in dbgs]


def feats(raw, fb_prev):
    d = raw[0::2]
    life = (d[:, 252] + d[:, 253] * 2**24) - (d[:, 250] + d[:, 251] * 2**24)
    hw = d[:, 254].astype(np.int64); xcc = d[:, 255].astype(np.int64)
    skey = ((((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)) * 4 + ((hw >> 4) & 3))
    nc = raw[:, 160].reshape(-1, 2)
    f = dict(ncon=nc.sum(1), nmax=nc.max(1), one=((nc.min(1) == 0) & (nc.max(1) > 0)).astype(float), ntog=raw[0::2, 156],
             hop=raw[:, 159].reshape(-1, 2).sum(1), fb=(raw[:, 158] - fb_prev).reshape(-1, 2).sum(1),
             pgs=raw[:, 163].reshape(-1, 2).sum(1), nos=raw[:, 164].reshape(-1, 2).sum(1))
    return life, skey, f


for rep in range(3):
    lifes = []
    for k, (e, d) in enumerate(zip(envs, dbgs)):
        e.step(acts[rep]); torch.cuda.synchronize()
        raw = d.cpu().numpy().astype(np.float64)
        life, skey, f = feats(raw, fbp[k]); fbp[k] = raw[:, 158].copy()
        lifes.append(life)
        if k == 0:
            groups = collections.defaultdict(list)
            for i, s in enumerate(skey.tolist()): groups[s].append(i)
            partner = np.zeros(len(life), dtype=np.int64)
            for g in groups.values():
                if len(g) == 2: partner[g[0]], partner[g[1]] = g[1], g[0]
            names = list(f)
            X = np.c_[np.ones(len(life))] + 0
            X = np.column_stack([np.ones(len(life))] + [f[n] for n in names] + [f[n][partner] for n in names])
            coef, *_ = np.linalg.lstsq(X, life, rcond=None)
            res = life - X @ coef
            print(f"step {rep}: life p50 {np.median(life):.0f} max {life.max():.0f}; fit on own + partner features: residual std {res.std():.0f} (total std {life.std():.0f})")
            print("   own    :", " ".join(f"{n} {c:+.0f}" for n, c in zip(names, coef[1:1 + len(names)])))
            print("   partner:", " ".join(f"{n} {c:+.0f}" for n, c in zip(names, coef[1 + len(names):])))
            print("   means  :", " ".join(f"{n} {f[n].mean():.2f}" for n in names))
            for nt in (0, 1, 2):
                m = f["ntog"] == nt
                print(f"   two-env pass ran in {nt} substeps: {m.sum():4d} waves, life mean {life[m].mean():.0f}, ncon mean {f['ncon'][m].mean():.2f}")
            for pat, m in (("both 0", f["ncon"] == 0), ("exactly one env in contact", f["one"] == 1), ("both in contact", (f["one"] == 0) & (f["ncon"] > 0))):
                print(f"   {pat:28s}: {m.sum():4d} waves, life mean {life[m].mean():.0f} p90 {np.percentile(life[m], 90) if m.sum() else 0:.0f}, ncon mean {f['ncon'][m].mean() if m.sum() else 0:.2f}")
            pair_max = np.maximum(life, life[partner])
            print(f"   per-SIMD max lifetime p50 {np.median(pair_max):.0f} p99 {np.percentile(pair_max, 99):.0f} max {pair_max.max():.0f}; per-SIMD sum of lifetimes p50 {np.median(life + life[partner]):.0f} max {(life + life[partner]).max():.0f}")
    print(f"   same data on two env objects: corr of per-wave lifetimes {np.corrcoef(lifes[0], lifes[1])[0, 1]:.3f}; |diff| p50 {np.median(np.abs(lifes[0] - lifes[1])):.0f} p99 {np.percentile(np.abs(lifes[0] - lifes[1]), 99):.0f}")
# the slowest waves of the last step taken on env object 0, with their SIMD partner
raw = dbgs[0].cpu().numpy().astype(np.float64)
life, skey, f = feats(raw, fbp[0] * 0)
first = raw[1::2, 156].astype(int)
order = np.argsort(-life)
print("slowest waves: life | first substep (n0,n1) | last substep (n0,n1) hops pgs nos || partner: life, first (n0,n1), last (n0,n1), hops")
nc = raw[:, 160].reshape(-1, 2).astype(int); hop = raw[:, 159].reshape(-1, 2).sum(1).astype(int)
for i in order[:25]:
    j = partner[i]
    print(f"  {life[i]:7.0f} | ({first[i] // 64},{first[i] % 64}) | ({nc[i, 0]},{nc[i, 1]}) {hop[i]:2d} {int(f['pgs'][i])} {int(f['nos'][i])} || {life[j]:7.0f} ({first[j] // 64},{first[j] % 64}) ({nc[j, 0]},{nc[j, 1]}) {hop[j]:2d}")
print("fastest:")
for i in order[-8:]:
    j = partner[i]
    print(f"  {life[i]:7.0f} | ({first[i] // 64},{first[i] % 64}) | ({nc[i, 0]},{nc[i, 1]}) {hop[i]:2d} {int(f['pgs'][i])} {int(f['nos'][i])} || {life[j]:7.0f} ({first[j] // 64},{first[j] % 64}) ({nc[j, 0]},{nc[j, 1]}) {hop[j]:2d}")
tot = (first // 64 + first % 64) + nc.sum(1)
print("corr(life, contacts over both substeps) = %.3f ; corr(life, own + partner contacts over both substeps) = %.3f" % (np.corrcoef(life, tot)[0, 1], np.corrcoef(life, tot + tot[partner])[0, 1]))
X = np.column_stack([np.ones(len(life)), tot, tot[partner], hop, hop[partner]])
coef, *_ = np.linalg.lstsq(X, life, rcond=None)
print("life ~ %.0f + %.0f*own contacts(2 substeps) + %.0f*partner contacts + %.0f*own hops + %.0f*partner hops; residual std %.0f" % (*coef, (life - X @ coef).std()))
