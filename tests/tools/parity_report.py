"""Diagnostic behind the fp32 parity tests (run on the GPU box): prints the numbers the tests' thresholds come from.
  A  teacher-forced single-step error of the fp32 kernel vs the fp64 oracle; every env-step above 1e-4 with its decision margin
  B  free-running 1024 envs x 1300 steps: episode returns and final joint positions, fp32 kernel vs fp64 oracle
  C  one env through a whole episode on the fp32 kernel, re-synchronised every K steps: error by steps-since-sync
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_tools as pt  # noqa: E402
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config  # noqa: E402
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env  # noqa: E402
from oracle import oracle as orc  # noqa: E402

TH = int(os.environ.get("NM_ORACLE_THREADS", "16"))


def make_env(N, seed, dtype=torch.float32):
    cfg = NightmareV3Config()
    cfg.env.num_envs = N
    return NightmareV3Env(cfg, device="cuda:0", seed=seed, dtype=dtype)


def sync(env, ora):
    env.set_state(*ora.get_state())
    b = ora.get_buffers()
    env.set_buffers(dof_pos=b["dof_pos"], dof_vel=b["dof_vel"], actions=b["actions"], commands=b["commands"], episode_sums=b["episode_sums"].T)
    env.episode_length_buf = torch.from_numpy(b["ep_len"]).cuda()


def part_a(N=512, T=150, scale=1.0, settle=0, label="A"):
    """scale / settle: the same with the robots STANDING (0.12 x the actions after `settle` oracle steps: 5.3 contacts per env, the regime of a
    trained policy, where the floor contacts leave the collision stage through the batched pass) - part D"""
    env, ora = make_env(N, 5), orc.OracleEnv(N, seed=5, num_threads=TH)
    env.reset(); ora.reset()
    rng = np.random.default_rng(1)
    for t in range(settle):
        ora.step((rng.uniform(-1, 1, (N, 18)) * scale).astype(np.float32))
    errs, outl, nclose, nchecked = [], [], 0, 0
    ncon = []
    for t in range(T):
        a = (rng.uniform(-1, 1, (N, 18)) * scale).astype(np.float32)
        q, v, w = ora.get_state()
        b = ora.get_buffers()
        sync(env, ora)
        obs, _, rew, done, _ = env.step(torch.from_numpy(a))
        oobs, orew, odone, _ = ora.step(a)
        e = np.maximum(np.abs(obs.cpu().numpy() - oobs).max(axis=1), np.abs(rew.cpu().numpy() - orew))
        errs.append(e)
        if t % 10 == 0: ncon.append(np.mean([ora.data(i).ncon for i in range(0, N, 8)]))
        for i in np.nonzero(e > 1e-4)[0]:
            m, k, npair = pt.discrete_margin(orc, q[i], v[i], w[i], pt.servo_ctrl(a[i], b["dof_pos"][i]))
            outl.append((t, int(i), float(e[i]), m, k, npair))
        if t % 10 == 0:   # base rate of small margins
            for i in range(0, N, 16):
                m, k, _ = pt.discrete_margin(orc, q[i], v[i], w[i], pt.servo_ctrl(a[i], b["dof_pos"][i]))
                nclose += m < 1e-6
                nchecked += 1
    e = np.concatenate(errs)
    print(f"{label}: {len(e)} env-steps (action scale {scale}, {settle} settling steps, {np.mean(ncon):.2f} contacts per env): median {np.median(e):.2e} p99 {np.percentile(e, 99):.2e} "
          f"p99.9 {np.percentile(e, 99.9):.2e} max {e.max():.2e}; >1e-4: {int((e > 1e-4).sum())}; base rate of margin<1e-6: {nclose}/{nchecked}")
    for o in outl:
        print("   outlier t=%d env=%d err=%.2e margin=%.2e kind=%s pairs=%d" % o)


def part_b(N=1024, T=1300):
    env, ora = make_env(N, 21), orc.OracleEnv(N, seed=21, num_threads=TH)
    env.reset(); ora.reset()
    rng = np.random.default_rng(7)
    ret32, ret64 = np.zeros(N), np.zeros(N)
    first32, first64 = np.full(N, -1), np.full(N, -1)
    epi32, epi64 = [], []
    run32, run64 = np.zeros(N), np.zeros(N)
    t0 = time.time()
    for t in range(T):
        a = rng.uniform(-1, 1, (N, 18)).astype(np.float32)
        obs, _, rew, done, _ = env.step(torch.from_numpy(a))
        oobs, orew, odone, _ = ora.step(a)
        r, d = rew.cpu().numpy().astype(np.float64), done.cpu().numpy()
        run32 += r; run64 += orew
        ret32 += r; ret64 += orew
        for i in np.nonzero(d)[0]:
            epi32.append(run32[i]); run32[i] = 0
            if first32[i] < 0: first32[i] = t
        for i in np.nonzero(odone)[0]:
            epi64.append(run64[i]); run64[i] = 0
            if first64[i] < 0: first64[i] = t
    q32 = env.get_state()[0][:, 7:]
    q64 = ora.get_state()[0][:, 7:]
    print(f"B: {N} envs x {T} steps in {time.time() - t0:.0f}s")
    for name, x, y in (("sum of rewards over the horizon", ret32, ret64), ("completed-episode return", np.array(epi32), np.array(epi64))):
        se = np.sqrt(x.var(ddof=1) / len(x) + y.var(ddof=1) / len(y))
        print(f"   {name}: fp32 mean {x.mean():.4f} std {x.std():.4f} n {len(x)} | fp64 mean {y.mean():.4f} std {y.std():.4f} n {len(y)} | "
              f"diff {x.mean() - y.mean():+.4f} = {(x.mean() - y.mean()) / se:+.2f} se; std ratio {x.std() / y.std():.4f}")
    print(f"   early terminations: fp32 {int(((first32 >= 0) & (first32 < 1250)).sum())} fp64 {int(((first64 >= 0) & (first64 < 1250)).sum())}")
    dm = q32.mean(0) - q64.mean(0)
    se = np.sqrt(q32.var(0, ddof=1) / N + q64.var(0, ddof=1) / N)
    print("   final qpos[7:] mean diff / se per joint:", np.round(dm / se, 2).tolist())
    print("   final qpos[7:] std ratio per joint:", np.round(q32.std(0) / q64.std(0), 3).tolist())
    print("   final qpos[7:] mean fp32:", np.round(q32.mean(0), 4).tolist())
    print("   final qpos[7:] std  fp32:", np.round(q32.std(0), 4).tolist())


def part_c(T=1300, K=50):
    env, ora = make_env(1, 3), orc.OracleEnv(1, seed=3)
    env.reset(); ora.reset()
    rng = np.random.default_rng(0)
    by_age = [[] for _ in range(K)]
    ndone_mismatch = 0
    for t in range(T):
        if t % K == 0:
            sync(env, ora)
        a = rng.uniform(-1, 1, (1, 18)).astype(np.float32)
        obs, _, rew, done, _ = env.step(torch.from_numpy(a))
        oobs, orew, odone, _ = ora.step(a)
        ndone_mismatch += int(done[0]) != int(odone[0])
        by_age[t % K].append(max(float(np.abs(obs.cpu().numpy() - oobs).max()), abs(float(rew[0]) - float(orew[0]))))
    print(f"C: one env, {T} steps, sync every {K}: done mismatches {ndone_mismatch}")
    for lo, hi in ((0, 1), (1, 5), (5, 10), (10, 20), (20, 35), (35, 50)):
        x = np.concatenate([by_age[k] for k in range(lo, hi)])
        print(f"   steps since sync {lo:2d}..{hi - 1:2d}: n {len(x):4d} median {np.median(x):.2e} p90 {np.percentile(x, 90):.2e} max {x.max():.2e}")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "abcd"
    if "a" in which: part_a()
    if "b" in which: part_b()
    if "c" in which: part_c()
    if "d" in which: part_a(scale=0.12, settle=600, label="D")
