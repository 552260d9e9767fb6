"""fp32 error on tibia-tibia (MPR) states, measured on the GPU box:  python tests/tools/mpr_study.py [N] [T] > profiles/r03_mpr_fp32_study.txt

Population: the crossing-leg states of tests/test_self_collision.py (airborne robots, neighbouring legs swung into each other), N envs x T
teacher-forced steps from the oracle's state. Every env-step is classified by the oracle (a tibia-tibia contact in the last forward pass, or
flagged by the kernel's own pair counter) and the fp32 kernel's observation / reward error against the fp64 oracle is reported for that class;
the largest errors are printed with both sides' contact (depth, normal, position) so that a discrete portal / support choice can be told from
accumulated rounding. Test infrastructure only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc                                           # noqa: E402
from test_self_collision import crossing_states, squeeze_actions            # noqa: E402
from nightmare_rl_amd import _lib                                           # noqa: E402
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config      # noqa: E402
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env            # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dtype = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
    rng = np.random.default_rng(11)
    cfg = NightmareV3Config()
    cfg.env.num_envs = N
    env = NightmareV3Env(cfg, device="cuda:0", seed=0, dtype=dtype)
    real = torch.float64 if dtype == torch.float64 else torch.float32
    dbg = torch.zeros(N, 256, dtype=real, device="cuda")
    env.set_debug_buffer(dbg)
    ora = orc.OracleEnv(N, seed=0, num_threads=16)
    qpos, qvel = crossing_states(N, rng)
    qw = np.zeros((N, 24))
    rows = []
    for t in range(T):
        ora.set_state(qpos, qvel, qw)
        env.set_state(qpos, qvel, qw)
        b = ora.get_buffers()
        env.set_buffers(dof_pos=b["dof_pos"], dof_vel=b["dof_vel"], actions=b["actions"], commands=b["commands"])
        env.episode_length_buf = torch.from_numpy(b["ep_len"]).cuda()
        a = squeeze_actions(N, rng)
        oobs, orew, odone, _ = ora.step(a)
        obs, _, rew, done, _ = env.step(torch.from_numpy(a))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        d = dbg.cpu().numpy()
        err = np.maximum(np.abs(obs.astype(np.float64) - oobs).max(axis=1), np.abs(rew.astype(np.float64) - orew))
        for i in range(N):
            od = ora.data(i)
            nc = od.ncon
            pair = [c for c in range(nc) if od.con_body1[c] > 0]
            kpair = int(d[i, 157])
            rows.append((err[i], len(pair), kpair, t, i, int(done[i] != odone[i]), nc, int(d[i, 160]),
                         [(float(od.con_dist[c]), np.array(od.con_frame[c])[:3].copy(), np.array(od.con_pos[c]).copy()) for c in pair[:1]],
                         (d[i, 96:96 + 16].copy(), d[i, 165:171].copy(), d[i, 112:112 + 6].copy())))
        qpos, qvel, qw = ora.get_state()
        nd = odone != 0
        if nd.any():                                       # re-seed envs that terminated
            q2, v2 = crossing_states(int(nd.sum()), rng)
            qpos[nd], qvel[nd], qw[nd] = q2, v2, 0
    e = np.array([r[0] for r in rows])
    npair = np.array([r[1] for r in rows])
    kp = np.array([r[2] for r in rows])
    cls = {"no pair contact": (npair == 0) & (kp == 0), "pair contact (oracle, last forward pass)": npair > 0,
           "pair only in the kernel's flag (other substep / precision)": (npair == 0) & (kp != 0)}
    print(f"# fp32 kernel vs fp64 oracle, crossing-leg population, {N} envs x {T} teacher-forced steps = {len(e)} env-steps; dtype {dtype}")
    print(f"# done-flag mismatches: {sum(r[5] for r in rows)}; contact-count mismatches (last forward pass): {sum(int(r[6] != r[7]) for r in rows)}")
    for name, m in cls.items():
        x = e[m]
        if len(x) == 0:
            print(f"{name}: 0 env-steps")
            continue
        q = np.percentile(x, [50, 90, 99, 99.9, 100])
        print(f"{name}: {len(x)} env-steps; error median {q[0]:.2e} p90 {q[1]:.2e} p99 {q[2]:.2e} p99.9 {q[3]:.2e} max {q[4]:.2e}; above 1e-4: {(x > 1e-4).sum()} ({(x > 1e-4).mean() * 100:.3f} %)")
    print("# the 25 largest errors among pair-contact env-steps: err | t env | oracle ncon / kernel ncon | oracle pair contact depth normal pos | kernel contact list dists (first 4), normals of contacts 0-1")
    idx = [k for k in np.argsort(-e) if npair[k] > 0 or kp[k] != 0][:25]
    for k in idx:
        r = rows[k]
        oc = r[8][0] if r[8] else None
        kd, kn, kpz = r[9]
        os_ = f"depth {-oc[0]:.3e} n ({oc[1][0]:+.4f} {oc[1][1]:+.4f} {oc[1][2]:+.4f}) pos ({oc[2][0]:+.4f} {oc[2][1]:+.4f} {oc[2][2]:+.4f})" if oc else "none"
        print(f"{r[0]:.3e} | t={r[3]} env={r[4]} | {r[6]} / {r[7]} | {os_} | dists {np.array2string(kd[:4], precision=3)} n0 {np.array2string(kn[:3], precision=4)} n1 {np.array2string(kn[3:], precision=4)} pos0 {np.array2string(kpz[:3], precision=4)}")


if __name__ == "__main__":
    main()
