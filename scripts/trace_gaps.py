"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run of train.py: is the device ever waiting for the host between the
mini-batches of an update (the N > 1 iteration must have the shape of the N = 1 one)?
  python scripts/trace_gaps.py <rocprofv3 output dir>"""
import csv, glob, sys
import numpy as np
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f[0]))), key=lambda r: r[0])
names = [r[2] for r in rows]
def short(n):
    for k in ("k_env_rollout", "k_ppo_fwdbwd", "k_ppo_step", "k_ppo_reduce", "k_rollout_tail", "k_gae", "k_ppo_perm", "ncclDevKernel", "rccl", "AllReduce"):
        if k in n: return k
    return n[:40]
# the last 6 iterations: from the 7th-last rollout launch on
ro = [i for i, n in enumerate(names) if "k_env_rollout" in n]
if len(ro) < 8:
    print(d, ": fewer than 8 rollouts in the trace"); sys.exit(0)
i0, i1 = ro[-7], ro[-1]
seg = rows[i0:i1]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
gaps = np.array([seg[k + 1][0] - seg[k][1] for k in range(len(seg) - 1)], dtype=np.float64) * 1e-3
inside = [(gaps[k], short(seg[k][2]), short(seg[k + 1][2])) for k in range(len(gaps))]
upd = np.array([g for g, a, b in inside if a.startswith("k_ppo") or a.startswith("ncclDev") or "rccl" in a.lower()])
coll = sum(1 for n in names[i0:i1] if "nccl" in n.lower() or "rccl" in n.lower() or "allreduce" in n.lower())
print(f"{d}: 6 iterations, {len(seg)} kernels ({coll} RCCL collective kernels), span {span * 1e-6:.2f} ms, GPU busy {100.0 * busy / span:.1f} %")
print(f"  gaps after update kernels (fwdbwd / reduce / all-reduce / step): n={len(upd)} median {np.median(upd):.2f} us p99 {np.percentile(upd, 99):.2f} us max {upd.max():.2f} us")
big = sorted(inside, key=lambda t: -t[0])[:6]
print("  largest gaps (us, after, before):", [(round(g, 1), a, b) for g, a, b in big])
