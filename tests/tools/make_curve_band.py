"""tests/golden/curve_cpu_band_4096.json and the round's HIP-vs-CPU table from the runs of tests/tools/curve_vs_cpu.py:
  python tests/tools/make_curve_band.py <cpu runs json> <hip runs json> <out table json> [tag]
(CPU side: --kinds cpu in the build container; HIP side: --kinds hip on the GPU box. Test infrastructure.)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cpu = [r for r in json.load(open(sys.argv[1]))["runs"] if r["kind"] == "cpu"]
hip = [r for r in json.load(open(sys.argv[2]))["runs"] if r["kind"] == "hip"]
tag = sys.argv[4] if len(sys.argv) > 4 else "r05"
marks = (10, 25, 50, 75, 100, 150)
at = lambda r, m: float(np.mean(r["mean_step_reward"][max(0, m - 5):m]))
table = []
for m in marks:
    row = {"iteration": m}
    for kind, runs in (("hip", hip), ("cpu", cpu)):
        v = np.array([at(r, m) for r in runs if len(r["mean_step_reward"]) >= m])
        if len(v):
            row.update({kind + "_mean": float(v.mean()), kind + "_min": float(v.min()), kind + "_max": float(v.max()), kind + "_seeds": len(v)})
    table.append(row)
    print("| %d | %+.4f [%+.4f, %+.4f] | %+.4f [%+.4f, %+.4f] |" % (m, row["hip_mean"], row["hip_min"], row["hip_max"], row["cpu_mean"], row["cpu_min"], row["cpu_max"]))
json.dump(dict(envs=4096, table=table, runs=hip + cpu), open(sys.argv[3], "w"))
band = {str(m): {"mean": r["cpu_mean"], "min": r["cpu_min"], "max": r["cpu_max"], "seeds": [x["seed"] for x in cpu]} for m, r in zip(marks, table) if m <= 50}
json.dump({"what": "mean reward per env-step (average over the 5 iterations before the checkpoint) of the on-policy runner trained on the CPU ORACLE env (fp64, torch PPO), "
                   "4096 envs, 80 steps per iteration, reference hyper-parameters; made by tests/tools/curve_vs_cpu.py --kinds cpu --envs 4096 --iters 150 --seeds 3 in the build "
                   f"container on the round-5 model tables (profiles/{tag}_curve_cpu_runs.json holds the full runs), condensed by tests/tools/make_curve_band.py",
           "envs": 4096, "band": band}, open(os.path.join(ROOT, "tests", "golden", "curve_cpu_band_4096.json"), "w"))
