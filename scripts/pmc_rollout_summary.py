"""Per-launch medians of the counters scripts/pmc_rollout.sh collected for k_env_rollout (80 steps of 4096 envs per launch, policy in the wave),
the derived figures DESIGN / bench.py quote, and a JSON (<dir>_summary.json -> profiles/rNN_pmc_rollout.json) for bench.py's rollout roofline.
  python scripts/pmc_rollout_summary.py gpurun_out/pmc_rollout_<tag> [envs=4096] [steps=80]"""
import csv, glob, json, sys
import numpy as np
d = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 80
pat = "k_env_rollout"
vals, dur = {}, []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
med = {k: float(np.median(v[len(v) // 3:])) for k, v in vals.items()}      # skip the first launches (cold caches, landing transients)
for k in sorted(med):
    print(f"{k:32s} n={len(vals[k]):3d} median per launch = {med[k]:18.1f}")
g = med.get
us = float(np.median(dur[len(dur) // 3:])) if dur else float("nan")
waves = N // 2
print(f"k_env_rollout under the profiler: median {us:.1f} us per launch = {us / T:.2f} us per step over {len(dur)} launches ({N} envs, {T} steps, {waves} waves)")
out = dict(med, envs=N, steps=T, kernel_us_under_profiler=us)
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    print(f"wave cycles: active {100 * g('SQ_ACTIVE_INST_ANY', 0) / wc:.1f} %  wait_any {100 * g('SQ_WAIT_ANY', 0) / wc:.1f} %  wait_inst {100 * g('SQ_WAIT_INST_ANY', 0) / wc:.1f} %"
          f"  | VALU {100 * g('SQ_ACTIVE_INST_VALU', 0) / wc:.1f} %  LDS {100 * g('SQ_ACTIVE_INST_LDS', 0) / wc:.1f} %  VMEM {100 * g('SQ_ACTIVE_INST_VMEM', 0) / wc:.1f} %")
if g("SQ_INSTS_VALU"):
    per = lambda k: g(k, 0) / waves / T
    print(f"instructions per wave and step: VALU {per('SQ_INSTS_VALU'):.0f} (of which MFMA {per('SQ_INSTS_MFMA'):.0f})  SALU {per('SQ_INSTS_SALU'):.0f}  LDS {per('SQ_INSTS_LDS'):.0f}  "
          f"VMEM {per('SQ_INSTS_VMEM'):.0f}  SMEM {per('SQ_INSTS_SMEM'):.0f}")
    if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        # SQ_BUSY_CYCLES counts per SE/XCD ... use GRBM_GUI_ACTIVE x SIMDs when present
        pass
if g("GRBM_GUI_ACTIVE"):
    cyc = g("GRBM_GUI_ACTIVE")                        # summed over the 8 XCDs by rocprofv3: per-XCD active cycles x 8
    simd_cycles = cyc / 8.0 * 1024                    # 1024 SIMDs, each for the launch's duration
    if g("SQ_INSTS_VALU"):
        print(f"VALU issue: SQ_INSTS_VALU x 4 cycles / SIMD-cycles = {100 * g('SQ_INSTS_VALU') * 4 / simd_cycles:.1f} % ; "
              f"matrix pipe: SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles = {100 * g('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd_cycles:.2f} %")
    out["simd_cycles"] = simd_cycles
if g("TCP_TCC_READ_REQ_sum"):
    rd = g("TCP_TCC_READ_REQ_sum")
    print(f"L1 -> L2 read requests per launch {rd:.0f} = {rd / T / 1e6:.2f} M per step; at 64 B each {rd * 64 / T / 1e6:.1f} MB per step"
          f" (the policy's weight stream: 81 fragments x 1 KB x {waves} waves = {81 * 1024 * waves / 1e6:.1f} MB per step); "
          f"mean L1->L2 read latency {g('TCP_TCC_READ_REQ_LATENCY_sum', 0) / rd:.0f} cycles")
    out["l2_read_bytes_per_step_at_64B"] = rd * 64 / T
    if us == us:
        out["l2_read_GBps"] = rd * 64 / (us * 1e-6) / 1e9
        print(f"L2 -> CU read bandwidth {out['l2_read_GBps']:.0f} GB/s under the profiler")
if g("TCC_REQ_sum"):
    print(f"L2: {g('TCC_REQ_sum'):.0f} requests, hit rate {100 * g('TCC_HIT_sum', 0) / max(g('TCC_HIT_sum', 0) + g('TCC_MISS_sum', 0), 1):.1f} %, EA read requests {g('TCC_EA0_RDREQ_sum', 0):.0f}")
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    fetch, write = g("FETCH_SIZE") * 1024.0, g("WRITE_SIZE") * 1024.0     # rocprofv3 reports KiB
    alg_step = 1092 * N                                                      # bytes per env-step of step()
    stor = (66 + 18 + 1 + 1 + 18 + 18 + 1 + 1) * 4 * N                       # + the storage row the rollout writes per step: obs, actions, logp, value, mu, sigma, reward, done(1 B ~ 4)
    print(f"HBM side per launch: FETCH_SIZE {fetch / 1e6:.1f} MB (gfx950 may count wide reads at half: upper bound {2 * fetch / 1e6:.1f} MB), WRITE_SIZE {write / 1e6:.1f} MB"
          f" = {(fetch + write) / T / 1e6:.2f} MB per step; algorithmic: {alg_step / 1e6:.2f} MB of env state / outputs + {stor / 1e6:.2f} MB of storage rows per step")
    out.update(hbm_bytes_per_launch=fetch + write, hbm_bytes_per_step=(fetch + write) / T, fetch_bytes=fetch, write_bytes=write,
               algorithmic_bytes_per_step=alg_step + stor)
json.dump(out, open(d.rstrip("/") + "_summary.json", "w"), indent=1)
