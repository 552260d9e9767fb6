"""Copy what scripts/measure_round.sh <tag> left under gpurun_out/ into profiles/ (the tracked, judged copies) and derive
profiles/<tag>_pmc_step_kernel.json, which bench.py reads for roofline.traffic / roofline.valu.   usage: python scripts/collect_profiles.py r02"""
import json, os, shutil, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
for name in ("bench.json.log", "bench_kernel_stats.csv", "train_kernel_stats.csv", "mlp_kernel_stats.csv", "pmc_step_summary.txt", "pmc_mlp_summary.txt",
             "curve_vs_cpu.json", "curve.log", "train40.log", "play.log", "parity_report.txt", "wavetimes.txt", "stage_stamps.txt", "rolloutbench.txt", "rolloutwaves.txt",
             "pmc_lanes_summary.txt", "pmc_lanes.json", "bench_driverargs.json.log", "ppobench.txt", "ppo_stage_stamps.txt", "pmc_ppo_summary.txt",
             "pmc_ppo_mem_summary.txt"):
    src = os.path.join(G, f"{tag}_{name}")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{tag}_{name}"))
        print("copied", name)
sj = os.path.join(G, f"pmc_{tag}_summary.json")
if os.path.exists(sj):
    m = json.load(open(sj))
    kb = 1024.0
    n_simd, n_xcd = 1024, 8
    gui = m.get("GRBM_GUI_ACTIVE")            # summed over the XCDs
    out = {
        "source": f"rocprofv3 --kernel-trace --pmc passes of scripts/pmcrun.py (4096 envs, 360 launches), medians per launch: gpurun_out/pmc_{tag}",
        "kernel": "k_env_step<float,2>",
        "FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
        "hbm_bytes_per_launch": (m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * kb,
        "hbm_bytes_per_launch_if_reads_are_half_counted": (2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * kb,
        "note": "FETCH_SIZE under-counts wide (16 B/lane) reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section); this kernel's reads are mostly dword gathers, "
                "so the width is uncalibrated: the true value lies between the two figures. Infinity-Cache hits are included in both counters.",
        "SQ_INSTS_VALU": m.get("SQ_INSTS_VALU"), "SQ_WAVES": m.get("SQ_WAVES"), "SQ_WAVE_CYCLES_quad": m.get("SQ_WAVE_CYCLES"),
        "wave_active_frac": m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
        "wave_wait_any_frac": m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
    }
    if gui and m.get("SQ_INSTS_VALU"):
        simd_cycles = gui / n_xcd * n_simd
        out["valu"] = {"definition": "SQ_INSTS_VALU x 4 cycles / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs)", "frac_of_4_cycle_issue": m["SQ_INSTS_VALU"] * 4 / simd_cycles,
                       "frac_of_simd32_2_cycle_issue": m["SQ_INSTS_VALU"] * 2 / simd_cycles, "insts_per_wave": m["SQ_INSTS_VALU"] / max(m.get("SQ_WAVES", 1), 1)}
    json.dump(out, open(os.path.join(P, f"{tag}_pmc_step_kernel.json"), "w"), indent=1)
    print("wrote", f"profiles/{tag}_pmc_step_kernel.json")
