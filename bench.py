#!/usr/bin/env python
"""bench.py - env-steps/s of NightmareV3Env.step() on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu 4096]

One "step" = one step() call over all envs of a rank = `decimation` (2) physics substeps of 8 ms + the env
epilogue (obs / rewards / termination / reset), random actions already resident in HBM. N>1: one rank per GPU, either started by
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` or - when called directly - by bench.py itself, which then
starts exactly that command as a child process before touching the GPU; envs shard contiguously by global id (rank r owns [r*E, (r+1)*E)), the
rollout needs no communication, and every 80 steps (num_steps_per_env, reference envs/nightmare_v3_config.py:135)
the ranks all-gather their per-env returns over RCCL, as a PPO-update boundary would. Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_FULL = 1092  # algorithmic HBM bytes per env-step, full step() (452 read + 640 written)
B_DYN = 656    # dynamics-only (config 2)
B_ROLLOUT = 1513  # one env-step inside the one-launch rollout: step() without the action read, + the observation read back from storage + the transition row
MFMA_F32_PEAK_TF = 157.3  # dense f32 MFMA: 256 CUs x 256 flop/clk x 2.4 GHz (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def host_cores():
    """(threads the CPU leg will use, CPUs visible to this process). Threads = the affinity mask, capped by the cgroup CPU quota when
    one is visible; on a host with more than 64 visible CPUs and no visible quota (a GPU box is a share of a big host) the leg stays
    within the 16-CPU share documented for one GPU. `cores` in the JSON line is the number of OpenMP threads actually used."""
    vis = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = vis
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return (max(1, min(n, 16)) if n > 64 else n), vis


def _time_oracle(n, threads, seconds, physics_only=False):
    from oracle import oracle as orc
    env = orc.OracleEnv(n, seed=0, num_threads=threads)
    env.reset()
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-1, 1, (n, 18)).astype(np.float32) for _ in range(8)]
    step = env.step_physics if physics_only else env.step
    for i in range(3):
        step(acts[i % 8])
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        step(acts[k % 8])
        k += 1
    return n * k / (time.perf_counter() - t0), k


def live_pmc(E):
    """HBM-side bytes per launch and VALU issue of the step kernel, measured DURING this run (the line carried the builder's
    box's counters): rocprofv3 --pmc passes over scripts/pmcrun.py (the same kernel, 4096 envs, 360 random-action steps) as CHILD processes -
    the profiler cannot attach to this one -, one pass per counter group with --kernel-trace only, as MI355X_MICROARCH.md prescribes. Medians
    over the launches after the first quarter. None when rocprofv3 is missing or a pass fails (the committed profile is used then)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="nm_bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    med = {}
    try:
        lanes_ctrs = ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32"]
        def one_pass(k, ctrs, script, pat, prefix=""):
            d = os.path.join(out, f"p{k}")
            r = subprocess.run([exe, "--kernel-trace", "--output-format", "csv", "--pmc", *ctrs, "-d", d, "-o", "run", "--", "python3", os.path.join(ROOT, "scripts", script), "4096"],
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=90)
            if r.returncode != 0:
                raise RuntimeError("rocprofv3 pass failed")
            vals = {}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if pat in row["Kernel_Name"]:
                        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for c in ctrs:
                v = sorted(vals[c][len(vals[c]) // 4:])
                med[prefix + c] = v[len(v) // 2]
        for k, ctrs in enumerate((["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_WAVES"], ["GRBM_GUI_ACTIVE"], lanes_ctrs)):
            one_pass(k, ctrs, "pmcrun.py", "k_env_step")
        rollout = None
        try:      # the one-launch rollout kernel's HBM side (4096 envs x 80 steps per launch, scripts/pmcrollout.py): two more passes
            one_pass(5, ["FETCH_SIZE"], "pmcrollout.py", "k_env_rollout", "roll_")
            one_pass(6, ["WRITE_SIZE"], "pmcrollout.py", "k_env_rollout", "roll_")
            rollout = {"hbm_bytes_per_launch": (med["roll_FETCH_SIZE"] + med["roll_WRITE_SIZE"]) * 1024.0, "envs": 4096, "steps": 80}
        except Exception:
            rollout = None
        simd_cycles = med["GRBM_GUI_ACTIVE"] / 8 * 1024          # the counter is summed over the 8 XCDs; 1024 SIMDs
        lanes = med["SQ_THREAD_CYCLES_VALU"] / max(med["SQ_ACTIVE_INST_VALU"], 1.0)
        flop = (med["SQ_INSTS_VALU_ADD_F32"] + med["SQ_INSTS_VALU_MUL_F32"] + 2 * med["SQ_INSTS_VALU_FMA_F32"] + med["SQ_INSTS_VALU_TRANS_F32"]) * lanes * (E / 4096.0)
        return {"fp32_flop_per_launch": flop, "active_lanes_per_valu_instruction": lanes, "rollout": rollout,
                "FETCH_SIZE_KB": med["FETCH_SIZE"], "WRITE_SIZE_KB": med["WRITE_SIZE"],
                "hbm_bytes_per_launch": (med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024.0 * (E / 4096.0),
                "hbm_bytes_per_launch_if_reads_are_half_counted": (2 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024.0 * (E / 4096.0),
                "valu": {"definition": "SQ_INSTS_VALU x 4 cycles / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs)", "frac_of_4_cycle_issue": med["SQ_INSTS_VALU"] * 4 / simd_cycles,
                         "frac_of_simd32_2_cycle_issue": med["SQ_INSTS_VALU"] * 2 / simd_cycles, "insts_per_wave": med["SQ_INSTS_VALU"] / max(med["SQ_WAVES"], 1.0)}}
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def cpu_baseline(envs=4096):
    """The CPU oracle (a port: MuJoCo itself is not installable) on this box's host cores: the same workload (N envs, random
    actions, full step()) on a bounded sample, plus the grid SURVEY 8(d) / BASELINE.md ask for - N in {1, N} x T in {1, all cores},
    with the env epilogue (obs / reward / reset: what `value` is) and without it (mj_step x decimation only, the shape of the
    reference's simple_test.py:25-45). It is a C restatement with no Python in the loop, i.e. an upper bound for the reference's own
    MuJoCo + numpy path. ~25 s in total."""
    cores, visible = host_cores()
    v_all, k = _time_oracle(envs, cores, 10.0)
    grid = {f"N{envs}_T{cores}": v_all, "N1_T1": _time_oracle(1, 1, 3.0)[0], f"N{envs}_T1": _time_oracle(envs, 1, 4.0)[0],
            f"N{envs}_T{cores}_physics_only": _time_oracle(envs, cores, 4.0, True)[0], "N1_T1_physics_only": _time_oracle(1, 1, 2.0, True)[0],
            f"N{envs}_T1_physics_only": _time_oracle(envs, 1, 3.0, True)[0]}
    return {"value": v_all, "unit": "env-steps/s", "cores": cores, "host_cpus_visible": visible, "kind": "port",
            "sample": f"{envs} envs x {k} random-action steps from reset (fp64 C restatement of mj_step + env epilogue, OpenMP over envs)",
            "grid_env_steps_per_s": grid}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks, one per GPU (default: WORLD_SIZE under a launcher, else 1)")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not start the rocprofv3 --pmc child passes (roofline.traffic / valu then come from profiles/)")
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import __graft_entry__ as ge
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` called directly: build once, then start the N ranks as child processes of THIS process (which has
        # not touched the GPU and does not) and leave with their status; rank 0's JSON line goes to the inherited stdout
        from nightmare_rl_amd.distributed import self_launch
        ge.compile_only()
        raise SystemExit(self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: pass the rank count the launcher started")
    if rank == 0:
        ge.build()
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # NM_DIST_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks then share devices)
        backend = os.environ.get("NM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank %= max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
        dist.barrier()
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
    from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
    from nightmare_rl_amd.policy import ActorMLP
    from nightmare_rl_amd.distributed import gather_returns

    E = args.envs_per_gpu
    cfg = NightmareV3Config()
    cfg.env.num_envs = E
    env = NightmareV3Env(cfg, device=dev, seed=0, env_id_offset=rank * E)
    env.reset()
    # synthetic random actions ~ U(-1,1), keyed by global env id so results do not depend on the GPU count
    pool = 64
    gen = torch.Generator().manual_seed(1234)
    acts = (torch.rand(pool, world * E, 18, generator=gen) * 2 - 1)[:, rank * E:(rank + 1) * E].contiguous().to(dev)
    returns = torch.zeros(E, device=dev)
    gathered = torch.zeros(world * E, device=dev) if world > 1 else None
    horizon = 80

    ncoll = 0
    mism = torch.zeros((), dtype=torch.int64, device=dev)      # gathered[rank*E:(rank+1)*E] != what this rank contributed (device-side count)

    env.accumulate_rewards_into(returns)     # returns[env] += reward inside the step kernel (nm_set_return_accumulator), not a launch of its own

    def one_step(i, last=False):
        nonlocal returns, gathered, ncoll, mism
        env.step(acts[i % pool])
        # PPO-update boundary: one all-gather of per-env returns over xGMI, every `horizon` steps and at the end of the run (so a
        # short timed region still contains the collective)
        if ((i + 1) % horizon == 0 or last) and world > 1:
            gathered = gather_returns(returns, total_envs=world * E)
            mism += (gathered[rank * E:(rank + 1) * E] != returns).sum()     # ordered by global env id: no host synchronisation here
            returns.zero_()
            ncoll += 1

    def sync(collective=True):
        if world > 1 and collective:
            dist.barrier()
        # spin on an event first: the blocking wait behind torch.cuda.synchronize() wakes up tens of microseconds after the last
        # kernel has ended, which a 20-step timed region (1.2 ms) would carry as 3 us per step
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        while not ev.query():
            pass
        torch.cuda.synchronize(dev)

    # Steady state first ("steady state, after >= 100 warm-up steps"): a timed region of a millisecond at the start of a process
    # runs at the clocks of an idle GPU (scripts/warmclock.py: 57.8 us/step cold, 53.4 after 0.2 s of sustained load - round 3 reported the
    # difference as `same_region_at_sustained_clock`). So before the W warm-up steps the caller asks for, the same step() runs untimed until
    # about 0.25 s of load have passed (and at least 100 warm-up steps in total); the count is in the line as `settle_steps`. No collective in here.
    settle = max(0, 100 - args.warmup) + 4608          # a FIXED count: 4608 steps = 0.25 s at 54 us; the state entering the timed region is reproducible
    for i in range(settle):
        env.step(acts[i % pool])
    torch.cuda.synchronize(dev)
    returns.zero_()
    for i in range(args.warmup):
        one_step(i, last=i == args.warmup - 1)
    sync()
    ncoll = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i, last=i == args.steps - 1)
    sync()
    dt = time.perf_counter() - t0
    env.accumulate_rewards_into(None)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if world > 1:
        dist.all_reduce(mism)

    out = None
    if rank == 0:
        value = world * E * args.steps / dt
        # roofline leg: HIP events around the dominant (step) kernel on its launch stream, separate pass
        # (a) ONE event pair on the launch stream around n back-to-back launches: span / n is the average launch duration including the
        #     ~1 us between two launches, i.e. an upper bound of what rocprofv3 --kernel-trace reports per launch - the roofline uses it;
        # (b) the library's own event pair around every launch (nm_profile): each pair costs the queue ~2 us, reported for reference.
        n_leg = max(300, min(args.steps, 1000))               # >= 300 launches whatever --steps is
        stream = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(20):
            env.step(acts[i % pool])
        e0.record(stream)
        for i in range(n_leg):
            env.step(acts[i % pool])
        e1.record(stream)
        e1.synchronize()
        k_avg = e0.elapsed_time(e1) / n_leg * 1e-3
        env.profile(True)
        for i in range(n_leg):
            env.step(acts[i % pool])
        k_ms, k_n = env.profile(False)
        k_avg_pairs = k_ms / max(k_n, 1) * 1e-3
        achieved = B_FULL * E / k_avg / 1e9
        # The regime training runs in: random U(-1,1) actions keep the robots flailing (about 1.4 floor contacts per
        # env), a policy that has learnt to stand keeps 4..6 feet down (5.3 on average after 150 PPO iterations, profiles/r04_trained_policy_regime.txt)
        # and the constraint stage then solves 16..24 rows per env. The standing regime is reproduced without a checkpoint by SMALL actions
        # around the servo's rest pose: 0.12 x the same U(-1,1) stream (zero actions = all six feet down). Same kernel, same launch count,
        # its own event pair; the contact histogram of both regimes is read from the kernel's debug buffer in untimed steps.
        def contact_hist(action_scale, nsteps=40):
            dbg = torch.zeros(E, 256, device=dev)
            env.set_debug_buffer(dbg)
            h = np.zeros(48, dtype=np.int64)
            for i in range(nsteps):
                env.step(acts[i % pool] * action_scale)
                h += np.bincount(dbg[:, 160].cpu().numpy().astype(np.int64), minlength=48)[:48]
            env.set_debug_buffer(None)
            return {"mean_contacts_per_env": float((h * np.arange(48)).sum() / h.sum()), "histogram": {int(k): int(v) for k, v in enumerate(h) if v}}
        regime = None
        try:
            hist_random = contact_hist(1.0)
            kScale = 0.12
            acts_stand = (acts * kScale).contiguous()
            env.reset()
            for i in range(300):                                  # settle into the stance
                env.step(acts_stand[i % pool])
            hist_stand = contact_hist(kScale)
            for i in range(20):
                env.step(acts_stand[i % pool])
            e0.record(stream)
            for i in range(n_leg):
                env.step(acts_stand[i % pool])
            e1.record(stream)
            e1.synchronize()
            ks = e0.elapsed_time(e1) / n_leg * 1e-3
            sync(collective=False)                               # rank 0 only in here: no barrier
            t1 = time.perf_counter()
            for i in range(args.steps):
                env.step(acts_stand[i % pool])
            sync(collective=False)
            dts = time.perf_counter() - t1
            regime = {"actions": f"{kScale} x U(-1,1)^18 around the servo rest pose: the robots stand (the regime of a trained policy: 5.3 contacts per env)",
                      "value": E * args.steps / dts, "ms_per_step": dts / args.steps * 1e3, "kernel_avg_us": ks * 1e6, "kernel_launches_timed": n_leg,
                      "roofline_frac_hbm": B_FULL * E / ks / 1e9 / HBM_PEAK_GBS, **hist_stand,
                      "headline_regime": {"actions": "U(-1,1)^18", "kernel_avg_us": k_avg * 1e6, **hist_random}}
            env.reset()
            for i in range(100):
                env.step(acts[i % pool])
        except Exception as exc:       # a secondary figure: never fails the headline
            regime = f"failed: {type(exc).__name__}: {exc}"
        # The same region once more (reset, W warm-up steps, K timed steps) right after the >= 600 back-to-back launches above: a short
        # timed region at the start of a process runs at the clocks of an idle GPU (scripts/warmclock.py: 57.8 us/step cold, 53.4 us/step
        # after 0.2 s of sustained load); `value` above is the cold one when --steps is small. Reported, not used for `value`.
        sustained = None
        if world == 1:
            env.reset()
            for i in range(args.warmup):
                env.step(acts[i % pool])
            sync()
            t1 = time.perf_counter()
            for i in range(args.steps):
                env.step(acts[(args.warmup + i) % pool])
            sync()
            dts = time.perf_counter() - t1
            sustained = {"value": E * args.steps / dts, "ms_per_step": dts / args.steps * 1e3,
                         "note": "the timed region repeated (reset, warm-up, timed steps) right after >= 600 back-to-back launches: GPU at its sustained clock"}
        # HBM-side bytes per launch and VALU issue utilisation are NOT measured by this process: they come from the committed
        # rocprofv3 --pmc passes of this same command (separate runs, as the counter guide prescribes); null when there are none
        traffic = valu = None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_step_kernel.json")))     # the newest round's counters
        pmc_file = os.path.relpath(cands[-1], ROOT) if cands else None
        try:
            tj = json.load(open(os.path.join(ROOT, pmc_file)))
            traffic = tj["hbm_bytes_per_launch"] * (E / 4096.0)
            valu = tj.get("valu")
        except Exception:
            pmc_file = None
        # secondary roofline: fp32 operations the kernel actually executes (rocprofv3 --pmc pass committed under profiles/, the same
        # way as `traffic`) over THIS run's kernel time, against the 157.3 TFLOP/s fp32 vector peak of MI355X_MICROARCH.md
        valu_fp32 = None
        lf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_lanes.json")))
        if lf:
            try:
                lj = json.load(open(lf[-1]))
                flop = lj["fp32_flop_per_launch"] * (E / float(lj["envs"]))
                valu_fp32 = {"bound": "valu-f32", "achieved": flop / k_avg / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flop / k_avg / 1e12 / 157.3,
                             "fp32_flop_per_launch": flop, "active_lanes_per_valu_instruction": lj["active_lanes_per_valu_instruction"],
                             "source": os.path.relpath(lf[-1], ROOT) + " (counters not measured in this run; kernel time is)"}
            except Exception:
                valu_fp32 = None
        # physics-only (BASELINE config 2) and step + 2x256 MLP policy forward (config 3), for DESIGN.md / the log
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(200):
            env.step_physics(acts[i % pool])
        torch.cuda.synchronize(dev)
        phys = E * 200 / (time.perf_counter() - t1)
        # config 3: policy forward (66->256->256->18, exact-f32 MFMA) + step(), closed loop: observation -> actions -> step -> observation.
        # One HIP graph holds 40 such steps (the PPO runner captures its 80-step rollout the same way), so neither the host launch
        # path nor a per-step graph launch sits between the kernels. An even number of steps: the env alternates two observation
        # buffers, and a graph replays fixed addresses.
        net = ActorMLP([66, 256, 256, 18]).to(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for i in range(6):
                env.step(net(env.get_observations()))
        torch.cuda.current_stream(dev).wait_stream(side)
        per_graph = 40
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            o = env.get_observations()
            for i in range(per_graph):
                o = env.step(net(o))[0]
        for i in range(3):
            graph.replay()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(10):
            graph.replay()
        torch.cuda.synchronize(dev)
        closed = E * 10 * per_graph / (time.perf_counter() - t1)
        # ... and the policy kernel of that loop alone (k_mlp_fused, 66 -> 256 -> 256 -> 18 at E rows = 174 080 FLOP per row): its share of the
        # dense f32 MFMA peak, driver-observed
        o_fix = env.get_observations().clone()
        for i in range(20):
            net(o_fix)
        e0.record(stream)
        for i in range(200):
            net(o_fix)
        e1.record(stream)
        e1.synchronize()
        t_mlp = e0.elapsed_time(e1) / 200 * 1e-3
        mlp = {"kernel": "k_mlp_fused", "shape": "66-256-256-18", "rows": E, "avg_us": t_mlp * 1e6, "tflops": 174080.0 * E / t_mlp / 1e12,
               "frac_of_f32_mfma_peak": 174080.0 * E / t_mlp / 1e12 / MFMA_F32_PEAK_TF, "peak_tflops": MFMA_F32_PEAK_TF}
        # the collection loop of the PPO runner (reference train.py:54: act -> step -> process_env_step, 80 steps, the reference's 66-54-42-30-18|1
        # networks) as ONE launch with the policy inside the env's wave (nm_rollout): policy + step + transition record per env-step
        roll = roll_roof = None
        try:
            from nightmare_rl_amd.rl import ActorCritic, RolloutStorage
            from nightmare_rl_amd.rl.fused import FusedCollector, FusedUpdate
            torch.manual_seed(0)
            ac = ActorCritic(66, 66, 18, actor_hidden_dims=[54, 42, 30], critic_hidden_dims=[54, 42, 30], activation="elu", init_noise_std=1.0).to(dev)
            fu = FusedUpdate(ac, torch.optim.Adam(ac.parameters(), lr=1e-3), dev, lr=1e-3)
            col = FusedCollector(ac, E, dev, seed=1, update=fu)
            if col.can_rollout(env):
                T = horizon
                st = RolloutStorage(E, T, [66], [None], [18], dev)
                zz = lambda *sh: torch.zeros(*sh, device=dev)
                cr, cl, fin = zz(E), zz(E), zz(3)
                eidx = torch.tensor([env._stat_names.index(k[4:]) for k in sorted(env.extras["episode"])], dtype=torch.int32, device=dev)
                eacc = zz(eidx.numel())
                for i in range(2):
                    col.rollout(env, st, T, 0.99, cr, cl, fin, ep=(eidx, eacc))
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for i in range(5):
                    col.rollout(env, st, T, 0.99, cr, cl, fin, ep=(eidx, eacc))
                torch.cuda.synchronize(dev)
                roll = E * T * 5 / (time.perf_counter() - t1)
                # its roofline: one launch = T steps of E envs; algorithmic bytes per env-step = step()'s 1092 minus the
                # 72 B of actions it no longer reads, plus what the collection loop keeps: the previous observation read back from its storage
                # row (264) and the transition row written (actions 72, log-prob 4, value 4, mean 72, sigma 72, reward 4, done 1) = 1513 B
                e0.record(stream)
                for i in range(5):
                    col.rollout(env, st, T, 0.99, cr, cl, fin, ep=(eidx, eacc))
                e1.record(stream)
                e1.synchronize()
                t_roll = e0.elapsed_time(e1) / 5 * 1e-3
                rj = None
                cr_ = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_rollout.json")))
                if cr_:
                    try:
                        rj = json.load(open(cr_[-1]))
                    except Exception:
                        rj = None
                roll_roof = {"bound": "hbm", "kernel": "k_env_rollout (+ k_rollout_tail)", "achieved": B_ROLLOUT * E * T / t_roll / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": B_ROLLOUT * E * T / t_roll / 1e9 / HBM_PEAK_GBS, "launch_avg_us": t_roll * 1e6, "us_per_step": t_roll * 1e6 / T, "steps_per_launch": T,
                             "algorithmic_bytes_per_env_step": B_ROLLOUT,
                             "traffic": rj and rj.get("hbm_bytes_per_launch") and rj["hbm_bytes_per_launch"] * (E / float(rj.get("envs", 4096))) * (T / float(rj.get("steps", 80))),
                             "l2_to_cu_read_GBps_under_profiler": rj and rj.get("l2_read_GBps"),
                             "traffic_source": cr_ and rj and os.path.relpath(cr_[-1], ROOT) + " (rocprofv3 --pmc passes over scripts/pmcrollout.py; not measured in this run)"}
        except Exception as exc:       # a secondary figure: never fails the headline
            roll = f"failed: {type(exc).__name__}: {exc}"
        # BASELINE config 5 on this one GPU: the reference's full PPO loop (train.py:54 -> OnPolicyRunner.learn: 80 steps per env, 5 epochs x 4
        # mini-batches, the 54/42/30 networks) on the hand-written kernels, env-steps/s end to end, mean of iterations 6..15 of 16
        ppo_e2e = None
        if world == 1:
            try:
                from nightmare_rl_amd.envs.helpers import class_to_dict
                from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3ConfigPPO
                from nightmare_rl_amd.rl import OnPolicyRunner
                torch.manual_seed(0)
                cfgp = NightmareV3Config()
                cfgp.env.num_envs = E
                envp = NightmareV3Env(cfgp, device=dev, seed=0)
                runner = OnPolicyRunner(envp, class_to_dict(NightmareV3ConfigPPO()), log_dir=None, device=str(dev))
                runner.learn(16, init_at_random_ep_len=True)
                h = runner.history[6:]
                ppo_e2e = sum(r["fps"] for r in h) / len(h)
                envp.close()
            except Exception as exc:   # a secondary figure: never fails the headline
                ppo_e2e = f"failed: {type(exc).__name__}: {exc}"
        # the fp64 verification build of the same kernel (what the exact-parity tests run)
        cfg64 = NightmareV3Config()
        cfg64.env.num_envs = E
        env64 = NightmareV3Env(cfg64, device=dev, seed=0, dtype=torch.float64)
        env64.reset()
        for i in range(5):
            env64.step(acts[i % pool])
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(40):
            env64.step(acts[i % pool])
        torch.cuda.synchronize(dev)
        f64 = E * 40 / (time.perf_counter() - t1)
        env64.close()
        out = {
            "metric": "env-steps/sec at N parallel Nightmare-v3 envs, 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "settle_steps": settle, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{E} Nightmare-v3 envs per GPU, random-action rollout U(-1,1)^18, full step(): 2 x 8 ms substeps "
                                   "(18-DoF dynamics + floor contact, PGS x3 + noslip x4) + obs/reward/termination/reset",
                       "envs_per_gpu": E, "decimation": 2,
                       "sharding": f"dp{world} by env id, all-gather of returns every {horizon} steps and on the last timed step"},
            "collectives_timed": ncoll, "gather_order_mismatches": int(mism.item()),
            "same_region_at_sustained_clock": sustained,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": pmc_file and f"{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; not measured in this run)",
                         "kernel": "k_env_step<float,2>", "kernel_avg_us": k_avg * 1e6, "kernel_launches_timed": n_leg,
                         "kernel_avg_us_event_pair_per_launch": k_avg_pairs * 1e6,
                         "algorithmic_bytes_per_env_step": B_FULL,
                         "valu": valu, "valu_fp32": valu_fp32,
                         "note": "latency/VALU-issue bound, not HBM bound: see DESIGN.md"},
            "contact_regime": regime,
            "physics_only_env_steps_per_s": phys,
            "closed_loop_mlp_2x256_env_steps_per_s": closed,
            "mlp_2x256": mlp,
            "policy_rollout_one_launch_env_steps_per_s": roll, "rollout_roofline": roll_roof, "ppo_end_to_end_env_steps_per_s": ppo_e2e,
            "fp64_verification_kernel_env_steps_per_s": f64,
            "counters": env.counters(),
        }
        if not args.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(E)
        if not args.no_cpu_baseline and not args.no_live_pmc and world == 1 and E == 4096:
            # after every timed leg: the counters of THIS box, this run (child processes; this process issues nothing meanwhile)
            lp = live_pmc(E)
            if lp:
                out["roofline"]["traffic"] = lp["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes over scripts/pmcrun.py (KB x 1024; "
                                                     f"reads {lp['FETCH_SIZE_KB']:.0f} KB, possibly counted at half on gfx950: upper bound {lp['hbm_bytes_per_launch_if_reads_are_half_counted']:.0f} B; writes {lp['WRITE_SIZE_KB']:.0f} KB)")
                out["roofline"]["valu"] = dict(lp["valu"], source="measured in this run (rocprofv3 --pmc SQ_INSTS_VALU / SQ_WAVES / GRBM_GUI_ACTIVE child passes)")
                if lp.get("rollout") and isinstance(out.get("rollout_roofline"), dict):
                    rr = lp["rollout"]
                    out["rollout_roofline"]["traffic"] = rr["hbm_bytes_per_launch"] * (E / float(rr["envs"])) * (out["rollout_roofline"].get("steps_per_launch", 80) / float(rr["steps"]))
                    out["rollout_roofline"]["traffic_source"] = "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes over scripts/pmcrollout.py (KB x 1024, per launch of 80 steps)"
                tf = lp["fp32_flop_per_launch"] / k_avg / 1e12
                out["roofline"]["valu_fp32"] = {"bound": "valu-f32", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3, "fp32_flop_per_launch": lp["fp32_flop_per_launch"],
                                                "active_lanes_per_valu_instruction": lp["active_lanes_per_valu_instruction"],
                                                "source": "measured in this run: (add + mul + 2 fma + trans f32 wave-instructions) x active lanes per VALU instruction "
                                                          "(SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU), rocprofv3 --pmc child pass; over this run's kernel time"}
    if world > 1:
        # BASELINE config 5 at N > 1: the full PPO loop, `E` envs per rank, data-parallel update (one RCCL all-reduce of
        # gradient | KL per mini-batch, inside the update's graph), whole-job env-steps/s. Runs LAST and under a deadline: the headline
        # numbers above are complete, and if this leg should hang on some node (it is the one part of this file that only a multi-GPU
        # node ever runs over RCCL) every rank still leaves with status 0 after rank 0 has printed the line, the leg marked "timed out".
        import threading

        def bail():
            if rank == 0:
                if out.get("ppo_end_to_end_env_steps_per_s") is None:
                    out["ppo_end_to_end_env_steps_per_s"] = "timed out (multi-rank PPO leg exceeded its deadline)"
                print(json.dumps(out), flush=True)
            os._exit(0)

        dist.barrier()
        watchdog = threading.Timer(float(os.environ.get("NM_BENCH_PPO_DEADLINE_S", "240")), bail)
        watchdog.daemon = True
        watchdog.start()
        ppo_e2e = None
        try:
            from nightmare_rl_amd.envs.helpers import class_to_dict
            from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3ConfigPPO
            from nightmare_rl_amd.rl import OnPolicyRunner
            torch.manual_seed(rank)
            cfgp = NightmareV3Config()
            cfgp.env.num_envs = E
            envp = NightmareV3Env(cfgp, device=dev, seed=0, env_id_offset=rank * E)
            runner = OnPolicyRunner(envp, class_to_dict(NightmareV3ConfigPPO()), log_dir=None, device=str(dev))
            runner.learn(16, init_at_random_ep_len=True)
            h = runner.history[6:]
            ppo_e2e = {"value": sum(r["fps"] for r in h) / len(h), "envs_total": world * E, "update": "data-parallel, all-reduce of gradient | KL per mini-batch",
                       "update_graph": getattr(runner.alg, "_upd_graph", None) not in (None, "failed"), "logging": runner.logging_mode, "rollout": runner.rollout_mode}
            envp.close()
        except Exception as exc:
            ppo_e2e = f"failed: {type(exc).__name__}: {exc}"
        if rank == 0:
            out["ppo_end_to_end_env_steps_per_s"] = ppo_e2e
        dist.barrier()                      # still under the deadline: a communicator left unusable by a failed leg must not hang the line either
        dist.destroy_process_group()
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
