#!/usr/bin/env python
"""bench.py - env-steps/s of NightmareV3Env.step() on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu 4096]

One "step" = one step() call over all envs of a rank = `decimation` (2) physics substeps of 8 ms + the env
epilogue (obs / rewards / termination / reset), random actions already resident in HBM. N>1 is launched by
torch.distributed.run, one rank per GPU: envs shard contiguously by global id (rank r owns [r*E, (r+1)*E)), the
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def host_cores():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota (a GPU box gives a share of a big host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 16)) if n > 64 else n   # no quota visible on a >64-thread host: stay within the documented 16-CPU share


def cpu_baseline(seconds_budget=15.0):
    """The CPU oracle (a port: MuJoCo itself is not installable) on this box's host cores, bounded sample."""
    from oracle import oracle as orc
    cores = host_cores()
    n = 1024
    env = orc.OracleEnv(n, seed=0, num_threads=cores)
    env.reset()
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-1, 1, (n, 18)).astype(np.float32) for _ in range(8)]
    for i in range(10):
        env.step(acts[i % 8])
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds_budget:
        env.step(acts[k % 8])
        k += 1
    dt = time.perf_counter() - t0
    return {"value": n * k / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {k} random-action steps from reset (fp64 C restatement of mj_step + env epilogue, OpenMP over envs)"}


def cpu_baseline_detail(seconds_each=6.0):
    """the CPU path at T = 1 and T = all cores, N = 1 and N = 4096 (the reference's own case is N = 1, T = 1)."""
    from oracle import oracle as orc
    out = {}
    for n in (1, 4096):
        for threads in (1, host_cores()):
            if n == 1 and threads > 1:
                continue
            env = orc.OracleEnv(n, seed=0, num_threads=threads)
            env.reset()
            rng = np.random.default_rng(0)
            acts = [rng.uniform(-1, 1, (n, 18)).astype(np.float32) for _ in range(8)]
            for i in range(3):
                env.step(acts[i % 8])
            t0 = time.perf_counter()
            k = 0
            while time.perf_counter() - t0 < seconds_each:
                env.step(acts[k % 8])
                k += 1
            out[f"N{n}_T{threads}"] = n * k / (time.perf_counter() - t0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-detail", action="store_true", help="also time the CPU oracle at N in {1, 4096} x T in {1, all cores} (adds ~20 s)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    import __graft_entry__ as ge
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

    def one_step(i):
        nonlocal returns, gathered
        _, _, rew, done, _ = env.step(acts[i % pool])
        returns += rew
        if (i + 1) % horizon == 0 and world > 1:     # PPO-update boundary: one all-gather of per-env returns over xGMI
            gathered = gather_returns(returns, total_envs=world * E)
            returns.zero_()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        one_step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    out = None
    if rank == 0:
        value = world * E * args.steps / dt
        # roofline leg: HIP events around the dominant (step) kernel on its launch stream, separate pass
        env.profile(True)
        for i in range(min(args.steps, 300)):
            env.step(acts[i % pool])
        k_ms, k_n = env.profile(False)
        k_avg = k_ms / max(k_n, 1) * 1e-3
        achieved = B_FULL * E / k_avg / 1e9
        traffic = None   # HBM-side bytes per launch from the committed rocprofv3 PMC passes of this kernel (profiles/)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            traffic = tj["bytes_per_launch"] * (E / 4096.0)
        except Exception:
            pass
        # physics-only (BASELINE config 2) and step + 2x256 MLP policy forward (config 3), for DESIGN.md / the log
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(200):
            env.step_physics(acts[i % pool])
        torch.cuda.synchronize(dev)
        phys = E * 200 / (time.perf_counter() - t1)
        # config 3: policy forward (66->256->256->18, exact-f32 MFMA) + step(), closed loop, replayed as ONE HIP graph per
        # step (policy kernel + step kernel + extras kernel) so the host launch path is off the critical path
        net = ActorMLP([66, 256, 256, 18]).to(dev)
        obs = env.get_observations()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for i in range(5):
                env.step(net(obs))
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            env.step(net(obs))
        for i in range(20):
            graph.replay()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(300):
            graph.replay()
        torch.cuda.synchronize(dev)
        closed = E * 300 / (time.perf_counter() - t1)
        out = {
            "metric": "env-steps/sec at N parallel Nightmare-v3 envs, 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{E} Nightmare-v3 envs per GPU, random-action rollout U(-1,1)^18, full step(): 2 x 8 ms substeps "
                                   "(18-DoF dynamics + floor contact, PGS x3 + noslip x4) + obs/reward/termination/reset",
                       "envs_per_gpu": E, "decimation": 2, "sharding": f"dp{world} by env id, all-gather of returns every {horizon} steps"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_env_step<float,2>", "kernel_avg_us": k_avg * 1e6,
                         "algorithmic_bytes_per_env_step": B_FULL,
                         "note": "latency/VALU-issue bound, not HBM bound: see DESIGN.md"},
            "physics_only_env_steps_per_s": phys,
            "closed_loop_mlp_2x256_env_steps_per_s": closed,
            "counters": env.counters(),
        }
        if not args.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline()
            if args.cpu_detail:
                out["cpu_baseline"]["detail_env_steps_per_s"] = cpu_baseline_detail()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
