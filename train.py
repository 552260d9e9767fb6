#!/usr/bin/env python
"""train.py - the reference's training entry point (reference train.py:1-54) on the MI355X backend.
Same flags (-r resume, -v render [rejected: no viewer], -n num_threads [accepted, unused], -e envs, -p resume_path) plus
--iters / --seed / --gpus. Multi-GPU: `python train.py --gpus G -e <total envs>` (starts its G ranks as child processes) or
`python -m torch.distributed.run --nproc-per-node G train.py -e <total envs>`."""
import argparse
import datetime
import os

import torch

from nightmare_rl_amd.distributed import shard_range
from nightmare_rl_amd.envs.helpers import class_to_dict, get_load_path
from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config, NightmareV3ConfigPPO
from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
from nightmare_rl_amd.rl import OnPolicyRunner


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-r", "--resume", action="store_true", default=False, dest="resume")
    ap.add_argument("-v", "--render", action="store_true", default=False, dest="render")
    ap.add_argument("-n", "--num_threads", type=int, default=1, dest="num_threads")
    ap.add_argument("-e", "--envs", type=int, default=2048, dest="num_envs")
    ap.add_argument("-p", "--resume_path", type=str, default=None, dest="resume_path")
    ap.add_argument("--iters", type=int, default=None, help="learning iterations (default: cfg.runner.max_iterations)")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); without torch.distributed.run, train.py starts them itself")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus is not None and args.gpus > 1:
        # called directly: the ranks become child processes of this one, which has not touched the GPU (and leaves with their status)
        import sys
        from nightmare_rl_amd.distributed import self_launch
        raise SystemExit(self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("NM_DIST_BACKEND", "nccl")   # gloo: rehearse the multi-rank path on a box with fewer GPUs than ranks
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("NM_FORCE_DATA_PARALLEL") == "1":      # (the switch: one rank down the multi-rank path, for a kernel trace)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    log_root = "logs/nightmare_v3/"
    log_dir = f"logs/nightmare_v3/{datetime.datetime.now()}/"
    cfg, train_cfg = NightmareV3Config(), NightmareV3ConfigPPO()
    cfg.viewer.render = args.render
    lo, hi = shard_range(args.num_envs, rank, world)
    cfg.env.num_envs = hi - lo
    train_cfg.runner.resume = args.resume
    seed = train_cfg.seed if args.seed is None else args.seed
    torch.manual_seed(seed + rank)
    env = NightmareV3Env(cfg, log_dir=log_dir, num_threads=args.num_threads, device=f"cuda:{local_rank}", seed=seed, env_id_offset=lo)
    runner = OnPolicyRunner(env, class_to_dict(train_cfg), log_dir=log_dir, device=f"cuda:{local_rank}")
    if train_cfg.runner.resume:
        path = get_load_path(args.resume_path or log_root, load_run=train_cfg.runner.load_run, checkpoint=train_cfg.runner.checkpoint)
        print(f"Loading model from: {path}")
        runner.load(path)
    runner.learn(num_learning_iterations=args.iters if args.iters is not None else train_cfg.runner.max_iterations, init_at_random_ep_len=True)
    counters = env.counters()                                # contacts dropped (provably 0), MuJoCo-style bad-state resets, search fallbacks
    if world > 1:
        # every rank must hold the same policy after the data-parallel updates: the largest difference to rank 0's parameters, and the
        # env counters summed over the shards
        dev = f"cuda:{local_rank}"
        flat = torch.cat([p.detach().reshape(-1).float() for p in runner.alg.actor_critic.parameters()]).to(dev)
        ref = flat.clone()
        dist.broadcast(ref, 0)
        diff = (flat - ref).abs().max().reshape(1)
        dist.all_reduce(diff, op=dist.ReduceOp.MAX)
        c = torch.tensor([counters[k] for k in sorted(counters)], device=dev, dtype=torch.float64)
        dist.all_reduce(c)
        counters = {k: int(v) for k, v in zip(sorted(counters), c.tolist())}
        if rank == 0:
            print(f"replicas: {world} ranks, max |parameter difference to rank 0| = {float(diff):.3e}", flush=True)
    if rank == 0:
        print("env counters:", counters, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
