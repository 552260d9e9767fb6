"""Multi-GPU layout of the env batch: contiguous shards by global env id, one process per GPU, and the single
collective the path has - an all-gather of per-env episode returns at PPO-update boundaries (RCCL over xGMI on
GPUs; gloo on CPU tensors in tests). The rollout itself needs no communication."""
import torch
import torch.distributed as dist


def shard_range(total_envs, rank, world_size):
    """Global env ids [lo, hi) owned by `rank`: contiguous blocks, remainder spread over the first ranks."""
    base, rem = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_returns(local_returns, total_envs=None, force=False):
    """All ranks get the per-env returns of every env, ordered by global env id. One collective call.
    force: issue the collective also in a one-rank group (the RCCL probe of tests/tools/rccl_probe.py)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return local_returns.clone()
    world = dist.get_world_size()
    n = torch.tensor([local_returns.numel()], device=local_returns.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    if total_envs is not None:
        counts = [shard_range(total_envs, r, world)[1] - shard_range(total_envs, r, world)[0] for r in range(world)]
    else:
        dist.all_gather(sizes, n)
        counts = [int(s.item()) for s in sizes]
    if len(set(counts)) == 1:
        out = torch.empty(sum(counts), dtype=local_returns.dtype, device=local_returns.device)
        dist.all_gather_into_tensor(out, local_returns.contiguous())
        return out
    # unequal shards (total not a multiple of the rank count): neither gloo nor RCCL gathers ragged tensors - pad to the longest shard,
    # still ONE collective, then drop the padding
    m = max(counts)
    padded = torch.zeros(m, dtype=local_returns.dtype, device=local_returns.device)
    padded[:local_returns.numel()] = local_returns
    out = torch.empty(world * m, dtype=local_returns.dtype, device=local_returns.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * m:r * m + c] for r, c in enumerate(counts)])


def global_advantage_stats(adv, force=False):
    """Mean / std of advantages over ALL ranks (rsl_rl normalises over envs x steps): one 3-float all-reduce."""
    s = torch.stack([adv.sum(), (adv * adv).sum(), torch.tensor(float(adv.numel()), device=adv.device, dtype=adv.dtype)])
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dist.all_reduce(s)
    mean = s[0] / s[2]
    var = (s[1] / s[2] - mean * mean).clamp_min(0) * (s[2] / (s[2] - 1).clamp_min(1))
    return mean, var.sqrt()


def self_launch(script, argv, nproc):
    """`python bench.py --gpus N` / `python train.py --gpus N` started WITHOUT torch.distributed.run: start the N ranks as CHILD
    processes (one `python -m torch.distributed.run --nproc-per-node N script argv...`), let them inherit stdout / stderr (rank 0's
    output is the job's output) and return their exit status. Must be called before this process has touched the GPU: nothing in here
    does (torch.cuda.device_count() does not initialise HIP on this image), and the caller exits with the returned code. With fewer
    devices than ranks the launch is refused unless NM_DIST_BACKEND=gloo asks for the shared-card rehearsal explicitly."""
    import os
    import socket
    import subprocess
    import sys
    nproc = int(nproc)
    ndev = torch.cuda.device_count()
    if ndev < nproc and os.environ.get("NM_DIST_BACKEND", "nccl") == "nccl":
        print(f"{os.path.basename(script)}: {nproc} ranks asked for, {ndev} HIP device(s) visible (RCCL needs one device per rank; "
              "NM_DIST_BACKEND=gloo rehearses the multi-rank path with the ranks sharing a card)", file=sys.stderr)
        return 2
    with socket.socket() as s:           # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between the ranks of one node on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    return subprocess.run(cmd, env=env).returncode
