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


def _visible_list(name, env):
    """Entries of a *_VISIBLE_DEVICES variable up to the first invalid one (the runtimes stop there), or None when it is unset."""
    if name not in env:
        return None
    out = []
    for tok in env[name].split(","):
        tok = tok.strip()
        if not tok or tok == "-1" or (tok.lstrip("-").isdigit() and int(tok) < 0):
            break
        out.append(tok)
    return out


def count_gpus_sysfs(sysfs_root="/sys/class/kfd/kfd/topology/nodes", dev_root="/dev/dri", env=None):
    """GPUs this process tree can open, WITHOUT loading the HIP / HSA runtime: KFD topology nodes with `simd_count` > 0 (CPU nodes
    have 0) whose DRM render node (`drm_render_minor`) is readable and writable here - the same test the ROCm runtime makes when it
    enumerates agents, so a container that was handed one card of eight counts one -, then narrowed by ROCR_VISIBLE_DEVICES and
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES. Returns None when the topology directory does not exist (no amdgpu driver here)."""
    import os
    env = os.environ if env is None else env
    if not os.path.isdir(sysfs_root):
        return None
    n = 0
    for node in sorted(os.listdir(sysfs_root)):
        props = {}
        try:
            with open(os.path.join(sysfs_root, node, "properties")) as f:
                for line in f:
                    kv = line.split()
                    if len(kv) == 2:
                        props[kv[0]] = kv[1]
        except OSError:
            continue                                  # a node the device cgroup hides
        if int(props.get("simd_count", "0")) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1"))
        if minor >= 0 and not os.access(os.path.join(dev_root, f"renderD{minor}"), os.R_OK | os.W_OK):
            continue
        n += 1
    for name in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        lst = _visible_list(name, env)
        if lst is not None:
            n = min(n, len(lst))
    return n


def count_gpus_in_child():
    """The fallback when sysfs says nothing: ask a throw-away child process (which may initialise whatever it likes) and read one number."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (IndexError, ValueError):
        return 0


def count_gpus():
    """Device count for the parent of a multi-rank job. No torch.cuda / HIP call in THIS process, by construction (a process that has
    initialised the GPU must not start the ranks by exec, and on some torch builds torch.cuda.device_count() is hipGetDeviceCount)."""
    n = count_gpus_sysfs()
    return count_gpus_in_child() if n is None else n


def self_launch(script, argv, nproc):
    """`python bench.py --gpus N` / `python train.py --gpus N` started WITHOUT torch.distributed.run: start the N ranks as CHILD
    processes (one `python -m torch.distributed.run --nproc-per-node N script argv...`), let them inherit stdout / stderr (rank 0's
    output is the job's output) and return their exit status. Must be called before this process has touched the GPU; nothing in here
    does: devices are counted from the KFD topology in sysfs (or by a throw-away child process), never through torch.cuda. With fewer
    devices than ranks the launch is refused (exit status 2) unless NM_DIST_BACKEND=gloo asks for the shared-card rehearsal explicitly.
    The rendezvous is torch.distributed.run's own `--standalone` on 127.0.0.1: the launcher picks and holds its port itself (no
    bind / close / reuse window in this process)."""
    import os
    import subprocess
    import sys
    nproc = int(nproc)
    if os.environ.get("NM_DIST_BACKEND", "nccl") == "nccl":
        ndev = count_gpus()
        if ndev < nproc:
            print(f"{os.path.basename(script)}: {nproc} ranks asked for, {ndev} HIP device(s) visible (RCCL needs one device per rank; "
                  "NM_DIST_BACKEND=gloo rehearses the multi-rank path with the ranks sharing a card)", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between the ranks of one node on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(nproc), script] + list(argv)
    return subprocess.run(cmd, env=env).returncode
