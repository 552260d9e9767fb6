"""N>1 path on CPU: world_size-2 gloo processes. Each rank steps ITS shard of the envs with the host emulation of the
device kernel (global-id keyed RNG), then the ranks all-gather returns; the result must equal the unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL, STEPS = 6, 12


def _rollout(lo, hi, seed):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emul import emul as em
    env = em.EmulEnv(hi - lo, double=False, seed=seed, env_off=lo)
    env.eplen[:] = 620                               # crosses the 625-step command resample -> exercises the RNG keying
    rng = np.random.default_rng(0)
    ret = np.zeros(hi - lo, np.float32)
    obs = None
    for t in range(STEPS):
        a = rng.uniform(-1, 1, (TOTAL, 18)).astype(np.float32)[lo:hi]
        obs, rew, done, _ = env.step(a)
        ret += rew
    return ret, obs, env.get("cmd")


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nightmare_rl_amd.distributed import gather_returns, global_advantage_stats, shard_range
    lo, hi = shard_range(TOTAL, rank, world)
    ret, obs, cmd = _rollout(lo, hi, seed=42)
    allret = gather_returns(torch.from_numpy(ret), total_envs=TOTAL)
    mean, std = global_advantage_stats(torch.from_numpy(ret).double())
    q.put((rank, lo, hi, allret.numpy(), float(mean), float(std), cmd))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_rollout_equals_single_rank():
    sys.path.insert(0, ROOT)
    from nightmare_rl_amd.distributed import shard_range
    assert [shard_range(7, r, 3) for r in range(3)] == [(0, 3), (3, 5), (5, 7)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full_ret, _, full_cmd = _rollout(0, TOTAL, seed=42)
    for rank, lo, hi, allret, mean, std, cmd in res:
        np.testing.assert_array_equal(allret, full_ret)              # bitwise: sharding does not change any env's result
        np.testing.assert_array_equal(cmd, full_cmd[lo:hi])
        assert abs(mean - full_ret.astype(np.float64).mean()) < 1e-9
        assert abs(std - full_ret.astype(np.float64).std(ddof=1)) < 1e-9
    assert np.abs(full_cmd).sum() > 0
