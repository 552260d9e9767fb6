"""bench.py's N>1 path rehearsed on the one-GPU box: two ranks (gloo; both share the card) launched the way the driver launches
them, `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --steps 20 --warmup 5`. The JSON line must report a
whole-job value for 2 x E envs and at least one all-gather INSIDE the timed region (with the driver's --steps 20 the every-80-steps
boundary alone would never fire). Named test_00_* so that it runs before anything in this pytest process has touched the GPU: the
ranks are started as child processes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_times_a_collective():
    env = dict(os.environ, NM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--envs-per-gpu", "1024"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["scaling"] == "weak"
    assert out["collectives_timed"] >= 1
    assert out["config"]["envs_per_gpu"] == 1024
    assert abs(out["value"] - 2 * 1024 * 20 / (out["ms_per_step"] * 20 / 1e3)) < 1e-6 * out["value"]   # whole-job aggregate
    assert "cpu_baseline" not in out                                                                    # rank 0 at N=1 only


def test_two_rank_fused_ppo_update_equals_the_single_process_update():
    """The data-parallel update path of rl/ppo.py (FusedUpdate.minibatch_data_parallel: gradient | KL in ONE all-reduce per mini-batch)
    with two ranks sharing the card over gloo: ranks stay bit-identical and match a full-batch single-process update."""
    env = dict(os.environ, NM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "tools", "ppo_two_ranks.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "PPO_TWO_RANKS_OK" in r.stdout


def test_two_rank_train_py_runs_the_fused_data_parallel_loop(tmp_path):
    """BASELINE config 5's launch shape in miniature: train.py under torch.distributed.run, env ids sharded over the ranks, fused
    collection per rank, data-parallel fused update."""
    env = dict(os.environ, NM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29535", os.path.join(ROOT, "train.py"), "-e", "1024", "--iters", "4"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("it ")]
    assert len(lines) == 4 and "nan" not in lines[-1], lines
    assert "contacts_dropped': 0" in r.stdout
