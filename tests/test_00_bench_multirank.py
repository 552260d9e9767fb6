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


def test_rccl_world_size_one_probe():
    """RCCL itself, on the card: a one-rank `nccl` group runs every collective the N>1 job issues (returns all-gather on device tensors,
    the gradient | KL all-reduce of the data-parallel mini-batch, advantage statistics, parameter broadcast) - tests/tools/rccl_probe.py."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29529", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NM_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "rccl_probe.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    try:      # the whole transcript for the record (gpurun_out/ is what comes back from the GPU box)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "rccl_probe.log"), "w") as f:
            f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    except OSError:
        pass
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "RCCL_PROBE_OK" in r.stdout
    print(r.stdout.strip().splitlines()[-1])


def test_bench_py_called_directly_starts_its_own_ranks():
    """`python bench.py --gpus 2 --steps 20 --warmup 5` - the shape of the driver's N=1 command, no torch.distributed.run in front:
    bench.py starts the two ranks as child processes itself (gloo rehearsal: they share the one card) and relays rank 0's line."""
    env = dict(os.environ, NM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--envs-per-gpu", "1024"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["collectives_timed"] >= 1 and out["gather_order_mismatches"] == 0
    # without the rehearsal switch a 2-rank RCCL job on a 1-GPU box is refused up front (exit 2), not left to fail inside RCCL
    import torch
    if torch.cuda.device_count() < 2:
        env.pop("NM_DIST_BACKEND")
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 2 and "HIP device(s) visible" in r.stderr


def test_train_py_called_directly_starts_its_own_ranks(tmp_path):
    env = dict(os.environ, NM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--gpus", "2", "-e", "512", "--iters", "2"], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert len([l for l in r.stdout.splitlines() if l.startswith("it ")]) == 2
    assert "replicas: 2 ranks, max |parameter difference to rank 0| = 0.000e+00" in r.stdout


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
    ppo = out["ppo_end_to_end_env_steps_per_s"]             # BASELINE config 5 at N > 1: the data-parallel PPO loop, whole-job env-steps/s
    assert isinstance(ppo, dict) and ppo["value"] > 0 and ppo["envs_total"] == 2 * 1024, ppo
    assert ppo["logging"].startswith("pipelined") and ppo["rollout"].startswith("one launch"), ppo
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


# BASELINE configs[3] / configs[4] name 8 ranks x 4096 envs. This pool's process guard allows at most 6 processes on one card, so the
# rehearsal on the one-GPU box keeps the TOTAL (32768 envs) and uses 4 ranks x 8192; the 8-shard layout itself is checked bit for bit in
# tests/test_gpu_configs.py::test_config4_32768_envs_equal_eight_shards_of_4096, and 8 gloo ranks on CPU tensors in
# tests/test_sharding_gloo.py::test_eight_rank_gather_is_ordered_by_global_env_id.
def test_four_rank_bench_at_the_config4_job_size():
    env = dict(os.environ, NM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "100", "--warmup", "20",
           "--envs-per-gpu", "8192"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["config"]["envs_per_gpu"] == 8192 and out["scaling"] == "weak"
    assert out["collectives_timed"] >= 2                       # step 80 of the rollout horizon and the last timed step
    assert out["gather_order_mismatches"] == 0                 # every rank found its own returns at [rank*E, (rank+1)*E) of the gathered vector
    assert out["counters"]["contacts_dropped"] == 0 and out["counters"]["bad_state_resets"] == 0
    assert abs(out["value"] - 4 * 8192 * 100 / (out["ms_per_step"] * 100 / 1e3)) < 1e-6 * out["value"]


def test_four_rank_train_py_at_the_config5_job_size(tmp_path):
    """BASELINE configs[4]: train.py at 32768 envs in total, sharded by global env id over the ranks, fused collection per rank, fused
    data-parallel update (one all-reduce of gradient | KL per mini-batch): 3 iterations, the replicas end bit-identical, no contact
    dropped, no bad-state reset on any shard."""
    env = dict(os.environ, NM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", "29539", os.path.join(ROOT, "train.py"), "-e", "32768", "--iters", "3"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("it ")]
    assert len(lines) == 3 and "nan" not in lines[-1], lines
    assert "replicas: 4 ranks, max |parameter difference to rank 0| = 0.000e+00" in r.stdout, r.stdout[-1500:]
    assert "'contacts_dropped': 0" in r.stdout and "'bad_state_resets': 0" in r.stdout
