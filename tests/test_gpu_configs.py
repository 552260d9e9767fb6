"""-m gpu tests AT THE SIZE of the BASELINE.json configurations that are not the bench line itself:
  configs[1]  4096 envs, dynamics + contact kernel only (nm_step_physics) - against the oracle's mj_step x 2, not against another HIP path
  configs[3]  32768 envs = 8 shards of 4096: one 32768-env batch is bit-equal, rows [g*4096, (g+1)*4096), to eight 4096-env objects
              built with env_id_offset = g*4096 (what the 8 ranks of one node own)
  boundary    the tensor step() hands out survives the next step() (what an unmodified rsl_rl runner relies on)
The multi-process halves of configs[3] / configs[4] (bench.py / train.py under torch.distributed.run) live in
tests/test_00_bench_multirank.py, because their ranks must start before this process has touched the GPU."""
import numpy as np
import pytest
import torch

from test_gpu_parity import check_outlier, kOutlierBound, kTol, make_env

pytestmark = pytest.mark.gpu


def _settled_population(oracle_mod, N, steps, seed):
    """N oracle envs driven by random actions from reset until most of them stand, lie or flail on the floor (contacts of every kind)."""
    ora = oracle_mod.OracleEnv(N, seed=seed, num_threads=16)
    ora.reset()
    rng = np.random.default_rng(seed)
    for t in range(steps):
        ora.step(rng.uniform(-1, 1, (N, 18)).astype(np.float32))
    return ora, rng


def _obs_equivalent_error(q, v, oq, ov):
    """State error in the units of the stated tolerance (obs scales, envs/nightmare_v3_config.py:68-71): joint angles x1, joint
    velocities x0.05, base linear velocity x2, base angular velocity x0.25; base position / quaternion x1."""
    return np.maximum.reduce([np.abs(q - oq).max(axis=1), 0.05 * np.abs(v[:, 6:] - ov[:, 6:]).max(axis=1),
                              2.0 * np.abs(v[:, :3] - ov[:, :3]).max(axis=1), 0.25 * np.abs(v[:, 3:6] - ov[:, 3:6]).max(axis=1)])


def test_config2_physics_only_4096_envs_fp64_kernel_equals_the_oracle(oracle_mod):
    """BASELINE configs[1] (reference simple_test.py:25-45 shape: ctrl, mj_step(model, data[i], decimation), nothing else) at 4096 envs:
    teacher-forced, the fp64 build of the step kernel in physics-only mode against the oracle's mj_step x 2 - qpos, qvel and
    qacc_warmstart to 1e-8 over 10 steps x 4096 envs of a population that is on the floor."""
    N, T = 4096, 10
    ora, rng = _settled_population(oracle_mod, N, 45, seed=31)
    env = make_env(N, dtype=torch.float64, seed=31)
    worst = 0.0
    ncon = 0
    for t in range(T):
        a = rng.uniform(-1, 1, (N, 18)).astype(np.float32)
        env.set_state(*ora.get_state())
        env.step_physics(torch.from_numpy(a))
        ora.step_physics(a)
        q, v, w = env.get_state()
        oq, ov, ow = ora.get_state()
        worst = max(worst, np.abs(q - oq).max(), np.abs(v - ov).max())
        assert np.abs(w - ow).max() < 1e-5 * max(1.0, np.abs(ow).max())       # accelerations are O(1e3): relative
        ncon += sum(int(ora.data(i).ncon > 0) for i in range(0, N, 16))         # a sample of the envs: contacts of the last substep
    assert worst < 1e-8, worst
    assert ncon > (N // 16) * T // 2, ncon                                     # the population really is in contact
    assert env.counters()["contacts_dropped"] == 0 and env.counters()["bad_state_resets"] == 0


def test_config2_physics_only_4096_envs_fp32_kernel_within_stated_tolerance(oracle_mod):
    """The production fp32 kernel in the same configuration: the state after one physics-only step is within 1e-4 of the oracle's in
    observation units; an env-step above it must be a proven discrete collision decision of bounded size (the contract of
    test_fp32_outliers_are_proven_discrete_events). One bound differs from that test's: here the POST-step velocity is compared
    (step() reports the velocity of the last forward pass, env.py:217), so a contact that exists in one precision only (`act`: its
    penetration is within 2e-7 m of zero) shows up with the whole approach speed of that vertex it removes: <= 0.5 m/s x the
    lin_vel scale 2 = 1.0 (measured on MI355X: one such env-step in 40 960, 0.32)."""
    bounds = dict(kOutlierBound, act=1.0)
    N, T = 4096, 10
    ora, rng = _settled_population(oracle_mod, N, 45, seed=32)
    env = make_env(N, seed=32)
    errs, nout = [], 0
    for t in range(T):
        a = rng.uniform(-1, 1, (N, 18)).astype(np.float32)
        q0, v0, w0 = ora.get_state()
        env.set_state(q0, v0, w0)
        env.step_physics(torch.from_numpy(a))
        ora.step_physics(a)
        q, v, _ = env.get_state()
        oq, ov, _ = ora.get_state()
        e = _obs_equivalent_error(q, v, oq, ov)
        errs.append(e)
        for i in np.nonzero(e > kTol)[0]:
            check_outlier(oracle_mod, e[i], q0[i], v0[i], w0[i], a[i], q0[i, 7:], f"t={t} env={i}", bounds=bounds)
            nout += 1
    e = np.concatenate(errs)
    assert np.median(e) < 2e-6 and np.percentile(e, 99) < 1e-5, (np.median(e), np.percentile(e, 99))
    assert nout <= len(e) // 500, (nout, len(e))
    assert env.counters()["contacts_dropped"] == 0 and env.counters()["bad_state_resets"] == 0


def test_config4_32768_envs_equal_eight_shards_of_4096():
    """BASELINE configs[3]: 32768 envs sharded over 8 GPUs by contiguous global env id. On one GPU: ONE 32768-env batch against EIGHT
    4096-env objects with env_id_offset = g*4096 (exactly what rank g owns): observations, rewards, dones, time-outs and commands are
    bit-equal row for row, through a command resample (step 625) and a time-out reset (step 1251), i.e. through every consumer of the
    per-env random stream."""
    G, E, T = 8, 4096, 12
    N = G * E
    gen = torch.Generator().manual_seed(4)
    acts = [(torch.rand(N, 18, generator=gen) * 2 - 1).cuda() for _ in range(T)]
    ep0 = torch.cat([torch.full((E // 2,), 620, dtype=torch.int64), torch.full((E // 2,), 1246, dtype=torch.int64)]).repeat(G).cuda()

    def run(n, off):
        env = make_env(n, seed=17, env_id_offset=off)
        env.reset()
        env.episode_length_buf = ep0[off:off + n].clone()
        out = []
        for t in range(T):
            obs, _, rew, done, extras = env.step(acts[t][off:off + n])
            out.append((obs.clone(), rew.clone(), done.clone(), extras["time_outs"].clone()))
        cmd = env.commands
        c = env.counters()
        env.close()
        return out, cmd, c

    full, fcmd, fc = run(N, 0)
    assert fc["contacts_dropped"] == 0 and fc["bad_state_resets"] == 0
    nto = sum(int((o[3] * o[2]).sum()) for o in full)      # extras['time_outs'] is only refreshed by a step in which an env reset (env.py:344)
    assert 0.9 * (N // 2) <= nto <= N // 2                                  # the late half times out once (minus the few that fell first)
    for g in range(G):
        shard, scmd, _ = run(E, g * E)
        sl = slice(g * E, (g + 1) * E)
        for t in range(T):
            for a, b in zip(full[t], shard[t]):
                assert torch.equal(a[sl], b), (g, t)
        np.testing.assert_array_equal(fcmd[sl], scmd)
    # shards differ from each other (the random stream is keyed by the GLOBAL id, not the local row)
    assert not np.array_equal(fcmd[:E], fcmd[E:2 * E])


def test_observation_survives_the_next_step_like_rsl_rl_needs():
    """rsl_rl v1.0.2's order (caller reference train.py:54): PPO.act keeps `transition.observations = obs` (no copy), then
    env.step(actions), then storage.add_transitions copies the kept tensor. The env alternates two observation buffers, so the kept
    tensor still holds what the policy acted on; without that the storage would silently receive the NEXT observation."""
    N, T = 512, 7
    env = make_env(N, seed=8)
    obs, _ = env.reset()
    store = torch.zeros(T, N, 66, device="cuda")
    gen = torch.Generator().manual_seed(0)
    for t in range(T):
        kept = obs                                         # PPO.act: transition.observations = obs
        witness = obs.clone()                              # test only: what the policy saw
        actions = (torch.rand(N, 18, generator=gen) * 2 - 1).cuda()
        obs, _, rew, done, _ = env.step(actions)           # the env writes its OTHER buffer
        store[t].copy_(kept)                               # add_transitions, after the step
        assert torch.equal(store[t], witness), t
        assert obs.data_ptr() != kept.data_ptr()
        assert env.get_observations() is obs
        assert not torch.equal(obs, witness)               # and the new observation really is new


def test_time_outs_refresh_is_incremental_on_its_own_buffer_and_full_on_any_other():
    """extras['time_outs'] (env.py:369-371) is refreshed by the launch's closing wave: on the buffer its last refresh wrote it only
    clears that refresh's ones and sets the new ones; a buffer it has not written (first call, a new tensor - here one filled with
    garbage) must come out completely rewritten. Steps without a reset leave the vector alone, like the reference (env.py:344)."""
    N = 256
    env = make_env(N, seed=4)
    env.reset()
    zero = torch.zeros(N, 18)

    def step_with_timeouts(ids):
        env.episode_length_buf = torch.zeros(N, dtype=torch.int64, device="cuda")
        env.episode_length_buf[ids] = 1250                      # -> 1251 > max_episode_length: these time out in this step
        _, _, _, done, extras = env.step(zero)
        return done.cpu().numpy(), extras["time_outs"].cpu().numpy()

    ids1 = torch.tensor([3, 64, 200])
    done, to = step_with_timeouts(ids1)
    want = np.zeros(N, np.float32); want[ids1.numpy()] = 1
    np.testing.assert_array_equal(to, want)
    assert done[ids1.numpy()].all()
    ids2 = torch.tensor([5, 64, 255, 17])                       # incremental: 3 and 200 cleared, 64 stays set, three new ones
    done, to = step_with_timeouts(ids2)
    want = np.zeros(N, np.float32); want[ids2.numpy()] = 1
    np.testing.assert_array_equal(to, want)
    env.time_out_buf = torch.full((N,), 7.0, device="cuda")     # another buffer: must be rewritten in full
    env.extras.pop("time_outs", None); env._fill_extras()
    ids3 = torch.tensor([9])
    done, to = step_with_timeouts(ids3)
    want = np.zeros(N, np.float32); want[9] = 1
    np.testing.assert_array_equal(to, want)
    # a step in which no env resets leaves the vector as it is (robots standing still on the ground: no fall within one step)
    env.episode_length_buf = torch.full((N,), 10, dtype=torch.int64, device="cuda")
    _, _, _, done, extras = env.step(zero)
    if int(done.sum()) == 0:
        np.testing.assert_array_equal(extras["time_outs"].cpu().numpy(), want)
    # a replacement that lands on the SAME ADDRESS (a caching allocator hands a freed block back) cannot be told from "the
    # buffer I refreshed last" by the device - the env notices the new tensor object and invalidates (nm_invalidate_time_outs)
    addr = env.time_out_buf.data_ptr()
    same_place = env.time_out_buf.view(-1)                      # another tensor object on the same memory ...
    same_place.fill_(7.0)                                       # ... with contents the incremental refresh would leave behind
    env.time_out_buf = same_place
    assert env.time_out_buf.data_ptr() == addr
    ids4 = torch.tensor([11, 12])
    done, to = step_with_timeouts(ids4)
    want = np.zeros(N, np.float32); want[ids4.numpy()] = 1
    np.testing.assert_array_equal(to, want)
    assert env.extras["time_outs"] is env.time_out_buf
