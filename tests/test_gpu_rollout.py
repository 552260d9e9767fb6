"""The K-steps-per-launch rollout with the policy inside the env's wavefront (nm_rollout; rsl_rl v1.0.2 OnPolicyRunner.learn's collection
loop, caller reference train.py:54, horizon envs/nightmare_v3_config.py:135) against
  (1) plain torch for the policy arithmetic (stated tolerance), and the 16x16x4-tile collection kernel nm_ppo_act for the noise keys;
  (2) the step-by-step path - one nm_rollout_act launch, one nm_step launch, one nm_ppo_record launch per step - BIT FOR BIT over an
      80-step rollout: storage rows, env state, episode bookkeeping (sums that go through float atomics: to rounding);
  (3) the runner: training through the one-launch rollout gives the same learning curve as through the captured step-by-step graph."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_gpu_parity import make_env

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _networks(seed=5, std=0.8):
    from nightmare_rl_amd.rl import ActorCritic
    from nightmare_rl_amd.rl.fused import FusedUpdate
    torch.manual_seed(seed)
    ac = ActorCritic(66, 66, 18, actor_hidden_dims=[54, 42, 30], critic_hidden_dims=[54, 42, 30], activation="elu", init_noise_std=std).to(DEV)
    with torch.no_grad():      # biases away from zero, distinct std per action: nothing in the packing may hide behind a default
        for m in list(ac.actor) + list(ac.critic):
            if isinstance(m, torch.nn.Linear):
                m.bias.uniform_(-0.3, 0.3)
        ac.std.mul_(torch.linspace(0.6, 1.4, 18, device=DEV))
    opt = torch.optim.Adam(ac.parameters(), lr=1e-3)
    fu = FusedUpdate(ac, opt, DEV, lr=1e-3)
    return ac, fu


def _storage(N, T):
    from nightmare_rl_amd.rl import RolloutStorage
    return RolloutStorage(N, T, [66], [None], [18], DEV)


@pytest.mark.parametrize("N", [4096, 1001])
def test_wave_policy_equals_torch_and_shares_the_noise_of_nm_ppo_act(N):
    """PPO.act on the rollout's wave code (4x4x1 MFMA blocks, two envs per wave; N = 1001 leaves half a wave empty): means and value vs
    torch fp32 (tolerance 2e-5 abs: exact-f32 FMA chains in another order), log-probability vs torch.distributions.Normal, sigma = std,
    the stored observation = the input, and the SAME standard-normal draw as nm_ppo_act for the same (seed, iteration, step, env, pair)."""
    from nightmare_rl_amd import _lib
    ac, fu = _networks()
    env = make_env(N)
    st, st2 = _storage(N, 3), _storage(N, 3)
    obs = torch.randn(N, 66, device=DEV) * 2.0
    it = torch.tensor([7], dtype=torch.int64, device=DEV)
    a = env.policy_act(fu.flat, obs, 1234, it, 2, st)
    with torch.no_grad():
        mu, v = ac.actor(obs), ac.critic(obs).squeeze(-1)
    torch.testing.assert_close(st.mu[2], mu, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(st.values[2].squeeze(-1), v, atol=2e-5, rtol=1e-5)
    assert torch.equal(st.sigma[2], ac.std.detach().expand(N, 18)) and torch.equal(st.observations[2], obs) and a.data_ptr() == st.actions[2].data_ptr()
    lp = torch.distributions.Normal(st.mu[2], st.sigma[2]).log_prob(st.actions[2]).sum(-1)
    torch.testing.assert_close(st.actions_log_prob[2].squeeze(-1), lp, atol=2e-4, rtol=1e-5)
    z = (st.actions[2] - st.mu[2]) / st.sigma[2]
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02
    assert float((st.mu[0].abs().sum() + st.mu[1].abs().sum())) == 0.0            # only row 2 was written
    L = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.nm_ppo_act(fu._h, fu.flat.data_ptr(), obs.data_ptr(), N, 1234, it.data_ptr(), 2, st2.actions[2].data_ptr(), st2.actions_log_prob[2].data_ptr(),
                            st2.values[2].data_ptr(), st2.mu[2].data_ptr(), st2.sigma[2].data_ptr(), st2.observations[2].data_ptr(), stream))
    z2 = (st2.actions[2] - st2.mu[2]) / st2.sigma[2]
    torch.testing.assert_close(z, z2, atol=2e-5, rtol=0)
    torch.testing.assert_close(st.mu[2], st2.mu[2], atol=2e-5, rtol=1e-5)
    env.close()


def _record(L, env, st, s, gamma, cur_ret, cur_len, fin, ep_idx, ep_acc):
    from nightmare_rl_amd import _lib
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tout = env.extras["time_outs"].data_ptr() if "time_outs" in env.extras else None      # what rsl_rl's `if 'time_outs' in infos` sees
    _lib.check(L.nm_ppo_record(env.rew_buf.data_ptr(), env.reset_buf.data_ptr(), tout, st.values[s].data_ptr(), float(gamma), env.num_envs,
                               st.rewards[s].data_ptr(), st.dones[s].data_ptr(), cur_ret.data_ptr(), cur_len.data_ptr(), fin.data_ptr(),
                               env._ep_stats.data_ptr(), ep_idx.data_ptr(), int(ep_idx.numel()), ep_acc.data_ptr(), stream))


@pytest.mark.parametrize("N,T,noise,send_timeouts", [(2048, 80, False, True), (63, 100, False, True), (256, 40, True, True), (128, 30, False, False)])
def test_one_launch_rollout_equals_the_step_by_step_path_bit_for_bit(N, T, noise, send_timeouts):
    """nm_rollout(K = T) against T x [nm_rollout_act, nm_step, nm_ppo_record] from the same start (random episode lengths, so that time-outs,
    falls and command resamples happen inside the rollout; the third case with observation noise on; N = 63: steps WITHOUT any reset occur, where extras['time_outs'] and
    extras['episode'] are stale and rsl_rl's process_env_step / the runner's running sum use the stale values - and half a wave is empty).
    Two rollouts back to back, so that what the first one leaves behind (time-out flags, episode sums, counters) is what the second starts from."""
    from nightmare_rl_amd import _lib
    L = _lib.load()
    ac, fu = _networks()
    gamma = 0.99
    envs = [make_env(N, seed=11, noise=noise), make_env(N, seed=11, noise=noise)]     # noise: cfg.noise.add_noise (env.py:304-305), keyed by the step count of the env
    for e in envs:
        e.cfg.env.send_timeouts = send_timeouts       # False: extras carry no 'time_outs' (env.py:369), PPO.process_env_step adds no bootstrap term
        e.reset()
        torch.manual_seed(3)
        e.episode_length_buf = torch.randint(0, 1250, (N,), device=DEV, dtype=torch.int64)
        e.episode_length_buf[: max(N // 8, 4)] = 1249 - torch.arange(max(N // 8, 4), device=DEV) % 60      # time-outs spread over the rollout
    it = torch.tensor([3], dtype=torch.int64, device=DEV)
    ep_idx = torch.tensor([envs[0]._stat_names.index(k[4:]) for k in sorted(envs[0].extras["episode"])], dtype=torch.int32, device=DEV)
    book = [dict(cur_ret=torch.zeros(N, device=DEV), cur_len=torch.zeros(N, device=DEV), fin=torch.zeros(3, device=DEV), ep_acc=torch.zeros(ep_idx.numel(), device=DEV))
            for _ in envs]
    sts = [_storage(N, T), _storage(N, T)]
    n_to = n_done = 0
    for rollout in range(2):
        it.fill_(3 + rollout)
        # A: one launch
        ea, ba, sa = envs[0], book[0], sts[0]
        obs_before = ea.get_observations()
        keep = obs_before.clone()
        lv = torch.full((N,), float("nan"), device=DEV)
        oa = ea.policy_rollout(T, fu.flat, 99, it, sa, gamma, ba["cur_ret"], ba["cur_len"], ba["fin"], ep=(ep_idx, ba["ep_acc"]), last_values=lv)
        assert torch.equal(obs_before, keep)                      # the tensor handed out before stays untouched (two observation buffers)
        # B: one launch per piece and step
        eb, bb, sb = envs[1], book[1], sts[1]
        o = eb.get_observations()
        stale_steps = 0
        for s in range(T):
            act = eb.policy_act(fu.flat, o, 99, it, s, sb)
            o, _, rew, done, infos = eb.step(act)
            _record(L, eb, sb, s, gamma, bb["cur_ret"], bb["cur_len"], bb["fin"], ep_idx, bb["ep_acc"])
            stale_steps += int(done.sum().item() == 0)
        torch.cuda.synchronize()
        for name in ("observations", "actions", "values", "actions_log_prob", "mu", "sigma", "rewards", "dones"):
            assert torch.equal(getattr(sa, name), getattr(sb, name)), (rollout, name, (getattr(sa, name).float() - getattr(sb, name).float()).abs().max())
        assert torch.equal(oa, o) and torch.equal(ea.rew_buf, eb.rew_buf) and torch.equal(ea.reset_buf, eb.reset_buf)
        # the value of the last observation (PPO.compute_returns' last_values), evaluated at the end of the launch: the wave policy's value
        # of that observation bit for bit, the torch critic's to rounding
        tmp = _storage(N, 1)
        eb.policy_act(fu.flat, o, 99, it, 0, tmp)
        assert torch.equal(lv, tmp.values[0].view(-1))
        with torch.no_grad():
            torch.testing.assert_close(lv, ac.evaluate(o).view(-1), atol=3e-5, rtol=1e-5)
        assert torch.equal(ea.episode_length_buf, eb.episode_length_buf) and torch.equal(ea.time_out_buf, eb.time_out_buf)
        for x, y in zip(ea.get_state(), eb.get_state()):
            np.testing.assert_array_equal(x, y)
        ba_, bb_ = ea.get_buffers(), eb.get_buffers()
        for k in ba_:
            np.testing.assert_array_equal(ba_[k], bb_[k], err_msg=k)
        assert torch.equal(ba["cur_ret"], bb["cur_ret"]) and torch.equal(ba["cur_len"], bb["cur_len"])
        torch.testing.assert_close(ba["fin"], bb["fin"], atol=1e-3, rtol=1e-5)                 # float atomics: order of the additions differs
        assert ba["fin"][2] == bb["fin"][2]
        torch.testing.assert_close(ba["ep_acc"], bb["ep_acc"], atol=1e-5, rtol=1e-4)
        torch.testing.assert_close(ea._ep_stats, eb._ep_stats, atol=1e-6, rtol=1e-4)
        assert ea.counters() == eb.counters() and ea.common_step_counter == eb.common_step_counter
        n_done += int(sb.dones.sum())
        boot = (sb.rewards.squeeze(-1) != 0) & False
        n_to += int(eb.time_out_buf.sum())
        if N < 100:
            assert stale_steps > 0, "this population must contain steps without a reset (stale extras)"
    assert n_done >= max(N // 8, 4)
    # and the env keeps working through plain step() afterwards, identically on both sides (time_outs ownership was handed back)
    a = torch.rand(N, 18, device=DEV) * 2 - 1
    for _ in range(3):
        ra, rb = envs[0].step(a), envs[1].step(a)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[2], rb[2]) and torch.equal(ra[3], rb[3]) and ("time_outs" in ra[4]) == send_timeouts
        if send_timeouts:
            assert torch.equal(ra[4]["time_outs"], rb[4]["time_outs"])
    for e in envs:
        e.close()


def test_rollout_refuses_what_it_cannot_do():
    from nightmare_rl_amd import _lib
    ac, fu = _networks()
    env64 = make_env(8, dtype=torch.float64)
    env64.reset()
    it = torch.zeros(1, dtype=torch.int64, device=DEV)
    z = lambda *s: torch.zeros(*s, device=DEV)
    with pytest.raises(_lib.NightmareHipError, match="fp32"):
        env64.policy_rollout(4, fu.flat, 0, it, _storage(8, 4), 0.99, z(8), z(8), z(3))
    env = make_env(8)
    env.reset()
    with pytest.raises(_lib.NightmareHipError, match="more steps than an episode"):
        env.policy_rollout(2000, fu.flat, 0, it, _storage(8, 2000), 0.99, z(8), z(8), z(3))
    L = _lib.load()
    dims = (C.c_int32 * 5)(66, 54, 42, 30, 18)
    cd = (C.c_int32 * 5)(66, 54, 42, 30, 1)
    assert L.nm_rollout_supported(dims, cd, 4) == 1
    wide = (C.c_int32 * 5)(66, 256, 256, 30, 18)
    assert L.nm_rollout_supported(wide, cd, 4) == 0 and L.nm_rollout_supported(dims, cd, 3) == 0
    env.close(); env64.close()


def test_runner_trains_through_the_one_launch_rollout(tmp_path):
    """OnPolicyRunner.learn (reference train.py:54) with the rollout as one launch vs the captured per-step graph: same networks, same
    seeds, same noise keys. The two collect with policy arithmetic that differs in rounding (4x4x1 blocks vs 16x16x4 tiles), so the
    comparison is statistical: first-iteration mean step reward within 1 %, both learn (reward rises over 12 iterations), and the
    one-launch collection is the faster one."""
    from nightmare_rl_amd.envs.helpers import class_to_dict
    from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3ConfigPPO
    from nightmare_rl_amd.rl import OnPolicyRunner
    hist = {}
    for mode in (True, False):
        cfg = class_to_dict(NightmareV3ConfigPPO())
        cfg["runner"]["fused_rollout"] = mode
        torch.manual_seed(0)
        env = make_env(2048, seed=1)
        r = OnPolicyRunner(env, cfg, log_dir=None, device=DEV)
        r.learn(12, init_at_random_ep_len=True)
        assert r.rollout_mode.startswith("one launch") == mode
        hist[mode] = r.history
        assert env.counters()["contacts_dropped"] == 0
        env.close()
    a, b = hist[True], hist[False]
    assert np.isfinite([h["value_loss"] for h in a + b]).all()
    assert abs(a[0]["mean_step_reward"] - b[0]["mean_step_reward"]) < 0.01 * abs(b[0]["mean_step_reward"]), (a[0]["mean_step_reward"], b[0]["mean_step_reward"])
    for h in (a, b):
        assert h[-1]["mean_step_reward"] > h[0]["mean_step_reward"]
    ta, tb = np.median([h["collection_time"] for h in a[3:]]), np.median([h["collection_time"] for h in b[3:]])
    print(f"collection per iteration: one launch {ta * 1e3:.2f} ms, captured per-step graph {tb * 1e3:.2f} ms")
    assert ta < tb


# ------------------------------------------------------------------------------------------------ the update side of the same round
def test_device_permutation_is_a_bijection_and_mixes():
    """nm_ppo_permutation (rsl_rl: torch.randperm in mini_batch_generator): for sizes around powers of two and the real one (4096 envs x 80
    steps) the image of 0..n-1 is 0..n-1; different (seed, counter) give different orders; no structure a mini-batch would inherit
    (the first quarter of the permuted order covers all 80 steps and all envs about evenly)."""
    from nightmare_rl_amd.rl.fused import FusedUpdate
    ac, fu = _networks()
    for n in (1, 2, 3, 17, 255, 256, 257, 4095, 4096, 4097, 100000, 4096 * 80):
        p = fu.permutation(n, seed=12345, counter=7)
        assert p.dtype == torch.int32 and p.numel() == n
        assert torch.equal(torch.sort(p.long()).values, torch.arange(n, device=DEV)), n
    n = 4096 * 80
    p0, p1, p2 = fu.permutation(n, 1, 1), fu.permutation(n, 1, 2), fu.permutation(n, 2, 1)
    assert float((p0 == p1).float().mean()) < 1e-3 and float((p0 == p2).float().mean()) < 1e-3
    assert torch.equal(p0, fu.permutation(n, 1, 1))
    q = p0[: n // 4].long()
    steps, envs = torch.bincount(q // 4096, minlength=80).float(), torch.bincount(q % 4096, minlength=4096).float()
    assert steps.min() > 0.85 * steps.mean() and steps.max() < 1.15 * steps.mean()            # ~1024 per step, sd 28
    assert envs.min() >= 5 and envs.max() <= 45                                               # ~20 per env, sd 3.9
    # consecutive indices do not land next to each other
    d = (p0[1:].long() - p0[:-1].long()).abs().float()
    assert float((d < 64).float().mean()) < 0.002


@pytest.mark.parametrize("shape", ["reference-fast", "reference-generic"])
def test_row_gather_inside_the_kernel_equals_the_gathered_copy_and_the_fused_step_equals_four_launches(shape, monkeypatch):
    """(a) nm_ppo_minibatch_rows on the unpermuted rollout + row numbers == nm_ppo_minibatch on torch-gathered copies: parameters and Adam
    moments bit for bit after 3 steps. (b) the one-launch step (reduce + norm + learning rate + Adam + packing behind one grid barrier) ==
    the four launches (NM_PPO_UNFUSED_STEP=1): same learning rate decisions, parameters to rounding of the gradient-norm sum, and the
    packed weights the next forward reads are the new parameters (the second and third step would diverge otherwise)."""
    import copy
    from nightmare_rl_amd.rl import ActorCritic
    from nightmare_rl_amd.rl.fused import FusedUpdate
    if shape == "reference-generic":
        monkeypatch.setenv("NM_PPO_GENERIC", "1")
    torch.manual_seed(2)
    mk = lambda: ActorCritic(66, 66, 18, actor_hidden_dims=[54, 42, 30], critic_hidden_dims=[54, 42, 30], activation="elu", init_noise_std=0.8).to(DEV)
    ac = mk()
    acs = [ac, copy.deepcopy(ac), copy.deepcopy(ac)]
    fus = []
    for k, a in enumerate(acs):
        if k == 2:
            monkeypatch.setenv("NM_PPO_UNFUSED_STEP", "1")
        fus.append(FusedUpdate(a, torch.optim.Adam(a.parameters(), lr=1e-3), DEV, lr=1e-3))
    monkeypatch.delenv("NM_PPO_UNFUSED_STEP")
    hp = dict(clip=0.2, value_coef=1.0, entropy_coef=0.0015, clip_value=True, desired_kl=0.01, adaptive=True, max_grad_norm=1.0)
    R = 4096 * 6 + 5
    gen = torch.Generator(device=DEV).manual_seed(8)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    obs = rn(R, 66)
    with torch.no_grad():
        old_mu = acs[0].actor(obs) + 0.05 * rn(R, 18)
        old_sigma = (acs[0].std * (1 + 0.05 * rn(18))).expand(R, 18).contiguous()
        actions = old_mu + old_sigma * rn(R, 18)
        old_logp = torch.distributions.Normal(old_mu, old_sigma).log_prob(actions).sum(-1)
        tv = acs[0].critic(obs).squeeze(-1) + 0.3 * rn(R)
    adv, ret = rn(R), tv + rn(R)
    full = (obs, actions, tv, adv, ret, old_logp, old_mu, old_sigma)
    perm = fus[0].permutation(R, 5, 1)
    B = 4096 * 2 + 3
    for step in range(3):
        rows = perm[step * B:(step + 1) * B]
        fus[0].minibatch(*full, hp, rows=rows)
        gathered = tuple(t[rows.long()] for t in full)
        fus[1].minibatch(*gathered, hp)
        fus[2].minibatch(*gathered, hp)
        s0, s1, s2 = (f.read_state() for f in fus)
        assert torch.equal(fus[0].flat, fus[1].flat) and torch.equal(fus[0].m, fus[1].m) and torch.equal(fus[0].v, fus[1].v), step
        assert s0 == s1
        assert s1["lr"] == s2["lr"] and abs(s1["grad_norm"] - s2["grad_norm"]) < 2e-6 * s2["grad_norm"] and abs(s1["kl"] - s2["kl"]) < 1e-6 * s2["kl"] + 1e-9
        torch.testing.assert_close(fus[1].flat, fus[2].flat, atol=1e-7, rtol=2e-6)
        torch.testing.assert_close(fus[1].m, fus[2].m, atol=1e-9, rtol=2e-6)


def test_a_failed_grid_barrier_makes_the_fused_step_a_no_op_and_is_reported():
    """k_ppo_step synchronises its ~236 workgroups with a spinning grid barrier. (a) nm_ppo_create asks the runtime
    whether that grid is co-resident (occupancy x CUs) - on a whole MI355X it is, so the handle takes the fused step. (b) When the barrier
    does time out (forced here by the test hook: the arrival counter is moved out of reach), the step is a NO-OP for every block - no
    parameter, Adam moment or packed weight is written, the Adam step counter does not advance - and stays one until the failure has been
    reported; nm_ppo_get_state then fails loudly, re-arms the barrier and moves the handle to the four-launch step, which works."""
    import copy
    from nightmare_rl_amd import _lib
    from nightmare_rl_amd.rl import ActorCritic
    from nightmare_rl_amd.rl.fused import FusedUpdate
    torch.manual_seed(4)
    ac = ActorCritic(66, 66, 18, actor_hidden_dims=[54, 42, 30], critic_hidden_dims=[54, 42, 30], activation="elu", init_noise_std=0.8).to(DEV)
    ref = copy.deepcopy(ac)
    fu = FusedUpdate(ac, torch.optim.Adam(ac.parameters(), lr=1e-3), DEV, lr=1e-3)
    fr = FusedUpdate(ref, torch.optim.Adam(ref.parameters(), lr=1e-3), DEV, lr=1e-3)
    L = _lib.load()
    assert L.nm_ppo_step_is_fused(fu._h) == 1                      # the occupancy query of nm_ppo_create found room for the whole grid
    hp = dict(clip=0.2, value_coef=1.0, entropy_coef=0.0015, clip_value=True, desired_kl=0.01, adaptive=True, max_grad_norm=1.0)
    R = 4096
    gen = torch.Generator(device=DEV).manual_seed(9)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    obs = rn(R, 66)
    with torch.no_grad():
        old_mu = ac.actor(obs) + 0.05 * rn(R, 18)
        old_sigma = ac.std.expand(R, 18).contiguous()
        actions = old_mu + old_sigma * rn(R, 18)
        old_logp = torch.distributions.Normal(old_mu, old_sigma).log_prob(actions).sum(-1)
        tv = ac.critic(obs).squeeze(-1) + 0.3 * rn(R)
    batch = (obs, actions, tv, rn(R), tv + rn(R), old_logp, old_mu, old_sigma)
    fu.minibatch(*batch, hp)
    fr.minibatch(*batch, hp)
    s_ok = fu.read_state()
    assert torch.equal(fu.flat, fr.flat) and s_ok["steps"] == 1.0
    before = (fu.flat.clone(), fu.m.clone(), fu.v.clone())
    stream = C.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)
    _lib.check(L.nm_ppo_debug_break_barrier(fu._h, stream))
    fu.minibatch(*batch, hp)                                       # its barrier times out: a no-op
    fu.minibatch(*batch, hp)                                       # sticky: returns at the barrier without arriving
    torch.cuda.synchronize()
    assert torch.equal(fu.flat, before[0]) and torch.equal(fu.m, before[1]) and torch.equal(fu.v, before[2])
    snap = torch.zeros(16, device=DEV)
    fu.snapshot_state(snap[:9])
    vals = snap[:9].tolist()
    assert vals[8] != 0.0 and vals[1] == 1.0                       # flagged; Adam's step count did not move
    with pytest.raises(_lib.NightmareHipError, match="grid barrier"):
        fu.state_from(vals)
    fu.minibatch(*batch, hp)                                       # still refused after a snapshot (the snapshot clears only the flag)
    with pytest.raises(_lib.NightmareHipError, match="grid barrier"):
        fu.read_state()
    assert torch.equal(fu.flat, before[0])
    assert L.nm_ppo_step_is_fused(fu._h) == 0                      # reported -> barrier re-armed, handle on the four-launch step
    fu.step_count = 1
    fu.minibatch(*batch, hp)
    fr.minibatch(*batch, hp)
    s2, r2 = fu.read_state(), fr.read_state()
    assert s2["steps"] == r2["steps"] == 2.0 and s2["lr"] == r2["lr"]
    torch.testing.assert_close(fu.flat, fr.flat, atol=1e-7, rtol=2e-6)          # four launches vs one: rounding of the norm sum


def test_learning_curve_at_4096_envs_lies_inside_the_cpu_reference_band():
    """BASELINE config 5's 'return curve vs CPU ref' as an assertion: the runner on the HIP env (fp32 kernels, one-launch rollout, fused
    update) for 25 iterations at 4096 envs against the band the SAME runner reached on the CPU oracle env (fp64, torch PPO; 3 seeds,
    tests/golden/curve_cpu_band_4096.json, made by tests/tools/curve_vs_cpu.py). The two sides differ in env precision, PPO kernels and
    noise generators, so the comparison is statistical: inside the CPU seeds' [min, max] widened by its own width on either side."""
    import json
    import os
    from nightmare_rl_amd.envs.helpers import class_to_dict
    from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3ConfigPPO
    from nightmare_rl_amd.rl import OnPolicyRunner
    band = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "curve_cpu_band_4096.json")))["band"]
    torch.manual_seed(100)
    env = make_env(4096, seed=100)
    r = OnPolicyRunner(env, class_to_dict(NightmareV3ConfigPPO()), log_dir=None, device=DEV)
    r.learn(25, init_at_random_ep_len=True)
    assert r.rollout_mode.startswith("one launch")
    for m in (10, 25):
        b = band[str(m)]
        v = float(np.mean([h["mean_step_reward"] for h in r.history[m - 5:m]]))
        w = b["max"] - b["min"]
        assert b["min"] - w <= v <= b["max"] + w, (m, v, b)
    env.close()
