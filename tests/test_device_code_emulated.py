"""The DEVICE kernel source (nightmare_rl_amd/csrc/nm_core.h) compiled for the host as a 64-lane lockstep SIMT
emulation (tests/emul) and checked against the reference-generated goldens. This is what can be verified about
the HIP path in a container without a GPU; the `-m gpu` tests repeat it on the real kernel through the C ABI."""
import numpy as np
import pytest

from conftest import load_golden

SCENARIOS = ["env_reset_rollout.npz", "env_timeouts.npz", "env_falls.npz", "env_noise.npz",
             "env_allrewards.npz", "env_modes22.npz", "env_modes01.npz",    # every reward function in the table; contact modes 2 and 0
             "env_manycontacts.npz"]                                       # more than 16 simultaneous contacts


def teacher_forced(make_env, g, steps=None):
    """Load the golden pre-step state each step, step once, return per-env max |obs error|, |rew error| and flags."""
    N, T = g["actions"].shape[1], g["actions"].shape[0] if steps is None else steps
    e = make_env(N)
    from test_oracle_env_golden import golden_config
    kw = golden_config(g)
    if kw is not None:
        e.configure(**kw)
    feet = (g["init_feet_air_time"], g["init_last_contacts"], g["init_last_contacts_filt"]) if "feet_air_time" in g else None
    qpos, qvel, qw = g["init_qpos"], g["init_qvel"], g["init_qacc_warmstart"]
    dofpos, dofvel, cmd, ep = g["init_dof_pos"], g["init_dof_vel"], g["init_commands"], g["init_ep_len"]
    act = np.zeros((N, 18))
    oerr, rerr, flags, state = [], [], 0, []
    for t in range(T):
        e.set("qpos", qpos); e.set("qvel", qvel); e.set("qwarm", qw); e.set("dofpos", dofpos); e.set("dofvel", dofvel)
        e.set("cmd", cmd); e.set("act", act)
        e.eplen[:] = ep
        if feet is not None:
            e.set_feet_state(*feet)
        if "noise_u" in g and g["noise_u"].shape[1] > 0:
            e.set_noise(g["noise_scale_vec"], g["noise_u"][t])
        obs, rew, done, to = e.step(g["actions"][t], cmd_u=g["cmd_u"][t])
        oerr.append(np.abs(obs.astype(np.float64) - g["obs"][t]).max(axis=1))
        rerr.append(np.abs(rew.astype(np.float64) - g["rew"][t]))
        flags += int((done != g["done"][t]).sum()) + int((to != g["time_outs"][t]).sum())
        state.append(max(np.abs(e.get("qpos") - g["qpos"][t]).max(), np.abs(e.get("qvel") - g["qvel"][t]).max()))
        np.testing.assert_array_equal(e.eplen, g["ep_len"][t])
        if feet is not None and kw["reward_scales"].get("feet_air_time", 0) != 0:
            air, last, filt = e.get_feet_state()
            np.testing.assert_allclose(air, g["feet_air_time"][t], atol=1e-5)
            np.testing.assert_array_equal(last, g["last_contacts"][t])
            np.testing.assert_array_equal(filt, g["last_contacts_filt"][t])
        if feet is not None:
            feet = (g["feet_air_time"][t], g["last_contacts"][t], g["last_contacts_filt"][t])
        qpos, qvel, qw = g["qpos"][t], g["qvel"][t], g["qacc_warmstart"][t]
        dofpos, dofvel, cmd, ep = g["dof_pos"][t], g["dof_vel"][t], g["commands"][t], g["ep_len"][t]
        act = np.clip(g["actions"][t].astype(np.float32) * np.float32(0.2), -1, 1).astype(np.float64)
    return np.concatenate(oerr), np.concatenate(rerr), flags, max(state)


@pytest.fixture(scope="module")
def emul():
    from emul import emul as em
    em.build()
    return em


@pytest.mark.parametrize("name", SCENARIOS)
def test_fp64_device_algorithm_is_exact(emul, name):
    g = load_golden(name)
    oerr, rerr, flags, serr = teacher_forced(lambda N: emul.EmulEnv(N, double=True), g)
    assert flags == 0
    # float32 rounding of the returned tensors only: two fp64 values that agree to 1e-12 can still round to neighbouring floats, so a
    # reward may be off by ONE float32 ulp of its magnitude (2.4e-7 for the |r| > 2 of a termination step), never more
    ulp = np.spacing(np.abs(g["rew"][:len(rerr) // g["rew"].shape[1]]).astype(np.float32)).astype(np.float64).reshape(-1)
    assert oerr.max() <= 1e-7 and (rerr <= np.maximum(2e-7, ulp)).all(), (oerr.max(), rerr.max())
    assert serr < 1e-9


@pytest.mark.parametrize("name", SCENARIOS)
def test_fp32_device_algorithm_within_tolerance(emul, oracle_mod, name):
    """Stated tolerance (BASELINE north_star): obs/reward within 1e-4 of the reference path, teacher-forced single step.
    An env-step above it is accepted only the way the -m gpu tests accept it (tests/test_gpu_parity.py): the ORACLE must show a discrete
    collision decision (support-vertex tie on a flat foot, contact on / off, extra contact on / off, the 0.3 rbound rule) within 2e-7 m of
    flipping at that very pre-step state, the error must stay under the bound of that kind of event, and such steps must stay rare."""
    import parity_tools as pt
    g = load_golden(name)
    oerr, rerr, flags, _ = teacher_forced(lambda N: emul.EmulEnv(N, double=False), g)
    assert flags == 0
    N = g["actions"].shape[1]
    bounds = {"tie": 2e-2, "nbr": 2e-2, "tol": 2e-2, "act": 1e-1}
    outliers = np.nonzero(oerr > 1e-4)[0]
    for k in outliers:
        t, i = divmod(int(k), N)
        pre = {n: (g["init_" + n][i] if t == 0 else g[n][t - 1][i]) for n in ("qpos", "qvel", "qacc_warmstart", "dof_pos")}
        m, kind, npair = pt.discrete_margin(oracle_mod, pre["qpos"], pre["qvel"], pre["qacc_warmstart"], pt.servo_ctrl(g["actions"][t][i], pre["dof_pos"]))
        assert npair == 0 and m < 2e-7 and oerr[k] <= bounds.get(kind, 0.0), \
            f"{name} step {t} env {i}: error {oerr[k]:.2e} above 1e-4 at a state whose nearest discrete decision is {m:.2e} m away ({kind})"
    assert len(outliers) <= max(2, len(oerr) // 200), (len(outliers), len(oerr))
    assert np.percentile(oerr, 99) < 2e-5 and np.median(oerr) < 2e-6
    assert np.percentile(rerr, 99) < 2e-5


def test_free_run_tracks_oracle(emul, oracle_mod):
    """Free-running fp32 device code vs the fp64 oracle from reset, same actions, internal counter RNG on both sides."""
    N, T = 8, 25
    rng = np.random.default_rng(5)
    e = emul.EmulEnv(N, double=False, seed=7)
    o = oracle_mod.OracleEnv(N, seed=7)
    for t in range(T):
        a = rng.uniform(-1, 1, (N, 18)).astype(np.float32)
        obs, rew, done, to = e.step(a)
        oobs, orew, odone, oto = o.step(a)
        np.testing.assert_array_equal(done, odone)
        np.testing.assert_allclose(obs, oobs, atol=5e-4)   # airborne + first contact: errors still at rounding level
        np.testing.assert_allclose(rew, orew, atol=5e-4)


def test_any_vertex_is_a_valid_warm_start_of_the_hull_search(emul, oracle_mod):
    """The support search starts from last step's vertex; force it to start from the widest fan centres of every hull (34
    neighbours, never a support vertex in practice): results must still equal the oracle's, step after step."""
    from nightmare_rl_amd.model.compile_model import load_tables
    T = load_tables()
    deg = (T["hull_nbr"] >= 0).sum(1)
    wide = []
    for g in range(7):
        va, nv = int(T["col_vadr"][g]), int(T["col_nvert"][g])
        wide.append(int(np.argmax(deg[va:va + nv])))
        assert deg[va + wide[-1]] > 15
    N, rng = 6, np.random.default_rng(3)
    e = emul.EmulEnv(N, double=True, seed=2)
    o = oracle_mod.OracleEnv(N, seed=2)
    for t in range(25):                       # fall from the reset height onto the floor
        a = rng.uniform(-1, 1, (N, 18)).astype(np.float32)
        if t in (0, 12, 20):
            e.hull_cache(np.tile(np.array(wide + [0], np.int32), (N, 1)))
        obs, rew, done, _ = e.step(a)
        oobs, orew, odone, _ = o.step(a)
        np.testing.assert_array_equal(done, odone)
        np.testing.assert_allclose(obs, oobs, atol=2e-7)
        np.testing.assert_allclose(e.get("qvel"), o.get_state()[1], atol=1e-9)
    assert max(o.data(i).ncon for i in range(N)) > 0
    hc = e.hull_cache()
    assert not any((hc[:, g] == wide[g]).all() for g in range(1, 7))       # the search moved away from the forced start


def test_counter_rng_matches_oracle(emul, oracle_mod):
    """Command resampling draws: device code and oracle share the (seed, global env id, counter) generator."""
    N = 5
    e = emul.EmulEnv(N, double=True, seed=123, env_off=40)
    o = oracle_mod.OracleEnv(N, seed=123, env_id_offset=40)
    e.eplen[:] = 624
    o.set_buffers(ep_len=np.full(N, 624))
    a = np.zeros((N, 18), np.float32)
    e.step(a)
    o.step(a)
    np.testing.assert_allclose(e.get("cmd"), o.get_buffers()["commands"], atol=1e-15)
    assert np.abs(e.get("cmd")).sum() > 0


def test_envs_per_wave_does_not_change_results(emul):
    """Two envs per wavefront vs one env per wave: bitwise identical, odd N padded. With two envs per wave the load, the epilogue and
    (when both envs have 1..8 floor contacts) the constraint stage handle both envs in one pass on half-waves: same arithmetic, same
    order - the counter proves that the two-env constraint pass really ran."""
    N, T = 7, 60
    rng = np.random.default_rng(8)
    e1 = emul.EmulEnv(N, double=False, seed=3, envs_per_wave=1)
    e2 = emul.EmulEnv(N, double=False, seed=3, envs_per_wave=2)
    before = emul.lib().emu_together_count()
    for t in range(T):
        a = rng.uniform(-1, 1, (N, 18)).astype(np.float32)
        o1 = e1.step(a)
        o2 = e2.step(a)
        for x, y in zip(o1, o2):
            np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(e1.get("qpos"), e2.get("qpos"))
    assert emul.lib().emu_together_count() - before > 50


def test_batched_floor_contacts_equal_the_mesh_by_mesh_emission(emul, monkeypatch):
    """From three touching meshes on, lone support-vertex contacts (certified by the table's per-vertex slack) leave stage B in one batched
    pass, mesh g on lane 57 + g. NM_COLLIDE_BATCH_MIN (read when an env is created) = 99 never batches, = 1 batches whenever the
    certificate holds: the two must agree bitwise, on a standing robot (small actions: 5-6 feet down) and on a flailing one."""
    N, T = 6, 70
    for scale, seed in ((0.12, 5), (1.0, 6)):
        rng = np.random.default_rng(seed)
        monkeypatch.setenv("NM_COLLIDE_BATCH_MIN", "99")
        e1 = emul.EmulEnv(N, double=False, seed=4, envs_per_wave=2)
        monkeypatch.setenv("NM_COLLIDE_BATCH_MIN", "1")
        e2 = emul.EmulEnv(N, double=False, seed=4, envs_per_wave=2)
        monkeypatch.delenv("NM_COLLIDE_BATCH_MIN")
        e3 = emul.EmulEnv(N, double=False, seed=4, envs_per_wave=1)          # the shipped threshold, one env per wave
        most = 0
        for t in range(T):
            a = (rng.uniform(-1, 1, (N, 18)) * scale).astype(np.float32)
            o1, o2, o3 = e1.step(a), e2.step(a, want_dbg=True), e3.step(a)
            for x, y, z in zip(o1, o2, o3):
                np.testing.assert_array_equal(x, y)
                np.testing.assert_array_equal(x, z)
            most += int((e2.dbg[:, 160] >= 3).sum())      # envs whose last forward pass had three or more contacts
        for k in ("qpos", "qvel", "qwarm"):
            np.testing.assert_array_equal(e1.get(k), e2.get(k))
            np.testing.assert_array_equal(e1.get(k), e3.get(k))
        assert most > (200 if scale < 1 else 20), most     # the batched pass had work to do


def test_noise_generator_matches_oracle(emul, oracle_mod):
    """Counter-RNG observation noise (no injected uniforms): device algorithm == oracle, keyed by global env id and step."""
    N, off = 4, 1000
    vec = np.linspace(0.01, 0.5, 66)
    e = emul.EmulEnv(N, double=True, seed=7, env_off=off)
    o = oracle_mod.OracleEnv(N, seed=7, env_id_offset=off)
    o.reset_idx(None)
    e.set("cmd", o.get_buffers()["commands"])
    e.set_noise(vec)
    o.set_noise(vec)
    rng = np.random.default_rng(0)
    clean = None
    for t in range(6):
        a = rng.uniform(-1, 1, (N, 18)).astype(np.float32)
        obs_e = e.step(a)[0]
        obs_o = o.step(a)[0]
        np.testing.assert_allclose(obs_e, obs_o, atol=2e-7)
    o2 = oracle_mod.OracleEnv(N, seed=7, env_id_offset=off)     # same run without noise: the difference is the noise
    o2.reset_idx(None)
    rng = np.random.default_rng(0)
    for t in range(6):
        clean = o2.step(rng.uniform(-1, 1, (N, 18)).astype(np.float32))[0]
    d = (obs_o - clean) / vec
    assert np.abs(d).max() <= 1.0 + 1e-5 and np.abs(d).std() > 0.2


def test_state_record_is_the_pre_reset_state(emul):
    """env.py:261-272 logs data[0] after the physics and before reset_idx."""
    g = load_golden("env_falls.npz")
    N = g["actions"].shape[1]
    for env_i in (1, 2):
        e = emul.EmulEnv(N, double=True)
        e.record(env_i)
        e.set("qpos", g["init_qpos"]); e.set("qvel", g["init_qvel"]); e.set("qwarm", g["init_qacc_warmstart"])
        e.set("dofpos", g["init_dof_pos"]); e.set("dofvel", g["init_dof_vel"]); e.set("cmd", g["init_commands"])
        e.eplen[:] = g["init_ep_len"]
        seen_reset = False
        for t in range(g["actions"].shape[0]):
            _, _, done, _ = e.step(g["actions"][t], cmd_u=g["cmd_u"][t])
            qpos, qvel, nbad = e.record(env_i)
            if done[env_i]:
                seen_reset = True
                assert np.abs(e.get("qvel")[env_i]).max() == 0 and np.abs(qvel).max() > 0      # log has the terminal state
            else:
                np.testing.assert_allclose(qpos, e.get("qpos")[env_i], atol=0)
                np.testing.assert_allclose(qvel, e.get("qvel")[env_i], atol=0)
            assert nbad == 0
        assert seen_reset
