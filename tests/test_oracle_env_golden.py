"""The oracle's env layer (oracle/nm_oracle_env.c) against golden vectors produced by the
reference's own NightmareV3Env class (tests/golden/make_goldens.py)."""
import numpy as np
import pytest

from conftest import load_golden

SCENARIOS = ["env_reset_rollout.npz", "env_timeouts.npz", "env_falls.npz", "env_noise.npz",
             "env_allrewards.npz", "env_modes22.npz", "env_modes01.npz",    # every reward function in the table; contact modes 2 and 0
             "env_manycontacts.npz"]                                       # more than 16 simultaneous contacts


def golden_config(g):
    """kwargs of OracleEnv.configure for a fixture generated with a non-default reference config (None = defaults)."""
    if "cfg_modes" not in g:
        return None
    m = g["cfg_modes"]
    return dict(reward_scales=dict(zip([str(n) for n in g["cfg_reward_names"]], g["cfg_reward_scales"])),
                tibia_contact_mode=int(m[0]), tibia_max_contact_force=float(m[1]), body_contact_mode=int(m[2]),
                body_max_contact_force=float(m[3]), base_height_target=float(m[4]), max_contact_force=float(m[5]))


def replay(orc, g, check):
    N = g["actions"].shape[1]
    env = orc.OracleEnv(N, seed=0)
    kw = golden_config(g)
    if kw is not None:
        env.configure(**kw)
    env.reset_idx(None, cmd_u=g["reset_u"])
    env.set_state(g["init_qpos"], g["init_qvel"], g["init_qacc_warmstart"])
    env.set_buffers(dof_pos=g["init_dof_pos"], dof_vel=g["init_dof_vel"], commands=g["init_commands"], ep_len=g["init_ep_len"])
    noisy = "noise_u" in g and g["noise_u"].shape[1] > 0
    for t in range(g["actions"].shape[0]):
        if noisy:   # the uniforms np.random.rand(N, 66) returned inside the reference step (env.py:305)
            env.set_noise(g["noise_scale_vec"], g["noise_u"][t])
        obs, rew, done, to = env.step(g["actions"][t], cmd_u=g["cmd_u"][t])
        check(t, env, obs, rew, done, to)
    return env


@pytest.mark.parametrize("name", SCENARIOS)
def test_env_layer_matches_reference(oracle_mod, name):
    g = load_golden(name)

    def check(t, env, obs, rew, done, to):
        np.testing.assert_array_equal(done, g["done"][t], err_msg=f"done t={t}")
        np.testing.assert_array_equal(to, g["time_outs"][t], err_msg=f"time_outs t={t}")
        np.testing.assert_allclose(obs, g["obs"][t], rtol=0, atol=2e-7, err_msg=f"obs t={t}")
        np.testing.assert_allclose(rew, g["rew"][t], rtol=1e-6, atol=1e-7, err_msg=f"rew t={t}")
        b = env.get_buffers()
        np.testing.assert_allclose(b["commands"], g["commands"][t], atol=1e-15)
        np.testing.assert_array_equal(b["ep_len"], g["ep_len"][t])
        names = [str(n) for n in g["reward_names"]]                      # the reference's table holds the non-zero scales only
        rows = [oracle_mod.REW_NAMES.index(k) for k in names]
        off = [i for i in range(oracle_mod.NREW) if i not in rows]
        np.testing.assert_allclose(b["episode_sums"][rows], g["episode_sums"][t], rtol=5e-7, atol=1e-12)  # action_rate is a float32 sum upstream
        assert (b["episode_sums"][off] == 0).all()
        dbg = env.debug()
        for k in ("base_lin_vel", "base_ang_vel", "projected_gravity", "tibia", "feet", "body"):
            np.testing.assert_allclose(dbg[k], g[k][t], rtol=0, atol=1e-12, err_msg=f"{k} t={t}")
        qpos, qvel, qw = env.get_state()
        np.testing.assert_allclose(qpos, g["qpos"][t], atol=1e-13)
        np.testing.assert_allclose(qvel, g["qvel"][t], atol=1e-12)
        if "feet_air_time" in g:
            air, last, filt = env.get_feet_state()
            np.testing.assert_allclose(air, g["feet_air_time"][t], atol=1e-12)
            np.testing.assert_array_equal(last, g["last_contacts"][t])
            np.testing.assert_array_equal(filt, g["last_contacts_filt"][t])
        n, stats = env.episode_stats()
        assert n == g["nreset"][t]
        if n:
            np.testing.assert_allclose(stats[rows], g["ep_stats"][t], rtol=1e-6, atol=1e-9)

    replay(oracle_mod, g, check)


def test_reference_constants(oracle_mod):
    g = load_golden("env_reset_rollout.npz")
    assert list(g["reward_names"]) == ["action_rate", "body_contact_forces", "default_position", "dof_acc", "orientation",
                                       "termination", "tracking_ang_vel", "tracking_lin_vel"]
    assert float(g["max_episode_length"]) == 1250.0
    assert abs(float(g["dt"]) - 0.016) < 1e-15
    scales = dict(zip(g["reward_names"], g["reward_scales"]))
    assert abs(scales["termination"] + 3.2) < 1e-12 and abs(scales["tracking_lin_vel"] - 0.128) < 1e-12


def test_noise_fixture_is_noisy_and_uses_the_upstream_index_ranges():
    g = load_golden("env_noise.npz")
    v = g["noise_scale_vec"]
    assert v.shape == (66,) and (v[12:36] > 0).all() and (v[36:] == 0).all() and (v[9:12] == 0).all()   # env.py:113-119
    assert g["noise_u"].shape[1:] == (5, 66) and 0 <= g["noise_u"].min() and g["noise_u"].max() < 1


def test_reward_fixtures_exercise_every_function_and_mode():
    g = load_golden("env_allrewards.npz")
    names = [str(n) for n in g["reward_names"]]
    assert sorted(names) == sorted(["action_rate", "ang_vel_xy", "base_height", "body_contact_forces", "default_position", "dof_acc", "dof_vel",
                                    "feet_air_time", "feet_contact_forces", "lin_vel_z", "orientation", "stand_still", "termination", "torques",
                                    "tracking_ang_vel", "tracking_lin_vel"])
    es = g["episode_sums"]                                   # [T, names, N]
    moved = {n: np.abs(np.diff(es[:, i], axis=0)).max() > 0 for i, n in enumerate(names)}
    assert all(moved[n] for n in names if n != "torques"), moved          # every term contributed; torques is identically 0 upstream
    assert not moved["torques"]
    assert g["last_contacts"].any() and (g["feet_air_time"] > 0.1).any() and (np.diff(g["last_contacts_filt"].astype(int), axis=0) != 0).any()
    assert g["nreset"].sum() >= 2
    m = load_golden("env_modes22.npz")
    # terminations that only contact mode 2 produces: no time-out, feet below 160 N, tilt below 60 degrees
    pg = m["projected_gravity"]
    tilt = np.arccos(-pg[..., 2] / np.linalg.norm(pg, axis=-1))
    only_mode2 = (m["done"] == 1) & (m["time_outs"] == 0) & (m["feet"].max(axis=2) <= 160) & (tilt <= np.pi / 3)
    assert only_mode2.sum() >= 2
    z = load_golden("env_modes01.npz")
    assert "tracking_ang_vel" not in [str(n) for n in z["reward_names"]] and "dof_vel" in [str(n) for n in z["reward_names"]]


def test_many_contact_fixture_exceeds_the_resident_solver():
    g = load_golden("env_manycontacts.npz")
    assert (g["ncon"] > 16).sum() >= 10 and g["ncon"].max() >= 20     # 16 = rows the register-resident solver of the HIP path holds / 4


def test_golden_covers_edge_cases():
    """The fixtures exercise every branch of E4-E6: timeouts, periodic resample, tilt and force terminations."""
    g = load_golden("env_timeouts.npz")
    assert g["time_outs"].sum() >= 3 and (np.abs(np.diff(g["commands"], axis=0)).sum() > 0)
    f = load_golden("env_falls.npz")
    assert f["done"].sum() >= 4
    assert (f["feet"].max(axis=2) > 160).any(), "no hard-landing termination in fixture"
    pg = f["projected_gravity"]
    tilt = np.arccos(-pg[..., 2] / np.linalg.norm(pg, axis=-1))
    assert (tilt > np.pi / 3).any(), "no tilt termination in fixture"
