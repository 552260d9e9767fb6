"""Tibia-vs-tibia contacts (P5: convex-convex through MPR). The reference's env reaches them through MuJoCo's
mjc_Convex; here the oracle restates libccd's MPR and the device code must agree with it."""
import numpy as np
import pytest


def crossing_states(n, rng, z=0.5):
    """Robots with pairs of neighbouring legs swung into each other: airborne (z = 0.5) or, lower, standing on the other legs."""
    qpos = np.zeros((n, 25))
    qpos[:, 2] = z
    qpos[:, 3] = 1.0
    qpos[:, 7:] = np.tile([0.0, -0.6, 0.4], 6) + rng.uniform(-0.1, 0.1, (n, 18))
    pairs = [(0, 1), (1, 2), (3, 4), (4, 5), (0, 1), (4, 5)]
    for i in range(n):
        a, b = pairs[i % len(pairs)]
        ang = 0.62 + 0.25 * rng.uniform()
        qpos[i, 7 + 3 * a] = ang
        qpos[i, 7 + 3 * b] = -ang
    qvel = rng.normal(size=(n, 24)) * 0.2
    return qpos, qvel


def squeeze_actions(n, rng):
    a = rng.uniform(-0.3, 0.3, (n, 18)).astype(np.float32)
    return a


def run_pairwise(make_dev, oracle_mod, double, steps=12, n=12, z=0.5, count_mixed=False):
    rng = np.random.default_rng(11)
    qpos, qvel = crossing_states(n, rng, z)
    nmixed = 0
    dev = make_dev(n, double)
    ora = oracle_mod.OracleEnv(n, seed=0)
    oerr, serr, ncontact = [], 0.0, 0
    qw = np.zeros((n, 24))
    for t in range(steps):
        ora.set_state(qpos, qvel, qw)
        dev.set_state(qpos, qvel, qw)
        b = ora.get_buffers()
        dev.set_buffers(b)
        a = squeeze_actions(n, rng)
        oobs, orew, odone, _ = ora.step(a)
        obs, rew, done = dev.step(a)
        for i in range(n):
            d = ora.data(i)
            npair = sum(1 for c in range(d.ncon) if d.con_body1[c] > 0)
            ncontact += npair
            nmixed += int(npair > 0 and d.ncon > npair)
        oerr.append(np.abs(obs.astype(np.float64) - oobs).max(axis=1))
        assert (done == odone).all()
        qpos, qvel, qw = ora.get_state()
        q2, v2, _ = dev.get_state()
        nd = odone == 0
        serr = max(serr, np.abs(q2[nd] - qpos[nd]).max(), np.abs(v2[nd] - qvel[nd]).max())
    if count_mixed:
        return np.concatenate(oerr), serr, ncontact, nmixed
    return np.concatenate(oerr), serr, ncontact


class EmulDev:
    def __init__(self, n, double):
        from emul import emul as em
        self.e = em.EmulEnv(n, double=double)

    def set_state(self, qpos, qvel, qw):
        self.e.set("qpos", qpos); self.e.set("qvel", qvel); self.e.set("qwarm", qw)

    def set_buffers(self, b):
        self.e.set("dofpos", b["dof_pos"]); self.e.set("dofvel", b["dof_vel"]); self.e.set("act", b["actions"]); self.e.set("cmd", b["commands"])
        self.e.eplen[:] = b["ep_len"]

    def get_state(self):
        return self.e.get("qpos"), self.e.get("qvel"), self.e.get("qwarm")

    def step(self, a):
        obs, rew, done, _ = self.e.step(a)
        return obs, rew, done


def test_oracle_mpr_contact_is_physical(oracle_mod):
    p = oracle_mod.Physics()
    p.qpos[2] = 0.5
    p.qpos[7], p.qpos[10] = 0.65, -0.65          # leg 1 and leg 2 coxae swung into each other
    p.forward()
    tt = [c for c in range(p.ncon) if p.con_body1[c] > 0]
    assert len(tt) == 1 and (p.con_body1[tt[0]], p.con_body[tt[0]]) == (4, 7)
    c = tt[0]
    n = p.con_frame[c][:3]
    assert abs(np.linalg.norm(n) - 1) < 1e-12 and -0.05 < p.con_dist[c] < 0
    assert n @ (p.xipos[7] - p.xipos[4]) > 0                      # normal points from geom1 (tibia 1) to geom2 (tibia 2)
    assert p.qfrc_constraint[6] < 0 < p.qfrc_constraint[9]        # the contact pushes the two coxae apart
    np.testing.assert_allclose(p.qfrc_constraint[:6], 0, atol=1e-12)   # internal force pair: nothing on the base dofs
    assert p.sensordata[0] > 0 and abs(p.sensordata[0] - p.sensordata[1]) < 1e-12   # both tibia sites register it
    for _ in range(40):
        p.step(1)
    assert not [c for c in range(p.ncon) if p.con_body1[c] > 0]   # separated again


def test_device_code_fp64_matches_oracle_with_leg_contacts(oracle_mod):
    oerr, serr, ncontact = run_pairwise(EmulDev, oracle_mod, double=True)
    assert ncontact >= 10, ncontact
    assert oerr.max() < 1e-6 and serr < 1e-8, (oerr.max(), serr)


def test_device_code_fp32_within_tolerance_with_leg_contacts(oracle_mod):
    oerr, serr, ncontact = run_pairwise(EmulDev, oracle_mod, double=False, steps=12, n=48)
    assert ncontact >= 40
    assert oerr.max() <= 1e-4 and np.median(oerr) < 5e-6, (np.median(oerr), oerr.max())      # no allowance: MPR runs in fp64 in both builds


def test_device_code_with_floor_and_leg_contacts_in_the_same_env(oracle_mod):
    """Tibia-tibia contacts in an env that also stands on the floor: the pair contacts follow the floor contacts in the list, and the
    floor contacts' frame / first body are only filled in (floor_frames) when such a general path runs. fp64 device code == oracle."""
    oerr, serr, ncontact, nmixed = run_pairwise(EmulDev, oracle_mod, double=True, steps=10, n=24, z=0.07, count_mixed=True)
    assert nmixed >= 10, (ncontact, nmixed)
    assert oerr.max() < 1e-6 and serr < 1e-8, (oerr.max(), serr)
