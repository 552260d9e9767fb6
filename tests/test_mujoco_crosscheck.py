"""Teacher-forced single-step cross-check of the oracle's physics against a REAL MuJoCo, wherever one is importable.

MuJoCo (pinned <= 3.1.2 by the reference's requirements.txt:2) is not installable in the build container or on the GPU box, so
this test skips there and the physics stays "parity unpinned" (oracle/nm_oracle.h). It exists so that the first environment which
does have `mujoco` and the reference's models/nightmare_v3/mjmodel.xml pins - or refutes - the restatement automatically."""
import os

import numpy as np
import pytest

mujoco = pytest.importorskip("mujoco")
XML = os.environ.get("NM_REFERENCE_XML", "/root/reference/models/nightmare_v3/mjmodel.xml")
pytestmark = pytest.mark.skipif(not os.path.exists(XML), reason="reference model description not present")


def test_model_constants_match_the_compiled_tables():
    from nightmare_rl_amd.model.compile_model import load_tables
    T = load_tables()
    m = mujoco.MjModel.from_xml_path(XML)
    np.testing.assert_allclose(m.body_mass, T["body_mass"], rtol=1e-6)
    np.testing.assert_allclose(m.body_inertia, T["body_inertia"], rtol=1e-5)
    np.testing.assert_allclose(m.body_ipos, T["body_ipos"], atol=1e-7)
    np.testing.assert_allclose(m.body_invweight0, T["body_invweight0"], rtol=1e-5)
    assert abs(m.stat.meaninertia - float(T["meaninertia"])) < 1e-6 * m.stat.meaninertia


def test_single_steps_match_mujoco(oracle_mod):
    m = mujoco.MjModel.from_xml_path(XML)
    d = mujoco.MjData(m)
    p = oracle_mod.Physics()
    rng = np.random.default_rng(0)
    ora = oracle_mod.OracleEnv(16, seed=1)
    ora.reset()
    for t in range(60):                                  # states along a random-action rollout of the oracle, through touch-down
        a = rng.uniform(-1, 1, (16, 18)).astype(np.float32)
        q, v, w = ora.get_state()
        import parity_tools as pt
        ctrl = [pt.servo_ctrl(a[i], ora.get_buffers()["dof_pos"][i]) for i in range(16)]    # what step() hands to the physics (E1)
        ora.step(a)
        i = t % 16
        mujoco.mj_resetData(m, d)
        d.qpos[:], d.qvel[:], d.qacc_warmstart[:], d.ctrl[:] = q[i], v[i], w[i], ctrl[i]
        p.reset()
        p.qpos[:], p.qvel[:], p.qacc_warmstart[:], p.ctrl[:] = q[i], v[i], w[i], ctrl[i]
        mujoco.mj_step(m, d)
        p.step(1)
        assert d.ncon == p.d.ncon, t
        np.testing.assert_allclose(p.qpos, d.qpos, atol=1e-9)
        np.testing.assert_allclose(p.qvel, d.qvel, atol=1e-7)
        np.testing.assert_allclose(p.sensordata, d.sensordata, atol=1e-6)
