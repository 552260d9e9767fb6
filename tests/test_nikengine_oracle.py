"""The gait/IK oracle (oracle/nik_oracle.py) against golden vectors produced by the reference's own nikengine package."""
import numpy as np

from conftest import load_golden


def _g():
    return load_golden("nikengine.npz")


def test_constants_match_reference_config():
    from oracle import nik_oracle as nk
    g = _g()
    np.testing.assert_allclose(nk.DEFAULT_POSE, g["cfg_default_pose"], atol=1e-15)
    np.testing.assert_allclose(nk.SIT_POSE, g["cfg_sit_pose"], atol=1e-15)
    np.testing.assert_allclose(nk.POSE_OFFSET, g["cfg_pose_offset"], atol=1e-15)
    np.testing.assert_allclose(nk.REL_CONVERT, g["cfg_rel_convert"])
    np.testing.assert_allclose(nk.SERVO_OFFSET, g["cfg_servo_offset"], atol=1e-15)
    np.testing.assert_allclose(nk.URDF_OFFSETS, g["cfg_urdf_offsets"])
    np.testing.assert_allclose(nk.DIM, g["cfg_dim"])


def test_helpers_match_reference():
    from oracle import nik_oracle as nk
    g = _g()
    np.testing.assert_allclose([nk.seg_seg_dist(*s) for s in g["seg_in"]], g["seg_out"], atol=1e-15)
    assert (g["seg_out"] == 0).sum() > 10                     # intersecting pairs are in the fixture
    np.testing.assert_allclose(nk.sigmoid(g["sig_in"]), g["sig_out"], rtol=1e-14)
    np.testing.assert_allclose([nk.bezier_point(t, list(g["bez_pts"])) for t in g["bez_t"]], g["bez_out"], atol=1e-15)


def test_relative_ik_including_unreachable_targets():
    from oracle import nik_oracle as nk
    g = _g()
    out = np.array([nk.relative_ik(p) for p in g["ik_in"]])
    np.testing.assert_allclose(out, g["ik_out"], atol=1e-12)
    assert np.isfinite(out).all()


def _replay(fps, inp, ts, walk=True):
    from oracle import nik_oracle as nk
    e = nk.Engine(engine_fps=fps)
    return np.array([e.update(l, a, awake=bool(s), walk=walk, now=t) for (l, a, s), t in zip(inp, ts)])


def test_walk_rollout_all_fsm_phases():
    g = _g()
    out = _replay(float(g["walk_fps"]), g["walk_in"], g["walk_t"])
    np.testing.assert_allclose(out, g["walk_out"], atol=1e-10)
    assert np.abs(np.diff(g["walk_out"][400:], axis=0)).max() > 1e-3      # it is actually walking at the end


def test_varying_commands_trigger_keepout_search():
    g = _g()
    out = _replay(float(g["var_fps"]), g["var_in"], g["var_t"])
    np.testing.assert_allclose(out, g["var_out"], atol=1e-10)


def test_stand_mode():
    g = _g()
    out = _replay(51.0, g["stand_in"], g["stand_t"], walk=False)
    np.testing.assert_allclose(out, g["stand_out"], atol=1e-10)


def _replay_gaits(out_ref, gait_sel, cmd, fps):
    from oracle import nik_oracle as nk
    e = nk.Engine(engine_fps=fps)
    names = ["tripod", "ripple", "wave"]
    res = []
    for k in range(len(out_ref)):
        e.gait_cmd = names[int(gait_sel[k])]
        res.append(e.update(cmd[0], cmd[1], awake=True, walk=True, now=k / fps))
    return np.array(res)


def test_ripple_and_wave_gaits_and_a_gait_change_while_walking():
    g = load_golden("nikengine_gaits.npz")
    fps = float(g["fps"])
    for name in ("ripple", "wave"):
        out = _replay_gaits(g[name + "_out"], g[name + "_gait"], g["cmd_fixed"], fps)
        np.testing.assert_allclose(out, g[name + "_out"], atol=1e-10, err_msg=name)
    out = _replay_gaits(g["switch_out"], g["switch_gait"], g["cmd_switch"], fps)
    np.testing.assert_allclose(out, g["switch_out"], atol=1e-10)
    assert np.abs(g["ripple_out"][400:] - g["wave_out"][400:]).max() > 1e-2       # the gaits really differ
