"""Batched gait/IK kernel (csrc/nm_nik.hip through the C-ABI) against the reference-generated golden vectors and the oracle."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-9          # f64 kernel vs numpy: libm differences only


def _replay(eng, nk, fps, inp, ts, walk="walk"):
    nk.config.ENGINE_FPS = fps
    out = []
    for (l, a, s), t in zip(inp, ts):
        nk.set_time_s(t)
        out.append(eng.update(float(l), float(a), "awake" if s else "idle", walk).copy())
    return np.array(out)


def test_reference_goldens_scalar_engine():
    import torch
    from nightmare_rl_amd import nikengine as nk
    g = load_golden("nikengine.npz")
    for key, mode in (("walk", "walk"), ("var", "walk"), ("stand", "stand")):
        eng = nk.EngineNode(1, dtype=torch.float64)
        fps = float(g[key + "_fps"]) if key + "_fps" in g else 51.0
        out = _replay(eng, nk, fps, g[key + "_in"], g[key + "_t"], mode)
        np.testing.assert_allclose(out, g[key + "_out"], atol=TOL, err_msg=key)
    nk.config.ENGINE_FPS = 51.0


def test_ik_table_through_the_kernel_matches_oracle_batch():
    """Many envs with different commands, flags and reset times vs one oracle engine per env."""
    import torch
    from nightmare_rl_amd import nikengine as nk
    from oracle import nik_oracle as no
    N, T, fps = 37, 420, 62.5
    rng = np.random.default_rng(5)
    nk.config.ENGINE_FPS = fps
    eng = nk.EngineNode(N, dtype=torch.float64)
    ref = [no.Engine(fps) for _ in range(N)]
    lin = rng.uniform(-0.25, 0.25, N).astype(np.float32)
    ang = rng.uniform(-1.2, 1.2, N).astype(np.float32)
    awake = np.ones(N, bool)
    walk = np.ones(N, bool)
    worst = 0.0
    seen = set()
    for t in range(T):
        now = t / fps
        if t % 40 == 0:
            lin = rng.uniform(-0.25, 0.25, N).astype(np.float32)
            ang = rng.uniform(-1.2, 1.2, N).astype(np.float32)
        if t == 300:
            awake[::5] = False           # some go back to sleep, some switch to stand
            walk[1::5] = False
        if t == 250:
            ids = np.array([3, 11, 36])
            eng.reset(ids)
            for i in ids:
                ref[i] = no.Engine(fps)
        out = eng.update(torch.as_tensor(lin).cuda(), torch.as_tensor(ang).cuda(), torch.as_tensor(awake), torch.as_tensor(walk), time_s=now).cpu().numpy()
        exp = np.array([ref[i].update(float(lin[i]), float(ang[i]), bool(awake[i]), bool(walk[i]), now) for i in range(N)])
        worst = max(worst, np.abs(out - exp).max())
        st = eng.get_state()
        assert (st["fsm"] == [r.fsm for r in ref]).all(), t
        seen |= set(st["fsm"].tolist())
    assert worst < TOL, worst
    assert {0, 1, 2, 5, 6} <= seen
    nk.config.ENGINE_FPS = 51.0


def test_ripple_and_wave_gaits_and_a_gait_change_while_walking():
    """Reference-generated golden (make_nik_goldens.py gaits): one env per schedule in ONE batched engine, so the per-env gait state
    is exercised too (env 0 ripple, env 1 wave, env 2 untouched tripod as the control against the plain walk golden)."""
    import torch
    from nightmare_rl_amd import nikengine as nk
    g = load_golden("nikengine_gaits.npz")
    fps = float(g["fps"])
    nk.config.ENGINE_FPS = fps
    eng = nk.EngineNode(2, dtype=torch.float64)
    eng.set_gait("ripple", [0])
    eng.set_gait("wave", [1])
    lin = torch.full((2,), float(g["cmd_fixed"][0]), dtype=torch.float64)
    ang = torch.full((2,), float(g["cmd_fixed"][1]), dtype=torch.float64)
    n = len(g["ripple_out"])
    out = np.array([eng.update(lin, ang, time_s=k / fps).cpu().numpy() for k in range(n)])
    np.testing.assert_allclose(out[:, 0], g["ripple_out"], atol=TOL)
    np.testing.assert_allclose(out[:, 1], g["wave_out"], atol=TOL)
    eng = nk.EngineNode(1, dtype=torch.float64)
    sel, names, res = g["switch_gait"], nk.GAIT_NAMES, []
    for k in range(len(sel)):
        if k == 0 or sel[k] != sel[k - 1]:
            eng.set_gait(names[int(sel[k])])
        nk.set_time_s(k / fps)
        res.append(eng.update(float(g["cmd_switch"][0]), float(g["cmd_switch"][1])).copy())
    np.testing.assert_allclose(np.array(res), g["switch_out"], atol=TOL)
    with pytest.raises(ValueError):
        eng.set_gait("gallop")
    nk.config.ENGINE_FPS = 51.0


def test_float32_output_and_errors():
    import torch
    from nightmare_rl_amd import nikengine as nk, _lib
    e = nk.EngineNode(8)
    out = e.update(0.0, 0.0, "idle", time_s=0.0)
    assert out.dtype == torch.float32 and out.shape == (8, 18) and torch.isfinite(out).all()
    with pytest.raises(_lib.NightmareHipError):
        e.reset(np.array([9]))
    with pytest.raises(_lib.NightmareHipError):
        nk.EngineNode(1, device="cpu")


def test_gait_engine_walks_the_simulated_robot():
    """custom_play.py:49-76 end to end on the device: gait kernel -> rate limit -> env servo -> rigid-body/contact kernel.
    The robot must get up to the stand height and cover about 0.2 m/s x walking time without any termination."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))
    from custom_play import play
    from nightmare_rl_amd import nikengine as nk
    r = play(num_envs=32, seconds=10.0, lin=0.05, ang=0.0)
    nk.config.ENGINE_FPS = 51.0
    dist = np.linalg.norm(r["displacement"][:, :2], axis=1)
    assert r["falls"] == 0
    assert (r["fsm"] == 6).all()
    assert 0.07 < r["height"].min() and r["height"].max() < 0.13, (r["height"].min(), r["height"].max())
    assert 0.6 < dist.min() and dist.max() < 1.5, (dist.min(), dist.max())       # 5.5 s of walking at ~0.2 m/s
