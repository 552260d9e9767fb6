"""C-ABI surface and host logic (no GPU needed: nothing here launches a kernel)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib_path():
    import __graft_entry__ as ge
    ge.build()
    from nightmare_rl_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    return _lib.LIB_PATH


def test_library_exports_every_declared_symbol(hip_lib_path):
    hdr = open(os.path.join(ROOT, "include", "nightmare_hip.h")).read()
    declared = set(re.findall(r"\b(nm_[a-z_0-9]+)\s*\(", hdr))
    L = ctypes.CDLL(hip_lib_path)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    from nightmare_rl_amd import _lib
    assert set(_lib.EXPORTS) <= declared


def test_default_config_matches_reference_defaults(hip_lib_path):
    from nightmare_rl_amd import _lib
    L = _lib.load()
    c = _lib.NmConfig()
    L.nm_default_config(ctypes.byref(c))
    assert c.decimation == 2 and c.p_gain == 20 and abs(c.action_scale - 0.2) < 1e-15
    assert abs(c.default_pos[1] - np.pi / 5) < 1e-15 and c.termination_contact_force == 160
    # every reward name of the reference config with a _reward_ function (env.py:399-497): alphabetical, termination last
    names = _lib.reward_names()
    assert names == ["action_rate", "ang_vel_xy", "base_height", "body_contact_forces", "default_position", "dof_acc", "dof_vel",
                     "feet_air_time", "feet_contact_forces", "lin_vel_z", "orientation", "stand_still", "torques", "tracking_ang_vel",
                     "tracking_lin_vel", "termination"]
    from nightmare_rl_amd.envs.helpers import class_to_dict
    from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
    cfg = NightmareV3Config()
    ref = class_to_dict(cfg.rewards.scales)
    assert dict(zip(names, c.reward_scales)) == {n: float(ref[n]) for n in names}
    assert sorted(set(ref) - set(names)) == ["collision", "feet_stumble"]       # config names with no function upstream (config.py:95-96)
    assert (c.tibia_contact_mode, c.tibia_max_contact_force, c.body_contact_mode, c.body_max_contact_force) == (1, 2.0, 1, 2.0)
    assert (c.base_height_target, c.max_contact_force) == (cfg.rewards.base_height_target, cfg.rewards.max_contact_force) == (0.1, 10.0)


def test_create_fails_loudly_without_gpu(hip_lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nightmare_rl_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    rc = L.nm_create(None, 4, 0, 0, 0, 0, ctypes.byref(h))
    assert rc != 0 and not h.value
    assert b"no HIP device" in L.nm_last_error() or b"hip" in L.nm_last_error().lower()
    from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config
    from nightmare_rl_amd.envs.nightmare_v3_env import NightmareV3Env
    cfg = NightmareV3Config()
    cfg.env.num_envs = 4
    with pytest.raises(_lib.NightmareHipError):
        NightmareV3Env(cfg)


def test_config_tree_matches_reference_surface():
    from nightmare_rl_amd.envs.helpers import class_to_dict
    from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config, NightmareV3ConfigPPO
    cfg = NightmareV3Config()
    assert cfg.env.num_envs == 8192 and cfg.env.num_obs == 66 and cfg.env.num_actions == 18
    cfg.env.num_envs = 16                      # instance attribute: must not leak into the class
    assert NightmareV3Config().env.num_envs == 8192
    assert cfg.control.decimation == 2 and len(cfg.control.default_pos) == 18
    scales = class_to_dict(cfg.rewards.scales)
    assert list(scales) == sorted(scales)      # dir() order = alphabetical (fixes reward evaluation order)
    nz = [k for k, v in scales.items() if v != 0]
    assert nz == ["action_rate", "body_contact_forces", "default_position", "dof_acc", "orientation", "termination",
                  "tracking_ang_vel", "tracking_lin_vel"]
    t = class_to_dict(NightmareV3ConfigPPO())
    assert t["policy"]["actor_hidden_dims"] == [54, 42, 30] and t["algorithm"]["num_mini_batches"] == 4
    assert t["runner"]["num_steps_per_env"] == 80 and t["seed"] == 1


def test_class_to_dict_matches_the_reference_dump():
    """tests/golden/config_class_to_dict.json = class_to_dict(...) of the reference's own config classes (train.py:37 hands it to the
    runner; env.py:80 builds the reward table from it): same nesting, same keys in the same (dir) order, same values - except the
    three defaults this backend changes on purpose (device, viewer.render, viewer.record_states)."""
    import json
    from nightmare_rl_amd.envs.helpers import class_to_dict
    from nightmare_rl_amd.envs.nightmare_v3_config import NightmareV3Config, NightmareV3ConfigPPO
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "config_class_to_dict.json")))
    ours = {"NightmareV3Config": class_to_dict(NightmareV3Config()), "NightmareV3ConfigPPO": class_to_dict(NightmareV3ConfigPPO())}
    assert ours["NightmareV3Config"].pop("device") == "cuda" and ref["NightmareV3Config"].pop("device") == "cpu"
    assert ours["NightmareV3Config"]["viewer"] == {"record_states": False, "render": False}
    assert ref["NightmareV3Config"]["viewer"] == {"record_states": True, "render": True}
    ours["NightmareV3Config"]["viewer"] = ref["NightmareV3Config"]["viewer"]

    def same(a, b, path=""):
        assert type(a) is type(b) or (isinstance(a, (int, float)) and isinstance(b, (int, float))), (path, a, b)
        if isinstance(a, dict):
            assert list(a) == list(b), (path, list(a), list(b))      # key order matters: it is the reward evaluation order
            for k in a:
                same(a[k], b[k], path + "." + k)
        elif isinstance(a, list):
            assert len(a) == len(b), path
            for i, (x, y) in enumerate(zip(a, b)):
                same(x, y, f"{path}[{i}]")
        else:
            assert a == b, (path, a, b)

    same(ours, ref)


def test_golden_config_agrees_with_goldens():
    """reward scale * dt and episode constants the reference computed at construction (fixtures) vs host constants."""
    from conftest import load_golden
    g = load_golden("env_reset_rollout.npz")
    ours = dict(action_rate=-0.02, body_contact_forces=-5.0, default_position=-0.01, dof_acc=-2.5e-5, orientation=-5.0,
                tracking_ang_vel=6.0, tracking_lin_vel=8.0, termination=-200.0)
    for n, s in zip(g["reward_names"], g["reward_scales"]):
        assert abs(ours[str(n)] * 0.016 - s) < 1e-15


def test_get_load_path(tmp_path):
    from nightmare_rl_amd.envs.helpers import get_load_path
    run = tmp_path / "2024-03-10 02:54:19"
    run.mkdir()
    for it in (50, 100, 1000):
        (run / f"model_{it}.pt").write_bytes(b"")
    assert get_load_path(str(tmp_path)).endswith("model_1000.pt")
    assert get_load_path(str(tmp_path), checkpoint=50).endswith("model_50.pt")
