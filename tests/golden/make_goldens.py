"""Generate golden vectors by running the REFERENCE's own NightmareV3Env class.

Runs only in the build container (needs /root/reference); the outputs (tests/golden/*.npz) are
committed and are what travels to the GPU box. Nothing from the reference's source is stored:
the fixtures are inputs (actions, uniforms, initial states) and outputs (obs, rewards, dones,
buffers) only.

`mujoco` (third-party, absent here) is replaced by a stub module whose MjData is backed by this
repo's CPU oracle (oracle/), so the fixtures pin the reference's *env logic* (E1-E9:
envs/nightmare_v3_env.py:145-371) exactly as written upstream, running on the oracle's physics.
`mju_negQuat` / `mju_rotVecQuat` are restated with their documented math.

Usage: python tests/golden/make_goldens.py
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import oracle as orc  # noqa: E402

import ctypes as C  # noqa: E402


# ------------------------------------------------------------------ stub mujoco backed by the oracle
class _Opt:
    timestep = 0.008


class StubModel:
    nv = 24
    ngeom = 20
    opt = _Opt()

    def __init__(self):
        q = np.zeros(25)
        q[2] = 0.15
        q[3] = 1.0
        self.qpos0 = q

    @staticmethod
    def from_xml_path(path):
        return StubModel()


class StubData:
    def __init__(self, model):
        self._d = orc.NmoData()
        self._s = SCRATCH
        orc.lib().nmo_reset_data(C.byref(self._d))
        self.qfrc_applied = np.zeros(24)
        self.act = np.zeros(0)

    def _np(self, name):
        return np.ctypeslib.as_array(getattr(self._d, name))

    qpos = property(lambda self: self._np("qpos"), lambda self, v: self._np("qpos").__setitem__(slice(None), v))
    qvel = property(lambda self: self._np("qvel"), lambda self, v: self._np("qvel").__setitem__(slice(None), v))
    ctrl = property(lambda self: self._np("ctrl"), lambda self, v: self._np("ctrl").__setitem__(slice(None), v))
    cvel = property(lambda self: self._np("cvel"))
    xipos = property(lambda self: self._np("xipos"))
    sensordata = property(lambda self: self._np("sensordata"))
    qacc_warmstart = property(lambda self: self._np("qacc_warmstart"))
    time = property(lambda self: self._d.time)


def _mj_step(model, data, nstep=1):
    orc.lib().nmo_step(C.byref(data._d), C.byref(data._s), nstep)


def _negQuat(res, q):
    res[0], res[1], res[2], res[3] = q[0], -q[1], -q[2], -q[3]


def _rotVecQuat(res, vec, quat):
    w, x, y, z = quat
    v = np.asarray(vec, dtype=np.float64)
    t = 2 * np.cross([x, y, z], v)
    res[:] = v + w * t + np.cross([x, y, z], t)


def install_stub():
    mj = types.ModuleType("mujoco")
    mj.MjModel = StubModel
    mj.MjData = StubData
    mj.mj_step = _mj_step
    mj.mju_negQuat = _negQuat
    mj.mju_rotVecQuat = _rotVecQuat
    mj.mj_name2id = lambda model, objtype, name: 1
    mj.mjtObj = types.SimpleNamespace(mjOBJ_BODY=1)
    viewer = types.ModuleType("mujoco.viewer")
    mj.viewer = viewer
    sys.modules["mujoco"] = mj
    sys.modules["mujoco.viewer"] = viewer


SCRATCH = orc.NmoScratch()


# ------------------------------------------------------------------ recording harness
class RandRecorder:
    """Wraps np.random.rand to log what _resample_commands drew (reference env.py:327,330)."""

    def __init__(self):
        self.calls = []
        self._orig = np.random.rand

    def __enter__(self):
        def rec(*shape):
            out = self._orig(*shape)
            self.calls.append(np.array(out, copy=True))
            return out

        np.random.rand = rec
        return self

    def __exit__(self, *a):
        np.random.rand = self._orig


def run_scenario(name, N, steps, seed, setup=None, action_fn=None, noise=False, cfg_fn=None, post_fn=None):
    import torch
    from envs.nightmare_v3_config import NightmareV3Config
    from envs.nightmare_v3_env import NightmareV3Env

    cfg = NightmareV3Config()
    cfg.env.num_envs = N
    cfg.viewer.render = False
    cfg.viewer.record_states = False
    cfg.noise.add_noise = noise
    if cfg_fn is not None:
        cfg_fn(cfg)
    np.random.seed(seed)
    env = NightmareV3Env(cfg, log_dir="/tmp/nm_golden_logs", num_threads=1)
    rng = np.random.default_rng(seed + 1000)
    log = {k: [] for k in ("actions", "obs", "rew", "done", "time_outs", "commands", "ep_len", "cmd_u", "qpos", "qvel", "qacc_warmstart",
                           "base_lin_vel", "base_ang_vel", "projected_gravity", "tibia", "feet", "body", "dof_pos", "dof_vel",
                           "episode_sums", "ep_stats", "nreset", "noise_u", "feet_air_time", "last_contacts", "last_contacts_filt", "base_heights", "ncon")}
    # reset() = reset_idx(all) + step(zeros)   (env.py:392-396)
    with RandRecorder() as rr:
        env.reset_idx(np.arange(N))
    reset_u = np.stack([rr.calls[0], rr.calls[1]], axis=1)  # [N,2]
    if setup is not None:
        setup(env)
    init = dict(qpos=np.stack([d.qpos.copy() for d in env.data]), qvel=np.stack([d.qvel.copy() for d in env.data]),
                qacc_warmstart=np.stack([d.qacc_warmstart.copy() for d in env.data]),
                ep_len=env.episode_length_buf.numpy().copy(), commands=env.commands.copy(),
                dof_pos=env.dof_pos.copy(), dof_vel=env.dof_vel.copy(), feet_air_time=env.feet_air_time.copy(),
                last_contacts=np.array(env.last_contacts, np.uint8), last_contacts_filt=np.array(env.last_contacts_filt, np.uint8))
    names = list(env.episode_sums.keys())
    for t in range(steps):
        a = np.zeros((N, 18), np.float32) if (t == 0 and action_fn is None and name == "reset_rollout") else None
        if a is None:
            a = action_fn(t, rng, N) if action_fn else rng.uniform(-1, 1, (N, 18)).astype(np.float32) * 5.0
        ep_before = env.episode_length_buf.numpy().copy()
        with RandRecorder() as rr:
            obs, _, rew, done, extras = env.step(torch.tensor(a))
        noise_calls = [c for c in rr.calls if np.ndim(c) == 2]          # np.random.rand(N, 66) at env.py:305
        rr.calls = [c for c in rr.calls if np.ndim(c) != 2]
        assert len(noise_calls) == (1 if noise else 0)
        log["noise_u"].append(noise_calls[0] if noise else np.zeros((0, 66)))
        # reconstruct which envs drew what: periodic resample first (always called), then reset_idx (only if any reset)
        cu = np.full((N, 4), 0.5)
        per_ids = np.nonzero((ep_before + 1) % 625 == 0)[0]
        assert len(rr.calls[0]) == len(per_ids)
        cu[per_ids, 0], cu[per_ids, 1] = rr.calls[0], rr.calls[1]
        rst_ids = np.nonzero(done.numpy())[0]
        if len(rst_ids):
            assert len(rr.calls) == 4 and len(rr.calls[2]) == len(rst_ids)
            cu[rst_ids, 2], cu[rst_ids, 3] = rr.calls[2], rr.calls[3]
        else:
            assert len(rr.calls) == 2
        log["actions"].append(a)
        log["obs"].append(obs.numpy().copy())
        log["rew"].append(rew.numpy().copy())
        log["done"].append(done.numpy().copy())
        log["time_outs"].append(env.time_out_buf.astype(np.float32).copy())
        log["commands"].append(env.commands.copy())
        log["ep_len"].append(env.episode_length_buf.numpy().copy())
        log["cmd_u"].append(cu)
        log["qpos"].append(np.stack([d.qpos.copy() for d in env.data]))
        log["qvel"].append(np.stack([d.qvel.copy() for d in env.data]))
        log["qacc_warmstart"].append(np.stack([d.qacc_warmstart.copy() for d in env.data]))
        for k, attr in (("base_lin_vel", "base_lin_vel"), ("base_ang_vel", "base_ang_vel"), ("projected_gravity", "projected_gravity"),
                        ("tibia", "tibia_contact_forces"), ("feet", "feet_contact_forces"), ("body", "body_contact_force"),
                        ("dof_pos", "dof_pos"), ("dof_vel", "dof_vel")):
            log[k].append(np.array(getattr(env, attr), dtype=np.float64, copy=True))
        log["feet_air_time"].append(env.feet_air_time.copy())
        log["last_contacts"].append(np.array(env.last_contacts, np.uint8))
        log["last_contacts_filt"].append(np.array(env.last_contacts_filt, np.uint8))
        log["base_heights"].append(env.base_heights.copy())
        log["ncon"].append(np.array([d._d.ncon for d in env.data]))       # contacts of the step's last forward pass (diagnostic)
        log["episode_sums"].append(np.stack([env.episode_sums[k].copy() for k in names]))
        if len(rst_ids):
            log["ep_stats"].append(np.array([float(extras["episode"]["rew_" + k]) for k in names]))
        else:
            log["ep_stats"].append(np.full(len(names), np.nan))
        log["nreset"].append(len(rst_ids))
    out = {k: np.stack(v) for k, v in log.items()}
    out["reset_u"] = reset_u
    out["reward_names"] = np.array(names)
    out["reward_scales"] = np.array([env.reward_scales[k] for k in names])
    out["max_episode_length"] = np.array(env.max_episode_length)
    out["dt"] = np.array(env.dt)
    out["noise_scale_vec"] = env.noise_scale_vec
    # the configuration the scenario ran with (raw config values; the tests configure the oracle / the HIP env from these)
    from envs.helpers import class_to_dict
    raw = class_to_dict(cfg.rewards.scales)
    out["cfg_reward_names"] = np.array(list(raw.keys()))
    out["cfg_reward_scales"] = np.array([float(raw[k]) for k in raw])
    out["cfg_modes"] = np.array([cfg.env.tibia_contact_mode, cfg.env.tibia_max_contact_force, cfg.env.body_contact_mode,
                                 cfg.env.body_max_contact_force, cfg.rewards.base_height_target, cfg.rewards.max_contact_force], dtype=np.float64)
    for k, v in init.items():
        out["init_" + k] = v
    if post_fn is not None:
        post_fn(env, out)
    path = os.path.join(ROOT, "tests", "golden", f"env_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "steps", steps, "N", N, "resets", int(out["nreset"].sum()), "timeouts", int(out["time_outs"].sum()),
          "->", os.path.relpath(path, ROOT), os.path.getsize(path) // 1024, "KiB")


def main():
    install_stub()
    sys.path.insert(0, REF)
    os.chdir(REF)

    # (a) reset + random-action rollout through first ground contact
    run_scenario("reset_rollout", N=6, steps=120, seed=1)

    # (b) timeouts and the periodic command resample: episode lengths placed just before 625 / 1250
    def setup_timeouts(env):
        import torch
        env.episode_length_buf = torch.tensor([620, 621, 1246, 1247, 1248, 100, 623, 1249], dtype=torch.int64)

    run_scenario("timeouts", N=8, steps=12, seed=2, setup=setup_timeouts,
                 action_fn=lambda t, rng, N: rng.uniform(-1, 1, (N, 18)).astype(np.float32))

    # (c) tilt termination (> 60 deg) and hard landings (foot force > 160 N): tilted / dropped initial states
    def setup_falls(env):
        for i, d in enumerate(env.data):
            ang = [0.9, 1.2, 2.0, 3.0, 0.0, 0.0, 0.0, 0.5][i]
            ax = np.array([[1, 0, 0], [0, 1, 0], [1, 1, 0], [1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, 0]][i], dtype=float)
            ax /= np.linalg.norm(ax)
            q = d.qpos.copy()
            q[3] = np.cos(ang / 2)
            q[4:7] = np.sin(ang / 2) * ax
            q[2] = [0.25, 0.25, 0.3, 0.3, 0.6, 0.9, 0.4, 0.5][i]
            if i in (4, 5, 6):  # legs pointing down so a foot lands first
                q[7:] = np.tile([0.0, -0.9, 0.6], 6)
            d.qpos = q
            v = np.zeros(24)
            if i in (4, 5):
                v[2] = -4.0
            d.qvel = v

    run_scenario("falls", N=8, steps=60, seed=3, setup=setup_falls,
                 action_fn=lambda t, rng, N: rng.uniform(-1, 1, (N, 18)).astype(np.float32) * 3.0)


def main_noise():
    """(d) observation noise on (cfg.noise.add_noise): the uniforms np.random.rand returned are part of the fixture."""
    install_stub()
    sys.path.insert(0, REF)
    os.chdir(REF)
    run_scenario("noise", N=5, steps=40, seed=4, noise=True,
                 action_fn=lambda t, rng, N: rng.uniform(-1, 1, (N, 18)).astype(np.float32) * 2.0)


def _setup_mixed(env):
    """Upright, tilted (terminate at once), dropped and belly-down robots; zero commands for some (stand_still)."""
    for i, d in enumerate(env.data):
        ang = [0.0, 0.0, 1.3, 0.0, 0.7, 0.0, 2.2, 0.3][i]
        ax = np.array([[1, 0, 0], [0, 1, 0], [1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 0]][i], dtype=float)
        ax /= np.linalg.norm(ax)
        q = d.qpos.copy()
        q[3] = np.cos(ang / 2)
        q[4:7] = np.sin(ang / 2) * ax
        q[2] = [0.15, 0.2, 0.25, 0.5, 0.2, 0.12, 0.3, 0.18][i]
        if i in (3, 5):   # legs pointing down: feet take the load
            q[7:] = np.tile([0.0, -0.9, 0.6], 6)
        if i == 1:        # legs folded up: the belly lands
            q[7:] = np.tile([0.0, 0.9, -0.3], 6)
        d.qpos = q
        v = np.zeros(24)
        if i == 3:
            v[2] = -3.0
        d.qvel = v
    env.commands[0:3] = 0.0


def _many_contact_states(n_want=8, seed=1):
    """Initial states from which a robot lying on folded legs shows more than 16 simultaneous contacts while its servos hold the
    pose (found by simulating 256 random folded poses with the oracle; > 16 contacts is what the register-resident solver of the
    HIP path cannot hold, so these exercise its matrix-free path)."""
    rng = np.random.default_rng(seed)
    N = 256
    o = orc.OracleEnv(N, seed=seed, num_threads=8)
    q = np.zeros((N, 25))
    q[:, 2] = rng.uniform(0.08, 0.2, N)
    for i in range(N):
        ang = rng.uniform(0, 0.4)
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        q[i, 3] = np.cos(ang / 2)
        q[i, 4:7] = np.sin(ang / 2) * ax
        leg = [rng.uniform(-0.5, 0.5), rng.uniform(-1.5, -0.8), rng.uniform(0.3, 1.5)]
        q[i, 7:] = np.tile(leg, 6) + rng.normal(size=18) * 0.05
    o.set_state(q, np.zeros((N, 24)), np.zeros((N, 24)))
    o.set_buffers(dof_pos=q[:, 7:])
    default = np.tile([0, np.pi / 5, 0], 6)
    count = np.zeros(N, int)
    for t in range(80):
        a = np.clip((o.get_buffers()["dof_pos"] + default) / 0.2, -5, 5).astype(np.float32)
        o.step(a)
        count += np.array([o.data(i).ncon for i in range(N)]) > 16
    pick = np.argsort(-count, kind="stable")[:n_want]
    assert (count[pick] > 0).sum() >= 4, count[pick]
    return q[pick]


def main_manycontacts():
    """(h) more than 16 simultaneous contacts (robot lying on folded legs, servos holding the pose)."""
    install_stub()
    sys.path.insert(0, REF)
    os.chdir(REF)
    q0 = _many_contact_states()
    holder = {}

    def setup(env):
        holder["env"] = env
        for i, d in enumerate(env.data):
            d.qpos = q0[i]
            d.qvel = np.zeros(24)
        env.dof_pos[:] = q0[:, 7:]

    def act(t, rng, N):
        env = holder["env"]
        return np.clip((env.dof_pos + env.default_dof_pos) / 0.2, -5, 5).astype(np.float32)

    run_scenario("manycontacts", N=8, steps=80, seed=9, setup=setup, action_fn=act)


def main_statelog():
    """(i) cfg.viewer.record_states (env.py:261-272): the pickle the reference class writes when env 0 resets, stored as arrays
    (time, qpos, qvel, act of every record), plus class_to_dict of both reference config classes as JSON (helpers.py:3-18)."""
    import glob
    import json
    import pickle
    import shutil
    install_stub()
    sys.path.insert(0, REF)
    os.chdir(REF)
    shutil.rmtree("/tmp/nm_golden_logs", ignore_errors=True)

    def cfg_fn(cfg):
        cfg.viewer.record_states = True

    def setup(env):
        import torch
        env.episode_length_buf = torch.tensor([1238, 300], dtype=torch.int64)

    def post(env, out):
        files = sorted(glob.glob("/tmp/nm_golden_logs/*.pkl"))
        assert len(files) == 1, files
        with open(files[0], "rb") as f:
            rec = pickle.load(f)
        assert isinstance(rec, list) and all(isinstance(r, tuple) and len(r) == 4 for r in rec)
        out["log_time"] = np.array([r[0] for r in rec])
        out["log_qpos"] = np.stack([r[1] for r in rec])
        out["log_qvel"] = np.stack([r[2] for r in rec])
        out["log_act_size"] = np.array([np.asarray(r[3]).size for r in rec])
        out["log_types"] = np.array([type(rec).__name__, type(rec[0]).__name__, type(rec[0][0]).__name__, type(rec[0][1]).__name__,
                                     str(rec[0][1].dtype), str(rec[0][1].shape), str(rec[0][2].shape), str(np.asarray(rec[0][3]).shape)])
        pend = env.recorded_states           # what has been logged since the dump (the next file's beginning)
        out["pending_time"] = np.array([r[0] for r in pend])
        out["pending_qpos"] = np.stack([r[1] for r in pend])

    run_scenario("statelog", N=2, steps=20, seed=10, setup=setup, cfg_fn=cfg_fn, post_fn=post,
                 action_fn=lambda t, rng, N: rng.uniform(-1, 1, (N, 18)).astype(np.float32))
    from envs.helpers import class_to_dict
    from envs.nightmare_v3_config import NightmareV3Config, NightmareV3ConfigPPO
    dump = {"NightmareV3Config": class_to_dict(NightmareV3Config()), "NightmareV3ConfigPPO": class_to_dict(NightmareV3ConfigPPO())}
    with open(os.path.join(ROOT, "tests", "golden", "config_class_to_dict.json"), "w") as f:
        json.dump(dump, f, indent=1, sort_keys=False)
    print("config_class_to_dict.json", {k: len(v) for k, v in dump.items()})


def main_rewards():
    """(e) every reward function of the reference in the table (the scales config.py:88-96 keeps as comments), (f)/(g) contact
    modes 2 (terminate on tibia / body contact) and 0."""
    install_stub()
    sys.path.insert(0, REF)
    os.chdir(REF)

    def all_rewards(cfg):
        sc = cfg.rewards.scales
        sc.lin_vel_z, sc.ang_vel_xy, sc.feet_air_time, sc.torques = -2.0, -5.0, -4.0, -0.00001
        sc.base_height, sc.feet_contact_forces, sc.dof_vel, sc.stand_still = -2000.0, -0.05, -0.001, -1.0

    def act(t, rng, N):   # slow sweeps + noise so that feet lift off and land (feet_air_time changes state)
        ph = 0.35 * t + np.arange(N)[:, None] * 0.7 + np.arange(18)[None, :] * 1.1
        return (3.0 * np.sin(ph) + rng.uniform(-1, 1, (N, 18))).astype(np.float32)

    run_scenario("allrewards", N=8, steps=140, seed=6, setup=_setup_mixed, action_fn=act, cfg_fn=all_rewards)

    def modes22(cfg):
        cfg.env.tibia_contact_mode = 2
        cfg.env.body_contact_mode = 2
        cfg.rewards.scales.feet_air_time = -4.0

    run_scenario("modes22", N=8, steps=60, seed=7, setup=_setup_mixed, action_fn=act, cfg_fn=modes22)

    def modes01(cfg):
        cfg.env.tibia_contact_mode = 0
        cfg.env.body_contact_mode = 1
        cfg.rewards.scales.tracking_ang_vel = 0       # a default term dropped from the table
        cfg.rewards.scales.dof_vel = -0.001

    run_scenario("modes01", N=8, steps=40, seed=8, setup=_setup_mixed, action_fn=act, cfg_fn=modes01)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "noise":
        main_noise()
    elif len(sys.argv) > 1 and sys.argv[1] == "rewards":
        main_rewards()
    elif len(sys.argv) > 1 and sys.argv[1] == "manycontacts":
        main_manycontacts()
    elif len(sys.argv) > 1 and sys.argv[1] == "statelog":
        main_statelog()
    else:
        main()
