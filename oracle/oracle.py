"""ctypes binding of the CPU oracle (oracle/libnm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (nightmare_rl_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnm_oracle.so")

NBODY, NV, NQ, NU, NSENS, MAXCON = 20, 24, 25, 18, 13, 48
MAXEFC = 4 * MAXCON
NREW = 16
# every reward name of the reference config with a _reward_ function behind it (env.py:399-497): alphabetical, termination last
REW_NAMES = ["action_rate", "ang_vel_xy", "base_height", "body_contact_forces", "default_position", "dof_acc", "dof_vel", "feet_air_time",
             "feet_contact_forces", "lin_vel_z", "orientation", "stand_still", "torques", "tracking_ang_vel", "tracking_lin_vel", "termination"]
DEFAULT_SCALES = dict(termination=-200.0, tracking_lin_vel=8.0, tracking_ang_vel=6.0, dof_acc=-2.5e-5, action_rate=-0.02,
                      body_contact_forces=-5.0, default_position=-0.01, orientation=-5.0)   # config.py:78-86; all others 0

d_ = C.c_double


class NmoData(C.Structure):
    _fields_ = [
        ("qpos", d_ * NQ), ("qvel", d_ * NV), ("qacc_warmstart", d_ * NV), ("ctrl", d_ * NU), ("time", d_),
        ("xpos", d_ * 3 * NBODY), ("xquat", d_ * 4 * NBODY), ("xmat", d_ * 9 * NBODY),
        ("xipos", d_ * 3 * NBODY), ("ximat", d_ * 9 * NBODY),
        ("xanchor", d_ * 3 * NU), ("xaxis", d_ * 3 * NU),
        ("subtree_com", d_ * 3),
        ("cinert", d_ * 10 * NBODY), ("cdof", d_ * 6 * NV),
        ("qM", d_ * NV * NV), ("qLD", d_ * NV * NV), ("qLDiagInv", d_ * NV),
        ("cvel", d_ * 6 * NBODY), ("cdof_dot", d_ * 6 * NV), ("qfrc_bias", d_ * NV),
        ("qfrc_actuator", d_ * NV), ("qfrc_smooth", d_ * NV), ("qacc_smooth", d_ * NV),
        ("qfrc_constraint", d_ * NV), ("qacc", d_ * NV),
        ("ncon", C.c_int32), ("nefc", C.c_int32),
        ("con_pos", d_ * 3 * MAXCON), ("con_frame", d_ * 9 * MAXCON), ("con_dist", d_ * MAXCON),
        ("con_body", C.c_int32 * MAXCON), ("con_body1", C.c_int32 * MAXCON), ("con_geom", C.c_int32 * MAXCON),
        ("efc_force", d_ * MAXEFC),
        ("sensordata", d_ * NSENS),
        ("solver_niter", C.c_int32), ("noslip_niter", C.c_int32), ("nwarning", C.c_int32), ("pad", C.c_int32),
    ]

    def np(self, name):
        return np.ctypeslib.as_array(getattr(self, name))


class NmoScratch(C.Structure):
    _fields_ = [
        ("J", d_ * NV * MAXEFC), ("JM2", d_ * NV * MAXEFC),
        ("pos", d_ * MAXEFC), ("diagApprox", d_ * MAXEFC), ("R", d_ * MAXEFC), ("D", d_ * MAXEFC),
        ("K", d_ * MAXEFC), ("B", d_ * MAXEFC), ("imp", d_ * MAXEFC),
        ("vel", d_ * MAXEFC), ("aref", d_ * MAXEFC), ("b", d_ * MAXEFC), ("jar", d_ * MAXEFC),
        ("AR", d_ * (MAXEFC * MAXEFC)),
    ]

    def np(self, name):
        return np.ctypeslib.as_array(getattr(self, name))


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        assert L.nmo_sizeof_data() == C.sizeof(NmoData), (L.nmo_sizeof_data(), C.sizeof(NmoData))
        assert L.nmo_sizeof_scratch() == C.sizeof(NmoScratch)
        P = C.POINTER
        L.nmo_reset_data.argtypes = [P(NmoData)]
        L.nmo_forward.argtypes = [P(NmoData), P(NmoScratch)]
        L.nmo_step.argtypes = [P(NmoData), P(NmoScratch), C.c_int]
        L.nmo_set_collide_self.argtypes = [C.c_int]
        L.nmo_env_create.restype = C.c_void_p
        L.nmo_env_create.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int]
        L.nmo_env_destroy.argtypes = [C.c_void_p]
        L.nmo_env_reset_idx.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.nmo_env_step.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.nmo_env_step_physics.argtypes = [C.c_void_p, C.c_void_p]
        L.nmo_env_set_noise.argtypes = [C.c_void_p] * 3
        L.nmo_env_configure.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double]
        L.nmo_env_get_feet_state.argtypes = [C.c_void_p] * 4
        L.nmo_env_set_feet_state.argtypes = [C.c_void_p] * 4
        L.nmo_env_get_state.argtypes = [C.c_void_p] * 4
        L.nmo_env_set_state.argtypes = [C.c_void_p] * 4
        L.nmo_env_get_buffers.argtypes = [C.c_void_p] * 7
        L.nmo_env_set_buffers.argtypes = [C.c_void_p] * 7
        L.nmo_env_episode_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.nmo_env_episode_stats.restype = C.c_int
        L.nmo_env_data.argtypes = [C.c_void_p, C.c_int]
        L.nmo_env_data.restype = P(NmoData)
        L.nmo_env_get_debug.argtypes = [C.c_void_p] * 8
        L.nmo_rand_u24.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.nmo_rand_u24.restype = C.c_double
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Physics:
    """One MjData-equivalent + scratch (mj_forward / mj_step for tests)."""

    def __init__(self):
        self.L = lib()
        self.d = NmoData()
        self.s = NmoScratch()
        self.L.nmo_reset_data(C.byref(self.d))

    def reset(self):
        self.L.nmo_reset_data(C.byref(self.d))

    def forward(self):
        self.L.nmo_forward(C.byref(self.d), C.byref(self.s))

    def step(self, n=1):
        self.L.nmo_step(C.byref(self.d), C.byref(self.s), n)

    def __getattr__(self, name):
        d = object.__getattribute__(self, "d")
        if any(name == f[0] for f in NmoData._fields_):
            v = getattr(d, name)
            return np.ctypeslib.as_array(v) if hasattr(v, "_length_") else v
        raise AttributeError(name)

    def efc(self, name):
        n = self.d.nefc
        a = self.s.np(name)
        if name == "AR":
            return a[: n * n].reshape(n, n).copy()
        return a[:n].copy()


class OracleEnv:
    """NightmareV3Env restated on the CPU (reference envs/nightmare_v3_env.py), numpy in / numpy out."""

    def __init__(self, num_envs, seed=0, env_id_offset=0, num_threads=1):
        self.L = lib()
        self.N = num_envs
        self.h = self.L.nmo_env_create(num_envs, seed, env_id_offset, num_threads)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.nmo_env_destroy(self.h)
            self.h = None

    def configure(self, reward_scales=None, tibia_contact_mode=1, tibia_max_contact_force=2.0, body_contact_mode=1,
                  body_max_contact_force=2.0, base_height_target=0.1, max_contact_force=10.0):
        """Non-default reward table (dict name -> RAW config scale; missing names = 0) and contact modes (config.py:17-21,77-100)."""
        sc = None
        if reward_scales is not None:
            sc = np.array([float(reward_scales.get(n, 0.0)) for n in REW_NAMES])
        self.L.nmo_env_configure(self.h, _ptr(sc), tibia_contact_mode, tibia_max_contact_force, body_contact_mode, body_max_contact_force,
                                 base_height_target, max_contact_force)

    def get_feet_state(self):
        air, last, filt = np.empty((self.N, 6)), np.empty((self.N, 6), np.uint8), np.empty((self.N, 6), np.uint8)
        self.L.nmo_env_get_feet_state(self.h, _ptr(air), _ptr(last), _ptr(filt))
        return air, last, filt

    def set_feet_state(self, air, last, filt):
        air, last, filt = np.ascontiguousarray(air, np.float64), np.ascontiguousarray(last, np.uint8), np.ascontiguousarray(filt, np.uint8)
        self.L.nmo_env_set_feet_state(self.h, _ptr(air), _ptr(last), _ptr(filt))

    def reset_idx(self, ids=None, cmd_u=None):
        if ids is None:
            self.L.nmo_env_reset_idx(self.h, None, 0, _ptr(None if cmd_u is None else np.ascontiguousarray(cmd_u, np.float64)))
        else:
            ids = np.ascontiguousarray(ids, np.int32)
            cu = None if cmd_u is None else np.ascontiguousarray(cmd_u, np.float64)
            self.L.nmo_env_reset_idx(self.h, _ptr(ids), len(ids), _ptr(cu))

    def step(self, actions, cmd_u=None):
        N = self.N
        a = np.ascontiguousarray(actions, np.float32).reshape(N, 18)
        cu = None if cmd_u is None else np.ascontiguousarray(cmd_u, np.float64).reshape(N, 4)
        obs = np.empty((N, 66), np.float32)
        rew = np.empty(N, np.float32)
        done = np.empty(N, np.int64)
        to = np.empty(N, np.float32)
        obs64 = np.empty((N, 66), np.float64)
        rew64 = np.empty(N, np.float64)
        self.L.nmo_env_step(self.h, _ptr(a), _ptr(cu), _ptr(obs), _ptr(rew), _ptr(done), _ptr(to), _ptr(obs64), _ptr(rew64))
        self.obs64, self.rew64 = obs64, rew64
        return obs, rew, done, to

    def step_physics(self, actions):
        """mj_step x decimation only (BASELINE config 2; reference simple_test.py:25-45)."""
        a = np.ascontiguousarray(actions, np.float32).reshape(self.N, 18)
        self.L.nmo_env_step_physics(self.h, _ptr(a))

    def set_noise(self, noise_scale_vec=None, u=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, np.float64)
        vec, u = f(noise_scale_vec), f(u)
        self.L.nmo_env_set_noise(self.h, _ptr(vec), _ptr(u))

    def reset(self):
        self.reset_idx()
        return self.step(np.zeros((self.N, 18), np.float32))[0]

    def get_state(self):
        N = self.N
        qpos, qvel, qw = np.empty((N, NQ)), np.empty((N, NV)), np.empty((N, NV))
        self.L.nmo_env_get_state(self.h, _ptr(qpos), _ptr(qvel), _ptr(qw))
        return qpos, qvel, qw

    def set_state(self, qpos=None, qvel=None, qacc_warmstart=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, np.float64)
        qpos, qvel, qw = f(qpos), f(qvel), f(qacc_warmstart)
        self.L.nmo_env_set_state(self.h, _ptr(qpos), _ptr(qvel), _ptr(qw))

    def get_buffers(self):
        N = self.N
        out = dict(dof_pos=np.empty((N, 18)), dof_vel=np.empty((N, 18)), actions=np.empty((N, 18)), commands=np.empty((N, 3)),
                   ep_len=np.empty(N, np.int64), episode_sums=np.empty((NREW, N)))
        self.L.nmo_env_get_buffers(self.h, *[_ptr(out[k]) for k in ("dof_pos", "dof_vel", "actions", "commands", "ep_len", "episode_sums")])
        return out

    def set_buffers(self, dof_pos=None, dof_vel=None, actions=None, commands=None, ep_len=None, episode_sums=None):
        f = lambda a, t=np.float64: None if a is None else np.ascontiguousarray(a, t)
        args = [f(dof_pos), f(dof_vel), f(actions), f(commands), f(ep_len, np.int64), f(episode_sums)]
        self.L.nmo_env_set_buffers(self.h, *[_ptr(a) for a in args])

    def episode_stats(self):
        out = np.empty(NREW)
        n = self.L.nmo_env_episode_stats(self.h, _ptr(out))
        return n, out

    def data(self, i):
        return self.L.nmo_env_data(self.h, i).contents

    def debug(self):
        N = self.N
        out = dict(base_lin_vel=np.empty((N, 3)), base_ang_vel=np.empty((N, 3)), projected_gravity=np.empty((N, 3)),
                   tibia=np.empty((N, 6)), feet=np.empty((N, 6)), body=np.empty(N), rew_terms=np.empty((NREW, N)))
        self.L.nmo_env_get_debug(self.h, *[_ptr(out[k]) for k in ("base_lin_vel", "base_ang_vel", "projected_gravity", "tibia", "feet", "body", "rew_terms")])
        return out


def rand_u24(seed, genv, ctr):
    return lib().nmo_rand_u24(seed, genv, ctr)
