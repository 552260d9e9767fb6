"""ctypes binding of the host SIMT emulation of the device kernel (tests only; see nm_emul.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libnm_emul.so")
SRC = [os.path.join(HERE, "nm_emul.cpp")] + [os.path.join(HERE, "..", "..", "nightmare_rl_amd", "csrc", f)
                                             for f in ("nm_core.h", "simt.h", "nm_host_model.h", os.path.join("..", "model", "nm_model_data.h"))]


def build(force=False):
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRC):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-missing-braces", "-DNM_DEBUG_SOLVER",
                               "-o", LIB, SRC[0]])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.emu_create.restype = C.c_void_p
        L.emu_create.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_int]
        L.emu_destroy.argtypes = [C.c_void_p]
        L.emu_step.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int] + [C.c_void_p] * 3
        L.emu_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.emu_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.emu_set_noise.argtypes = [C.c_void_p] * 3
        L.emu_record.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.emu_configure.argtypes = [C.c_void_p] * 3
        L.emu_feet.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.emu_hull_cache.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.emu_eplen.argtypes = [C.c_void_p]
        L.emu_eplen.restype = C.POINTER(C.c_int64)
        L.emu_together_count.restype = C.c_long
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


WHAT = dict(qpos=(0, 25), qvel=(1, 24), qwarm=(2, 24), dofpos=(3, 18), dofvel=(4, 18), act=(5, 18), cmd=(6, 3), epsum=(7, 16))
REW_NAMES = ["action_rate", "ang_vel_xy", "base_height", "body_contact_forces", "default_position", "dof_acc", "dof_vel", "feet_air_time",
             "feet_contact_forces", "lin_vel_z", "orientation", "stand_still", "torques", "tracking_ang_vel", "tracking_lin_vel", "termination"]


class EmulEnv:
    def __init__(self, N, double=False, seed=0, env_off=0, envs_per_wave=2):
        self.L = lib()
        self.N = N
        self.h = self.L.emu_create(N, int(double), seed, env_off, envs_per_wave)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.emu_destroy(self.h)
            self.h = None

    def get(self, name):
        w, k = WHAT[name]
        out = np.empty((self.N, k))
        self.L.emu_get(self.h, w, _p(out))
        return out

    def set(self, name, val):
        w, k = WHAT[name]
        v = np.ascontiguousarray(val, np.float64).reshape(self.N, k)
        self.L.emu_set(self.h, w, _p(v))

    def configure(self, reward_scales, tibia_contact_mode=1, tibia_max_contact_force=2.0, body_contact_mode=1, body_max_contact_force=2.0,
                  base_height_target=0.1, max_contact_force=10.0):
        sc = np.array([float(reward_scales.get(n, 0.0)) for n in REW_NAMES])
        m = np.array([tibia_contact_mode, tibia_max_contact_force, body_contact_mode, body_max_contact_force, base_height_target, max_contact_force], np.float64)
        self.L.emu_configure(self.h, _p(sc), _p(m))

    def get_feet_state(self):
        air, fl = np.zeros((self.N, 6)), np.zeros(self.N, np.int32)
        self.L.emu_feet(self.h, 0, _p(air), _p(fl))
        bits = (fl[:, None] >> np.arange(12)[None, :]) & 1
        return air, bits[:, :6].astype(np.uint8), bits[:, 6:].astype(np.uint8)

    def set_feet_state(self, air, last, filt):
        air = np.ascontiguousarray(air, np.float64)
        fl = ((np.asarray(last, np.int32) << np.arange(6)).sum(1) | (np.asarray(filt, np.int32) << (6 + np.arange(6))).sum(1)).astype(np.int32)
        self.L.emu_feet(self.h, 1, _p(air), _p(fl))

    def hull_cache(self, value=None):
        """[N,8] ints: warm-start vertex of the 7 colliding meshes + the env's running exhaustive-scan count; set when `value` given."""
        hc = np.zeros((self.N, 8), np.int32) if value is None else np.ascontiguousarray(value, np.int32).reshape(self.N, 8)
        self.L.emu_hull_cache(self.h, int(value is not None), _p(hc))
        return hc

    def set_noise(self, vec=None, u=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, np.float64)
        vec, u = f(vec), f(u)
        self.L.emu_set_noise(self.h, _p(vec), _p(u))

    def record(self, env):
        """Select the env whose pre-reset state is logged; returns the last record (qpos25, qvel24, nbad)."""
        out = np.zeros(50)
        self.L.emu_record(self.h, env, _p(out))
        return out[:25], out[25:49], int(out[49])

    @property
    def eplen(self):
        return np.ctypeslib.as_array(self.L.emu_eplen(self.h), (self.N,))

    def step(self, actions, cmd_u=None, nsub=2, physics_only=False, want_dbg=False):
        N = self.N
        a = np.ascontiguousarray(actions, np.float32).reshape(N, 18)
        cu = None if cmd_u is None else np.ascontiguousarray(cmd_u, np.float64).reshape(N, 4)
        obs = np.zeros((N, 66), np.float32)
        rew = np.zeros(N, np.float32)
        done = np.zeros(N, np.int64)
        to = np.zeros(N, np.float32)
        dbg = np.zeros((N, 256)) if want_dbg else None
        ssum = np.zeros(16)
        scnt = np.zeros(2, np.int32)
        self.L.emu_step(self.h, _p(a), _p(cu), _p(obs), _p(rew), _p(done), _p(to), nsub, int(physics_only), _p(dbg), _p(ssum), _p(scnt))
        self.dbg, self.stat_sum, self.stat_cnt = dbg, ssum, scnt
        return obs, rew, done, to
