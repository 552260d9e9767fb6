"""Scripted gait / inverse-kinematics engine on the GPU, with the reference's calling convention.

Mirrors reference nikengine/engine.py: `set_time_s` (:12-19), `config.ENGINE_FPS` (read on every tick, custom_play.py:51
changes it after construction) and `EngineNode.update(lin_speed, ang_speed, state, mode)` (:710-715), but one object drives
`num_envs` independent engines in a single kernel launch (csrc/nm_nik.hip). There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

GAIT_NAMES = ["tripod", "ripple", "wave"]      # engine.py:214-225
FSM_NAMES = ["idle", "adjust_get_up", "get_up", "sit", "adjust_sit", "stand", "walk"]


class _Config:
    """The knobs custom_play.py touches. Geometry (poses, link lengths, gait table) is compiled into the kernel."""
    ENGINE_FPS = 51.0          # engine.py:24


config = _Config()
_time_s = 0.0


def set_time_s(t):
    global _time_s
    _time_s = float(t)


def get_time_s():
    return _time_s


class EngineNode:
    """`num_envs` gait engines. `update` returns the 18 joint targets per env ([18] numpy for the reference's scalar call
    with num_envs == 1, else a [N,18] float32 tensor on the device)."""

    def __init__(self, num_envs=1, device="cuda:0", dtype=torch.float32):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.NightmareHipError("the gait engine runs on the GPU only (device must be cuda:N)")
        if dtype not in (torch.float32, torch.float64):
            raise ValueError("dtype must be float32 or float64")
        self.L = _lib.load()
        self.num_envs, self.device, self.dtype = int(num_envs), dev, dtype
        torch.cuda.init()
        self.h = self.L.nm_nik_create(self.num_envs, dev.index or 0)
        if not self.h:
            raise _lib.NightmareHipError(self.L.nm_last_error().decode())
        self.angles = torch.zeros(self.num_envs, 18, device=dev, dtype=dtype)
        self._lin = torch.zeros(self.num_envs, device=dev, dtype=torch.float64)
        self._ang = torch.zeros(self.num_envs, device=dev, dtype=torch.float64)
        self._flag = {}

    def __del__(self):
        if getattr(self, "h", None):
            self.L.nm_nik_destroy(self.h)
            self.h = None

    def _vec(self, buf, v):
        if torch.is_tensor(v):
            buf.copy_(v.reshape(self.num_envs))
        else:
            buf.fill_(float(v)) if np.ndim(v) == 0 else buf.copy_(torch.as_tensor(np.asarray(v, np.float64)).reshape(self.num_envs))
        return buf.data_ptr()

    def _flags(self, name, v, yes):
        """None (= all set) for the common scalar case, else a cached u8 device mask."""
        if isinstance(v, str):
            if v == yes:
                return None
            v = torch.zeros(self.num_envs, dtype=torch.uint8)
        elif not torch.is_tensor(v):
            v = torch.as_tensor(np.array([x == yes if isinstance(x, str) else bool(x) for x in v], np.uint8))
        buf = self._flag.setdefault(name, torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device))
        buf.copy_(v.reshape(self.num_envs).to(torch.uint8))
        return buf.data_ptr()

    def update(self, lin_speed, ang_speed, state="awake", mode="walk", time_s=None):
        now = _time_s if time_s is None else float(time_s)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        f32 = self.angles.data_ptr() if self.dtype == torch.float32 else None
        f64 = self.angles.data_ptr() if self.dtype == torch.float64 else None
        _lib.check(self.L.nm_nik_update(self.h, self._vec(self._lin, lin_speed), self._vec(self._ang, ang_speed),
                                        self._flags("awake", state, "awake"), self._flags("walk", mode, "walk"),
                                        now, float(config.ENGINE_FPS), f32, f64, stream))
        if self.num_envs == 1 and not torch.is_tensor(lin_speed):
            return self.angles[0].double().cpu().numpy()
        return self.angles

    def set_gait(self, gait, env_ids=None):
        """state.cmd.gait (engine.py:297) of all or the listed engines: 'tripod' (default), 'ripple' or 'wave'. A walking engine takes
        the new gait over when its current step completes (:627), a starting one when it begins to walk (:543). Upstream the Command
        object is a class attribute shared by every EngineNode of the process (:402-406); here it is per env."""
        if gait not in GAIT_NAMES:
            raise ValueError("gait must be one of %s" % GAIT_NAMES)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if env_ids is None:
            _lib.check(self.L.nm_nik_set_gait(self.h, None, self.num_envs, GAIT_NAMES.index(gait), stream))
        else:
            ids = np.ascontiguousarray(torch.as_tensor(env_ids).cpu().numpy(), np.int32)
            _lib.check(self.L.nm_nik_set_gait(self.h, ids.ctypes.data_as(C.c_void_p), len(ids), GAIT_NAMES.index(gait), stream))

    def reset(self, env_ids=None):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if env_ids is None:
            _lib.check(self.L.nm_nik_reset(self.h, None, 0, stream))
        else:
            ids = np.ascontiguousarray(torch.as_tensor(env_ids).cpu().numpy(), np.int32)
            _lib.check(self.L.nm_nik_reset(self.h, ids.ctypes.data_as(C.c_void_p), len(ids), stream))

    def get_state(self):
        pose = np.empty((self.num_envs, 18))
        fsm = np.empty(self.num_envs, np.int32)
        gss = np.empty(self.num_envs)
        _lib.check(self.L.nm_nik_get_state(self.h, pose.ctypes.data_as(C.c_void_p), fsm.ctypes.data_as(C.c_void_p), gss.ctypes.data_as(C.c_void_p)))
        return dict(pose=pose.reshape(self.num_envs, 6, 3), fsm=fsm, gait_step_state=gss)
