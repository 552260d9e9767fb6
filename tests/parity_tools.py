"""Helpers of the parity tests (test infrastructure; uses the CPU oracle).

`discrete_margin` answers, from the fp64 oracle alone, how close one env-step is to a DISCRETE decision of the collision stage
(reference path: MuJoCo's mjc_PlaneConvex driven from envs/nightmare_v3_env.py:200). A contact set is a discontinuous
function of the state at exactly these places:
  tie   two hull vertices of a mesh are the lowest within `m` metres            -> which vertex is the support point
  act   the support vertex is within `m` of the floor plane                     -> contact exists or not
  nbr   a hull neighbour of the support vertex is within `m` of the plane       -> extra contact exists or not
  tol   a penetrating neighbour is within `m` of the 0.3*rbound distance rule    -> extra contact kept or skipped
  pair  a tibia-tibia (MPR) contact is present or a pair is within `m` of touching
Two implementations that agree to rounding (fp32 kernel vs fp64 oracle) can only produce a LARGE one-step difference at a
state whose margin is at rounding level; the GPU tests assert exactly that for every env-step above the 1e-4 tolerance.
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_model = None


def model():
    global _model
    if _model is None:
        _model = dict(np.load(os.path.join(ROOT, "nightmare_rl_amd", "model", "nm_model.npz")))
    return _model


def plane_margins(xpos, xmat, reach=2e-6):
    """Margins (metres) of the floor-vs-hull decisions for body poses xpos[20,3], xmat[20,9] (oracle layout).
    Returns (min margin, kind). Meshes whose lowest vertex is more than `reach` above the plane decide nothing."""
    M = model()
    best, kind = np.inf, "none"
    for g in range(int(M["ncol"])):
        b = int(M["col_body"][g])
        nv, va = int(M["col_nvert"][g]), int(M["col_vadr"][g])
        V = M["hull_vert"][va:va + nv]
        R = np.asarray(xmat[b]).reshape(3, 3)
        P = V @ R.T + np.asarray(xpos[b])
        z = P[:, 2]
        order = np.argsort(z, kind="stable")
        i0 = int(order[0])
        if z[i0] > reach:
            continue
        cand = [(z[order[1]] - z[i0], "tie"), (abs(z[i0]), "act")]
        first = P[i0].copy()
        first[2] -= 0.5 * z[i0]
        tol = 0.3 * float(M["col_rbound"][g])
        for nb in M["hull_nbr"][va + i0]:
            if nb < 0:
                break
            cand.append((abs(z[nb]), "nbr"))
            if z[nb] < reach:
                cand.append((abs(np.linalg.norm(P[nb] - first) - tol), "tol"))
        m, k = min(cand)
        if m < best:
            best, kind = m, k
    return best, kind


def discrete_margin(orc, qpos, qvel, qwarm, ctrl, nsub=2):
    """Smallest decision margin over the `nsub` forward passes of one env-step that starts from (qpos, qvel, qacc_warmstart)
    with servo command ctrl[18]. Returns (margin in metres, kind, number of tibia-tibia contacts seen)."""
    p = orc.Physics()
    p.d.time = 0.0
    p.qpos[:] = qpos
    p.qvel[:] = qvel
    p.qacc_warmstart[:] = qwarm
    p.ctrl[:] = ctrl
    best, kind, npair = np.inf, "none", 0
    for s in range(nsub):
        p.forward()
        m, k = plane_margins(p.xpos, p.xmat)
        nb1 = int((p.con_body1[: p.d.ncon] > 0).sum())
        npair += nb1
        if nb1:
            m, k = 0.0, "pair"
        if m < best:
            best, kind = m, k
        p.step(1)
    return best, kind, npair


def servo_ctrl(actions_f32, dof_pos, action_scale=0.2, clip=1.0, p_gain=20.0):
    """E1 (env.py:152-156,181-192): the velocity command the env hands to the physics for raw policy actions."""
    a = np.clip(np.asarray(actions_f32, np.float32) * np.float32(action_scale), -clip, clip).astype(np.float64)
    default = np.tile([0.0, np.pi / 5, 0.0], 6)
    return ((a - default) - np.asarray(dof_pos)) * p_gain
