"""Golden vectors for the gait/IK engine, produced by importing the reference's own nikengine package
(/root/reference/nikengine; importable in the build container). Fixtures hold inputs and outputs only.

Usage: python tests/golden/make_nik_goldens.py"""
import io
import contextlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/root/reference")

with contextlib.redirect_stdout(io.StringIO()):
    from nikengine import engine as ref
    from nikengine.modules.math import asymmetrical_sigmoid, shortest_distance_two_segments_2d
    from nikengine.modules.bezier import Bezier


def rollout(fps, cmds, dt, ticks, mode="walk"):
    """cmds: function tick -> (lin, ang, state). Returns times, inputs, 18 angles per tick."""
    ref.config.ENGINE_FPS = fps
    with contextlib.redirect_stdout(io.StringIO()):
        node = ref.EngineNode()
    out, inp, ts = [], [], []
    for k in range(ticks):
        t = k * dt
        lin, ang, state = cmds(k)
        ref.set_time_s(t)
        with contextlib.redirect_stdout(io.StringIO()):
            a = node.update(lin, ang, state, mode)
        out.append(np.array(a, dtype=np.float64))
        inp.append([lin, ang, 1.0 if state == "awake" else 0.0])
        ts.append(t)
    return np.array(ts), np.array(inp), np.array(out)


def main_gaits():
    """ripple / wave gaits (engine.py:214-225) and a gait change while walking. EngineNode.update has no gait argument: upstream the
    gait is state.cmd.gait, a field of a Command object that every EngineNode shares (RobotState.cmd is a class attribute, :402-406);
    it is read when WalkState is built (:543) and again whenever a step completes (:627)."""
    fps = 51.0
    data = {}

    def rollout_gait(schedule, ticks, cmd):
        ref.config.ENGINE_FPS = fps
        with contextlib.redirect_stdout(io.StringIO()):
            node = ref.EngineNode()
        node._robot_state.cmd.gait = "tripod"
        out, gaits = [], []
        for k in range(ticks):
            if k in schedule:
                node._robot_state.cmd.gait = schedule[k]
            ref.set_time_s(k / fps)
            with contextlib.redirect_stdout(io.StringIO()):
                out.append(np.array(node.update(cmd[0], cmd[1], "awake", "walk"), dtype=np.float64))
            gaits.append(["tripod", "ripple", "wave"].index(node._robot_state.cmd.gait))
        node._robot_state.cmd.gait = "tripod"        # the Command object is shared by every later EngineNode
        return np.array(out), np.array(gaits)

    for name in ("ripple", "wave"):
        o, gsel = rollout_gait({0: name}, 700, (0.06, 0.25))
        data[name + "_out"], data[name + "_gait"] = o, gsel
    o, gsel = rollout_gait({0: "tripod", 300: "wave", 520: "ripple", 700: "tripod"}, 900, (-0.05, -0.3))
    data["switch_out"], data["switch_gait"] = o, gsel
    data["fps"] = np.array(fps)
    data["cmd_fixed"] = np.array([0.06, 0.25])
    data["cmd_switch"] = np.array([-0.05, -0.3])
    path = os.path.join(ROOT, "tests", "golden", "nikengine_gaits.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    rng = np.random.default_rng(0)
    data = {}
    # (1) the documented caller (custom_play.py:49-52,67-76): idle tick, then awake/walk at constant command, fps = 1/(0.008*2)
    fps = 1.0 / 0.008 / 2
    t, i, o = rollout(fps, lambda k: (0.05, 0.2, "idle" if k == 0 else "awake"), 0.016, 700)
    data.update(walk_t=t, walk_in=i, walk_out=o, walk_fps=np.array(fps))
    # (2) varying commands incl. fast ones that trigger the keep-out line search, default fps 51
    seq = [(rng.uniform(-0.25, 0.25), rng.uniform(-1.2, 1.2)) for _ in range(40)]
    t, i, o = rollout(51, lambda k: (*seq[min(k // 30, 39)], "awake"), 1 / 51, 1000)
    data.update(var_t=t, var_in=i, var_out=o, var_fps=np.array(51.0))
    # (3) stand mode (GetUp -> Stand)
    t, i, o = rollout(51, lambda k: (0.0, 0.0, "awake"), 1 / 51, 260, mode="stand")
    data.update(stand_t=t, stand_in=i, stand_out=o)
    # (4) relative_ik table incl. unreachable targets (too far / too close)
    pts = np.concatenate([rng.uniform([0.05, -0.2, -0.25], [0.35, 0.2, -0.02], (200, 3)),
                          rng.uniform([0.4, -0.3, -0.3], [0.6, 0.3, 0.0], (20, 3)),        # too far
                          rng.uniform([0.066, -0.01, -0.03], [0.09, 0.01, -0.005], (20, 3))])  # too close
    dim = ref.config.DEFAULT_DIM
    with contextlib.redirect_stdout(io.StringIO()):
        ik = np.array([ref.EngineNode.relative_ik(p.copy(), dim) for p in pts])
    data.update(ik_in=pts, ik_out=ik)
    # (5) helpers
    segs = rng.uniform(-0.4, 0.4, (300, 4, 2))
    data.update(seg_in=segs, seg_out=np.array([shortest_distance_two_segments_2d(*s) for s in segs], dtype=np.float64))
    xs = np.linspace(-0.2, 1.2, 57)
    data.update(sig_in=xs, sig_out=np.array([asymmetrical_sigmoid(x) for x in xs]))
    P = [rng.uniform(-1, 1, (6, 3)) for _ in range(4)]
    ts = np.linspace(0, 1, 21)
    data.update(bez_pts=np.array(P), bez_t=ts, bez_out=np.array([Bezier.Point(float(t), P) for t in ts]))
    # constants of MyConfig the engine is specialised for
    c = ref.config
    data.update(cfg_default_pose=c.DEFAULT_POSE, cfg_sit_pose=c.DEFAULT_SIT_POSE, cfg_pose_offset=c.POSE_OFFSET,
                cfg_rel_convert=c.POSE_REL_CONVERT.astype(np.float64), cfg_servo_offset=c.SERVO_OFFSET, cfg_urdf_offsets=c.URDF_JOINT_OFFSETS,
                cfg_dim=c.DEFAULT_DIM)
    path = os.path.join(ROOT, "tests", "golden", "nikengine.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", "walk out range", o.min(), o.max())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gaits":
        main_gaits()
    else:
        main()
