"""What a change of the model compiler did to the collision tables: python scripts/hulltable_diff.py OLD.npz [NEW.npz]

Per colliding mesh: hull vertex counts, whether the vertex SETS agree (positions within 1e-7 m - the vertex coordinates themselves may
move by a float32 rounding), how many vertices have another neighbour ORDER / another neighbour SET; then the plane-hull contact sets
of the 200 poses of tests/test_physics_known_answers.py under either table (the numpy restatement of the documented rule).
Used for DESIGN section 2's table (round 5: qhull fed the raw, unscaled STL floats)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load(path):
    z = np.load(path)
    return {k: (z[k].item() if z[k].shape == () else z[k]) for k in z.files}


def main():
    from nightmare_rl_amd.model import compile_model as cm
    A = load(sys.argv[1])
    B = load(sys.argv[2]) if len(sys.argv) > 2 else cm.load_tables()
    names = ["base_link"] + [f"leg_{i}_tibia" for i in range(1, 7)]
    print("mesh            verts old/new  same set  nbr order differs  nbr set differs  max |dv| of matched vertices")
    for g in range(int(A["ncol"])):
        na, va = int(A["col_nvert"][g]), int(A["col_vadr"][g])
        nb, vb = int(B["col_nvert"][g]), int(B["col_vadr"][g])
        Va, Vb = A["hull_vert"][va:va + na], B["hull_vert"][vb:vb + nb]
        d = np.linalg.norm(Va[:, None, :] - Vb[None, :, :], axis=2)
        ja = d.argmin(axis=1)
        matched = d[np.arange(na), ja] < 1e-7
        same = na == nb and matched.all() and len(set(ja.tolist())) == na
        order = sets = 0
        if same:
            for i in range(na):
                ra = [int(ja[k]) for k in A["hull_nbr"][va + i] if k >= 0]
                rb = [int(k) for k in B["hull_nbr"][vb + ja[i]] if k >= 0]
                order += ra != rb
                sets += set(ra) != set(rb)
        print(f"{names[g]:14s}  {na:4d} / {nb:4d}   {'yes' if same else 'NO ':3s}       {order if same else '-':>5}              {sets if same else '-':>5}"
              f"            {d[np.arange(na), ja][matched].max():.2e}  (unmatched old vertices: {int((~matched).sum())})")

    import test_physics_known_answers as ka
    rng = np.random.default_rng(7)
    differ = ncon_a = ncon_b = 0
    for trial in range(200):                                     # the pose population of test_plane_hull_contact_set_...
        q = np.array(B["qpos0"], dtype=np.float64)
        kind = trial % 4
        if kind == 0:
            q[7:] = np.tile([0.0, -0.9, 0.6], 6) + rng.uniform(-0.3, 0.3, 18)
            q[2] = rng.uniform(0.05, 0.12)
        elif kind == 1:
            q[7:] = rng.uniform(-1.0, 1.0, 18)
            q[2] = rng.uniform(-0.01, 0.03)
        elif kind == 2:
            q[7:] = np.tile([0.0, 0.9, -2.2], 6) + rng.uniform(-0.25, 0.25, 18)
            q[2] = rng.uniform(0.0, 0.06)
        else:
            q[7:] = np.tile([0.0, -0.5, 0.3], 6) + rng.uniform(-0.6, 0.6, 18)
            q[2] = rng.uniform(0.02, 0.1)
        quat = np.array([1.0, 0, 0, 0]) + (0.02 if kind != 3 else 0.35) * rng.normal(size=4)
        q[3:7] = quat / np.linalg.norm(quat)
        out = []
        for T in (A, B):
            ka.T = T
            want, _ = ka.plane_hull_contacts_np(q)
            out.append(want)
        ncon_a += len(out[0])
        ncon_b += len(out[1])
        a = sorted((b, *np.round(p, 6)) for b, p, _ in out[0])
        b = sorted((b, *np.round(p, 6)) for b, p, _ in out[1])
        differ += a != b
    print(f"200 known-answer poses: contact set (body, position to 1e-6 m) differs in {differ} poses; contacts old {ncon_a}, new {ncon_b}")


if __name__ == "__main__":
    main()
