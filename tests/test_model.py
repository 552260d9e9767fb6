"""Known-answer tests of the model compiler (nightmare_rl_amd/model/compile_model.py): the rules that set every body's mass and
inertia, the convex-hull graphs the collision stage walks, and the qpos0 constants (body_invweight0, meaninertia).

MuJoCo 3.1.2 is not installable here, so these pin the compiler against numbers worked out BY HAND (boxes, planar-face pyramids)
or by an independent route (finite-difference Jacobians), not against a real MjModel: parity with MuJoCo stays unpinned."""
import numpy as np
import pytest

from nightmare_rl_amd.model.compile_model import hull_with_graph, load_tables, mesh_props

T = load_tables()


def l_prism(arm):
    """Triangulated L-shaped prism: ([0,arm] x [0,1]  U  [0,1] x [1,arm]) x [0,1], outward normals."""
    a = float(arm)
    poly = np.array([[0, 0], [a, 0], [a, 1], [1, 1], [1, a], [0, a]], float)     # counter-clockwise
    V = np.vstack([np.c_[poly, np.zeros(6)], np.c_[poly, np.ones(6)]])
    tris2d = [(0, 1, 2), (0, 2, 3), (0, 3, 4), (0, 4, 5)]                           # fan from the reflex-free corner (0,0)
    F = []
    for (i, j, k) in tris2d:
        F.append((i, k, j))                     # bottom, normal -z
        F.append((6 + i, 6 + j, 6 + k))         # top, normal +z
    for i in range(6):
        j = (i + 1) % 6
        F.append((i, j, 6 + j))
        F.append((i, 6 + j, 6 + i))
    return V, np.array(F)


def box_inertia(m, size, c):
    """Inertia tensor of a uniform box (mass m, edge lengths size) about the point -c from its centre (parallel axes)."""
    a, b, h = size
    I = np.diag([m * (b * b + h * h) / 12, m * (a * a + h * h) / 12, m * (a * a + b * b) / 12])
    return I + m * (np.dot(c, c) * np.eye(3) - np.outer(c, c))


def test_mesh_inertia_rule_on_a_star_convex_L_prism_by_hand():
    """L of arm 2: both reference points of the legacy rule (area-weighted face centroid (6/7, 6/7, 1/2) and the COM) lie in the
    kernel [0,1]^2 of the L, so |pyramid volumes| = signed volumes and the rule must give the exact solid:
    two boxes 2x1x1 and 1x1x1 -> volume 3, COM (5/6, 5/6, 1/2), inertia by the parallel-axis theorem."""
    V, F = l_prism(2)
    vol, com, I = mesh_props(V, F, legacy=True)
    assert abs(vol - 3.0) < 1e-12
    np.testing.assert_allclose(com, [5 / 6, 5 / 6, 0.5], atol=1e-12)
    cA, cB = np.array([1.0, 0.5, 0.5]), np.array([0.5, 1.5, 0.5])
    hand = box_inertia(2.0, (2, 1, 1), cA - com) + box_inertia(1.0, (1, 1, 1), cB - com)
    np.testing.assert_allclose(I, hand, atol=1e-12)
    vol_e, com_e, I_e = mesh_props(V, F, legacy=False)
    np.testing.assert_allclose([vol_e, *com_e], [vol, *com], atol=1e-12)
    np.testing.assert_allclose(I_e, I, atol=1e-12)


def test_legacy_rule_overcounts_a_non_star_convex_L_prism_by_hand():
    """L of arm 4: the area-weighted face centroid p = (1.4, 1.4, 0.5) lies OUTSIDE the solid, so the two inner walls are seen from
    their outer side and the legacy rule adds their pyramids instead of subtracting them. By hand, pyramid = area x distance / 3:
      top + bottom 2 x 7 x 0.5/3; walls y=0 and x=0: 4 x 1.4/3 each; x=4 and y=4: 1 x 2.6/3 each; inner walls y=1, x=1: 3 x 0.4/3 each
      -> 7.8 + 0.8 = 8.6 (exact volume 7.8 - 0.8 = 7).  COM from the same weights: x = 0.75 x (12.2/8.6) + 0.25 x 1.4.
    The mass then comes from the volume re-accumulated about that COM (what mesh_props returns)."""
    V, F = l_prism(4)
    vol_e, com_e, _ = mesh_props(V, F, legacy=False)
    assert abs(vol_e - 7.0) < 1e-12
    np.testing.assert_allclose(com_e, [9.5 / 7, 9.5 / 7, 0.5], atol=1e-12)
    cx = 0.75 * (12.2 / 8.6) + 0.25 * 1.4
    vol2 = 7 / 3 + 2 * (4 * cx / 3) + 2 * ((4 - cx) / 3) + 2 * (3 * (cx - 1) / 3)
    vol, com, _ = mesh_props(V, F, legacy=True)
    np.testing.assert_allclose(com, [cx, cx, 0.5], atol=1e-12)
    assert abs(vol - vol2) < 1e-12 and vol > 8.6 and abs(vol2 - 8.6558) < 1e-3


def test_body_masses_follow_the_rule_and_settotalmass():
    assert abs(T["body_mass"].sum() - 3.0) < 1e-12                     # settotalmass=3 (mjmodel.xml:2)
    m = T["body_mass"]
    np.testing.assert_allclose(m[[1, 2, 3, 4]], [1.729, 0.0386, 0.0477, 0.1255], atol=5e-4)   # legacy rule
    # the legacy rule only ever adds volume: every body's legacy share of its exact share moves with the whole robot's overcount
    assert (T["body_mass_exactrule"][1:] > 0).all()


def regular_icosahedron():
    p = (1 + 5 ** 0.5) / 2
    return np.array([[0, s1, s2 * p] for s1 in (-1, 1) for s2 in (-1, 1)] + [[s1, s2 * p, 0] for s1 in (-1, 1) for s2 in (-1, 1)] +
                    [[s2 * p, 0, s1] for s1 in (-1, 1) for s2 in (-1, 1)], float)


def test_hull_graph_of_an_icosahedron_and_a_cube():
    ico = regular_icosahedron()
    rng = np.random.default_rng(0)
    pts = np.vstack([ico, rng.uniform(-0.5, 0.5, (40, 3))])            # interior points must not appear in the hull
    vid, nbr = hull_with_graph(pts)
    assert sorted(vid.tolist()) == list(range(12))
    assert all(len(n) == 5 for n in nbr)                                # every vertex of an icosahedron has 5 neighbours
    for i, n in enumerate(nbr):
        d = np.linalg.norm(pts[vid[n]] - pts[vid[i]], axis=1)
        np.testing.assert_allclose(d, 2.0, atol=1e-12)                 # ... at edge length 2
        assert all(i in nbr[j] for j in n)                              # symmetric
    cube = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], float)
    vid, nbr = hull_with_graph(cube)
    assert len(vid) == 8
    deg = np.array([len(n) for n in nbr])
    assert deg.sum() == 2 * 18 and deg.min() >= 3 and deg.max() <= 6    # 12 edges + one diagonal per triangulated face ('Qt')
    for i, n in enumerate(nbr):
        assert all(i in nbr[j] for j in n)
        assert all(np.linalg.norm(cube[vid[j]] - cube[vid[i]]) < 1.5 for j in n)   # edges and face diagonals, never a space diagonal


def test_compiled_hull_tables_are_consistent():
    for g in range(int(T["ncol"])):
        nv, va = int(T["col_nvert"][g]), int(T["col_vadr"][g])
        V = T["hull_vert"][va:va + nv]
        nb = T["hull_nbr"][va:va + nv]
        assert ((nb >= -1) & (nb < nv)).all()
        for i in range(nv):
            row = nb[i][nb[i] >= 0]
            assert len(row) >= 3 and len(set(row.tolist())) == len(row) and i not in row
            assert all(i in nb[j] for j in row)                         # undirected graph
        # a hill climb over the graph reaches the exhaustive support vertex from anywhere (what the warm-started search relies on)
        rng = np.random.default_rng(g)
        for _ in range(20):
            d = rng.normal(size=3)
            best = int(np.argmax(V @ d))
            cur = int(rng.integers(nv))
            for _hop in range(nv):
                row = nb[cur][nb[cur] >= 0]
                j = row[np.argmax(V[row] @ d)]
                if V[j] @ d <= V[cur] @ d:
                    break
                cur = int(j)
            assert abs(V[cur] @ d - V[best] @ d) < 1e-12


def test_invweight0_and_meaninertia_from_finite_difference_jacobians(oracle_mod):
    """engine_setconst.c set0: body_invweight0 = mean diagonal of J M^-1 J' (translational, rotational) at the body COM, qpos0;
    stat.meaninertia = mean diagonal of M. J here comes from finite differences of the oracle's forward kinematics, M from the
    oracle's CRBA (itself checked against an independent numpy CRBA in test_oracle_physics.py)."""
    p = oracle_mod.Physics()
    q0 = T["qpos0"].copy()

    def pose(q):
        p.qpos[:] = q
        p.qvel[:] = 0
        p.forward()
        return p.xipos.copy(), p.ximat.copy().reshape(-1, 3, 3), p.qM.copy()

    x0, R0, M = pose(q0)
    assert abs(np.trace(M) / 24 - float(T["meaninertia"])) < 1e-12
    eps = 1e-6
    nb = 20
    Jp, Jr = np.zeros((nb, 3, 24)), np.zeros((nb, 3, 24))
    for i in range(24):
        q = q0.copy()
        if i < 3:
            q[i] += eps
        elif i < 6:                                   # body-frame rotation about axis i-3
            w = np.zeros(3)
            w[i - 3] = eps
            a = np.linalg.norm(w)
            dq = np.r_[np.cos(a / 2), np.sin(a / 2) * w / a]
            b = q0[3:7]
            q[3:7] = [b[0] * dq[0] - b[1:] @ dq[1:], *(b[0] * dq[1:] + dq[0] * b[1:] + np.cross(b[1:], dq[1:]))]
        else:
            q[7 + i - 6] += eps
        x1, R1, _ = pose(q)
        Jp[:, :, i] = (x1 - x0) / eps
        for b in range(nb):
            S = (R1[b] @ R0[b].T - np.eye(3)) / eps
            Jr[b, :, i] = [S[2, 1], S[0, 2], S[1, 0]]
    Minv = np.linalg.inv(M)
    for b in range(1, nb):
        At = Jp[b] @ Minv @ Jp[b].T
        Ar = Jr[b] @ Minv @ Jr[b].T
        np.testing.assert_allclose([np.trace(At) / 3, np.trace(Ar) / 3], T["body_invweight0"][b], rtol=2e-5)
    assert (T["body_invweight0"][0] == 0).all()


# ------------------------------------------------------------------------------------------------ the MuJoCo comparer itself
def test_mjmodel_comparer_accepts_the_tables_and_names_the_stage_that_differs():
    """tests/mjmodel_compare.py is what the first box with `import mujoco` runs (tests/test_mujoco_crosscheck.py). Here it runs against a
    stand-in with MjModel's attribute layout assembled from the tables: no finding on the tables themselves; a hull numbered the other way
    round is a note, not a finding; and each tampering below is found and attributed to its stage."""
    import copy

    from mjmodel_compare import compare_model, graph_of, synthetic_mjmodel
    m = synthetic_mjmodel(T)
    assert compare_model(m, T) == []
    gid, nbr = graph_of(m, 1)
    va, nv = int(T["col_vadr"][1]), int(T["col_nvert"][1])
    assert gid.tolist() == T["hull_point_id"][va:va + nv].tolist() and len(nbr) == nv
    rev = compare_model(synthetic_mjmodel(T, hull_numbering="reversed"), T)
    assert len(rev) == int(T["ncol"]) and all(": note:" in f for f in rev)

    def tampered(fn):
        m2 = copy.deepcopy(m)
        fn(m2)
        return [f for f in compare_model(m2, T) if ": note:" not in f]

    def swap_two_neighbours(m2):                       # another Qt triangulation: same sets, another order
        adr = int(m2.mesh_graphadr[2])
        nvg = int(m2.mesh_graph[adr])
        e = adr + 2 + 2 * nvg + int(m2.mesh_graph[adr + 2 + 5])
        m2.mesh_graph[e], m2.mesh_graph[e + 1] = m2.mesh_graph[e + 1], m2.mesh_graph[e]

    def another_hull_vertex(m2):                       # a nearly coplanar point that qhull keeps on one input and drops on the other
        adr = int(m2.mesh_graphadr[3])
        nvg = int(m2.mesh_graph[adr])
        free = sorted(set(range(int(m2.mesh_vertnum[3]))) - set(m2.mesh_graph[adr + 2 + nvg: adr + 2 + 2 * nvg].tolist()))
        m2.mesh_graph[adr + 2 + nvg + 7] = free[0]

    def one_ulp_in_a_vertex(m2):                       # float32 x float32(0.001) instead of one rounding of the double product, magnified
        i = int(m2.mesh_vertadr[4]) + int(T["hull_point_id"][int(T["col_vadr"][4]) + 3])
        m2.mesh_vert[i, 0] += np.float32(3e-7)

    f = tampered(swap_two_neighbours)
    assert len(f) == 1 and "col mesh 2" in f[0] and "1 vertices with another neighbour ORDER, 0 with another neighbour SET" in f[0]
    f = tampered(another_hull_vertex)
    assert any("col mesh 3" in x and "hull vertex SET differs" in x and "MakeGraph" in x for x in f)
    f = tampered(one_ulp_in_a_vertex)
    assert any("col mesh 4" in x and "positions in the body frame" in x for x in f)
    f = tampered(lambda m2: m2.geom_rbound.__setitem__(1, m2.geom_rbound[1] * 1.001))
    assert len(f) == 1 and "geom_rbound" in f[0]
    f = tampered(lambda m2: m2.body_inertia.__setitem__((4, 0), m2.body_inertia[4, 0] * 1.01))
    assert any("inertia tensor" in x for x in f)


def test_qhull_sees_the_raw_unscaled_stl_vertices_and_the_scale_rounds_once():
    """The order upstream's mesh compile works in (user_mesh.cc mjCMesh::Compile, 3.1.2): MakeGraph BEFORE Process(), i.e. qhull gets the
    de-duplicated float32 file values cast to double, unscaled; the `scale` attribute is applied afterwards as float32(double(v) * scale).
    Checked on a synthetic point cloud: (a) scale_verts rounds the double product once - it differs from float32 x float32(0.001) in some
    entries, never by more than one ulp; (b) the graph built from raw points indexes the same points after scaling."""
    from nightmare_rl_amd.model.compile_model import scale_verts
    rng = np.random.default_rng(5)
    raw = (rng.uniform(-150, 150, (4000, 3))).astype(np.float32)
    once = scale_verts(raw, [0.001, 0.001, 0.001])
    twice = raw * np.float32(0.001)
    assert once.dtype == np.float32
    exact = raw.astype(np.float64) * 0.001
    assert (np.abs(once.astype(np.float64) - exact) <= np.abs(twice.astype(np.float64) - exact) + 1e-30).all()
    differs = once != twice
    assert differs.sum() > 0                       # float32(0.001) is off by 0.8 half-ulps: more than half of the entries move
    assert (np.abs(once[differs].astype(np.float64) - twice[differs]) <= np.spacing(np.abs(once[differs])).astype(np.float64)).all()
    # the compiled tables carry what the cross-check needs, consistently
    for g in range(int(T["ncol"])):
        va, nv = int(T["col_vadr"][g]), int(T["col_nvert"][g])
        pid = T["hull_point_id"][va:va + nv]
        assert (np.diff(pid) > 0).all() and pid[-1] < T["col_mesh_nvert"][g]
        Rg = np.asarray(__import__("mjmodel_compare").quat_to_mat(T["col_geom_quat"][g]))
        Pb = T["col_geom_pos"][g] + T["hull_mesh_vert"][va:va + nv].astype(np.float64) @ Rg.T
        np.testing.assert_allclose(Pb, T["hull_vert"][va:va + nv], atol=1e-9)       # quaternion round trip of the geom frame only
        assert abs(np.linalg.norm(np.abs(T["hull_mesh_vert"][va:va + nv]).max(axis=0)) - T["col_rbound"][g]) < 1e-3
