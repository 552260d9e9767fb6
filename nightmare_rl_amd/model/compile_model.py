"""Model compiler: MJCF + STL robot description -> flat constant tables.

Reads the Nightmare-v3 robot description (reference `models/nightmare_v3/mjmodel.xml`
and the STL meshes beside it) ONCE, offline, and emits

  * ``nm_model_data.h``  - C tables compiled into the HIP library and the CPU oracle
  * ``nm_model.npz``     - the same tables for the Python tests

Nothing at run time parses XML or reads the reference tree: both outputs are committed.

What is restated here is the part of MuJoCo 3.1.2's *model compiler* that matters for this
model class (third-party, not vendored by the reference):

  * `inertiafromgeom=true`: every body's mass/COM/inertia from its single mesh geom at
    density 1000, with the 3.1.2 default "legacy" mesh-inertia rule (absolute tetrahedron
    volumes about the area-weighted face centroid; user_mesh.cc `mjCMesh::Process`), then
    `settotalmass=3` rescaling (mjmodel.xml:2).
  * colliding meshes (contype/conaffinity != 0: base_link and the six tibias,
    mjmodel.xml:34,47...) are replaced by their convex hulls (qhull "Qt", here through
    scipy.spatial.ConvexHull which wraps the same qhull) plus a vertex adjacency graph.
  * `geom_rbound` of a mesh geom = norm of the per-axis max |coordinate| of the mesh in
    its centred principal frame.
  * `body_invweight0` (mean diagonal of J M^-1 J^T at qpos0, translational / rotational)
    and `stat.meaninertia` (mean diagonal of M at qpos0): engine_setconst.c `set0`.

PARITY NOTE: MuJoCo is not installable here, so these tables are "parity unpinned" against
a real `MjModel`; they are pinned by the known-answer tests in tests/test_model.py (the mesh-inertia
rule on L-shaped prisms worked out by hand - star-convex: equals the exact solid; non-star-convex:
the predicted over-count -, hull graphs of an icosahedron and a cube, body_invweight0 / meaninertia
from finite-difference Jacobians) and tests/test_oracle_physics.py (total mass, symmetry,
exact-volume cross-check, M against an independent CRBA).

Usage:  python -m nightmare_rl_amd.model.compile_model /root/reference/models/nightmare_v3/mjmodel.xml
"""
from __future__ import annotations

import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- quaternions
def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def mat_to_quat(R):
    """Rotation matrix -> unit quaternion (w,x,y,z), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q /= np.linalg.norm(q)
    if q[0] < 0:
        q = -q
    return q


def axis_angle_quat(axis, angle):
    s = np.sin(angle / 2)
    return np.array([np.cos(angle / 2), axis[0] * s, axis[1] * s, axis[2] * s])


# ----------------------------------------------------------------------------- STL / mesh
def load_stl_raw(path):
    """Binary STL -> (verts float32 [n,3] de-duplicated in file order, faces int [m,3]); the RAW file floats, nothing scaled.

    MuJoCo reads the triangle list into a float32 vertex array and removes repeated vertices IN FILE ORDER (user_mesh.cc LoadSTL /
    RemoveRepeated: the i-th distinct vertex is the i-th one to appear in the triangle list; the array is compressed in place, not
    sorted). The order matters downstream: qhull's facet list - and with it the neighbour order of the hull graph, which decides WHICH
    <= 3 extra plane-mesh contacts are kept - depends on the order of its input points. (np.unique alone would hand back the vertices
    sorted by coordinate.) The de-duplication runs on the raw floats, BEFORE any scaling, as upstream's does.
    """
    raw = open(path, "rb").read()
    n = struct.unpack("<I", raw[80:84])[0]
    assert len(raw) == 84 + 50 * n, f"{path}: not a binary STL"
    rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), offset=84, count=n)
    tri = rec["v"].astype(np.float32).reshape(-1, 3)
    uniq, first, inv = np.unique(tri, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")         # distinct vertices by first occurrence
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    faces = rank[np.asarray(inv).reshape(-1)].reshape(-1, 3)
    return np.ascontiguousarray(uniq[order], dtype=np.float32), faces


def scale_verts(raw32, scale):
    """The mesh `scale` attribute as upstream applies it (user_mesh.cc `mjCMesh::Process`: `vert_[i] *= scale[j]` on a float array with a
    double scale): the product is formed in double and rounded ONCE to float32 - not float32 x float32(scale)."""
    s = np.asarray(scale, dtype=np.float64)
    assert np.prod(s) > 0, "a negative scale flips the face winding (LoadSTL); not needed for this model"
    return (raw32.astype(np.float64) * s).astype(np.float32)


def load_stl(path, scale):
    """(verts float64 [n,3] = the float32 vertices MuJoCo holds after scaling, faces): see load_stl_raw / scale_verts."""
    raw32, faces = load_stl_raw(path)
    return scale_verts(raw32, scale).astype(np.float64), faces


def mesh_props(V, F, legacy=True):
    """Volume, COM and inertia-about-COM (density 1) of a triangle mesh.

    legacy=True : MuJoCo 3.1.2 default rule - |volume| of each face pyramid about the
                  area-weighted face centroid (COM/volume), then about the COM (inertia).
    legacy=False: signed volumes ("exact" rule) - used only as a cross-check in tests.
    """
    A, B, C = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    nrm = np.cross(B - A, C - A)
    a2 = np.linalg.norm(nrm, axis=1)
    keep = a2 > 1e-30
    A, B, C, nrm, a2 = A[keep], B[keep], C[keep], nrm[keep], a2[keep]
    area = 0.5 * a2
    nhat = nrm / a2[:, None]
    cen = (A + B + C) / 3.0
    facecen = (area[:, None] * cen).sum(0) / area.sum()
    vol = np.einsum("ij,ij->i", cen - facecen, nhat) * area / 3.0
    if legacy:
        vol = np.abs(vol)
    volume = vol.sum()
    com = (vol[:, None] * (0.75 * cen + 0.25 * facecen)).sum(0) / volume
    # second moments of each pyramid (apex at COM) about the COM
    D, E, Fv = A - com, B - com, C - com
    vol2 = np.einsum("ij,ij->i", (D + E + Fv) / 3.0, nhat) * area / 3.0
    if legacy:
        vol2 = np.abs(vol2)
    P = np.zeros((3, 3))
    for a in range(3):
        for b in range(3):
            P[a, b] = (vol2 / 20.0 * (
                2 * (D[:, a] * D[:, b] + E[:, a] * E[:, b] + Fv[:, a] * Fv[:, b])
                + D[:, a] * E[:, b] + D[:, b] * E[:, a]
                + D[:, a] * Fv[:, b] + D[:, b] * Fv[:, a]
                + E[:, a] * Fv[:, b] + E[:, b] * Fv[:, a])).sum()
    inertia = np.trace(P) * np.eye(3) - P
    # MuJoCo re-accumulates the volume in the inertia loop (pyramids about the COM); that second
    # value is what the geom mass is taken from.
    return vol2.sum(), com, inertia


def hull_with_graph(V):
    """Convex hull vertices (indices into V) + adjacency lists, qhull 'Qt' via scipy."""
    from scipy.spatial import ConvexHull

    hull = ConvexHull(V, qhull_options="Qt")
    vid = np.array(sorted(set(hull.simplices.reshape(-1).tolist())))
    local = {int(g): i for i, g in enumerate(vid)}
    nbr = [[] for _ in vid]
    for simplex in hull.simplices:
        for g in simplex:
            lst = nbr[local[int(g)]]
            for h in simplex:
                if h != g and local[int(h)] not in lst:
                    lst.append(local[int(h)])
    return vid, nbr


# ----------------------------------------------------------------------------- MJCF
def fvec(s, n=None):
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None:
        assert len(v) == n
    return v


def parse_mjcf(xml_path):
    root = ET.parse(xml_path).getroot()
    comp = root.find("compiler").attrib
    opt = root.find("option").attrib
    assert comp.get("angle") == "radian" and comp.get("inertiafromgeom") == "true"
    model = {
        "settotalmass": float(comp["settotalmass"]),
        "gravity": fvec(opt["gravity"], 3),
        "timestep": float(opt["timestep"]),
        "iterations": int(opt["iterations"]),
        "noslip_iterations": int(opt["noslip_iterations"]),
        "noslip_tolerance": float(opt["noslip_tolerance"]),
        "integrator": opt["integrator"],
        "solver": opt["solver"],
    }
    assert model["integrator"] == "implicitfast" and model["solver"] == "PGS"
    meshes = {}
    for m in root.find("asset").findall("mesh"):
        meshes[m.attrib["name"]] = (m.attrib["file"], fvec(m.attrib["scale"], 3))
    bodies = [dict(name="world", parent=-1, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), joint=None, geom=None, sites=[])]
    geoms = []

    def geom_of(el, body_id):
        a = el.attrib
        g = dict(name=a.get("name"), body=body_id, type=a.get("type", "sphere"),
                 pos=fvec(a.get("pos", "0 0 0"), 3), quat=fvec(a.get("quat", "1 0 0 0"), 4),
                 mesh=a.get("mesh"), contype=int(a.get("contype", 1)), conaffinity=int(a.get("conaffinity", 1)))
        g["quat"] = g["quat"] / np.linalg.norm(g["quat"])
        geoms.append(g)
        return g

    wb = root.find("worldbody")
    for g in wb.findall("geom"):
        geom_of(g, 0)

    def walk(el, parent):
        a = el.attrib
        bid = len(bodies)
        q = fvec(a.get("quat", "1 0 0 0"), 4)
        b = dict(name=a["name"], parent=parent, pos=fvec(a.get("pos", "0 0 0"), 3), quat=q / np.linalg.norm(q), sites=[])
        bodies.append(b)
        j = el.find("joint")
        assert j is not None and len(el.findall("joint")) == 1
        ja = j.attrib
        if ja.get("type") == "free":
            b["joint"] = dict(type="free", name=ja.get("name"))
        else:
            assert fvec(ja.get("pos", "0 0 0"), 3).tolist() == [0, 0, 0]
            ax = fvec(ja["axis"], 3)
            b["joint"] = dict(type="hinge", name=ja.get("name"), axis=ax / np.linalg.norm(ax))
            assert "range" not in ja and "damping" not in ja and "armature" not in ja and "frictionloss" not in ja
        gl = el.findall("geom")
        assert len(gl) == 1
        b["geom"] = geom_of(gl[0], bid)
        for s in el.findall("site"):
            sa = s.attrib
            b["sites"].append(dict(name=sa["name"], pos=fvec(sa.get("pos", "0 0 0"), 3), size=float(sa["size"].split()[0])))
        for c in el.findall("body"):
            walk(c, bid)

    for b in wb.findall("body"):
        walk(b, 0)
    model.update(meshes=meshes, bodies=bodies, geoms=geoms)
    acts = []
    for a in root.find("actuator"):
        assert a.tag == "velocity" and a.attrib["ctrllimited"] == "true"
        acts.append(dict(joint=a.attrib["joint"], kv=float(a.attrib["kv"]), ctrlrange=fvec(a.attrib["ctrlrange"], 2)))
    model["actuators"] = acts
    model["sensors"] = [(s.tag, s.attrib["site"]) for s in root.find("sensor")]
    return model


# ----------------------------------------------------------------------------- numpy dynamics (qpos0 constants + test cross-check)
def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def kinematics_np(T, qpos):
    """World poses of all bodies. T = compiled tables dict. Returns xpos[nb,3], xmat[nb,3,3]."""
    nb = T["nbody"]
    xpos = np.zeros((nb, 3))
    xmat = np.zeros((nb, 3, 3))
    xmat[0] = np.eye(3)
    q = qpos[3:7] / np.linalg.norm(qpos[3:7])
    xpos[1] = qpos[0:3]
    xmat[1] = quat_to_mat(q)
    for b in range(2, nb):
        p = T["body_parent"][b]
        xpos[b] = xpos[p] + xmat[p] @ T["body_pos"][b]
        Rl = quat_to_mat(quat_mul(T["body_quat"][b], axis_angle_quat(T["jnt_axis"][b - 2], qpos[7 + b - 2])))
        xmat[b] = xmat[p] @ Rl
    return xpos, xmat


def mass_matrix_np(T, qpos):
    """Dense joint-space inertia M(q) [24,24] by composite rigid bodies (textbook form)."""
    nb, nv = T["nbody"], T["nv"]
    xpos, xmat = kinematics_np(T, qpos)
    # spatial inertia about the world origin, world-aligned, as 6x6 ([ang;lin] ordering)
    I6 = np.zeros((nb, 6, 6))
    for b in range(1, nb):
        Ri = xmat[b] @ quat_to_mat(T["body_iquat"][b])
        Ic = Ri @ np.diag(T["body_inertia"][b]) @ Ri.T
        c = xpos[b] + xmat[b] @ T["body_ipos"][b]
        m = T["body_mass"][b]
        S = skew(c)
        I6[b, :3, :3] = Ic - m * S @ S
        I6[b, :3, 3:] = m * S
        I6[b, 3:, :3] = -m * S
        I6[b, 3:, 3:] = m * np.eye(3)
    # motion subspace columns about the world origin
    Sm = np.zeros((nv, 6))
    dof_body = np.zeros(nv, dtype=int)
    for k in range(3):
        Sm[k, 3 + k] = 1.0
        a = xmat[1][:, k]
        Sm[3 + k, :3] = a
        Sm[3 + k, 3:] = np.cross(a, -xpos[1])
        dof_body[k] = dof_body[3 + k] = 1
    for j in range(nv - 6):
        b = 2 + j
        a = xmat[b] @ T["jnt_axis"][j]
        Sm[6 + j, :3] = a
        Sm[6 + j, 3:] = np.cross(a, -xpos[b])
        dof_body[6 + j] = b
    Ic = I6.copy()
    for b in range(nb - 1, 1, -1):
        Ic[T["body_parent"][b]] += Ic[b]
    # ancestors-or-self test
    anc = np.zeros((nb, nb), dtype=bool)
    for b in range(nb):
        a = b
        while a >= 0:
            anc[b, a] = True
            a = T["body_parent"][a] if a > 0 else -1
    M = np.zeros((nv, nv))
    for i in range(nv):
        F = Ic[dof_body[i]] @ Sm[i]
        for j in range(nv):
            if anc[dof_body[i], dof_body[j]] and (dof_body[j] != dof_body[i] or True):
                if anc[dof_body[i], dof_body[j]]:
                    M[i, j] = M[j, i] = Sm[j] @ F
    return M, xpos, xmat, Sm, dof_body


def body_jac_com_np(T, qpos, b):
    """6 x nv Jacobian ([lin; ang]) of body b's COM (mj_jacBodyCom)."""
    M, xpos, xmat, Sm, dof_body = mass_matrix_np(T, qpos)
    c = xpos[b] + xmat[b] @ T["body_ipos"][b]
    nv = T["nv"]
    J = np.zeros((6, nv))
    a = b
    chain = set()
    while a > 0:
        chain.add(a)
        a = T["body_parent"][a]
    for i in range(nv):
        if dof_body[i] in chain:
            w, v0 = Sm[i, :3], Sm[i, 3:]
            J[:3, i] = v0 + np.cross(w, c)
            J[3:, i] = w
    return J, M


# ----------------------------------------------------------------------------- compile
def compile_model(xml_path):
    mj = parse_mjcf(xml_path)
    mdir = os.path.dirname(xml_path)
    bodies = mj["bodies"]
    nb = len(bodies)
    assert nb == 20
    T = dict(nbody=nb, nq=25, nv=24, nu=18)
    T["body_parent"] = np.array([max(b["parent"], 0) for b in bodies], dtype=np.int32)
    T["body_pos"] = np.array([b["pos"] for b in bodies])
    T["body_quat"] = np.array([b["quat"] for b in bodies])
    T["jnt_axis"] = np.array([b["joint"]["axis"] for b in bodies[2:]])
    names = [b["name"] for b in bodies]
    # topology the kernels are specialised for: base + 6 x (coxa, femur, tibia)
    for leg in range(6):
        for k, part in enumerate(("coxa", "femur", "tibia")):
            b = 2 + 3 * leg + k
            assert names[b] == f"leg_{leg + 1}_{part}", names[b]
            assert T["body_parent"][b] == (1 if k == 0 else b - 1)

    mass = np.zeros(nb)
    ipos = np.zeros((nb, 3))
    iquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
    inertia = np.zeros((nb, 3))
    mass_exact = np.zeros(nb)
    col = []  # colliding mesh geoms
    density = 1000.0
    for b in range(1, nb):
        g = bodies[b]["geom"]
        fn, scale = mj["meshes"][g["mesh"]]
        raw32, F = load_stl_raw(os.path.join(mdir, fn))
        V = scale_verts(raw32, scale).astype(np.float64)       # what Process() computes mass / inertia on
        vol, com, I = mesh_props(V, F, legacy=True)
        vol_e, _, _ = mesh_props(V, F, legacy=False)
        Rg = quat_to_mat(g["quat"])
        mass[b] = density * vol
        mass_exact[b] = density * vol_e
        ipos[b] = g["pos"] + Rg @ com
        w, U = np.linalg.eigh(I)
        order = np.argsort(-w)  # descending, as mju_eig3
        w, U = w[order], U[:, order]
        if np.linalg.det(U) < 0:
            U[:, 2] = -U[:, 2]
        iquat[b] = mat_to_quat(Rg @ U)
        inertia[b] = density * w
        if g["contype"] or g["conaffinity"]:
            # Upstream's order (user_mesh.cc mjCMesh::Compile, 3.1.2 [3P]): load + RemoveRepeated -> MakeGraph (qhull "Qt" on the vertex
            # array AS LOADED: raw float32 file values cast to double, no scale, no centring) -> CopyGraph / MakeNormal -> Process()
            # ("scale, center, orient, compute mass and inertia"). The graph holds point ids, so the later scaling does not touch it -
            # but qhull's facet list, its Qt triangulation and even WHICH nearly coplanar points count as hull vertices depend on the
            # last bits of its input, so the hull must be built from the unscaled values.
            vid, nbr = hull_with_graph(raw32.astype(np.float64))
            # mesh_vert as the model stores it: float32, re-centred at the COM and rotated into the principal frame, each step rounded to
            # float32 (Process(): `vert -= CoM`, then `vert = eigvec' vert`); the geom frame absorbs COM and principal rotation
            # (mjCGeom::Compile: geom pos/quat composed with the mesh's pos_volume / quat_volume), so at run time a hull vertex in the BODY
            # frame is geom_pos + R(geom_quat) mesh_vert, evaluated in double.
            Vc32 = (V - com).astype(np.float32)
            Vp32 = (Vc32.astype(np.float64) @ U).astype(np.float32)
            Vp = Vp32.astype(np.float64)
            gpos = g["pos"] + Rg @ com
            gmat = Rg @ U
            rb = np.linalg.norm(np.max(np.abs(Vp), axis=0))     # geom_rbound: norm of the centred principal frame's half extents
            Vb = gpos + Vp[vid] @ gmat.T                         # hull vertices in the BODY frame
            # oriented bounding box of the hull in the mesh's principal frame -> body frame (conservative pair cull on the GPU)
            Hp = Vp[vid]
            lo, hi = Hp.min(axis=0), Hp.max(axis=0)
            obb_c = gpos + gmat @ ((lo + hi) / 2)
            obb_ax = gmat.T                # rows = box axes in the body frame
            obb_h = (hi - lo) / 2
            col.append(dict(body=b, name=g["name"], verts=Vb, nbr=nbr, rbound=rb, center=ipos[b].copy(), obb_c=obb_c, obb_ax=obb_ax, obb_h=obb_h,
                            contype=g["contype"], conaffinity=g["conaffinity"], mesh_vert=Vp32, vid=vid, gpos=gpos, gquat=mat_to_quat(gmat),
                            nmeshvert=len(V)))
    scale = mj["settotalmass"] / mass.sum()
    mass *= scale
    inertia *= scale
    mass_exact *= mj["settotalmass"] / mass_exact.sum()
    T.update(body_mass=mass, body_ipos=ipos, body_iquat=iquat, body_inertia=inertia, body_mass_exactrule=mass_exact)

    # collision tables (floor plane z=0 is geom 0; contype 1 / conaffinity 1)
    assert [c["body"] for c in col] == [1, 4, 7, 10, 13, 16, 19]
    for c in col:  # every colliding mesh must collide with the floor (1/1)
        assert (c["contype"] & 1) or (c["conaffinity"] & 1)
    T["ncol"] = len(col)
    T["col_body"] = np.array([c["body"] for c in col], dtype=np.int32)
    T["col_rbound"] = np.array([c["rbound"] for c in col])
    T["col_center"] = np.array([c["center"] for c in col])
    T["col_obb_center"] = np.array([c["obb_c"] for c in col])
    T["col_obb_axes"] = np.array([c["obb_ax"].reshape(9) for c in col])
    T["col_obb_half"] = np.array([c["obb_h"] for c in col])
    T["col_nvert"] = np.array([len(c["verts"]) for c in col], dtype=np.int32)
    T["col_vadr"] = np.concatenate([[0], np.cumsum(T["col_nvert"])[:-1]]).astype(np.int32)
    T["hull_vert"] = np.concatenate([c["verts"] for c in col])
    maxnbr = max(len(l) for c in col for l in c["nbr"])
    nbr_tab = -np.ones((len(T["hull_vert"]), maxnbr), dtype=np.int32)  # LOCAL vertex ids within the mesh
    row = 0
    for c in col:
        for l in c["nbr"]:
            nbr_tab[row, : len(l)] = l
            row += 1
    T["hull_nbr"] = nbr_tab
    T["hull_maxnbr"] = maxnbr
    # what tests/test_mujoco_crosscheck.py compares with a real MjModel (npz only; the kernels use the body-frame tables above):
    # the hull vertices as mesh_vert rows (float32, centred principal frame), their point ids in the de-duplicated mesh, the geom frame
    T["col_mesh_nvert"] = np.array([c["nmeshvert"] for c in col], dtype=np.int32)
    T["hull_mesh_vert"] = np.concatenate([c["mesh_vert"][c["vid"]] for c in col]).astype(np.float32)
    T["hull_point_id"] = np.concatenate([c["vid"] for c in col]).astype(np.int32)
    T["col_geom_pos"] = np.array([c["gpos"] for c in col])
    T["col_geom_quat"] = np.array([c["gquat"] for c in col])

    # touch-sensor sites: 6 tibia (r=10), 6 foot (r=.007), base (r=10)  (mjmodel.xml:156-170)
    site = {s["name"]: (b, s) for b, bd in enumerate(bodies) for s in bd["sites"]}
    sens = mj["sensors"]
    assert [s[1] for s in sens] == [f"leg_{i}_tibia" for i in range(1, 7)] + [f"leg_{i}_foot" for i in range(1, 7)] + ["base_link"]
    T["sens_body"] = np.array([site[s[1]][0] for s in sens], dtype=np.int32)
    T["sens_pos"] = np.array([site[s[1]][1]["pos"] for s in sens])
    T["sens_radius"] = np.array([site[s[1]][1]["size"] for s in sens])

    # actuators: velocity servo kv on hinge j, ctrlrange
    jn = [b["joint"]["name"] for b in bodies[2:]]
    assert [a["joint"] for a in mj["actuators"]] == jn
    assert all(a["kv"] == mj["actuators"][0]["kv"] for a in mj["actuators"])
    T["kv"] = mj["actuators"][0]["kv"]
    T["ctrl_max"] = float(mj["actuators"][0]["ctrlrange"][1])
    T["timestep"] = mj["timestep"]
    T["gravity"] = mj["gravity"]
    T["iterations"] = mj["iterations"]
    T["noslip_iterations"] = mj["noslip_iterations"]
    T["noslip_tolerance"] = mj["noslip_tolerance"]
    # defaults not written in the XML (MuJoCo 3.1.2 defaults)
    T["solref"] = np.array([0.02, 1.0])
    T["solimp"] = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
    T["friction"] = 1.0
    T["impratio"] = 1.0
    T["tolerance"] = 1e-8
    qpos0 = np.zeros(25)
    qpos0[0:3] = bodies[1]["pos"]
    qpos0[3:7] = bodies[1]["quat"]
    T["qpos0"] = qpos0

    # constants evaluated at qpos0 (engine_setconst.c set0)
    M, *_ = mass_matrix_np(T, qpos0)
    T["meaninertia"] = float(np.mean(np.diag(M)))
    Minv = np.linalg.inv(M)
    invw = np.zeros((nb, 2))
    for b in range(1, nb):
        J, _ = body_jac_com_np(T, qpos0, b)
        A = J @ Minv @ J.T
        invw[b, 0] = np.trace(A[:3, :3]) / 3
        invw[b, 1] = np.trace(A[3:, 3:]) / 3
    T["body_invweight0"] = invw
    return T


# ----------------------------------------------------------------------------- emit
def c_array(name, arr, ctype):
    arr = np.asarray(arr)
    flat = arr.reshape(-1)
    dims = "".join(f"[{d}]" for d in arr.shape)
    if ctype == "int":
        body = ", ".join(str(int(x)) for x in flat)
    else:
        body = ", ".join(repr(float(x)) for x in flat)
    lines = []
    cur = ""
    for tok in body.split(", "):
        if len(cur) + len(tok) > 110:
            lines.append(cur)
            cur = ""
        cur += tok + ", "
    lines.append(cur.rstrip(", "))
    return f"static const {ctype} {name}{dims} = {{\n  " + "\n  ".join(lines) + "\n};\n"


def emit_header(T, path):
    out = []
    out.append("/* GENERATED by nightmare_rl_amd/model/compile_model.py from the Nightmare-v3 robot description\n"
               " * (reference models/nightmare_v3/mjmodel.xml + STL meshes). Data only - do not edit.\n"
               " * Layout: body 0 world, 1 base_link, 2+3L+{0,1,2} = leg L coxa/femur/tibia. dof 0-5 free joint,\n"
               " * 6+3L+k hinge of body 2+3L+k. */\n#ifndef NM_MODEL_DATA_H\n#define NM_MODEL_DATA_H\n")
    for k in ("nbody", "nq", "nv", "nu", "ncol", "hull_maxnbr", "iterations", "noslip_iterations"):
        out.append(f"#define NM_{k.upper()} {int(T[k])}\n")
    out.append(f"#define NM_NHULLVERT {len(T['hull_vert'])}\n")
    for k in ("kv", "ctrl_max", "timestep", "noslip_tolerance", "friction", "impratio", "tolerance", "meaninertia"):
        out.append(f"#define NM_{k.upper()} {float(T[k])!r}\n")
    out.append("#ifndef NM_NO_TABLES\n")
    for k in ("body_parent", "col_body", "col_nvert", "col_vadr", "sens_body", "hull_nbr"):
        out.append(c_array("nm_" + k, T[k], "int"))
    for k in ("body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia", "jnt_axis", "col_rbound",
              "col_center", "col_obb_center", "col_obb_axes", "col_obb_half", "hull_vert", "sens_pos", "sens_radius", "gravity", "solref", "solimp", "qpos0", "body_invweight0"):
        out.append(c_array("nm_" + k, T[k], "double"))
    out.append("#endif /* NM_NO_TABLES */\n#endif\n")
    with open(path, "w") as f:
        f.write("".join(out))


def main(argv):
    xml = argv[1] if len(argv) > 1 else "/root/reference/models/nightmare_v3/mjmodel.xml"
    T = compile_model(xml)
    emit_header(T, os.path.join(HERE, "nm_model_data.h"))
    np.savez_compressed(os.path.join(HERE, "nm_model.npz"), **{k: np.asarray(v) for k, v in T.items()})
    print("total mass", T["body_mass"].sum())
    print("mass base/coxa/femur/tibia (legacy rule):", T["body_mass"][[1, 2, 3, 4]])
    print("mass base/coxa/femur/tibia (exact rule): ", T["body_mass_exactrule"][[1, 2, 3, 4]])
    print("hull verts", T["col_nvert"], "max nbr", T["hull_maxnbr"], "rbound", T["col_rbound"])
    print("meaninertia", T["meaninertia"], "invweight0 base/tibia", T["body_invweight0"][[1, 4]])


def load_tables():
    """Load the committed tables (no reference tree needed)."""
    z = np.load(os.path.join(HERE, "nm_model.npz"))
    T = {}
    for k in z.files:
        v = z[k]
        T[k] = v.item() if v.shape == () else v
    for k in ("nbody", "nq", "nv", "nu", "ncol"):
        T[k] = int(T[k])
    return T


if __name__ == "__main__":
    main(sys.argv)
