"""Stage-by-stage comparison of a compiled `MjModel` (real MuJoCo, wherever `import mujoco` works) with this repo's model tables
(nightmare_rl_amd/model/nm_model.npz). Test infrastructure: used by tests/test_mujoco_crosscheck.py against the real thing and by
tests/test_model.py against a synthetic stand-in assembled from the tables themselves (so the comparer has run before it matters).

`compare_model(m, T)` returns a list of findings, one string per stage that differs, each naming the stage of MuJoCo's model compiler
it points at (user_mesh.cc LoadSTL / RemoveRepeated / MakeGraph / Process, user_objects.cc mjCGeom::Compile, engine_setconst.c) -
every stage is checked, not only the first that fails. Only attribute names of the public `MjModel` struct are used:
body_mass / body_inertia / body_ipos / body_iquat / body_invweight0, geom_bodyid / geom_dataid / geom_pos / geom_quat / geom_rbound,
mesh_vertadr / mesh_vertnum / mesh_vert / mesh_graphadr / mesh_graph, stat.meaninertia.

mesh_graph layout (mjmodel.h): at mesh_graphadr[i]: numvert, numface, vert_edgeadr[numvert], vert_globalid[numvert],
edge_localid[numvert + 3 numface] (per vertex: local neighbour ids, terminated by -1), face_globalid[3 numface]."""
import numpy as np


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def graph_of(m, mesh):
    """(global point ids [numvert], neighbour lists as GLOBAL point ids in edge order) of mesh `mesh`'s convex hull graph, or None."""
    adr = int(m.mesh_graphadr[mesh])
    if adr < 0:
        return None
    G = np.asarray(m.mesh_graph)
    nv, nf = int(G[adr]), int(G[adr + 1])
    edgeadr = G[adr + 2: adr + 2 + nv]
    gid = G[adr + 2 + nv: adr + 2 + 2 * nv]
    edges = G[adr + 2 + 2 * nv: adr + 2 + 2 * nv + nv + 3 * nf]
    nbr = []
    for i in range(nv):
        row = []
        k = int(edgeadr[i])
        while edges[k] >= 0:
            row.append(int(gid[edges[k]]))
            k += 1
        nbr.append(row)
    return np.asarray(gid, dtype=np.int64), nbr


def compare_model(m, T):
    out = []

    def close(name, a, b, stage, **kw):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        if a.shape != b.shape:
            out.append(f"{name}: shape {a.shape} vs tables {b.shape}  [{stage}]")
        elif not np.allclose(a, b, **kw):
            out.append(f"{name}: max |diff| {np.abs(a - b).max():.3e} (tolerance {kw})  [{stage}]")

    # ---- mass model: legacy mesh inertia + settotalmass (mjCMesh::Process, mjCModel::Compile)
    close("body_mass", m.body_mass, T["body_mass"], "Process(): legacy volume rule / settotalmass", rtol=1e-6, atol=0)
    close("body_ipos", m.body_ipos, T["body_ipos"], "Process(): COM", rtol=0, atol=1e-7)
    nb = int(T["nbody"])
    Im = np.array([quat_to_mat(m.body_iquat[b]) @ np.diag(m.body_inertia[b]) @ quat_to_mat(m.body_iquat[b]).T for b in range(1, nb)])
    It = np.array([quat_to_mat(T["body_iquat"][b]) @ np.diag(T["body_inertia"][b]) @ quat_to_mat(T["body_iquat"][b]).T for b in range(1, nb)])
    close("body inertia tensor in the body frame (iquat-invariant)", Im, It, "Process(): inertia about the COM", rtol=1e-5, atol=1e-12)
    close("body_inertia (principal moments)", m.body_inertia, T["body_inertia"], "mju_eig3 ordering", rtol=1e-5, atol=1e-12)
    close("body_invweight0", m.body_invweight0, T["body_invweight0"], "engine_setconst.c set0", rtol=1e-5, atol=0)
    if abs(m.stat.meaninertia - float(T["meaninertia"])) > 1e-6 * float(T["meaninertia"]):
        out.append(f"stat.meaninertia {m.stat.meaninertia!r} vs {float(T['meaninertia'])!r}  [engine_setconst.c set0]")

    # ---- the seven colliding meshes: vertex array, hull graph, geom frame, bounding radius
    gbody = np.asarray(m.geom_bodyid)
    for g in range(int(T["ncol"])):
        b = int(T["col_body"][g])
        gids = np.nonzero(gbody == b)[0]
        if len(gids) != 1:
            out.append(f"body {b}: {len(gids)} geoms, expected one mesh geom")
            continue
        gi = int(gids[0])
        mesh = int(m.geom_dataid[gi])
        tag = f"col mesh {g} (body {b}, geom {gi}, mesh {mesh})"
        nv, va = int(T["col_nvert"][g]), int(T["col_vadr"][g])
        if int(m.mesh_vertnum[mesh]) != int(T["col_mesh_nvert"][g]):
            out.append(f"{tag}: mesh_vertnum {int(m.mesh_vertnum[mesh])} vs {int(T['col_mesh_nvert'][g])} distinct STL vertices  [LoadSTL / RemoveRepeated]")
        gr = graph_of(m, mesh)
        if gr is None:
            out.append(f"{tag}: no convex hull graph in the model (mesh_graphadr < 0)  [needhull / MakeGraph]")
            continue
        gid, nbr_m = gr
        pid = np.asarray(T["hull_point_id"][va:va + nv], dtype=np.int64)          # our hull vertices as point ids, ascending
        if len(gid) != nv or set(gid.tolist()) != set(pid.tolist()):
            only_m = sorted(set(gid.tolist()) - set(pid.tolist()))
            only_t = sorted(set(pid.tolist()) - set(gid.tolist()))
            out.append(f"{tag}: hull vertex SET differs: {len(gid)} vs {nv} vertices, {len(only_m)} only in MuJoCo {only_m[:6]}, {len(only_t)} only in the tables {only_t[:6]}"
                       "  [MakeGraph: qhull input (raw unscaled floats?) / options]")
        # vertex coordinates: mesh_vert rows of the hull vertices, taken to the BODY frame through the geom frame (independent of the
        # sign / order conventions of the principal axes), against the tables' body-frame hull vertices
        R = quat_to_mat(m.geom_quat[gi])
        mv = np.asarray(m.mesh_vert, dtype=np.float64).reshape(-1, 3)[int(m.mesh_vertadr[mesh]):int(m.mesh_vertadr[mesh]) + int(m.mesh_vertnum[mesh])]
        common = [p for p in pid.tolist() if p in set(gid.tolist()) and p < len(mv)]
        if common:
            row_of = {int(p): i for i, p in enumerate(pid.tolist())}
            Pm = np.asarray(m.geom_pos[gi]) + mv[common] @ R.T
            Pt = T["hull_vert"][va:va + nv][[row_of[p] for p in common]]
            if not np.allclose(Pm, Pt, rtol=0, atol=1e-7):
                out.append(f"{tag}: hull vertex positions in the body frame differ by up to {np.abs(Pm - Pt).max():.3e} m  [Process(): scale / centre / orient roundings, or geom frame]")
            # mesh_vert itself (float32, centred principal frame) up to the sign of each principal axis
            Mt = np.asarray(T["hull_mesh_vert"][va:va + nv], dtype=np.float64)[[row_of[p] for p in common]]
            if not np.allclose(np.abs(mv[common]), np.abs(Mt), rtol=0, atol=1e-7):
                out.append(f"{tag}: |mesh_vert| rows differ by up to {np.abs(np.abs(mv[common]) - np.abs(Mt)).max():.3e}  [Process(): principal frame]")
        # adjacency: same neighbours in the same ORDER (the order decides which <= 3 extra plane-mesh contacts survive the cap)
        local_of = {int(p): i for i, p in enumerate(pid.tolist())}
        order = sets = 0
        for i, p in enumerate(gid.tolist()):
            if p not in local_of:
                continue
            mine = [int(pid[k]) for k in T["hull_nbr"][va + local_of[p]] if k >= 0]
            order += mine != nbr_m[i]
            sets += set(mine) != set(nbr_m[i])
        if order or sets:
            out.append(f"{tag}: hull graph: {order} vertices with another neighbour ORDER, {sets} with another neighbour SET  [MakeGraph: facet order / Qt triangulation]")
        if gid.tolist() != pid.tolist():
            # not a finding by itself: the tables number hull vertices by point id, upstream by qhull's vertex list - observable only through
            # exact support ties; reported so that a tie-break difference can be traced
            out.append(f"{tag}: note: hull vertex NUMBERING differs (qhull vertex-list order vs ascending point id) - affects exact ties only")
        close(f"{tag} geom_rbound", m.geom_rbound[gi], T["col_rbound"][g], "mjCGeom::Compile GetRBound (aabb of the centred principal frame)", rtol=1e-6, atol=0)
        close(f"{tag} geom_pos", m.geom_pos[gi], T["col_geom_pos"][g], "mjCGeom::Compile: geom frame x mesh pos_volume", rtol=0, atol=1e-7)
        Rt = quat_to_mat(T["col_geom_quat"][g])
        if not np.allclose(np.abs(R.T @ Rt), np.eye(3), atol=1e-5):
            out.append(f"{tag}: geom_quat is another frame (|R_mj' R_tab| != I)  [mjCGeom::Compile: geom frame x mesh quat_volume]")
    return out


def synthetic_mjmodel(T, hull_numbering="tables"):
    """A stand-in with MjModel's attribute layout, assembled from the tables (colliding meshes only; one mesh per colliding geom): lets
    the comparer run without MuJoCo. hull_numbering='reversed' numbers each hull's vertices the other way round (like a qhull vertex list
    that is not sorted by point id)."""
    from types import SimpleNamespace
    nb = int(T["nbody"])
    ncol = int(T["ncol"])
    m = SimpleNamespace(body_mass=T["body_mass"].copy(), body_inertia=T["body_inertia"].copy(), body_ipos=T["body_ipos"].copy(),
                        body_iquat=T["body_iquat"].copy(), body_invweight0=T["body_invweight0"].copy(),
                        stat=SimpleNamespace(meaninertia=float(T["meaninertia"])))
    m.geom_bodyid = np.r_[0, np.arange(1, nb)].astype(np.int32)                        # floor + one geom per body
    m.geom_dataid = -np.ones(nb, dtype=np.int32)
    m.geom_pos = np.zeros((nb, 3))
    m.geom_quat = np.tile([1.0, 0, 0, 0], (nb, 1))
    m.geom_rbound = np.zeros(nb)
    vert, vertadr, vertnum, graph, graphadr = [], [], [], [], []
    for g in range(ncol):
        b = int(T["col_body"][g])
        nv, va = int(T["col_nvert"][g]), int(T["col_vadr"][g])
        gi = b                                                                       # geom id of body b in this stand-in
        m.geom_dataid[gi] = g
        m.geom_pos[gi] = T["col_geom_pos"][g]
        m.geom_quat[gi] = T["col_geom_quat"][g]
        m.geom_rbound[gi] = T["col_rbound"][g]
        nmv = int(T["col_mesh_nvert"][g])
        mv = np.zeros((nmv, 3), dtype=np.float32)
        pid = np.asarray(T["hull_point_id"][va:va + nv])
        mv[pid] = T["hull_mesh_vert"][va:va + nv]
        vertadr.append(sum(len(v) for v in vert))
        vertnum.append(nmv)
        vert.append(mv)
        perm = np.arange(nv) if hull_numbering == "tables" else np.arange(nv)[::-1]   # graph vertex i = tables' local vertex perm[i]
        inv = np.empty(nv, dtype=np.int64)
        inv[perm] = np.arange(nv)
        rows = [[int(inv[k]) for k in T["hull_nbr"][va + perm[i]] if k >= 0] for i in range(nv)]
        nedge = sum(len(r) for r in rows)
        nf = -(-nedge // 3)                                                          # a triangulated hull has 3 numface directed edges
        edgeadr, edges = [], []
        for r in rows:
            edgeadr.append(len(edges))
            edges += r + [-1]
        edges += [-1] * (nv + 3 * nf - len(edges))
        graphadr.append(len(graph))
        graph += [nv, nf] + edgeadr + [int(pid[perm[i]]) for i in range(nv)] + edges + [0] * (3 * nf)
    m.mesh_vert = np.concatenate(vert)
    m.mesh_vertadr = np.array(vertadr, dtype=np.int32)
    m.mesh_vertnum = np.array(vertnum, dtype=np.int32)
    m.mesh_graph = np.array(graph, dtype=np.int32)
    m.mesh_graphadr = np.array(graphadr, dtype=np.int32)
    return m
