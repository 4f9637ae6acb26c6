"""Multi-rank (x-slab) path: shared-DOF partial sums exchanged per stage, identical state on all
sharers.  (1) CPU, world_size 2 over gloo: the partition data (slab meshes, interface planes,
exterior-facet tags) and the exchange algorithm, with the oracle as the per-rank operator, against
the single-rank oracle.  (2) GPU: the library's own pack / ordered-sum / stage kernels with the
in-process transport (all slabs on one MI355X), against the single-rank oracle."""
import os
import socket

import numpy as np
import pytest

import fenicsxfus_amd as fa
from util import Problem

P, N_GLOBAL, HI = 4, (6, 3, 3), [0.024, 0.012, 0.012]
F0, P0, S0 = 0.5e6, 60000.0, 1500.0
NSTEPS = 6


def material(mesh):
    nc = mesh.num_cells
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    cx = mesh.cell_centroids()[:, 0]
    sel = (cx > 0.4 * HI[0]) & (cx < 0.6 * HI[0])   # bone slab straddling the 2-rank interface
    c[sel], rho[sel] = 2800.0, 1850.0
    return c, rho


def dt_value():
    return 0.5 * (HI[0] / N_GLOBAL[0]) / (2800.0 * P**2)


def single_rank_reference(orc):
    pr = Problem(orc, N_GLOBAL, P, hi=HI, perturb=0.1)
    c, rho = material(pr.mesh)
    tags = fa.tag_box_boundary(pr.mesh)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    dt = dt_value()
    # tf a hair past NSTEPS*dt: NSTEPS full steps (+ a ~1e-12*dt remainder step, far below tolerance)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, F0, P0, S0, 0.0, NSTEPS * dt * (1 + 1e-12), dt, u, v)
    return pr, m, u, v


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _exchange_sum(dist, rank, V, vec):
    """Every sharer adds the partials of an interface plane in ascending rank order."""
    import torch

    out = vec.copy()
    recv = {}
    for nb, idx in V.neighbours:
        send = torch.from_numpy(vec[idx].copy())
        buf = torch.empty_like(send)
        if rank < nb:
            dist.send(send, nb), dist.recv(buf, nb)
        else:
            dist.recv(buf, nb), dist.send(send, nb)
        recv[nb] = (idx, buf.numpy())
    for nb, (idx, buf) in recv.items():
        out[idx] = (buf + vec[idx]) if nb < rank else (vec[idx] + buf)
    return out


def _gloo_worker(rank, size, port, q):
    import torch.distributed as dist

    import oracle as orc

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    pr = Problem(orc, N_GLOBAL, P, hi=HI, perturb=0.1, rank=rank, size=size)
    c, rho = material(pr.mesh)
    tags = fa.tag_box_boundary(pr.mesh)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, tags)
    m, src, absb = (_exchange_sum(dist, rank, pr.V, a) for a in (m, src, absb))
    # host restatement of the library's stage loop (fusmi.hip stage_begin / stage_end)
    n = pr.ndofs
    u0, v0 = np.zeros(n), np.zeros(n)
    u_, v_, un, vn = (np.zeros(n) for _ in range(4))
    dt, t = dt_value(), 0.0
    a_r, b_r, c_r = [0, .5, .5, 1, 0], [1 / 6, 1 / 3, 1 / 3, 1 / 6], [0, .5, .5, 1]
    for _ in range(NSTEPS):
        for i in range(4):
            us, vs = (u0, v0) if i == 0 else (un, vn)
            b = _exchange_sum(dist, rank, pr.V, pr.K(us, coeff))
            tn = t + c_r[i] * dt
            win = 0.5 * (1 - np.cos(F0 * np.pi * tn / 4.0)) if tn < 4.0 / F0 else 1.0
            g = win * P0 * 2 * np.pi * F0 / S0 * np.cos(2 * np.pi * F0 * tn)
            b = b + g * src - absb * vs
            kv = b / m
            adt, bdt = dt * a_r[i + 1], dt * b_r[i]
            if i == 0:
                u_, v_, un, vn = v0 * bdt + u0, kv * bdt + v0, v0 * adt + u0, kv * adt + v0
            elif i == 3:
                u0, v0 = vn * bdt + u_, kv * bdt + v_
            else:
                u_, v_, un, vn = vn * bdt + u_, kv * bdt + v_, vn * adt + u0, kv * adt + v0
        t += dt
    q.put((rank, pr.V.global_offset, m, u0, v0))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_cpu(orc):
    import torch.multiprocessing as mp

    ref, m_ref, u_ref, v_ref = single_rank_reference(orc)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    planes = {}
    for rank, off, m, u, v in res:
        n = len(u)
        assert np.abs(m - m_ref[off:off + n]).max() < 1e-14 * np.abs(m_ref).max()
        assert np.abs(u - u_ref[off:off + n]).max() < 1e-10 * np.abs(u_ref).max()
        assert np.abs(v - v_ref[off:off + n]).max() < 1e-10 * np.abs(v_ref).max()
        planes[rank] = (off, u)
    # the shared plane is bit-identical on both sharers
    off1 = planes[1][0]
    plane = len(planes[0][1]) - off1
    assert np.array_equal(planes[0][1][off1:], planes[1][1][:plane])


@pytest.mark.gpu
@pytest.mark.parametrize("geometry", ["stream", None])   # streamed G / default (G recomputed from the cell maps)
@pytest.mark.parametrize("size", [2, 3])
def test_slabs_in_process_gpu(orc, size, geometry):
    ref, m_ref, u_ref, v_ref = single_rank_reference(orc)
    ctxs = [fa.Context(0, geometry=geometry) for _ in range(size)]
    fa.Context.init_local_group(ctxs)
    models, offs = [], []
    dt = dt_value()
    for r in range(size):
        mesh = fa.BoxMesh([0, 0, 0], HI, N_GLOBAL, rank=r, size=size, perturb=0.1)
        V = fa.FunctionSpace(mesh, P)
        c, rho = material(mesh)
        models.append(fa.LinearSpectralExplicit(mesh, fa.tag_box_boundary(mesh), P, c, rho, F0, P0, S0, 4, dt, V=V,
                                                ctx=ctxs[r]))
        offs.append(V.global_offset)
    fa.group_finish_setup(models)
    for m in models:
        m.init()
    fa.group_rk4_steps(models, 0.0, dt, NSTEPS)
    us = []
    for r, mdl in enumerate(models):
        n = mdl.data.ndofs
        assert np.abs(mdl.mass_vector() - m_ref[offs[r]:offs[r] + n]).max() < 1e-14 * np.abs(m_ref).max()
        u = mdl.u_sol().x.array
        us.append(u)
        assert np.abs(u - u_ref[offs[r]:offs[r] + n]).max() < 1e-10 * np.abs(u_ref).max()
        assert np.abs(mdl.v_n.x.array - v_ref[offs[r]:offs[r] + n]).max() < 1e-10 * np.abs(v_ref).max()
    for r in range(size - 1):       # interface planes bit-identical on both sharers
        plane = len(us[r]) - (offs[r + 1] - offs[r])
        assert np.array_equal(us[r][-plane:], us[r + 1][:plane])
    # the ranks' local parts of the squared L2 norm add up to the single-rank integral, and the
    # global smallest cell size is the minimum of the local ones (main.cpp:60-68, 151-157)
    w = ref.M(np.ones(ref.ndofs))
    assert abs(sum(mdl.data.norm2(u) for mdl, u in zip(models, us)) - w @ u_ref**2) < 1e-9 * (w @ u_ref**2)
    X, dm = ref.mesh.geometry.x, ref.mesh.geometry.dofmap
    h = min(max(np.linalg.norm(X[a] - X[b]) for a in cell for b in cell) for cell in dm)
    assert abs(min(mdl.data.hmin() for mdl in models) - h) < 1e-14
    for mdl in models:
        mdl.close()
    for c in ctxs:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [0, 1])
def test_slabs_overlap_blocks_option_gpu(orc, overlap):
    """Option "overlap_blocks": interface blocks launched first, the remaining blocks after the pack.
    Elongated box so that the 1 x 2 x 2 blocks next to the cut are a strict subset of each rank's
    blocks; both settings must give the single-rank oracle's state."""
    n, hi, size, nsteps = (12, 2, 2), [0.048, 0.008, 0.008], 2, 5
    pr = Problem(orc, n, P, hi=hi)
    nc = pr.mesh.num_cells
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, fa.tag_box_boundary(pr.mesh))
    dt = 0.5 * (hi[0] / n[0]) / (1500.0 * P**2)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, F0, P0, S0, 0.0, nsteps * dt * (1 + 1e-12), dt, u, v)
    ctxs = [fa.Context(0, block_elems=4) for _ in range(size)]
    for cx in ctxs:
        cx.set_option("overlap_blocks", overlap)
    fa.Context.init_local_group(ctxs)
    models, offs = [], []
    for r in range(size):
        mesh = fa.BoxMesh([0, 0, 0], hi, n, rank=r, size=size)
        V = fa.FunctionSpace(mesh, P)
        k = mesh.num_cells
        models.append(fa.LinearSpectralExplicit(mesh, fa.tag_box_boundary(mesh), P, np.full(k, 1500.0),
                                                np.full(k, 1000.0), F0, P0, S0, 4, dt, V=V, ctx=ctxs[r]))
        assert models[-1].data.info()["nblocks"] == 6
        offs.append(V.global_offset)
    fa.group_finish_setup(models)
    for mdl in models:
        mdl.init()
    fa.group_rk4_steps(models, 0.0, dt, nsteps)
    for r, mdl in enumerate(models):
        k = mdl.data.ndofs
        assert np.abs(u).max() > 0
        assert np.abs(mdl.u_sol().x.array - u[offs[r]:offs[r] + k]).max() < 1e-10 * np.abs(u).max()
        assert np.abs(mdl.v_n.x.array - v[offs[r]:offs[r] + k]).max() < 1e-10 * np.abs(v).max()
    for mdl in models:
        mdl.close()
    for cx in ctxs:
        cx.close()


@pytest.mark.gpu
def test_westervelt_slabs_in_process_gpu(orc):
    # the nonlinear model across 2 slabs: m0 and the M(nlin1) diagonal are summed over the sharers
    pr = Problem(orc, N_GLOBAL, P, hi=HI, perturb=0.1)
    c, rho = material(pr.mesh)
    tags = fa.tag_box_boundary(pr.mesh)
    w0 = 2 * np.pi * F0
    delta, beta = fa.compute_diffusivity_of_sound(w0, 1500.0, 0.2), 3.5
    m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, tags)
    n1 = -2.0 * beta / rho**2 / c**4
    u_ref, v_ref = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    dt = dt_value()
    p0 = 6.0e6
    orc.westervelt_rk4(3, pr.N, pr.dm, pr.G, pr.detJ, pr.D, lin, att, n1, -n1, m, src, absb, src2, F0, p0, S0, 0.0,
                       NSTEPS * dt * (1 + 1e-12), dt, u_ref, v_ref)
    size = 2
    ctxs = [fa.Context(0) for _ in range(size)]
    fa.Context.init_local_group(ctxs)
    models, offs = [], []
    for r in range(size):
        mesh = fa.BoxMesh([0, 0, 0], HI, N_GLOBAL, rank=r, size=size, perturb=0.1)
        V = fa.FunctionSpace(mesh, P)
        cr, rr = material(mesh)
        nc = mesh.num_cells
        models.append(fa.WesterveltSpectralExplicit(mesh, fa.tag_box_boundary(mesh), P, cr, rr, np.full(nc, delta),
                                                    np.full(nc, beta), F0, p0, S0, 4, dt, V=V, ctx=ctxs[r]))
        offs.append(V.global_offset)
    fa.group_finish_setup(models)
    for mdl in models:
        mdl.init()
    fa.group_rk4_steps(models, 0.0, dt, NSTEPS)
    for r, mdl in enumerate(models):
        n = mdl.data.ndofs
        u = mdl.u_sol().x.array
        assert np.abs(u - u_ref[offs[r]:offs[r] + n]).max() < 1e-10 * np.abs(u_ref).max()
        mdl.close()
    for cx in ctxs:
        cx.close()


@pytest.mark.gpu
def test_rccl_binding_selftest():
    # grouped ncclSend/ncclRecv (to the own rank) through the dlopen'ed RCCL on the library stream
    c = fa.Context(0)
    c.comm_selftest(1 << 18)
    c.close()


@pytest.mark.gpu
def test_general_partition_four_quadrants_gpu(orc):
    """A partition DOLFINx could produce rather than slabs: the box cut into 2 x 2 quadrants in x-y,
    every rank an unstructured local mesh with its own (geometric) DOF numbering, neighbour lists
    built from global DOF identity and ordered by it.  Interface faces are not contiguous index
    ranges, and the DOFs on the central line are held by all four ranks -- each adds the four partial
    sums in ascending rank order.  In-process transport on one GPU against the single-rank oracle."""
    from fenicsxfus_amd.unstructured import HexFunctionSpace, HexMesh
    n, hi, Pq, nsteps = (4, 4, 3), [0.016, 0.016, 0.012], 3, 5
    pr = Problem(orc, n, Pq, hi=hi, perturb=0.1)
    nc = pr.mesh.num_cells
    cen = pr.mesh.cell_centroids()
    c = np.where(cen[:, 2] > 0.5 * hi[2], 2800.0, 1500.0)
    rho = np.where(cen[:, 2] > 0.5 * hi[2], 1850.0, 1000.0)
    gtags = fa.tag_box_boundary(pr.mesh)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, gtags)
    dt = 0.4 * (hi[0] / n[0]) / (2800.0 * Pq**2)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, F0, P0, S0, 0.0, nsteps * dt * (1 + 1e-12), dt, u, v)
    assert np.abs(u).max() > 0
    from scipy.spatial import cKDTree
    # true (mapped) positions of the global dofs: the same trilinear map the local spaces use
    Xg = np.zeros((pr.ndofs, 3))
    Xg[pr.dm] = HexFunctionSpace(HexMesh(pr.mesh.geometry.x, pr.mesh.geometry.dofmap), Pq)._node_x
    tree = cKDTree(Xg)

    def global_ids(X):
        d, i = tree.query(X)
        assert d.max() < 1e-9 * max(hi)
        return i
    quad = (cen[:, 0] > 0.5 * hi[0]).astype(int) + 2 * (cen[:, 1] > 0.5 * hi[1]).astype(int)
    size = 4
    ctxs = [fa.Context(0, block_elems=4) for _ in range(size)]
    fa.Context.init_local_group(ctxs)
    models, gids = [], []
    for r in range(size):
        cells = np.nonzero(quad == r)[0]
        used, inv = np.unique(pr.mesh.geometry.dofmap[cells], return_inverse=True)
        lmesh = HexMesh(pr.mesh.geometry.x[used], inv.reshape(len(cells), 8))
        V = HexFunctionSpace(lmesh, Pq)
        gids.append(global_ids(V.tabulate_dof_coordinates()))
        # global-boundary facets of this rank's cells, as (local cell, local facet, tag)
        loc_of = {g: i for i, g in enumerate(cells)}
        sel = np.isin(gtags.cells, cells)
        tags = fa.FacetTags(np.array([loc_of[g] for g in gtags.cells[sel]], np.int32), gtags.local_facets[sel],
                            gtags.values[sel])
        models.append((lmesh, V, tags, c[cells], rho[cells]))
    for r in range(size):       # neighbour lists: shared global dofs, ordered by global id on both sides
        lmesh, V, tags, cr, rr = models[r]
        V.neighbours = []
        mine = {g: i for i, g in enumerate(gids[r])}
        for q in range(size):
            if q == r:
                continue
            shared = np.intersect1d(gids[r], gids[q])
            if len(shared):
                V.neighbours.append((q, np.array([mine[g] for g in shared], dtype=np.int32)))
        assert len(V.neighbours) == 3                     # every quadrant touches the central line
    four = set(gids[0]) & set(gids[1]) & set(gids[2]) & set(gids[3])
    assert len(four) == n[2] * Pq + 1                     # dofs held by all four ranks
    mods = [fa.LinearSpectralExplicit(lm, tg, Pq, cr, rr, F0, P0, S0, 4, dt, V=V, ctx=ctxs[r])
            for r, (lm, V, tg, cr, rr) in enumerate(models)]
    fa.group_finish_setup(mods)
    for mdl in mods:
        mdl.init()
    fa.group_rk4_steps(mods, 0.0, dt, nsteps)
    sols = []
    for r, mdl in enumerate(mods):
        assert np.abs(mdl.mass_vector() - m[gids[r]]).max() < 1e-14 * np.abs(m).max()
        ur = mdl.u_sol().x.array
        sols.append(dict(zip(gids[r].tolist(), ur.tolist())))
        assert np.abs(ur - u[gids[r]]).max() < 1e-10 * np.abs(u).max()
        assert np.abs(mdl.v_n.x.array - v[gids[r]]).max() < 1e-10 * np.abs(v).max()
    for g in four:                                        # identical bits on all four sharers
        assert len({sols[r][g] for r in range(size)}) == 1
    for mdl in mods:
        mdl.close()
    for cx in ctxs:
        cx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model_kind", ["linear", "westervelt"])
def test_external_transport_gpu(orc, model_kind):
    """The external-transport entry points (what a GPU-aware-MPI caller uses instead of the built-in
    RCCL exchange): three slab ranks on one GPU, the exchange done HERE by device-to-device copies
    between the ranks' send / receive buffers, the stage and setup halves driven one by one."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    size, nsteps = 3, NSTEPS
    pr = Problem(orc, N_GLOBAL, P, hi=HI, perturb=0.1)
    c, rho = material(pr.mesh)
    gt = fa.tag_box_boundary(pr.mesh)
    dt = dt_value()
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    w0 = 2 * np.pi * F0
    delta = np.where(c > 2000.0, fa.compute_diffusivity_of_sound(w0, 2800.0, 46.0), fa.compute_diffusivity_of_sound(w0, 1500.0, 0.2))
    beta = np.where(c > 2000.0, 6.0, 3.5)
    p0 = P0 if model_kind == "linear" else 100 * P0
    if model_kind == "linear":
        m, src, absb, coeff = pr.linear_model_vectors(c, rho, gt)
        orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, F0, p0, S0, 0.0, nsteps * dt * (1 + 1e-12), dt, u, v)
    else:
        m, src, absb, src2, lin, att = pr.lossy_model_vectors(c, rho, delta, gt)
        n1 = -2.0 * beta / rho**2 / c**4
        orc.westervelt_rk4(3, pr.N, pr.dm, pr.G, pr.detJ, pr.D, lin, att, n1, -n1, m, src, absb, src2, F0, p0, S0,
                           0.0, nsteps * dt * (1 + 1e-12), dt, u, v)
    ctxs, models, offs = [], [], []
    for r in range(size):
        cx = fa.Context(0)
        cx.init_external(r, size)
        mesh = fa.BoxMesh([0, 0, 0], HI, N_GLOBAL, rank=r, size=size, perturb=0.1)
        V = fa.FunctionSpace(mesh, P)
        cr, rr = material(mesh)
        cxr = mesh.cell_centroids()[:, 0]
        if model_kind == "linear":
            mdl = fa.LinearSpectralExplicit(mesh, fa.tag_box_boundary(mesh), P, cr, rr, F0, p0, S0, 4, dt, V=V, ctx=cx)
        else:
            dr = np.where(cr > 2000.0, delta.max(), delta.min())
            br = np.where(cr > 2000.0, 6.0, 3.5)
            mdl = fa.WesterveltSpectralExplicit(mesh, fa.tag_box_boundary(mesh), P, cr, rr, dr, br, F0, p0, S0, 4, dt,
                                                V=V, ctx=cx)
        ctxs.append(cx), models.append(mdl), offs.append(V.global_offset)
    layouts = [mdl.data.halo_layout() for mdl in models]
    bufs = [mdl.data.halo_buffers() for mdl in models]
    assert [len(l[0]) for l in layouts] == [1, 2, 1]

    def exchange():
        """rank r's send range for neighbour q -> q's receive range for neighbour r (8-byte values)."""
        for r in range(size):
            for q, cnt, off in zip(*layouts[r]):
                k = list(layouts[q][0]).index(r)
                assert layouts[q][1][k] == cnt
                rc = hip.hipMemcpy(bufs[q][1] + 8 * int(layouts[q][2][k]), bufs[r][0] + 8 * int(off), 8 * int(cnt), 3)
                assert rc == 0

    for k in range(models[0].setup_count()):
        for mdl in models:
            mdl.setup_pack(k)
        exchange()
        for mdl in models:
            mdl.setup_unpack(k)
    for mdl in models:
        mdl.setup_finish()
        mdl.init()
    t = 0.0
    for _ in range(nsteps):
        for i in range(4):
            for mdl in models:
                mdl.stage_begin(i, t, dt)
            exchange()
            for mdl in models:
                mdl.stage_end(i, t, dt)
        t += dt
    for r, mdl in enumerate(models):
        k = mdl.data.ndofs
        assert np.abs(mdl.mass_vector() - m[offs[r]:offs[r] + k]).max() < 1e-14 * np.abs(m).max()
        assert np.abs(u).max() > 0
        assert np.abs(mdl.u_sol().x.array - u[offs[r]:offs[r] + k]).max() < 1e-10 * np.abs(u).max()
        assert np.abs(mdl.v_n.x.array - v[offs[r]:offs[r] + k]).max() < 1e-10 * np.abs(v).max()
    with pytest.raises(fa.FusError):                     # the built-in loop refuses: no transport of its own
        models[0].rk4_steps(0.0, dt, 1)
    for mdl in models:
        mdl.close()
    for cx in ctxs:
        cx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("Ph", [5, 7])
def test_slabs_higher_degree_gpu(orc, Ph):
    """Two slabs at p = 5 and p = 7: the single-register-set block kernel (fp64, streamed geometry) with
    interface DOFs, against the single-rank oracle."""
    n, hi, nsteps, size = (4, 2, 2), [0.016, 0.008, 0.008], 4, 2
    pr = Problem(orc, n, Ph, hi=hi, perturb=0.1)
    nc = pr.mesh.num_cells
    c, rho = np.full(nc, 1500.0), np.full(nc, 1000.0)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, fa.tag_box_boundary(pr.mesh))
    dt = 0.4 * (hi[0] / n[0]) / (1500.0 * Ph**2)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, F0, P0, S0, 0.0, nsteps * dt * (1 + 1e-12), dt, u, v)
    ctxs = [fa.Context(0) for _ in range(size)]
    fa.Context.init_local_group(ctxs)
    models, offs = [], []
    for r in range(size):
        mesh = fa.BoxMesh([0, 0, 0], hi, n, rank=r, size=size, perturb=0.1)
        V = fa.FunctionSpace(mesh, Ph)
        k = mesh.num_cells
        models.append(fa.LinearSpectralExplicit(mesh, fa.tag_box_boundary(mesh), Ph, np.full(k, 1500.0),
                                                np.full(k, 1000.0), F0, P0, S0, 4, dt, V=V, ctx=ctxs[r]))
        assert not models[-1].data.is_affine()
        offs.append(V.global_offset)
    fa.group_finish_setup(models)
    for mdl in models:
        mdl.init()
    fa.group_rk4_steps(models, 0.0, dt, nsteps)
    assert np.abs(u).max() > 0
    for r, mdl in enumerate(models):
        k = mdl.data.ndofs
        assert np.abs(mdl.u_sol().x.array - u[offs[r]:offs[r] + k]).max() < 1e-10 * np.abs(u).max()
        assert np.abs(mdl.v_n.x.array - v[offs[r]:offs[r] + k]).max() < 1e-10 * np.abs(v).max()
    for mdl in models:
        mdl.close()
    for cx in ctxs:
        cx.close()


def _gpu_rank_worker(rank, size, port, q):
    """One RANK PROCESS of the N > 1 path on the HIP library: its own x-slab, the library's pack / ordered-sum / stage
    kernels, interface values moved between the processes by gloo through host memory (the external-transport entry
    points of fusmi.h; on a multi-GPU node the library's RCCL send/recv takes this place)."""
    import ctypes

    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    import torch

    hip = ctypes.CDLL("libamdhip64.so")
    cx = fa.Context(0)
    cx.init_external(rank, size)
    mesh = fa.BoxMesh([0, 0, 0], HI, N_GLOBAL, rank=rank, size=size, perturb=0.1)
    V = fa.FunctionSpace(mesh, P)
    c, rho = material(mesh)
    dt = dt_value()
    mdl = fa.LinearSpectralExplicit(mesh, fa.tag_box_boundary(mesh), P, c, rho, F0, P0, S0, 4, dt, V=V, ctx=cx)
    ranks, counts, offs = mdl.data.halo_layout()
    sp, rp, n = mdl.data.halo_buffers()
    hs, hr = np.zeros(n), np.zeros(n)

    def exchange():
        assert hip.hipMemcpy(ctypes.c_void_p(hs.ctypes.data), ctypes.c_void_p(sp), ctypes.c_size_t(8 * n), 2) == 0
        ops, bufs = [], []
        for qr, cnt, off in zip(ranks.tolist(), counts.tolist(), offs.tolist()):
            rb = torch.empty(cnt, dtype=torch.float64)
            bufs.append((off, cnt, rb))
            ops.append(dist.P2POp(dist.irecv, rb, qr))
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(hs[off:off + cnt].copy()), qr))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for off, cnt, rb in bufs:
            hr[off:off + cnt] = rb.numpy()
        assert hip.hipMemcpy(ctypes.c_void_p(rp), ctypes.c_void_p(hr.ctypes.data), ctypes.c_size_t(8 * n), 1) == 0

    mdl.external_setup(exchange)
    mdl.init()
    mdl.external_rk_steps(0.0, dt, NSTEPS, exchange)
    q.put((rank, V.global_offset, mdl.mass_vector(), mdl.u_sol().x.array.copy(), mdl.v_n.x.array.copy()))
    dist.barrier()
    mdl.close()
    cx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [2, 3])
def test_distinct_rank_processes_gpu(orc, size):
    """Two / three distinct rank PROCESSES on the HIP path (they share this box's one GPU; gloo moves the interface
    values): state equal to the single-rank oracle, interface planes bit-identical on both sharers."""
    # (no torch in THIS process: it holds the HIP runtime through libfusmi already; the rank processes are fresh
    # interpreters that import torch first)
    import multiprocessing as mp

    ref, m_ref, u_ref, v_ref = single_rank_reference(orc)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_rank_worker, args=(r, size, port, q)) for r in range(size)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, off, m, u, v in res:
        k = len(u)
        assert np.abs(m - m_ref[off:off + k]).max() < 1e-14 * np.abs(m_ref).max()
        assert np.abs(u - u_ref[off:off + k]).max() < 1e-10 * np.abs(u_ref).max()
        assert np.abs(v - v_ref[off:off + k]).max() < 1e-10 * np.abs(v_ref).max()
    for a, b in zip(res[:-1], res[1:]):
        plane = len(a[3]) - (b[1] - a[1])
        assert plane > 0 and np.array_equal(a[3][-plane:], b[3][:plane])
