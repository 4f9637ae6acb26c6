"""The DOLFINx-present adapter run on duck-typed stand-ins (tests/fake_dolfinx.py):
the tensor dofmap of `reorder_dofmap` (cpp/fenicsx-sf/common/permute.hpp:15-42), the 1-D node order
taken from Basix at run time, and neighbour lists from an `IndexMap` in which every pair of holders of a
DOF lists the other -- also the holders that are not its owner (partition edges / corners).  CPU: the
lists against the ones built from global DOF identity.  GPU: a four-rank run through the wrapped spaces
against the single-rank oracle."""
import numpy as np
import pytest

import fenicsxfus_amd as fa
from fake_dolfinx import FakeBasix, exchange_all, partition
from fenicsxfus_amd import dolfinx_adapter as ad
from util import Problem

F0, P0, S0 = 0.5e6, 60000.0, 1500.0


def _quadrants(orc, P, n=(4, 4, 3), hi=(0.016, 0.016, 0.012), seed=1, endpoints_first=True):
    pr = Problem(orc, n, P, hi=list(hi), perturb=0.1)
    cen = pr.mesh.cell_centroids()
    part = (cen[:, 0] > 0.5 * hi[0]).astype(int) + 2 * (cen[:, 1] > 0.5 * hi[1]).astype(int)
    bx = FakeBasix(P, 3, seed=seed, endpoints_first=endpoints_first)
    ranks = partition(pr, part, P, bx)
    msgs = [ad.sharer_messages(rk["V"].dofmap.index_map) for rk in ranks]
    recv = exchange_all(msgs)
    spaces = [ad.wrap_function_space(rk["V"], P, basix=bx, alltoall=(lambda send, r=r: recv[r])) for r, rk in enumerate(ranks)]
    return pr, part, bx, ranks, spaces


@pytest.mark.parametrize("P", [2, 3])
def test_tensor_dofmap_is_reorder_dofmap(orc, P):
    pr, part, bx, ranks, spaces = _quadrants(orc, P)
    perm = np.argsort(bx.tp)                                   # permute.hpp:27-32
    for rk, V in zip(ranks, spaces):
        lst = rk["V"].dofmap.list
        expect = np.stack([lst[:, perm[i]] for i in range(lst.shape[1])], axis=1)   # permute.hpp:38-41, entry by entry
        assert np.array_equal(V.tensor_dofmap, expect)
        assert np.array_equal(V.tensor_dofmap, rk["tensor_local"])
        assert np.array_equal(V.nodes1d, bx.pts) and V.nodes1d[1] == 1.0           # Basix order: end points first
        assert V.num_dofs == len(rk["oracle_ids"])


def test_neighbour_lists_cover_every_pair_of_holders(orc):
    P = 3
    pr, part, bx, ranks, spaces = _quadrants(orc, P)
    size = len(ranks)
    ids = [rk["oracle_ids"] for rk in ranks]
    for r, V in enumerate(spaces):
        nb = dict(V.neighbours)
        assert sorted(nb) == [q for q in range(size) if q != r]      # every quadrant touches the central line
        for q, idx in nb.items():
            # both sides list the same dofs in the same order
            other = dict(spaces[q].neighbours)[r]
            assert len(idx) == len(other)
            assert np.array_equal(ids[r][idx], ids[q][other])
            # ... and they are exactly the dofs both ranks hold
            assert set(ids[r][idx].tolist()) == set(ids[r].tolist()) & set(ids[q].tolist())
    four = set(ids[0]) & set(ids[1]) & set(ids[2]) & set(ids[3])
    assert len(four) == 3 * P + 1
    # the case the owner-only pairing misses: ranks 1, 2, 3 all ghost the central line from rank 0
    im1 = ranks[1]["V"].dofmap.index_map
    assert set(im1.owners.tolist()) == {0} or 0 in set(im1.owners.tolist())
    central_on_1 = [k for k, g in enumerate(ids[1]) if g in four]
    assert set(central_on_1) <= set(dict(spaces[1].neighbours)[2].tolist())
    assert set(central_on_1) <= set(dict(spaces[1].neighbours)[3].tolist())


def test_sharer_message_for_unknown_dof_is_an_error(orc):
    pr, part, bx, ranks, spaces = _quadrants(orc, 2)
    im = ranks[1]["V"].dofmap.index_map
    with pytest.raises(ValueError):
        ad.neighbours_from_index_map(im, {0: np.array([10**9, 1, 2], np.int64)})


def test_single_rank_space_has_no_neighbours(orc):
    pr = Problem(orc, (2, 2, 2), 2)
    bx = FakeBasix(2, 3)
    rk = partition(pr, np.zeros(pr.mesh.num_cells, int), 2, bx)[0]
    V = ad.wrap_function_space(rk["V"], 2, basix=bx)
    assert V.neighbours == [] and V.num_dofs == pr.ndofs


@pytest.mark.gpu
@pytest.mark.parametrize("endpoints_first", [False, True])
def test_four_ranks_through_the_adapter_gpu(orc, endpoints_first):
    """Linear RK4 on four quadrant ranks whose spaces come from `wrap_function_space`: operator action and
    5 steps against the single-rank oracle, results in each rank's DOLFINx-local numbering, DOFs held by all
    four ranks bit-identical on every one of them."""
    P, nsteps = 3, 5
    n, hi = (4, 4, 3), (0.016, 0.016, 0.012)
    pr, part, bx, ranks, spaces = _quadrants(orc, P, n, hi, endpoints_first=endpoints_first)
    cen = pr.mesh.cell_centroids()
    c = np.where(cen[:, 2] > 0.5 * hi[2], 2800.0, 1500.0)
    rho = np.where(cen[:, 2] > 0.5 * hi[2], 1850.0, 1000.0)
    gtags = fa.tag_box_boundary(pr.mesh)
    m, src, absb, coeff = pr.linear_model_vectors(c, rho, gtags)
    dt = 0.4 * (hi[0] / n[0]) / (2800.0 * P**2)
    u, v = np.zeros(pr.ndofs), np.zeros(pr.ndofs)
    orc.linear_rk4(3, pr.N, pr.dm, pr.G, pr.D, coeff, m, src, absb, F0, P0, S0, 0.0, nsteps * dt * (1 + 1e-12), dt, u, v)
    assert np.abs(u).max() > 0
    # single-rank operator action through a wrapped space (rank-local part: every cell of that rank)
    x = np.random.default_rng(0).standard_normal(pr.ndofs)
    yref = pr.K(x, coeff)
    size = len(ranks)
    ctxs = [fa.Context(0, block_elems=4) for _ in range(size)]
    fa.Context.init_local_group(ctxs)
    mods = []
    ysum = np.zeros(pr.ndofs)
    for r, (rk, V) in enumerate(zip(ranks, spaces)):
        cells = rk["cells"]
        loc_of = {g: i for i, g in enumerate(cells)}
        sel = np.isin(gtags.cells, cells)
        tags = fa.FacetTags(np.array([loc_of[g] for g in gtags.cells[sel]], np.int32), gtags.local_facets[sel],
                            gtags.values[sel])
        mdl = fa.LinearSpectralExplicit(rk["mesh"], tags, P, c[cells], rho[cells], F0, P0, S0, 4, dt, V=V, ctx=ctxs[r])
        mods.append(mdl)
        ids = rk["oracle_ids"]
        yl = mdl.data.stiffness(x[ids], coeff[cells], np.zeros(len(ids)))
        np.add.at(ysum, ids, yl)                      # the ranks' partial actions add up to the global one
    assert np.abs(ysum - yref).max() < 1e-12 * np.abs(yref).max()
    fa.group_finish_setup(mods)
    for mdl in mods:
        mdl.init()
    fa.group_rk4_steps(mods, 0.0, dt, nsteps)
    sols = []
    for rk, mdl in zip(ranks, mods):
        ids = rk["oracle_ids"]
        assert np.abs(mdl.mass_vector() - m[ids]).max() < 1e-14 * np.abs(m).max()
        ur = mdl.u_sol().x.array
        sols.append(dict(zip(ids.tolist(), ur.tolist())))
        assert np.abs(ur - u[ids]).max() < 1e-10 * np.abs(u).max()
        assert np.abs(mdl.v_n.x.array - v[ids]).max() < 1e-10 * np.abs(v).max()
    four = set.intersection(*[set(rk["oracle_ids"].tolist()) for rk in ranks])
    for g in four:
        assert len({s[g] for s in sols}) == 1
    for mdl in mods:
        mdl.close()
    for cx in ctxs:
        cx.close()
