#!/usr/bin/env python3
"""Headline benchmark: DOF-updates/s of the Linear RK4 step (p=4 hex, fp64) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full classical RK4 step (4 stages: stiffness action + shared-DOF reduction +
boundary terms + fused stage update) of the Linear acoustic model on BASELINE.json configs[1]:
3-D homogeneous water box, 64^3 hexes, p=4, fp64 (16 974 593 DOFs per GPU).  At N>1 each rank owns
a 64^3 x-slab of a (64 N) x 64 x 64 box (weak scaling) and exchanges one interface plane per
neighbour per stage over RCCL.  All state is resident in HBM when the timed region starts.
The headline goes through the library's kernel for general first-order hexahedral meshes (J and G
recomputed per point from each cell's trilinear map, --geometry trilinear: the synthetic box is not
allowed its affine shortcut); the reference's data path (per-point G streamed from HBM) and the
affine path are timed beside it at N=1 ('streamed_geometry', 'other_geometry').

Prints ONE JSON line on rank 0 (see the contract in the task statement) including
  roofline     -- the dominant kernel (block stiffness operator): algorithmic bytes per launch
                  (SURVEY 8d stiffness term: rho_e (s + 4 + g) + s per DOF, g = the path's geometry
                  bytes per element-DOF: 6 s streamed, 21 s / N^3 trilinear, 7 s / N^3 affine; plus
                  the fused stage update of the interior DOFs) / its average duration measured
                  with HIP events on the library's stream
  cpu_baseline -- the CPU oracle (port of the reference loop, -Ofast) timed on this host
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "fenicsx-fus_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def workload(n_per_gpu, P, rank, size, dtype=np.float64):
    """SURVEY 8d synthetic inputs: box edge 0.12 m per 64 cells, water, source on x=0 (tag 1),
    absorbing elsewhere (tag 2), f = 0.5 MHz, p0 = 60 kPa, CFL 0.5 snapped to steps/period."""
    import fenicsxfus_amd as fa

    L1 = 0.12 * n_per_gpu / 64.0
    mesh = fa.BoxMesh([0, 0, 0], [L1 * size, L1, L1], (n_per_gpu * size, n_per_gpu, n_per_gpu), rank=rank,
                      size=size, dtype=dtype)
    V = fa.FunctionSpace(mesh, P)
    tags = fa.tag_box_boundary(mesh)
    c0, rho0, freq, p0 = 1500.0, 1000.0, 0.5e6, 60000.0
    h = L1 / n_per_gpu
    dt = 0.5 * h / (c0 * P**2)
    period = 1.0 / freq
    dt = period / np.ceil(period / dt)
    return mesh, V, tags, c0, rho0, freq, p0, dt


def cpu_baseline(P, n, steps):
    """Oracle (restatement of Linear.hpp:228-314 + spectral_op.hpp:173-243, -Ofast -march=native) on a
    bounded sample of the same workload, timed on this host: threaded like the reference's one MPI
    rank per core (one contiguous x-slab of cells per thread) and, for scale,
    on a single thread."""
    import oracle

    oracle.build()

    mesh, V, tags, c0, rho0, freq, p0, dt = workload(n, P, 0, 1)
    nc, nd = mesh.num_cells, V.num_dofs
    wts = oracle.gll_weights_at(V.nodes1d)
    D = oracle.dphi(V.nodes1d)
    G, detJ = oracle.geometry(3, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d, wts)
    c, r = np.full(nc, c0), np.full(nc, rho0)
    m = oracle.mass(3, P + 1, V.tensor_dofmap, detJ, 1.0 / (r * c * c), np.ones(nd), np.zeros(nd))
    fd = lambda tag, cc: oracle.facet_diag(3, tags.cells[tags.find(tag)], tags.local_facets[tags.find(tag)], cc,  # noqa
                                           mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d, wts, V.tensor_dofmap, nd)
    src, absb = fd(1, 1.0 / r), fd(2, 1.0 / (r * c))
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    threads = max(1, min(ncpu, 16, n))            # the GPU box gives one GPU's share of the host cores
    layers = np.linspace(0, n, threads + 1).astype(np.int64) * (n * n)   # cells are x-major
    u, v = np.zeros(nd), np.zeros(nd)
    t0 = time.perf_counter()
    ns = oracle.linear_rk4_mt(P + 1, V.tensor_dofmap, G, D, -1.0 / r, m, src, absb, freq, p0, c0, 0.0,
                              steps * dt * (1 - 1e-9), dt, u, v, layers, fast=True)
    el = time.perf_counter() - t0
    s1 = max(2, steps // 8)
    u1, v1 = np.zeros(nd), np.zeros(nd)
    t0 = time.perf_counter()
    n1 = oracle.linear_rk4(3, P + 1, V.tensor_dofmap, G, D, -1.0 / r, m, src, absb, freq, p0, c0, 0.0,
                           s1 * dt * (1 - 1e-9), dt, u1, v1, fast=True)
    el1 = time.perf_counter() - t0
    return {"value": nd * ns / el, "unit": "DOF-updates/s", "cores": threads, "kind": "port",
            "sample": f"{n}^3 hex p={P} fp64 ({nd} DOFs), {ns} RK4 steps, {el:.1f} s, oracle -Ofast, "
                      f"{threads} threads (one x-slab of cells each)",
            "single_thread_value": nd * n1 / el1}


class _DevBuf:
    """A raw device buffer as seen by torch (``__cuda_array_interface__``), no copy."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def torch_exchange(torch, dist, model, rank, loopback=False):
    """exchange() for the external transport: neighbour ranges of the library's send buffer go to the
    neighbours' receive buffers with torch.distributed P2P (NCCL = RCCL); loopback: to the own one."""
    ranks, counts, offs = model.data.halo_layout()
    sp, rp, n = model.data.halo_buffers()
    if n == 0:
        return lambda: None
    ts = "<f8" if model.data.dtype == np.float64 else "<f4"
    send = torch.as_tensor(_DevBuf(sp, n, ts), device="cuda")
    recv = torch.as_tensor(_DevBuf(rp, n, ts), device="cuda")

    def exchange():
        if loopback:
            recv.copy_(send)
        else:
            ops = []
            for q, c, o in zip(ranks.tolist(), counts.tolist(), offs.tolist()):
                ops.append(dist.P2POp(dist.irecv, recv[o:o + c], q))
                ops.append(dist.P2POp(dist.isend, send[o:o + c], q))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        torch.cuda.synchronize()
    return exchange


GEOM_NAMES = {"stream": "general, G streamed (6 values per point from HBM: the reference's data path, B_general)",
              "affine": "affine cells (7 values per cell, B_affine)",
              "trilinear": "general first-order hexahedra, trilinear (21 values per cell, J and G recomputed per point, "
                           "B_affine + 14 s / N^3 per element-DOF)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", type=int, default=64, help="cells per axis per GPU")
    ap.add_argument("--P", type=int, default=4)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--block-elems", type=int, default=None)
    ap.add_argument("--waves", type=int, default=None)
    ap.add_argument("--deterministic", type=int, default=None, help="1: conflict-free rounds, 0: LDS atomics")
    ap.add_argument("--geometry", choices=["auto", "stream", "trilinear"], default="trilinear",
                    help="trilinear (headline): J and G recomputed per point from 21 numbers per cell -- what the library "
                         "does by default on any first-order hexahedral mesh with non-affine cells, taken here without "
                         "the affine shortcut the synthetic box would allow; stream: per-point G from HBM (the reference's "
                         "data path); auto: the library default (this box is affine: 7 numbers per cell)")
    ap.add_argument("--graph", type=int, default=None, help="1: replay each RK step as one hipGraph (launch-bound sizes)")
    ap.add_argument("--both-geometries", type=int, default=1,
                    help="1: at N=1 also time the other two geometry paths -> 'other_geometry' (affine), 'trilinear_geometry', "
                         "'streamed_geometry'; 2: at every N; 0: never")
    ap.add_argument("--halo-loopback", action="store_true",
                    help="diagnostic: time the middle slab of 3 with its RCCL exchange looped back to this GPU "
                         "(exchange overhead rehearsal on one GPU; the solution is not the physical one)")
    ap.add_argument("--model", choices=["linear", "lossy", "westervelt"], default="linear",
                    help="linear is BASELINE's metric; lossy / westervelt (SURVEY 8f-1) are reported as diagnostics")
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: no per-kernel HIP events in the timed region")
    ap.add_argument("--transport", choices=["rccl", "torch"], default="rccl",
                    help="N > 1 exchange: the library's RCCL send/recv (default; falls back to 'torch' if its "
                         "communicator cannot be created) or torch.distributed P2P through the external-transport "
                         "entry points (slower: the host drives every stage half)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-n", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=400)
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON result: whatever libraries print on file descriptor 1
    # (RCCL's start-up banner under torch.distributed.run, for one) is sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    if "FUSMI_BENCH_DEVICE" in os.environ:   # rehearsal of N>1 on a one-GPU box
        local_rank = int(os.environ["FUSMI_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)

    import fenicsxfus_amd as fa

    ctx = fa.Context(local_rank, block_elems=args.block_elems, waves=args.waves, deterministic=args.deterministic, geometry=args.geometry)
    if args.graph is not None:
        ctx.set_option("graph", args.graph)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torch.distributed.run
    transport = args.transport
    if args.halo_loopback:
        assert world == 1 and not launched
        if transport == "torch":
            ctx.init_external(1, 3)
        else:
            ctx.set_option("halo_loopback", 1)
            ctx.comm_init(1, 3, fa.Context.unique_id())
        args.both_geometries = 0
    ids2 = ids3 = [None]
    if world > 1 or launched:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        if transport == "rccl":
            try:
                ids = [fa.Context.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                ctx.comm_init(rank, world, ids[0])
                ids2 = [fa.Context.unique_id() if rank == 0 else None]   # communicators of the secondary runs
                dist.broadcast_object_list(ids2, src=0)
                ids3 = [fa.Context.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids3, src=0)
            except fa.FusError as e:            # the library's own RCCL communicator could not be made
                print(f"[bench rank {rank}] RCCL transport unavailable ({e}); using torch.distributed P2P", file=sys.stderr)
                transport = "torch"
                ctx = fa.Context(local_rank, block_elems=args.block_elems, waves=args.waves,
                                 deterministic=args.deterministic, geometry=args.geometry)
        if transport == "torch":
            ctx.init_external(rank, world)
            args.both_geometries = 0

    def barrier():
        if world > 1 or launched:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    P, n = args.P, args.cells
    np_dtype = np.float64 if args.dtype == "f64" else np.float32
    if args.halo_loopback:
        mesh, V, tags, c0, rho0, freq, p0, dt = workload(n, P, 1, 3, dtype=np_dtype)
    else:
        mesh, V, tags, c0, rho0, freq, p0, dt = workload(n, P, rank, world, dtype=np_dtype)
    nc = mesh.num_cells
    ndofs_global = V.dofmap.index_map.size_global

    def run(context, steps, warmup, profile):
        """Build the model on `context`, run warmup + timed steps; returns timings and model info."""
        if args.model == "linear":
            model = fa.LinearSpectralExplicit(mesh, tags, P, np.full(nc, c0), np.full(nc, rho0), freq, p0, c0, 4, dt,
                                              V=V, ctx=context)
        else:   # attenuating, weakly nonlinear water-like medium (BM7-SC1/main.cpp:43-46 style coefficients)
            delta = np.full(nc, fa.compute_diffusivity_of_sound(2 * np.pi * freq, c0, 0.2))
            if args.model == "lossy":
                model = fa.LossySpectralExplicit(mesh, tags, P, np.full(nc, c0), np.full(nc, rho0), delta, freq, p0, c0,
                                                 4, dt, V=V, ctx=context)
            else:
                model = fa.WesterveltSpectralExplicit(mesh, tags, P, np.full(nc, c0), np.full(nc, rho0), delta,
                                                      np.full(nc, 3.5), freq, p0, c0, 4, dt, V=V, ctx=context)
        if transport == "torch" and (world > 1 or args.halo_loopback):
            # external transport: the host drives the two halves of every stage around a torch P2P exchange
            exch = torch_exchange(torch, dist if world > 1 else None, model, rank, loopback=args.halo_loopback)
            model.external_setup(exch)

            def advance(t, n):
                model.external_rk_steps(t, dt, n, exch)
        else:
            def advance(t, n):
                model.rk4_steps(t, dt, n, sync=False)
        model.init()
        info = model.data.info()
        affine = model.data.geometry_mode()        # "stream" | "affine" | "trilinear"
        advance(0.0, warmup)
        if profile:
            context.profile_enable(2)      # HIP events around the dominant kernel only (see fusmi.h)
        barrier()
        t0 = time.perf_counter()
        advance(warmup * dt, steps)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1 or launched:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        prof = {}
        if profile:
            prof = {k: context.profile_get(k) for k in ("stiffness", "stiffness_if")}
            # per-kernel breakdown of a step: separate, untimed pass with events around every kernel
            nb = min(steps, 5)
            context.profile_enable(1)
            advance((warmup + steps) * dt, nb)
            context.synchronize()
            prof["breakdown_ms_per_step"] = {k: context.profile_get(k)[0] / nb for k in
                                             ("stiffness", "stiffness_if", "shared", "boundary", "stage", "halo")}
            context.profile_enable(0)
        u = model.u_sol().x.array
        finite = bool(np.isfinite(u).all()) and float(np.abs(u).max()) > 0.0
        model.close()
        return elapsed, prof, info, affine, finite

    elapsed, prof, info, affine, finite = run(ctx, args.steps, args.warmup, not args.no_profile)
    if args.no_profile:
        prof = {"stiffness": (0.0, 0), "stiffness_if": (0.0, 0), "breakdown_ms_per_step": {}}
    # secondary measurements: the same workload through the other geometry paths ("auto": the box
    # mesh is affine, G rebuilt from 7 numbers per cell; "trilinear": J and G recomputed per point
    # from 21 numbers per cell, valid for any first-order hexahedral mesh; "stream": 6 per point from HBM)
    others = []
    # (one GPU only unless --both-geometries 2: the scaling runs need nothing but the headline line)
    if args.both_geometries and (world == 1 or args.both_geometries >= 2):
        for g, ids_k in zip([g for g in ("stream", "auto", "trilinear") if g != args.geometry], (ids2, ids3)):
            ctx2 = fa.Context(local_rank, block_elems=args.block_elems, waves=args.waves,
                              deterministic=args.deterministic, geometry=g)
            if world > 1:
                ctx2.comm_init(rank, world, ids_k[0])
            e2, _, _, aff2, fin2 = run(ctx2, args.steps, args.warmup, False)
            others.append((e2, aff2, fin2))
            ctx2.close()

    if rank == 0:
        s = 8 if args.dtype == "f64" else 4
        N3 = (P + 1) ** 3
        ndl = V.num_dofs
        rho_e = nc * N3 / ndl
        # SURVEY 8d, per stage and DOF: stiffness = rho_e (s + 4 + 6 s) + s ; stage update = 12 s.
        # One launch of the dominant kernel does the stiffness action for every DOF and the fused
        # stage update for the block-interior DOFs it completes (the shared DOFs' update runs in
        # k_stage on the shared range).
        # B_general streams 6 s of G per element-DOF; B_affine rebuilds G from per-cell numbers
        # lossy / Westervelt: one more gathered operator input, Westervelt two more vector reads
        extra_x = {"linear": 0, "lossy": 1, "westervelt": 1}[args.model]
        extra_v = {"linear": 0, "lossy": 0, "westervelt": 2}[args.model]
        geo_b = {"stream": 6 * s, "affine": 7 * s / N3, "trilinear": 21 * s / N3}   # geometry bytes per element-DOF
        b_stiff = rho_e * (s + 4 + geo_b[affine] + extra_x * s) + s
        b_general = 4 * (b_stiff + (12 + extra_v) * s)
        n_int = info["interior_dofs"]
        alg_launch = b_stiff * ndl + (12 + extra_v) * s * n_int
        # N > 1: the stage's block kernel runs as two launches (interface blocks first, then the rest)
        k_ms, k_cnt = prof["stiffness"][0] + prof["stiffness_if"][0], prof["stiffness"][1]
        avg_ms = k_ms / max(k_cnt, 1)
        achieved = alg_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        value = (ndl if args.halo_loopback else ndofs_global) * args.steps / elapsed
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this
        # command (FETCH_SIZE x2 on gfx950 + WRITE_SIZE; profiles/r01l_pmc_traffic.json (trilinear), r01i_pmc_traffic.json (streamed), tools/gpu_profile.sh); only quoted
        # for the configuration those passes were taken on
        traffic = None
        pmc_name = {"stream": "r01i_pmc_traffic.json", "trilinear": "r01l_pmc_traffic.json"}.get(affine, "none")
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        if (os.path.exists(pmc) and n == 64 and P == 4 and args.block_elems is None and args.waves is None
                and not args.deterministic and args.dtype == "f64" and args.model == "linear" and world == 1
                and not args.halo_loopback):
            traffic = json.load(open(pmc))["k_block_op_fused"]["hbm_bytes_per_launch"]
        triad = ctx.measure_bandwidth()     # measured streaming bandwidth of this device
        out = {
            "metric": ("DOF-updates/sec (RK4 step) at p=4 hex fp64" if (P == 4 and args.dtype == "f64") else
                       f"DOF-updates/sec (RK4 step) at p={P} hex {args.dtype}") + ("" if args.model == "linear" else f" [{args.model} model]"),
            "value": value,
            "unit": "DOF-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"3D homogeneous wave, {n}^3 hex p={P} {args.dtype} per GPU, {args.model.capitalize()} RK4 "
                                   + ("(BASELINE.json configs[1])" if (n, P, args.dtype) == (64, 4, "f64") else
                                     "(BASELINE.json configs[2])" if (n, P, args.dtype) == (128, 7, "f64") else
                                     "(parity/diagnostic configuration)"), "ndofs_global": int(ndofs_global),
                       "cells_per_gpu": int(nc), "geometry": GEOM_NAMES[affine],
                       "partition": "middle x-slab of 3, exchange looped back (diagnostic)" if args.halo_loopback
                       else f"x-slabs x{world}", "transport": transport if (world > 1 or args.halo_loopback) else "none",
                       "blocks": info["nblocks"],
                       "lds_bytes_per_block": info["lds_bytes"], "dt": dt},
            "roofline": {"bound": "hbm", "kernel": f"k_block_op<{'double' if args.dtype == 'f64' else 'float'},{P},stiffness,+fused RK4 stage>", "achieved": achieved,
                         "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         # measured streaming bandwidth of this device (nt triad, 3 x 512 MiB) and, where the PMC
                         # traffic applies, the kernel's real HBM rate against it
                         "measured_triad_GBps": triad, "frac_of_measured_triad": achieved / triad,
                         "real_traffic_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None,
                         "real_traffic_frac_of_triad": (traffic / (avg_ms * 1e-3) / 1e9 / triad) if (traffic and avg_ms > 0) else None,
                         "traffic_source": f"profiles/{pmc_name} (separate rocprofv3 --pmc passes)" if traffic else None,
                         "algorithmic_bytes_per_launch": alg_launch, "interior_dofs": n_int, "avg_launch_ms": avg_ms,
                         "launches": k_cnt},
            "step_roofline": {"algorithmic_bytes_per_dof_update": b_general,
                              "achieved_GBps": b_general * value / world / 1e9,
                              "frac_of_8TBps": b_general * value / world / 8e12,
                              "frac_of_measured_triad": b_general * value / world / 1e9 / triad},
            # per-operator-action rate, comparable to the reference's logged stiffness actions
            # (2.0e9 DOF/s on 76 Icelake cores, p=4 fp64); here one action also does the fused stage update
            "operator_action_dofs_per_s": (ndl / (avg_ms * 1e-3)) if avg_ms > 0 else None,
            "kernel_ms_per_step": prof["breakdown_ms_per_step"],
            "finite_nonzero_solution": finite,
        }
        for e2, aff2, fin2 in others:
            b2 = 4 * (rho_e * (s + 4 + geo_b[aff2] + extra_x * s) + s + (12 + extra_v) * s)
            v2 = ndofs_global * args.steps / e2
            key = {"affine": "other_geometry", "trilinear": "trilinear_geometry", "stream": "streamed_geometry"}[aff2]
            out[key] = {"geometry": GEOM_NAMES[aff2], "value": v2, "unit": "DOF-updates/s",
                        "ms_per_step": 1e3 * e2 / args.steps, "algorithmic_bytes_per_dof_update": b2,
                        "frac_of_8TBps": b2 * v2 / world / 8e12, "finite_nonzero_solution": fin2}
        if not args.no_cpu and args.dtype == "f64" and world == 1:   # CPU leg on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(P, args.cpu_n, args.cpu_steps)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or launched:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
