#!/usr/bin/env python3
"""Headline benchmark: DOF-updates/s of the Linear RK4 step (p=4 hex, fp64) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full classical RK4 step (4 stages: stiffness action + shared-DOF reduction +
boundary terms + fused stage update) of the Linear acoustic model.

Workloads (config.workload names the one that ran):
  default            BASELINE.json configs[1]: 3-D homogeneous water box, 64^3 hexes, p=4, fp64
                     (16 974 593 DOFs per GPU); at N>1 each rank owns a 64^3 x-slab of a
                     (64 N) x 64 x 64 box (weak scaling) and exchanges one interface plane per
                     neighbour per stage over RCCL.
  --cells 128 --P 7  configs[2] (single GPU).
  --global-cells 256 --P 4 --medium skull            configs[3]: a fixed 256^3 box cut into N x-slabs
                     (strong scaling; 8 slabs of 32 x 256 x 256), heterogeneous c / rho map.
  --global-cells 256 --P 6 --dtype f32               configs[4].
All state is resident in HBM when the timed region starts.  The timed region is K steps, bracketed
by barrier + synchronize; it is repeated --repeats times (each repeat continues the same simulation)
and `value` comes from the MEDIAN repeat, with min / max beside it.

The headline goes through the library's kernel for general first-order hexahedral meshes (J and G
recomputed per point from each cell's trilinear map, --geometry trilinear: the synthetic box is not
allowed its affine shortcut); the reference's data path (per-point G streamed from HBM) and the
affine path are timed beside it at N=1 ('streamed_geometry', 'other_geometry').

Prints ONE JSON line on rank 0 including
  roofline     -- the dominant kernel (block stiffness operator + fused stage update):
                  `achieved` = algorithmic bytes per launch (rho_e (s + 4 + g) + s per DOF,
                  g = the path's geometry bytes per element-DOF: 6 s streamed, 21 s / N^3 trilinear,
                  7 s / N^3 affine; plus 12 s for each interior DOF the fused update finishes) / its
                  average duration measured with HIP events on the library's stream;
                  `traffic` = HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x 2 +
                  WRITE_SIZE, separate passes) -- measured live by child processes at N=1
                  (--traffic live) or taken from the committed profile of the same command;
                  `frac_real` = traffic / duration / 8 TB/s: the rate at which the kernel really moves bytes.
  cpu_baseline -- the CPU oracle (port of the reference loop, -Ofast) timed on this host
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "fenicsx-fus_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

WATER = (1500.0, 1000.0)
# tag -> (c, rho) of the reference's layered head model
# (cpp/fenicsx-sf/experiments/measure_vector_assembly_speed/main.cpp:43-83, 126-160)
LAYERS = [(1500.0, 1000.0), (1610.0, 1090.0), (2800.0, 1850.0), (2300.0, 1700.0), (2800.0, 1850.0), (1560.0, 1040.0)]


def medium(kind, mesh, L):
    """Per-cell (c, rho): water; "skull" = cortical-bone slab x in [0.4 L, 0.5 L] in water
    (BM7-SC1/main.cpp:37-40); "layers" = the six-material head model as x-slabs."""
    nc = mesh.num_cells
    c, rho = np.full(nc, WATER[0]), np.full(nc, WATER[1])
    if kind == "water":
        return c, rho
    cx = mesh.cell_centroids()[:, 0]
    if kind == "skull":
        sel = (cx > 0.4 * L) & (cx < 0.5 * L)
        c[sel], rho[sel] = 2800.0, 1850.0
    else:
        edges = np.array([0.0, 0.30, 0.36, 0.42, 0.50, 0.56, 1.0001]) * L
        tag = np.clip(np.searchsorted(edges, cx, side="right") - 1, 0, 5)
        tab = np.array(LAYERS)
        c, rho = tab[tag, 0].copy(), tab[tag, 1].copy()
    return c, rho


def workload(args, rank, size, dtype=np.float64, n_override=None):
    """SURVEY 8d synthetic inputs: box edge 0.12 m per 64 cells, source on x=0 (tag 1), absorbing
    elsewhere (tag 2), f = 0.5 MHz, p0 = 60 kPa, CFL 0.5 with the largest sound speed, snapped to an
    integer number of steps per period."""
    import fenicsxfus_amd as fa

    P = args.P
    if args.global_cells and n_override is None:        # strong scaling: a fixed global box in `size` x-slabs
        g = args.global_cells
        L = 0.12 * g / 64.0
        mesh = fa.BoxMesh([0, 0, 0], [L, L, L], (g, g, g), rank=rank, size=size, dtype=dtype)
        h, Lx = L / g, L
    else:                                               # weak scaling: n^3 (or nx x ny x nz) cells per rank
        n = n_override or args.cells
        nx, ny, nz = (n, n, n) if (n_override or not args.cells_xyz) else args.cells_xyz
        h = 0.12 / 64.0
        mesh = fa.BoxMesh([0, 0, 0], [h * nx * size, h * ny, h * nz], (nx * size, ny, nz), rank=rank, size=size, dtype=dtype)
        Lx = h * nx * size
    V = fa.FunctionSpace(mesh, P)
    tags = fa.tag_box_boundary(mesh)
    c, rho = medium(args.medium, mesh, Lx)
    cmax = max(v[0] for v in ([WATER] if args.medium == "water" else LAYERS))
    freq, p0 = 0.5e6, 60000.0
    dt = 0.5 * h / (cmax * P**2)
    period = 1.0 / freq
    dt = period / np.ceil(period / dt)
    return mesh, V, tags, c, rho, freq, p0, dt


def cpu_baseline(args, n, steps):
    """Oracle (restatement of Linear.hpp:228-314 + spectral_op.hpp:173-243, -Ofast -march=native) on a
    bounded sample of the same workload, timed on this host: threaded like the reference's one MPI
    rank per core (one contiguous x-slab of cells per thread) and, for scale,
    on a single thread."""
    import oracle

    oracle.build()

    P = args.P
    mesh, V, tags, c, r, freq, p0, dt = workload(args, 0, 1, n_override=n)
    nc, nd = mesh.num_cells, V.num_dofs
    wts = oracle.gll_weights_at(V.nodes1d)
    D = oracle.dphi(V.nodes1d)
    G, detJ = oracle.geometry(3, mesh.geometry.x, mesh.geometry.dofmap, V.nodes1d, wts)
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
    ns = oracle.linear_rk4_mt(P + 1, V.tensor_dofmap, G, D, -1.0 / r, m, src, absb, freq, p0, 1500.0, 0.0,
                              steps * dt * (1 - 1e-9), dt, u, v, layers, fast=True)
    el = time.perf_counter() - t0
    s1 = max(2, steps // 16)
    u1, v1 = np.zeros(nd), np.zeros(nd)
    t0 = time.perf_counter()
    n1 = oracle.linear_rk4(3, P + 1, V.tensor_dofmap, G, D, -1.0 / r, m, src, absb, freq, p0, 1500.0, 0.0,
                           s1 * dt * (1 - 1e-9), dt, u1, v1, fast=True)
    el1 = time.perf_counter() - t0
    return {"value": nd * ns / el, "unit": "DOF-updates/s", "cores": threads, "kind": "port",
            "sample": f"{n}^3 hex p={P} fp64 ({nd} DOFs, {args.medium}), {ns} RK4 steps, {el:.1f} s, oracle -Ofast, "
                      f"{threads} threads (one x-slab of cells each)",
            "single_thread_value": nd * n1 / el1}


class _DevBuf:
    """A raw device buffer as seen by torch (``__cuda_array_interface__``), no copy."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def torch_exchange(torch, dist, model, rank, loopback=False, host_staged=False):
    """exchange() for the external transport: neighbour ranges of the library's send buffer go to the
    neighbours' receive buffers with torch.distributed P2P (NCCL = RCCL); loopback: to the own one;
    host_staged (rehearsal of N ranks on fewer devices): device -> pinned host -> gloo -> device."""
    ranks, counts, offs = model.data.halo_layout()
    sp, rp, n = model.data.halo_buffers()
    if n == 0:
        return lambda: None
    ts = "<f8" if model.data.dtype == np.float64 else "<f4"
    send = torch.as_tensor(_DevBuf(sp, n, ts), device="cuda")
    recv = torch.as_tensor(_DevBuf(rp, n, ts), device="cuda")
    if host_staged:
        hs, hr = torch.empty_like(send, device="cpu").pin_memory(), torch.empty_like(recv, device="cpu").pin_memory()

    def exchange():
        if loopback:
            recv.copy_(send)
        elif host_staged:
            hs.copy_(send)
            torch.cuda.synchronize()
            ops = []
            for q, c, o in zip(ranks.tolist(), counts.tolist(), offs.tolist()):
                ops.append(dist.P2POp(dist.irecv, hr[o:o + c], q))
                ops.append(dist.P2POp(dist.isend, hs[o:o + c], q))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            recv.copy_(hr)
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
              "affine_diag": "affine cells with orthogonal edges (7 values per cell, B_affine; diagonal metric: three 1-D "
                             "stiffness contractions per element)",
              "trilinear": "general first-order hexahedra, trilinear (21 values per cell, J and G recomputed per point, "
                           "B_affine + 14 s / N^3 per element-DOF)"}


def fused_kernel_filter(name, P, dtype):
    """Kernel-trace / counter rows of the fused block kernel: k_block_op<T, P, OP=0 (stiffness), ATOMIC,
    STAGE in {0, 1, 3}, ...> -- not the plain operator action (STAGE = -1)."""
    t = "double" if dtype == "f64" else "float"
    if f"k_block_op<{t}, {P}, 0, " not in name:
        return False
    targs = name.split("k_block_op<")[1].split(",")
    return targs[4].strip() != "-1"


def live_traffic(argv_tail, P, dtype):
    """HBM bytes per launch of the fused block kernel, measured now: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, kernel-trace only beside them, the
    program directly after `--`), FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM) prescribes for wide
    coalesced streams on gfx950; counters are in KiB.  Runs before this process touches the GPU."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    per = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"fus_pmc_{ctr}_", dir="/tmp")
        cmd = [rocprof, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--pmc-child", "--no-cpu", "--steps", "4", "--warmup", "1", "--repeats", "1",
               "--both-geometries", "0", "--traffic", "none", *argv_tail]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        except subprocess.TimeoutExpired:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"{ctr} pass timed out"
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"{ctr} pass failed (rc {r.returncode})"
        tot, seen = 0.0, set()
        for row in csv.DictReader(open(files[0])):
            if row["Counter_Name"] == ctr and fused_kernel_filter(row["Kernel_Name"], P, dtype):
                tot += float(row["Counter_Value"])
                seen.add(row["Dispatch_Id"])
        shutil.rmtree(d, ignore_errors=True)
        if not seen:
            return None, f"no fused block-kernel dispatch in the {ctr} pass"
        per[ctr] = (tot / len(seen) * 1024.0, len(seen))
    rd, wr = 2.0 * per["FETCH_SIZE"][0], per["WRITE_SIZE"][0]
    return {"hbm_bytes_per_launch": rd + wr, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
            "launches": per["FETCH_SIZE"][1]}, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command"


# Interior-dof vector streams of the fused stage update, in values per dof, by stage of one RK4 step
# (kernels.hpp stage kinds 4-7 = lean RK4 without accumulators, V_i = the stage velocities in three rotating buffers:
#  minv, v0 | un, V_1;  minv, V_1, v0 | un, V_2;  minv, V_2, v0, V_1 | un, V_3;  minv, V_3, v0, V_1, V_2 | u0, v0 --
#  the stage input x is counted apart and is still in LDS where the update needs it;
#  kinds 0, 1, 1, 3 with lean_rk4 = 0: 7, 10, 10, 6).  Shared dofs: k_shared_stage_planes has no LDS copy of the stage
#  input and reads it (u0 at stage 0, un later): one value more per stage.
LEAN_INTERIOR = (4, 5, 6, 7)
EVENT_SAMPLE = 5   # HIP events around every 5th launch of the dominant kernel in the timed region (see run())
FULL_INTERIOR = (7, 10, 10, 6)
LEAN_SHARED = (5, 6, 7, 8)


def compulsory_bytes(info, nc, N3, s, geom, model="linear", lean=True):
    """HBM bytes THIS data layout has to move (not the reference's: SURVEY 8d charges an int32 dofmap entry and a gathered
    value per element-dof): per launch of the block kernel, averaged over the four stages of an RK4 step, and per
    launch of the shared-dof stage kernel.  x once per block-local dof (interior once, a shared dof once per sharing
    block); the gather's index and the partial sum's position (2 x int32) and the partial sum itself per (block,
    shared dof) pair; geometry and coefficient per cell; the fused update's vector streams per interior dof; uint16
    dofmaps once per distinct block shape (L2 resident afterwards)."""
    nint, nsh, npairs = info["interior_dofs"], info["shared_dofs"], info["pairs"]
    geo = {"stream": 6 * N3, "affine": 7, "affine_diag": 7, "trilinear": 21}[geom] * s * nc
    nin = {"linear": 1, "lossy": 2, "westervelt": 2}[model]           # operator inputs gathered per block pass
    extra_v = {"linear": 0, "lossy": 0, "westervelt": 2}[model]
    streams = (LEAN_INTERIOR if lean else FULL_INTERIOR)
    epi = (sum(streams) / 4.0 + extra_v) * s * nint
    det = {"x_block_local": nin * (nint + npairs) * s, "gather_index_and_partial_position": 8 * npairs,
           "partial_sums_written": npairs * s, "geometry_per_cell": geo, "coefficient_per_cell": nin * nc * s,
           "stage_update_streams_interior": epi,
           "dofmap_uint16_first_touch": 2 * N3 * info["shapes"] * (nc / max(info["nblocks"], 1))}
    blk = sum(det.values())
    shared = (npairs + (sum(LEAN_SHARED if lean else FULL_INTERIOR) / 4.0 + extra_v) * nsh) * s
    return blk, shared, det



def _free_port():
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launcher_command(n, argv, port):
    """The rank processes of `bench.py --gpus n` run as a plain command: one torch.distributed.run child (n ranks of
    this script with the same arguments, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
            "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(n, argv, ndev, run=subprocess.run):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: this parent never touches the GPU (it only
    counted the devices); it starts the N ranks as a CHILD process tree (no exec), relays rank 0's single JSON line
    to its own stdout and returns the children's exit code.  With fewer devices than ranks (a one-GPU box) the
    ranks run as a rehearsal: they share the devices round-robin and the exchange is staged through host memory
    over gloo (RCCL refuses two ranks on one device); the JSON line says so (`rehearsal`)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if ndev < n:
        env["FUSMI_BENCH_REHEARSAL"] = str(ndev)
        print(f"[bench] {n} ranks on {ndev} device(s): rehearsal (shared devices, host-staged gloo exchange)", file=sys.stderr)
    cmd = launcher_command(n, argv, _free_port())
    r = run(cmd, stdout=subprocess.PIPE, env=env, cwd=os.getcwd())
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{") and ln.rstrip().endswith("}")]
    if r.returncode == 0 and not lines:
        print("[bench] the ranks exited 0 without a result line", file=sys.stderr)
        return 4
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    else:
        print(f"[bench] the {n}-rank launch failed (exit code {r.returncode}); command: {' '.join(cmd)}", file=sys.stderr)
    return r.returncode



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="RK4 steps per timed repeat (default: about 0.26 s per repeat at configs[1])")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="the timed K-step block is run this many times; value = median")
    ap.add_argument("--cells", type=int, default=64, help="cells per axis per GPU (weak scaling)")
    ap.add_argument("--cells-xyz", type=int, nargs=3, default=None, metavar=("NX", "NY", "NZ"),
                    help="cells per GPU as a box NX x NY x NZ instead of --cells^3 (32 256 256 = one rank's x-slab of the "
                         "256^3 configs[3] / [4] at 8 GPUs, run as a box of its own)")
    ap.add_argument("--global-cells", type=int, default=0,
                    help="strong scaling: a fixed G^3 box cut into N x-slabs (BASELINE configs[3], [4]: 256)")
    ap.add_argument("--medium", choices=["water", "skull", "layers"], default="water",
                    help="skull: cortical-bone slab in water (BM7-SC1/main.cpp:37-40); layers: the reference's six-material "
                         "head model as x-slabs (measure_vector_assembly_speed/main.cpp:43-83)")
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
    ap.add_argument("--mfma", type=int, default=None, choices=[-1, 0, 1],
                    help="degrees 6 and 7: index-1 / index-2 contractions on the matrix cores (1), on the vector ALUs (0), "
                         "or the library's measured default (-1)")
    ap.add_argument("--pack32", type=int, default=None, choices=[-1, 0, 1],
                    help="fp32 degrees 5-7: two elements per wave in packed float2 (1), scalar kernel (0), library default (-1)")
    ap.add_argument("--diag-metric", type=int, default=None, choices=[0, 1],
                    help="affine meshes with orthogonal cell edges: diagonal-metric form of the stiffness kernel (default: on)")
    ap.add_argument("--walk", type=int, default=None,
                    help="block-kernel workgroups per CU that walk several blocks each (0: one workgroup per block)")
    ap.add_argument("--lean-rk4", type=int, default=None, choices=[0, 1],
                    help="0: keep the RK4 accumulators u_, v_ in HBM at every stage (Linear.hpp:282-294); default 1")
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
                    help="N > 1 exchange: the library's RCCL send/recv (default) or torch.distributed P2P through the "
                         "external-transport entry points (slower: the host drives every stage half)")
    ap.add_argument("--overlap", choices=["ab", "on", "off"], default="ab",
                    help="N > 1: launch the interface blocks first and overlap the exchange with the remaining blocks "
                         "(option overlap_blocks). ab: time both during warm-up on this node's links, run the timed "
                         "region with the faster, report both")
    ap.add_argument("--traffic", choices=["live", "profile", "none"], default="live",
                    help="roofline.traffic: live = two rocprofv3 --pmc child passes of this command before the run (N=1 "
                         "only), profile = the committed profiles/ file of the same configuration, none")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-n", type=int, default=64, help="cells per axis of the CPU sample (64 = the GPU workload itself)")
    ap.add_argument("--cpu-steps", type=int, default=0,
                    help="RK4 steps of the CPU sample; 0 = sized for about 15 s of threaded CPU work")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON result: whatever libraries print on file descriptor 1
    # (RCCL's start-up banner under torch.distributed.run, for one) is sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torch.distributed.run

    import torch

    ndev = torch.cuda.device_count()        # (counting devices does not initialise the GPU)
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    if world != args.gpus:
        if not launched and args.gpus > 1:
            # a plain `python bench.py --gpus N`: start the N ranks ourselves (child processes), relay the line
            os.dup2(result_fd, 1)
            os.close(result_fd)
            sys.exit(self_launch(args.gpus, sys.argv[1:], ndev))
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # N ranks on fewer devices (set by self_launch on a one-GPU box, or by hand): shared devices, gloo, host-staged exchange
    rehearsal = int(os.environ.get("FUSMI_BENCH_REHEARSAL", "0")) if world > 1 else 0

    # ---- live HBM traffic of the dominant kernel (child processes, before this one touches the GPU) ----
    traffic, traffic_src = None, None
    if (args.traffic == "live" and world == 1 and not launched and not args.pmc_child and not args.halo_loopback
            and args.model == "linear"):
        tail = ["--cells", str(args.cells), "--P", str(args.P), "--dtype", args.dtype, "--geometry", args.geometry,
                "--medium", args.medium]
        if args.global_cells:
            tail += ["--global-cells", str(args.global_cells)]
        if args.cells_xyz:
            tail += ["--cells-xyz", *[str(k) for k in args.cells_xyz]]
        for k, v in (("--block-elems", args.block_elems), ("--waves", args.waves), ("--deterministic", args.deterministic),
                     ("--mfma", args.mfma), ("--lean-rk4", args.lean_rk4), ("--walk", args.walk), ("--pack32", args.pack32),
                     ("--diag-metric", args.diag_metric)):
            if v is not None:
                tail += [k, str(v)]
        traffic, traffic_src = live_traffic(tail, args.P, args.dtype)
        if traffic is None:
            print(f"[bench] live traffic unavailable: {traffic_src}", file=sys.stderr)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    if "FUSMI_BENCH_DEVICE" in os.environ:   # every rank on one named device
        local_rank = int(os.environ["FUSMI_BENCH_DEVICE"])
    elif rehearsal:
        local_rank = local_rank % rehearsal
    torch.cuda.set_device(local_rank)

    import fenicsxfus_amd as fa

    ctx = fa.Context(local_rank, block_elems=args.block_elems, waves=args.waves, deterministic=args.deterministic, geometry=args.geometry)
    if args.graph is not None:
        ctx.set_option("graph", args.graph)
    for key, val in (("mfma", args.mfma), ("lean_rk4", args.lean_rk4), ("walk", args.walk), ("pack32", args.pack32),
                     ("diag_metric", args.diag_metric)):
        if val is not None:
            ctx.set_option(key, val)
    transport = "torch" if rehearsal else args.transport
    if args.halo_loopback:
        assert world == 1 and not launched
        if transport == "torch":
            ctx.init_external(1, 3)
        else:
            ctx.set_option("halo_loopback", 1)
            ctx.comm_init(1, 3, fa.Context.unique_id())
        args.both_geometries = 0
    ids2 = ids3 = [None]
    dist = None
    if world > 1 or launched:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        if transport == "rccl":
            # the library's own communicator; a failure here is fatal (exit non-zero, nothing is re-launched)
            try:
                ids = [fa.Context.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                ctx.comm_init(rank, world, ids[0])
                ids2 = [fa.Context.unique_id() if rank == 0 else None]   # communicators of the secondary runs
                dist.broadcast_object_list(ids2, src=0)
                ids3 = [fa.Context.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids3, src=0)
            except fa.FusError as e:
                print(f"[bench rank {rank}] RCCL communicator could not be created: {e}", file=sys.stderr)
                sys.stderr.flush()
                os._exit(3)
        else:
            ctx.init_external(rank, world)
            args.both_geometries = 0

    def barrier():
        if world > 1 or launched:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def max_over_ranks(x):
        if world > 1 or launched:
            tt = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return x

    P = args.P
    np_dtype = np.float64 if args.dtype == "f64" else np.float32
    if args.halo_loopback:
        mesh, V, tags, c0, rho0, freq, p0, dt = workload(args, 1, 3, dtype=np_dtype)
    else:
        mesh, V, tags, c0, rho0, freq, p0, dt = workload(args, rank, world, dtype=np_dtype)
    nc = mesh.num_cells
    ndofs_global = V.dofmap.index_map.size_global
    overlap_ab = {}

    def run(context, steps, warmup, repeats, profile, headline=False):
        """Build the model on `context`, run warmup, then `repeats` timed blocks of `steps` steps (each bracketed
        by barrier + synchronize, max over ranks); returns the block times and model info."""
        if args.model == "linear":
            model = fa.LinearSpectralExplicit(mesh, tags, P, c0, rho0, freq, p0, 1500.0, 4, dt, V=V, ctx=context)
        else:   # attenuating, weakly nonlinear water-like medium (BM7-SC1/main.cpp:43-46 style coefficients)
            delta = np.full(nc, fa.compute_diffusivity_of_sound(2 * np.pi * freq, 1500.0, 0.2))
            if args.model == "lossy":
                model = fa.LossySpectralExplicit(mesh, tags, P, c0, rho0, delta, freq, p0, 1500.0, 4, dt, V=V, ctx=context)
            else:
                model = fa.WesterveltSpectralExplicit(mesh, tags, P, c0, rho0, delta, np.full(nc, 3.5), freq, p0, 1500.0,
                                                      4, dt, V=V, ctx=context)
        if transport == "torch" and (world > 1 or args.halo_loopback):
            # external transport: the host drives the two halves of every stage around a torch P2P exchange
            exch = torch_exchange(torch, dist if world > 1 else None, model, rank, loopback=args.halo_loopback,
                                  host_staged=bool(rehearsal))
            model.external_setup(exch)

            def advance(t, n):
                model.external_rk_steps(t, dt, n, exch)
        else:
            def advance(t, n):
                model.rk4_steps(t, dt, n, sync=False)
        model.init()
        info = model.data.info()
        info["mfma"] = model.data.uses_mfma()
        info["mfma4"] = model.data.uses_mfma4()
        info["pack32"] = model.data.uses_pack32()
        info["diag_metric"] = model.data.uses_diag_metric()
        affine = model.data.geometry_mode()        # "stream" | "affine" | "trilinear"
        advance(0.0, warmup)
        done = warmup

        def timed(n):
            nonlocal done
            barrier()
            t0 = time.perf_counter()
            advance(done * dt, n)
            barrier()
            el = time.perf_counter() - t0
            done += n
            return max_over_ranks(el)

        # exchange overlapped with the non-interface blocks, or only with the shared-dof stage kernel:
        # decided on this node's links during warm-up (both timings are reported)
        if headline and (world > 1 or args.halo_loopback) and transport == "rccl":
            if args.overlap == "ab":
                ab = max(3, min(steps, 10))
                for name, flag in (("off", 0), ("on", 1), ("off", 0), ("on", 1)):
                    context.set_option("overlap_blocks", flag)
                    timed(1)
                    overlap_ab.setdefault(name, []).append(1e3 * timed(ab) / ab)
                best = min(overlap_ab, key=lambda k: min(overlap_ab[k]))
                context.set_option("overlap_blocks", 1 if best == "on" else 0)
                overlap_ab["chosen"] = best
                timed(1)
            else:
                context.set_option("overlap_blocks", 1 if args.overlap == "on" else 0)
                overlap_ab["chosen"] = args.overlap
                timed(1)
        if profile:
            # HIP events around the dominant kernel only, and around every EVENT_SAMPLE-th launch of it: an event record
            # drains the queue between two kernels (events around every launch cost the timed region 1.7 %, around every
            # 5th 0.3 %); 5 is coprime with the 4 stages of a step, so the four stage kinds are sampled equally
            context.set_option("profile_sample", EVENT_SAMPLE)
            context.profile_enable(2)
        times = [timed(steps) for _ in range(repeats)]
        prof = {}
        if profile:
            prof = {k: context.profile_get(k) for k in ("stiffness", "stiffness_if")}
            # per-kernel breakdown of a step: separate, untimed pass with events around every kernel
            nb = min(steps, 5)
            context.set_option("profile_sample", 1)
            context.profile_enable(1)
            advance(done * dt, nb)
            context.synchronize()
            prof["breakdown_ms_per_step"] = {k: context.profile_get(k)[0] / nb for k in
                                             ("stiffness", "stiffness_if", "shared", "boundary", "stage", "halo")}
            context.profile_enable(0)
        u = model.u_sol().x.array
        finite = bool(np.isfinite(u).all()) and float(np.abs(u).max()) > 0.0
        model.close()
        if info["diag_metric"]:
            affine = "affine_diag"
        return times, prof, info, affine, finite

    times, prof, info, affine, finite = run(ctx, args.steps, args.warmup, max(1, args.repeats), not args.no_profile,
                                            headline=True)
    elapsed = float(np.median(times))
    mfma_used = bool(info.get("mfma"))
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
            # the same kernel options as the headline context: a like-for-like comparison
            if args.graph is not None:
                ctx2.set_option("graph", args.graph)
            for key, val in (("mfma", args.mfma), ("lean_rk4", args.lean_rk4), ("walk", args.walk), ("pack32", args.pack32),
                             ("diag_metric", args.diag_metric)):
                if val is not None:
                    ctx2.set_option(key, val)
            if world > 1:
                ctx2.comm_init(rank, world, ids_k[0])
            t2, _, info2, aff2, fin2 = run(ctx2, args.steps, args.warmup, min(3, max(1, args.repeats)), False)
            others.append((float(np.median(t2)), aff2, fin2, info2))
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
        geo_b = {"stream": 6 * s, "affine": 7 * s / N3, "affine_diag": 7 * s / N3, "trilinear": 21 * s / N3}   # geometry bytes per element-DOF
        b_stiff = rho_e * (s + 4 + geo_b[affine] + extra_x * s) + s
        b_general = 4 * (b_stiff + (12 + extra_v) * s)
        n_int = info["interior_dofs"]
        alg_launch = b_stiff * ndl + (12 + extra_v) * s * n_int
        # N > 1: the stage's block kernel may run as two launches (interface blocks first, then the rest)
        k_ms, k_cnt = prof["stiffness"][0] + prof["stiffness_if"][0], prof["stiffness"][1]
        avg_ms = k_ms / max(k_cnt, 1)
        achieved = alg_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        ndofs_counted = ndl if args.halo_loopback else ndofs_global
        value = ndofs_counted * args.steps / elapsed
        if traffic is None and args.traffic != "none" and world == 1 and not args.halo_loopback and args.model == "linear":
            # the committed rocprofv3 --pmc passes of this configuration (tools/gpu_profile.sh), if there are any
            idx = os.path.join(ROOT, "profiles", "traffic_index.json")
            key = f"{affine}:{args.cells if not args.global_cells else 'g%d' % args.global_cells}:p{P}:{args.dtype}"
            if (os.path.exists(idx) and args.block_elems is None and args.waves is None and not args.deterministic
                    and args.medium == "water"):
                ent = json.load(open(idx)).get(key)
                if ent:
                    traffic = {"hbm_bytes_per_launch": ent["hbm_bytes_per_launch"]}
                    traffic_src = f"profiles/{ent['file']} (separate rocprofv3 --pmc passes of this configuration)"
        tbytes = traffic["hbm_bytes_per_launch"] if traffic else None
        lean = (args.lean_rk4 is None or args.lean_rk4 == 1)
        comp_blk, comp_sh, comp_det = compulsory_bytes(info, nc, N3, s, affine, args.model, lean)
        comp_step = 4.0 * (comp_blk + comp_sh) / ndl            # bytes per DOF-update (4 stages)
        copy_bw = ctx.measure_bandwidth()     # measured streaming bandwidth of this device
        cellsdesc = "x".join(str(k) for k in args.cells_xyz) if args.cells_xyz else f"{args.cells}^3"
        mode = (f"strong scaling: {args.global_cells}^3 box in {world} x-slab(s)" if args.global_cells else
                f"weak scaling: {cellsdesc} cells per GPU")
        cfgname = {(64, 0, 4, "f64", "water"): "BASELINE.json configs[1]", (128, 0, 7, "f64", "water"): "BASELINE.json configs[2]"}.get(
            (args.cells, args.global_cells, P, args.dtype, args.medium))
        if args.global_cells == 256 and P == 4 and args.dtype == "f64" and args.medium != "water":
            cfgname = "BASELINE.json configs[3]"
        if args.global_cells == 256 and P == 6 and args.dtype == "f32":
            cfgname = "BASELINE.json configs[4]"
        if args.cells_xyz:
            cfgname = None
            if tuple(args.cells_xyz) == (32, 256, 256) and world == 1:
                if P == 6 and args.dtype == "f32":
                    cfgname = "one rank's 32x256x256 x-slab of BASELINE.json configs[4] (256^3 over 8 GPUs) as a box of its own"
                if P == 4 and args.dtype == "f64":
                    cfgname = "one rank's 32x256x256 x-slab of BASELINE.json configs[3] (256^3 over 8 GPUs) as a box of its own"
        ms_rep = [1e3 * t / args.steps for t in times]
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
            "scaling": "strong" if args.global_cells else "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"3D {'homogeneous' if args.medium == 'water' else 'heterogeneous (' + args.medium + ')'} wave, "
                                   f"hex p={P} {args.dtype}, {args.model.capitalize()} RK4, {mode} "
                                   f"({cfgname or 'parity/diagnostic configuration'})",
                       "ndofs_global": int(ndofs_global), "cells_per_gpu": int(nc), "geometry": GEOM_NAMES[affine],
                       "medium": args.medium,
                       "partition": "middle x-slab of 3, exchange looped back (diagnostic)" if args.halo_loopback
                       else f"x-slabs x{world}", "transport": transport if (world > 1 or args.halo_loopback) else "none",
                       "blocks": info["nblocks"], "mfma_contractions": mfma_used,
                       # index-1 contraction on v_mfma_f64_4x4x4_4b_f64 from the registers (p=7 fp64 trilinear kernel: default)
                       "mfma_4x4x4_index1": bool(info.get("mfma4")), "packed_fp32": bool(info.get("pack32")),
                       "lds_bytes_per_block": info["lds_bytes"], "dt": dt,
                       # the RK4 stage update: fused into the kernels' epilogues; values per interior dof and step
                       "rk4_update": ("fused, classical RK4 without accumulator vectors (kernels.hpp stage kinds 4-7): "
                                      f"{sum(LEAN_INTERIOR)} values per interior dof and step"
                                      if (args.lean_rk4 is None or args.lean_rk4 == 1) else
                                      f"fused, accumulators streamed as in Linear.hpp:282-294: {sum(FULL_INTERIOR)} values per "
                                      "interior dof and step"),
                       **({"n1_note": "BASELINE configs[4] on ONE GPU would be 3 630 961 153 local DOFs, beyond the int32 local "
                                      "indices of the reference's dofmap (and of this library): its smallest run is N = 2; "
                                      "per-GPU-size lines use --cells-xyz 32 256 256"}
                          if (P == 6 and args.dtype == "f32" and (args.cells_xyz or args.global_cells)) else {})},
            # the timed K-step block repeated: value / ms_per_step are the median repeat
            "repeats": {"n": len(times), "ms_per_step": ms_rep, "min": min(ms_rep), "max": max(ms_rep),
                        "median": 1e3 * elapsed / args.steps,
                        "value_min": ndofs_counted * args.steps / max(times), "value_max": ndofs_counted * args.steps / min(times)},
            "roofline": {"bound": "hbm", "kernel": f"k_block_op<{'double' if args.dtype == 'f64' else 'float'},{P},stiffness,+fused RK4 stage>", "achieved": achieved,
                         "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": tbytes,
                         # the rate at which the kernel really moves bytes (PMC traffic / HIP-event duration)
                         "real_traffic_GBps": (tbytes / (avg_ms * 1e-3) / 1e9) if (tbytes and avg_ms > 0) else None,
                         "frac_real": (tbytes / (avg_ms * 1e-3) / 8e12) if (tbytes and avg_ms > 0) else None,
                         "traffic_source": traffic_src if tbytes else None,
                         "traffic_detail": traffic if tbytes else None,
                         # measured streaming bandwidth of this device (16-byte copy/triad kernel of the library)
                         "measured_stream_GBps": copy_bw,
                         "real_traffic_frac_of_measured_stream": (tbytes / (avg_ms * 1e-3) / 1e9 / copy_bw) if (tbytes and avg_ms > 0) else None,
                         "algorithmic_bytes_per_launch": alg_launch, "interior_dofs": n_int, "avg_launch_ms": avg_ms,
                         "launches": k_cnt,
                         "event_sampling": f"HIP events on the library's stream around every {EVENT_SAMPLE}th launch of the kernel inside "
                                           "the timed region (all four RK4 stage kinds sampled equally); launches = sampled launches",
                         # what THIS layout must move (compulsory_bytes above): the ceiling this design can be held to.
                         # `frac` / `achieved` stay SURVEY 8d's accounting figure against the reference's data path --
                         # an equivalent bandwidth, not a rate of bytes moved; read frac_real / frac_compulsory for that
                         "compulsory_bytes_per_launch": comp_blk,
                         "compulsory_detail": comp_det,
                         "frac_compulsory": (comp_blk / (avg_ms * 1e-3) / 8e12) if avg_ms > 0 else None,
                         "raw_FETCH_SIZE_bytes_per_launch": (traffic["read_bytes_per_launch"] / 2.0) if (traffic and "read_bytes_per_launch" in traffic) else None,
                         "fetch_size_correction": "x2 (MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half of a wide coalesced read on gfx950)",
                         "achieved_is": "algorithmic-equivalent GB/s (SURVEY 8d byte model / time), not bytes moved",
                         "achieved_exceeds_measured_stream": bool(achieved > copy_bw)},
            "step_roofline": {"algorithmic_bytes_per_dof_update": b_general,
                              "achieved_GBps": b_general * value / world / 1e9,
                              "frac_of_8TBps": b_general * value / world / 8e12,
                              "compulsory_bytes_per_dof_update": comp_step,
                              "compulsory_GBps": comp_step * value / world / 1e9,
                              "frac_compulsory": comp_step * value / world / 8e12},
            # per-operator-action rate, comparable to the reference's logged stiffness actions
            # (2.0e9 DOF/s on 76 Icelake cores, p=4 fp64); here one action also does the fused stage update
            "operator_action_dofs_per_s": (ndl / (avg_ms * 1e-3)) if avg_ms > 0 else None,
            "kernel_ms_per_step": prof["breakdown_ms_per_step"],
            "finite_nonzero_solution": finite,
        }
        # the step's second kernel: stage update of the rank-local shared dofs = their block partial sums (8.. planes at
        # the dof's own index) + the stage's model vectors; RK4 stage kinds 4, 5, 6, 3 stream 5, 7, 10, 6 vectors
        st_ms = prof["breakdown_ms_per_step"].get("stage", 0.0)
        if st_ms > 0 and args.model == "linear" and world == 1 and not args.halo_loopback:
            nsh, npairs = info["shared_dofs"], info["pairs"]
            st_bytes = (npairs + 7.0 * nsh) * s            # average stage of the four
            out["second_kernel"] = {"kernel": "k_shared_stage_planes (rank-local shared dofs: partial sums + fused RK4 stage)",
                                    "shared_dofs": int(nsh), "partial_sums": int(npairs), "avg_launch_ms": st_ms / 4,
                                    "algorithmic_bytes_per_launch": st_bytes,
                                    "achieved_GBps": st_bytes / (st_ms / 4 * 1e-3) / 1e9,
                                    "frac_of_8TBps": st_bytes / (st_ms / 4 * 1e-3) / 8e12,
                                    "share_of_step": st_ms / (1e3 * elapsed / args.steps)}
        if world > 1 or args.halo_loopback:
            # who moved the interface values: the library's own RCCL communicator (ncclSend / ncclRecv between `nranks`
            # ranks, replacing la::Vector::scatter_fwd / scatter_rev of Linear.hpp:196-206), or torch.distributed
            out["comm"] = {"transport": transport, "nranks": int(getattr(ctx, "nranks", world)),
                           "backend": ("gloo, interface values staged through host memory" if rehearsal else
                                       ("RCCL ncclSend/ncclRecv on the library's comm stream" if transport == "rccl"
                                        else "torch.distributed P2P (NCCL = RCCL)")),
                           "devices_visible": int(ndev)}
        if rehearsal:
            out["rehearsal"] = (f"{world} ranks shared {rehearsal} device(s); exchange staged through host memory over gloo "
                                "-- correctness rehearsal of the N>1 path, NOT a scaling measurement")
        if overlap_ab:
            out["exchange_overlap"] = {"what": "ms per step with the exchange hidden behind the shared-dof stage kernel only "
                                               "(off) or also behind the non-interface blocks (on), timed during warm-up",
                                       **overlap_ab}
        for e2, aff2, fin2, info2 in others:
            b2 = 4 * (rho_e * (s + 4 + geo_b[aff2] + extra_x * s) + s + (12 + extra_v) * s)
            v2 = ndofs_global * args.steps / e2
            cb2, cs2, _ = compulsory_bytes(info2, nc, N3, s, aff2, args.model, lean)
            c2 = 4.0 * (cb2 + cs2) / ndl
            key = {"affine": "other_geometry", "affine_diag": "other_geometry", "trilinear": "trilinear_geometry", "stream": "streamed_geometry"}[aff2]
            out[key] = {"geometry": GEOM_NAMES[aff2], "value": v2, "unit": "DOF-updates/s",
                        "ms_per_step": 1e3 * e2 / args.steps, "algorithmic_bytes_per_dof_update": b2,
                        "frac_of_8TBps": b2 * v2 / world / 8e12,
                        "compulsory_bytes_per_dof_update": c2, "frac_compulsory": c2 * v2 / world / 8e12,
                        "finite_nonzero_solution": fin2,
                        "options": {"mfma": bool(info2.get("mfma")), "pack32": bool(info2.get("pack32")),
                                    "diag_metric": bool(info2.get("diag_metric"))}}
        if not args.no_cpu and args.dtype == "f64" and world == 1 and not args.global_cells and not args.cells_xyz:   # CPU leg on rank 0 at N=1 only
            nd_cpu = (args.cpu_n * P + 1) ** 3
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_n, args.cpu_steps or max(4, int(15 * 9e7 / nd_cpu)))
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or launched:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
