/*
 * fusmi.h -- C ABI of libfusmi, the MI355X-native drop-in for the fenicsx-fus hot path:
 * sum-factorised mass/stiffness operator action on hex spectral elements + explicit RK4
 * stage update + shared-DOF halo exchange.  (SURVEY.md section 8b is the contract.)
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository adeebkor/fenicsx-fus @ 2024-10-08).  Plain pointers and sizes only; no C++ or
 * torch types cross this boundary.  All functions return FUS_OK (0) or a negative error code;
 * fus_last_error() returns the message of the calling thread's last failure.  Handles are not
 * thread-safe: one ctx/op/model per GPU per thread, like the reference's one object per MPI rank
 * (the reference operator holds mutable scratch, spectral_op.hpp:267-283).
 *
 * The library REQUIRES a HIP device: there is no CPU fallback.  Calls that need the device fail
 * with FUS_ERR_HIP when none is present.
 */
#ifndef FUSMI_H
#define FUSMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FUS_OK 0
#define FUS_ERR_ARG (-1)     /* invalid argument / unsupported P, tdim, dtype, geometry order */
#define FUS_ERR_HIP (-2)     /* HIP runtime error (including: no device) */
#define FUS_ERR_RCCL (-3)    /* RCCL error */
#define FUS_ERR_STATE (-4)   /* call sequence error (e.g. model used before init) */
#define FUS_ERR_LIMIT (-5)   /* a block does not fit the 160 KB LDS budget / index overflow */

enum { FUS_F32 = 0, FUS_F64 = 1 };           /* scalar type T of the reference templates */
enum { FUS_HOST = 0, FUS_DEVICE = 1 };       /* memory space of caller vectors */
enum { FUS_LINEAR = 0, FUS_LOSSY = 1, FUS_WESTERVELT = 2 };
enum { FUS_U = 0, FUS_V = 1 };

typedef struct fus_ctx fus_ctx;
typedef struct fus_op fus_op;
typedef struct fus_model fus_model;

const char* fus_last_error(void);
/* ABI version, for the binding to check. */
int fus_version(void);

/* ---- context --------------------------------------------------------------------------------
 * Replaces the implicit per-rank process state of the reference (MPI_COMM_WORLD rank,
 * Linear.hpp:64-65).  Binds to HIP device `device`, creates the compute and comm streams. */
int fus_init(int device, fus_ctx** ctx);
int fus_finalize(fus_ctx* ctx);
int fus_synchronize(fus_ctx* ctx);
/* Tunables, set before fus_op_create: "block_elems" (elements per LDS block) and "waves"
 * (waves per workgroup, 1..8): default 0 = auto (hexahedra with G streamed: 128 / 64 / 32 /
 * 20 / 12 / 8 elements at P = 2..7 in fp64, 128 / 64 / 48 / 24 / 24 / 16 in fp32, about half of
 * that -- 8 at P >= 5 -- on the affine and trilinear paths; 4 waves), "geometry" (0 auto: 7 numbers
 * per cell when every cell is a parallelepiped, else -- first-order hexahedra -- the 21 coefficients
 * of each cell's trilinear map with J and G recomputed per point in the kernel, else the streamed
 * per-point factors | 1 always stream the per-point factors, the reference's data path | 2 as auto
 * without the affine shortcut), "fields" (1 | 2: operator inputs the block kernel
 * stages per pass; 2 is required by FUS_LOSSY), "deterministic" (1: elements accumulate in
 * conflict-free rounds, results bitwise reproducible; 0 (default): LDS floating-point atomics, the
 * order of the <= 8 adds per DOF inside a block is free).
 * "forms" (set before fus_model_create; FUS_LOSSY / FUS_WESTERVELT): 0 (default) the C++ benchmark
 * forms -- absorbing and delta-mass terms on every listed boundary facet (BM7-SC1/forms.py:37-42),
 * source doubled (Lossy.hpp:216-220); 1 the Python package's -- those terms on tag 2 only, source
 * not doubled (python/src/fenicsxfus/_lossy.py:107-128, :186-189).
 * "external_transport" (1, before fus_comm_init: the caller exchanges the interface values, see below).
 * Multi-rank, set before fus_comm_init / fus_model_create: "overlap_blocks" (1: the blocks touching
 * interface DOFs are launched first and the exchange overlaps the remaining blocks; default 0: it
 * overlaps the shared-DOF kernel only), "halo_loopback" (1: timing rehearsal on one GPU -- a 1-rank
 * communicator, every send/receive goes to the own rank; results are not the physical ones).
 * "graph" (1: on one rank the launches of an RK step are captured and replayed as one hipGraph, the
 * executable graph being updated in place with each step's stage scalars; for launch-bound sizes
 * such as BASELINE config 1; default 0).
 * "lean_rk4" (set before fus_model_create; default 1): the classical RK4 keeps no accumulators u_, v_ of
 * Linear.hpp:282-294 in HBM -- the stage slopes are affine in the stage velocities, so the last stage builds the
 * new state from the three stage velocities (three rotating buffers), and u0 is rebuilt from the stage input the
 * kernel already holds in LDS (208 instead of 296 bytes of vector traffic per DOF and step, same arithmetic up
 * to rounding); 0 keeps them in HBM at every stage.  The Runge-Kutta orders 1-3 always keep them.
 * "mfma" (-1 auto (default) | 0 | 1, before fus_op_create): degrees 6 and 7 on the per-cell geometry
 * paths -- the index-1 / index-2 contractions of an element, the (N x N).(N x N^2) products of the
 * reference's contract<> (sum_factorisation.hpp:70-86), as 16x16x4 MFMA tiles on the matrix cores
 * instead of vector FMAs.  Auto = where it measures faster on MI355X: nowhere at present (the re-mapped vector form is
 * 3 % ahead of it at degree 7, fp64, trilinear geometry, the one case it used to win; profiles/r02_experiments.md).
 * "pack32" (-1 auto (default) | 0 | 1, before fus_op_create): fp32, degrees 5-7, per-cell geometry paths -- a
 * wave works on two elements at once, every tile exchange and FMA packed as float2 (half the LDS and vector
 * instructions per element of the scalar fp32 kernel, which is LDS bound).
 * "walk" (0 (default) | 1..8 | -1, any time): block-kernel workgroups per CU that walk several blocks each
 * with the next block's prologue loads in flight under the current block's epilogue; 0 = one workgroup
 * per block (measured faster everywhere so far, profiles/r02_experiments.md), -1 = as many as are resident.
 * "diag_metric" (1 (default) | 0, before fus_op_create): affine meshes whose cells have mutually orthogonal edges
 * take the diagonal-metric form of the stiffness kernel (fus_op_uses_diag_metric); 0 keeps the general affine form.
 * "planes" (1 (default) | 0 | 2..16, any time): the shared-dof stage kernel reads the block partial sums of a dof as
 * planes at the dof's own index (no index list; every access coalesced) or through the shared-dof CSR; the
 * sums and their order are the same.  The CSR form is also taken when a dof has more sharing blocks than the
 * kernel has planes (16; a value k = 2..16 lowers that limit to k).
 * Unknown keys -> FUS_ERR_ARG. */
int fus_set_option(fus_ctx* ctx, const char* key, int64_t value);

/* Multi-GPU: one process per GPU.  fus_comm_unique_id fills a 128-byte RCCL id on rank 0; the
 * caller broadcasts it (any transport) and every rank calls fus_comm_init.  Replaces
 * MPI_Init/MPI_COMM_WORLD inside PetscInitialize (BM7-SC1/main.cpp:22). */
int fus_comm_unique_id(void* id128);
int fus_comm_init(fus_ctx* ctx, int rank, int nranks, const void* id128);
/* Diagnostic: n doubles through grouped ncclSend/ncclRecv to the own rank on the library stream
 * (checks the run-time RCCL binding; RCCL is dlopen'ed, preferring a copy already resident in the
 * process such as PyTorch's, or $FUSMI_RCCL). */
int fus_comm_selftest(fus_ctx* ctx, int64_t n);
/* In-place all-reduce of n host doubles over the ranks of fus_comm_init (RCCL): the global minimum
 * cell size behind the time step (MPI_Reduce(MIN) + MPI_Bcast,
 * cpp/fenicsx-sf-naive/examples/linear_planewave2d_1/main.cpp:67-68) and the sums behind norms
 * (:151-157).  One rank: no-op.  In-process groups / external transport: FUS_ERR_STATE (the caller
 * holds every rank's value, or its own transport). */
#define FUS_SUM 0
#define FUS_MIN 1
#define FUS_MAX 2
int fus_comm_allreduce(fus_ctx* ctx, double* values, int n, int op);

/* In-process transport for rehearsing the multi-rank path on ONE GPU (tests): the n contexts of
 * this process become ranks 0..n-1 and interface planes move by device copies instead of RCCL.
 * Models of such contexts are finished with fus_group_finish_setup (the sharers' parts of m and
 * of the boundary weights) and advanced in lock-step with fus_group_rk4_steps. */
int fus_comm_init_local(fus_ctx** ctxs, int n);

/* ---- operator data ------------------------------------------------------------------------
 * Replaces the constructors StiffnessSpectral3D<T,P>(V) / MassSpectral3D<T,P>(V)
 * (cpp/fenicsx-sf/common/spectral_op.hpp:135-171, :32-63): takes the tensor-ordered cell dofmap
 * (what reorder_dofmap, permute.hpp:15-42, produces), the 1-D node coordinates on [0,1] in the
 * caller's local order (any order; the library derives GLL weights and the derivative table
 * for it, replacing tabulate_1d precompute.hpp:217-234) and the mesh geometry, and computes on
 * the device the scaled geometric factors G = J^-1 J^-T |detJ| w and |detJ| w
 * (compute_scaled_geometrical_factor / _jacobian_determinant, precompute.hpp:101-213, 33-94).
 * One object serves both operators (the reference builds one per operator, Lossy.hpp:152-153).
 *   tdim        3 (hexahedra) or 2 (quadrilaterals: StiffnessSpectral2D / MassSpectral2D,
 *               cpp/fenicsx-sf-naive/common/spectral_op.hpp:29-107, 226-359; N^2 nodes per cell,
 *               geom_dofmap int32[ncells * 4] with v = vx + 2vy (order 1) or int32[ncells * 9] with
 *               n = nx + 3ny (order 2, biquadratic), G has 3 entries (xx, xy, yy) per point, local
 *               facets 0..3 = y=0, x=0, x=1, y=1)
 *   P           polynomial degree 2..10 (the reference's Qdegree map, spectral_op.hpp:35-44); N = P+1 nodes per
 *               direction.  Degrees 8-10: first-order hexahedra through the per-cell geometry paths only
 *               ("geometry" 0 or 2); a tensor plane then has more than 64 columns and two waves share an element
 *   dtype       FUS_F64 | FUS_F32: type of geom_x and of every vector/coefficient argument later
 *   tensor_dofmap  int32[ncells * N^tdim], local DOF indices < ndofs, x-slowest tensor order
 *   nodes1d     double[N]
 *   geom_x      T[nnodes * 3]
 *   geom_order  1: geom_dofmap int32[ncells * 8], vertex order v = vx + 2vy + 4vz (DOLFINx's);
 *               2: geom_dofmap int32[ncells * 27], nodes in tensor order n = nx + 3ny + 9nz with
 *                  n_d in {0,1,2} <-> reference coordinate {0, 1/2, 1} (the caller permutes from
 *                  DOLFINx's vertices-edges-faces-interior order).  Other orders -> FUS_ERR_ARG.
 * All arrays are caller-owned host memory, copied during the call. */
int fus_op_create(fus_ctx* ctx, int tdim, int P, int dtype, int64_t ncells, int64_t ndofs,
                  const int32_t* tensor_dofmap, const double* nodes1d, const void* geom_x,
                  int64_t nnodes, const int32_t* geom_dofmap, int geom_order, fus_op** op);
int fus_op_destroy(fus_op* op);

/* y += K(coeffs) x.  Replaces StiffnessSpectral3D::operator()(x, coeffs, y)
 * (spectral_op.hpp:173-243): y is ACCUMULATED, x must hold every local DOF value (the caller's
 * scatter_fwd, Linear.hpp:196), coeffs has one scalar per local cell.  No inter-rank reduction is
 * done (the caller's scatter_rev, Linear.hpp:206).  x, coeffs, y: T arrays in `space`. */
int fus_stiffness_apply(fus_op* op, const void* x, const void* coeffs, void* y, int space);
/* y += M(coeffs) x.  Replaces MassSpectral3D::operator() (spectral_op.hpp:69-86). */
int fus_mass_apply(fus_op* op, const void* x, const void* coeffs, void* y, int space);

/* Inspection (parity tests): geometry factors in the REFERENCE layout G[cell][point][6]
 * (xx,xy,xz,yy,yz,zz), detJ[cell][point] (precompute.hpp:198-208); host T arrays, may be NULL. */
int fus_op_get_geometry(fus_op* op, void* G, void* detJ);
/* 1-D tables in the caller's node order: weights[N], dphi[N*N] row = point (spectral_op.hpp:168). */
int fus_op_get_tables(fus_op* op, double* weights, double* dphi);
/* Layout statistics: out[0]=nblocks out[1]=interior dofs out[2]=shared dofs out[3]=(block,dof)
 * pairs out[4]=max local dofs per block out[5]=unique block shapes out[6]=LDS bytes per block
 * out[7]=padded internal vector length. */
int fus_op_info(fus_op* op, int64_t out[8]);
/* 1 when every cell was found to be a parallelepiped and the operator rebuilds G = Gc w_q from 7
 * numbers per cell instead of streaming 6 per point (option "geometry" = 0, the default); 0
 * otherwise (see fus_op_geometry_mode). */
int fus_op_is_affine(fus_op* op);
/* Geometry source of the block operator: 0 = per-point factors streamed from HBM (the reference's
 * data path, precompute.hpp:101-213), 1 = affine cells (7 numbers per cell), 2 = first-order
 * hexahedra with the Jacobian recomputed per point from the cell's trilinear map (21 numbers per
 * cell; the default for first-order meshes with non-affine cells). */
int fus_op_geometry_mode(fus_op* op);
/* 1 when the operator's block kernel runs its index-1 / index-2 contractions on the matrix cores
 * (MFMA 16x16x4; degrees 6 and 7 on the per-cell geometry paths, option "mfma"). */
int fus_op_uses_mfma(fus_op* op);
/* 1 when the affine kernel runs in its diagonal-metric form: every cell a parallelepiped with mutually orthogonal
 * edges (boxes in any orientation; J^T J and with it G of spectral_op.hpp:113-130 are diagonal), degrees <= 7 --
 * the stiffness action as three 1-D stiffness contractions, sum_d g_d (M x K1 x M) x with K1 = D^T diag(w) D,
 * instead of the six derivative contractions and the pointwise transform (option "diag_metric"). */
int fus_op_uses_diag_metric(fus_op* op);
/* 1 when the block kernel runs its index-1 contraction on v_mfma_f64_4x4x4_4b_f64 straight from the registers (degree 7,
 * fp64, first-order hexahedra with non-affine cells -- the trilinear geometry kernel; the default there): the reference's
 * contract<T,8,8,8,8,bool> of spectral_op.hpp:199-201 / :222-227 (sum_factorisation.hpp:70-86) on the matrix cores. */
int fus_op_uses_mfma4(fus_op* op);
/* 1 when the fp32 stiffness kernel works on two elements per wave in packed float2 (degrees 5-7, per-cell
 * geometry paths, LDS-atomic accumulation; option "pack32"). */
int fus_op_uses_pack32(fus_op* op);
/* Smallest cell size of the local mesh, the size of a cell being its largest vertex-to-vertex
 * distance (dolfinx::mesh::h, linear_planewave2d_1/main.cpp:60-64); dt = CFL hmin / (c P^2), :102. */
int fus_op_hmin(fus_op* op, double* hmin);
/* out = sum over the local cells of the GLL-quadrature integral of x^2 (caller numbering; loc =
 * FUS_HOST | FUS_DEVICE); the sum over ranks is the squared L2 norm of
 * fem::assemble_scalar(u*u*dx), linear_planewave2d_1/main.cpp:151-157. */
int fus_op_norm2(fus_op* op, const void* x, int loc, double* out);

/* out[dof] += cellcoef[cell] * |J_facet| w_a w_b at the GLL nodes of each listed boundary facet,
 * facets given as (cell, local facet) pairs in DOLFINx numbering (hex: 0:z=0 1:y=0 2:x=0 3:x=1
 * 4:y=1 5:z=1), the pairs fem::compute_integration_domains returns (Linear.hpp:113-118).
 * Replaces the FFCx facet kernels of the GLL-collocated forms L/a (SC1-BM1/forms.py:36-39,
 * BM7-SC1/forms.py:37-42), which are diagonal.  Host arrays; out is T[ndofs]. */
int fus_facet_diag(fus_op* op, int64_t nfacets, const int32_t* facet_cells,
                   const int32_t* facet_local, const void* cellcoef, void* out);

/* Shared-DOF description for >1 rank (replaces the IndexMap ghost/owner data behind
 * la::Vector::scatter_fwd/scatter_rev, Linear.hpp:196-206): for neighbour k, the local DOF
 * indices shared with rank ranks[k], counts[k] of them, concatenated in dof_idx; both sides
 * must list a shared set in the same (global id) order.  Call once, right after fus_op_create and
 * before any model is created on the op (the block layout is rebuilt so that these DOFs are never
 * finished inside a block's fused epilogue). */
int fus_op_set_neighbours(fus_op* op, int nneigh, const int32_t* ranks, const int64_t* counts,
                          const int32_t* dof_idx);

/* ---- model ------------------------------------------------------------------------------------
 * Replaces LinearSpectral3D<T,P>(element, mesh, facet_tags, c0, rho0, freq, amp, speed)
 * (Linear.hpp:55-158): c0, rho0 are the DG0 arrays (T[ncells]); boundary facets as
 * (cell, local facet, tag) with tag 1 = source, 2 = absorbing (forms.py:38-39).  Builds the lumped
 * mass m (Linear.hpp:127-134) and the operator coefficient -1/rho (:154-155) on the device.
 * FUS_LOSSY replaces LossySpectral3D (Lossy.hpp:56-173): delta0 = diffusivity of sound per cell; the
 * two operator actions of a stage, lin_op(u_n, -1/rho) and att_op(v_n, -delta/(rho c^2))
 * (Lossy.hpp:231-232), run as ONE pass of the block kernel (both share G); absorbing term on every
 * listed facet, dg source term and the delta/(rho c^3) boundary mass term as in
 * BM7-SC1/forms.py:37-42; source scaling 2 W p0 w0/s0 as live in Lossy.hpp:216-220.  The operator
 * data must have been created with option "fields" = 2.
 * FUS_WESTERVELT replaces WesterveltSpectral3D (Westervelt.hpp:58-193): beta0 = coefficient of
 * nonlinearity per cell.  The per-stage mass re-assembly m = m0 + M(nlin1) u_n and the RHS term
 * M(nlin2)(v_n^2) (Westervelt.hpp:246-265) are diagonal (M(c) x = diag(M(c) 1) .* x), so they fold
 * into the fused stage update: kv = (b - mn1 v_n^2) / (m0 + mn1 u_n), mn1 = M(-2 beta/(rho^2 c^4)) 1. */
int fus_model_create(fus_ctx* ctx, int kind, fus_op* op, const void* c0, const void* rho0,
                     const void* delta0, const void* beta0, int64_t nfacets,
                     const int32_t* facet_cells, const int32_t* facet_local,
                     const int32_t* facet_tags, double freq, double amp, double speed,
                     fus_model** model);
int fus_model_destroy(fus_model* model);
/* Explicit Runge-Kutta scheme of the Python reference (python/src/fenicsxfus/_linear.py:286-311):
 * 1 forward Euler, 2 / 3 Ralston, 4 classical (default; the only one the C++ reference has,
 * Linear.hpp:263-265).  The entry points named rk4 run the selected scheme. */
int fus_model_set_rk_order(fus_model* model, int order);
/* u_n = v_n = 0 (Linear.hpp:161-164). */
int fus_model_init(fus_model* model);
/* Classical RK4 from t0 to tf with step dt, `while (t < tf) { dt = min(dt, tf - t); ... }`
 * (Linear.hpp:228-314); *nsteps receives the number of steps taken (may be NULL). */
int fus_model_rk4(fus_model* model, double t0, double tf, double dt, int64_t* nsteps);
/* Exactly nsteps full steps of size dt starting at t0 (benchmark entry; same stage arithmetic). */
int fus_model_rk4_steps(fus_model* model, double t0, double dt, int64_t nsteps);
/* Copy u_n (FUS_U) or v_n (FUS_V) out / in, caller DOF numbering, T[ndofs] in `space`
 * (u_sol(), Linear.hpp:316). */
int fus_model_get(fus_model* model, int which, void* out, int space);
int fus_model_set(fus_model* model, int which, const void* in, int space);
/* Lumped mass vector m in caller numbering (host T[ndofs]); parity inspection. */
int fus_model_get_mass(fus_model* model, void* out);
int64_t fus_model_ndofs(fus_model* model); /* number_of_dofs(), Linear.hpp:318 (local) */

/* ---- receivers: point samples of the resident solution ------------------------------------------
 * Replaces Function::eval(points, cells) after the cell search of compute_eval_params
 * (python/src/fenicsxfus/utils.py:10-47) / geometry::compute_colliding_cells
 * (cpp/mwe/parallel_eval_line/main.cpp:49-84): the caller locates each point (cell = local cell index in caller
 * numbering, refcoords = its reference coordinates in [0,1]^tdim, double[npts*tdim], X0 pairing with tensor index
 * 0 of the dofmap) and drops points outside the local mesh, as the reference does (points_on_proc).  The library
 * keeps, per receiver, the internal indices of its cell's dofs and the 1-D Lagrange basis values, and evaluates
 * u_h (FUS_U) or v_h (FUS_V) at the receivers from the vectors resident in HBM -- no full-vector copy.
 *   fus_model_sample       T[npts] now, into host or device memory (`space`)
 *   fus_model_record       sample `which` after every `every`-th step of fus_model_rk4 / fus_model_rk4_steps into a
 *                          device buffer of `capacity` records (every = 0: off); restarts the record count
 *   fus_model_get_records  copies the records taken so far (T[nrec*npts], row = record) and their times; out / times
 *                          may be NULL to query *nrec only */
int fus_model_set_receivers(fus_model* model, int64_t npts, const int32_t* cells, const double* refcoords);
int fus_model_sample(fus_model* model, int which, void* out, int space);
int fus_model_record(fus_model* model, int which, int every, int64_t capacity);
int fus_model_get_records(fus_model* model, void* out, double* times, int64_t* nrec);

int fus_group_finish_setup(fus_model** models, int n);
int fus_group_rk4_steps(fus_model** models, int n, double t0, double dt, int64_t nsteps);

/* ---- external transport ---------------------------------------------------------------------
 * For callers that move the interface values themselves -- GPU-aware MPI in the reference's setting
 * (its scatter_fwd/scatter_rev, Linear.hpp:196-206, are MPI neighbourhood exchanges) -- instead of
 * the built-in RCCL exchange.  fus_set_option(ctx, "external_transport", 1), then
 * fus_comm_init(ctx, rank, nranks, NULL), fus_op_create, fus_op_set_neighbours, fus_model_create.
 *   fus_op_halo_layout   neighbours in the order of the buffers: rank, number of values, offset (values)
 *   fus_op_halo_buffers  device pointers of the send / receive buffers (element type T) and their length;
 *                        neighbour k sends send[off_k .. off_k + count_k) and receives into the same
 *                        range of recv
 * Setup (once): for k in [0, fus_model_setup_count): fus_model_setup_pack(k) -> exchange ->
 * fus_model_setup_unpack(k); then fus_model_setup_finish, fus_model_init.
 * Every RK stage i of a step at time t: fus_model_stage_begin(i, t, dt) -> exchange ->
 * fus_model_stage_end(i, t, dt).  *_pack / stage_begin return with the send buffer complete; the
 * receive buffer must be complete when *_unpack / stage_end are called.  Every sharer adds the ranks'
 * values in ascending rank order, so all of them end with identical bits. */
int fus_op_halo_layout(fus_op* op, int* nneigh, int32_t* ranks, int64_t* counts, int64_t* offsets);
int fus_op_halo_buffers(fus_op* op, void** send_dev, void** recv_dev, int64_t* nvalues);
int fus_model_setup_count(fus_model* model);
int fus_model_setup_pack(fus_model* model, int k);
int fus_model_setup_unpack(fus_model* model, int k);
int fus_model_setup_finish(fus_model* model);
int fus_model_stage_begin(fus_model* model, int stage, double t, double dt);
int fus_model_stage_end(fus_model* model, int stage, double t, double dt);

/* ---- measurement -----------------------------------------------------------------------------
 * HIP-event timing of the library's own kernels on the stream they run on.  Names:
 * "stiffness" (block operator kernel), "shared" (shared-DOF reduction), "stage" (fused RK stage
 * update), "boundary", "halo".  total_ms/count accumulate since the last enable.
 * on = 1: every kernel; on = 2: only the block operator kernel ("stiffness", and "stiffness_if" when
 * the interface blocks are launched separately) -- an event record drains the queue between two
 * kernels, so timed runs use 2 (bench.py) and take the full breakdown in a separate pass.  Option
 * "profile_sample" = k (fus_set_option, default 1): level 2 puts its events around every k-th launch of the
 * block operator kernel only (k = 5 samples the four RK4 stage kinds equally and costs a timed run 0.3 % instead
 * of 1.7 %); fus_profile_get then returns the time and the count of the sampled launches. */
int fus_profile_enable(fus_ctx* ctx, int on);
int fus_profile_get(fus_ctx* ctx, const char* name, double* total_ms, int64_t* count);
/* Measured streaming bandwidth of the device (16-byte non-temporal copy of nbytes, one vector per
 * thread; bytes read + bytes written per second, best of reps launches, GB/s): reported beside the
 * roofline fractions. */
int fus_measure_bandwidth(fus_ctx* ctx, int64_t nbytes, int reps, double* gbps);

/* Host-only layout builder (no device needed): runs the block partitioner / DOF renumbering on
 * a dofmap and returns statistics as fus_op_info does; used by the CPU test-suite. */
int fus_layout_check(int P, int64_t ncells, int64_t ndofs, const int32_t* tensor_dofmap,
                     const double* centroids /* [ncells*3] */, int block_elems, int waves,
                     int64_t out[8]);
/* The same for tdim = 2 | 3 and with the optional mask of DOFs other ranks hold as well
 * (force_shared, uint8[ndofs] or NULL): those DOFs are classified shared and the blocks touching
 * them come first in the layout (fus_op_set_neighbours does this on the device path). */
int fus_layout_check_ex(int tdim, int P, int64_t ncells, int64_t ndofs, const int32_t* tensor_dofmap,
                        const double* centroids, int block_elems, int waves,
                        const uint8_t* force_shared, int64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif
