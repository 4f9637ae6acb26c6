// fusmi.hip -- C ABI of libfusmi (include/fusmi.h) and host orchestration: handles, dtype/degree
// dispatch, kernel launches, RK4 loop, halo exchange over RCCL.
//
// Reference counterparts: StiffnessSpectral3D / MassSpectral3D (spectral_op.hpp:29-107, 132-284),
// LinearSpectral3D (Linear.hpp:52-347).  There is no CPU fallback in this file: every compute
// entry point needs the HIP device and fails with FUS_ERR_HIP without one.
#include "../../include/fusmi.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <limits>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <memory>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "layout.hpp"
#include "tables.hpp"

using namespace fus;

// -------------------------------------------------------------------------------------------------
// errors
// -------------------------------------------------------------------------------------------------
// One shared object is linked from several translation units of this file: the "main" unit (C ABI,
// degree-independent code) and one unit per polynomial degree (-DFUS_TU_DEGREE=k) holding the
// k_block_op instantiations of that degree, so that the degrees compile in parallel (build.py).
#define FUS_HIDDEN __attribute__((visibility("hidden")))
#ifdef FUS_TU_DEGREE
extern FUS_HIDDEN thread_local std::string g_err;
#else
FUS_HIDDEN thread_local std::string g_err;
#endif
static int fail(int code, const std::string& msg)
{
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                               \
  do                                                                                               \
  {                                                                                                \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(FUS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
  } while (0)
#define NCCLCHK(expr)                                                                              \
  do                                                                                               \
  {                                                                                                \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess)                                                                         \
      return fail(FUS_ERR_RCCL, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));            \
  } while (0)
#define FUSCHK(expr)                                                                               \
  do                                                                                               \
  {                                                                                                \
    int r_ = (expr);                                                                               \
    if (r_ != FUS_OK)                                                                              \
      return r_;                                                                                   \
  } while (0)

// -------------------------------------------------------------------------------------------------
// RCCL, bound at run time.  A host process may already carry an RCCL (PyTorch bundles its own
// librccl.so); binding to the resident copy instead of linking a second one keeps a single RCCL
// in the process.  Order: $FUSMI_RCCL (explicit path), an already loaded librccl, the system one.
// -------------------------------------------------------------------------------------------------
struct RcclApi
{
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
#ifdef FUS_TU_DEGREE
extern FUS_HIDDEN RcclApi g_rccl;
#else
FUS_HIDDEN RcclApi g_rccl;

static int rccl_load()
{
  if (g_rccl.handle)
    return FUS_OK;
  void* h = nullptr;
  if (const char* path = getenv("FUSMI_RCCL"))
    h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  for (const char* name : {"librccl.so.1", "librccl.so"})
    if (!h)
      h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
    if (!h)
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  if (!h)
    return fail(FUS_ERR_RCCL, std::string("cannot load librccl: ") + dlerror());
  RcclApi a;
  a.handle = h;
#define FUS_SYM(field, name)                                                                       \
  *reinterpret_cast<void**>(&a.field) = dlsym(h, name);                                            \
  if (!a.field)                                                                                    \
    return fail(FUS_ERR_RCCL, "librccl lacks " name);
  FUS_SYM(GetUniqueId, "ncclGetUniqueId")
  FUS_SYM(CommInitRank, "ncclCommInitRank")
  FUS_SYM(CommDestroy, "ncclCommDestroy")
  FUS_SYM(Send, "ncclSend")
  FUS_SYM(Recv, "ncclRecv")
  FUS_SYM(GroupStart, "ncclGroupStart")
  FUS_SYM(GroupEnd, "ncclGroupEnd")
  FUS_SYM(AllReduce, "ncclAllReduce")
  FUS_SYM(GetErrorString, "ncclGetErrorString")
#undef FUS_SYM
  g_rccl = a;
  return FUS_OK;
}
#endif  // !FUS_TU_DEGREE

// -------------------------------------------------------------------------------------------------
// handles
// -------------------------------------------------------------------------------------------------
struct Prof
{
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  double done_ms = 0;
  int64_t done_count = 0;
  int64_t seen = 0;   // scopes of this name since fus_profile_enable (level 2 samples every prof_sample-th)
};

struct fus_ctx
{
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t comm_stream = nullptr;            // RCCL exchange, overlapped with local shared dofs
  hipEvent_t ev_packed = nullptr, ev_recv = nullptr;
  int deterministic = 0;  // 1: conflict-free rounds (bitwise reproducible); 0: LDS atomics
  int fields = 1;         // operator inputs per block pass the ops are sized for (2: Lossy)
  int graph = 0;          // 1: replay the RK step as one hipGraph (launch-bound sizes)
  int geometry = 0;       // 0: auto (affine cells: 7 numbers per cell; other first-order hexahedra: the
                          // cell's trilinear map, G recomputed per point; else streamed), 1: always
                          // stream G, 2: as auto without the affine shortcut
  // 0 = auto: 32 elements / 4 waves when G is streamed, 16 / 4 on the affine path (measured best
  // on MI355X at p=4 fp64, profiles/r01_block_sweep.txt)
  int block_elems = 0, waves = 0;
  int prof = 0;  // 0 off, 1 all scopes, 2 block-operator kernel only
  int prof_sample = 1;  // level 2: events around every prof_sample-th launch of the block-operator kernel (option "profile_sample")
  std::map<std::string, Prof> profs;
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  bool local_group = false;  // in-process transport (single-GPU rehearsal of the multi-rank path)
  // timing rehearsal of the RCCL exchange on one GPU (option "halo_loopback"): the context keeps its
  // logical (rank, nranks) but owns a 1-rank communicator and every send / receive goes to itself,
  // so a middle slab exercises pack -> ncclSend/ncclRecv -> ordered unpack with realistic launch
  // and synchronisation cost (the received planes are its own: results are NOT the physical ones)
  bool loopback = false;
  bool overlap_blocks = false;  // launch interface blocks first, overlap the exchange with the rest
  bool external_transport = false;  // the caller exchanges the packed interface values (e.g. GPU-aware MPI)
  int num_cus = 256;                // hipDeviceProp_t::multiProcessorCount
  // Lossy / Westervelt boundary forms: 0 = the C++ benchmarks (BM7-SC1/forms.py:37-42: absorbing and
  // delta-mass terms on every boundary facet, source doubled, Lossy.hpp:216-220); 1 = the Python
  // package (python/src/fenicsxfus/_lossy.py:107-128, :186-189: those terms on tag 2 only, source
  // not doubled)
  int forms = 0;
  // classical RK4 without the redundant accumulator streams (stage kinds 4-6, kernels.hpp): 1 (default);
  // 0 keeps u_, v_ in HBM at every stage like Linear.hpp:282-294
  int lean_rk4 = 1;
  // index-1 / index-2 contractions of the degrees 6 and 7 on the matrix cores (per-cell geometry kernels):
  // -1 auto (the measured choice per degree, scalar type and geometry), 0 never, 1 wherever a variant exists
  int mfma = -1;
  int pack32 = -1;  // fp32, degrees 5-7, per-cell geometry: two elements per wave in packed float2 (-1 auto, 0, 1)
  // shared-dof stage kernel reads the partial sums as planes at the dof's own index (1, default) or through the
  // shared-dof CSR (0; also taken when a dof has more than FUS_MAX_PLANES sharing blocks): same sums, same order
  int planes = 1;
  // affine meshes whose cells have mutually orthogonal edges (boxes): the stiffness action in its diagonal-metric
  // form (three 1-D stiffness contractions, kernels.hpp elem_compute) -- 1 (default) where the mesh allows, 0 never
  int diag_metric = 1;
  int walk = 0;   // block-kernel workgroups per CU that walk several blocks each (0: one workgroup per block --
                  // the measured best everywhere so far; -1: as many as are resident), see launch_block_op_v
};

struct Neigh
{
  int rank;
  int64_t count;
  int64_t off;  // first entry of this neighbour in the send / receive buffers
};

struct fus_op
{
  fus_ctx* ctx;
  int P, N, Nd, dtype;
  int tdim = 3;  // 3: hexahedra, 2: quadrilaterals (Nd = N^tdim)
  size_t ts;  // sizeof(T)
  int64_t ncells, ndofs, nnodes;
  Layout L;
  std::vector<double> nodes, wts, D;
  std::vector<char> h_geom_x;        // caller geometry (host copy, T)
  std::vector<int32_t> h_geom_dm;    // [ncells*geom_nv]
  int geom_order = 1, geom_nv = 8;   // 1: 2^tdim vertices; 2: 27 nodes in tensor order (hexahedra)
  std::vector<int32_t> h_dofmap;     // caller tensor dofmap
  // device
  BlockArgs A{};
  std::vector<void*> allocs;
  int32_t *d_cell_perm = nullptr, *d_dof_perm = nullptr;
  int64_t *d_sh_ptr = nullptr, *d_sh_pairs = nullptr;
  void *d_G = nullptr, *d_detJ = nullptr, *d_Dg = nullptr, *d_partial = nullptr;
  void *d_tmp_x = nullptr, *d_tmp_b = nullptr, *d_tmp_c = nullptr, *d_tmp_coef = nullptr;
  size_t lds_bytes = 0;
  int deterministic = 0;
  int nfields = 1;
  bool affine = false;     // GEOM_AFFINE path in use (d_Gc), streamed G/detJ built only on demand
  bool trilinear = false;  // GEOM_TRILINEAR path in use (d_Gc holds 21 map coefficients per cell)
  bool mfma = false;       // MFMA contraction variants of the block kernel in use (degrees 6, 7)
  bool ortho_mesh = false; // every cell a parallelepiped with mutually orthogonal edges (found in op_build)
  bool diag = false;       // GEOM_AFFINE in its diagonal-metric form (ortho_mesh, degrees <= 7, option "diag_metric")
  bool pk = false;         // packed fp32 variants in use (degrees 5-7: two elements per wave, layout slots doubled)
  void* d_Gc = nullptr;
  void *d_xg = nullptr, *d_pts = nullptr, *d_wts = nullptr;
  int32_t* d_xdm = nullptr;
  // neighbours (multi-GPU)
  std::vector<Neigh> neigh;
  // halo buffers: concatenated neighbour lists (ascending rank), one send and one receive buffer;
  // unique interface dofs with, per dof, its addends in ascending rank order (-1 = own partial)
  int64_t n_halo = 0, n_uidx = 0;
  int32_t *d_pack_idx = nullptr, *d_uidx = nullptr, *d_uptr = nullptr, *d_usrc = nullptr;
  void *d_sendbuf = nullptr, *d_recvbuf = nullptr;
  // the pseudo partial slots (next-stage boundary terms of shared boundary dofs) live in d_partial,
  // i.e. per op, while several models may share one op: the model that wrote them last
  const struct fus_model* bnd_owner = nullptr;
};

struct fus_model
{
  fus_ctx* ctx;
  fus_op* op;
  int kind;
  hipGraphExec_t gexec = nullptr;  // option "graph": the RK step as one executable graph (see d_model_step)
  int64_t gsteps = 0;              // steps taken since init / set (the first one is launched directly)
  double freq, amp, speed;
  void *u0 = nullptr, *v0 = nullptr, *u_ = nullptr, *v_ = nullptr, *un = nullptr, *vn = nullptr,
       *b = nullptr, *minv = nullptr, *m = nullptr, *coef = nullptr, *coef2 = nullptr;
  void* d_bsrc2 = nullptr;  // lossy: delta/(rho c^2) w_f on the source facets (dg term)
  void* mn1 = nullptr;      // Westervelt: diag of M(-2 beta/(rho^2 c^4)), internal numbering
  // boundary dofs (diagonal source / absorbing weights), sorted by internal index:
  // [0, nb_int) are block-interior (applied in the fused epilogue through d_blk_bnd_off),
  // [nb_int, nb) are shared dofs (their terms ride in pseudo partial slots: boundary_next / k_boundary_partial)
  int64_t nb = 0, nb_int = 0;
  int32_t* d_bidx = nullptr;
  int32_t* d_blk_bnd_off = nullptr;
  int32_t *d_sh_ptr32 = nullptr, *d_sh_pairs32 = nullptr;  // shared CSR incl. boundary pseudo pairs
  uint64_t* d_bnd_mask = nullptr;                           // rank-local shared dofs: boundary dof bits ...
  int32_t* d_bnd_base = nullptr;                            // ... and boundary dofs ahead of each 64-dof word
  void *d_bsrc = nullptr, *d_babs = nullptr;
  std::vector<void*> allocs;
  bool initialised = false;
  bool setup_done = false;
  int rk_order = 4;  // explicit Runge-Kutta scheme, tables of python/src/fenicsxfus/_linear.py:286-311
  int forms = 0;     // fus_ctx::forms at creation
  int lean_rk4 = 1;  // fus_ctx::lean_rk4 at creation
  // The boundary term of the shared boundary dofs rides in pseudo partial slots.  With RK4 the
  // shared-dof stage kernels write the NEXT stage's values there (boundary_next, kernels.hpp);
  // bnd_valid / bnd_tn say for which stage time the slots are current, and stage_begin launches
  // k_boundary_partial only when they are not (first stage after init / set, other RK orders).
  bool bnd_valid = false;
  double bnd_tn = 0.0;
  // receivers (fus_model_set_receivers): per point the internal indices of its cell's dofs and the 1-D basis
  // values at its reference coordinates; record buffer for sampling every k steps inside the RK loops
  int64_t rc_n = 0;
  int32_t* d_rc_idx = nullptr;
  void *d_rc_bas = nullptr, *d_rc_out = nullptr, *d_rec = nullptr;
  int rec_every = 0, rec_which = 0;
  int64_t rec_cap = 0, rec_n = 0, rec_step = 0;
  std::vector<double> rec_times;
};

// -------------------------------------------------------------------------------------------------
// small helpers
// -------------------------------------------------------------------------------------------------
template <typename U>
static int dalloc(std::vector<void*>& pool, U** p, size_t n)
{
  void* q = nullptr;
  HIPCHK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(U)));
  pool.push_back(q);
  *p = static_cast<U*>(q);
  return FUS_OK;
}
static int dalloc_bytes(std::vector<void*>& pool, void** p, size_t bytes, bool zero, hipStream_t st)
{
  void* q = nullptr;
  HIPCHK(hipMalloc(&q, std::max<size_t>(bytes, 16)));
  pool.push_back(q);
  if (zero)
    HIPCHK(hipMemsetAsync(q, 0, std::max<size_t>(bytes, 16), st));
  *p = q;
  return FUS_OK;
}
template <typename U>
static int upload(std::vector<void*>& pool, U** p, const std::vector<U>& v, hipStream_t st)
{
  FUSCHK(dalloc(pool, p, v.size()));
  if (!v.empty())
    HIPCHK(hipMemcpyAsync(*p, v.data(), v.size() * sizeof(U), hipMemcpyHostToDevice, st));
  return FUS_OK;
}
static inline unsigned nblk(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

struct ProfScope
{
  fus_ctx* c;
  Prof* p = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ProfScope(fus_ctx* c_, const char* name) : c(c_)
  {
    // level 1: every scope; level 2: the dominant kernel only ("stiffness", "stiffness_if") -- each
    // event record drains the queue between two kernels, so a timed run keeps them to a minimum
    bool take = c->prof == 1;
    if (c->prof == 2 && !strncmp(name, "stiffness", 9))
      take = (c->profs[name].seen++ % c->prof_sample) == 0;
    if (take)
    {
      p = &c->profs[name];
      (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, c->stream);
    }
  }
  ~ProfScope()
  {
    if (p)
    {
      (void)hipEventRecord(e1, c->stream);
      p->ev.emplace_back(e0, e1);
    }
  }
};

// -------------------------------------------------------------------------------------------------
// typed implementation
// -------------------------------------------------------------------------------------------------
template <typename T, int P, int OP, int ATOMIC, int STAGE, int NF, int GEOM, int TD = 3, int MF = 0, int PK = 0>
static int launch_block_op_v(fus_op* op, const T* geo, const T* coef, const T* x, T* bvec,
                             const StageArgs<T>& S, int blk_begin, int blk_count)
{
  if (blk_count <= 0)
    return FUS_OK;
  constexpr int N = P + 1;
  KArgs<T, N> K;
  for (int i = 0; i < N * N; ++i)
    K.Dk.d[i] = (T)op->D[i], K.Dk.dt[i] = (T)op->D[(i % N) * N + i / N];

  for (int i = 0; i < N; ++i)
    K.Dk.w[i] = (T)op->wts[i], K.Dk.x[i] = (T)op->nodes[i];
  // the attribute is per device: one bit per device and instantiation (set again harmlessly if two
  // threads race on the first launch)
  static std::atomic<uint64_t> attr_set{0};
  const uint64_t dev_bit = 1ull << (op->ctx->device & 63);
  if (!(attr_set.load(std::memory_order_relaxed) & dev_bit))
  {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_block_op<T, P, OP, ATOMIC, STAGE, NF, GEOM, TD, MF, PK>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  K.A = op->A;
  if (GEOM == GEOM_DIAG)
  {
    // the 1-D stiffness matrix K1 = D^T diag(w) D takes the derivative table's place (symmetric: d = dt)
    for (int q = 0; q < N; ++q)
      for (int i = 0; i < N; ++i)
      {
        double acc = 0;
        for (int m = 0; m < N; ++m)
          acc += op->D[m * N + q] * op->wts[m] * op->D[m * N + i];
        K.Dk.d[q * N + i] = K.Dk.dt[q * N + i] = (T)acc;
      }
  }
  K.A.blk_begin = blk_begin;
  K.A.blk_count = blk_count;
  K.Dg = static_cast<const T*>(op->d_Dg), K.geo = geo, K.coef = coef, K.x = x, K.bvec = bvec;
  K.partial = static_cast<T*>(op->d_partial);
  K.S = S;
  // option "walk" = w > 0: w workgroups per CU, each walking every (w * CUs)-th block of the range with
  // the next block's prologue loads in flight under the current block's epilogue (per-cell geometry
  // kernels; the streamed-geometry kernel keeps one workgroup per block).  Off by default: measured on
  // MI355X (config 2, profiles/r02_experiments.md) one workgroup per block is 13-17 % faster -- the
  // hardware's own dispatch of a fresh workgroup into a freed slot overlaps the phases of different
  // blocks better than the walking workgroup's software pipeline does.
  int grid = blk_count;
  if (GEOM != GEOM_STREAM && TD == 3 && op->ctx->walk != 0)
  {
    int per_cu = op->ctx->walk;
    if (per_cu < 0)  // as many workgroups per CU as are resident at once, where each then has >= 4 blocks to walk
    {
      // resident workgroups per CU of this instantiation, per (device, waves per workgroup, LDS bytes)
      static std::mutex occ_mu;
      static std::map<std::tuple<int, int, size_t>, int> occ_cache;
      int occ = 0;
      {
        std::lock_guard<std::mutex> lk(occ_mu);
        const auto key = std::make_tuple(op->ctx->device, op->L.waves, op->lds_bytes);
        auto it = occ_cache.find(key);
        if (it == occ_cache.end())
        {
          int nb = 0;
          HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(
              &nb, reinterpret_cast<const void*>(&k_block_op<T, P, OP, ATOMIC, STAGE, NF, GEOM, TD, MF, PK>), 64 * op->L.waves,
              op->lds_bytes));
          it = occ_cache.emplace(key, std::max(nb, 1)).first;
        }
        occ = it->second;
      }
      per_cu = (blk_count >= 4 * occ * op->ctx->num_cus) ? occ : 0;
    }
    if (per_cu > 0)
      grid = std::min(blk_count, per_cu * op->ctx->num_cus);
  }
  hipLaunchKernelGGL((k_block_op<T, P, OP, ATOMIC, STAGE, NF, GEOM, TD, MF, PK>), dim3(grid),
                     dim3(64 * op->L.waves), op->lds_bytes, op->ctx->stream, K);
  HIPCHK(hipGetLastError());
  return FUS_OK;
}

// Blocks [blk_begin, blk_begin + blk_count) of the layout; blk_count < 0: all blocks.
template <typename T, int P, int OP, int STAGE, int NF = 1>
static int launch_block_op(fus_op* op, const T* geo, const T* coef, const T* x, T* bvec,
                           const StageArgs<T>& S, int blk_begin = 0, int blk_count = -1)
{
  if (NF > op->nfields)
    return fail(FUS_ERR_STATE, "operator data was not created for two-field models (option fields=2)");
  if (blk_count < 0)
    blk_count = op->L.nblocks - blk_begin;
  const int b0 = blk_begin, nb = blk_count;
  if constexpr (P >= 8)
  {
    // degrees 8-10: two waves per element (kernels.hpp, elem_compute_hi); per-cell geometry where the mesh allows,
    // per-point factors streamed otherwise (second-order geometry, option "geometry" = 1)
    if (op->tdim == 2)   // quadrilaterals: an element's N^2 nodes span two waves (elem_compute2d, HI)
      return op->deterministic
                 ? launch_block_op_v<T, P, OP, 0, STAGE, NF, GEOM_STREAM, 2>(op, geo, coef, x, bvec, S, b0, nb)
                 : launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_STREAM, 2>(op, geo, coef, x, bvec, S, b0, nb);
    if (!(op->affine || op->trilinear))
      return op->deterministic
                 ? launch_block_op_v<T, P, OP, 0, STAGE, NF, GEOM_STREAM>(op, geo, coef, x, bvec, S, b0, nb)
                 : launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_STREAM>(op, geo, coef, x, bvec, S, b0, nb);
    const T* gc = static_cast<const T*>(op->d_Gc);
    if (op->affine)
      return op->deterministic
                 ? launch_block_op_v<T, P, OP, 0, STAGE, NF, GEOM_AFFINE>(op, gc, coef, x, bvec, S, b0, nb)
                 : launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_AFFINE>(op, gc, coef, x, bvec, S, b0, nb);
    return op->deterministic
               ? launch_block_op_v<T, P, OP, 0, STAGE, NF, GEOM_TRILINEAR>(op, gc, coef, x, bvec, S, b0, nb)
               : launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_TRILINEAR>(op, gc, coef, x, bvec, S, b0, nb);
  }
  else
  {
  if (op->tdim == 2)  // quadrilaterals: streamed geometry only
    return op->deterministic
               ? launch_block_op_v<T, P, OP, 0, STAGE, NF, GEOM_STREAM, 2>(op, geo, coef, x, bvec, S, b0, nb)
               : launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_STREAM, 2>(op, geo, coef, x, bvec, S, b0, nb);
  // degrees 6 and 7, per-cell geometry, LDS-atomic accumulation: the index-1 / index-2 contractions on the
  // matrix cores (kernels.hpp, elem_compute_mfma) where option "mfma" / the measured default says so; the
  // variants exist for the stiffness operator as the plain action and as the lean RK4 stages
  if constexpr ((P == 6 || P == 7) && OP == OP_STIFFNESS && (STAGE == STAGE_NONE || STAGE >= 3))
  {
    if (op->mfma && !op->deterministic && op->tdim == 3 && (op->affine || op->trilinear))
    {
      const T* gc = static_cast<const T*>(op->d_Gc);
      return op->affine ? launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_AFFINE, 3, 1>(op, gc, coef, x, bvec, S, b0, nb)
                        : launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_TRILINEAR, 3, 1>(op, gc, coef, x, bvec, S, b0, nb);
    }
  }
  // fp32, degrees 5-7, per-cell geometry, LDS-atomic accumulation: two elements per wave in packed float2
  // (kernels.hpp, elem_compute_pk); stiffness operator only (the mass action runs through the scalar kernel)
  if constexpr (sizeof(T) == 4 && P >= 5 && P <= 7 && OP == OP_STIFFNESS)
  {
    if (op->pk && !op->mfma && !op->deterministic && op->tdim == 3 && (op->affine || op->trilinear))
    {
      const T* gc = static_cast<const T*>(op->d_Gc);
      return op->affine ? launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_AFFINE, 3, 0, 1>(op, gc, coef, x, bvec, S, b0, nb)
                        : launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_TRILINEAR, 3, 0, 1>(op, gc, coef, x, bvec, S, b0, nb);
    }
  }
  // geometry operand: per-cell factors (affine meshes) or the streamed per-point arrays
  if constexpr (OP == OP_STIFFNESS && P <= 7)
  {
    if (op->affine && op->diag)   // cells with orthogonal edges: diagonal-metric form of the stiffness action
    {
      const T* gc = static_cast<const T*>(op->d_Gc);
      return op->deterministic
                 ? launch_block_op_v<T, P, OP, 0, STAGE, NF, GEOM_DIAG>(op, gc, coef, x, bvec, S, b0, nb)
                 : launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_DIAG>(op, gc, coef, x, bvec, S, b0, nb);
    }
  }
  if (op->affine)
  {
    const T* gc = static_cast<const T*>(op->d_Gc);
    return op->deterministic
               ? launch_block_op_v<T, P, OP, 0, STAGE, NF, GEOM_AFFINE>(op, gc, coef, x, bvec, S, b0, nb)
               : launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_AFFINE>(op, gc, coef, x, bvec, S, b0, nb);
  }
  if (op->trilinear)
  {
    const T* gc = static_cast<const T*>(op->d_Gc);
    return op->deterministic
               ? launch_block_op_v<T, P, OP, 0, STAGE, NF, GEOM_TRILINEAR>(op, gc, coef, x, bvec, S, b0, nb)
               : launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_TRILINEAR>(op, gc, coef, x, bvec, S, b0, nb);
  }
  return op->deterministic
             ? launch_block_op_v<T, P, OP, 0, STAGE, NF, GEOM_STREAM>(op, geo, coef, x, bvec, S, b0, nb)
             : launch_block_op_v<T, P, OP, 1, STAGE, NF, GEOM_STREAM>(op, geo, coef, x, bvec, S, b0, nb);
  }
}

template <typename T>
static int shared_reduce(fus_op* op, T* bvec)
{
  if (op->L.n_shared > 0)
  {
    ProfScope ps(op->ctx, "shared");
    hipLaunchKernelGGL((k_shared_reduce<T, int64_t>), dim3(nblk(op->L.n_shared)), dim3(256), 0,
                       op->ctx->stream, (int64_t)0, op->L.n_shared, op->d_sh_ptr, op->d_sh_pairs,
                       static_cast<const T*>(op->d_partial), bvec + op->L.n_int_pad);
    HIPCHK(hipGetLastError());
  }
  return FUS_OK;
}

// b_internal = A x_internal  (all dofs: interior written by the block kernel, shared reduced)
template <typename T, int P, int OP>
static int apply_internal(fus_op* op, const T* coef, const T* x, T* bvec)
{
  fus_ctx* c = op->ctx;
  {
    ProfScope ps(c, OP == OP_STIFFNESS ? "stiffness" : "mass");
    const T* geo = static_cast<const T*>(OP == OP_STIFFNESS ? op->d_G : op->d_detJ);
    StageArgs<T> none{};
    FUSCHK((launch_block_op<T, P, OP, STAGE_NONE>(op, geo, coef, x, bvec, none)));
  }
  return shared_reduce<T>(op, bvec);
}

// Shared-DOF exchange of a partial-sum vector (replaces b->scatter_rev(std::plus) +
// the two scatter_fwd of Linear.hpp:196-206): every sharer ends with the identical total because
// each adds the partials in ascending rank order.  Three phases so that the transport can sit
// between them: pack (own partials -> send buffers), exchange (RCCL send/recv over xGMI, or
// device copies for the in-process transport), unpack (ordered sum).
template <typename T>
static int halo_pack(fus_op* op, const T* vec)
{
  if (op->neigh.empty())
    return FUS_OK;
  fus_ctx* c = op->ctx;
  hipLaunchKernelGGL((k_pack<T>), dim3(nblk(op->n_halo)), dim3(256), 0, c->stream, op->n_halo,
                     op->d_pack_idx, vec, static_cast<T*>(op->d_sendbuf));
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(c->ev_packed, c->stream));
  return FUS_OK;
}

// grouped send/recv with every neighbour on the comm stream, ordered after the pack by an event
static int halo_exchange_rccl(fus_op* op)
{
  fus_ctx* c = op->ctx;
  if (op->neigh.empty())
    return FUS_OK;
  if (!c->comm)
    return fail(FUS_ERR_STATE, "neighbours set but fus_comm_init was not called");
  const ncclDataType_t dt = op->ts == 8 ? ncclDouble : ncclFloat;
  HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_packed, 0));
  NCCLCHK(g_rccl.GroupStart());
  for (auto& nb : op->neigh)
  {
    const int peer = c->loopback ? 0 : nb.rank;
    NCCLCHK(g_rccl.Send(static_cast<char*>(op->d_sendbuf) + nb.off * op->ts, nb.count, dt, peer,
                        c->comm, c->comm_stream));
    NCCLCHK(g_rccl.Recv(static_cast<char*>(op->d_recvbuf) + nb.off * op->ts, nb.count, dt, peer,
                        c->comm, c->comm_stream));
  }
  NCCLCHK(g_rccl.GroupEnd());
  HIPCHK(hipEventRecord(c->ev_recv, c->comm_stream));
  return FUS_OK;
}

template <typename T>
static int halo_unpack(fus_op* op, T* vec)
{
  fus_ctx* c = op->ctx;
  if (op->neigh.empty())
    return FUS_OK;
  if (!c->local_group)
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_recv, 0));
  hipLaunchKernelGGL((k_unpack_ordered<T>), dim3(nblk(op->n_uidx)), dim3(256), 0, c->stream,
                     op->n_uidx, op->d_uidx, op->d_uptr, op->d_usrc,
                     static_cast<const T*>(op->d_recvbuf), vec);
  HIPCHK(hipGetLastError());
  return FUS_OK;
}

// RCCL transport: the three phases back to back
template <typename T>
static int halo_sum(fus_op* op, T* vec)
{
  if (op->neigh.empty())
    return FUS_OK;
  if (op->ctx->local_group)
    return fail(FUS_ERR_STATE, "in-process transport: use the fus_group_* entry points");
  ProfScope ps(op->ctx, "halo");
  FUSCHK(halo_pack<T>(op, vec));
  FUSCHK(halo_exchange_rccl(op));
  return halo_unpack<T>(op, vec);
}

// In-process transport: after every member has packed, move each send buffer to the matching
// receive buffer of the peer op (device copy), then every member unpacks.
static int halo_exchange_local(fus_op** ops, int n)
{
  for (int i = 0; i < n; ++i)
    HIPCHK(hipStreamSynchronize(ops[i]->ctx->stream));
  for (int i = 0; i < n; ++i)
    for (auto& nb : ops[i]->neigh)
    {
      fus_op* peer = nullptr;
      for (int j = 0; j < n; ++j)
        if (ops[j]->ctx->rank == nb.rank)
          peer = ops[j];
      if (!peer)
        return fail(FUS_ERR_STATE, "neighbour rank not in the local group");
      Neigh* back = nullptr;
      for (auto& pn : peer->neigh)
        if (pn.rank == ops[i]->ctx->rank)
          back = &pn;
      if (!back || back->count != nb.count)
        return fail(FUS_ERR_STATE, "asymmetric neighbour lists");
      HIPCHK(hipMemcpyAsync(static_cast<char*>(peer->d_recvbuf) + back->off * peer->ts,
                            static_cast<const char*>(ops[i]->d_sendbuf) + nb.off * ops[i]->ts,
                            nb.count * ops[i]->ts, hipMemcpyDeviceToDevice, ops[i]->ctx->stream));
    }
  // device-to-device copies need not block the host: wait for them before anyone unpacks
  for (int i = 0; i < n; ++i)
    HIPCHK(hipStreamSynchronize(ops[i]->ctx->stream));
  return FUS_OK;
}

// Per-point geometry factors in the streaming layouts (precompute.hpp:101-213, 33-94), computed on
// the device.  Built at setup for non-affine meshes, on demand (fus_op_get_geometry) otherwise.
template <typename T, int P>
static int ensure_stream_geometry(fus_op* op)
{
  constexpr int N = P + 1;
  if (op->d_G)
    return FUS_OK;
  hipStream_t st = op->ctx->stream;
  const int Nd = op->Nd, ng = op->tdim == 3 ? 6 : 3;
  FUSCHK(dalloc_bytes(op->allocs, &op->d_G, (size_t)op->ncells * ng * Nd * sizeof(T), false, st));
  FUSCHK(dalloc_bytes(op->allocs, &op->d_detJ, (size_t)op->ncells * Nd * sizeof(T), false, st));
  if (op->tdim == 2 && op->geom_order == 1)
    hipLaunchKernelGGL((k_geometry2d<T, N, 1>), dim3(nblk(op->ncells * Nd)), dim3(256), 0, st,
                       op->ncells, op->d_cell_perm, static_cast<const T*>(op->d_xg), op->d_xdm,
                       static_cast<const double*>(op->d_pts), static_cast<const double*>(op->d_wts),
                       static_cast<T*>(op->d_G), static_cast<T*>(op->d_detJ));
  else if (op->tdim == 2)
    hipLaunchKernelGGL((k_geometry2d<T, N, 2>), dim3(nblk(op->ncells * Nd)), dim3(256), 0, st,
                       op->ncells, op->d_cell_perm, static_cast<const T*>(op->d_xg), op->d_xdm,
                       static_cast<const double*>(op->d_pts), static_cast<const double*>(op->d_wts),
                       static_cast<T*>(op->d_G), static_cast<T*>(op->d_detJ));
  else if (op->geom_order == 1)
    hipLaunchKernelGGL((k_geometry<T, N, 1>), dim3(nblk(op->ncells * Nd)), dim3(256), 0, st,
                       op->ncells, op->d_cell_perm, static_cast<const T*>(op->d_xg), op->d_xdm,
                       static_cast<const double*>(op->d_pts), static_cast<const double*>(op->d_wts),
                       static_cast<T*>(op->d_G), static_cast<T*>(op->d_detJ));
  else
    hipLaunchKernelGGL((k_geometry<T, N, 2>), dim3(nblk(op->ncells * Nd)), dim3(256), 0, st,
                       op->ncells, op->d_cell_perm, static_cast<const T*>(op->d_xg), op->d_xdm,
                       static_cast<const double*>(op->d_pts), static_cast<const double*>(op->d_wts),
                       static_cast<T*>(op->d_G), static_cast<T*>(op->d_detJ));
  HIPCHK(hipGetLastError());
  return FUS_OK;
}

// Measured choice (MI355X, profiles/r02_experiments.md sections 5 and 15) between the vector and the matrix-core
// form of the index-1 / index-2 contractions.
static bool mfma_default(int P, bool f64, bool affine)
{
  // Nowhere at present.  Degree 7, fp64, trilinear geometry was the one case where the matrix-core form won
  // (-5.8 % against the tile-read vector form); with the re-mapped vector contractions at N = 8 (derivative-table
  // rows by scalar loads) the vector form is 2.7 % ahead: 1.423 against 1.469 ms per launch at 64^3.  Affine
  // geometry at degree 7 (+6 %: little vector work to relieve) and fp32 at degree 6 (+45 %: the f32 MFMA issues
  // at the vector-FMA rate) were slower from the start.  Option "mfma" = 1 still selects the variants.
  (void)P, (void)f64, (void)affine;
  return false;
}

template <typename T, int P>
static int op_setup_device(fus_op* op)
{
  constexpr int N = P + 1;
  fus_ctx* c = op->ctx;
  hipStream_t st = c->stream;
  Layout& L = op->L;
  auto& pool = op->allocs;

  std::vector<ShapeDev> shapes(L.shapes.size());
  for (size_t i = 0; i < shapes.size(); ++i)
  {
    shapes[i].nelem = L.shapes[i].nelem, shapes[i].nloc = L.shapes[i].nloc;
    shapes[i].nint = L.shapes[i].nint, shapes[i].nrounds = L.shapes[i].nrounds;
    shapes[i].rounds_off = L.shapes[i].rounds_off, shapes[i].ldm_off = L.shapes[i].ldm_off;
  }
  int32_t *d_blk_shape, *d_elem_off, *d_int_off, *d_sh_gidx;
  int64_t* d_sh_off;
  ShapeDev* d_shapes;
  int16_t* d_rounds;
  uint16_t* d_ldm;
  FUSCHK(upload(pool, &d_blk_shape, L.blk_shape, st));
  FUSCHK(upload(pool, &d_shapes, shapes, st));
  FUSCHK(upload(pool, &d_elem_off, L.blk_elem_off, st));
  FUSCHK(upload(pool, &d_int_off, L.blk_int_off, st));
  FUSCHK(upload(pool, &d_sh_off, L.blk_sh_off, st));
  FUSCHK(upload(pool, &d_sh_gidx, L.sh_gidx, st));
  FUSCHK(upload(pool, &d_rounds, L.rounds, st));
  L.ldm.resize((L.ldm.size() + 15) & ~(size_t)15);  // 16-byte tail for the vector copy
  FUSCHK(upload(pool, &d_ldm, L.ldm, st));
  FUSCHK(upload(pool, &op->d_cell_perm, L.cell_perm, st));
  FUSCHK(upload(pool, &op->d_dof_perm, L.dof_perm, st));
  FUSCHK(upload(pool, &op->d_sh_ptr, L.sh_ptr, st));
  {
    // the CSR the kernels read holds where each pair's partial sum is stored (Layout::pair_pos), not its pair id
    std::vector<int64_t> pos(L.sh_pairs.size());
    for (size_t k = 0; k < pos.size(); ++k)
      pos[k] = L.pair_pos[L.sh_pairs[k]];
    FUSCHK(upload(pool, &op->d_sh_pairs, pos, st));
  }
  int32_t* d_sh_ppos;
  FUSCHK(upload(pool, &d_sh_ppos, L.pair_pos, st));
  op->A.sh_ppos = d_sh_ppos;
  op->A.blk_shape = d_blk_shape, op->A.shapes = d_shapes, op->A.blk_elem_off = d_elem_off;
  op->A.blk_int_off = d_int_off, op->A.blk_sh_off = d_sh_off, op->A.sh_gidx = d_sh_gidx;
  op->A.rounds = d_rounds, op->A.ldm = d_ldm, op->A.nblocks = L.nblocks;
  op->A.lds_nloc = (L.max_nloc + 1) & ~1;
  op->A.waves = L.waves;
  op->A.lds_nelem = (L.max_nelem + 7) & ~7;
  op->lds_bytes = L.lds_bytes(sizeof(T), op->nfields);
  if (op->lds_bytes > 160 * 1024)
    return fail(FUS_ERR_LIMIT, "block does not fit 160 KB of LDS; lower block_elems");

  // derivative table followed by the 1-D weights and points (the per-cell geometry paths rebuild
  // w_q / J(q) from them)
  std::vector<T> Dg(N * N + 2 * N);
  for (int i = 0; i < N * N; ++i)
    Dg[i] = (T)op->D[i];
  for (int i = 0; i < N; ++i)
    Dg[N * N + i] = (T)op->wts[i], Dg[N * N + N + i] = (T)op->nodes[i];
  T* d_Dg;
  FUSCHK(upload(pool, &d_Dg, Dg, st));
  op->d_Dg = d_Dg;

  // mesh geometry stays on the device for the (possibly deferred) per-point factors
  T* d_xg;
  FUSCHK(dalloc(pool, &d_xg, (size_t)op->nnodes * 3));
  HIPCHK(hipMemcpyAsync(d_xg, op->h_geom_x.data(), (size_t)op->nnodes * 3 * sizeof(T),
                        hipMemcpyHostToDevice, st));
  double *d_pts, *d_wts;
  FUSCHK(upload(pool, &op->d_xdm, op->h_geom_dm, st));
  FUSCHK(upload(pool, &d_pts, op->nodes, st));
  FUSCHK(upload(pool, &d_wts, op->wts, st));
  op->d_xg = d_xg, op->d_pts = d_pts, op->d_wts = d_wts;
  op->d_G = op->d_detJ = nullptr;

  // per-cell factors + affinity test (every cell a parallelepiped?)
  T* d_Gc;
  unsigned int* d_err;
  FUSCHK(dalloc(pool, &d_Gc, (size_t)op->ncells * 21));
  FUSCHK(dalloc(pool, &d_err, 1));
  HIPCHK(hipMemsetAsync(d_err, 0, sizeof(unsigned int), st));
  float rel_err = 1.0f;
  if (op->geom_order == 1 && op->tdim == 3)
  {
    hipLaunchKernelGGL((k_geometry_affine<T>), dim3(nblk(op->ncells)), dim3(256), 0, st, op->ncells,
                       op->d_cell_perm, d_xg, op->d_xdm, d_Gc, d_err);
    HIPCHK(hipGetLastError());
    unsigned int err_bits = 0;
    HIPCHK(hipMemcpyAsync(&err_bits, d_err, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(&rel_err, &err_bits, sizeof(float));
  }
  op->d_Gc = d_Gc;
  op->affine = c->geometry == 0 && rel_err <= (sizeof(T) == 8 ? 1e-12f : 1e-6f);
  op->trilinear = !op->affine && c->geometry != 1 && op->geom_order == 1 && op->tdim == 3;
  // matrix-core contraction variants: degrees 6 and 7 on the per-cell geometry paths; "auto" follows the
  // A/B measurements on MI355X (profiles/r02_mfma.md)
  op->mfma = (P == 6 || P == 7) && (op->affine || op->trilinear) && !op->deterministic
             && (c->mfma == 1 || (c->mfma < 0 && mfma_default(P, sizeof(T) == 8, op->affine)));
  op->diag = op->affine && op->ortho_mesh && c->diag_metric && P <= 7 && !op->mfma && !op->pk;
  if (op->affine)
    op->lds_bytes = L.lds_bytes(sizeof(T), op->nfields, 7);
  else if (op->trilinear)
  {
    hipLaunchKernelGGL((k_geometry_trilinear<T>), dim3(nblk(op->ncells)), dim3(256), 0, st, op->ncells,
                       op->d_cell_perm, d_xg, op->d_xdm, d_Gc);
    HIPCHK(hipGetLastError());
    op->lds_bytes = L.lds_bytes(sizeof(T), op->nfields, 21);
  }
  else
    FUSCHK((ensure_stream_geometry<T, P>(op)));

  FUSCHK(dalloc_bytes(pool, &op->d_partial, (size_t)(L.n_partial + L.n_shared) * sizeof(T), true, st));
  FUSCHK(dalloc_bytes(pool, &op->d_tmp_x, (size_t)L.n_internal * sizeof(T), true, st));
  FUSCHK(dalloc_bytes(pool, &op->d_tmp_b, (size_t)L.n_internal * sizeof(T), true, st));
  FUSCHK(dalloc_bytes(pool, &op->d_tmp_c, (size_t)op->ndofs * sizeof(T), true, st));
  FUSCHK(dalloc_bytes(pool, &op->d_tmp_coef, (size_t)op->ncells * sizeof(T) * 2, true, st));
  HIPCHK(hipStreamSynchronize(st));
  return FUS_OK;
}

// y += A(coeffs) x with caller-numbered vectors (the reference operator call)
template <typename T, int P, int OP>
static int op_apply(fus_op* op, const void* x, const void* coeffs, void* y, int space)
{
  fus_ctx* c = op->ctx;
  hipStream_t st = c->stream;
  T* xin = static_cast<T*>(op->d_tmp_x);
  T* bint = static_cast<T*>(op->d_tmp_b);
  T* tc = static_cast<T*>(op->d_tmp_c);
  T* coef_c = static_cast<T*>(op->d_tmp_coef);
  T* coef_i = coef_c + op->ncells;
  const T* xc = static_cast<const T*>(x);
  const T* cc = static_cast<const T*>(coeffs);
  if (space == FUS_HOST)
  {
    HIPCHK(hipMemcpyAsync(tc, x, op->ndofs * sizeof(T), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(coef_c, coeffs, op->ncells * sizeof(T), hipMemcpyHostToDevice, st));
    xc = tc, cc = coef_c;
  }
  hipLaunchKernelGGL((k_to_internal<T>), dim3(nblk(op->ndofs)), dim3(256), 0, st, op->ndofs,
                     op->d_dof_perm, xc, xin);
  hipLaunchKernelGGL((k_cells_to_internal<T>), dim3(nblk(op->ncells)), dim3(256), 0, st, op->ncells,
                     op->d_cell_perm, cc, coef_i);
  FUSCHK((apply_internal<T, P, OP>(op, coef_i, xin, bint)));
  if (space == FUS_HOST)
  {
    HIPCHK(hipMemcpyAsync(tc, y, op->ndofs * sizeof(T), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL((k_from_internal<T, 1>), dim3(nblk(op->ndofs)), dim3(256), 0, st, op->ndofs,
                       op->d_dof_perm, bint, tc);
    HIPCHK(hipMemcpyAsync(y, tc, op->ndofs * sizeof(T), hipMemcpyDeviceToHost, st));
  }
  else
    hipLaunchKernelGGL((k_from_internal<T, 1>), dim3(nblk(op->ndofs)), dim3(256), 0, st, op->ndofs,
                       op->d_dof_perm, bint, static_cast<T*>(y));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  return FUS_OK;
}

template <typename T, int P>
static int op_get_geometry(fus_op* op, void* G, void* detJ)
{
  constexpr int N = P + 1;
  hipStream_t st = op->ctx->stream;
  FUSCHK((ensure_stream_geometry<T, P>(op)));
  const int Nd = op->Nd, ng = op->tdim == 3 ? 6 : 3;
  std::vector<void*> tmp;
  T *dG = nullptr, *dd = nullptr;
  if (G)
    FUSCHK(dalloc(tmp, &dG, (size_t)op->ncells * Nd * ng));
  if (detJ)
    FUSCHK(dalloc(tmp, &dd, (size_t)op->ncells * Nd));
  if (op->tdim == 2)
    hipLaunchKernelGGL((k_geometry_export2d<T, N>), dim3(nblk(op->ncells * Nd)), dim3(256), 0, st,
                       op->ncells, op->d_cell_perm, static_cast<const T*>(op->d_G),
                       static_cast<const T*>(op->d_detJ), dG, dd);
  else
    hipLaunchKernelGGL((k_geometry_export<T, N>), dim3(nblk(op->ncells * Nd)), dim3(256), 0, st,
                       op->ncells, op->d_cell_perm, static_cast<const T*>(op->d_G),
                       static_cast<const T*>(op->d_detJ), dG, dd);
  HIPCHK(hipGetLastError());
  if (G)
    HIPCHK(hipMemcpyAsync(G, dG, (size_t)op->ncells * Nd * ng * sizeof(T), hipMemcpyDeviceToHost, st));
  if (detJ)
    HIPCHK(hipMemcpyAsync(detJ, dd, (size_t)op->ncells * Nd * sizeof(T), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (void* q : tmp)
    (void)hipFree(q);
  return FUS_OK;
}

// Host builder of the diagonal facet weights (setup)
template <typename T>
static void facet_diag_host(const fus_op* op, int64_t nfacets, const int32_t* fc, const int32_t* fl,
                            const T* cellcoef, T* out)
{
  static const int axis3[6] = {2, 1, 0, 0, 1, 2}, side3[6] = {0, 0, 0, 1, 1, 1};
  const int N = op->N, Nd = op->Nd;
  const T* xg = reinterpret_cast<const T*>(op->h_geom_x.data());
  int i_lo = 0, i_hi = 0;
  for (int i = 0; i < N; ++i)
  {
    if (op->nodes[i] < op->nodes[i_lo])
      i_lo = i;
    if (op->nodes[i] > op->nodes[i_hi])
      i_hi = i;
  }
  if (op->tdim == 2)
  {
    // edges of quadrilaterals (local facet 0: y=0, 1: x=0, 2: x=1, 3: y=1): weight = edge-length
    // element |dx/dX_t| times the 1-D quadrature weight
    static const int axis2[4] = {1, 0, 0, 1}, side2[4] = {0, 0, 1, 1};
    for (int64_t f = 0; f < nfacets; ++f)
    {
      const int64_t cell = fc[f];
      const int ax = axis2[fl[f]], sd = side2[fl[f]], d1 = 1 - ax;
      T cd[9][3];
      const int nv2 = op->geom_nv;
      for (int v = 0; v < nv2; ++v)
        for (int j = 0; j < 3; ++j)
          cd[v][j] = xg[3 * (int64_t)op->h_geom_dm[cell * nv2 + v] + j];
      for (int a = 0; a < N; ++a)
      {
        int idx[2];
        idx[ax] = sd ? i_hi : i_lo, idx[d1] = a;
        T J[2][2];
        if (op->geom_order == 1)
          jacobian2<T>(reinterpret_cast<const T(*)[3]>(cd), op->nodes[idx[0]], op->nodes[idx[1]], J);
        else
          jacobian2_q2<T>(cd, op->nodes[idx[0]], op->nodes[idx[1]], J);
        const T len = (T)std::sqrt((double)(J[0][d1] * J[0][d1] + J[1][d1] * J[1][d1]));
        out[op->h_dofmap[cell * Nd + idx[0] * N + idx[1]]] += cellcoef[cell] * len * (T)op->wts[a];
      }
    }
    return;
  }
  for (int64_t f = 0; f < nfacets; ++f)
  {
    const int64_t cell = fc[f];
    const int ax = axis3[fl[f]], sd = side3[fl[f]];
    const int d1 = (ax + 1) % 3, d2 = (ax + 2) % 3;
    T cd[27][3];
    const int nv = op->geom_nv;
    for (int v = 0; v < nv; ++v)
      for (int j = 0; j < 3; ++j)
        cd[v][j] = xg[3 * (int64_t)op->h_geom_dm[cell * nv + v] + j];
    for (int a = 0; a < N; ++a)
      for (int b = 0; b < N; ++b)
      {
        int idx[3];
        idx[ax] = sd ? i_hi : i_lo, idx[d1] = a, idx[d2] = b;
        T J[3][3];
        if (op->geom_order == 1)
          jacobian3<T>(reinterpret_cast<const T(*)[3]>(cd), op->nodes[idx[0]], op->nodes[idx[1]],
                       op->nodes[idx[2]], J);
        else
          jacobian3_q2<T>(cd, op->nodes[idx[0]], op->nodes[idx[1]], op->nodes[idx[2]], J);
        const T t1[3] = {J[0][d1], J[1][d1], J[2][d1]}, t2[3] = {J[0][d2], J[1][d2], J[2][d2]};
        const T n0 = t1[1] * t2[2] - t1[2] * t2[1], n1 = t1[2] * t2[0] - t1[0] * t2[2],
                n2 = t1[0] * t2[1] - t1[1] * t2[0];
        const T area = (T)std::sqrt((double)(n0 * n0 + n1 * n1 + n2 * n2));
        const int li = (idx[0] * N + idx[1]) * N + idx[2];
        out[op->h_dofmap[cell * Nd + li]] += cellcoef[cell] * area * (T)(op->wts[a] * op->wts[b]);
      }
  }
}

template <typename T, int P>
static int model_setup(fus_model* m, const void* c0_, const void* rho0_, const void* delta0_,
                       const void* beta0_, int64_t nfacets, const int32_t* fc, const int32_t* fl,
                       const int32_t* ft)
{
  fus_op* op = m->op;
  fus_ctx* c = m->ctx;
  hipStream_t st = c->stream;
  const Layout& L = op->L;
  const int64_t n = L.n_internal;
  const T* c0 = static_cast<const T*>(c0_);
  const T* rho0 = static_cast<const T*>(rho0_);
  const T* delta0 = static_cast<const T*>(delta0_);
  const bool lossy = m->kind == FUS_LOSSY || m->kind == FUS_WESTERVELT;
  const T* beta0 = static_cast<const T*>(beta0_);
  auto& pool = m->allocs;
  for (void** v : {&m->u0, &m->v0, &m->u_, &m->v_, &m->un, &m->vn, &m->b, &m->minv, &m->m})
    FUSCHK(dalloc_bytes(pool, v, n * sizeof(T), true, st));

  // operator coefficients -1/rho (Linear.hpp:154-155) [and -delta/(rho c^2), Lossy.hpp:166-169]
  // and the mass coefficient 1/(rho c^2) (forms.py:36), internal element order
  std::vector<T> coef(op->ncells), coef2(lossy ? op->ncells : 0), mcoef(op->ncells);
  for (int64_t e = 0; e < op->ncells; ++e)
  {
    const int64_t cell = L.cell_perm[e];
    coef[e] = T(-1.0) / rho0[cell];
    if (lossy)
      coef2[e] = -delta0[cell] / rho0[cell] / c0[cell] / c0[cell];
    mcoef[e] = T(1.0) / rho0[cell] / c0[cell] / c0[cell];
  }
  T *d_coef, *d_coef2 = nullptr, *d_mcoef;
  FUSCHK(upload(pool, &d_coef, coef, st));
  if (lossy)
    FUSCHK(upload(pool, &d_coef2, coef2, st));
  FUSCHK(upload(pool, &d_mcoef, mcoef, st));
  m->coef = d_coef, m->coef2 = d_coef2;

  // lumped mass, this rank's cells only: m = M(1/(rho c^2)) 1  (Linear.hpp:127-133)
  T* ones = static_cast<T*>(m->un);
  hipLaunchKernelGGL((k_fill<T>), dim3(1024), dim3(256), 0, st, n, ones, T(1));
  FUSCHK((apply_internal<T, P, OP_MASS>(op, d_mcoef, ones, static_cast<T*>(m->m))));
  if (m->kind == FUS_WESTERVELT)
  {
    // diagonal of the nonlinear mass action: mn1 = M(-2 beta/(rho^2 c^4)) 1  (Westervelt.hpp:185,
    // 249-254); nlin2 = -nlin1 (:186), so the RHS term is -mn1 .* v_n^2
    std::vector<T> n1(op->ncells);
    for (int64_t e = 0; e < op->ncells; ++e)
    {
      const int64_t cell = L.cell_perm[e];
      const T cc = c0[cell], rr = rho0[cell];
      n1[e] = T(-2.0) * beta0[cell] / rr / rr / cc / cc / cc / cc;
    }
    T* d_n1;
    FUSCHK(upload(pool, &d_n1, n1, st));
    FUSCHK(dalloc_bytes(pool, &m->mn1, n * sizeof(T), true, st));
    FUSCHK((apply_internal<T, P, OP_MASS>(op, d_n1, ones, static_cast<T*>(m->mn1))));
  }
  HIPCHK(hipMemsetAsync(m->un, 0, n * sizeof(T), st));

  // boundary weights of this rank's facets (diagonal: GLL collocation)
  //   Linear (SC1-BM1/forms.py:38-39): src = (1/rho) w_f on tag 1, abs = (1/(rho c)) w_f on tag 2
  //   Lossy  (BM7-SC1/forms.py:37-42): abs on EVERY boundary facet, src2 = (delta/(rho c^2)) w_f on
  //   tag 1 (dg term) and the mass gains (delta/(rho c^3)) w_f on every boundary facet
  std::vector<T> src(op->ndofs, T(0)), absb(op->ndofs, T(0)), src2, mb;
  std::vector<T> cs(op->ncells), ca(op->ncells), cs2, cm;
  for (int64_t k = 0; k < op->ncells; ++k)
    cs[k] = T(1.0) / rho0[k], ca[k] = T(1.0) / rho0[k] / c0[k];
  if (lossy)
  {
    src2.assign(op->ndofs, T(0)), mb.assign(op->ndofs, T(0));
    cs2.resize(op->ncells), cm.resize(op->ncells);
    for (int64_t k = 0; k < op->ncells; ++k)
    {
      cs2[k] = delta0[k] / rho0[k] / c0[k] / c0[k];
      cm[k] = delta0[k] / rho0[k] / c0[k] / c0[k] / c0[k];
    }
  }
  std::vector<int32_t> c1, l1, c2, l2;
  for (int64_t f = 0; f < nfacets; ++f)
  {
    if (fc[f] < 0 || fc[f] >= op->ncells || fl[f] < 0 || fl[f] > 5)
      return fail(FUS_ERR_ARG, "facet (cell, local facet) out of range");
    if (ft[f] == 1)
      c1.push_back(fc[f]), l1.push_back(fl[f]);
    if ((lossy && m->forms == 0) || ft[f] == 2)
      c2.push_back(fc[f]), l2.push_back(fl[f]);   // lossy, C++ forms: plain ds = every listed facet
  }
  facet_diag_host<T>(op, (int64_t)c1.size(), c1.data(), l1.data(), cs.data(), src.data());
  facet_diag_host<T>(op, (int64_t)c2.size(), c2.data(), l2.data(), ca.data(), absb.data());
  if (lossy)
  {
    facet_diag_host<T>(op, (int64_t)c1.size(), c1.data(), l1.data(), cs2.data(), src2.data());
    facet_diag_host<T>(op, (int64_t)c2.size(), c2.data(), l2.data(), cm.data(), mb.data());
  }
  T* tmpc = static_cast<T*>(op->d_tmp_c);
  // scratch until fus_model_init: full-length src / abs / src2 weights; b holds the mass term
  void* dst[4] = {m->u_, m->v_, m->vn, m->b};
  std::vector<T>* hv[4] = {&src, &absb, &src2, &mb};
  for (int k = 0; k < (lossy ? 4 : 2); ++k)
  {
    HIPCHK(hipMemcpyAsync(tmpc, hv[k]->data(), op->ndofs * sizeof(T), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL((k_to_internal<T>), dim3(nblk(op->ndofs)), dim3(256), 0, st, op->ndofs,
                       op->d_dof_perm, tmpc, static_cast<T*>(dst[k]));
    HIPCHK(hipStreamSynchronize(st));
  }
  if (lossy)
  {
    hipLaunchKernelGGL((k_add_vec<T>), dim3(1024), dim3(256), 0, st, n, static_cast<const T*>(m->b),
                       static_cast<T*>(m->m));
    HIPCHK(hipMemsetAsync(m->b, 0, n * sizeof(T), st));
  }
  HIPCHK(hipGetLastError());
  return FUS_OK;
}

// Setup vectors: m (needs the sharers' contributions, m.scatter_rev(+), Linear.hpp:134), src
// weights (parked in u_) and abs weights (parked in v_) of this rank's own facets
static void* setup_halo_vector(fus_model* m, int k) { return k == 0 ? m->m : (k == 1 ? m->mn1 : m->v_); }

template <typename T>
static int model_setup_finish(fus_model* m)
{
  fus_op* op = m->op;
  hipStream_t st = m->ctx->stream;
  const int64_t n = op->L.n_internal;
  hipLaunchKernelGGL((k_reciprocal<T>), dim3(nblk(n)), dim3(256), 0, st, n,
                     static_cast<const T*>(m->m), static_cast<T*>(m->minv));
  const bool lossy = m->kind == FUS_LOSSY || m->kind == FUS_WESTERVELT;
  std::vector<T> src(n), absb(n), src2(lossy ? n : 0);
  HIPCHK(hipMemcpyAsync(src.data(), m->u_, n * sizeof(T), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(absb.data(), m->v_, n * sizeof(T), hipMemcpyDeviceToHost, st));
  if (lossy)
    HIPCHK(hipMemcpyAsync(src2.data(), m->vn, n * sizeof(T), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  std::vector<int32_t> bidx;
  std::vector<T> bsrc, babs, bsrc2;
  for (int64_t i = 0; i < n; ++i)
    if (src[i] != T(0) || absb[i] != T(0) || (lossy && src2[i] != T(0)))
    {
      bidx.push_back((int32_t)i);
      bsrc.push_back(src[i]);
      babs.push_back(absb[i]);
      if (lossy)
        bsrc2.push_back(src2[i]);
    }
  m->nb = (int64_t)bidx.size();
  {
    const Layout& L = op->L;
    m->nb_int = std::lower_bound(bidx.begin(), bidx.end(), (int32_t)L.n_int_pad) - bidx.begin();
    std::vector<int32_t> off(L.nblocks + 1);
    for (int32_t b = 0; b < L.nblocks; ++b)
      off[b] = (int32_t)(std::lower_bound(bidx.begin(), bidx.begin() + m->nb_int, L.blk_int_off[b])
                         - bidx.begin());
    off[L.nblocks] = (int32_t)m->nb_int;
    FUSCHK(upload(m->allocs, &m->d_blk_bnd_off, off, st));
    // shared CSR of this model: where the op's pairs keep their partial sums + one trailing pseudo pair (slot
    // n_partial + k) for the k-th shared boundary dof, so the boundary term is the last addend like
    // Linear.hpp:204-205
    if (L.n_partial + L.n_shared > 2000000000ll)
      return fail(FUS_ERR_LIMIT, "partial slab exceeds int32 indexing");
    std::vector<int32_t> extra(L.n_shared, -1);
    for (int64_t k = m->nb_int; k < m->nb; ++k)
      extra[bidx[k] - L.n_int_pad] = (int32_t)(L.n_partial + (k - m->nb_int));
    std::vector<int32_t> ptr(L.n_shared + 1), prs;
    prs.reserve(L.npairs + (m->nb - m->nb_int));
    for (int64_t sidx = 0; sidx < L.n_shared; ++sidx)
    {
      ptr[sidx] = (int32_t)prs.size();
      for (int64_t k = L.sh_ptr[sidx]; k < L.sh_ptr[sidx + 1]; ++k)
        prs.push_back(L.pair_pos[L.sh_pairs[k]]);
      if (extra[sidx] >= 0)
        prs.push_back(extra[sidx]);
    }
    ptr[L.n_shared] = (int32_t)prs.size();
    FUSCHK(upload(m->allocs, &m->d_sh_ptr32, ptr, st));
    FUSCHK(upload(m->allocs, &m->d_sh_pairs32, prs, st));
    // the same for the plane form of the rank-local shared dofs (k_shared_stage_planes): one bit per dof and the
    // number of boundary dofs ahead of each 64-dof word (bidx ascends, so the k-th set bit is the k-th term)
    m->d_bnd_mask = nullptr, m->d_bnd_base = nullptr;
    if (m->nb > m->nb_int && L.n_shared_local > 0)
    {
      const int64_t nw = (L.n_shared_local + 63) / 64;
      std::vector<uint64_t> mask(nw, 0);
      std::vector<int32_t> base(nw, 0);
      for (int64_t sidx = 0; sidx < L.n_shared_local; ++sidx)
        if (extra[sidx] >= 0)
          mask[sidx >> 6] |= 1ull << (sidx & 63);
      // boundary dofs of the interface range come after every local one in bidx, so the local ranks start at 0
      for (int64_t w = 1; w < nw; ++w)
        base[w] = base[w - 1] + __builtin_popcountll(mask[w - 1]);
      FUSCHK(upload(m->allocs, &m->d_bnd_mask, mask, st));
      FUSCHK(upload(m->allocs, &m->d_bnd_base, base, st));
    }
  }
  T *d_bsrc, *d_babs;
  FUSCHK(upload(m->allocs, &m->d_bidx, bidx, st));
  FUSCHK(upload(m->allocs, &d_bsrc, bsrc, st));
  FUSCHK(upload(m->allocs, &d_babs, babs, st));
  m->d_bsrc = d_bsrc, m->d_babs = d_babs;
  if (lossy)
  {
    T* d_bsrc2;
    FUSCHK(upload(m->allocs, &d_bsrc2, bsrc2, st));
    m->d_bsrc2 = d_bsrc2;
  }
  HIPCHK(hipMemsetAsync(m->u_, 0, n * sizeof(T), st));
  HIPCHK(hipMemsetAsync(m->v_, 0, n * sizeof(T), st));
  HIPCHK(hipMemsetAsync(m->vn, 0, n * sizeof(T), st));
  HIPCHK(hipStreamSynchronize(st));
  m->setup_done = true;
  return FUS_OK;
}

struct StageScalars
{
  double gval, dgval, adt, bdt;
  double b0dt, pdt;    // dt b_0 and dt a_i (lean RK4 stage kinds: the stage's input is u0 + pdt V_{i-1})
  double tn;  // the stage's time t + c_i dt as the scalars above saw it (in T)
};

// source scalar g(t_n) (Linear.hpp:185-192) and the stage's axpy factors (:282-294), in T
template <typename T>
static StageScalars stage_scalars(const fus_model* m, int i, double t_, double dt_)
{
  const T t = (T)t_, dt = (T)dt_;
  // Runge-Kutta tables (_linear.py:286-311): forward Euler, Ralston 2nd / 3rd order, classical RK4
  // (Linear.hpp:263-265); a_runge carries one trailing 0 for "the stage after the last"
  T a_runge[5] = {0.0, 0.5, 0.5, 1.0, 0.0};
  T b_runge[4] = {(T)(1.0 / 6.0), (T)(1.0 / 3.0), (T)(1.0 / 3.0), (T)(1.0 / 6.0)};
  T c_runge[4] = {0.0, 0.5, 0.5, 1.0};
  if (m->rk_order == 1)
  {
    a_runge[0] = 0, a_runge[1] = 0, b_runge[0] = 1, c_runge[0] = 0;
  }
  else if (m->rk_order == 2)
  {
    a_runge[0] = 0, a_runge[1] = (T)(2.0 / 3.0), a_runge[2] = 0;
    b_runge[0] = (T)(1.0 / 4.0), b_runge[1] = (T)(3.0 / 4.0);
    c_runge[0] = 0, c_runge[1] = (T)(2.0 / 3.0);
  }
  else if (m->rk_order == 3)
  {
    a_runge[0] = 0, a_runge[1] = (T)(1.0 / 2.0), a_runge[2] = (T)(3.0 / 4.0), a_runge[3] = 0;
    b_runge[0] = (T)(2.0 / 9.0), b_runge[1] = (T)(1.0 / 3.0), b_runge[2] = (T)(4.0 / 9.0);
    c_runge[0] = 0, c_runge[1] = (T)(1.0 / 2.0), c_runge[2] = (T)(3.0 / 4.0);
  }
  const T freq = (T)m->freq, p0 = (T)m->amp, s0 = (T)m->speed;
  const T w0 = (T)(2 * M_PI * m->freq);
  const T period = (T)(1.0 / m->freq), window_length = (T)4.0;
  const T tn = t + c_runge[i] * dt;
  T window, dwindow;
  if (tn < period * window_length)
  {
    window = (T)(0.5 * (1.0 - std::cos((double)(freq * (T)M_PI * tn / window_length))));
    dwindow = (T)(0.5 * M_PI) * freq / window_length
              * (T)std::sin((double)(freq * (T)M_PI * tn / window_length));
  }
  else
    window = 1.0, dwindow = 0.0;
  StageScalars sc;
  if (m->kind == FUS_LOSSY || m->kind == FUS_WESTERVELT)
  {
    // heterogeneous-domain scaling, live in Lossy.hpp:216-220 (factor 2) and its derivative dg
    const T two = m->forms == 0 ? (T)2.0 : (T)1.0;  // "heterogenous domain" doubling, Lossy.hpp:216-220
    sc.gval = (double)(window * two * p0 * w0 / s0 * (T)std::cos((double)(w0 * tn)));
    sc.dgval = (double)(dwindow * two * p0 * w0 / s0 * (T)std::cos((double)(w0 * tn))
                        - window * two * p0 * w0 * w0 / s0 * (T)std::sin((double)(w0 * tn)));
  }
  else
  {
    sc.gval = (double)(window * p0 * w0 / s0 * (T)std::cos((double)(w0 * tn)));  // Linear.hpp:192
    sc.dgval = 0.0;
  }
  sc.adt = (double)(dt * a_runge[i + 1]);
  sc.bdt = (double)(dt * b_runge[i]);
  sc.b0dt = (double)(dt * b_runge[0]);
  sc.pdt = (double)(dt * a_runge[i]);
  sc.tn = (double)tn;
  return sc;
}

// Which fused-update variant stage i uses: 0 = first stage (reads u0, v0 only), 1 = middle stage,
// 3 = last stage of RK4 (writes the new u0, v0 directly).  The lower-order schemes end on a middle
// stage and swap (u_, v_) with (u0, v0) afterwards.
static int stage_kind(const fus_model* m, int i)
{
  if (m->rk_order == 4 && m->lean_rk4)   // classical RK4 without accumulator streams (kernels.hpp, stage kinds 4-7)
    return 4 + i;
  if (i == 0)
    return 0;
  return (m->rk_order == 4 && i == 3) ? 3 : 1;
}

// The stage's velocity buffers as the fused updates name them.  Stage kinds 0, 1, 3: the model's own vn, v_, u_.
// Lean RK4 (kinds 4-7): the three stage velocities V_1, V_2, V_3 rotate through those three buffers (A = vn, B = v_,
// C = u_): `vn` is the velocity this stage starts from (read; stage 0 starts from v0), `v_` the one it leaves for the
// next stage (written; at stage 3: V_2, read) and `u_` is V_1 (read at stages 2 and 3).
template <typename T>
struct StageVel
{
  T *vn, *v_, *u_;
};
template <typename T>
static StageVel<T> stage_vel(const fus_model* m, int i)
{
  T *A = static_cast<T*>(m->vn), *B = static_cast<T*>(m->v_), *C = static_cast<T*>(m->u_);
  if (!(m->rk_order == 4 && m->lean_rk4))
    return {A, B, C};
  switch (i)
  {
  case 0: return {A, A, C};   // writes V_1 -> A
  case 1: return {A, B, C};   // reads V_1, writes V_2 -> B
  case 2: return {B, C, A};   // reads V_2 and V_1, writes V_3 -> C
  default: return {C, B, A};  // reads V_3, V_2 and V_1
  }
}

template <typename T>
static StageArgs<T> stage_args(fus_model* m, int i, const StageScalars& sc)
{
  StageArgs<T> S;
  const StageVel<T> V = stage_vel<T>(m, i);
  S.minv = static_cast<const T*>(m->minv);
  S.vn = V.vn, S.un = static_cast<T*>(m->un);
  S.u0 = static_cast<T*>(m->u0), S.v0 = static_cast<T*>(m->v0);
  S.u_ = V.u_, S.v_ = V.v_;
  S.adt = (T)sc.adt, S.bdt = (T)sc.bdt, S.gval = (T)sc.gval;
  S.b0dt = (T)sc.b0dt, S.pdt = (T)sc.pdt, S.third = T(1) / T(3);
  S.blk_bnd_off = m->d_blk_bnd_off, S.bnd_idx = m->d_bidx;
  S.bnd_src = static_cast<const T*>(m->d_bsrc), S.bnd_abs = static_cast<const T*>(m->d_babs);
  S.x2 = nullptr, S.coef2 = static_cast<const T*>(m->coef2);
  S.bnd_src2 = static_cast<const T*>(m->d_bsrc2), S.dgval = (T)sc.dgval;
  S.m0 = static_cast<const T*>(m->m), S.mn1 = static_cast<const T*>(m->mn1);
  return S;
}

// Stage i, first half: the block kernel applies K(-1/rho) to u_stage and, for every block-interior
// dof, finishes the stage right away (boundary terms, kv = b/m, axpys); shared dofs leave as partial
// sums, are reduced into b's shared range, and the interface entries are packed for the exchange.
template <typename T, int P>
static int stage_begin(fus_model* m, int i, double t, double dt)
{
  fus_op* op = m->op;
  const T* ustage = static_cast<const T*>(i == 0 ? m->u0 : m->un);  // a_0 = 0: un == u0
  T* b = static_cast<T*>(m->b);
  StageArgs<T> S = stage_args<T>(m, i, stage_scalars<T>(m, i, t, dt));
  const T* vstage = i == 0 ? static_cast<const T*>(m->v0) : S.vn;   // the velocity this stage starts from
  S.x2 = vstage;   // lossy: second operator input v_n
  const T* G = static_cast<const T*>(op->d_G);
  const T* coef = static_cast<const T*>(m->coef);
  const int kind = stage_kind(m, i);
  auto launch = [&](int b0, int nb) -> int
  {
#define FUS_STAGE_CASE(K, NFV)                                                                       \
  case K:                                                                                          \
    return launch_block_op<T, P, OP_STIFFNESS, K, NFV>(op, G, coef, ustage, b, S, b0, nb);
    if (m->kind == FUS_LOSSY || m->kind == FUS_WESTERVELT)
    {
      switch (kind)
      {
        FUS_STAGE_CASE(0, 2) FUS_STAGE_CASE(3, 2) FUS_STAGE_CASE(4, 2) FUS_STAGE_CASE(5, 2) FUS_STAGE_CASE(6, 2) FUS_STAGE_CASE(7, 2)
      default:
        return launch_block_op<T, P, OP_STIFFNESS, 1, 2>(op, G, coef, ustage, b, S, b0, nb);
      }
    }
    switch (kind)
    {
      FUS_STAGE_CASE(0, 1) FUS_STAGE_CASE(3, 1) FUS_STAGE_CASE(4, 1) FUS_STAGE_CASE(5, 1) FUS_STAGE_CASE(6, 1) FUS_STAGE_CASE(7, 1)
    default:
      return launch_block_op<T, P, OP_STIFFNESS, 1>(op, G, coef, ustage, b, S, b0, nb);
    }
#undef FUS_STAGE_CASE
  };
  // Multi-rank, option "overlap_blocks": the blocks that touch interface dofs (first in the layout)
  // run ahead, their partials are reduced, packed and handed to the exchange, and the remaining
  // blocks (the bulk of the stage) run while the planes are in flight.  Off by default: the exchange
  // already overlaps k_shared_stage, and on one GPU with the exchange looped back the extra small
  // launch costs more (2.47 vs 2.41 ms per step at 64^3 p=4) than it hides.  A block's epilogue only writes that block's interior dofs,
  // which no other block and none of the small kernels below reads.
  const int nb_if = op->L.nblocks_if;
  const bool split = m->ctx->overlap_blocks && !op->neigh.empty() && nb_if > 0 && nb_if < op->L.nblocks;
  {
    ProfScope ps(m->ctx, split ? "stiffness_if" : "stiffness");
    FUSCHK(launch(0, split ? nb_if : -1));
  }
  // boundary terms of the shared boundary dofs become one more partial each
  hipStream_t st = m->ctx->stream;
  const int64_t nbs = m->nb - m->nb_int;
  const StageScalars sc_now = stage_scalars<T>(m, i, t, dt);
  if (nbs > 0 && !(m->bnd_valid && op->bnd_owner == m && m->bnd_tn == sc_now.tn))
  {
    ProfScope ps(m->ctx, "boundary");
    hipLaunchKernelGGL((k_boundary_partial<T>), dim3(nblk(nbs)), dim3(256), 0, st, nbs,
                       m->d_bidx + m->nb_int, static_cast<const T*>(m->d_bsrc) + m->nb_int,
                       static_cast<const T*>(m->d_babs) + m->nb_int, S.gval,
                       m->d_bsrc2 ? static_cast<const T*>(m->d_bsrc2) + m->nb_int : nullptr, S.dgval,
                       vstage, static_cast<T*>(op->d_partial) + op->L.n_partial);
  }
  // interface dofs (held by other ranks too): this rank's partials are summed into b and into the
  // send buffer by one kernel; the event hands the buffer to the exchange
  if (!op->neigh.empty())
  {
    ProfScope ps(m->ctx, "halo");
    hipLaunchKernelGGL((k_if_reduce_pack<T>), dim3(nblk(op->n_halo)), dim3(256), 0, st, op->n_halo,
                       op->d_pack_idx, op->L.n_int_pad, m->d_sh_ptr32, m->d_sh_pairs32,
                       static_cast<const T*>(op->d_partial), b, static_cast<T*>(op->d_sendbuf));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(m->ctx->ev_packed, st));
  }
  HIPCHK(hipGetLastError());
  if (split)
  {
    ProfScope ps(m->ctx, "stiffness");
    FUSCHK(launch(nb_if, op->L.nblocks - nb_if));
  }
  return FUS_OK;
}

// Stage i, second half, shared dofs only.  Rank-local shared dofs: fixed-order sum of the block
// partials fused with the stage update (k_shared_stage).  Interface dofs: ordered sum of the
// sharers' totals fused with the same update (k_if_unpack_stage).
template <typename T>
static int stage_end(fus_model* m, int i, double t, double dt)
{
  fus_op* op = m->op;
  fus_ctx* c = m->ctx;
  hipStream_t st = c->stream;
  const StageScalars sc = stage_scalars<T>(m, i, t, dt);
  const T adt = (T)sc.adt, bdt = (T)sc.bdt;
  const StageVel<T> V = stage_vel<T>(m, i);
  T *u0 = static_cast<T*>(m->u0), *v0 = static_cast<T*>(m->v0), *u_ = V.u_, *v_ = V.v_,
    *un = static_cast<T*>(m->un), *vn = V.vn, *b = static_cast<T*>(m->b);
  const T* minv = static_cast<const T*>(m->minv);
  const T* partial = static_cast<const T*>(op->d_partial);
  const int64_t off = op->L.n_int_pad, nloc = op->L.n_shared_local;
  // Westervelt: m0 and mn1 on the shared range (nullptr otherwise)
  const T* m0p = m->mn1 ? static_cast<const T*>(m->m) + off : nullptr;
  const T* mn1p = m->mn1 ? static_cast<const T*>(m->mn1) + off : nullptr;
  // RK4: this stage's kernels also leave the next stage's boundary terms of the shared boundary dofs
  // in their pseudo partial slots (next stage of this step, or stage 0 of the next step at t + dt,
  // advanced the way the step loops do: in T for fus_model_rk4, whose t is a T)
  BndNext<T> B{};
  const int64_t nbs = m->nb - m->nb_int;
  m->bnd_valid = false;
  if (m->rk_order == 4 && nbs > 0 && op->L.n_partial + nbs < INT32_MAX)
  {
    const StageScalars scn = i < 3 ? stage_scalars<T>(m, i + 1, t, dt)
                                   : stage_scalars<T>(m, 0, (double)((T)t + (T)dt), dt);
    B.enabled = 1, B.npairs = (int32_t)op->L.n_partial;
    B.srcw = static_cast<const T*>(m->d_bsrc) + m->nb_int;
    B.absw = static_cast<const T*>(m->d_babs) + m->nb_int;
    B.src2w = m->d_bsrc2 ? static_cast<const T*>(m->d_bsrc2) + m->nb_int : nullptr;
    B.gnext = (T)scn.gval, B.dgnext = (T)scn.dgval;
    B.partial = static_cast<T*>(op->d_partial);
    m->bnd_valid = true, m->bnd_tn = scn.tn;
    op->bnd_owner = m;
  }
  const LeanRK<T> R{(T)sc.b0dt, (T)sc.pdt, T(1) / T(3)};
  const int kind = stage_kind(m, i);
  if (nloc > 0 && c->planes && (int)op->L.plane_cnt.size() <= (c->planes == 1 ? FUS_MAX_PLANES : c->planes))
  {
    ProfScope ps(c, "stage");
    const dim3 grid(nblk(nloc)), blk(256);
    PartialPlanes PL{};
    PL.bnd0 = (int32_t)op->L.n_partial;
    for (size_t j = 0; j < op->L.plane_cnt.size(); ++j)
      PL.cnt[j] = (int32_t)op->L.plane_cnt[j], PL.off[j] = (int32_t)op->L.plane_off[j];
#define FUS_PLANES_CASE(K)                                                                         \
  case K:                                                                                          \
    hipLaunchKernelGGL((k_shared_stage_planes<T, K>), grid, blk, 0, st, nloc, PL, m->d_bnd_mask, m->d_bnd_base,      \
                       partial, minv + off, vn + off, un + off, u0 + off, v0 + off, u_ + off, v_ + off, adt, bdt,    \
                       m0p, mn1p, B, R);                                                           \
    break;
    switch (kind)
    {
      FUS_PLANES_CASE(0) FUS_PLANES_CASE(3) FUS_PLANES_CASE(4) FUS_PLANES_CASE(5) FUS_PLANES_CASE(6) FUS_PLANES_CASE(7)
    default:
      hipLaunchKernelGGL((k_shared_stage_planes<T, 1>), grid, blk, 0, st, nloc, PL, m->d_bnd_mask, m->d_bnd_base,
                         partial, minv + off, vn + off, un + off, u0 + off, v0 + off, u_ + off, v_ + off, adt, bdt,
                         m0p, mn1p, B, R);
    }
#undef FUS_PLANES_CASE
  }
  else if (nloc > 0)
  {
    ProfScope ps(c, "stage");
    const dim3 grid(nblk(nloc)), blk(256);
#define FUS_SHARED_CASE(K)                                                                         \
  case K:                                                                                          \
    hipLaunchKernelGGL((k_shared_stage<T, K>), grid, blk, 0, st, nloc, m->d_sh_ptr32, m->d_sh_pairs32, partial,      \
                       minv + off, vn + off, un + off, u0 + off, v0 + off, u_ + off, v_ + off, adt, bdt, m0p, mn1p, B, \
                       R);                                                                         \
    break;
    switch (kind)
    {
      FUS_SHARED_CASE(0) FUS_SHARED_CASE(3) FUS_SHARED_CASE(4) FUS_SHARED_CASE(5) FUS_SHARED_CASE(6) FUS_SHARED_CASE(7)
    default:
      hipLaunchKernelGGL((k_shared_stage<T, 1>), grid, blk, 0, st, nloc, m->d_sh_ptr32, m->d_sh_pairs32, partial,
                         minv + off, vn + off, un + off, u0 + off, v0 + off, u_ + off, v_ + off, adt, bdt, m0p, mn1p, B,
                         R);
    }
#undef FUS_SHARED_CASE
  }
  if (!op->neigh.empty())
  {
    // (the rank-local shared dofs above ran while the interface planes were in flight)
    // ordered sum of the sharers' totals fused with the stage update of the interface dofs
    if (!c->local_group)
      HIPCHK(hipStreamWaitEvent(st, c->ev_recv, 0));
    ProfScope ps(c, "stage");
    const dim3 grid(nblk(op->n_uidx)), blk(256);
    const T* recv = static_cast<const T*>(op->d_recvbuf);
    const T* m0f = m->mn1 ? static_cast<const T*>(m->m) : nullptr;
    const T* mn1f = m->mn1 ? static_cast<const T*>(m->mn1) : nullptr;
#define FUS_IF_CASE(K)                                                                             \
  case K:                                                                                          \
    hipLaunchKernelGGL((k_if_unpack_stage<T, K>), grid, blk, 0, st, op->n_uidx, op->d_uidx, op->d_uptr, op->d_usrc,  \
                       recv, b, minv, vn, un, u0, v0, u_, v_, adt, bdt, m0f, mn1f, op->L.n_int_pad, m->d_sh_ptr32,    \
                       m->d_sh_pairs32, B, R);                                                     \
    break;
    switch (kind)
    {
      FUS_IF_CASE(0) FUS_IF_CASE(3) FUS_IF_CASE(4) FUS_IF_CASE(5) FUS_IF_CASE(6) FUS_IF_CASE(7)
    default:
      hipLaunchKernelGGL((k_if_unpack_stage<T, 1>), grid, blk, 0, st, op->n_uidx, op->d_uidx, op->d_uptr, op->d_usrc,
                         recv, b, minv, vn, un, u0, v0, u_, v_, adt, bdt, m0f, mn1f, op->L.n_int_pad, m->d_sh_ptr32,
                         m->d_sh_pairs32, B, R);
    }
#undef FUS_IF_CASE
  }
  HIPCHK(hipGetLastError());
  return FUS_OK;
}

// One classical RK4 step (Linear.hpp:273-295), state in (u0, v0) on entry and exit; RCCL (or
// single-rank) transport.
template <typename T, int P>
static int model_step(fus_model* m, double t, double dt)
{
  for (int i = 0; i < m->rk_order; ++i)
  {
    FUSCHK((stage_begin<T, P>(m, i, t, dt)));
    if (!m->op->neigh.empty())
    {
      ProfScope ps(m->ctx, "halo");
      FUSCHK(halo_exchange_rccl(m->op));
    }
    FUSCHK(stage_end<T>(m, i, t, dt));
  }
  if (m->rk_order != 4)  // the accumulated solution becomes the next step's start state
    std::swap(m->u_, m->u0), std::swap(m->v_, m->v0);
  return FUS_OK;
}

template <typename T>
static int model_getset(fus_model* m, int which, void* host_or_dev, int space, bool set)
{
  fus_op* op = m->op;
  hipStream_t st = m->ctx->stream;
  T* vec = static_cast<T*>(which == FUS_U ? m->u0 : m->v0);
  T* tc = static_cast<T*>(op->d_tmp_c);
  if (set)
  {
    const T* src = static_cast<const T*>(host_or_dev);
    if (space == FUS_HOST)
    {
      HIPCHK(hipMemcpyAsync(tc, host_or_dev, op->ndofs * sizeof(T), hipMemcpyHostToDevice, st));
      src = tc;
    }
    hipLaunchKernelGGL((k_to_internal<T>), dim3(nblk(op->ndofs)), dim3(256), 0, st, op->ndofs,
                       op->d_dof_perm, src, vec);
  }
  else
  {
    T* dst = space == FUS_HOST ? tc : static_cast<T*>(host_or_dev);
    hipLaunchKernelGGL((k_from_internal<T, 0>), dim3(nblk(op->ndofs)), dim3(256), 0, st, op->ndofs,
                       op->d_dof_perm, vec, dst);
    if (space == FUS_HOST)
      HIPCHK(hipMemcpyAsync(host_or_dev, tc, op->ndofs * sizeof(T), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  return FUS_OK;
}

// -------------------------------------------------------------------------------------------------
// dispatch over (dtype, P): the degree-dependent entry points live in the per-degree units
// -------------------------------------------------------------------------------------------------
struct DegreeImpl
{
  int (*op_setup)(fus_op*);
  int (*op_apply)(fus_op*, int, const void*, const void*, void*, int);
  int (*op_get_geometry)(fus_op*, void*, void*);
  int (*model_setup)(fus_model*, const void*, const void*, const void*, const void*, int64_t,
                     const int32_t*, const int32_t*, const int32_t*);
  int (*model_step)(fus_model*, double, double);
  int (*stage_begin)(fus_model*, int, double, double);
};
#define FUS_CAT_(a, b) a##b
#define FUS_CAT(a, b) FUS_CAT_(a, b)

#ifdef FUS_TU_DEGREE
template <typename T, int P>
static int op_apply_kind(fus_op* op, int kind, const void* x, const void* cf, void* y, int space)
{
  return kind == OP_STIFFNESS ? op_apply<T, P, OP_STIFFNESS>(op, x, cf, y, space)
                              : op_apply<T, P, OP_MASS>(op, x, cf, y, space);
}
template <typename T, int P>
static DegreeImpl make_degree_impl()
{
  return {&op_setup_device<T, P>, &op_apply_kind<T, P>, &op_get_geometry<T, P>,
          &model_setup<T, P>,     &model_step<T, P>,    &stage_begin<T, P>};
}
// one unit per (degree, scalar type): -DFUS_TU_DEGREE=k -DFUS_TU_DTYPE=64|32
#if FUS_TU_DTYPE == 64
FUS_HIDDEN const DegreeImpl* FUS_CAT(FUS_CAT(fus_degree_impl_, FUS_TU_DEGREE), _f64)()
{
  static const DegreeImpl d = make_degree_impl<double, FUS_TU_DEGREE>();
  return &d;
}
#else
FUS_HIDDEN const DegreeImpl* FUS_CAT(FUS_CAT(fus_degree_impl_, FUS_TU_DEGREE), _f32)()
{
  static const DegreeImpl d = make_degree_impl<float, FUS_TU_DEGREE>();
  return &d;
}
#endif
#else  // main unit: everything from here to the end of the file
#define FUS_DECL_DEGREE(k)                                                                         \
  FUS_HIDDEN const DegreeImpl* FUS_CAT(FUS_CAT(fus_degree_impl_, k), _f64)();                      \
  FUS_HIDDEN const DegreeImpl* FUS_CAT(FUS_CAT(fus_degree_impl_, k), _f32)();
#define FUS_CASE_DEGREE(k)                                                                         \
  case k:                                                                                          \
    return dtype == FUS_F64 ? FUS_CAT(FUS_CAT(fus_degree_impl_, k), _f64)() : FUS_CAT(FUS_CAT(fus_degree_impl_, k), _f32)();
#ifdef FUS_DEV_BUILD  // developer iteration build: ONE degree (never shipped; build.py --dev)
#ifndef FUS_DEV_DEGREE
#define FUS_DEV_DEGREE 4
#endif
FUS_DECL_DEGREE(FUS_DEV_DEGREE)
static const DegreeImpl* degree_impl(int dtype, int P)
{
  switch (P)
  {
    FUS_CASE_DEGREE(FUS_DEV_DEGREE)
  default: return nullptr;
  }
}
#else
FUS_DECL_DEGREE(2) FUS_DECL_DEGREE(3) FUS_DECL_DEGREE(4) FUS_DECL_DEGREE(5) FUS_DECL_DEGREE(6) FUS_DECL_DEGREE(7)
FUS_DECL_DEGREE(8) FUS_DECL_DEGREE(9) FUS_DECL_DEGREE(10)
static const DegreeImpl* degree_impl(int dtype, int P)
{
  switch (P)
  {
    FUS_CASE_DEGREE(2) FUS_CASE_DEGREE(3) FUS_CASE_DEGREE(4) FUS_CASE_DEGREE(5) FUS_CASE_DEGREE(6) FUS_CASE_DEGREE(7)
    FUS_CASE_DEGREE(8) FUS_CASE_DEGREE(9) FUS_CASE_DEGREE(10)
  default: return nullptr;
  }
}
#endif
#define FUS_DEGREE(d, dtype_, P_)                                                                  \
  const DegreeImpl* d = degree_impl(dtype_, P_);                                                   \
  if (!d)                                                                                          \
    return fail(FUS_ERR_ARG, "unsupported polynomial degree (2..10)");

static int d_op_setup(fus_op* op)
{
  FUS_DEGREE(d, op->dtype, op->P);
  return d->op_setup(op);
}
static int d_op_apply(fus_op* op, int kind, const void* x, const void* cf, void* y, int space)
{
  FUS_DEGREE(d, op->dtype, op->P);
  return d->op_apply(op, kind, x, cf, y, space);
}
static int d_op_get_geometry(fus_op* op, void* G, void* dJ)
{
  FUS_DEGREE(d, op->dtype, op->P);
  return d->op_get_geometry(op, G, dJ);
}
static int d_model_setup(fus_model* m, const void* c0, const void* rho0, const void* delta0,
                         const void* beta0, int64_t nf, const int32_t* fc, const int32_t* fl,
                         const int32_t* ft)
{
  FUS_DEGREE(d, m->op->dtype, m->op->P);
  return d->model_setup(m, c0, rho0, delta0, beta0, nf, fc, fl, ft);
}
// One RK step.  Option "graph" (one rank, no per-kernel events): the step's launches are captured
// into a graph and replayed as ONE submission -- for launch-bound sizes (BASELINE config 1: eight
// kernels of a few microseconds).  The stage scalars (source value at the stage times) are kernel
// arguments, so every step is re-captured and the executable graph updated in place
// (hipGraphExecUpdate: same topology, new arguments); the first step after init / set runs directly
// (it has one more launch, k_boundary_partial, and sets the kernels' attributes).
static int d_model_step(fus_model* m, double t, double dt)
{
  FUS_DEGREE(d, m->op->dtype, m->op->P);
  fus_ctx* c = m->ctx;
  const bool graphed = c->graph && c->nranks <= 1 && !c->prof && m->op->neigh.empty() && m->gsteps > 0;
  ++m->gsteps;
  if (!graphed)
    return d->model_step(m, t, dt);
  HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  const int r = d->model_step(m, t, dt);
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(c->stream, &g);
  if (r != FUS_OK || e != hipSuccess || !g)
  {
    if (g)
      (void)hipGraphDestroy(g);
    return r != FUS_OK ? r : fail(FUS_ERR_HIP, std::string("graph capture of the RK step failed: ") + hipGetErrorString(e));
  }
  if (m->gexec)
  {
    hipGraphNode_t bad = nullptr;
    hipGraphExecUpdateResult res;
    if (hipGraphExecUpdate(m->gexec, g, &bad, &res) != hipSuccess)
    {
      (void)hipGetLastError();
      (void)hipGraphExecDestroy(m->gexec);
      m->gexec = nullptr;
    }
  }
  if (!m->gexec)
  {
    const hipError_t ei = hipGraphInstantiate(&m->gexec, g, nullptr, nullptr, 0);
    if (ei != hipSuccess)
    {
      (void)hipGraphDestroy(g);
      m->gexec = nullptr;
      return fail(FUS_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
    }
  }
  const hipError_t el = hipGraphLaunch(m->gexec, c->stream);
  (void)hipGraphDestroy(g);
  if (el != hipSuccess)
    return fail(FUS_ERR_HIP, std::string("hipGraphLaunch: ") + hipGetErrorString(el));
  return FUS_OK;
}
static int d_stage_begin(fus_model* m, int i, double t, double dt)
{
  FUS_DEGREE(d, m->op->dtype, m->op->P);
  return d->stage_begin(m, i, t, dt);
}
static int d_stage_end(fus_model* m, int i, double t, double dt)
{
  return m->op->dtype == FUS_F64 ? stage_end<double>(m, i, t, dt) : stage_end<float>(m, i, t, dt);
}
static int d_setup_finish(fus_model* m)
{
  return m->op->dtype == FUS_F64 ? model_setup_finish<double>(m) : model_setup_finish<float>(m);
}
static int d_halo_sum(fus_op* op, void* v)
{
  return op->dtype == FUS_F64 ? halo_sum<double>(op, static_cast<double*>(v))
                              : halo_sum<float>(op, static_cast<float*>(v));
}
static int d_halo_pack(fus_op* op, const void* v)
{
  return op->dtype == FUS_F64 ? halo_pack<double>(op, static_cast<const double*>(v))
                              : halo_pack<float>(op, static_cast<const float*>(v));
}
static int d_halo_unpack(fus_op* op, void* v)
{
  return op->dtype == FUS_F64 ? halo_unpack<double>(op, static_cast<double*>(v))
                              : halo_unpack<float>(op, static_cast<float*>(v));
}

// (Re)build the block layout and the device-side operator data.  force_shared marks dofs held by
// other ranks too: they must never be finished inside a block's fused epilogue.
static int op_build(fus_op* op, const uint8_t* force_shared)
{
  fus_ctx* c = op->ctx;
  for (void* q : op->allocs)
    (void)hipFree(q);
  op->allocs.clear();
  // centroids for the block partitioner
  std::vector<double> cen((size_t)op->ncells * 3, 0.0);
  const int nv = op->geom_nv;
  for (int64_t cidx = 0; cidx < op->ncells; ++cidx)
    for (int v = 0; v < nv; ++v)
      for (int j = 0; j < 3; ++j)
      {
        const size_t k = 3 * (size_t)op->h_geom_dm[cidx * nv + v] + j;
        cen[3 * cidx + j] +=
            (1.0 / nv) * (op->dtype == FUS_F64 ? reinterpret_cast<const double*>(op->h_geom_x.data())[k]
                                          : (double)reinterpret_cast<const float*>(op->h_geom_x.data())[k]);
      }
  // will the per-cell (affine) geometry path be taken?  (same test as k_geometry_affine, on the host
  // copy: every vertex of every cell on the parallelepiped spanned by vertices 0, 1, 2, 4)
  bool affine_mesh = c->geometry == 0 && op->geom_order == 1 && op->tdim == 3;
  bool ortho_mesh = affine_mesh;
  for (int64_t cidx = 0; cidx < op->ncells && affine_mesh; ++cidx)
  {
    double cd[8][3], h2 = 0, e2 = 0;
    for (int v = 0; v < 8; ++v)
      for (int j = 0; j < 3; ++j)
      {
        const size_t k = 3 * (size_t)op->h_geom_dm[cidx * 8 + v] + j;
        cd[v][j] = op->dtype == FUS_F64 ? reinterpret_cast<const double*>(op->h_geom_x.data())[k]
                                        : (double)reinterpret_cast<const float*>(op->h_geom_x.data())[k];
      }
    for (int v = 0; v < 8; ++v)
      for (int i = 0; i < 3; ++i)
      {
        const double e0 = cd[1][i] - cd[0][i], e1 = cd[2][i] - cd[0][i], e4 = cd[4][i] - cd[0][i];
        const double d = cd[v][i] - (cd[0][i] + (v & 1) * e0 + ((v >> 1) & 1) * e1 + (v >> 2) * e4);
        e2 = std::max(e2, d * d);
        if (v == 0)
          h2 += e0 * e0 + e1 * e1 + e4 * e4;
      }
    const double tol = op->dtype == FUS_F64 ? 1e-12 : 1e-6;
    if (std::sqrt(e2 / h2) > 0.5 * tol)
      affine_mesh = false;
    // mutually orthogonal edges: J^T J, and with it G = |det J| (J^T J)^-1, is diagonal
    double ed[3][3];
    for (int i = 0; i < 3; ++i)
      ed[0][i] = cd[1][i] - cd[0][i], ed[1][i] = cd[2][i] - cd[0][i], ed[2][i] = cd[4][i] - cd[0][i];
    for (int u = 0; u < 3; ++u)
      for (int v = u + 1; v < 3; ++v)
      {
        double dot = 0, nu = 0, nv2 = 0;
        for (int i = 0; i < 3; ++i)
          dot += ed[u][i] * ed[v][i], nu += ed[u][i] * ed[u][i], nv2 += ed[v][i] * ed[v][i];
        if (std::fabs(dot) > (op->dtype == FUS_F64 ? 1e-14 : 2e-7) * std::sqrt(nu * nv2))
          ortho_mesh = false;
      }
  }
  op->ortho_mesh = affine_mesh && ortho_mesh;
  // diagonal-metric form of the affine kernel: scalar kernels of the degrees <= 7 (no packed / matrix-core variant)
  const bool diag_mesh = op->ortho_mesh && c->diag_metric && op->P <= 7 && !(c->mfma == 1) && !(c->pack32 == 1);
  // auto block size: about 2-3 thousand local dofs per block when G is streamed (128 / 64 / 32
  // elements at P = 2 / 3 / >= 4; larger P shrink further to fit LDS), half of that on the affine
  // path (profiles/r01_block_sweep.txt)
  // quadrilaterals: N^2 nodes per element, so many more elements make a block of that size
  if (op->P >= 8)
  {
    // degrees 8-10 (two waves per element): 4-element blocks, an even number of waves
    const bool aff = affine_mesh;
    const bool tri8 = !affine_mesh && c->geometry != 1 && op->geom_order == 1 && op->tdim == 3;   // else: per-point factors streamed
    const int gcs8 = aff ? 7 : (tri8 ? 21 : 0);
    int waves8 = c->waves > 0 ? std::min(4, (c->waves + 1) & ~1) : 4;
    // (fp64 distorted cells at degree 8: 8-element blocks, +8.5 % over 4 -- fewer shared dofs; profiles/r03_experiments.md 7)
    const int be_hi = op->tdim == 2 ? 32 : ((op->P == 8 && op->dtype == FUS_F64 && tri8) ? 8 : 4);
    for (int be = c->block_elems > 0 ? c->block_elems : be_hi;; be = (be + 1) / 2)
    {
      std::string err = build_layout(op->L, op->P, op->ncells, op->ndofs, op->h_dofmap.data(), cen.data(), be, waves8,
                                     force_shared, op->tdim);
      const bool too_big = err.empty() ? op->L.lds_bytes(op->ts, op->nfields, gcs8) + 64 > 160 * 1024
                                       : err.find("65535") != std::string::npos;
      if (too_big && be > 1)
        continue;
      if (!err.empty())
        return fail(FUS_ERR_ARG, "layout: " + err);
      if (too_big)
        return fail(FUS_ERR_LIMIT, "a one-element block does not fit the LDS budget at this degree");
      break;
    }
    int r8 = d_op_setup(op);
    if (r8 != FUS_OK)
    {
      for (void* q : op->allocs)
        (void)hipFree(q);
      op->allocs.clear();
    }
    return r8;
  }
  static const int be_quad[8] = {0, 0, 512, 256, 256, 128, 128, 64};
  // fp32 halves the LDS per dof: there the best sizes keep two or three blocks per CU resident
  // (p=4: 48, p=5/6: 24, p=7: 16 -- +8 / +15 / +45 / +42 % over 32 elements, profiles/r01_block_sweep.txt)
  static const int be_hex_f32[8] = {0, 0, 128, 64, 48, 24, 24, 16};
  const int be_hex = op->dtype == FUS_F32 ? be_hex_f32[op->P] : (op->P == 2 ? 128 : (op->P == 3 ? 64 : 32));
  const int be_stream = op->tdim == 2 ? be_quad[op->P] : be_hex;
  // fp64, p >= 5, streamed geometry, LDS-atomic mode: the block kernel keeps ONE geometry register
  // set there (kernels.hpp, FUS_PF1) and fits two waves per SIMD, which pays only if two blocks per
  // CU are resident: blocks of at most 80 KB of LDS (20 / 12 / 8 elements at p = 5 / 6 / 7 with one
  // operator input, fewer with two).  +50 / +30 / +35 % over one 32->16-element block per CU.
  // first-order hexahedra that are not all affine: J and G recomputed per point from each cell's
  // trilinear map (GEOM_TRILINEAR) unless the streamed factors are asked for
  const bool trilinear_mesh = !affine_mesh && c->geometry != 1 && op->geom_order == 1 && op->tdim == 3;
  const bool two_per_cu = op->dtype == FUS_F64 && op->tdim == 3 && op->P >= 5 && !affine_mesh && !trilinear_mesh
                          && c->block_elems <= 0;
  static const int be_two[8] = {0, 0, 0, 0, 0, 20, 12, 8};
  // per-cell geometry paths (affine, trilinear): about half the streamed size; at p >= 5 blocks of 8
  // elements, small enough for four / three / two blocks per CU at the register budgets of those
  // kernels (p=5: +3-5 % over 12, p=6: +19-20 %; profiles/r01_block_sweep.txt)
  // (fp32, round 3: the fp64 block accumulator costs 4 more bytes of LDS per local dof; re-swept at 64^3,
  // profiles/r03_experiments.md section 4: affine 8 / 8 / 4 elements at p = 5 / 6 / 7)
  static const int be_aff_hi[8] = {0, 0, 0, 0, 0, 8, 8, 8};
  const int be_affine = (op->tdim == 3 && op->P >= 5) ? ((op->dtype == FUS_F32 && op->P == 7) ? 4 : be_aff_hi[op->P])
                                                      : be_stream / 2;
  const int gcs = affine_mesh ? 7 : (trilinear_mesh ? 21 : 0);
  // packed fp32 kernels (two elements per wave): degrees 5-7, per-cell geometry, LDS atomics, MFMA variants off
  // "auto" = where it measured faster on MI355X (profiles/r02_experiments.md section 7): degree 5 (+4 %) and degree 6
  // on affine cells (+2 %); slower at degree 6 trilinear (-12 %: 137 registers, three waves per SIMD) and degree 7
  op->pk = op->dtype == FUS_F32 && op->tdim == 3 && op->P >= 5 && op->P <= 7 && (affine_mesh || trilinear_mesh)
           && !c->deterministic && c->mfma != 1
           && (c->pack32 == 1 || (c->pack32 < 0 && !diag_mesh && (op->P == 5 || (op->P == 6 && affine_mesh))));
  // fp32 halves the LDS per block: the trilinear kernel takes 16 elements at p >= 5 (+9-12 %), the
  // affine one at p = 5 only (+5 %; 16 is 2-9 % slower at p = 6, 7)
  // (round 3, fp64 accumulator: 16 / 8 / 8 elements at p = 5 / 6 / 7)
  const bool hi32 = op->dtype == FUS_F32 && op->tdim == 3 && op->P >= 5;
  const int be_tri = hi32 ? (op->P == 5 ? 16 : 8) : be_affine;
  const int be0 = c->block_elems > 0
                      ? c->block_elems
                      : ((trilinear_mesh || affine_mesh) && op->P == 4 && op->nfields == 1 && c->waves <= 0) ? 32
                      : (two_per_cu ? be_two[op->P]
                                    : (trilinear_mesh ? be_tri : (affine_mesh ? be_affine : be_stream)));
  const size_t lds_cap = two_per_cu ? 80 * 1024 : 160 * 1024;
  int waves = c->waves > 0 ? c->waves : 4;
  if (op->P > 4 && waves > FUS_MID_THREADS / 64)
    waves = FUS_MID_THREADS / 64;  // launch bound of the block kernel for the higher degrees
  // per-cell geometry paths at p = 4 in fp64: 32 elements / 8 waves (two 8-wave blocks per CU) leave fewer
  // shared dofs than 16 / 4 -- the block kernel is 2-6 % slower, the step 1.5-3 % faster
  // (one operator input only: with two, such a block takes 92 KB of LDS and a CU holds one)
  // (fp32: the same 32 elements, +7-8 % over 24; eight waves on affine cells, four on the trilinear kernel)
  // (fp32 trilinear too since the fp64 accumulator: 1.58 -> 1.62e10 with eight waves)
  const bool tri_p4 = (trilinear_mesh || affine_mesh) && op->P == 4 && op->nfields == 1 && c->waves <= 0
                      && c->block_elems <= 0;
  if (tri_p4)
    waves = 8;
  // blocks must fit the LDS budget (the CU's 160 KB, or half of it): shrink the block until they do
  for (int be = be0;; be = two_per_cu && be > 4 ? be - std::max(1, be / 4) : (be + 1) / 2)
  {
    std::string err = build_layout(op->L, op->P, op->ncells, op->ndofs, op->h_dofmap.data(),
                                   cen.data(), be, waves, force_shared, op->tdim, op->pk ? 2 : 1);
    const bool too_big = err.empty() ? op->L.lds_bytes(op->ts, op->nfields, gcs) + 64 > lds_cap
                                     : err.find("65535") != std::string::npos;
    if (too_big && be > 1)
      continue;
    if (!err.empty())
      return fail(FUS_ERR_ARG, "layout: " + err);
    if (too_big)
      return fail(FUS_ERR_LIMIT, "a one-element block does not fit the LDS budget at this degree");
    break;
  }
  int r = d_op_setup(op);
  if (r != FUS_OK)
  {
    for (void* q : op->allocs)
      (void)hipFree(q);
    op->allocs.clear();
  }
  return r;
}

// -------------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------------
extern "C"
{

const char* fus_last_error(void) { return g_err.c_str(); }
int fus_version(void) { return 1; }

int fus_init(int device, fus_ctx** out)
{
  if (!out)
    return fail(FUS_ERR_ARG, "null ctx pointer");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(FUS_ERR_HIP, "no HIP device: libfusmi has no CPU fallback");
  if (device < 0 || device >= ndev)
    return fail(FUS_ERR_ARG, "device index out of range");
  HIPCHK(hipSetDevice(device));
  auto* c = new fus_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    c->num_cus = prop.multiProcessorCount;
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_recv, hipEventDisableTiming));
  *out = c;
  return FUS_OK;
}

int fus_finalize(fus_ctx* c)
{
  if (!c)
    return FUS_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& kv : c->profs)
    for (auto& ev : kv.second.ev)
      (void)hipEventDestroy(ev.first), (void)hipEventDestroy(ev.second);
  (void)hipStreamSynchronize(c->comm_stream);
  if (c->comm)
    g_rccl.CommDestroy(c->comm);
  (void)hipEventDestroy(c->ev_packed), (void)hipEventDestroy(c->ev_recv);
  (void)hipStreamDestroy(c->comm_stream);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return FUS_OK;
}

int fus_synchronize(fus_ctx* c)
{
  HIPCHK(hipStreamSynchronize(c->stream));
  return FUS_OK;
}

int fus_set_option(fus_ctx* c, const char* key, int64_t value)
{
  if (!c || !key)
    return fail(FUS_ERR_ARG, "null argument");
  if (!strcmp(key, "block_elems"))
  {
    if (value < 0 || value > 4096)
      return fail(FUS_ERR_ARG, "block_elems out of range");
    c->block_elems = (int)value;
  }
  else if (!strcmp(key, "waves"))
  {
    if (value < 0 || value > 8)
      return fail(FUS_ERR_ARG, "waves must be 0 (auto) or 1..8");
    c->waves = (int)value;
  }
  else if (!strcmp(key, "deterministic"))
    c->deterministic = value != 0;
  else if (!strcmp(key, "graph"))
    c->graph = value != 0;
  else if (!strcmp(key, "geometry"))
  {
    if (value < 0 || value > 2)
      return fail(FUS_ERR_ARG, "geometry must be 0 (auto), 1 (stream) or 2 (trilinear)");
    c->geometry = (int)value;
  }
  else if (!strcmp(key, "halo_loopback"))
    c->loopback = value != 0;
  else if (!strcmp(key, "overlap_blocks"))
    c->overlap_blocks = value != 0;
  else if (!strcmp(key, "external_transport"))
    c->external_transport = value != 0;
  else if (!strcmp(key, "forms"))
  {
    if (value != 0 && value != 1)
      return fail(FUS_ERR_ARG, "forms must be 0 (C++ benchmark forms) or 1 (Python package forms)");
    c->forms = (int)value;
  }
  else if (!strcmp(key, "lean_rk4"))
    c->lean_rk4 = value != 0;
  else if (!strcmp(key, "pack32"))
  {
    if (value < -1 || value > 1)
      return fail(FUS_ERR_ARG, "pack32 must be -1 (auto), 0 or 1");
    c->pack32 = (int)value;
  }
  else if (!strcmp(key, "profile_sample"))
  {
    if (value < 1 || value > 1000)
      return fail(FUS_ERR_ARG, "profile_sample must be 1..1000");
    c->prof_sample = (int)value;
  }
  else if (!strcmp(key, "planes"))
  {
    if (value < 0 || value > FUS_MAX_PLANES)
      return fail(FUS_ERR_ARG, "planes must be 0 (CSR form), 1 (planes) or 2..16 (planes for meshes with at most so many sharers of a dof)");
    c->planes = (int)value;
  }
  else if (!strcmp(key, "diag_metric"))
    c->diag_metric = value != 0;
  else if (!strcmp(key, "walk"))
  {
    if (value < -1 || value > 8)
      return fail(FUS_ERR_ARG, "walk must be -1 (auto), 0 (one workgroup per block) or 1..8 workgroups per CU");
    c->walk = (int)value;
  }
  else if (!strcmp(key, "mfma"))
  {
    if (value < -1 || value > 1)
      return fail(FUS_ERR_ARG, "mfma must be -1 (auto), 0 or 1");
    c->mfma = (int)value;
  }
  else if (!strcmp(key, "fields"))
  {
    if (value != 1 && value != 2)
      return fail(FUS_ERR_ARG, "fields must be 1 or 2");
    c->fields = (int)value;
  }
  else
    return fail(FUS_ERR_ARG, std::string("unknown option ") + key);
  return FUS_OK;
}

int fus_comm_unique_id(void* id128)
{
  static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
  FUSCHK(rccl_load());
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, 128);
  return FUS_OK;
}

int fus_comm_init(fus_ctx* c, int rank, int nranks, const void* id128)
{
  if (!c || nranks < 1 || rank < 0 || rank >= nranks)
    return fail(FUS_ERR_ARG, "bad rank/nranks");
  c->rank = rank, c->nranks = nranks;
  if (c->external_transport)
  {
    // the caller moves the packed interface values itself (fus_op_halo_buffers): no communicator, the
    // stage / setup halves are driven through fus_model_stage_begin/_end, fus_model_setup_*
    c->local_group = nranks > 1;
    return FUS_OK;
  }
  if (nranks == 1 && !id128)
    return FUS_OK;
  if (!id128)
    return fail(FUS_ERR_ARG, "fus_comm_init: RCCL id missing (or set option external_transport)");
  FUSCHK(rccl_load());
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  HIPCHK(hipSetDevice(c->device));
  if (c->loopback)
    NCCLCHK(g_rccl.CommInitRank(&c->comm, 1, id, 0));
  else
    NCCLCHK(g_rccl.CommInitRank(&c->comm, nranks, id, rank));
  return FUS_OK;
}

// Round trip of n doubles through ncclSend/ncclRecv to the own rank on the library's stream:
// checks the run-time RCCL binding and the grouped send/recv pattern the halo exchange uses.
int fus_comm_selftest(fus_ctx* c, int64_t n)
{
  if (!c || n < 2)
    return fail(FUS_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  FUSCHK(rccl_load());
  ncclComm_t comm = c->comm;
  bool own = false;
  if (!comm)
  {
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    NCCLCHK(g_rccl.CommInitRank(&comm, 1, id, 0));
    own = true;
  }
  std::vector<double> h(n), back(n, 0.0);
  for (int64_t i = 0; i < n; ++i)
    h[i] = 0.5 * (double)i - 3.0;
  double *a = nullptr, *b = nullptr;
  HIPCHK(hipMalloc((void**)&a, n * sizeof(double)));
  HIPCHK(hipMalloc((void**)&b, n * sizeof(double)));
  HIPCHK(hipMemcpyAsync(a, h.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(b, 0, n * sizeof(double), c->stream));
  const int self = own ? 0 : c->rank;
  NCCLCHK(g_rccl.GroupStart());
  NCCLCHK(g_rccl.Send(a, n, ncclDouble, self, comm, c->stream));
  NCCLCHK(g_rccl.Recv(b, n, ncclDouble, self, comm, c->stream));
  NCCLCHK(g_rccl.GroupEnd());
  HIPCHK(hipMemcpyAsync(back.data(), b, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  // and the all-reduce of fus_comm_allreduce (over this communicator's ranks; a 1-rank one returns its input)
  double red[2] = {0, 0};
  if (own || c->nranks == 1)
  {
    NCCLCHK(g_rccl.AllReduce(a, a, 2, ncclDouble, ncclMin, comm, c->stream));
    HIPCHK(hipMemcpyAsync(red, a, sizeof(red), hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  (void)hipFree(a), (void)hipFree(b);
  if (own)
    g_rccl.CommDestroy(comm);
  for (int64_t i = 0; i < n; ++i)
    if (back[i] != h[i])
      return fail(FUS_ERR_RCCL, "RCCL self send/recv returned wrong data");
  if ((own || c->nranks == 1) && n >= 2 && (red[0] != h[0] || red[1] != h[1]))
    return fail(FUS_ERR_RCCL, "RCCL all-reduce over one rank changed the data");
  return FUS_OK;
}

// Scalars every rank needs: the global minimum cell size behind the time step
// (MPI_Reduce(MIN) + MPI_Bcast, fenicsx-sf-naive/examples/linear_planewave2d_1/main.cpp:67-68) and the
// sums behind norms (fem::assemble_scalar + MPI sum, :151-157), over the library's RCCL communicator.
int fus_comm_allreduce(fus_ctx* c, double* values, int n, int op)
{
  if (!c || !values || n < 1 || op < FUS_SUM || op > FUS_MAX)
    return fail(FUS_ERR_ARG, "bad argument");
  if (c->nranks <= 1 || c->loopback)
    return FUS_OK;
  if (c->local_group || !c->comm)
    return fail(FUS_ERR_STATE, "fus_comm_allreduce needs the RCCL communicator (in-process groups and the "
                               "external transport reduce on the caller's side)");
  HIPCHK(hipSetDevice(c->device));
  double* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, n * sizeof(double)));
  HIPCHK(hipMemcpyAsync(d, values, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  const ncclRedOp_t rop = op == FUS_SUM ? ncclSum : (op == FUS_MIN ? ncclMin : ncclMax);
  ncclResult_t r = g_rccl.AllReduce(d, d, (size_t)n, ncclDouble, rop, c->comm, c->stream);
  if (r == ncclSuccess)
  {
    HIPCHK(hipMemcpyAsync(values, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  (void)hipFree(d);
  NCCLCHK(r);
  return FUS_OK;
}

// Smallest cell size of the local mesh, size = largest vertex-to-vertex distance of a cell
// (dolfinx::mesh::h as used by linear_planewave2d_1/main.cpp:60-64).
int fus_op_hmin(fus_op* op, double* hmin)
{
  if (!op || !hmin)
    return fail(FUS_ERR_ARG, "null argument");
  const int nv1 = op->tdim == 3 ? 8 : 4;
  const int nvg = op->geom_order == 1 ? nv1 : (op->tdim == 3 ? 27 : 9);
  int vert[8];
  for (int v = 0; v < nv1; ++v)  // second order: tensor nodes with every index in {0, 2}
    vert[v] = op->geom_order == 1 ? v : 2 * (v & 1) + 6 * ((v >> 1) & 1) + 18 * (v >> 2);
  auto coord = [&](int64_t node, int j) -> double
  {
    const size_t k = 3 * (size_t)node + j;
    return op->dtype == FUS_F64 ? reinterpret_cast<const double*>(op->h_geom_x.data())[k]
                                : (double)reinterpret_cast<const float*>(op->h_geom_x.data())[k];
  };
  double best = std::numeric_limits<double>::infinity();
  for (int64_t cidx = 0; cidx < op->ncells; ++cidx)
  {
    double h2 = 0;
    for (int a = 0; a < nv1; ++a)
      for (int b = a + 1; b < nv1; ++b)
      {
        double d2 = 0;
        for (int j = 0; j < 3; ++j)
        {
          const double d = coord(op->h_geom_dm[cidx * nvg + vert[a]], j) - coord(op->h_geom_dm[cidx * nvg + vert[b]], j);
          d2 += d * d;
        }
        h2 = std::max(h2, d2);
      }
    best = std::min(best, h2);
  }
  *hmin = std::sqrt(best);
  return FUS_OK;
}

int fus_op_create(fus_ctx* c, int tdim, int P, int dtype, int64_t ncells, int64_t ndofs,
                  const int32_t* tensor_dofmap, const double* nodes1d, const void* geom_x,
                  int64_t nnodes, const int32_t* geom_dofmap, int geom_order, fus_op** out)
{
  if (!c || !out || !tensor_dofmap || !nodes1d || !geom_x || !geom_dofmap)
    return fail(FUS_ERR_ARG, "null argument");
  if (tdim != 2 && tdim != 3)
    return fail(FUS_ERR_ARG, "tdim must be 3 (hexahedra) or 2 (quadrilaterals)");
  if (P < 2 || P > 10)
    return fail(FUS_ERR_ARG, "unsupported polynomial degree (2..10)");
  if (dtype != FUS_F64 && dtype != FUS_F32)
    return fail(FUS_ERR_ARG, "dtype must be FUS_F32 or FUS_F64");
  if (geom_order != 1 && geom_order != 2)
    return fail(FUS_ERR_ARG, "geometry order must be 1 (2^tdim vertices) or 2 (3^tdim nodes, tensor order)");
  if (ncells <= 0 || ndofs <= 0 || nnodes <= 0)
    return fail(FUS_ERR_ARG, "empty mesh");
  const int N = P + 1;
  if (!is_gll_node_set(N, nodes1d))
    return fail(FUS_ERR_ARG, "nodes1d are not the GLL points of [0,1]");
  HIPCHK(hipSetDevice(c->device));
  std::unique_ptr<fus_op> op(new fus_op());
  op->deterministic = c->deterministic;
  op->nfields = c->fields;
  op->ctx = c, op->P = P, op->N = N, op->tdim = tdim, op->dtype = dtype;
  op->Nd = tdim == 3 ? N * N * N : N * N;
  op->ts = dtype == FUS_F64 ? 8 : 4;
  op->ncells = ncells, op->ndofs = ndofs, op->nnodes = nnodes;
  op->nodes.assign(nodes1d, nodes1d + N);
  op->wts = gll_weights_at(N, nodes1d);
  op->D = dphi_table(N, nodes1d);
  op->h_geom_x.assign(static_cast<const char*>(geom_x),
                      static_cast<const char*>(geom_x) + (size_t)nnodes * 3 * op->ts);
  op->geom_order = geom_order;
  op->geom_nv = geom_order == 2 ? (tdim == 3 ? 27 : 9) : (tdim == 3 ? 8 : 4);
  op->h_geom_dm.assign(geom_dofmap, geom_dofmap + ncells * op->geom_nv);
  op->h_dofmap.assign(tensor_dofmap, tensor_dofmap + ncells * op->Nd);
  for (int64_t k = 0; k < ncells * op->geom_nv; ++k)
    if (geom_dofmap[k] < 0 || geom_dofmap[k] >= nnodes)
      return fail(FUS_ERR_ARG, "geometry dofmap entry out of range");
  int r = op_build(op.get(), nullptr);
  if (r != FUS_OK)
    return r;
  *out = op.release();
  return FUS_OK;
}

int fus_op_destroy(fus_op* op)
{
  if (!op)
    return FUS_OK;
  (void)hipSetDevice(op->ctx->device);
  (void)hipStreamSynchronize(op->ctx->stream);
  for (void* q : op->allocs)
    (void)hipFree(q);
  for (void* q : {(void*)op->d_pack_idx, (void*)op->d_uidx, (void*)op->d_uptr, (void*)op->d_usrc,
                  op->d_sendbuf, op->d_recvbuf})
    (void)hipFree(q);
  delete op;
  return FUS_OK;
}

int fus_stiffness_apply(fus_op* op, const void* x, const void* coeffs, void* y, int space)
{
  if (!op || !x || !coeffs || !y)
    return fail(FUS_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(op->ctx->device));
  return d_op_apply(op, OP_STIFFNESS, x, coeffs, y, space);
}

int fus_mass_apply(fus_op* op, const void* x, const void* coeffs, void* y, int space)
{
  if (!op || !x || !coeffs || !y)
    return fail(FUS_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(op->ctx->device));
  return d_op_apply(op, OP_MASS, x, coeffs, y, space);
}

// sum over the LOCAL cells of the GLL-quadrature integral of x^2 (= x . M(1) x with the local,
// un-exchanged mass action): summed over ranks (fus_comm_allreduce, FUS_SUM) it is the squared L2
// norm the examples print (fem::assemble_scalar of u*u*dx, linear_planewave2d_1/main.cpp:151-157).
int fus_op_norm2(fus_op* op, const void* x, int space, double* out)
{
  if (!op || !x || !out)
    return fail(FUS_ERR_ARG, "null argument");
  const size_t ts = op->ts;
  std::vector<char> xh(op->ndofs * ts), yh(op->ndofs * ts, 0), ones(op->ncells * ts);
  if (space == FUS_HOST)
    memcpy(xh.data(), x, xh.size());
  else
  {
    HIPCHK(hipSetDevice(op->ctx->device));
    HIPCHK(hipMemcpy(xh.data(), x, xh.size(), hipMemcpyDeviceToHost));
  }
  for (int64_t i = 0; i < op->ncells; ++i)
    if (ts == 8)
      reinterpret_cast<double*>(ones.data())[i] = 1.0;
    else
      reinterpret_cast<float*>(ones.data())[i] = 1.0f;
  FUSCHK(fus_mass_apply(op, xh.data(), ones.data(), yh.data(), FUS_HOST));
  long double acc = 0;
  for (int64_t i = 0; i < op->ndofs; ++i)
    acc += ts == 8 ? (long double)reinterpret_cast<double*>(xh.data())[i] * reinterpret_cast<double*>(yh.data())[i]
                   : (long double)reinterpret_cast<float*>(xh.data())[i] * reinterpret_cast<float*>(yh.data())[i];
  *out = (double)acc;
  return FUS_OK;
}

int fus_op_get_geometry(fus_op* op, void* G, void* detJ)
{
  if (!op)
    return fail(FUS_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(op->ctx->device));
  return d_op_get_geometry(op, G, detJ);
}

int fus_op_get_tables(fus_op* op, double* weights, double* dphi)
{
  if (!op)
    return fail(FUS_ERR_ARG, "null argument");
  if (weights)
    memcpy(weights, op->wts.data(), sizeof(double) * op->N);
  if (dphi)
    memcpy(dphi, op->D.data(), sizeof(double) * op->N * op->N);
  return FUS_OK;
}

static void layout_info(const Layout& L, size_t ts, int64_t out[8])
{
  out[0] = L.nblocks, out[1] = L.n_interior, out[2] = L.n_shared, out[3] = L.npairs;
  out[4] = L.max_nloc, out[5] = (int64_t)L.shapes.size();
  out[6] = (int64_t)L.lds_bytes(ts);
  out[7] = L.n_internal;
}

int fus_op_is_affine(fus_op* op) { return (op && op->affine) ? 1 : 0; }
int fus_op_geometry_mode(fus_op* op) { return !op ? 0 : (op->affine ? 1 : (op->trilinear ? 2 : 0)); }
int fus_op_uses_mfma(fus_op* op) { return (op && op->mfma) ? 1 : 0; }
int fus_op_uses_diag_metric(fus_op* op) { return (op && op->diag) ? 1 : 0; }
int fus_op_uses_mfma4(fus_op* op)
{
  // kernels.hpp mf4_contract_b: N = 8, fp64, trilinear geometry kernel, scalar (not the opt-in 16x16x4) form
  return (op && FUS_MF4 && op->P == 7 && op->dtype == FUS_F64 && op->tdim == 3 && op->trilinear && !op->mfma) ? 1 : 0;
}
int fus_op_uses_pack32(fus_op* op) { return (op && op->pk && !op->mfma) ? 1 : 0; }

int fus_op_info(fus_op* op, int64_t out[8])
{
  if (!op || !out)
    return fail(FUS_ERR_ARG, "null argument");
  layout_info(op->L, op->ts, out);
  out[6] = (int64_t)op->lds_bytes;  // what the block kernel is launched with (fields, geometry mode)
  return FUS_OK;
}

int fus_layout_check(int P, int64_t ncells, int64_t ndofs, const int32_t* tensor_dofmap,
                     const double* centroids, int block_elems, int waves, int64_t out[8])
{
  return fus_layout_check_ex(3, P, ncells, ndofs, tensor_dofmap, centroids, block_elems, waves, nullptr, out);
}

int fus_layout_check_ex(int tdim, int P, int64_t ncells, int64_t ndofs, const int32_t* tensor_dofmap,
                        const double* centroids, int block_elems, int waves,
                        const uint8_t* force_shared, int64_t out[8])
{
  if (!tensor_dofmap || !centroids || !out)
    return fail(FUS_ERR_ARG, "null argument");
  Layout L;
  std::string err = build_layout(L, P, ncells, ndofs, tensor_dofmap, centroids, block_elems, waves,
                                 force_shared, tdim);
  if (!err.empty())
    return fail(FUS_ERR_ARG, "layout: " + err);
  err = verify_layout(L, tensor_dofmap);
  if (!err.empty())
    return fail(FUS_ERR_STATE, "layout verification: " + err);
  layout_info(L, 8, out);
  return FUS_OK;
}

int fus_facet_diag(fus_op* op, int64_t nfacets, const int32_t* fc, const int32_t* fl,
                   const void* cellcoef, void* out)
{
  if (!op || !out || (nfacets > 0 && (!fc || !fl || !cellcoef)))
    return fail(FUS_ERR_ARG, "null argument");
  for (int64_t f = 0; f < nfacets; ++f)
    if (fc[f] < 0 || fc[f] >= op->ncells || fl[f] < 0 || fl[f] > 5)
      return fail(FUS_ERR_ARG, "facet (cell, local facet) out of range");
  if (op->dtype == FUS_F64)
    facet_diag_host<double>(op, nfacets, fc, fl, static_cast<const double*>(cellcoef),
                            static_cast<double*>(out));
  else
    facet_diag_host<float>(op, nfacets, fc, fl, static_cast<const float*>(cellcoef),
                           static_cast<float*>(out));
  return FUS_OK;
}

int fus_op_set_neighbours(fus_op* op, int nneigh, const int32_t* ranks, const int64_t* counts,
                          const int32_t* dof_idx)
{
  if (!op || nneigh < 0 || (nneigh > 0 && (!ranks || !counts || !dof_idx)))
    return fail(FUS_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(op->ctx->device));
  hipStream_t st = op->ctx->stream;
  if (!op->neigh.empty())
    return fail(FUS_ERR_STATE, "neighbours already set");
  {
    // dofs other ranks hold as well are classified shared: rebuild the layout with that mask
    int64_t total = 0;
    for (int k = 0; k < nneigh; ++k)
      total += counts[k];
    std::vector<uint8_t> mask(op->ndofs, 0);
    for (int64_t j = 0; j < total; ++j)
    {
      if (dof_idx[j] < 0 || dof_idx[j] >= op->ndofs)
        return fail(FUS_ERR_ARG, "shared dof index out of range");
      mask[dof_idx[j]] = 1;
    }
    HIPCHK(hipStreamSynchronize(st));
    FUSCHK(op_build(op, mask.data()));
  }
  std::vector<std::pair<int, int>> order;  // (rank, k)
  std::vector<int64_t> off(nneigh + 1, 0);
  for (int k = 0; k < nneigh; ++k)
    off[k + 1] = off[k] + counts[k], order.emplace_back(ranks[k], k);
  std::sort(order.begin(), order.end());
  if (off[nneigh] > 2000000000ll)
    return fail(FUS_ERR_LIMIT, "halo exceeds int32 indexing");
  // concatenated neighbour lists in ascending rank order
  std::vector<int32_t> pack_idx;
  for (auto& rk : order)
  {
    const int k = rk.second;
    if (rk.first == op->ctx->rank)
      return fail(FUS_ERR_ARG, "a rank cannot be its own neighbour");
    Neigh nb;
    nb.rank = rk.first, nb.count = counts[k], nb.off = (int64_t)pack_idx.size();
    for (int64_t j = 0; j < nb.count; ++j)
      pack_idx.push_back(op->L.dof_perm[dof_idx[off[k] + j]]);
    op->neigh.push_back(nb);
  }
  op->n_halo = (int64_t)pack_idx.size();
  // unique interface dofs and, per dof, its addends in ascending rank order (own = -1)
  std::vector<int32_t> uniq(pack_idx);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  op->n_uidx = (int64_t)uniq.size();
  std::vector<std::vector<int32_t>> addends(uniq.size());
  for (size_t n = 0; n < op->neigh.size(); ++n)
  {
    const Neigh& nb = op->neigh[n];
    // the own partial goes before the first neighbour of higher rank
    const bool first_higher = nb.rank > op->ctx->rank
                              && (n == 0 || op->neigh[n - 1].rank < op->ctx->rank);
    if (first_higher)
      for (auto& a : addends)
        a.push_back(-1);
    for (int64_t j = 0; j < nb.count; ++j)
    {
      const size_t u = std::lower_bound(uniq.begin(), uniq.end(), pack_idx[nb.off + j]) - uniq.begin();
      addends[u].push_back((int32_t)(nb.off + j));
    }
  }
  if (op->neigh.empty() || op->neigh.back().rank < op->ctx->rank)
    for (auto& a : addends)
      a.push_back(-1);
  std::vector<int32_t> uptr(uniq.size() + 1, 0), usrc;
  for (size_t u = 0; u < uniq.size(); ++u)
  {
    // a dof not shared with the lower ranks still has its own partial in rank position: the -1
    // was appended for every dof at the right place above, so the list is already ordered
    uptr[u] = (int32_t)usrc.size();
    usrc.insert(usrc.end(), addends[u].begin(), addends[u].end());
  }
  uptr[uniq.size()] = (int32_t)usrc.size();
  auto up = [&](int32_t** d, const std::vector<int32_t>& v) -> int
  {
    HIPCHK(hipMalloc((void**)d, std::max<size_t>(1, v.size()) * sizeof(int32_t)));
    HIPCHK(hipMemcpy(*d, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return FUS_OK;
  };
  FUSCHK(up(&op->d_pack_idx, pack_idx));
  FUSCHK(up(&op->d_uidx, uniq));
  FUSCHK(up(&op->d_uptr, uptr));
  FUSCHK(up(&op->d_usrc, usrc));
  HIPCHK(hipMalloc(&op->d_sendbuf, std::max<size_t>(1, pack_idx.size()) * op->ts));
  HIPCHK(hipMalloc(&op->d_recvbuf, std::max<size_t>(1, pack_idx.size()) * op->ts));
  return FUS_OK;
}

int fus_model_create(fus_ctx* c, int kind, fus_op* op, const void* c0, const void* rho0,
                     const void* delta0, const void* beta0, int64_t nfacets,
                     const int32_t* facet_cells, const int32_t* facet_local,
                     const int32_t* facet_tags, double freq, double amp, double speed,
                     fus_model** out)
{
  if (!c || !op || !c0 || !rho0 || !out)
    return fail(FUS_ERR_ARG, "null argument");
  if (kind != FUS_LINEAR && kind != FUS_LOSSY && kind != FUS_WESTERVELT)
    return fail(FUS_ERR_ARG, "unknown model kind");
  if (kind == FUS_LINEAR && (delta0 || beta0))
    return fail(FUS_ERR_ARG, "delta0/beta0 must be NULL for FUS_LINEAR");
  if (kind == FUS_LOSSY && (!delta0 || beta0))
    return fail(FUS_ERR_ARG, "FUS_LOSSY needs delta0 (and no beta0)");
  if (kind == FUS_WESTERVELT && (!delta0 || !beta0))
    return fail(FUS_ERR_ARG, "FUS_WESTERVELT needs delta0 and beta0");
  if (kind != FUS_LINEAR && op->nfields < 2)
    return fail(FUS_ERR_STATE, "FUS_LOSSY / FUS_WESTERVELT need operator data created with option fields=2");
  if (nfacets > 0 && (!facet_cells || !facet_local || !facet_tags))
    return fail(FUS_ERR_ARG, "null facet arrays");
  if (!(freq > 0) || !(speed > 0))
    return fail(FUS_ERR_ARG, "freq and speed must be positive");
  HIPCHK(hipSetDevice(c->device));
  std::unique_ptr<fus_model> m(new fus_model());
  m->ctx = c, m->op = op, m->kind = kind, m->freq = freq, m->amp = amp, m->speed = speed;
  m->forms = c->forms;
  m->lean_rk4 = c->lean_rk4;
  int r = d_model_setup(m.get(), c0, rho0, delta0, beta0, nfacets, facet_cells, facet_local, facet_tags);
  if (r == FUS_OK && !c->local_group)
  {
    // add the sharers' parts of the lumped mass over RCCL, then finish; with the in-process
    // transport this happens in fus_group_finish_setup once every member exists.  The boundary
    // weights of interface dofs stay per-rank: each rank adds the term of its own facets to its
    // partial b and the exchange sums them (v_n is identical on all sharers).
    r = d_halo_sum(op, setup_halo_vector(m.get(), 0));
    if (r == FUS_OK && m->mn1)
      r = d_halo_sum(op, setup_halo_vector(m.get(), 1));
    if (r == FUS_OK)
      r = d_setup_finish(m.get());
  }
  if (r != FUS_OK)
  {
    for (void* q : m->allocs)
      (void)hipFree(q);
    return r;
  }
  *out = m.release();
  return FUS_OK;
}

// ---- in-process transport (single-GPU rehearsal of the multi-rank path) -----------------------
int fus_comm_init_local(fus_ctx** ctxs, int n)
{
  if (!ctxs || n < 1)
    return fail(FUS_ERR_ARG, "bad group");
  for (int i = 0; i < n; ++i)
  {
    if (!ctxs[i])
      return fail(FUS_ERR_ARG, "null ctx in group");
    ctxs[i]->rank = i, ctxs[i]->nranks = n, ctxs[i]->local_group = n > 1;
  }
  return FUS_OK;
}

static int group_halo(fus_model** ms, int n, int which_setup_vec)
{
  std::vector<fus_op*> ops(n);
  for (int i = 0; i < n; ++i)
  {
    ops[i] = ms[i]->op;
    void* v = which_setup_vec >= 0 ? setup_halo_vector(ms[i], which_setup_vec) : ms[i]->b;
    FUSCHK(d_halo_pack(ops[i], v));
  }
  FUSCHK(halo_exchange_local(ops.data(), n));
  if (which_setup_vec >= 0)
    for (int i = 0; i < n; ++i)
      FUSCHK(d_halo_unpack(ops[i], setup_halo_vector(ms[i], which_setup_vec)));
  return FUS_OK;
}

int fus_group_finish_setup(fus_model** ms, int n)
{
  if (!ms || n < 1)
    return fail(FUS_ERR_ARG, "bad group");
  FUSCHK(group_halo(ms, n, 0));  // lumped mass (see fus_model_create)
  if (ms[0]->mn1)
    FUSCHK(group_halo(ms, n, 1));  // Westervelt: diagonal of M(nlin1)
  for (int i = 0; i < n; ++i)
    FUSCHK(d_setup_finish(ms[i]));
  return FUS_OK;
}

int fus_group_rk4_steps(fus_model** ms, int n, double t0, double dt, int64_t nsteps)
{
  if (!ms || n < 1)
    return fail(FUS_ERR_ARG, "bad group");
  for (int i = 0; i < n; ++i)
    if (!ms[i]->setup_done || !ms[i]->initialised)
      return fail(FUS_ERR_STATE, "group member not set up / initialised");
  double t = t0;
  for (int64_t s = 0; s < nsteps; ++s)
  {
    for (int st = 0; st < ms[0]->rk_order; ++st)
    {
      for (int i = 0; i < n; ++i)
        FUSCHK(d_stage_begin(ms[i], st, t, dt));   // includes the pack
      std::vector<fus_op*> ops(n);
      for (int i = 0; i < n; ++i)
        ops[i] = ms[i]->op;
      FUSCHK(halo_exchange_local(ops.data(), n));
      for (int i = 0; i < n; ++i)
        FUSCHK(d_stage_end(ms[i], st, t, dt));
    }
    if (ms[0]->rk_order != 4)
      for (int i = 0; i < n; ++i)
        std::swap(ms[i]->u_, ms[i]->u0), std::swap(ms[i]->v_, ms[i]->v0);
    t += dt;
  }
  for (int i = 0; i < n; ++i)
    HIPCHK(hipStreamSynchronize(ms[i]->ctx->stream));
  return FUS_OK;
}

// ---- external transport: the stage / setup halves one by one, the exchange is the caller's ----------
int fus_op_halo_layout(fus_op* op, int* nneigh, int32_t* ranks, int64_t* counts, int64_t* offsets)
{
  if (!op || !nneigh)
    return fail(FUS_ERR_ARG, "null argument");
  *nneigh = (int)op->neigh.size();
  for (size_t k = 0; k < op->neigh.size(); ++k)
  {
    if (ranks)
      ranks[k] = op->neigh[k].rank;
    if (counts)
      counts[k] = op->neigh[k].count;
    if (offsets)
      offsets[k] = op->neigh[k].off;
  }
  return FUS_OK;
}

int fus_op_halo_buffers(fus_op* op, void** send_dev, void** recv_dev, int64_t* nvalues)
{
  if (!op)
    return fail(FUS_ERR_ARG, "null op");
  if (send_dev)
    *send_dev = op->d_sendbuf;
  if (recv_dev)
    *recv_dev = op->d_recvbuf;
  if (nvalues)
    *nvalues = op->n_halo;
  return FUS_OK;
}

static int external_model(fus_model* m)
{
  if (!m)
    return fail(FUS_ERR_ARG, "null model");
  if (!m->ctx->external_transport)
    return fail(FUS_ERR_STATE, "set option external_transport before fus_comm_init to drive the halves yourself");
  HIPCHK(hipSetDevice(m->ctx->device));
  return FUS_OK;
}

int fus_model_setup_count(fus_model* m) { return m ? (m->mn1 ? 2 : 1) : 0; }

int fus_model_setup_pack(fus_model* m, int k)
{
  FUSCHK(external_model(m));
  if (k < 0 || k >= fus_model_setup_count(m))
    return fail(FUS_ERR_ARG, "setup vector index out of range");
  FUSCHK(d_halo_pack(m->op, setup_halo_vector(m, k)));
  HIPCHK(hipStreamSynchronize(m->ctx->stream));   // the send buffer is complete on return
  return FUS_OK;
}

int fus_model_setup_unpack(fus_model* m, int k)
{
  FUSCHK(external_model(m));
  if (k < 0 || k >= fus_model_setup_count(m))
    return fail(FUS_ERR_ARG, "setup vector index out of range");
  return d_halo_unpack(m->op, setup_halo_vector(m, k));
}

int fus_model_setup_finish(fus_model* m)
{
  FUSCHK(external_model(m));
  return d_setup_finish(m);
}

int fus_model_stage_begin(fus_model* m, int stage, double t, double dt)
{
  FUSCHK(external_model(m));
  if (!m->initialised || !m->setup_done)
    return fail(FUS_ERR_STATE, "fus_model_setup_finish and fus_model_init come first");
  if (stage < 0 || stage >= m->rk_order)
    return fail(FUS_ERR_ARG, "stage out of range");
  FUSCHK(d_stage_begin(m, stage, t, dt));
  HIPCHK(hipStreamSynchronize(m->ctx->stream));   // the send buffer is complete on return
  return FUS_OK;
}

int fus_model_stage_end(fus_model* m, int stage, double t, double dt)
{
  FUSCHK(external_model(m));
  if (stage < 0 || stage >= m->rk_order)
    return fail(FUS_ERR_ARG, "stage out of range");
  FUSCHK(d_stage_end(m, stage, t, dt));
  if (stage == m->rk_order - 1 && m->rk_order != 4)
    std::swap(m->u_, m->u0), std::swap(m->v_, m->v0);  // the accumulated solution is the new state
  return FUS_OK;
}

int fus_model_set_rk_order(fus_model* m, int order)
{
  if (!m || order < 1 || order > 4)
    return fail(FUS_ERR_ARG, "rk order must be 1, 2, 3 or 4");
  m->rk_order = order;
  m->bnd_valid = false;
  return FUS_OK;
}

int fus_model_destroy(fus_model* m)
{
  if (!m)
    return FUS_OK;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  if (m->gexec)
    (void)hipGraphExecDestroy(m->gexec);
  if (m->op && m->op->bnd_owner == m)
    m->op->bnd_owner = nullptr;
  for (void* q : m->allocs)
    (void)hipFree(q);
  delete m;
  return FUS_OK;
}

int fus_model_init(fus_model* m)
{
  if (!m)
    return fail(FUS_ERR_ARG, "null model");
  HIPCHK(hipSetDevice(m->ctx->device));
  const size_t bytes = (size_t)m->op->L.n_internal * m->op->ts;
  for (void* v : {m->u0, m->v0, m->u_, m->v_, m->un, m->vn, m->b})
    HIPCHK(hipMemsetAsync(v, 0, bytes, m->ctx->stream));
  m->initialised = true;
  m->bnd_valid = false;  // state changed: the pseudo boundary partials are stale
  m->gsteps = 0;
  return FUS_OK;
}

// ---- receivers: point samples of the resident solution (reference: Function::eval at located cells,
// python/src/fenicsxfus/utils.py:10-47, cpp/mwe/parallel_eval_line/main.cpp:49-84) ----
static void launch_sample(fus_model* m, int which, void* out)
{
  fus_op* op = m->op;
  const void* vec = which == FUS_U ? m->u0 : m->v0;
  const dim3 grid((unsigned)((m->rc_n + 3) / 4)), blockd(256);
  hipStream_t st = m->ctx->stream;
#define FUS_SAMPLE(T, TD)                                                                          \
  hipLaunchKernelGGL((k_sample<T, TD>), grid, blockd, 0, st, m->rc_n, op->N, m->d_rc_idx,          \
                     static_cast<const T*>(m->d_rc_bas), static_cast<const T*>(vec), static_cast<T*>(out))
  if (op->dtype == FUS_F64)
  {
    if (op->tdim == 3)
      FUS_SAMPLE(double, 3);
    else
      FUS_SAMPLE(double, 2);
  }
  else
  {
    if (op->tdim == 3)
      FUS_SAMPLE(float, 3);
    else
      FUS_SAMPLE(float, 2);
  }
#undef FUS_SAMPLE
}

// after a step that ended at time t: every rec_every-th step, one sample of the receivers into the record buffer
static int model_record_step(fus_model* m, double t)
{
  if (m->rec_every <= 0)
    return FUS_OK;
  ++m->rec_step;
  if (m->rec_step % m->rec_every != 0 || m->rec_n >= m->rec_cap)
    return FUS_OK;
  launch_sample(m, m->rec_which, static_cast<char*>(m->d_rec) + (size_t)m->rec_n * m->rc_n * m->op->ts);
  HIPCHK(hipGetLastError());
  m->rec_times.push_back(t);
  ++m->rec_n;
  return FUS_OK;
}

int fus_model_set_receivers(fus_model* m, int64_t npts, const int32_t* cells, const double* refcoords)
{
  if (!m || npts < 0 || (npts > 0 && (!cells || !refcoords)))
    return fail(FUS_ERR_ARG, "bad argument");
  fus_op* op = m->op;
  HIPCHK(hipSetDevice(m->ctx->device));
  const int N = op->N, Nd = op->Nd, td = op->tdim;
  for (int64_t r = 0; r < npts; ++r)
    if (cells[r] < 0 || cells[r] >= op->ncells)
      return fail(FUS_ERR_ARG, "receiver cell index out of range (points outside the local mesh must be dropped by the caller)");
  // internal indices of each receiver's cell dofs; 1-D Lagrange basis on the operator's nodes at the reference coordinates
  std::vector<int32_t> idx((size_t)npts * Nd);
  std::vector<double> bas((size_t)npts * td * N);
  for (int64_t r = 0; r < npts; ++r)
  {
    const int32_t* dm = op->h_dofmap.data() + (size_t)cells[r] * Nd;
    for (int k = 0; k < Nd; ++k)
      idx[(size_t)r * Nd + k] = op->L.dof_perm[dm[k]];
    for (int d = 0; d < td; ++d)
    {
      const double X = refcoords[r * td + d];
      for (int i = 0; i < N; ++i)
      {
        double l = 1.0;
        for (int j = 0; j < N; ++j)
          if (j != i)
            l *= (X - op->nodes[j]) / (op->nodes[i] - op->nodes[j]);
        bas[((size_t)r * td + d) * N + i] = l;
      }
    }
  }
  hipStream_t st = m->ctx->stream;
  HIPCHK(hipStreamSynchronize(st));
  m->rc_n = npts;
  m->rec_every = 0, m->rec_n = 0, m->rec_cap = 0, m->rec_times.clear();
  FUSCHK(upload(m->allocs, &m->d_rc_idx, idx, st));
  if (op->dtype == FUS_F64)
  {
    double* q = nullptr;
    FUSCHK(upload(m->allocs, &q, bas, st));
    m->d_rc_bas = q;
  }
  else
  {
    std::vector<float> bf(bas.begin(), bas.end());
    float* q = nullptr;
    FUSCHK(upload(m->allocs, &q, bf, st));
    m->d_rc_bas = q;
    HIPCHK(hipStreamSynchronize(st));  // bf leaves scope
  }
  FUSCHK(dalloc_bytes(m->allocs, &m->d_rc_out, (size_t)npts * op->ts, true, st));
  HIPCHK(hipStreamSynchronize(st));
  return FUS_OK;
}

int fus_model_sample(fus_model* m, int which, void* out, int space)
{
  if (!m || !out || (which != FUS_U && which != FUS_V))
    return fail(FUS_ERR_ARG, "bad argument");
  if (m->rc_n == 0)
    return m->d_rc_idx ? FUS_OK : fail(FUS_ERR_STATE, "fus_model_set_receivers has not been called");
  HIPCHK(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  launch_sample(m, which, space == FUS_HOST ? m->d_rc_out : out);
  HIPCHK(hipGetLastError());
  if (space == FUS_HOST)
    HIPCHK(hipMemcpyAsync(out, m->d_rc_out, (size_t)m->rc_n * m->op->ts, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return FUS_OK;
}

int fus_model_record(fus_model* m, int which, int every, int64_t capacity)
{
  if (!m || (which != FUS_U && which != FUS_V) || every < 0 || capacity < 0)
    return fail(FUS_ERR_ARG, "bad argument");
  if (every > 0 && !m->d_rc_idx)
    return fail(FUS_ERR_STATE, "fus_model_set_receivers has not been called");
  HIPCHK(hipSetDevice(m->ctx->device));
  m->rec_every = every, m->rec_which = which, m->rec_n = 0, m->rec_step = 0, m->rec_times.clear();
  if (every > 0 && capacity > m->rec_cap)
  {
    FUSCHK(dalloc_bytes(m->allocs, &m->d_rec, (size_t)capacity * std::max<int64_t>(m->rc_n, 1) * m->op->ts, false, m->ctx->stream));
    m->rec_cap = capacity;
  }
  return FUS_OK;
}

int fus_model_get_records(fus_model* m, void* out, double* times, int64_t* nrec)
{
  if (!m || !nrec)
    return fail(FUS_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(m->ctx->device));
  const int64_t n = m->rec_n;
  if (out && n > 0 && m->rc_n > 0)
  {
    HIPCHK(hipMemcpyAsync(out, m->d_rec, (size_t)n * m->rc_n * m->op->ts, hipMemcpyDeviceToHost, m->ctx->stream));
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
  }
  if (times)
    for (int64_t k = 0; k < n; ++k)
      times[k] = m->rec_times[k];
  *nrec = n;
  return FUS_OK;
}

int fus_model_rk4(fus_model* m, double t0, double tf_, double dt_, int64_t* nsteps)
{
  if (!m)
    return fail(FUS_ERR_ARG, "null model");
  if (!m->initialised || !m->setup_done)
    return fail(FUS_ERR_STATE, "fus_model_init (or fus_model_set) must be called before rk4");
  if (!(dt_ > 0))
    return fail(FUS_ERR_ARG, "dt must be positive");
  if (m->ctx->local_group)
    return fail(FUS_ERR_STATE, "in-process transport: use fus_group_rk4_steps");
  HIPCHK(hipSetDevice(m->ctx->device));
  int64_t step = 0;
  if (m->op->dtype == FUS_F64)
  {
    double t = t0, tf = tf_, dt = dt_;
    while (t < tf)
    {
      dt = std::min(dt, tf - t);
      FUSCHK(d_model_step(m, t, dt));
      t += dt;
      ++step;
      FUSCHK(model_record_step(m, t));
    }
  }
  else
  {
    float t = (float)t0, tf = (float)tf_, dt = (float)dt_;
    while (t < tf)
    {
      dt = std::min(dt, tf - t);
      FUSCHK(d_model_step(m, t, dt));
      t += dt;
      ++step;
      FUSCHK(model_record_step(m, t));
    }
  }
  HIPCHK(hipStreamSynchronize(m->ctx->stream));
  if (nsteps)
    *nsteps = step;
  return FUS_OK;
}

int fus_model_rk4_steps(fus_model* m, double t0, double dt, int64_t nsteps)
{
  if (!m)
    return fail(FUS_ERR_ARG, "null model");
  if (!m->initialised || !m->setup_done)
    return fail(FUS_ERR_STATE, "fus_model_init (or fus_model_set) must be called before rk4");
  if (m->ctx->local_group)
    return fail(FUS_ERR_STATE, "in-process transport: use fus_group_rk4_steps");
  HIPCHK(hipSetDevice(m->ctx->device));
  double t = t0;
  for (int64_t s = 0; s < nsteps; ++s)
  {
    FUSCHK(d_model_step(m, t, dt));
    t += dt;
    FUSCHK(model_record_step(m, t));
  }
  return FUS_OK;
}

int fus_model_get(fus_model* m, int which, void* out, int space)
{
  if (!m || !out || (which != FUS_U && which != FUS_V))
    return fail(FUS_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(m->ctx->device));
  return m->op->dtype == FUS_F64 ? model_getset<double>(m, which, out, space, false)
                                 : model_getset<float>(m, which, out, space, false);
}

int fus_model_set(fus_model* m, int which, const void* in, int space)
{
  if (!m || !in || (which != FUS_U && which != FUS_V))
    return fail(FUS_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(m->ctx->device));
  m->initialised = true;
  m->bnd_valid = false;  // state changed: the pseudo boundary partials are stale
  m->gsteps = 0;
  return m->op->dtype == FUS_F64
             ? model_getset<double>(m, which, const_cast<void*>(in), space, true)
             : model_getset<float>(m, which, const_cast<void*>(in), space, true);
}

int fus_model_get_mass(fus_model* m, void* out)
{
  if (!m || !out)
    return fail(FUS_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(m->ctx->device));
  fus_op* op = m->op;
  hipStream_t st = m->ctx->stream;
  if (op->dtype == FUS_F64)
    hipLaunchKernelGGL((k_from_internal<double, 0>), dim3(nblk(op->ndofs)), dim3(256), 0, st,
                       op->ndofs, op->d_dof_perm, static_cast<const double*>(m->m),
                       static_cast<double*>(op->d_tmp_c));
  else
    hipLaunchKernelGGL((k_from_internal<float, 0>), dim3(nblk(op->ndofs)), dim3(256), 0, st,
                       op->ndofs, op->d_dof_perm, static_cast<const float*>(m->m),
                       static_cast<float*>(op->d_tmp_c));
  HIPCHK(hipMemcpyAsync(out, op->d_tmp_c, op->ndofs * op->ts, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return FUS_OK;
}

int64_t fus_model_ndofs(fus_model* m) { return m ? m->op->ndofs : 0; }

// Measured streaming bandwidth of this device: triad y = x + a z over three arrays of nbytes each
// (16-byte accesses, all CUs), best of `reps` launches; GB/s counts 3 * nbytes per launch.
int fus_measure_bandwidth(fus_ctx* c, int64_t nbytes, int reps, double* gbps)
{
  if (!c || !gbps || nbytes < (1 << 20) || reps < 1)
    return fail(FUS_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  typedef double D2 __attribute__((ext_vector_type(2)));
  const int64_t nvec = nbytes / 16;
  D2 *x = nullptr, *y = nullptr;
  HIPCHK(hipMalloc(&x, nvec * 16));
  HIPCHK(hipMalloc(&y, nvec * 16));
  HIPCHK(hipMemsetAsync(x, 0, nvec * 16, c->stream));
  HIPCHK(hipMemsetAsync(y, 0, nvec * 16, c->stream));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < reps + 1; ++r)  // first launch is a warm-up
  {
    HIPCHK(hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL((k_stream_copy<D2>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, c->stream, nvec,
                       static_cast<const D2*>(x), y);
    HIPCHK(hipEventRecord(e1, c->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best)
      best = ms;
  }
  (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
  (void)hipFree(x), (void)hipFree(y);
  *gbps = 2.0 * (double)(nvec * 16) / (best * 1e-3) / 1e9;   // bytes read + bytes written
  return FUS_OK;
}

int fus_profile_enable(fus_ctx* c, int on)
{
  if (!c)
    return fail(FUS_ERR_ARG, "null ctx");
  HIPCHK(hipStreamSynchronize(c->stream));
  for (auto& kv : c->profs)
    for (auto& ev : kv.second.ev)
      (void)hipEventDestroy(ev.first), (void)hipEventDestroy(ev.second);
  c->profs.clear();
  c->prof = on < 0 ? 0 : (on > 2 ? 1 : on);
  return FUS_OK;
}

int fus_profile_get(fus_ctx* c, const char* name, double* total_ms, int64_t* count)
{
  if (!c || !name)
    return fail(FUS_ERR_ARG, "null argument");
  HIPCHK(hipStreamSynchronize(c->stream));
  auto it = c->profs.find(name);
  double ms = 0;
  int64_t n = 0;
  if (it != c->profs.end())
  {
    Prof& p = it->second;
    for (auto& ev : p.ev)
    {
      float f = 0;
      HIPCHK(hipEventElapsedTime(&f, ev.first, ev.second));
      p.done_ms += f, p.done_count += 1;
      (void)hipEventDestroy(ev.first), (void)hipEventDestroy(ev.second);
    }
    p.ev.clear();
    ms = p.done_ms, n = p.done_count;
  }
  if (total_ms)
    *total_ms = ms;
  if (count)
    *count = n;
  return FUS_OK;
}

} // extern "C"
#endif  // !FUS_TU_DEGREE

#ifndef FUS_TRACE_BITS
#define FUS_TRACE_BITS 64   // scalar type whose unit exports the trace (-DFUS_TRACE_BITS=32 for the fp32 kernels)
#endif
#if defined(FUS_TRACE) && defined(FUS_TU_DEGREE) && FUS_TU_DTYPE == FUS_TRACE_BITS
// experiment builds: phase timestamps of the last k_block_op launch of this degree's unit
extern "C" int fus_debug_trace(unsigned long long* out, long long nblocks)
{
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fus::g_fus_trace), (size_t)nblocks * 8 * sizeof(unsigned long long))
                 == hipSuccess
             ? 0
             : -2;
}
#endif
