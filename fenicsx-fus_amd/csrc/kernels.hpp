// kernels.hpp -- CDNA4 (gfx950) device code of libfusmi.  Included by fusmi.hip only.
//
// Hot kernels
//   k_block_op<T,P,OP>   the sum-factorised operator action on one LDS block of elements
//                        (reference: StiffnessSpectral3D / MassSpectral3D ::operator(),
//                        cpp/fenicsx-sf/common/spectral_op.hpp:173-243, 69-86)
//   k_shared_reduce<T>   fixed-order sum of per-block partials of shared DOFs
//                        (replaces the `+=` scatter of spectral_op.hpp:240-241 across blocks)
//   k_shared_stage<T,S>  / epilogue of k_block_op: fused RK4 stage update (reference: 9 separate
//                        vector passes per stage, Linear.hpp:274-294 + :212-221)
//
// Thread mapping of k_block_op (wave = 64 lanes): lane p = (b, c) of the N x N plane
// (tensor indices 1 and 2), registers run along tensor index 0; EPW = 64 / N^2 elements per wave
// (N=5: two elements, 50 live lanes).  Index-0 contractions are pure register FMAs against the
// derivative table held in SGPRs (kernel argument); index-1/2 contractions exchange through a
// per-element LDS tile.  Geometry factors stream from HBM with 16-byte loads in a per-lane
// vector layout; nothing is read twice.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geom.hpp"

namespace fus
{

struct ShapeDev
{
  int32_t nelem, nloc, nint, nrounds;
  int64_t rounds_off, ldm_off;
};

struct BlockArgs
{
  const int32_t* blk_shape;
  const ShapeDev* shapes;
  const int32_t* blk_elem_off;
  const int32_t* blk_int_off;
  const int64_t* blk_sh_off;
  const int32_t* sh_gidx;
  const int32_t* sh_ppos;  // where each (block, shared slot) pair's partial sum goes (Layout::pair_pos)
  const int16_t* rounds;
  const uint16_t* ldm;
  int32_t nblocks;
  int32_t blk_begin; // first block of this launch (a contiguous range of blocks)
  int32_t blk_count; // blocks of this launch; workgroup w walks blocks blk_begin + w, + gridDim.x, ...
  int32_t lds_nloc;  // LDS array length (>= max nloc, even)
  int32_t lds_nelem; // LDS per-element table length (>= max elements per block, multiple of 8)
  int32_t waves;
};

// Arguments of the fused RK4 stage epilogue of k_block_op (STAGE >= 0): the model vectors in
// internal numbering, the stage's scalars and the boundary entries of block-interior dofs
// (sorted by internal index, blk_bnd_off[b]..blk_bnd_off[b+1] belong to block b).
template <typename T>
struct StageArgs
{
  const T* minv;
  T *vn, *un, *u0, *v0, *u_, *v_;
  T adt, bdt, gval;
  // classical RK4 only (stage kinds 4-7, see below): dt b_0, dt a_i (the factor this stage's input was built with:
  // un = u0 + pdt * V_{i-1}) and 1 / 3
  T b0dt, pdt, third;
  const int32_t* blk_bnd_off;
  const int32_t* bnd_idx;
  const T* bnd_src;
  const T* bnd_abs;
  // second operator input (NF == 2, lossy model: K(coef) x + K(coef2) x2, Lossy.hpp:231-232) and
  // the dg source term (BM7-SC1/forms.py:42)
  const T* x2;
  const T* coef2;
  const T* bnd_src2;
  T dgval;
  // Westervelt (nullptr otherwise): lumped mass m0 and the diagonal of M(nlin1) = M(-2 beta/(rho^2
  // c^4)); per stage the LHS is m0 + mn1 .* u_n and the RHS gains -mn1 .* v_n^2
  // (Westervelt.hpp:246-265; the mass operator is diagonal, so M(c) x = diag(M(c) 1) .* x)
  const T* m0;
  const T* mn1;
};

enum
{
  STAGE_NONE = -1  // plain operator action: b / partial slab are written, no update
};

// Fused stage-update variants (template parameter STAGE of the block and shared-dof kernels).
//   0 first stage, 1 middle stage, 3 last stage of RK4: every stage keeps the accumulators u_, v_ in HBM
//     the way Linear.hpp:282-294 does (used by the Runge-Kutta orders 1-3 and option "lean_rk4" = 0);
//   4, 5, 6, 7 = stages 0-3 of the classical RK4 WITHOUT accumulators.  With V_0 = v0, U_0 = u0 and
//         U_i = u0 + a_i dt V_{i-1},   V_i = v0 + a_i dt k_{i-1},   k_i = M^-1 (b_i - K U_i)
//     the stage slopes are affine in the stage velocities, k_i = (V_{i+1} - v0) / (a_{i+1} dt), so
//         u1 = u0 + dt/6 (v0 + 2 V_1 + 2 V_2 + V_3),   v1 = (V_1 + 2 V_2 + V_3 - v0) / 3 + dt/6 k_3
//     need nothing but the three stage velocities, which live in three rotating buffers (the host passes them as
//     vn = V_i [read], v_ = V_{i+1} [written; stage 3: V_2, read], u_ = V_1 [read, stages 2 and 3]); and u0 itself
//     is only read at stage 0, where it IS the operator's input: later stages take it from the input already in
//     LDS, u0 = U_i - a_i dt V_{i-1}.  Interior dofs move 4 / 5 / 6 / 7 values per stage (Linear.hpp's form: 5 / 9 / 9 / 6;
//     round 2's lean form: 4 / 7 / 9 / 6); the same arithmetic up to rounding (|error| ~ eps |u|, the size of the
//     reference's own accumulation error).
__host__ __device__ constexpr bool stage_is_first(int st) { return st == 0 || st == 4; }

template <typename T, int N>
struct DTab
{
  T d[N * N];    // derivative table, d[q * N + i] = phi_i'(x_q)
  T w[N], x[N];  // 1-D GLL weights and points (kernel argument -> scalar registers)
  T dt[N * N];   // its transpose, dt[i * N + q] = d[q * N + i] (read by dtab_row at N = 8)
};

// One row of the derivative table (TR = 0: d[r][.]) or of its transpose (TR = 1: d[.][r]) as scalar operands.
// fp64, N = 8: the table is 160 scalar registers and does not fit beside everything else; left to the
// register allocator it lives in vector-register lanes and every FMA that uses an entry is preceded by a
// v_readlane (+100 % vector instructions in the element trips at degree 7).  Here the row (one
// s_load_dwordx16) is loaded from the kernarg segment where it is used, through a pointer made opaque per
// use (scalar cache), and is dead after its N FMAs: -5.7 % kernel time at degree 7 trilinear.  Lower degrees
// keep the whole table in scalar registers (k_block_op copies it there before the trips): per-row loads
// measured 3-6 % slower at degree 6 (also with rows padded to one load each) and degree 5.
// Only callable from k_block_op (whose only argument starts with BlockArgs, then the DTab: KArgs).
// rows by scalar loads where the table does not fit the scalar registers: fp64 from N = 8, every type from N = 9
template <typename T, int N>
__host__ __device__ constexpr bool dtab_by_rows()
{
  return N >= 9 || (sizeof(T) == 8 && N == 8);
}

template <typename T, int N, int TR>
__device__ __forceinline__ void dtab_row(const DTab<T, N>& Dk, int r, T (&out)[N])
{
  if constexpr (dtab_by_rows<T, N>())
  {
    typedef const T __attribute__((address_space(4))) * CP;
    typedef const char __attribute__((address_space(4))) * CC;
    // KArgs = { BlockArgs A; DTab Dk; ... } (checked in k_block_op); dt follows d, w, x inside the DTab
    CP pr = (CP)((CC)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(BlockArgs) + (TR ? sizeof(T) * (N * N + 2 * N) : 0)) + r * N;
    asm volatile("" : "+s"(pr));
#pragma unroll
    for (int i = 0; i < N; ++i)
      out[i] = pr[i];
  }
  else
  {
#pragma unroll
    for (int i = 0; i < N; ++i)
      out[i] = TR ? Dk.d[i * N + r] : Dk.d[r * N + i];
  }
}

// values per 16-byte (or 8-byte) geometry load
template <typename T, int N>
struct GLoad
{
  static constexpr int VW = (sizeof(T) == 8) ? 2 : (((6 * N) % 4 == 0) ? 4 : 2);
  static constexpr int NV = 6 * N / VW;
  typedef T type __attribute__((ext_vector_type(VW)));
};

// position of geometry value v = g*N + a of lane p inside an element's 6*Nd block
template <typename T, int N>
__host__ __device__ inline int64_t g_index(int v, int p)
{
  constexpr int VW = GLoad<T, N>::VW;
  return (int64_t)((v / VW) * (N * N) + p) * VW + (v % VW);
}

#define FUS_WAVE_SYNC()                                                                            \
  do                                                                                               \
  {                                                                                                \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                         \
    __builtin_amdgcn_wave_barrier();                                                               \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                         \
  } while (0)

enum
{
  OP_STIFFNESS = 0,
  OP_MASS = 1
};

// Phase timestamps of k_block_op exist in development builds only (build.py --dev -DFUS_TRACE pulls in
// dev_trace.hpp); the product carries none.
#ifdef FUS_TRACE
#include "dev_trace.hpp"
#else
#define FUS_STAMP(blk, k)
#define FUS_TRACE_END(blk)
#endif

// waves per SIMD the GEOM_TRILINEAR block kernel is compiled for: 4 (128 VGPRs) up to degree 5, 2 above
// (degrees 6 / 7 need 161 / 184 VGPRs; capping 7 at 168 measured no gain)
#define FUS_TRI_WAVES(P) ((P) <= 5 ? 4 : 2)
// the same for the fp32 kernels of the degrees 6 and 7 (158-174 VGPRs when compiled for two waves per SIMD: three
// resident 4-wave blocks per CU; FUS_TRI32_WAVES = 4 caps them at 128)
#ifndef FUS_TRI32_WAVES
#define FUS_TRI32_WAVES 2
#endif
#define FUS_TRI_WAVES_T(T, P) ((sizeof(T) == 4 && (P) >= 6) ? FUS_TRI32_WAVES : FUS_TRI_WAVES(P))
// waves per SIMD the kernels of the degrees 8-10 are compiled for
// interior ranges of the fused stage update in flight per pass at the degrees >= 6 (one operator input, per-cell geometry)
#ifndef FUS_EPIU
#define FUS_EPIU 4
#endif
#ifndef FUS_EPIU_F32P7AFF
#define FUS_EPIU_F32P7AFF 2
#endif
// threads per workgroup the kernels of the degrees 5-10 are compiled for (256; 512 = developer probe of 8-wave blocks)
// first-batch prologue registers per thread at p = 7 (interior vectors) and at the degrees 8-10 (interior vectors, shared dofs)
#ifndef FUS_UI_P7
#define FUS_UI_P7 5
#endif
#ifndef FUS_UI_HI
#define FUS_UI_HI 7
#endif
#ifndef FUS_XCD_REMAP
#define FUS_XCD_REMAP 0   // (measured: loses, see k_block_op)
#endif
#ifndef FUS_MID_THREADS
#define FUS_MID_THREADS 256
#endif
#ifndef FUS_HI_WAVES
#define FUS_HI_WAVES 1
#endif
// Geometry source of the block operator
//   GEOM_STREAM: per-point factors G / detJw streamed from HBM (any trilinear mesh; the reference's
//                data path, precompute.hpp:101-213)
//   GEOM_AFFINE: every cell is a parallelepiped (J constant per cell): 6 + 1 numbers per CELL,
//                G(q) = Gc * w_q and detJw(q) = detc * w_q rebuilt in registers
//   GEOM_TRILINEAR: any first-order hexahedron: 21 numbers per CELL (the coefficients of the trilinear
//                map without its constant), J(q) and from it G(q) / detJw(q) recomputed per point
//                in registers (the formulas of geom.hpp); trades the 6 N^3 streamed numbers per
//                cell for ~65 flops per point
enum
{
  GEOM_STREAM = 0,
  GEOM_AFFINE = 1,
  GEOM_TRILINEAR = 2,
  // GEOM_AFFINE data on cells with mutually orthogonal edges (J^T J diagonal): the stiffness action in its
  // diagonal-metric form (elem_compute); everything else as GEOM_AFFINE
  GEOM_DIAG = 3
};
__host__ __device__ constexpr bool is_aff(int geom) { return geom == GEOM_AFFINE || geom == GEOM_DIAG; }

template <typename T>
__host__ __device__ constexpr int epiu_of(int P, int geom, int nf)
{
  // measured per kernel at 64^3 (profiles/r03_experiments.md section 11): where the kernel has no registers to spare
  // (p=5: compiled for four waves per SIMD; the streamed kernels) more ranges spill or cost a resident wave
  if (nf != 1 || geom == GEOM_STREAM || P < 5)
    return 1;
  if (P == 5)
    return (sizeof(T) == 4 && geom == GEOM_TRILINEAR) ? FUS_EPIU : 1;   // (the packed fp32 kernel: +4.5 %)
  if (sizeof(T) == 4 && P == 7)
    return geom == GEOM_TRILINEAR ? FUS_EPIU + 2 : FUS_EPIU_F32P7AFF;
  if (sizeof(T) == 8 && P == 6 && is_aff(geom))
    return 1;   // (four ranges: -4.5 % on the fp64 affine kernels at p=6)
  return FUS_EPIU;
}
// the stage update's HBM operands of the first pass requested BEFORE the barrier that ends the element trips (their
// round trip runs under the barrier wait): fp64 at the degrees <= 5 (+1.3 % trilinear / +4.3 % affine at p=4, +1.4 / +4 %
// at p=5); even or slightly negative elsewhere (fp32 p=4 -1.5 %, p=7 fp64 -2 %: the operands are live across the barrier)
template <typename T>
__host__ __device__ constexpr bool epi_early(int P)
{
  return sizeof(T) == 8 && P <= 5;
}

// numbers per cell held in LDS by the per-cell geometry modes
__host__ __device__ constexpr int geom_cell_stride(int geom)
{
  return is_aff(geom) ? 7 : (geom == GEOM_TRILINEAR ? 21 : 0);
}

// 1 / x to working precision from the hardware estimate (the per-point G of GEOM_TRILINEAR)
__device__ __forceinline__ double fast_rcp(double x)
{
  // (two Newton steps.  One cubic step r (1 + e + e^2) -- three FMAs instead of four, enough for v_rcp_f64's ~2^-23 -- and
  // hoisting the loop-invariant weight product out of the transform took 4 of a trip's 453 fp64 instructions and 8 VGPRs
  // away and measured SLOWER at every degree: p=4 -0.6 %, p=5 -5.4 %, p=6 -1.7 %, p=7 -0.8 %; profiles/r03_experiments.md
  // section 16.  The compiler's schedule of this form stays.)
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float fast_rcp(float x)
{
  float r = __builtin_amdgcn_rcpf(x);
  r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
  return r;
}

// Lane-constant part of a trilinear cell's Jacobian for the lane's tensor column (X1, X2) = (pb, pc):
// with x(X) = c0 + c100 X0 + c010 X1 + c001 X2 + c110 X0 X1 + c101 X0 X2 + c011 X1 X2 + c111 X0 X1 X2
// the columns of J are  j0 = c100 + c110 X1 + c101 X2 + c111 X1 X2  (independent of X0),
// j1 = (c010 + c011 X2) + X0 (c110 + c111 X2),  j2 = (c001 + c011 X1) + X0 (c101 + c111 X1).
// cc: the cell's 21 coefficients [c100 c010 c001 c110 c101 c011 c111][3].
template <typename T>
struct TriLane
{
  T j0[3], a1[3], d1[3], a2[3], d2[3];
  __device__ __forceinline__ void init(const T* __restrict__ cc, T pb, T pc)
  {
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
      const T c100 = cc[i], c010 = cc[3 + i], c001 = cc[6 + i], c110 = cc[9 + i], c101 = cc[12 + i],
              c011 = cc[15 + i], c111 = cc[18 + i];
      d2[i] = c101 + pb * c111;
      j0[i] = (c100 + pb * c110) + pc * d2[i];
      a1[i] = c010 + pc * c011;
      d1[i] = c110 + pc * c111;
      a2[i] = c001 + pb * c011;
    }
  }
  // stiffness::transform at X0 = pa without forming G: with r_i = rows of det * J^-1,
  // G = (w / |det|) R R^T, so  G (f0, f1, f2) = (w / |det|) R (R^T f):  f <- cf * G f
  __device__ __forceinline__ void transform(T pa, T wcf, T& f0, T& f1, T& f2) const
  {
    T j1[3], j2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
      j1[i] = a1[i] + pa * d1[i];
      j2[i] = a2[i] + pa * d2[i];
    }
    const T r0[3] = {j1[1] * j2[2] - j1[2] * j2[1], j1[2] * j2[0] - j1[0] * j2[2], j1[0] * j2[1] - j1[1] * j2[0]};
    const T r1[3] = {j2[1] * j0[2] - j2[2] * j0[1], j2[2] * j0[0] - j2[0] * j0[2], j2[0] * j0[1] - j2[1] * j0[0]};
    const T r2[3] = {j0[1] * j1[2] - j0[2] * j1[1], j0[2] * j1[0] - j0[0] * j1[2], j0[0] * j1[1] - j0[1] * j1[0]};
    const T det = j0[0] * r0[0] + j0[1] * r0[1] + j0[2] * r0[2];
    const T sc = wcf * fast_rcp(det < T(0) ? -det : det);
    T t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      t[i] = sc * (f0 * r0[i] + f1 * r1[i] + f2 * r2[i]);
    f0 = r0[0] * t[0] + r0[1] * t[1] + r0[2] * t[2];
    f1 = r1[0] * t[0] + r1[1] * t[1] + r1[2] * t[2];
    f2 = r2[0] * t[0] + r2[1] * t[1] + r2[2] * t[2];
  }
  __device__ __forceinline__ T detw(T pa, T w) const
  {
    T j1[3], j2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
      j1[i] = a1[i] + pa * d1[i];
      j2[i] = a2[i] + pa * d2[i];
    }
    const T det = j0[0] * (j1[1] * j2[2] - j1[2] * j2[1]) + j0[1] * (j1[2] * j2[0] - j1[0] * j2[2])
                  + j0[2] * (j1[0] * j2[1] - j1[1] * j2[0]);
    return (det < T(0) ? -det : det) * w;
  }
};

// ---------------------------------------------------------------------------------------------
// Per-element inputs fetched from HBM one round ahead of their use (software pipeline):
// local dof indices, geometry factors (stiffness) or detJw (mass), cell coefficient.
// TD = 2 (quadrilaterals; one value per lane, geometry [elem][3][N^2] / detJw [elem][N^2]) keeps
// its three factors in g2.
template <typename T, int N, int OP, int GEOM, int TD = 3>
struct ElemIn
{
  typedef typename GLoad<T, N>::type GV;
  static constexpr int NV = GLoad<T, N>::NV;
  int er;
  GV g[(TD == 3 && OP == OP_STIFFNESS && GEOM == GEOM_STREAM) ? NV : 1];
  T dj[(TD == 3 && OP == OP_MASS && GEOM == GEOM_STREAM) ? N : 1];
  T g2[(TD == 2) ? 3 : 1];
};

template <typename T, int N, int OP, int GEOM, int TD>
__device__ __forceinline__ void elem_fetch(ElemIn<T, N, OP, GEOM, TD>& in, int er,
                                           const T* __restrict__ geo, int elem_off, int p)
{
  constexpr int N2 = N * N, Nd = (TD == 3) ? N * N * N : N * N;
  constexpr int VW = GLoad<T, N>::VW, NV = GLoad<T, N>::NV;
  typedef typename GLoad<T, N>::type GV;
  in.er = er;
  if constexpr (TD == 2)
  {
    if (er >= 0)
    {
      const int64_t e = elem_off + er;
      if (OP == OP_STIFFNESS)
      {
#pragma unroll
        for (int k = 0; k < 3; ++k)
          in.g2[k] = geo[e * (3 * Nd) + k * Nd + p];
      }
      else
        in.g2[0] = geo[e * Nd + p];
    }
    return;
  }
  if (GEOM == GEOM_STREAM && er >= 0 && N2 <= 64)   // (degrees 8-10 read their factors where used: elem_compute_hi)
  {
    const int64_t e = elem_off + er;
    if (OP == OP_STIFFNESS)
    {
      const T* Ge = geo + e * (6 * Nd);
#pragma unroll
      for (int t = 0; t < NV; ++t)
        in.g[t] = *reinterpret_cast<const GV*>(Ge + (size_t)(t * N2 + p) * VW);
    }
    else
    {
#pragma unroll
      for (int a = 0; a < N; ++a)
        in.dj[a] = geo[e * Nd + a * N2 + p];
    }
  }
}

// One element's operator action, accumulated into the block's LDS vector y_l.
// ATOMIC: the accumulation is an LDS floating-point atomic (ds_add_f64 / ds_add_f32), so waves need
// not proceed in conflict-free rounds; otherwise a plain read-modify-write (deterministic).
// One quadrilateral element (TD = 2): lane p = (b, c) holds the single value at tensor node (b, c);
// both derivative directions exchange through the element's LDS tile
// (cpp/fenicsx-sf-naive/common/spectral_op.hpp:273-323 with the transform of :195-207; G = (xx, xy,
// yy) with xx pairing with the derivative along tensor index 0).
// HI (N^2 > 64, degrees 8-10): the element's N^2 nodes span two waves -- lane pair index p, columns p < N^2 --, the tile
// exchanges are fenced by workgroup barriers and nothing returns early (every wave of the workgroup meets every barrier;
// has_col / in.er guard the memory operations of lanes without a node or an element).
template <typename T, int N, int OP, int ATOMIC, int NF, bool HI = false>
__device__ __forceinline__ void elem_compute2d(const ElemIn<T, N, OP, GEOM_STREAM, 2>& in,
                                               const T (&Drb)[N], const T (&Drc)[N],
                                               const T (&Dcb)[N], const T (&Dcc)[N],
                                               const T* __restrict__ x_l, double* __restrict__ y_l,
                                               T* __restrict__ sA,
                                               const uint16_t* __restrict__ ldm_l,
                                               const T* __restrict__ cf_l,
                                               const T* __restrict__ x2_l,
                                               const T* __restrict__ cf2_l, int p, int b, int c,
                                               bool has_col = true)
{
  constexpr int Nd = N * N;
#define FUS_SYNC2D()                                                                               \
  do                                                                                               \
  {                                                                                                \
    if constexpr (HI)                                                                              \
      __syncthreads();                                                                             \
    else                                                                                           \
      FUS_WAVE_SYNC();                                                                             \
  } while (0)
  if (!HI && in.er < 0)
    return;
  const bool on = in.er >= 0 && has_col;
  const int er = in.er >= 0 ? in.er : 0, pp = has_col ? p : 0, bb = has_col ? b : 0, cc = has_col ? c : 0;
  const int li = on ? (int)ldm_l[er * Nd + pp] : 0;
  const T cf = (NF == 2) ? T(1) : cf_l[er];
  T Y;
  if (OP == OP_STIFFNESS)
  {
    const T X = (NF == 2) ? cf_l[er] * x_l[li] + cf2_l[er] * x2_l[li] : x_l[li];
    if (on)
      sA[pp] = X;
    FUS_SYNC2D();
    T d0 = T(0), d1 = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j)
    {
      d0 += Drb[j] * sA[j * N + cc];
      d1 += Drc[j] * sA[bb * N + j];
    }
    const T F0 = cf * (in.g2[0] * d0 + in.g2[1] * d1);
    const T F1 = cf * (in.g2[1] * d0 + in.g2[2] * d1);
    FUS_SYNC2D();
    if (on)
      sA[pp] = F0;
    FUS_SYNC2D();
    T acc = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j)
      acc += Dcb[j] * sA[j * N + cc];
    FUS_SYNC2D();
    if (on)
      sA[pp] = F1;
    FUS_SYNC2D();
#pragma unroll
    for (int j = 0; j < N; ++j)
      acc += Dcc[j] * sA[bb * N + j];
    Y = acc;
    if constexpr (HI)
      __syncthreads();   // the tile is free for the next element
  }
  else
    Y = cf * x_l[li] * in.g2[0];
#undef FUS_SYNC2D
  if (on)
  {
    if (ATOMIC)
      __hip_atomic_fetch_add(&y_l[li], (double)Y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      y_l[li] += Y;
  }
}

// Index of entry (i0, i1, i2) of an element's exchange tile in the re-mapped contractions.  N = 8: the tile is read and
// written by lane (b, c) as (a, b, c) for every a, as (b, k, c) for every k and as (b, c, k) for every k; with a linear
// image two of the three patterns put the 32 lanes of an LDS lane group on 4-8 banks (42 % of the LDS cycles of the
// p=7 kernels were bank conflicts, profiles/r02_p7_remap_counters.json).  XOR-ing index 1 with the low bits of index
// 0 and index 2 with index 1 makes all three conflict free, without padding.  Other N: planes of TS entries.
// Plane stride of the exchange tile in the re-mapped contractions, and the entries reserved per element slot
// (Layout::lds_bytes reserves the same).  N = 4: one entry of padding per plane; N = 8: XOR swizzle, no padding (rtix);
// fp32 at N = 7 (the LDS-bound kernel of BASELINE configs[4]): planes of 56 -- 63 -> 56 bank-conflict cycles per element
// over the three read patterns, by enumeration (ideal 42).
template <typename T, int N>
__host__ __device__ constexpr int tile_plane_stride()
{
  return (N == 8 || N == 4) ? N * N + 1 : ((N == 7 && sizeof(T) == 4) ? 56 : N * N);
}
template <typename T, int N, int TD>
__host__ __device__ constexpr int tile_slot_entries()
{
  return (TD == 3 && N == 7 && sizeof(T) == 4) ? N * 56 : (TD == 3 ? N * N * N : N * N) + N;
}

template <int N, int TS>
__device__ __forceinline__ int rtix(int i0, int i1, int i2)
{
  if constexpr (N == 8)
    return i0 * 64 + ((i1 ^ (i0 & 3)) << 3) + (i2 ^ i1);
  else
    return i0 * TS + i1 * N + i2;
}

// Cross-lane reads without LDS (DPP modifiers of the vector ALU): lane l receives x of another lane of its 16-lane
// row.  CTRL: quad_perm (0x00-0xFF; broadcast of quad lane j = j * 0x55) or row_ror:n (0x120 + n: lane l reads lane
// (l - n) & 15 of its row).  fp32: one v_mov_b32_dpp, which hipcc folds into the consuming v_fmac_f32; fp64: two moves.
template <int CTRL>
__device__ __forceinline__ float dpp_read(float x)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_read(double x)
{
  const uint64_t u = __builtin_bit_cast(uint64_t, x);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// sum_m cb[m] * (v of lane (b - m, c))  +  sum_j cc[j] * (w of lane (b, j)):  the index-1 and index-2 parts of a
// contraction at N = 4, where lane (b, c) of an element sits at position 4 b + c of a 16-lane row
template <typename T>
__device__ __forceinline__ T dpp_contract_b(const T (&cb)[4], T v)
{
  T acc = cb[0] * v;
  acc += cb[1] * dpp_read<0x124>(v);
  acc += cb[2] * dpp_read<0x128>(v);
  acc += cb[3] * dpp_read<0x12C>(v);
  return acc;
}
template <typename T>
__device__ __forceinline__ T dpp_contract_c(const T (&cc)[4], T w)
{
  T acc = cc[0] * dpp_read<0x00>(w);
  acc += cc[1] * dpp_read<0x55>(w);
  acc += cc[2] * dpp_read<0xAA>(w);
  acc += cc[3] * dpp_read<0xFF>(w);
  return acc;
}

// N = 4 (degree 3) per-cell geometry kernels: index-1 / index-2 contractions by wavefront shuffles (option / build flag
// FUS_DPP4; profiles/r03_experiments.md section on DPP)
// 0: never, 1: the general affine kernel (where it measured faster: -4 % fp64, -7 % fp32 at 64^3; the trilinear kernel
// lost 2 % in fp64 and was even in fp32), 2: every per-cell geometry kernel
#ifndef FUS_DPP4
#define FUS_DPP4 1
#endif
__host__ __device__ constexpr bool use_dpp4(int N, int geom)
{
  return N == 4 && ((FUS_DPP4 == 1 && geom == GEOM_AFFINE) || (FUS_DPP4 == 2 && (geom == GEOM_AFFINE || geom == GEOM_TRILINEAR)));
}

// N = 8: lane (b, c) at 8 b + c.  Value of lane (b, j) of the lane's own group of eight, through ds_swizzle in bit-mask
// mode (and_mask 0x18 keeps the group, or_mask j selects the lane; the instruction runs in the LDS pipe but touches no
// LDS memory).  A/B only (build flag FUS_SWZ8): 16 swizzles per fp64 value and contraction against 2 tile accesses.
#ifndef FUS_SWZ8
#define FUS_SWZ8 0
#endif

// N = 8, fp64: the index-1 contraction on the matrix cores straight from the registers, v_mfma_f64_4x4x4_4b_f64 (four
// independent 4x4x4 products per instruction: no padding at N = 8).  Lane layout of the instruction, found by brute force
// (tools/mfma4_probe.hip, profiles/r03_mfma4_lane_layout.txt):  A[i][k] at lane 16 k + 4 blk + i,  B[k][j] at lane
// 16 k + 4 blk + j,  D[i][j] at lane 16 i + 4 blk + j.  With the kernel's own lane = 8 b + c the B operand of block
// (beta = b & 1, c >> 2) IS the lane's register X[a]: k = b >> 1, j = c & 3 -- the four b of one parity; the other
// parity's values come from lane ^ 8 (DPP row_ror:8), and the result D (row i = q >> 1 of parity beta) lands in lane
// (q, c): no exchange tile, no wave barrier.  out(q, c) = sum_b M[q][b] in(b, c) with
//   a_own = M[2 i + beta][2 k + beta],  a_swp = M[2 i + beta][2 k + 1 - beta]   (i = lane & 3, k = lane >> 4).
#ifndef FUS_MF4
#define FUS_MF4 1
#endif
__device__ __forceinline__ double mf4_contract_b(double a_own, double a_swp, double v)
{
  const double w = dpp_read<0x128>(v);   // the value of lane ^ 8: the other parity of b
  double acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a_swp, w, 0.0, 0, 0, 0);
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a_own, v, acc, 0, 0, 0);
}
__device__ __forceinline__ float mf4_contract_b(float, float, float v) { return v; }   // (fp64 only)
template <int J>
__device__ __forceinline__ float swz8_read(float x)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), (J << 5) | 0x18));
}
template <int J>
__device__ __forceinline__ double swz8_read(double x)
{
  const uint64_t u = __builtin_bit_cast(uint64_t, x);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)u, (J << 5) | 0x18);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)(u >> 32), (J << 5) | 0x18);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
template <typename T>
__device__ __forceinline__ T swz8_contract_c(const T (&cc)[8], T w)
{
  T acc = cc[0] * swz8_read<0>(w);
  acc += cc[1] * swz8_read<1>(w);
  acc += cc[2] * swz8_read<2>(w);
  acc += cc[3] * swz8_read<3>(w);
  acc += cc[4] * swz8_read<4>(w);
  acc += cc[5] * swz8_read<5>(w);
  acc += cc[6] * swz8_read<6>(w);
  acc += cc[7] * swz8_read<7>(w);
  return acc;
}

template <typename T, int N, int OP, int ATOMIC, int NF, int GEOM>
__device__ __forceinline__ void elem_compute(const ElemIn<T, N, OP, GEOM>& in, const DTab<T, N>& Dk,
                                             const T (&Drb)[N], const T (&Drc)[N], const T (&Dcb)[N],
                                             const T (&Dcc)[N], const T* __restrict__ x_l,
                                             double* __restrict__ y_l, T* __restrict__ sA,
                                             T* __restrict__ sB, const uint16_t* __restrict__ ldm_l,
                                             const T* __restrict__ cf_l, const T* __restrict__ x2_l,
                                             const T* __restrict__ cf2_l, const T* __restrict__ gc_l,
                                             const T (&w3)[N], const T* __restrict__ D_l, T wbc, T pb,
                                             T pc, int p, int b, int c)
{
  constexpr int N2 = N * N, Nd = N * N * N;
  constexpr int VW = GLoad<T, N>::VW;
  // GEOM_TRILINEAR keeps the lane's rows / columns of the derivative table in LDS (read where used)
  // so that the kernel fits four waves per SIMD
  constexpr bool DLDS = (GEOM == GEOM_TRILINEAR) || (is_aff(GEOM) && (N == 6 || N == 7));
  if (in.er < 0)
    return;
  TriLane<T> tri;
#ifdef FUS_ABL_NOGC   // developer ablation (wrong results): the cell's map coefficients cost no LDS reads
  if (GEOM == GEOM_TRILINEAR)
  {
    T fake[21];
#pragma unroll
    for (int i = 0; i < 21; ++i)
      fake[i] = (i % 4 == 0) ? T(1) + pb * T(i) : T(0.01) * T(i) + pc;
    tri.init(fake, pb, pc);
  }
#else
  if (GEOM == GEOM_TRILINEAR)
    tri.init(gc_l + in.er * 21, pb, pc);
#endif
  int li[N];
#pragma unroll
  for (int a = 0; a < N; ++a)
    li[a] = ldm_l[in.er * Nd + a * N2 + p];
  // NF == 2: the element input is coef x + coef2 x2 (both operators share G, the action is linear)
  // and the transform coefficient becomes 1
  const T cf = (NF == 2) ? T(1) : cf_l[in.er];
  T Y[N];
  if (OP == OP_STIFFNESS)
  {
    T X[N], F0[N], F1[N], F2[N];
    if (NF == 2)
    {
      const T c1 = cf_l[in.er], c2 = cf2_l[in.er];
#pragma unroll
      for (int a = 0; a < N; ++a)
        X[a] = c1 * x_l[li[a]] + c2 * x2_l[li[a]];
    }
    else
    {
#if defined(FUS_ABL_NOGS) || defined(FUS_ABL_NOGATHER)   // developer ablation (wrong results): no gather from x_l
                                                         // (NOGS: and one scatter-add instead of N, below)
#pragma unroll
      for (int a = 0; a < N; ++a)
        X[a] = T(li[a]) * T(1e-3);
#else
#pragma unroll
      for (int a = 0; a < N; ++a)
        X[a] = x_l[li[a]];
#endif
    }
    // REMAP: the index-1 and index-2 contractions also run in registers.  Lane (b, c) re-reads the tile
    // as lane (a' = b, c) with index 1 along its registers (then as (a' = b, b' = c) with index 2 along
    // them), contracts with the derivative table in scalar registers and writes the result back for
    // the (b, c) owner: 5 reads + 5 writes + 5 reads per direction instead of 25 reads + the lane's
    // derivative rows -- the element trips of the per-cell geometry kernels are bound by the LDS port.
    // Per-cell geometry kernels of every degree up to 7 (N = 8: since the derivative-table rows come by scalar
    // loads, dtab_row, the re-mapped form is 3-17 % faster than the tile reads there too and 3 % faster than
    // the matrix-core form; in round 1, table in registers, it was 5-14 % slower); the streamed kernel keeps the
    // tile reads (+1.5 % only: it sits on the bandwidth roof).
    constexpr bool REMAP = GEOM != GEOM_STREAM && N <= 8;
    // Diagonal metric (affine cells with mutually orthogonal edges -- boxes in any orientation): G(q) = diag(g) w_q, so
    //   K x = sum_d g_d (M x .. x K1 x .. x M) x,   K1 = D^T diag(w) D  (the 1-D stiffness matrix, in Dk.d here),
    // three contractions instead of six and no pointwise transform.  Index 0 in registers; indices 1 and 2 as in
    // the re-mapped form below: one store of X, the two re-mapped reads, and one store + read per result.
    if constexpr (use_dpp4(N, GEOM))
    {
      // Wavefront-shuffle form (north_star: "per-direction 1D contractions done with wavefront shuffles"): at N = 4 an
      // element is one 16-lane DPP row, lane (b, c) at 4 b + c.  The index-1 contraction reads lanes (b - m, c) with
      // row_ror:4m, the index-2 contraction the lanes of its own quad with quad_perm broadcasts; the lane's table
      // entries for both come from the table in LDS once per element trip.  No exchange tile, no wave barrier.
      T cb[4], cc[4], tb[4], tc[4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
      {
        const int jb = (b - m) & 3;
        cb[m] = D_l[b * 4 + jb], tb[m] = D_l[jb * 4 + b];   // D[b][b-m] and its transpose D[b-m][b]
        cc[m] = D_l[c * 4 + m], tc[m] = D_l[m * 4 + c];     // D[c][j], D[j][c]
      }
#pragma unroll
      for (int q = 0; q < N; ++q)
      {
        T acc = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i)
          acc += Dk.d[q * N + i] * X[i];
        F0[q] = acc;
      }
#pragma unroll
      for (int a = 0; a < N; ++a)
      {
        F1[a] = dpp_contract_b<T>(cb, X[a]);
        F2[a] = dpp_contract_c<T>(cc, X[a]);
      }
#pragma unroll
      for (int a = 0; a < N; ++a)
      {
        if (GEOM == GEOM_TRILINEAR)
          tri.transform(Dk.x[a], Dk.w[a] * wbc * cf, F0[a], F1[a], F2[a]);
        else
        {
          T G6[6];
#pragma unroll
          for (int gi = 0; gi < 6; ++gi)
            G6[gi] = gc_l[in.er * 7 + gi] * w3[a];
          const T w0 = F0[a], w1 = F1[a], w2 = F2[a];
          F0[a] = cf * (G6[0] * w0 + G6[1] * w1 + G6[2] * w2);
          F1[a] = cf * (G6[1] * w0 + G6[3] * w1 + G6[4] * w2);
          F2[a] = cf * (G6[2] * w0 + G6[4] * w1 + G6[5] * w2);
        }
      }
#pragma unroll
      for (int a = 0; a < N; ++a)
      {
        T acc = T(0);
#pragma unroll
        for (int q = 0; q < N; ++q)
          acc += Dk.d[q * N + a] * F0[q];
        // transposed index-1 part: sum_j D[j][b] F1(a, j, c) -- lane (b - m, c) holds F1(a, b - m, c)
        Y[a] = acc + dpp_contract_b<T>(tb, F1[a]) + dpp_contract_c<T>(tc, F2[a]);
      }
      (void)sA;
    }
    else if constexpr (GEOM == GEOM_DIAG)
    {
      constexpr int TS = tile_plane_stride<T, N>();
      const T g0 = gc_l[in.er * 7 + 0] * cf * wbc, g1 = gc_l[in.er * 7 + 3] * cf * wbc,
              g2 = gc_l[in.er * 7 + 5] * cf * wbc;   // w_b w_c of the lane's column, whichever two indices it spans
#pragma unroll
      for (int q = 0; q < N; ++q)
      {
        T acc = T(0), kr[N];
        dtab_row<T, N, 0>(Dk, q, kr);
#pragma unroll
        for (int i = 0; i < N; ++i)
          acc += kr[i] * X[i];
        Y[q] = g0 * acc;
      }
#pragma unroll
      for (int a = 0; a < N; ++a)
        sA[rtix<N, TS>(a, b, c)] = X[a];
      FUS_WAVE_SYNC();
      T Tb[N], Uc[N];
#pragma unroll
      for (int k = 0; k < N; ++k)
      {
        Tb[k] = sA[rtix<N, TS>(b, k, c)];
        Uc[k] = sA[rtix<N, TS>(b, c, k)];
      }
      FUS_WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < N; ++q)
      {
        T acc = T(0), kr[N];
        dtab_row<T, N, 0>(Dk, q, kr);
#pragma unroll
        for (int k = 0; k < N; ++k)
          acc += kr[k] * Tb[k];
        sA[rtix<N, TS>(b, q, c)] = g1 * acc;   // index-1 term at point (b, q, c)
      }
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
        Y[a] += sA[rtix<N, TS>(a, b, c)];
      FUS_WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < N; ++q)
      {
        T acc = T(0), kr[N];
        dtab_row<T, N, 0>(Dk, q, kr);
#pragma unroll
        for (int k = 0; k < N; ++k)
          acc += kr[k] * Uc[k];
        sA[rtix<N, TS>(b, c, q)] = g2 * acc;   // index-2 term at point (b, c, q)
      }
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
        Y[a] += sA[rtix<N, TS>(a, b, c)];
    }
#ifdef FUS_ABL_NOEXCH   // developer ablation (wrong results): the six contractions and the transform on the lane's own
                         // registers, no exchange through the LDS tile and none of its wave barriers
    else if constexpr (REMAP)
    {
      auto mul = [&](const T (&in)[N], T (&out)[N], bool tr)
      {
#pragma unroll
        for (int q = 0; q < N; ++q)
        {
          T acc = T(0), dr[N];
          if (tr)
            dtab_row<T, N, 1>(Dk, q, dr);
          else
            dtab_row<T, N, 0>(Dk, q, dr);
#pragma unroll
          for (int i = 0; i < N; ++i)
            acc += dr[i] * in[i];
          out[q] = acc;
        }
      };
      mul(X, F0, false), mul(X, F1, false), mul(X, F2, false);
#pragma unroll
      for (int a = 0; a < N; ++a)
      {
        F1[a] += T(0.5) * F0[a], F2[a] -= T(0.25) * F0[a];
        if (GEOM == GEOM_TRILINEAR)
          tri.transform(Dk.x[a], Dk.w[a] * wbc * cf, F0[a], F1[a], F2[a]);
        else
          F0[a] *= cf * w3[a];
      }
      T Y1[N], Y2[N];
      mul(F0, Y, true), mul(F1, Y1, true), mul(F2, Y2, true);
#pragma unroll
      for (int a = 0; a < N; ++a)
        Y[a] += Y1[a] + Y2[a];
      (void)sA;
    }
#endif
    else if constexpr (REMAP)
    {
      // plane stride of the tile: N^2, padded by one where the re-mapped accesses (lanes (b, c) at
      // b * TS + ...) would otherwise fall on the same LDS banks for every b (N = 8: 8-way, N = 4: 2-way)
      constexpr int TS = tile_plane_stride<T, N>();
#pragma unroll
      for (int q = 0; q < N; ++q)
      {
        T acc = T(0), dr[N];
        dtab_row<T, N, 0>(Dk, q, dr);
#pragma unroll
        for (int i = 0; i < N; ++i)
          acc += dr[i] * X[i];
        F0[q] = acc;
      }
#pragma unroll
      for (int a = 0; a < N; ++a)
        sA[rtix<N, TS>(a, b, c)] = X[a];
      FUS_WAVE_SYNC();
      constexpr bool SWZ = (N == 8) && FUS_SWZ8;   // index-2 contractions by ds_swizzle instead of the tile (A/B)
      // index-1 contractions on the matrix cores (mf4_contract_b): the trilinear kernel, where it measured +5 ... +7 %
      // (the general affine kernel and the diagonal-metric form were even: profiles/r03_experiments.md section 6)
      constexpr bool MF4 = (N == 8) && sizeof(T) == 8 && FUS_MF4 && GEOM == GEOM_TRILINEAR;
      T Tb[N], Uc[N], swc[8], swt[8];
      T mf_own = T(0), mf_swp = T(0), mt_own = T(0), mt_swp = T(0);
      if constexpr (MF4)
      {
        const int ln = b * N + c, be = b & 1, mi = 2 * (ln & 3) + be, mk = 2 * (ln >> 4);
        mf_own = D_l[mi * N + mk + be], mf_swp = D_l[mi * N + mk + 1 - be];       // D[q][b]
        mt_own = D_l[(mk + be) * N + mi], mt_swp = D_l[(mk + 1 - be) * N + mi];   // D^T
      }
      if constexpr (SWZ)
      {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          swc[j] = D_l[c * N + j], swt[j] = D_l[j * N + c];   // D[c][j], D[j][c]
      }
#pragma unroll
      for (int k = 0; k < N; ++k)
      {
        if constexpr (!MF4)
          Tb[k] = sA[rtix<N, TS>(b, k, c)];
        if constexpr (!SWZ)
          Uc[k] = sA[rtix<N, TS>(b, c, k)];
      }
      if constexpr (MF4)
      {
#pragma unroll
        for (int a = 0; a < N; ++a)
          F1[a] = mf4_contract_b(mf_own, mf_swp, X[a]);   // d/dX1 at the lane's own points (a, b, c)
      }
      else
      {
      FUS_WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < N; ++q)
      {
        T acc = T(0), dr[N];
        dtab_row<T, N, 0>(Dk, q, dr);
#pragma unroll
        for (int k = 0; k < N; ++k)
          acc += dr[k] * Tb[k];
        sA[rtix<N, TS>(b, q, c)] = acc;  // d/dX1 at point (b, q, c)
      }
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
        F1[a] = sA[rtix<N, TS>(a, b, c)];
      }
      if constexpr (SWZ)
      {
#pragma unroll
        for (int a = 0; a < N; ++a)
          F2[a] = swz8_contract_c<T>(swc, X[a]);   // d/dX2 at the lane's own points (a, b, c)
      }
      else
      {
      FUS_WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < N; ++q)
      {
        T acc = T(0), dr[N];
        dtab_row<T, N, 0>(Dk, q, dr);
#pragma unroll
        for (int k = 0; k < N; ++k)
          acc += dr[k] * Uc[k];
        sA[rtix<N, TS>(b, c, q)] = acc;  // d/dX2 at point (b, c, q)
      }
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
        F2[a] = sA[rtix<N, TS>(a, b, c)];
      }
      // stiffness::transform (spectral_op.hpp:113-130)
#pragma unroll
      for (int a = 0; a < N; ++a)
      {
#ifdef FUS_ABL_NOTRANS   // developer ablation (wrong results): no per-point Jacobian / transform arithmetic
        if (GEOM == GEOM_TRILINEAR)
        {
          const T sc = Dk.w[a] * wbc * cf * tri.j0[0];
          F0[a] *= sc, F1[a] *= sc, F2[a] *= sc;
        }
#else
        if (GEOM == GEOM_TRILINEAR)
          tri.transform(Dk.x[a], Dk.w[a] * wbc * cf, F0[a], F1[a], F2[a]);
#endif
        else
        {
          T G6[6];
#pragma unroll
          for (int gi = 0; gi < 6; ++gi)
          {
            const int v = gi * N + a;
            G6[gi] = (GEOM == GEOM_STREAM) ? in.g[v / VW][v % VW] : gc_l[in.er * 7 + gi] * w3[a];
          }
          const T w0 = F0[a], w1 = F1[a], w2 = F2[a];
          F0[a] = cf * (G6[0] * w0 + G6[1] * w1 + G6[2] * w2);
          F1[a] = cf * (G6[1] * w0 + G6[3] * w1 + G6[4] * w2);
          F2[a] = cf * (G6[2] * w0 + G6[4] * w1 + G6[5] * w2);
        }
      }
      // transposed contractions, the same way round
      if constexpr (MF4)
      {
#pragma unroll
        for (int a = 0; a < N; ++a)
        {
          T acc = mf4_contract_b(mt_own, mt_swp, F1[a]), dc[N];   // sum_q D[q][b] F1(a, q, c)
          dtab_row<T, N, 1>(Dk, a, dc);
#pragma unroll
          for (int q = 0; q < N; ++q)
            acc += dc[q] * F0[q];
          Y[a] = acc;
        }
      }
      else
      {
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
        sA[rtix<N, TS>(a, b, c)] = F1[a];
      FUS_WAVE_SYNC();
#pragma unroll
      for (int k = 0; k < N; ++k)
        Tb[k] = sA[rtix<N, TS>(b, k, c)];
      FUS_WAVE_SYNC();
#pragma unroll
      for (int j = 0; j < N; ++j)
      {
        T acc = T(0), dc[N];
        dtab_row<T, N, 1>(Dk, j, dc);
#pragma unroll
        for (int q = 0; q < N; ++q)
          acc += dc[q] * Tb[q];
        sA[rtix<N, TS>(b, j, c)] = acc;
      }
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
      {
        T acc = sA[rtix<N, TS>(a, b, c)], dc[N];
        dtab_row<T, N, 1>(Dk, a, dc);
#pragma unroll
        for (int q = 0; q < N; ++q)
          acc += dc[q] * F0[q];
        Y[a] = acc;
      }
      }
      if constexpr (SWZ)
      {
#pragma unroll
        for (int a = 0; a < N; ++a)
          Y[a] += swz8_contract_c<T>(swt, F2[a]);   // sum_j D[j][c] F2(a, b, j)
      }
      else
      {
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
        sA[rtix<N, TS>(a, b, c)] = F2[a];
      FUS_WAVE_SYNC();
#pragma unroll
      for (int k = 0; k < N; ++k)
        Uc[k] = sA[rtix<N, TS>(b, c, k)];
      FUS_WAVE_SYNC();
#pragma unroll
      for (int j = 0; j < N; ++j)
      {
        T acc = T(0), dc[N];
        dtab_row<T, N, 1>(Dk, j, dc);
#pragma unroll
        for (int q = 0; q < N; ++q)
          acc += dc[q] * Uc[q];
        sA[rtix<N, TS>(b, c, j)] = acc;
      }
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
        Y[a] += sA[rtix<N, TS>(a, b, c)];
      }
    }
    else
    {
    // derivative along tensor index 0: registers only (spectral_op.hpp:194-196)
#pragma unroll
    for (int q = 0; q < N; ++q)
    {
      T acc = T(0), dr[N];
      dtab_row<T, N, 0>(Dk, q, dr);
#pragma unroll
      for (int i = 0; i < N; ++i)
        acc += dr[i] * X[i];
      F0[q] = acc;
    }
#pragma unroll
    for (int a = 0; a < N; ++a)
      sA[a * N2 + p] = X[a];
    FUS_WAVE_SYNC();
    // derivatives along tensor indices 1 and 2 (spectral_op.hpp:199-210)
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      T f1 = T(0), f2 = T(0);
#pragma unroll
      for (int j = 0; j < N; ++j)
      {
        f1 += (DLDS ? D_l[b * N + j] : Drb[j]) * sA[a * N2 + j * N + c];
        f2 += (DLDS ? D_l[c * N + j] : Drc[j]) * sA[a * N2 + b * N + j];
      }
      F1[a] = f1;
      F2[a] = f2;
    }
    // stiffness::transform (spectral_op.hpp:113-130)
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      if (GEOM == GEOM_TRILINEAR)
      {
        tri.transform(Dk.x[a], Dk.w[a] * wbc * cf, F0[a], F1[a], F2[a]);
        continue;
      }
      T G6[6];
      {
#pragma unroll
        for (int gi = 0; gi < 6; ++gi)
        {
          const int v = gi * N + a;
          if (GEOM == GEOM_STREAM)
            G6[gi] = in.g[v / VW][v % VW];
          else
            G6[gi] = gc_l[in.er * 7 + gi] * w3[a];   // affine cell: G(q) = Gc w_q
        }
      }
      const T w0 = F0[a], w1 = F1[a], w2 = F2[a];
      F0[a] = cf * (G6[0] * w0 + G6[1] * w1 + G6[2] * w2);
      F1[a] = cf * (G6[1] * w0 + G6[3] * w1 + G6[4] * w2);
      F2[a] = cf * (G6[2] * w0 + G6[4] * w1 + G6[5] * w2);
    }
    // transposed contractions (spectral_op.hpp:222-238); the index-1 and index-2 parts reuse ONE
    // exchange tile one after the other (LDS footprint per wave halves -> more blocks per CU)
    FUS_WAVE_SYNC();
#pragma unroll
    for (int a = 0; a < N; ++a)
      sA[a * N2 + p] = F1[a];
    FUS_WAVE_SYNC();
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      T acc = T(0), dc[N];
      dtab_row<T, N, 1>(Dk, a, dc);
#pragma unroll
      for (int q = 0; q < N; ++q)
        acc += dc[q] * F0[q];
#pragma unroll
      for (int j = 0; j < N; ++j)
        acc += (DLDS ? D_l[j * N + b] : Dcb[j]) * sA[a * N2 + j * N + c];
      Y[a] = acc;
    }
    FUS_WAVE_SYNC();
#pragma unroll
    for (int a = 0; a < N; ++a)
      sA[a * N2 + p] = F2[a];
    FUS_WAVE_SYNC();
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      T acc = Y[a];
#pragma unroll
      for (int j = 0; j < N; ++j)
        acc += (DLDS ? D_l[j * N + c] : Dcc[j]) * sA[a * N2 + b * N + j];
      Y[a] = acc;
    }
    }
  }
  else
  {
    // mass::transform (spectral_op.hpp:19-26)
#pragma unroll
    for (int a = 0; a < N; ++a)
      Y[a] = cf * x_l[li[a]]
             * (GEOM == GEOM_STREAM ? in.dj[a]
                                    : (GEOM == GEOM_TRILINEAR ? tri.detw(Dk.x[a], Dk.w[a] * wbc) : gc_l[in.er * 7 + 6] * w3[a]));
  }
  // scatter-add into the block accumulator (spectral_op.hpp:240-241); elements of one round
  // share no dof and rounds are ordered -> deterministic
#if defined(FUS_ABL_NOGS) || defined(FUS_ABL_NOSCATTER)
  {
    T sum = T(0);
#pragma unroll
    for (int a = 0; a < N; ++a)
      sum += Y[a];
    __hip_atomic_fetch_add(&y_l[li[0]], (double)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }
#endif
#ifdef FUS_ABL_RTN32   // developer probe (wrong results): fp32 scatter with the RETURNING fp32 LDS atomic
  if constexpr (sizeof(T) == 4)
  {
    float seen = 0.0f;
#pragma unroll
    for (int a = 0; a < N; ++a)
      seen += __hip_atomic_fetch_add(reinterpret_cast<float*>(y_l) + li[a], (float)Y[a], __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
    if (seen == 1.2345e30f)
      y_l[0] = seen;
    return;
  }
#endif
#ifdef FUS_ABL_SCWRITE   // developer ablation (wrong results): the scatter as plain stores to the same addresses
#pragma unroll
  for (int a = 0; a < N; ++a)
    y_l[li[a]] = (double)Y[a];
  return;
#endif
#pragma unroll
  for (int a = 0; a < N; ++a)
  {
    if (ATOMIC)
      __hip_atomic_fetch_add(&y_l[li[a]], (double)Y[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      y_l[li[a]] += Y[a];
  }
}

// ---------------------------------------------------------------------------------------------
// Degrees 8-10 (N = 9, 10, 11; the reference's Qdegree map goes to P = 10, spectral_op.hpp:35-44): a tensor
// plane has more than 64 columns, so an element is worked on by TWO waves (lane pair index p = 0..127,
// columns p < N^2), the exchanges through the element's LDS tile are fenced by workgroup barriers instead
// of wave barriers, and nothing returns early (every wave of the workgroup meets every barrier; `on` guards
// the memory operations of lanes without an element or a column).  Per-cell geometry (affine / trilinear).
template <typename T, int N, int OP, int ATOMIC, int NF, int GEOM>
__device__ __forceinline__ void elem_compute_hi(int er_, bool has_col, const DTab<T, N>& Dk, const T* __restrict__ x_l,
                                                double* __restrict__ y_l, T* __restrict__ sA,
                                                const uint16_t* __restrict__ ldm_l, const T* __restrict__ cf_l,
                                                const T* __restrict__ x2_l, const T* __restrict__ cf2_l,
                                                const T* __restrict__ gc_l, const T* __restrict__ D_l,
                                                const T* __restrict__ w_l, const T* __restrict__ pt_l, int p, int b,
                                                int c, const T* __restrict__ geo = nullptr, int elem_off = 0)
{
  // GEOM_STREAM (any first- or second-order hexahedron, precompute.hpp:101-213): the lane's per-point factors are read
  // from HBM where the transform uses them (no register prefetch: correctness path of the Qdegree range, not a tuned one)
  constexpr int N2 = N * N, Nd = N * N * N;
  const bool on = er_ >= 0 && has_col;
  const int er = er_ >= 0 ? er_ : 0;
  const int pp = has_col ? p : 0, bb = has_col ? b : 0, cc = has_col ? c : 0;
  TriLane<T> tri;
  T wbc = T(0);
  if (GEOM == GEOM_TRILINEAR)
  {
    tri.init(gc_l + er * 21, pt_l[bb], pt_l[cc]);
    wbc = w_l[bb] * w_l[cc];
  }
  int li[N];
#pragma unroll
  for (int a = 0; a < N; ++a)
    li[a] = on ? (int)ldm_l[er * Nd + a * N2 + pp] : 0;
  const T cf = (NF == 2) ? T(1) : cf_l[er];
  T Y[N];
  if (OP == OP_STIFFNESS)
  {
    T X[N], F0[N], F1[N], F2[N];
    if (NF == 2)
    {
      const T c1 = cf_l[er], c2 = cf2_l[er];
#pragma unroll
      for (int a = 0; a < N; ++a)
        X[a] = c1 * x_l[li[a]] + c2 * x2_l[li[a]];
    }
    else
    {
#pragma unroll
      for (int a = 0; a < N; ++a)
        X[a] = x_l[li[a]];
    }
    // Two forms of the index-1 / index-2 contractions.  Re-mapped (as elem_compute; table rows by scalar loads): the
    // faster one on affine cells at p = 8 (fp64 5.9 -> 9.3e9 DOF-updates/s, fp32 +10 %) and at p = 9 in fp32 (+10 %).
    // Tile reads with the table in LDS: everywhere else (the re-mapped form measured 4-12 % slower there: its 14
    // workgroup barriers per element against 6).
    constexpr bool REMAP_HI = is_aff(GEOM) && (N == 9 || (N == 10 && sizeof(T) == 4));
    if constexpr (REMAP_HI)
    {
      // Re-mapped form of the index-1 / index-2 contractions (as elem_compute): lane pair (b, c) re-reads the tile as
      // (a' = b, c) with index 1 along its registers, then as (a' = b, b' = c) with index 2 along them; derivative-table
      // rows by scalar loads (dtab_row); every exchange fenced by a workgroup barrier (two waves share the element).
      constexpr int TS = N2 + 1;   // N planes of N^2 + 1: exactly the Nd + N entries of an element slot
      T Tb[N], Uc[N];
  #pragma unroll
      for (int q = 0; q < N; ++q)
      {
        T acc = T(0), dr[N];
        dtab_row<T, N, 0>(Dk, q, dr);
  #pragma unroll
        for (int i = 0; i < N; ++i)
          acc += dr[i] * X[i];
        F0[q] = acc;
      }
      if (on)
      {
  #pragma unroll
        for (int a = 0; a < N; ++a)
          sA[a * TS + pp] = X[a];
      }
      __syncthreads();
  #pragma unroll
      for (int k = 0; k < N; ++k)
      {
        Tb[k] = sA[bb * TS + k * N + cc];
        Uc[k] = sA[bb * TS + cc * N + k];
      }
      __syncthreads();
  #pragma unroll
      for (int q = 0; q < N; ++q)
      {
        T acc = T(0), dr[N];
        dtab_row<T, N, 0>(Dk, q, dr);
  #pragma unroll
        for (int k = 0; k < N; ++k)
          acc += dr[k] * Tb[k];
        if (on)
          sA[bb * TS + q * N + cc] = acc;   // d/dX1 at point (b, q, c)
      }
      __syncthreads();
  #pragma unroll
      for (int a = 0; a < N; ++a)
        F1[a] = sA[a * TS + pp];
      __syncthreads();
  #pragma unroll
      for (int q = 0; q < N; ++q)
      {
        T acc = T(0), dr[N];
        dtab_row<T, N, 0>(Dk, q, dr);
  #pragma unroll
        for (int k = 0; k < N; ++k)
          acc += dr[k] * Uc[k];
        if (on)
          sA[bb * TS + cc * N + q] = acc;   // d/dX2 at point (b, c, q)
      }
      __syncthreads();
  #pragma unroll
      for (int a = 0; a < N; ++a)
        F2[a] = sA[a * TS + pp];
      // stiffness::transform (spectral_op.hpp:113-130)
  #pragma unroll
      for (int a = 0; a < N; ++a)
      {
        if (GEOM == GEOM_TRILINEAR)
          tri.transform(pt_l[a], w_l[a] * wbc * cf, F0[a], F1[a], F2[a]);
        else
        {
          const T w3 = w_l[a] * w_l[bb] * w_l[cc];
          T G6[6];
  #pragma unroll
          for (int gi = 0; gi < 6; ++gi)
            G6[gi] = gc_l[er * 7 + gi] * w3;   // affine cell: G(q) = Gc w_q
          const T w0 = F0[a], w1 = F1[a], w2 = F2[a];
          F0[a] = cf * (G6[0] * w0 + G6[1] * w1 + G6[2] * w2);
          F1[a] = cf * (G6[1] * w0 + G6[3] * w1 + G6[4] * w2);
          F2[a] = cf * (G6[2] * w0 + G6[4] * w1 + G6[5] * w2);
        }
      }
      // transposed contractions (spectral_op.hpp:222-238), the same way round
      __syncthreads();
      if (on)
      {
  #pragma unroll
        for (int a = 0; a < N; ++a)
          sA[a * TS + pp] = F1[a];
      }
      __syncthreads();
  #pragma unroll
      for (int k = 0; k < N; ++k)
        Tb[k] = sA[bb * TS + k * N + cc];
      __syncthreads();
  #pragma unroll
      for (int j = 0; j < N; ++j)
      {
        T acc = T(0), dc[N];
        dtab_row<T, N, 1>(Dk, j, dc);
  #pragma unroll
        for (int q = 0; q < N; ++q)
          acc += dc[q] * Tb[q];
        if (on)
          sA[bb * TS + j * N + cc] = acc;
      }
      __syncthreads();
  #pragma unroll
      for (int a = 0; a < N; ++a)
      {
        T acc = sA[a * TS + pp], dc[N];
        dtab_row<T, N, 1>(Dk, a, dc);
  #pragma unroll
        for (int q = 0; q < N; ++q)
          acc += dc[q] * F0[q];
        Y[a] = acc;
      }
      __syncthreads();
      if (on)
      {
  #pragma unroll
        for (int a = 0; a < N; ++a)
          sA[a * TS + pp] = F2[a];
      }
      __syncthreads();
  #pragma unroll
      for (int k = 0; k < N; ++k)
        Uc[k] = sA[bb * TS + cc * N + k];
      __syncthreads();
  #pragma unroll
      for (int j = 0; j < N; ++j)
      {
        T acc = T(0), dc[N];
        dtab_row<T, N, 1>(Dk, j, dc);
  #pragma unroll
        for (int q = 0; q < N; ++q)
          acc += dc[q] * Uc[q];
        if (on)
          sA[bb * TS + cc * N + j] = acc;
      }
      __syncthreads();
  #pragma unroll
      for (int a = 0; a < N; ++a)
        Y[a] += sA[a * TS + pp];
      __syncthreads();   // the tile is free for the next element
    }
    else
    {
      // derivative along tensor index 0: registers only (spectral_op.hpp:194-196)
      // (scheduling fences keep the table reads of one output at a time in registers: with everything
      // unrolled and hoisted these kernels spill at the 256-register cap)
  #pragma unroll
      for (int q = 0; q < N; ++q)
      {
        T acc = T(0);
  #pragma unroll
        for (int i = 0; i < N; ++i)
          acc += D_l[q * N + i] * X[i];
        F0[q] = acc;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (on)
      {
  #pragma unroll
        for (int a = 0; a < N; ++a)
          sA[a * N2 + pp] = X[a];
      }
      __syncthreads();
      // derivatives along tensor indices 1 and 2 (spectral_op.hpp:199-210)
  #pragma unroll
      for (int a = 0; a < N; ++a)
        F1[a] = F2[a] = T(0);
  #pragma unroll 1
      for (int j = 0; j < N; ++j)
      {
        const T d1 = D_l[bb * N + j], d2 = D_l[cc * N + j];
  #pragma unroll
        for (int a = 0; a < N; ++a)
        {
          F1[a] += d1 * sA[a * N2 + j * N + cc];
          F2[a] += d2 * sA[a * N2 + bb * N + j];
        }
      }
      // stiffness::transform (spectral_op.hpp:113-130)
  #pragma unroll
      for (int a = 0; a < N; ++a)
      {
        if (GEOM == GEOM_TRILINEAR)
          tri.transform(pt_l[a], w_l[a] * wbc * cf, F0[a], F1[a], F2[a]);
        else
        {
          T G6[6];
          if constexpr (GEOM == GEOM_STREAM)
          {
            const T* __restrict__ Ge = geo + (int64_t)(elem_off + er) * (6 * Nd);
  #pragma unroll
            for (int gi = 0; gi < 6; ++gi)
              G6[gi] = on ? Ge[g_index<T, N>(gi * N + a, pp)] : T(0);
          }
          else
          {
          const T w3 = w_l[a] * w_l[bb] * w_l[cc];
  #pragma unroll
          for (int gi = 0; gi < 6; ++gi)
            G6[gi] = gc_l[er * 7 + gi] * w3;   // affine cell: G(q) = Gc w_q
          }
          const T w0 = F0[a], w1 = F1[a], w2 = F2[a];
          F0[a] = cf * (G6[0] * w0 + G6[1] * w1 + G6[2] * w2);
          F1[a] = cf * (G6[1] * w0 + G6[3] * w1 + G6[4] * w2);
          F2[a] = cf * (G6[2] * w0 + G6[4] * w1 + G6[5] * w2);
        }
      }
      // transposed contractions (spectral_op.hpp:222-238)
      __syncthreads();
      if (on)
      {
  #pragma unroll
        for (int a = 0; a < N; ++a)
          sA[a * N2 + pp] = F1[a];
      }
      __syncthreads();
  #pragma unroll
      for (int a = 0; a < N; ++a)
      {
        T acc = T(0);
  #pragma unroll
        for (int q = 0; q < N; ++q)
          acc += D_l[q * N + a] * F0[q];
        Y[a] = acc;
        __builtin_amdgcn_sched_barrier(0);
      }
  #pragma unroll 1
      for (int j = 0; j < N; ++j)
      {
        const T d1 = D_l[j * N + bb];
  #pragma unroll
        for (int a = 0; a < N; ++a)
          Y[a] += d1 * sA[a * N2 + j * N + cc];
      }
      __syncthreads();
      if (on)
      {
  #pragma unroll
        for (int a = 0; a < N; ++a)
          sA[a * N2 + pp] = F2[a];
      }
      __syncthreads();
  #pragma unroll 1
      for (int j = 0; j < N; ++j)
      {
        const T d2 = D_l[j * N + cc];
  #pragma unroll
        for (int a = 0; a < N; ++a)
          Y[a] += d2 * sA[a * N2 + bb * N + j];
      }
      __syncthreads();   // the tile is free for the next element
    }
  }
  else
  {
    // mass::transform (spectral_op.hpp:19-26)
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      T dj;
      if constexpr (GEOM == GEOM_STREAM)
        dj = on ? geo[(int64_t)(elem_off + er) * Nd + a * N2 + pp] : T(0);
      else
        dj = GEOM == GEOM_TRILINEAR ? tri.detw(pt_l[a], w_l[a] * wbc) : gc_l[er * 7 + 6] * (w_l[a] * w_l[bb] * w_l[cc]);
      Y[a] = cf * x_l[li[a]] * dj;
    }
  }
  if (on)
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      if (ATOMIC)
        __hip_atomic_fetch_add(&y_l[li[a]], (double)Y[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else
        y_l[li[a]] += Y[a];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Packed fp32 path (template parameter PK of k_block_op; degrees 5-7, per-cell geometry kernels).
// In fp32 the scalar kernels issue as many LDS and vector instructions per element as the fp64 ones while
// each moves half the bytes -- counters show fp32 p=6 bound by the LDS pipe (profiles/r02_experiments.md
// section 2).  Here a wave works on TWO elements at once: lane (b, c) holds float2 = (element A, element B)
// of its tensor column, every exchange through the tile is one 8-byte access for both, and the arithmetic
// is packed (v_pk_fma_f32), so LDS and vector instructions per element halve.  Same algorithm as the
// re-mapped form of elem_compute (spectral_op.hpp:194-238); the gather and scatter stay per element.
typedef float F2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ F2 pk_rcp(F2 x)
{
  F2 r = {__builtin_amdgcn_rcpf(x[0]), __builtin_amdgcn_rcpf(x[1])};
  r = (F2(1.0f) - x * r) * r + r;
  return r;
}
__device__ __forceinline__ F2 pk_abs(F2 x) { return F2{__builtin_fabsf(x[0]), __builtin_fabsf(x[1])}; }

template <int N, int ATOMIC, int NF, int GEOM>
__device__ __forceinline__ void elem_compute_pk(int e0, int e1, const DTab<float, N>& Dk, const float* __restrict__ x_l,
                                                double* __restrict__ y_l, F2* __restrict__ sA,
                                                const uint16_t* __restrict__ ldm_l, const float* __restrict__ cf_l,
                                                const float* __restrict__ x2_l, const float* __restrict__ cf2_l,
                                                const float* __restrict__ gc_l, const float* __restrict__ w_l,
                                                const float* __restrict__ pt_l, int p, int b, int c)
{
  static_assert(is_aff(GEOM) || GEOM == GEOM_TRILINEAR, "per-cell geometry kernels");
  constexpr int N2 = N * N, Nd = N * N * N;
  constexpr int TS = (N == 8 || N == 4) ? N2 + 1 : N2;   // plane stride of the tile of float2 entries (see elem_compute, REMAP)
  if (e0 < 0)
    return;
  const bool two = e1 >= 0;
  const int eb = two ? e1 : e0;   // a lone element is computed twice, scattered once
  int li0[N], li1[N];
#pragma unroll
  for (int a = 0; a < N; ++a)
  {
    li0[a] = ldm_l[e0 * Nd + a * N2 + p];
    li1[a] = ldm_l[eb * Nd + a * N2 + p];
  }
  const F2 cf = (NF == 2) ? F2(1.0f) : F2{cf_l[e0], cf_l[eb]};
  F2 X[N], F0[N], F1[N], F2v[N], Y[N];
  if (NF == 2)
  {
    const F2 c1 = {cf_l[e0], cf_l[eb]}, c2 = {cf2_l[e0], cf2_l[eb]};
#pragma unroll
    for (int a = 0; a < N; ++a)
      X[a] = c1 * F2{x_l[li0[a]], x_l[li1[a]]} + c2 * F2{x2_l[li0[a]], x2_l[li1[a]]};
  }
  else
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
      X[a] = F2{x_l[li0[a]], x_l[li1[a]]};
  }
  // index 0: registers
#pragma unroll
  for (int q = 0; q < N; ++q)
  {
    F2 acc = F2(0.0f);
#pragma unroll
    for (int i = 0; i < N; ++i)
      acc += Dk.d[q * N + i] * X[i];
    F0[q] = acc;
  }
#pragma unroll
  for (int a = 0; a < N; ++a)
    sA[a * TS + p] = X[a];
  FUS_WAVE_SYNC();
  F2 Tb[N], Uc[N];
#pragma unroll
  for (int k = 0; k < N; ++k)
  {
    Tb[k] = sA[b * TS + k * N + c];
    Uc[k] = sA[b * TS + c * N + k];
  }
  FUS_WAVE_SYNC();
#pragma unroll
  for (int q = 0; q < N; ++q)
  {
    F2 acc = F2(0.0f);
#pragma unroll
    for (int k = 0; k < N; ++k)
      acc += Dk.d[q * N + k] * Tb[k];
    sA[b * TS + q * N + c] = acc;  // d/dX1 at point (b, q, c)
  }
  FUS_WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < N; ++a)
    F1[a] = sA[a * TS + p];
  FUS_WAVE_SYNC();
#pragma unroll
  for (int q = 0; q < N; ++q)
  {
    F2 acc = F2(0.0f);
#pragma unroll
    for (int k = 0; k < N; ++k)
      acc += Dk.d[q * N + k] * Uc[k];
    sA[b * TS + c * N + q] = acc;  // d/dX2 at point (b, c, q)
  }
  FUS_WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < N; ++a)
    F2v[a] = sA[a * TS + p];
  // stiffness::transform (spectral_op.hpp:113-130), both elements at once
  if (GEOM == GEOM_TRILINEAR)
  {
    // the lane-constant parts of the two cells' Jacobians (TriLane, packed)
    const float pb = pt_l[b], pc = pt_l[c];
    const F2 wbc = F2(w_l[b] * w_l[c]) * cf;
    F2 j0[3], a1[3], d1[3], a2[3], d2[3];
    const float* ca = gc_l + e0 * 21;
    const float* cb = gc_l + eb * 21;
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
      const F2 c100 = {ca[i], cb[i]}, c010 = {ca[3 + i], cb[3 + i]}, c001 = {ca[6 + i], cb[6 + i]},
               c110 = {ca[9 + i], cb[9 + i]}, c101 = {ca[12 + i], cb[12 + i]}, c011 = {ca[15 + i], cb[15 + i]},
               c111 = {ca[18 + i], cb[18 + i]};
      d2[i] = c101 + pb * c111;
      j0[i] = (c100 + pb * c110) + pc * d2[i];
      a1[i] = c010 + pc * c011;
      d1[i] = c110 + pc * c111;
      a2[i] = c001 + pb * c011;
    }
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      const float pa = Dk.x[a];
      F2 j1[3], j2[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
      {
        j1[i] = a1[i] + pa * d1[i];
        j2[i] = a2[i] + pa * d2[i];
      }
      const F2 r0[3] = {j1[1] * j2[2] - j1[2] * j2[1], j1[2] * j2[0] - j1[0] * j2[2], j1[0] * j2[1] - j1[1] * j2[0]};
      const F2 r1[3] = {j2[1] * j0[2] - j2[2] * j0[1], j2[2] * j0[0] - j2[0] * j0[2], j2[0] * j0[1] - j2[1] * j0[0]};
      const F2 r2[3] = {j0[1] * j1[2] - j0[2] * j1[1], j0[2] * j1[0] - j0[0] * j1[2], j0[0] * j1[1] - j0[1] * j1[0]};
      const F2 det = j0[0] * r0[0] + j0[1] * r0[1] + j0[2] * r0[2];
      const F2 sc = (Dk.w[a] * wbc) * pk_rcp(pk_abs(det));
      const F2 f0 = F0[a], f1 = F1[a], f2 = F2v[a];
      F2 t[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        t[i] = sc * (f0 * r0[i] + f1 * r1[i] + f2 * r2[i]);
      F0[a] = r0[0] * t[0] + r0[1] * t[1] + r0[2] * t[2];
      F1[a] = r1[0] * t[0] + r1[1] * t[1] + r1[2] * t[2];
      F2v[a] = r2[0] * t[0] + r2[1] * t[1] + r2[2] * t[2];
    }
  }
  else
  {
    const float wbc = w_l[b] * w_l[c];
    F2 Gc[6];
#pragma unroll
    for (int gi = 0; gi < 6; ++gi)
      Gc[gi] = F2{gc_l[e0 * 7 + gi], gc_l[eb * 7 + gi]} * cf;
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      const float w3 = w_l[a] * wbc;   // affine cell: G(q) = Gc w_q
      const F2 w0 = F0[a] * w3, w1 = F1[a] * w3, w2 = F2v[a] * w3;
      F0[a] = Gc[0] * w0 + Gc[1] * w1 + Gc[2] * w2;
      F1[a] = Gc[1] * w0 + Gc[3] * w1 + Gc[4] * w2;
      F2v[a] = Gc[2] * w0 + Gc[4] * w1 + Gc[5] * w2;
    }
  }
  // transposed contractions, the same way round
  FUS_WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < N; ++a)
    sA[a * TS + p] = F1[a];
  FUS_WAVE_SYNC();
#pragma unroll
  for (int k = 0; k < N; ++k)
    Tb[k] = sA[b * TS + k * N + c];
  FUS_WAVE_SYNC();
#pragma unroll
  for (int j = 0; j < N; ++j)
  {
    F2 acc = F2(0.0f);
#pragma unroll
    for (int q = 0; q < N; ++q)
      acc += Dk.d[q * N + j] * Tb[q];
    sA[b * TS + j * N + c] = acc;
  }
  FUS_WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < N; ++a)
  {
    F2 acc = sA[a * TS + p];
#pragma unroll
    for (int q = 0; q < N; ++q)
      acc += Dk.d[q * N + a] * F0[q];
    Y[a] = acc;
  }
  FUS_WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < N; ++a)
    sA[a * TS + p] = F2v[a];
  FUS_WAVE_SYNC();
#pragma unroll
  for (int k = 0; k < N; ++k)
    Uc[k] = sA[b * TS + c * N + k];
  FUS_WAVE_SYNC();
#pragma unroll
  for (int j = 0; j < N; ++j)
  {
    F2 acc = F2(0.0f);
#pragma unroll
    for (int q = 0; q < N; ++q)
      acc += Dk.d[q * N + j] * Uc[q];
    sA[b * TS + c * N + j] = acc;
  }
  FUS_WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < N; ++a)
  {
    Y[a] += sA[a * TS + p];
    if (ATOMIC)
    {
      __hip_atomic_fetch_add(&y_l[li0[a]], (double)Y[a][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (two)
        __hip_atomic_fetch_add(&y_l[li1[a]], (double)Y[a][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    else
    {
      y_l[li0[a]] += Y[a][0];
      if (two)
        y_l[li1[a]] += Y[a][1];
    }
  }
  FUS_WAVE_SYNC();   // the tile is free for the next pair
}

// ---------------------------------------------------------------------------------------------
// MFMA contraction path (degrees 6 and 7, per-cell geometry kernels; template parameter MF of k_block_op).
// At N = 7, 8 one element fills a wave, and each of the index-1 / index-2 contractions of an element is an
// (N x N) . (N x N^2) product -- the reference's contract<T, N, N, N, N, bool>
// (cpp/fenicsx-sf/common/sum_factorisation.hpp:70-86).  It runs here on the matrix cores as 16 x 16 x 4
// tiles (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32; N = 7 is padded to 8 with zero table entries):
//   A operand (lane l: row l & 15, k = l >> 4): the derivative table, D[q][k] (rows q >= N are zero),
//                                               or its transpose for the second half of the operator;
//   B operand (lane l: k = l >> 4, column l & 15): 16 columns = two tensor planes a = 2 t + (h >> 3) x 8
//                                               values of the free index, read from the element's LDS tile;
//   two k-steps (k = 4 s + g) per tile, four tiles (t) per direction: 8 MFMAs per direction and element,
//   half of each tile's 16 result rows are padding (N <= 8 < 16).
// The results go back through the tile to the lane-per-column layout the transform and the scatter use
// (8 reads + 8 writes + 8 reads per lane and direction, like the re-mapped vector form); the index-0
// contraction stays on the vector ALUs (registers only).  f64 MFMA issues at the vector FMA rate on gfx950,
// so the gain, if any, is the vector ALU freed for the geometry recomputation running beside it.
template <typename T>
struct MfmaOp;
template <>
struct MfmaOp<double>
{
  typedef double V4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ V4 mma(double a, double b, V4 c)
  {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of the f64 form: column = lane & 15, row = (lane >> 4) + 4 reg
  static __device__ __forceinline__ int row(int g, int r) { return g + 4 * r; }
};
template <>
struct MfmaOp<float>
{
  typedef float V4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ V4 mma(float a, float b, V4 c)
  {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of the f32 form: column = lane & 15, row = 4 (lane >> 4) + reg
  static __device__ __forceinline__ int row(int g, int r) { return 4 * g + r; }
};

// Position of tensor entry (a, b, c) in the element's LDS tile.  N = 8: the fastest index is XOR-swizzled
// with bits of the other two, c ^ (2 (b >> 2) | 4 (a & 1)), which keeps the lane-per-column accesses
// (a fixed, lanes (b, c)) conflict-free and spreads the operand reads of the index-2 contraction (lanes
// (b, c = k) for two planes a) over the banks -- 4-way conflicts otherwise (57 % of the LDS cycles of the
// unswizzled form were bank conflicts: profiles/r02_mfma.md).
template <int N>
__device__ __forceinline__ int mfma_tix(int a, int b, int c)
{
  if constexpr (N == 8)
    return a * 64 + b * 8 + (c ^ (((b >> 2) << 1) | ((a & 1) << 2)));
  else
    return (a * N + b) * N + c;
}

// One directional contraction of the element in the tile: out[.., q, ..] = sum_k A[q][k] in[.., k, ..];
// DIR = 1: the contracted index is tensor index 1 (free index 2 on the columns), DIR = 2: the reverse.
// Bm[t][s]: this lane's B operands (read from the tile by the caller before it is overwritten).
template <typename T, int N, int DIR>
__device__ __forceinline__ void mfma_tile_contract(const T (&Am)[2], const T (&Bm)[4][2], T* __restrict__ sA, int g,
                                                   int ha, int hc)
{
  typedef typename MfmaOp<T>::V4 V4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
  {
    V4 acc = {T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int s = 0; s < 2; ++s)
      acc = MfmaOp<T>::mma(Am[s], Bm[t][s], acc);
    const int a = 2 * t + ha;
#pragma unroll
    for (int r = 0; r < 4; ++r)
    {
      const int q = MfmaOp<T>::row(g, r);
      if (q < N && a < N && hc < N)
        sA[DIR == 1 ? mfma_tix<N>(a, q, hc) : mfma_tix<N>(a, hc, q)] = acc[r];
    }
  }
}

template <typename T, int N, int DIR>
__device__ __forceinline__ void mfma_tile_operands(T (&Bm)[4][2], const T* __restrict__ sA, int g, int ha, int hc)
{
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s)
    {
      const int a = 2 * t + ha, k = 4 * s + g;
      // padding (N = 7) reads a valid, finite tile entry; its table entry is zero
      const bool ok = a < N && k < N && hc < N;
      Bm[t][s] = sA[ok ? (DIR == 1 ? mfma_tix<N>(a, k, hc) : mfma_tix<N>(a, hc, k)) : 0];
    }
}

template <typename T, int N, int ATOMIC, int NF, int GEOM>
__device__ __forceinline__ void elem_compute_mfma(const ElemIn<T, N, OP_STIFFNESS, GEOM>& in, const DTab<T, N>& Dk,
                                                  const T* __restrict__ x_l, double* __restrict__ y_l,
                                                  T* __restrict__ sA, const uint16_t* __restrict__ ldm_l,
                                                  const T* __restrict__ cf_l, const T* __restrict__ x2_l,
                                                  const T* __restrict__ cf2_l, const T* __restrict__ gc_l,
                                                  const T (&w3)[N], const T* __restrict__ D_l, T wbc, T pb, T pc,
                                                  int p, int lane)
{
  static_assert(N == 7 || N == 8, "one element per wave, table padded to 8");
  static_assert(is_aff(GEOM) || GEOM == GEOM_TRILINEAR, "per-cell geometry kernels");
  constexpr int N2 = N * N, Nd = N * N * N;
  // one element per wave (EPW = 1): lane 0 always holds it; lanes >= N^2 (N = 7) carry no tensor column
  // but take part in the matrix instructions
  const int er = __builtin_amdgcn_readfirstlane(in.er);
  if (er < 0)
    return;
  const bool on = in.er >= 0;
  const int g = lane >> 4, h = lane & 15, ha = h >> 3, hc = h & 7;
  const int pb_i = on ? p / N : 0, pc_i = on ? p - (p / N) * N : 0;   // the lane's tensor column (b, c)
  T Af[2], At[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
  {
    const int k = 4 * s + g;
    const bool ok = h < N && k < N;
    Af[s] = ok ? D_l[h * N + k] : T(0);  // A[row q = h][k]     = D[q][k]
    At[s] = ok ? D_l[k * N + h] : T(0);  // A[row j = h][k = q] = D[q][j]
  }
  TriLane<T> tri;
  if (GEOM == GEOM_TRILINEAR)
    tri.init(gc_l + er * 21, pb, pc);
  int li[N];
#pragma unroll
  for (int a = 0; a < N; ++a)
    li[a] = on ? (int)ldm_l[er * Nd + a * N2 + p] : 0;
  const T cf = (NF == 2) ? T(1) : cf_l[er];
  T X[N], F0[N], F1[N], F2[N], Y[N];
  if (NF == 2)
  {
    const T c1 = cf_l[er], c2 = cf2_l[er];
#pragma unroll
    for (int a = 0; a < N; ++a)
      X[a] = c1 * x_l[li[a]] + c2 * x2_l[li[a]];
  }
  else
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
      X[a] = x_l[li[a]];
  }
  // derivative along tensor index 0: registers only (spectral_op.hpp:194-196)
#pragma unroll
  for (int q = 0; q < N; ++q)
  {
    T acc = T(0), dr[N];
    dtab_row<T, N, 0>(Dk, q, dr);
#pragma unroll
    for (int i = 0; i < N; ++i)
      acc += dr[i] * X[i];
    F0[q] = acc;
  }
  if (on)
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
      sA[mfma_tix<N>(a, pb_i, pc_i)] = X[a];
  }
  FUS_WAVE_SYNC();
  // derivatives along tensor indices 1 and 2 (spectral_op.hpp:199-210) on the matrix cores
  T B1[4][2], B2[4][2];
  mfma_tile_operands<T, N, 1>(B1, sA, g, ha, hc);   // contracted index 1 (stride N), free index 2
  mfma_tile_operands<T, N, 2>(B2, sA, g, ha, hc);   // contracted index 2 (stride 1), free index 1
  FUS_WAVE_SYNC();
  mfma_tile_contract<T, N, 1>(Af, B1, sA, g, ha, hc);
  FUS_WAVE_SYNC();
  if (on)
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
      F1[a] = sA[mfma_tix<N>(a, pb_i, pc_i)];
  }
  FUS_WAVE_SYNC();
  mfma_tile_contract<T, N, 2>(Af, B2, sA, g, ha, hc);
  FUS_WAVE_SYNC();
  if (on)
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
      F2[a] = sA[mfma_tix<N>(a, pb_i, pc_i)];
  }
  // stiffness::transform (spectral_op.hpp:113-130)
  if (on)
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      if (GEOM == GEOM_TRILINEAR)
        tri.transform(Dk.x[a], Dk.w[a] * wbc * cf, F0[a], F1[a], F2[a]);
      else
      {
        T G6[6];
#pragma unroll
        for (int gi = 0; gi < 6; ++gi)
          G6[gi] = gc_l[er * 7 + gi] * w3[a];
        const T w0 = F0[a], w1 = F1[a], w2 = F2[a];
        F0[a] = cf * (G6[0] * w0 + G6[1] * w1 + G6[2] * w2);
        F1[a] = cf * (G6[1] * w0 + G6[3] * w1 + G6[4] * w2);
        F2[a] = cf * (G6[2] * w0 + G6[4] * w1 + G6[5] * w2);
      }
    }
  }
  // transposed contractions (spectral_op.hpp:222-238)
  FUS_WAVE_SYNC();
  if (on)
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
      sA[mfma_tix<N>(a, pb_i, pc_i)] = F1[a];
  }
  FUS_WAVE_SYNC();
  mfma_tile_operands<T, N, 1>(B1, sA, g, ha, hc);
  FUS_WAVE_SYNC();
  mfma_tile_contract<T, N, 1>(At, B1, sA, g, ha, hc);
  FUS_WAVE_SYNC();
  if (on)
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      T acc = sA[mfma_tix<N>(a, pb_i, pc_i)], dc[N];
      dtab_row<T, N, 1>(Dk, a, dc);
#pragma unroll
      for (int q = 0; q < N; ++q)
        acc += dc[q] * F0[q];
      Y[a] = acc;
    }
  }
  FUS_WAVE_SYNC();
  if (on)
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
      sA[mfma_tix<N>(a, pb_i, pc_i)] = F2[a];
  }
  FUS_WAVE_SYNC();
  mfma_tile_operands<T, N, 2>(B2, sA, g, ha, hc);
  FUS_WAVE_SYNC();
  mfma_tile_contract<T, N, 2>(At, B2, sA, g, ha, hc);
  FUS_WAVE_SYNC();
  if (on)
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      Y[a] += sA[mfma_tix<N>(a, pb_i, pc_i)];
      if (ATOMIC)
        __hip_atomic_fetch_add(&y_l[li[a]], (double)Y[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else
        y_l[li[a]] += Y[a];
    }
  }
}

// The stiffness pass of elem_compute in two halves, for the single-register-set schedule of the
// degrees 5 and 6 (k_block_op, PF1): elem_stiff_fwd ends with the transform -- the last reader of in.g --,
// the kernel then requests the next element's geometry into the same registers, and elem_stiff_bwd
// does the transposed contractions and the scatter.  Same arithmetic in the same order as
// elem_compute.
// DL = 1 (degree 7): the lane's rows / columns of the derivative table are read from LDS (D_l) at the
// start of each half instead of living in registers for the whole kernel.
template <typename T, int N, int NF, int DL = 0>
__device__ __forceinline__ bool elem_stiff_fwd(const ElemIn<T, N, OP_STIFFNESS, GEOM_STREAM>& in,
                                               const DTab<T, N>& Dk, const T (&Drb_)[N], const T (&Drc_)[N],
                                               const T* __restrict__ D_l,
                                               const T* __restrict__ x_l, T* __restrict__ sA,
                                               const uint16_t* __restrict__ ldm_l,
                                               const T* __restrict__ cf_l, const T* __restrict__ x2_l,
                                               const T* __restrict__ cf2_l, int p, int b, int c,
                                               int (&li)[N], T (&F0)[N], T (&F1)[N], T (&F2)[N])
{
  constexpr int N2 = N * N, Nd = N * N * N;
  constexpr int VW = GLoad<T, N>::VW;
  if (in.er < 0)
    return false;
#pragma unroll
  for (int a = 0; a < N; ++a)
    li[a] = ldm_l[in.er * Nd + a * N2 + p];
  const T cf = (NF == 2) ? T(1) : cf_l[in.er];
  T X[N];
  if (NF == 2)
  {
    const T c1 = cf_l[in.er], c2 = cf2_l[in.er];
#pragma unroll
    for (int a = 0; a < N; ++a)
      X[a] = c1 * x_l[li[a]] + c2 * x2_l[li[a]];
  }
  else
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
      X[a] = x_l[li[a]];
  }
#pragma unroll
  for (int q = 0; q < N; ++q)
  {
    T acc = T(0), dr[N];
    dtab_row<T, N, 0>(Dk, q, dr);
#pragma unroll
    for (int i = 0; i < N; ++i)
      acc += dr[i] * X[i];
    F0[q] = acc;
  }
#pragma unroll
  for (int a = 0; a < N; ++a)
    sA[a * N2 + p] = X[a];
  T Drb[N], Drc[N];
#pragma unroll
  for (int j = 0; j < N; ++j)
  {
    Drb[j] = DL ? D_l[b * N + j] : Drb_[j];
    Drc[j] = DL ? D_l[c * N + j] : Drc_[j];
  }
  FUS_WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < N; ++a)
  {
    T f1 = T(0), f2 = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j)
    {
      f1 += Drb[j] * sA[a * N2 + j * N + c];
      f2 += Drc[j] * sA[a * N2 + b * N + j];
    }
    F1[a] = f1;
    F2[a] = f2;
  }
#pragma unroll
  for (int a = 0; a < N; ++a)
  {
    T G6[6];
#pragma unroll
    for (int gi = 0; gi < 6; ++gi)
    {
      const int v = gi * N + a;
      G6[gi] = in.g[v / VW][v % VW];
    }
    const T w0 = F0[a], w1 = F1[a], w2 = F2[a];
    F0[a] = cf * (G6[0] * w0 + G6[1] * w1 + G6[2] * w2);
    F1[a] = cf * (G6[1] * w0 + G6[3] * w1 + G6[4] * w2);
    F2[a] = cf * (G6[2] * w0 + G6[4] * w1 + G6[5] * w2);
  }
  return true;
}

template <typename T, int N, int DL = 0, int ATOMIC = 1>
__device__ __forceinline__ void elem_stiff_bwd(const DTab<T, N>& Dk, const T (&Dcb_)[N], const T (&Dcc_)[N],
                                               const T* __restrict__ D_l,
                                               double* __restrict__ y_l, T* __restrict__ sA, int p, int b,
                                               int c, const int (&li)[N], const T (&F0)[N],
                                               const T (&F1)[N], const T (&F2)[N])
{
  constexpr int N2 = N * N;
  T Y[N];
  T Dcb[N], Dcc[N];
#pragma unroll
  for (int j = 0; j < N; ++j)
  {
    Dcb[j] = DL ? D_l[j * N + b] : Dcb_[j];
    Dcc[j] = DL ? D_l[j * N + c] : Dcc_[j];
  }
  FUS_WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < N; ++a)
    sA[a * N2 + p] = F1[a];
  FUS_WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < N; ++a)
  {
    T acc = T(0), dc[N];
    dtab_row<T, N, 1>(Dk, a, dc);
#pragma unroll
    for (int q = 0; q < N; ++q)
      acc += dc[q] * F0[q];
#pragma unroll
    for (int j = 0; j < N; ++j)
      acc += Dcb[j] * sA[a * N2 + j * N + c];
    Y[a] = acc;
  }
  FUS_WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < N; ++a)
    sA[a * N2 + p] = F2[a];
  FUS_WAVE_SYNC();
#pragma unroll
  for (int a = 0; a < N; ++a)
  {
    T acc = Y[a];
#pragma unroll
    for (int j = 0; j < N; ++j)
      acc += Dcc[j] * sA[a * N2 + b * N + j];
    Y[a] = acc;
  }
#pragma unroll
  for (int a = 0; a < N; ++a)
  {
    if (ATOMIC)
      __hip_atomic_fetch_add(&y_l[li[a]], (double)Y[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      y_l[li[a]] += Y[a];   // conflict-free rounds: elements of one round share no dof
  }
}

// Single geometry register set (see elem_stiff_fwd): degrees 5-7, streamed geometry (fp64 and fp32, LDS-atomic
// and conflict-free-round accumulation).  With two sets those kernels need 290-380 registers, i.e. one wave per
// SIMD and one block per CU; with one set they fit 256: two waves per SIMD, two blocks per CU.
#define FUS_PF1(T, P, OP, ATOMIC, GEOM, TD)                                                         \
  (((P) >= 5 && (P) <= 7 && (OP) == OP_STIFFNESS && (GEOM) == GEOM_STREAM && (TD) == 3) ? 1 : 0)

// Fused RK4 stage update of ONE dof s whose right-hand side sum `acc` (= b) is complete.
// kv = b * minv  (Linear.hpp:212-221), ku = vn (f0, :171-174).
//   STAGE 0 : vn == v0, un == u0, u_ == u0, v_ == v0 (aliases are not read twice)
//             u_ = u0 + bdt*v0 ; v_ = v0 + bdt*kv ; un' = u0 + adt*v0 ; vn' = v0 + adt*kv
//   STAGE 1,2: u_ += bdt*vn ; v_ += bdt*kv ; un' = u0 + adt*vn ; vn' = v0 + adt*kv
//   STAGE 3 : u0 = u_ + bdt*vn ; v0 = v_ + bdt*kv        (next step's state, no copies)
// adt = dt*a_{i+1}, bdt = dt*b_i (Linear.hpp:282-294).
//
// Model vectors are read / written once per stage: non-temporal, so that the partial slab (written
// just before by k_block_op) stays cache resident.
template <typename T>
struct LeanRK
{
  T b0dt, pdt, third;   // see StageArgs
};

template <typename T, int STAGE>
__device__ __forceinline__ T stage_update_dof(int64_t s, T acc, const T* __restrict__ minv,
                                                 T* __restrict__ vn, T* __restrict__ un,
                                                 T* __restrict__ u0, T* __restrict__ v0,
                                                 T* __restrict__ u_, T* __restrict__ v_, T adt, T bdt,
                                                 const T* __restrict__ m0, const T* __restrict__ mn1,
                                                 const LeanRK<T> R)
{
#define FUS_LD(p) __builtin_nontemporal_load(&(p)[s])
#define FUS_ST(p, val) __builtin_nontemporal_store((val), &(p)[s])
  T kv;
  if (mn1)  // Westervelt (see StageArgs): stage inputs u_n, v_n are u0, v0 at the first stage
  {
    const T us = stage_is_first(STAGE) ? u0[s] : un[s], vs = stage_is_first(STAGE) ? v0[s] : vn[s];
    kv = (acc - mn1[s] * vs * vs) / (m0[s] + mn1[s] * us);
  }
  else
    kv = acc * FUS_LD(minv);
  T vnext;  // the velocity the NEXT stage starts from: vn' (stages 0-2) or the new v0 (stage 3)
  if (STAGE == 0)
  {
    const T u = FUS_LD(u0), v = FUS_LD(v0);
    FUS_ST(u_, v * bdt + u);
    FUS_ST(v_, kv * bdt + v);
    FUS_ST(un, v * adt + u);
    vnext = kv * adt + v;
    FUS_ST(vn, vnext);
  }
  else if (STAGE == 4)   // stage 0: the input is u0 itself
  {
    const T u = FUS_LD(u0), v = FUS_LD(v0);
    FUS_ST(un, v * adt + u);
    vnext = kv * adt + v;
    FUS_ST(v_, vnext);   // V_1
  }
  else if (STAGE == 3)
  {
    FUS_ST(u0, FUS_LD(vn) * bdt + FUS_LD(u_));
    vnext = kv * bdt + FUS_LD(v_);
    FUS_ST(v0, vnext);
  }
  else if (STAGE == 5)   // stage 1: un = u0 + pdt v0, vn = V_1
  {
    const T w = FUS_LD(vn), v = FUS_LD(v0), xs = FUS_LD(un);
    FUS_ST(un, xs + (w * adt - v * R.pdt));
    vnext = kv * adt + v;
    FUS_ST(v_, vnext);   // V_2
  }
  else if (STAGE == 6)   // stage 2: un = u0 + pdt V_1, vn = V_2, u_ = V_1
  {
    const T w = FUS_LD(vn), v = FUS_LD(v0), va = FUS_LD(u_), xs = FUS_LD(un);
    FUS_ST(un, xs + (w * adt - va * R.pdt));
    vnext = kv * adt + v;
    FUS_ST(v_, vnext);   // V_3
  }
  else if (STAGE == 7)   // stage 3: un = u0 + pdt V_2, vn = V_3, u_ = V_1, v_ = V_2; the new state
  {
    const T w = FUS_LD(vn), v = FUS_LD(v0), va = FUS_LD(u_), vb = FUS_LD(v_), xs = FUS_LD(un);
    FUS_ST(u0, (xs - vb * R.pdt) + ((va + vb) * T(2) + (v + w)) * R.b0dt);
    vnext = kv * bdt + ((vb * T(2) + (va + w)) - v) * R.third;
    FUS_ST(v0, vnext);
  }
  else
  {
    const T w = FUS_LD(vn);
    FUS_ST(u_, w * bdt + FUS_LD(u_));
    FUS_ST(v_, kv * bdt + FUS_LD(v_));
    FUS_ST(un, w * adt + FUS_LD(u0));
    vnext = kv * adt + FUS_LD(v0);
    FUS_ST(vn, vnext);
  }
#undef FUS_LD
#undef FUS_ST
  return vnext;
}

// The block kernel's only argument (read in the kernel through the kernarg segment pointer, see FUS_KARGS).
template <typename T, int N>
struct KArgs
{
  BlockArgs A;
  DTab<T, N> Dk;
  const T* Dg;
  const T* geo;
  const T* coef;
  const T* x;
  T* bvec;
  T* partial;
  StageArgs<T> S;
};

template <typename T, int N>
__device__ __forceinline__ void load_stage_args(const KArgs<T, N> __attribute__((address_space(4))) * q, StageArgs<T>& S)
{
  S.minv = q->S.minv;
  S.vn = q->S.vn, S.un = q->S.un, S.u0 = q->S.u0, S.v0 = q->S.v0, S.u_ = q->S.u_, S.v_ = q->S.v_;
  S.adt = q->S.adt, S.bdt = q->S.bdt, S.gval = q->S.gval;
  S.b0dt = q->S.b0dt, S.pdt = q->S.pdt, S.third = q->S.third;
  S.blk_bnd_off = q->S.blk_bnd_off, S.bnd_idx = q->S.bnd_idx;
  S.bnd_src = q->S.bnd_src, S.bnd_abs = q->S.bnd_abs;
  S.x2 = q->S.x2, S.coef2 = q->S.coef2, S.bnd_src2 = q->S.bnd_src2, S.dgval = q->S.dgval;
  S.m0 = q->S.m0, S.mn1 = q->S.mn1;
}

// Block operator:  bvec[interior dofs of block] = (A x)[...],  partial[(block, shared slot)] =
// this block's contribution to a shared dof.  x, bvec in internal numbering.
// geo = G (6*Nd per element, per-lane vector layout) for OP_STIFFNESS, detJw (Nd per element,
// tensor order) for OP_MASS.  coef: one scalar per internal element.
//
// STAGE >= 0 (stiffness only) fuses the RK4 stage update into the epilogue (Linear.hpp:274-294 +
// :203-221): for the block's interior dofs the sum in LDS is complete, so b never goes to HBM --
// boundary terms are added in LDS, kv = b * minv, and u_, v_, un', vn' (or the new u0, v0 at stage 3)
// are written straight from here.  Shared dofs still leave as partial sums.
// Launch bound: up to 8 waves per workgroup for P <= 4; the higher degrees are limited to 4 waves and
// either use the whole 512-entry register file with one wave per SIMD (two geometry register sets:
// fp32 and deterministic streamed variants, affine at degree 7), or -- FUS_PF1, fp64 streamed
// geometry -- keep one set and fit two waves per SIMD, or -- the per-cell geometry modes, which hold
// no geometry registers -- are compiled for FUS_TRI_WAVES(P) waves per SIMD with the lane's
// derivative-table rows read from LDS where used.
// TD = 2: the same block machinery for quadrilateral elements (Nd = N^2, GEOM_STREAM only).
template <typename T, int P, int OP, int ATOMIC, int STAGE, int NF, int GEOM, int TD = 3, int MF = 0, int PK = 0>
__global__ void __launch_bounds__((P <= 4) ? 512 : FUS_MID_THREADS, (P >= 8) ? FUS_HI_WAVES : (P <= 4 && is_aff(GEOM))
                                                            ? 4
                                                            : ((GEOM == GEOM_TRILINEAR || (is_aff(GEOM) && P <= 6)) ? FUS_TRI_WAVES_T(T, P) : (FUS_PF1(T, P, OP, ATOMIC, GEOM, TD) ? 2 : 1)))
k_block_op(const KArgs<T, P + 1> kernel_args)
{
  (void)kernel_args;
  static_assert(TD == 3 || GEOM == GEOM_STREAM, "quadrilaterals use the streamed geometry");
  typedef KArgs<T, P + 1> KArgsT;
  typedef DTab<T, P + 1> DTabT;
  static_assert(offsetof(KArgsT, Dk) == sizeof(BlockArgs) && offsetof(DTabT, dt) == sizeof(T) * ((P + 1) * (P + 1) + 2 * (P + 1)),
                "dtab_row reads the table at these kernarg offsets");
  constexpr int N = P + 1, N2 = N * N, Nd = (TD == 3) ? N * N * N : N * N;
  constexpr int EPW = (64 / N2) > 0 ? (64 / N2) : 1;
  // lanes that work on one element slot group: a wave, or two waves where a tensor plane has more than
  // 64 columns (degrees 8-10, elem_compute_hi)
  constexpr int LPE = (N2 <= 64) ? 64 : 128;
  constexpr int SLOT = tile_slot_entries<T, N, TD>();   // entries of one element's exchange tile
  static_assert(N2 <= 128, "degrees up to 10");
  static_assert(LPE == 64 || !MF, "degrees 8-10: no matrix-core variants");
  // packed fp32 (elem_compute_pk): a wave works on two elements at once
  static_assert(!PK || (sizeof(T) == 4 && EPW == 1 && LPE == 64 && TD == 3 && OP == OP_STIFFNESS && ATOMIC && !MF
                        && GEOM != GEOM_STREAM),
                "packed path: fp32 stiffness, degrees 5-7, per-cell geometry, LDS-atomic accumulation");
  constexpr int EPS = PK ? 2 : 1;   // elements per lane group and trip

  // Kernel arguments are read from the kernarg segment where they are used, through a pointer that is
  // made opaque again at the start of every phase (FUS_KARGS): a workgroup that walks several blocks
  // would otherwise keep all ~150 scalar registers' worth of arguments live around the block loop and
  // spill them.
  typedef const KArgs<T, N> __attribute__((address_space(4))) * KP;
#define FUS_KARGS(q)                                                                               \
  KP q = (KP)__builtin_amdgcn_kernarg_segment_ptr();                             \
  asm volatile("" : "+s"(q))
  constexpr int GCS = geom_cell_stride(GEOM);               // per-cell geometry numbers (7 / 21 / 0)
  // FUS_ACC: the block accumulator y_l is fp64 for every scalar type.  On gfx950 the LDS atomic ds_add_f32 is
  // an order of magnitude slower than ds_add_f64 / ds_add_u32 / a plain store (fp32 p=6: a third of the kernel,
  // profiles/r03_experiments.md section 3), so the fp32 kernels scatter-add with ds_add_f64 into an fp64 image and
  // round once when the sums leave LDS (which also makes the fp32 element sum exact to fp32 rounding).
  constexpr int AW = 8 / (int)sizeof(T);                    // accumulator entry in units of T
  typedef double A2 __attribute__((ext_vector_type(2)));
  extern __shared__ __align__(16) unsigned char smem_raw[];
  // LDS carve, re-derived from the kernel arguments in every phase (nothing of it is carried around the
  // block loop in scalar registers)
#define FUS_PHASE_LDS(q)                                                                           \
  const int lds_nloc = q->A.lds_nloc, lds_nelem = q->A.lds_nelem, nwaves = q->A.waves;             \
  const int slots = (nwaves * 64 / LPE) * EPW * EPS;                                               \
  T* x_l = reinterpret_cast<T*>(smem_raw);                                                         \
  double* y_l = reinterpret_cast<double*>(x_l + lds_nloc); /* accumulator: fp64 for every T (FUS_ACC) */ \
  T* x2_l = reinterpret_cast<T*>(y_l + lds_nloc); /* second input (NF == 2 only) */                \
  T* scratch = x2_l + (NF == 2 ? lds_nloc : 0);                                                    \
  T* D_l = scratch + (size_t)slots * SLOT; /* derivative table (tiles: SLOT entries each) */       \
  T* cf_l = D_l + N2; /* per-element coefficient(s) */                                             \
  T* cf2_l = cf_l + lds_nelem;                                                                     \
  T* gc_l = cf2_l + (NF == 2 ? lds_nelem : 0); /* affine: 6 G + 1 detJ; trilinear: 21 map coefficients */ \
  T* w_l = gc_l + GCS * lds_nelem; /* 1-D weights, 1-D points: 8 slots each (12: degrees 8-10) */  \
  T* pt_l = w_l + (N <= 8 ? 8 : 12);                                                               \
  uint16_t* ldm_l = reinterpret_cast<uint16_t*>(w_l + (GCS ? (N <= 8 ? 16 : 24) : 0)); /* 16-B aligned */ \
  int16_t* rt_l = reinterpret_cast<int16_t*>(ldm_l + (size_t)lds_nelem * Nd);                      \
  (void)y_l, (void)x2_l, (void)scratch, (void)D_l, (void)cf_l, (void)cf2_l, (void)gc_l, (void)w_l, (void)pt_l,      \
      (void)ldm_l, (void)rt_l, (void)slots

  const int tid0 = threadIdx.x, nthr = blockDim.x;
  // the thread index, opaque per phase like the lane coordinates (FUS_TID): address arithmetic of one phase
  // must not be hoisted out of the block loop and stay live through the others
#define FUS_TID()                                                                                  \
  int tid = tid0;                                                                                  \
  asm volatile("" : "+v"(tid))
  // lane coordinates: recomputed per block from a value the compiler cannot trace (FUS_LANE_COORDS), so that
  // the address arithmetic of the element trips is not hoisted out of the block loop of a walking
  // workgroup and kept live (and spilled) across its prologue / epilogue phases
#define FUS_LANE_COORDS(q)                                                                         \
  FUS_TID();                                                                                       \
  const int lane = tid % LPE, wave = tid / LPE; /* a wave, or a wave pair (degrees 8-10) */        \
  const int s = lane / N2, p = lane - s * N2;                                                      \
  const int b = p / N, c = p - b * N;                                                              \
  const bool active = s < EPW;                                                                     \
  const int myslot = (wave * EPW + (active ? s : 0)) * EPS;                                        \
  /* exchange tiles follow x_l, y_l (, x2_l); one per element slot */                              \
  T* sA = reinterpret_cast<T*>(smem_raw) + (size_t)((NF == 2 ? 2 : 1) + AW) * q->A.lds_nloc         \
          + (size_t)myslot * SLOT;                                                                 \
  T* sB = sA

  // A workgroup walks the blocks blk, blk + gridDim.x, ... of the launch's range (a launch with one
  // workgroup per block walks one).  While it writes out block k (the epilogue's loads and stores) the
  // global loads of block k + 1's prologue are already in flight, so the two memory-latency phases of
  // consecutive blocks overlap instead of following each other.
  struct Meta
  {
    ShapeDev sh;
    int elem_off, int_off;
    int64_t sh_off;
  };
  auto meta_of = [&](KP q, int bk) -> Meta
  {
    Meta M;
    M.sh = q->A.shapes[q->A.blk_shape[bk]];
    M.elem_off = q->A.blk_elem_off[bk];
    M.int_off = q->A.blk_int_off[bk];
    M.sh_off = q->A.blk_sh_off[bk];
    return M;
  };
  typedef T V2 __attribute__((ext_vector_type(2)));
  typedef uint32_t U4 __attribute__((ext_vector_type(4)));
  // per thread: interior 16-B vectors, shared dofs, dofmap 16-B vectors of the first batch (what a block
  // of the default size needs; larger blocks finish in the plain loops of p_commit).  Kept small: these
  // registers are live across the epilogue of the previous block when a workgroup walks several blocks.
  // (p=7: 1098 interior vectors per 8-element block and 256 threads -> five; degrees 8-10: up to 1688 interior vectors and
  // 1538 shared dofs -> seven each: FUS_UI_HI.  With four, the remainder cost the prologue a second memory round trip.)
  constexpr int UI = (P <= 3) ? 3 : ((P == 4) ? 2 : (P <= 6 ? 4 : (P == 7 ? FUS_UI_P7 : FUS_UI_HI))),
                US = (P <= 3) ? 4 : ((P == 4) ? 3 : (P <= 7 ? 5 : FUS_UI_HI)),
                UL = (P <= 3) ? 2 : ((P == 4) ? 1 : 4);
  // ---- prologue: stage the block's dof values, local dofmaps and coefficients in LDS, clear the
  // accumulator.  Every load that does not depend on another is issued first (one round trip), the
  // gather of the shared dofs (it needs their indices) second (p_load); the LDS stores follow in p_commit.
  // Blocks larger than the first batches finish in plain loops there. ----
  struct PLoad
  {
    V2 xi[UI], xi2[UI];
    T xs[US], xs2[US];
    U4 lq[UL];
    T cfv, cf2v, gcv, gcv2, dgv, xtail, xtail2;
  };
  // indices of the block's shared dofs (the gather of their values depends on them: requested a phase early)
  auto p_idx = [&](KP q, const Meta& M, bool enable, int (&gi)[US]) __attribute__((always_inline))
  {
    FUS_TID();
    const int nsh = M.sh.nloc - M.sh.nint;
    const int32_t* gix = q->A.sh_gidx + M.sh_off;
#pragma unroll
    for (int u = 0; u < US; ++u)
      gi[u] = (enable && tid + u * nthr < nsh) ? gix[tid + u * nthr] : -1;
  };
  auto p_load = [&](KP q, const Meta& M, PLoad& L, bool with_tables, bool enable, const int (&gi)[US])
      __attribute__((always_inline))
  {
    FUS_TID();
    const ShapeDev& sh = M.sh;
    const T* __restrict__ x = q->x;
    const T* __restrict__ x2 = (NF == 2) ? q->S.x2 : x;
    const T* __restrict__ coef = q->coef;
    const T* __restrict__ coef2 = (NF == 2) ? q->S.coef2 : coef;
    const T* __restrict__ geo = q->geo;
    const T* __restrict__ Dg = q->Dg;
    const V2* xg = reinterpret_cast<const V2*>(x + M.int_off);  // int_off is a multiple of 16
    const V2* xg2 = reinterpret_cast<const V2*>(x2 + M.int_off);
    const int nvec = sh.nint >> 1;
    const int nsh = sh.nloc - sh.nint;
    const int32_t* gix = q->A.sh_gidx + M.sh_off;
    const int n16 = (sh.nelem * Nd * 2 + 15) >> 4;  // ldm_off is a multiple of 8 entries
    const U4* lsrc = reinterpret_cast<const U4*>(q->A.ldm + sh.ldm_off);
    const int ngc = sh.nelem * GCS;
    // round trip 1
    // (every field is assigned unconditionally: a register that is only written under a condition
    // would have to stay live across the element trips of a walking workgroup)
#pragma unroll
    for (int u = 0; u < UI; ++u)
    {
      const bool ok = enable && tid + u * nthr < nvec;
      L.xi[u] = ok ? xg[tid + u * nthr] : V2(T(0));
      L.xi2[u] = (NF == 2 && ok) ? xg2[tid + u * nthr] : V2(T(0));
    }
#pragma unroll
    for (int u = 0; u < UL; ++u)
      L.lq[u] = (enable && tid + u * nthr < n16) ? lsrc[tid + u * nthr] : U4(0u);
    L.cfv = (enable && tid < sh.nelem) ? coef[M.elem_off + tid] : T(0);
    L.cf2v = (NF == 2 && enable && tid < sh.nelem) ? coef2[M.elem_off + tid] : T(0);
    L.gcv = (enable && tid < ngc) ? geo[(int64_t)M.elem_off * GCS + tid] : T(0);
    L.gcv2 = (GCS > 7 && enable && tid + nthr < ngc) ? geo[(int64_t)M.elem_off * GCS + tid + nthr] : T(0);
    L.dgv = (with_tables && enable && tid < N2 + 2 * N) ? Dg[tid] : T(0);  // derivative table, 1-D weights, 1-D points
    L.xtail = (enable && tid == 0 && (sh.nint & 1)) ? x[M.int_off + sh.nint - 1] : T(0);
    L.xtail2 = (NF == 2 && enable && tid == 0 && (sh.nint & 1)) ? x2[M.int_off + sh.nint - 1] : T(0);
    // the shared dofs' values (their indices came with p_idx)
#pragma unroll
    for (int u = 0; u < US; ++u)
    {
      L.xs[u] = gi[u] >= 0 ? x[gi[u]] : T(0);
      L.xs2[u] = (NF == 2 && gi[u] >= 0) ? x2[gi[u]] : T(0);
    }
  };
  auto p_commit = [&](KP q, const Meta& M, const PLoad& L, bool with_tables) __attribute__((always_inline))
  {
    FUS_PHASE_LDS(q);
    FUS_TID();
    const ShapeDev& sh = M.sh;
    const T* __restrict__ x = q->x;
    const T* __restrict__ x2 = (NF == 2) ? q->S.x2 : x;
    const T* __restrict__ coef = q->coef;
    const T* __restrict__ coef2 = (NF == 2) ? q->S.coef2 : coef;
    const T* __restrict__ geo = q->geo;
    const T* __restrict__ Dg = q->Dg;
    const V2* xg = reinterpret_cast<const V2*>(x + M.int_off);
    const V2* xg2 = reinterpret_cast<const V2*>(x2 + M.int_off);
    const int nvec = sh.nint >> 1;
    const int nsh = sh.nloc - sh.nint;
    const int32_t* gix = q->A.sh_gidx + M.sh_off;
    const int n16 = (sh.nelem * Nd * 2 + 15) >> 4;
    const U4* lsrc = reinterpret_cast<const U4*>(q->A.ldm + sh.ldm_off);
    const int ngc = sh.nelem * GCS;
    // LDS stores; the accumulator is cleared (16 bytes per store)
    for (int i = tid; i < (sh.nloc + 1) / 2; i += nthr)
      reinterpret_cast<A2*>(y_l)[i] = A2(0.0);
#pragma unroll
    for (int u = 0; u < UI; ++u)
      if (tid + u * nthr < nvec)
      {
        reinterpret_cast<V2*>(x_l)[tid + u * nthr] = L.xi[u];
        if (NF == 2)
          reinterpret_cast<V2*>(x2_l)[tid + u * nthr] = L.xi2[u];
      }
#pragma unroll
    for (int u = 0; u < UL; ++u)
      if (tid + u * nthr < n16)
        reinterpret_cast<U4*>(ldm_l)[tid + u * nthr] = L.lq[u];
    if (tid < sh.nelem)
    {
      cf_l[tid] = L.cfv;
      if (NF == 2)
        cf2_l[tid] = L.cf2v;
    }
    if (tid < ngc)
      gc_l[tid] = L.gcv;
    if (GCS > 7 && tid + nthr < ngc)
      gc_l[tid + nthr] = L.gcv2;
    if (with_tables)
    {
      if (tid < N2)
        D_l[tid] = L.dgv;
      if (GCS && tid >= N2 && tid < N2 + N)
        w_l[tid - N2] = L.dgv;
      if (GCS && tid >= N2 + N && tid < N2 + 2 * N)
        pt_l[tid - N2 - N] = L.dgv;
      // (a one-wave workgroup at the higher degrees has fewer threads than table entries)
      for (int k = tid + nthr; k < N2 + 2 * N; k += nthr)
      {
        const T v = Dg[k];
        if (k < N2)
          D_l[k] = v;
        else if (GCS && k < N2 + N)
          w_l[k - N2] = v;
        else if (GCS)
          pt_l[k - N2 - N] = v;
      }
    }
    if (tid == 0 && (sh.nint & 1))
    {
      x_l[sh.nint - 1] = L.xtail;
      if (NF == 2)
        x2_l[sh.nint - 1] = L.xtail2;
    }
#pragma unroll
    for (int u = 0; u < US; ++u)
      if (tid + u * nthr < nsh)
      {
        x_l[sh.nint + tid + u * nthr] = L.xs[u];
        if (NF == 2)
          x2_l[sh.nint + tid + u * nthr] = L.xs2[u];
      }
    // leftovers of large blocks (higher degrees), again with the loads of a batch ahead of its stores
    constexpr int UB = 8;
    for (int base = tid + UI * nthr; base < nvec; base += nthr * UB)
    {
      V2 v[UB], v2[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (base + u * nthr < nvec)
        {
          v[u] = xg[base + u * nthr];
          if (NF == 2)
            v2[u] = xg2[base + u * nthr];
        }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (base + u * nthr < nvec)
        {
          reinterpret_cast<V2*>(x_l)[base + u * nthr] = v[u];
          if (NF == 2)
            reinterpret_cast<V2*>(x2_l)[base + u * nthr] = v2[u];
        }
    }
    for (int base = tid + US * nthr; base < nsh; base += nthr * UB)
    {
      int g[UB];
      T v[UB], v2[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u)
        g[u] = (base + u * nthr < nsh) ? gix[base + u * nthr] : 0;
#pragma unroll
      for (int u = 0; u < UB; ++u)
      {
        v[u] = x[g[u]];
        if (NF == 2)
          v2[u] = x2[g[u]];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (base + u * nthr < nsh)
        {
          x_l[sh.nint + base + u * nthr] = v[u];
          if (NF == 2)
            x2_l[sh.nint + base + u * nthr] = v2[u];
        }
    }
    for (int base = tid + UL * nthr; base < n16; base += nthr * UB)
    {
      U4 q[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (base + u * nthr < n16)
          q[u] = lsrc[base + u * nthr];
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (base + u * nthr < n16)
          reinterpret_cast<U4*>(ldm_l)[base + u * nthr] = q[u];
    }
    for (int k = tid + nthr; k < sh.nelem; k += nthr)
    {
      cf_l[k] = coef[M.elem_off + k];
      if (NF == 2)
        cf2_l[k] = coef2[M.elem_off + k];
    }
    for (int k = tid + (GCS > 7 ? 2 : 1) * nthr; k < ngc; k += nthr)
      gc_l[k] = geo[(int64_t)M.elem_off * GCS + k];
    // the round table (deterministic mode only) -> LDS, so the per-round element lookup is not a
    // global load that would drain the geometry prefetch queue (vmcnt retires in order)
    if (!ATOMIC)
      for (int k = tid; k < sh.nrounds * slots; k += nthr)
        rt_l[k] = q->A.rounds[sh.rounds_off + k];
  };

  FUS_KARGS(q0);
  const int blk_end = q0->A.blk_begin + q0->A.blk_count;
  const int blk_stride = gridDim.x;
  // Developer probe (FUS_XCD_REMAP = 1; off).  Workgroups are dealt round-robin to the 8 XCDs (observed placement, a
  // speed matter only); the remap gives workgroup i position i / 8 of the (i % 8)-th contiguous eighth of the block
  // range, so that blocks next to each other in the layout -- which gather the same shared dofs -- run on one XCD and
  // meet in its L2 (bijective for any grid size).  It LOSES here (p=4 fp64 -1 %, streamed G -3.6 %, fp32 +0.9 %;
  // profiles/r03_experiments.md section 15): the kernel streams far more than it re-reads (54 % of the shared-dof
  // gathers, 3 % of its bytes, are the only re-use), and with the default placement the eight XCDs walk the same
  // region of every vector together instead of eight distant ones.
  int wg = blockIdx.x;
  if (FUS_XCD_REMAP)
  {
    const int G = gridDim.x, q8 = G >> 3, r8 = G & 7, xc = wg & 7;
    wg = xc * q8 + (xc < r8 ? xc : r8) + (wg >> 3);
  }
  int blk = wg + q0->A.blk_begin;
  FUS_STAMP(blk, 0);
  // first trip's geometry is requested before the block's dof values are staged
  ElemIn<T, N, OP, GEOM, TD> inA, inB;
  auto first_fetch = [&](KP q, const Meta& Mm) __attribute__((always_inline))
  {
    FUS_LANE_COORDS(q);
    (void)b, (void)c, (void)sA, (void)sB;
    const int slots = (q->A.waves * 64 / LPE) * EPW * EPS;
    const int nt0 = ATOMIC ? (Mm.sh.nelem + slots - 1) / slots : Mm.sh.nrounds;
    int e0 = -1;
    if (active && nt0 > 0)
      e0 = ATOMIC ? (myslot < Mm.sh.nelem ? myslot : -1) : (int)q->A.rounds[Mm.sh.rounds_off + myslot];
    elem_fetch<T, N, OP, GEOM, TD>(inA, e0, q->geo, Mm.elem_off, p);
  };
  PLoad L;
  {
    const Meta M0 = meta_of(q0, blk);
    first_fetch(q0, M0);
    int gi0[US];
    p_idx(q0, M0, true, gi0);
    p_load(q0, M0, L, true, true, gi0);
  }
  bool first = true;
  for (;;)
  {
  // ---- phase 1: this block's staged data -> LDS, then the element trips ----
  FUS_KARGS(qc);
  FUS_PHASE_LDS(qc);
  FUS_LANE_COORDS(qc);
  const Meta M = meta_of(qc, blk);
  const ShapeDev& sh = M.sh;
  const int elem_off = M.elem_off;
  // element of this lane group in trip r: conflict-free round table (deterministic mode) or simply
  // the next `slots` elements (atomic mode: no ordering constraint between waves)
  const int ntrips = ATOMIC ? (sh.nelem + slots - 1) / slots : sh.nrounds;
  auto elem_of = [&](int r) -> int
  {
    if (!active || r >= ntrips)
      return -1;
    if (ATOMIC)
      return (r * slots + myslot < sh.nelem) ? r * slots + myslot : -1;
    return (int)rt_l[r * slots + myslot];
  };
#ifdef FUS_PROBE_PFREE
  // developer probe (wrong results): a walked block's prologue costs nothing but clearing the accumulator -- the
  // upper bound of any scheme that has the next block's data in LDS before its trips start (LDS-DMA staging)
  if (first)
    p_commit(qc, M, L, true);
  else
  {
    FUS_TID();
    for (int i = tid; i < (M.sh.nloc + 1) / 2; i += nthr)
      reinterpret_cast<A2*>(y_l)[i] = A2(0.0);
  }
#else
  p_commit(qc, M, L, first);
#endif
  // the next block of this workgroup: its shared-dof indices are requested now and ride through the trips
  // (degrees <= 4: a few registers), so that all of its prologue loads are independent in phase 2
  const int blk_next = blk + blk_stride;
  const bool has_next = blk_next < blk_end;
  // (measured: requesting them here costs more than it saves -- the barrier ahead of the trips waits for
  // them, 0.2525 -> 0.2620 ms at config 2 -- so the indices are read in phase 2)
  constexpr bool IDX_EARLY = false;
  int gi_n[US];
  if constexpr (IDX_EARLY)
  {
    Meta Mi{};
    if (has_next)
      Mi = meta_of(qc, blk_next);
    p_idx(qc, Mi, has_next, gi_n);
  }
  __syncthreads();
  FUS_STAMP(blk, 1);
#ifdef FUS_STAGGER   // developer probe: the second half of an 8-wave workgroup starts its trips FUS_STAGGER x 64 cycles later
  if (tid0 >= 256)
    __builtin_amdgcn_s_sleep(FUS_STAGGER);
#endif
  // lane-dependent rows/columns of the derivative table, quadrature weights and points of the lane's
  // tensor column (the tables stay in LDS from the first block on)
  T Drb[N], Drc[N], Dcb[N], Dcc[N], w3[N];
  T wbc, pb, pc;
  {
#pragma unroll
    for (int j = 0; j < N; ++j)
    {
      constexpr bool inreg = (OP == OP_STIFFNESS) && !(FUS_PF1(T, P, OP, ATOMIC, GEOM, TD) && P >= 6)
                             && GEOM != GEOM_TRILINEAR && GEOM != GEOM_DIAG && !(is_aff(GEOM) && (P == 5 || P == 6)) && !MF;
      Drb[j] = inreg ? D_l[b * N + j] : T(0);
      Drc[j] = inreg ? D_l[c * N + j] : T(0);
      Dcb[j] = inreg ? D_l[j * N + b] : T(0);
      Dcc[j] = inreg ? D_l[j * N + c] : T(0);
    }
#pragma unroll
    for (int a = 0; a < N; ++a)
      w3[a] = (GEOM == GEOM_AFFINE) ? w_l[a] * w_l[b] * w_l[c] : T(0);  // w_q = w_a w_b w_c of this lane's points
    wbc = (GEOM == GEOM_TRILINEAR || GEOM == GEOM_DIAG) ? w_l[b] * w_l[c] : T(0);
    pb = (GEOM == GEOM_TRILINEAR) ? pt_l[b] : T(0), pc = (GEOM == GEOM_TRILINEAR) ? pt_l[c] : T(0);
  }

  // ---- trips, two per iteration: while one register set is consumed the other is in flight ----
  FUS_KARGS(qt);
  // derivative table (N = 8 in fp64: read row by row where used, dtab_row), 1-D weights and points -> scalar
  // registers for the trips
  DTab<T, N> Dk;
  if constexpr (!dtab_by_rows<T, N>())
  {
#pragma unroll
    for (int i = 0; i < N * N; ++i)
      Dk.d[i] = qt->Dk.d[i];
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
    Dk.w[i] = qt->Dk.w[i], Dk.x[i] = qt->Dk.x[i];
  const T* __restrict__ geo = qt->geo;
#define FUS_ELEM_COMPUTE(in)                                                                       \
  do                                                                                               \
  {                                                                                                \
    if constexpr (LPE == 128 && TD == 2)                                                           \
      elem_compute2d<T, N, OP, ATOMIC, NF, true>(in, Drb, Drc, Dcb, Dcc, x_l, y_l, sA, ldm_l,      \
                                                 cf_l, x2_l, cf2_l, p, b, c, s == 0);              \
    else if constexpr (LPE == 128)                                                                 \
      elem_compute_hi<T, N, OP, ATOMIC, NF, GEOM>(in.er, s == 0, Dk, x_l, y_l, sA, ldm_l, cf_l,    \
                                                  x2_l, cf2_l, gc_l, D_l, w_l, pt_l, p, b, c, geo, \
                                                  elem_off);                                       \
    else if constexpr (MF && TD == 3 && OP == OP_STIFFNESS)                                        \
      elem_compute_mfma<T, N, ATOMIC, NF, GEOM>(in, Dk, x_l, y_l, sA, ldm_l, cf_l, x2_l, cf2_l,    \
                                                gc_l, w3, D_l, wbc, pb, pc, p, lane);              \
    else if constexpr (TD == 3)                                                                    \
      elem_compute<T, N, OP, ATOMIC, NF, GEOM>(in, Dk, Drb, Drc, Dcb, Dcc, x_l, y_l, sA, sB,       \
                                               ldm_l, cf_l, x2_l, cf2_l, gc_l, w3, D_l, wbc, pb,   \
                                               pc, p, b, c);                                       \
    else                                                                                           \
      elem_compute2d<T, N, OP, ATOMIC, NF>(in, Drb, Drc, Dcb, Dcc, x_l, y_l, sA, ldm_l, cf_l,      \
                                           x2_l, cf2_l, p, b, c);                                  \
  } while (0)
  if constexpr (PK)
  {
    for (int r = 0; r < ntrips; ++r)
    {
      const int i0 = r * slots + myslot;
      const int e0 = (active && i0 < sh.nelem) ? i0 : -1, e1 = (active && i0 + 1 < sh.nelem) ? i0 + 1 : -1;
      elem_compute_pk<N, ATOMIC, NF, GEOM>(e0, e1, Dk, x_l, y_l, reinterpret_cast<F2*>(sA), ldm_l, cf_l, x2_l, cf2_l,
                                           gc_l, w_l, pt_l, p, b, c);
    }
  }
  else if constexpr (FUS_PF1(T, P, OP, ATOMIC, GEOM, TD))
  {
    for (int r = 0; r < ntrips; ++r)
    {
      int li[N];
      T F0[N], F1[N], F2[N];
      constexpr int DL = (P >= 6) ? 1 : 0;
      const bool act = elem_stiff_fwd<T, N, NF, DL>(inA, Dk, Drb, Drc, D_l, x_l, sA, ldm_l, cf_l, x2_l, cf2_l, p, b,
                                                    c, li, F0, F1, F2);
      elem_fetch<T, N, OP, GEOM, TD>(inA, elem_of(r + 1), geo, elem_off, p);
      if (act)
        elem_stiff_bwd<T, N, DL, ATOMIC>(Dk, Dcb, Dcc, D_l, y_l, sA, p, b, c, li, F0, F1, F2);
      if (!ATOMIC && nwaves > 1)
        __syncthreads();   // rounds are ordered
    }
  }
  else
  for (int r = 0; r < ntrips; r += 2)
  {
    const bool has1 = r + 1 < ntrips;
    elem_fetch<T, N, OP, GEOM, TD>(inB, elem_of(r + 1), geo, elem_off, p);
    FUS_ELEM_COMPUTE(inA);
    if (r == 0)
      FUS_STAMP(blk, 5);
    if (r == 2)
      FUS_STAMP(blk, 7);
    if (!ATOMIC && nwaves > 1)
      __syncthreads();
    elem_fetch<T, N, OP, GEOM, TD>(inA, elem_of(r + 2), geo, elem_off, p);
    if (has1)
    {
      FUS_ELEM_COMPUTE(inB);
      if (r == 0)
        FUS_STAMP(blk, 6);
      if (!ATOMIC && nwaves > 1)
        __syncthreads();
    }
  }
#undef FUS_ELEM_COMPUTE
  {
  // this block's epilogue operands (arguments, LDS carve and block data re-derived: see FUS_KARGS)
  FUS_KARGS(qe);
  FUS_PHASE_LDS(qe);
  FUS_TID();
  const Meta M = meta_of(qe, blk);
  const ShapeDev& sh = M.sh;
  const int int_off = M.int_off;
  const int64_t sh_off = M.sh_off;
  StageArgs<T> S;
  load_stage_args<T, N>(qe, S);
  T* __restrict__ bvec = qe->bvec;
  T* __restrict__ partial = qe->partial;
  // where this block's partial sums go: requested now, used after the interior dofs (the stores would otherwise
  // wait for a memory round trip of their own at the very end of the block)
  const int32_t* __restrict__ pp = qe->A.sh_ppos + sh_off;
  const int nsh = sh.nloc - sh.nint;
  int ppv[US];
#pragma unroll
  for (int u = 0; u < US; ++u)
    ppv[u] = (tid + u * nthr < nsh) ? pp[tid + u * nthr] : -1;

  // ---- epilogue: each dof written once ----
  const int nvec = sh.nint >> 1;
    // fused stage update on the contiguous interior range (16-byte accesses)
    // degrees <= 4, one operator input: two interior ranges per pass, the loads of both in flight before
    // the first store (+1-3 % on the per-cell geometry paths; with the second input's extra operands
    // -- Westervelt -- it measured 5 % slower, and the higher degrees were not measured)
    constexpr bool EPI2 = (P <= 4) && (NF == 1);
    struct Epi
    {
      bool on;
      int i, o;
      V2 bv, mi, w, a0, b0, au, av, m1, m0v, us;
    };
    // part 1: the operands from HBM (issued BEFORE the barrier that ends the trips: they do not depend on the block's
    // sums, and their round trip then runs under the barrier wait and the boundary terms); part 2: the sums / stage input
    // from LDS; part 0: both
    auto epi_load = [&](int i, Epi& E, int part) __attribute__((always_inline))
    {
      if (part != 2)
      {
      // (every field is assigned on every path -- zeros where the range has ended -- so that none of them
      // looks live across the block loop to the register allocator)
      E.on = i < nvec;
      E.i = i, E.o = int_off + 2 * i;
      E.bv = E.mi = E.w = E.a0 = E.b0 = E.au = E.av = E.m1 = E.m0v = E.us = V2(T(0));
#ifdef FUS_PROBE_EFREE
      // developer probe (wrong results): the stage update's operands cost no memory round trip
      auto ld = [&](const T* ptr) -> V2 { return E.on ? V2(T(1)) + V2(T((uintptr_t)ptr & 7)) : V2(T(0)); };
#else
      auto ld = [&](const T* ptr) -> V2
      { return E.on ? __builtin_nontemporal_load(reinterpret_cast<const V2*>(ptr + E.o)) : V2(T(0)); };
#endif
      constexpr bool WV = NF == 2;   // Westervelt operands possible (S.mn1 decides at run time)
      if (STAGE == 0)
        E.a0 = ld(S.u0), E.b0 = ld(S.v0);
      else if (STAGE == 4)
        E.b0 = ld(S.v0);
      else
        E.w = ld(S.vn);
      if (WV && S.mn1)
        E.m1 = ld(S.mn1), E.m0v = ld(S.m0);
      else
        E.mi = ld(S.minv);
      if (STAGE == 3)
        E.au = ld(S.u_), E.av = ld(S.v_);
      else if (STAGE == 1)
        E.au = ld(S.u_), E.av = ld(S.v_), E.a0 = ld(S.u0), E.b0 = ld(S.v0);
      else if (STAGE == 5)
        E.b0 = ld(S.v0);
      else if (STAGE == 6)
        E.b0 = ld(S.v0), E.au = ld(S.u_);
      else if (STAGE == 7)
        E.b0 = ld(S.v0), E.au = ld(S.u_), E.av = ld(S.v_);
      }
      if (part != 1)
      {
        constexpr bool WV2 = NF == 2;
        const A2 a2 = E.on ? reinterpret_cast<const A2*>(y_l)[E.i] : A2(0.0);
        E.bv = V2{(T)a2[0], (T)a2[1]};
        if ((WV2 && S.mn1) || STAGE >= 4)
          E.us = E.on ? reinterpret_cast<const V2*>(x_l)[E.i] : V2(T(0));   // the stage input u_n of the interior dofs is still in LDS
      }
    };
    auto epi_store = [&](const Epi& E) __attribute__((always_inline))
    {
      if (!E.on)
        return;
      auto st = [&](T* ptr, V2 val) { __builtin_nontemporal_store(val, reinterpret_cast<V2*>(ptr + E.o)); };
      V2 kv;
      if (NF == 2 && S.mn1)
      {
        const V2 vs = stage_is_first(STAGE) ? E.b0 : E.w;
        kv = (E.bv - E.m1 * vs * vs) / (E.m0v + E.m1 * E.us);
      }
      else
        kv = E.bv * E.mi;
      if (STAGE == 0)
      {
        st(S.u_, E.b0 * S.bdt + E.a0);
        st(S.v_, kv * S.bdt + E.b0);
        st(S.un, E.b0 * S.adt + E.a0);
        st(S.vn, kv * S.adt + E.b0);
      }
      else if (STAGE == 4)
      {
        st(S.un, E.b0 * S.adt + E.us);
        st(S.v_, kv * S.adt + E.b0);
      }
      else if (STAGE == 3)
      {
        st(S.u0, E.w * S.bdt + E.au);
        st(S.v0, kv * S.bdt + E.av);
      }
      else if (STAGE == 5)
      {
        st(S.un, E.us + (E.w * S.adt - E.b0 * S.pdt));
        st(S.v_, kv * S.adt + E.b0);
      }
      else if (STAGE == 6)
      {
        st(S.un, E.us + (E.w * S.adt - E.au * S.pdt));
        st(S.v_, kv * S.adt + E.b0);
      }
      else if (STAGE == 7)
      {
        st(S.u0, (E.us - E.av * S.pdt) + ((E.au + E.av) * T(2) + (E.b0 + E.w)) * S.b0dt);
        st(S.v0, kv * S.bdt + ((E.av * T(2) + (E.au + E.w)) - E.b0) * S.third);
      }
      else
      {
        st(S.u_, E.w * S.bdt + E.au);
        st(S.v_, kv * S.bdt + E.av);
        st(S.un, E.w * S.adt + E.a0);
        st(S.vn, kv * S.adt + E.b0);
      }
    };
    // degrees >= 6 (4-wave workgroups, blocks of up to 3375 dofs: four to five passes per thread): several ranges per pass --
    // at p=7 the one-range loop made the epilogue the longest phase of a block (6.9 of 19.2 us: a memory round trip per
    // pass).  Only where the kernel has registers to spare: at p=5 (compiled for four waves per SIMD) and on the streamed
    // kernels four ranges spill or cost a resident wave (-45 % / -30 %); profiles/r03_experiments.md section 11.
    constexpr int EPIU = EPI2 ? 2 : epiu_of<T>(P, GEOM, NF);
    constexpr bool EARLY = epi_early<T>(P);
    Epi Ef[EPIU];
    if (EARLY && STAGE != STAGE_NONE)
    {
#pragma unroll
      for (int u = 0; u < EPIU; ++u)
        epi_load(tid + u * nthr, Ef[u], 1);
    }
  __syncthreads();   // every element of the block has been accumulated
  FUS_STAMP(blk, 2);

  // ---- phase 2: the next block of this workgroup -- its prologue loads travel while this block is
  // written out ----
  {
    // (always executed, with every load predicated on has_next: the staged registers are then defined on
    // every path and not live across the trips)
    FUS_KARGS(qn);
    Meta Mn{};
    if (has_next)
      Mn = meta_of(qn, blk_next);
#ifdef FUS_PROBE_PFREE
    if constexpr (!IDX_EARLY)
      p_idx(qn, Mn, false, gi_n);
    p_load(qn, Mn, L, false, false, gi_n);
#else
    if constexpr (!IDX_EARLY)
      p_idx(qn, Mn, has_next, gi_n);
    p_load(qn, Mn, L, false, has_next, gi_n);
#endif
  }
  if (STAGE == STAGE_NONE)
  {
    V2* bg = reinterpret_cast<V2*>(bvec + int_off);
    for (int i = tid; i < nvec; i += nthr)
    {
      const A2 a2 = reinterpret_cast<const A2*>(y_l)[i];
      bg[i] = V2{(T)a2[0], (T)a2[1]};
    }
    if (tid == 0 && (sh.nint & 1))
      bvec[int_off + sh.nint - 1] = (T)y_l[sh.nint - 1];
  }
  else
  {
    // boundary terms of block-interior dofs (Linear.hpp:205; forms.py:38-39 collocated):
    // b += g(t) src - abs * v_stage ; every entry is a distinct dof
    const int k0 = S.blk_bnd_off[blk], k1 = S.blk_bnd_off[blk + 1];
    if (k1 > k0)
    {
      const T* vstage = stage_is_first(STAGE) ? S.v0 : S.vn;
      for (int k = k0 + tid; k < k1; k += nthr)
      {
        const int gi = S.bnd_idx[k];
        T add = S.gval * S.bnd_src[k] - S.bnd_abs[k] * vstage[gi];
        if (NF == 2)
          add += S.dgval * S.bnd_src2[k];
        y_l[gi - int_off] += add;
      }
      __syncthreads();
    }
    // the last dof of an odd interior range: the one-dof form of the same update
    if (tid == 0 && (sh.nint & 1))
      (void)stage_update_dof<T, STAGE>((int64_t)int_off + sh.nint - 1, (T)y_l[sh.nint - 1], S.minv, S.vn, S.un, S.u0,
                                       S.v0, S.u_, S.v_, S.adt, S.bdt, (NF == 2) ? S.m0 : nullptr,
                                       (NF == 2) ? S.mn1 : nullptr, LeanRK<T>{S.b0dt, S.pdt, S.third});
    // first pass: the ranges whose HBM operands were requested before the barrier
#pragma unroll
    for (int u = 0; u < EPIU; ++u)
      epi_load(tid + u * nthr, Ef[u], EARLY ? 2 : 0);
#pragma unroll
    for (int u = 0; u < EPIU; ++u)
      epi_store(Ef[u]);
    for (int i = tid + EPIU * nthr; i < nvec; i += EPIU * nthr)
    {
      Epi E[EPIU];
#pragma unroll
      for (int u = 0; u < EPIU; ++u)
        epi_load(i + u * nthr, E[u], 0);
#pragma unroll
      for (int u = 0; u < EPIU; ++u)
        epi_store(E[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < US; ++u)
    if (ppv[u] >= 0)
      partial[ppv[u]] = (T)y_l[sh.nint + tid + u * nthr];
  for (int k = tid + US * nthr; k < nsh; k += nthr)
    partial[pp[k]] = (T)y_l[sh.nint + k];
  }
  FUS_TRACE_END(blk);
  if (!has_next)
    break;
  __syncthreads();  // every wave has read what it needs of this block's x_l / y_l
  blk = blk_next;
  first = false;
  {
    // the next block's first trip: its element ids (and, streamed geometry, its factors)
    FUS_KARGS(qf);
    const Meta Mf = meta_of(qf, blk);
    first_fetch(qf, Mf);
  }
  }
#undef FUS_KARGS
#undef FUS_LANE_COORDS
#undef FUS_PHASE_LDS
#undef FUS_TID
}


// bsh[s] = sum over the (block, slot) pairs of shared dof s, ascending block order (a trailing
// pseudo pair carries the boundary term of a shared boundary dof, see k_boundary_partial)
template <typename T, typename I>
__global__ void k_shared_reduce(int64_t s0, int64_t s1, const I* __restrict__ sh_ptr,
                                const I* __restrict__ sh_pairs, const T* __restrict__ partial,
                                T* __restrict__ bsh)
{
  const int64_t s = s0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= s1)
    return;
  T acc = T(0);
  for (I k = sh_ptr[s]; k < sh_ptr[s + 1]; ++k)
    acc += partial[sh_pairs[k]];
  bsh[s] = acc;
}

// Boundary term of a shared boundary dof for the NEXT stage, produced where that dof's stage update
// has just finished (its new stage velocity is in a register): the dof's last CSR entry is a pseudo
// pair (index >= npairs) exactly when it is a boundary dof, and slot k = pair - npairs holds
//   g(t_next) src[k] - abs[k] v_next (+ dg(t_next) src2[k])        (Linear.hpp:205; forms.py:38-39)
// -- what k_boundary_partial computes in a launch of its own.  enabled = 0: leave the slots alone.
template <typename T>
struct BndNext
{
  int enabled;
  int32_t npairs;
  const T *srcw, *absw, *src2w;
  T gnext, dgnext;
  T* partial;
};

template <typename T>
__device__ __forceinline__ void boundary_next(const BndNext<T>& B, int32_t last_pair, T vnext)
{
  if (B.enabled && last_pair >= B.npairs)
  {
    const int32_t k = last_pair - B.npairs;
    T v = B.gnext * B.srcw[k] - B.absw[k] * vnext;
    if (B.src2w)
      v += B.dgnext * B.src2w[k];
    B.partial[last_pair] = v;
  }
}



// The partial sums of the rank-local shared dofs as planes (Layout::pair_pos): plane j holds, at the dof's own
// index, the contribution of the dof's j-th sharing block; cnt[j] dofs (a prefix of the range) have one.
constexpr int FUS_MAX_PLANES = 16;
struct PartialPlanes
{
  int32_t bnd0;                  // slot of the first boundary term (after every pair's partial sum)
  int32_t cnt[FUS_MAX_PLANES];   // 0 for the planes that do not exist
  int32_t off[FUS_MAX_PLANES];
};

// Shared dofs of one rank, partial sums in planes: sum over the planes in ascending order (= ascending block
// order, the order of the CSR form below: same bits), then the boundary term of a shared boundary dof
// (bnd_mask: one bit per dof, bnd_base: boundary dofs ahead of the 64-dof word; the k-th boundary dof's term is
// in slot PL.bnd0 + k), fused with the RK4 stage update.  No index list is read: every access is at s.
template <typename T, int STAGE>
__global__ void __launch_bounds__(256)
k_shared_stage_planes(int64_t n, const PartialPlanes PL, const uint64_t* __restrict__ bnd_mask,
                      const int32_t* __restrict__ bnd_base, const T* __restrict__ partial,
                      const T* __restrict__ minv, T* __restrict__ vn, T* __restrict__ un, T* __restrict__ u0,
                      T* __restrict__ v0, T* __restrict__ u_, T* __restrict__ v_, T adt, T bdt,
                      const T* __restrict__ m0, const T* __restrict__ mn1, const BndNext<T> B, const LeanRK<T> R)
{
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n)
    return;
  T acc = T(0);
#pragma unroll
  for (int j = 0; j < FUS_MAX_PLANES; ++j)
  {
    if (s < PL.cnt[j])
      acc += partial[(int64_t)PL.off[j] + s];
  }
  int32_t pair = -1;
  if (bnd_mask)
  {
    const uint64_t w = bnd_mask[s >> 6];
    const int bit = (int)(s & 63);
    if ((w >> bit) & 1ull)
    {
      pair = PL.bnd0 + bnd_base[s >> 6] + __builtin_popcountll(w & ((1ull << bit) - 1ull));
      acc += partial[pair];
    }
  }
  const T vnext = stage_update_dof<T, STAGE>(s, acc, minv, vn, un, u0, v0, u_, v_, adt, bdt, m0, mn1, R);
  boundary_next<T>(B, pair, vnext);
}

// Shared dofs of one rank, CSR form (more than FUS_MAX_PLANES sharers of one dof, or option "planes" = 0):
// fixed-order sum of the partials fused with the RK4 stage update of
// stage_update_dof; vectors are passed offset to the shared range.
template <typename T, int STAGE>
__global__ void __launch_bounds__(256)
k_shared_stage(int64_t n, const int32_t* __restrict__ sh_ptr, const int32_t* __restrict__ sh_pairs,
               const T* __restrict__ partial, const T* __restrict__ minv, T* __restrict__ vn,
               T* __restrict__ un, T* __restrict__ u0, T* __restrict__ v0, T* __restrict__ u_,
               T* __restrict__ v_, T adt, T bdt, const T* __restrict__ m0,
               const T* __restrict__ mn1, const BndNext<T> B, const LeanRK<T> R)
{
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n)
    return;
  T acc = T(0);
  int32_t pair = 0;
  for (int32_t k = sh_ptr[s]; k < sh_ptr[s + 1]; ++k)
  {
    pair = sh_pairs[k];
    acc += partial[pair];
  }
  const T vnext = stage_update_dof<T, STAGE>(s, acc, minv, vn, un, u0, v0, u_, v_, adt, bdt, m0, mn1, R);
  boundary_next<T>(B, pair, vnext);
}

// Interface dofs (held by other ranks too), first half of the exchange: this rank's total of each
// packed dof = fixed-order sum of its block partials, written to the send buffer and to b.
// pack_idx[k] is the dof's index in the internal vectors, sh0 the index of shared slot 0; a dof
// listed for two neighbours is summed twice to the same bits.
template <typename T>
__global__ void k_if_reduce_pack(int64_t n, const int32_t* __restrict__ pack_idx, int64_t sh0,
                                 const int32_t* __restrict__ sh_ptr,
                                 const int32_t* __restrict__ sh_pairs,
                                 const T* __restrict__ partial, T* __restrict__ b,
                                 T* __restrict__ sendbuf)
{
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n)
    return;
  const int64_t u = pack_idx[k], s = u - sh0;
  T acc = T(0);
  for (int32_t q = sh_ptr[s]; q < sh_ptr[s + 1]; ++q)
    acc += partial[sh_pairs[q]];
  sendbuf[k] = acc;
  b[u] = acc;
}

// Second half: every sharer adds the ranks' totals of an interface dof in ascending rank order
// (identical bits everywhere, see k_unpack_ordered) and finishes the RK stage for it.  Vectors are
// passed whole (internal numbering); sh0 = internal index of shared slot 0 (for the dof's CSR row).
template <typename T, int STAGE>
__global__ void __launch_bounds__(256)
k_if_unpack_stage(int64_t nu, const int32_t* __restrict__ uidx, const int32_t* __restrict__ uptr,
                  const int32_t* __restrict__ usrc, const T* __restrict__ recvbuf,
                  const T* __restrict__ b, const T* __restrict__ minv, T* __restrict__ vn,
                  T* __restrict__ un, T* __restrict__ u0, T* __restrict__ v0, T* __restrict__ u_,
                  T* __restrict__ v_, T adt, T bdt, const T* __restrict__ m0,
                  const T* __restrict__ mn1, int64_t sh0, const int32_t* __restrict__ sh_ptr,
                  const int32_t* __restrict__ sh_pairs, const BndNext<T> B, const LeanRK<T> R)
{
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nu)
    return;
  const int64_t u = uidx[j];
  const T own = b[u];
  T acc = T(0);
  for (int32_t k = uptr[j]; k < uptr[j + 1]; ++k)
  {
    const int32_t sidx = usrc[k];
    acc += (sidx < 0) ? own : recvbuf[sidx];
  }
  const T vnext = stage_update_dof<T, STAGE>(u, acc, minv, vn, un, u0, v0, u_, v_, adt, bdt, m0, mn1, R);
  if (B.enabled)
    boundary_next<T>(B, sh_pairs[sh_ptr[u - sh0 + 1] - 1], vnext);
}

// Boundary term of shared boundary dofs, written as one more partial (summed last):
// slot[k] = g(t) src[k] - abs[k] * v_stage[idx[k]]     (Linear.hpp:205; forms.py:38-39)
template <typename T>
__global__ void k_boundary_partial(int64_t nb, const int32_t* __restrict__ idx,
                                   const T* __restrict__ srcw, const T* __restrict__ absw, T gval,
                                   const T* __restrict__ src2w, T dgval,
                                   const T* __restrict__ vstage, T* __restrict__ slot)
{
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nb)
  {
    T v = gval * srcw[k] - absw[k] * vstage[idx[k]];
    if (src2w)
      v += dgval * src2w[k];
    slot[k] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Setup / plumbing kernels
// ---------------------------------------------------------------------------------------------

// Geometry factors for internal element e at point q (precompute.hpp:101-213, 33-94), written in
// the operator's streaming layouts.  GORD = geometry order: 1 (8 vertices) or 2 (27 nodes, tensor
// order).
template <typename T, int N, int GORD>
__global__ void k_geometry(int64_t ncells, const int32_t* __restrict__ cell_perm,
                           const T* __restrict__ xg, const int32_t* __restrict__ xdofmap,
                           const double* __restrict__ pts, const double* __restrict__ wts,
                           T* __restrict__ G, T* __restrict__ detJ)
{
  constexpr int N2 = N * N, Nd = N * N * N;
  constexpr int NVG = (GORD == 1) ? 8 : 27;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncells * Nd)
    return;
  const int64_t e = gid / Nd;
  const int q = (int)(gid - e * Nd);
  const int a = q / N2, p = q - a * N2, bb = p / N, cc = p - bb * N;
  const int64_t cell = cell_perm[e];
  T cd[NVG][3];
  for (int v = 0; v < NVG; ++v)
    for (int j = 0; j < 3; ++j)
      cd[v][j] = xg[3 * (int64_t)xdofmap[cell * NVG + v] + j];
  T J[3][3], G6[6];
  if constexpr (GORD == 1)
    jacobian3<T>(cd, pts[a], pts[bb], pts[cc], J);
  else
    jacobian3_q2<T>(cd, pts[a], pts[bb], pts[cc], J);
  const T w = (T)(wts[a] * wts[bb] * wts[cc]);
  const T dw = geometric_factor3<T>(J, w, G6);
  detJ[e * Nd + q] = dw;
  for (int gi = 0; gi < 6; ++gi)
    G[e * (6 * Nd) + g_index<T, N>(gi * N + a, p)] = G6[gi];
}

// Affine cells: Gc[e][0..5] = K K^T |det J| (no quadrature weight), Gc[e][6] = |det J| with the
// constant Jacobian J = [x1-x0, x2-x0, x4-x0] (same formulas as geometric_factor3 with w = 1).
// affine_err_bits receives the largest deviation of the other vertices from the parallelepiped,
// relative to the cell size: the host leaves the affine path (for GEOM_TRILINEAR, or GEOM_STREAM on request) when it is not ~0.
template <typename T>
__global__ void k_geometry_affine(int64_t ncells, const int32_t* __restrict__ cell_perm,
                                  const T* __restrict__ xg, const int32_t* __restrict__ xdofmap,
                                  T* __restrict__ Gc, unsigned int* __restrict__ affine_err_bits)
{
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ncells)
    return;
  const int64_t cell = cell_perm[e];
  T cd[8][3];
  for (int v = 0; v < 8; ++v)
    for (int j = 0; j < 3; ++j)
      cd[v][j] = xg[3 * (int64_t)xdofmap[cell * 8 + v] + j];
  T J[3][3], G6[6];
  T h2 = T(0);
  for (int i = 0; i < 3; ++i)
  {
    J[i][0] = cd[1][i] - cd[0][i];
    J[i][1] = cd[2][i] - cd[0][i];
    J[i][2] = cd[4][i] - cd[0][i];
    h2 += J[i][0] * J[i][0] + J[i][1] * J[i][1] + J[i][2] * J[i][2];
  }
  T err2 = T(0);
  for (int v = 0; v < 8; ++v)
    for (int i = 0; i < 3; ++i)
    {
      const T pred = cd[0][i] + (T)(v & 1) * J[i][0] + (T)((v >> 1) & 1) * J[i][1] + (T)(v >> 2) * J[i][2];
      const T d = cd[v][i] - pred;
      err2 = (d * d > err2) ? d * d : err2;
    }
  const T dw = geometric_factor3<T>(J, T(1), G6);
  for (int gi = 0; gi < 6; ++gi)
    Gc[e * 7 + gi] = G6[gi];
  Gc[e * 7 + 6] = dw;
  const float rel = (float)sqrt((double)(err2 / h2));
  atomicMax(affine_err_bits, __float_as_uint(rel));  // non-negative floats order like their bits
}

// First-order hexahedra, GEOM_TRILINEAR: Cc[e][7][3] = coefficients (c100 c010 c001 c110 c101 c011 c111)
// of the cell's trilinear map x(X) (vertex v = vx + 2 vy + 4 vz), formed as differences of edge
// vectors so that each keeps the relative accuracy of the edge lengths.
template <typename T>
__global__ void k_geometry_trilinear(int64_t ncells, const int32_t* __restrict__ cell_perm,
                                     const T* __restrict__ xg, const int32_t* __restrict__ xdofmap,
                                     T* __restrict__ Cc)
{
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ncells)
    return;
  const int64_t cell = cell_perm[e];
  for (int i = 0; i < 3; ++i)
  {
    T x[8];
    for (int v = 0; v < 8; ++v)
      x[v] = xg[3 * (int64_t)xdofmap[cell * 8 + v] + i];
    const T e10 = x[1] - x[0], e32 = x[3] - x[2], e54 = x[5] - x[4], e76 = x[7] - x[6];
    const T e20 = x[2] - x[0], e64 = x[6] - x[4];
    T* o = Cc + e * 21 + i;
    o[0] = e10;
    o[3] = e20;
    o[6] = x[4] - x[0];
    o[9] = e32 - e10;
    o[12] = e54 - e10;
    o[15] = e64 - e20;
    o[18] = (e76 - e54) - (e32 - e10);
  }
}

// Quadrilateral cells: G[e][3][N^2] (xx, xy, yy planes) and detJw[e][N^2] for internal element e
// (cpp/fenicsx-sf-naive/common/precompute.hpp; first two coordinates).  GORD = geometry order:
// 1 (4 vertices, bilinear) or 2 (9 nodes in tensor order, biquadratic).
template <typename T, int N, int GORD>
__global__ void k_geometry2d(int64_t ncells, const int32_t* __restrict__ cell_perm,
                             const T* __restrict__ xg, const int32_t* __restrict__ xdofmap,
                             const double* __restrict__ pts, const double* __restrict__ wts,
                             T* __restrict__ G, T* __restrict__ detJ)
{
  constexpr int Nd = N * N;
  constexpr int NVG = (GORD == 1) ? 4 : 9;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncells * Nd)
    return;
  const int64_t e = gid / Nd;
  const int q = (int)(gid - e * Nd);
  const int bb = q / N, cc = q - bb * N;
  const int64_t cell = cell_perm[e];
  T cd[NVG][3];
  for (int v = 0; v < NVG; ++v)
    for (int j = 0; j < 3; ++j)
      cd[v][j] = xg[3 * (int64_t)xdofmap[cell * NVG + v] + j];
  T J[2][2], G3[3];
  if constexpr (GORD == 1)
    jacobian2<T>(cd, pts[bb], pts[cc], J);
  else
    jacobian2_q2<T>(cd, pts[bb], pts[cc], J);
  const T dw = geometric_factor2<T>(J, (T)(wts[bb] * wts[cc]), G3);
  detJ[e * Nd + q] = dw;
  for (int gi = 0; gi < 3; ++gi)
    G[e * (3 * Nd) + gi * Nd + q] = G3[gi];
}

// internal layout -> reference layout G[cell][point][3], detJ[cell][point]
template <typename T, int N>
__global__ void k_geometry_export2d(int64_t ncells, const int32_t* __restrict__ cell_perm,
                                    const T* __restrict__ G, const T* __restrict__ detJ,
                                    T* __restrict__ Gout, T* __restrict__ dout)
{
  constexpr int Nd = N * N;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncells * Nd)
    return;
  const int64_t e = gid / Nd;
  const int q = (int)(gid - e * Nd);
  const int64_t cell = cell_perm[e];
  if (dout)
    dout[cell * Nd + q] = detJ[e * Nd + q];
  if (Gout)
    for (int gi = 0; gi < 3; ++gi)
      Gout[(cell * Nd + q) * 3 + gi] = G[e * (3 * Nd) + gi * Nd + q];
}

// internal streaming layout -> reference layout G[cell][point][6], detJ[cell][point]
template <typename T, int N>
__global__ void k_geometry_export(int64_t ncells, const int32_t* __restrict__ cell_perm,
                                  const T* __restrict__ G, const T* __restrict__ detJ,
                                  T* __restrict__ Gout, T* __restrict__ dout)
{
  constexpr int N2 = N * N, Nd = N * N * N;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncells * Nd)
    return;
  const int64_t e = gid / Nd;
  const int q = (int)(gid - e * Nd);
  const int a = q / N2, p = q - a * N2;
  const int64_t cell = cell_perm[e];
  if (dout)
    dout[cell * Nd + q] = detJ[e * Nd + q];
  if (Gout)
    for (int gi = 0; gi < 6; ++gi)
      Gout[(cell * Nd + q) * 6 + gi] = G[e * (6 * Nd) + g_index<T, N>(gi * N + a, p)];
}

// out_internal[perm[i]] = in_caller[i]
template <typename T>
__global__ void k_to_internal(int64_t n, const int32_t* __restrict__ perm, const T* __restrict__ in,
                              T* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    out[perm[i]] = in[i];
}

// y_caller[i] = (ACC ? y_caller[i] : 0) + in_internal[perm[i]]
template <typename T, int ACC>
__global__ void k_from_internal(int64_t n, const int32_t* __restrict__ perm,
                                const T* __restrict__ in, T* __restrict__ y)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    y[i] = (ACC ? y[i] : T(0)) + in[perm[i]];
}

// out[e] = in[cell_perm[e]]
template <typename T>
__global__ void k_cells_to_internal(int64_t n, const int32_t* __restrict__ cell_perm,
                                    const T* __restrict__ in, T* __restrict__ out)
{
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n)
    out[e] = in[cell_perm[e]];
}

template <typename T>
__global__ void k_fill(int64_t n, T* __restrict__ x, T v)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    x[i] = v;
}

// y += x
template <typename T>
__global__ void k_add_vec(int64_t n, const T* __restrict__ x, T* __restrict__ y)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] += x[i];
}

// minv = 1/m where m != 0 (padding slots stay 0)
template <typename T>
__global__ void k_reciprocal(int64_t n, const T* __restrict__ m, T* __restrict__ minv)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    minv[i] = (m[i] != T(0)) ? T(1) / m[i] : T(0);
}

// Streaming copy y = x with 16-byte non-temporal accesses, one vector per thread (a flat grid: on MI355X
// this form reaches the device's copy bandwidth, 6.3-6.6 TB/s, where grid-stride loops over a few thousand
// workgroups stay at 4.4-5.4 TB/s -- tools/bw_probe.hip): the measured streaming bandwidth beside the
// roofline fractions (fus_measure_bandwidth).
template <typename V>  // V = 2 doubles as an ext_vector (a template only so that every unit may include it)
__global__ void __launch_bounds__(256) k_stream_copy(int64_t nvec, const V* __restrict__ x, V* __restrict__ y)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nvec)
    __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i);
}

// halo helpers
// pack: sendbuf[k] = vec[idx[k]] over the concatenated neighbour lists
// Receiver sampling (the reference evaluates its solution at points with Function::eval after locating their cells:
// python/src/fenicsxfus/utils.py:10-47, cpp/mwe/parallel_eval_line/main.cpp:49-84).  One wave per receiver r:
//   out[r] = sum_{i0,i1,i2} l_i0(X0) l_i1(X1) l_i2(X2) vec[idx[r][i0,i1,i2]]
// on the RESIDENT vector in internal numbering (idx = dof_perm of the receiver's cell dofs, bas = the 1-D Lagrange
// basis values at the receiver's reference coordinates, both built once in fus_model_set_receivers); lane k adds
// the entries k, k + 64, ... in that order, then a fixed butterfly: the same bits on every call.
template <typename T, int TD>
__global__ void __launch_bounds__(256)
k_sample(int64_t npts, int N, const int32_t* __restrict__ idx, const T* __restrict__ bas,
         const T* __restrict__ vec, T* __restrict__ out)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= npts)
    return;  // (the whole wave leaves)
  const int N2 = N * N, Nd = (TD == 3) ? N2 * N : N2;
  const int32_t* __restrict__ ix = idx + r * Nd;
  const T* __restrict__ b = bas + r * (TD * N);
  T acc = T(0);
  for (int k = lane; k < Nd; k += 64)
  {
    T w;
    if (TD == 3)
    {
      const int i0 = k / N2, rem = k - i0 * N2, i1 = rem / N, i2 = rem - i1 * N;
      w = b[i0] * b[N + i1] * b[2 * N + i2];
    }
    else
    {
      const int i0 = k / N, i1 = k - i0 * N;
      w = b[i0] * b[N + i1];
    }
    acc += w * vec[ix[k]];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    acc += __shfl_xor(acc, o, 64);
  if (lane == 0)
    out[r] = acc;
}

template <typename T>
__global__ void k_pack(int64_t n, const int32_t* __restrict__ idx, const T* __restrict__ vec,
                       T* __restrict__ buf)
{
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n)
    buf[k] = vec[idx[k]];
}
// unpack: every sharer adds the partials of an interface dof in ascending rank order (identical
// bits on all ranks): vec[uidx[j]] = sum_k (src[k] < 0 ? own partial : recvbuf[src[k]])
template <typename T>
__global__ void k_unpack_ordered(int64_t nu, const int32_t* __restrict__ uidx,
                                 const int32_t* __restrict__ uptr, const int32_t* __restrict__ usrc,
                                 const T* __restrict__ recvbuf, T* __restrict__ vec)
{
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nu)
    return;
  const int32_t u = uidx[j];
  const T own = vec[u];
  T acc = T(0);
  for (int32_t k = uptr[j]; k < uptr[j + 1]; ++k)
  {
    const int32_t sidx = usrc[k];
    acc += (sidx < 0) ? own : recvbuf[sidx];
  }
  vec[u] = acc;
}

} // namespace fus
