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
  const int16_t* rounds;
  const uint16_t* ldm;
  int32_t nblocks;
  int32_t blk_begin; // first block of this launch (grid = a contiguous range of blocks)
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

template <typename T, int N>
struct DTab
{
  T d[N * N];
  T w[N], x[N];  // 1-D GLL weights and points (kernel argument -> scalar registers)
};

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
#ifndef FUS_TRI_WAVES
#define FUS_TRI_WAVES(P) ((P) <= 5 ? 4 : 2)
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
  GEOM_TRILINEAR = 2
};

// numbers per cell held in LDS by the per-cell geometry modes
__host__ __device__ constexpr int geom_cell_stride(int geom)
{
  return geom == GEOM_AFFINE ? 7 : (geom == GEOM_TRILINEAR ? 21 : 0);
}

// 1 / x to working precision from the hardware estimate (the per-point G of GEOM_TRILINEAR)
__device__ __forceinline__ double fast_rcp(double x)
{
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
  if (GEOM == GEOM_STREAM && er >= 0)
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
template <typename T, int N, int OP, int ATOMIC, int NF>
__device__ __forceinline__ void elem_compute2d(const ElemIn<T, N, OP, GEOM_STREAM, 2>& in,
                                               const T (&Drb)[N], const T (&Drc)[N],
                                               const T (&Dcb)[N], const T (&Dcc)[N],
                                               const T* __restrict__ x_l, T* __restrict__ y_l,
                                               T* __restrict__ sA,
                                               const uint16_t* __restrict__ ldm_l,
                                               const T* __restrict__ cf_l,
                                               const T* __restrict__ x2_l,
                                               const T* __restrict__ cf2_l, int p, int b, int c)
{
  constexpr int Nd = N * N;
  if (in.er < 0)
    return;
  const int li = ldm_l[in.er * Nd + p];
  const T cf = (NF == 2) ? T(1) : cf_l[in.er];
  T Y;
  if (OP == OP_STIFFNESS)
  {
    const T X = (NF == 2) ? cf_l[in.er] * x_l[li] + cf2_l[in.er] * x2_l[li] : x_l[li];
    sA[p] = X;
    FUS_WAVE_SYNC();
    T d0 = T(0), d1 = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j)
    {
      d0 += Drb[j] * sA[j * N + c];
      d1 += Drc[j] * sA[b * N + j];
    }
    const T F0 = cf * (in.g2[0] * d0 + in.g2[1] * d1);
    const T F1 = cf * (in.g2[1] * d0 + in.g2[2] * d1);
    FUS_WAVE_SYNC();
    sA[p] = F0;
    FUS_WAVE_SYNC();
    T acc = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j)
      acc += Dcb[j] * sA[j * N + c];
    FUS_WAVE_SYNC();
    sA[p] = F1;
    FUS_WAVE_SYNC();
#pragma unroll
    for (int j = 0; j < N; ++j)
      acc += Dcc[j] * sA[b * N + j];
    Y = acc;
  }
  else
    Y = cf * x_l[li] * in.g2[0];
  if (ATOMIC)
    __hip_atomic_fetch_add(&y_l[li], Y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    y_l[li] += Y;
}

template <typename T, int N, int OP, int ATOMIC, int NF, int GEOM>
__device__ __forceinline__ void elem_compute(const ElemIn<T, N, OP, GEOM>& in, const DTab<T, N>& Dk,
                                             const T (&Drb)[N], const T (&Drc)[N], const T (&Dcb)[N],
                                             const T (&Dcc)[N], const T* __restrict__ x_l,
                                             T* __restrict__ y_l, T* __restrict__ sA,
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
  constexpr bool DLDS = (GEOM == GEOM_TRILINEAR) || (GEOM == GEOM_AFFINE && (N == 6 || N == 7));
  if (in.er < 0)
    return;
  TriLane<T> tri;
  if (GEOM == GEOM_TRILINEAR)
    tri.init(gc_l + in.er * 21, pb, pc);
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
#pragma unroll
      for (int a = 0; a < N; ++a)
        X[a] = x_l[li[a]];
    }
#ifndef FUS_REMAP_ON
#define FUS_REMAP_ON 1
#endif
    // REMAP: the index-1 and index-2 contractions also run in registers.  Lane (b, c) re-reads the tile
    // as lane (a' = b, c) with index 1 along its registers (then as (a' = b, b' = c) with index 2 along
    // them), contracts with the derivative table in scalar registers and writes the result back for
    // the (b, c) owner: 5 reads + 5 writes + 5 reads per direction instead of 25 reads + the lane's
    // derivative rows -- the element trips of the per-cell geometry kernels are bound by the LDS port.
#ifndef FUS_REMAP_MAXN
#define FUS_REMAP_MAXN 7  // degree 7 (N = 8) keeps the tile-read form: re-mapped it measured 5-14 % slower
#endif
#ifndef FUS_REMAP_STREAM
#define FUS_REMAP_STREAM 0  // the streamed kernel sits on the bandwidth roofline: measured separately
#endif
    constexpr bool REMAP = FUS_REMAP_ON && (GEOM != GEOM_STREAM || FUS_REMAP_STREAM) && N <= FUS_REMAP_MAXN;
    if constexpr (REMAP)
    {
      // plane stride of the tile: N^2, padded by one where the re-mapped accesses (lanes (b, c) at
      // b * TS + ...) would otherwise fall on the same LDS banks for every b (N = 8: 8-way, N = 4: 2-way)
      constexpr int TS = (N == 8 || N == 4) ? N2 + 1 : N2;
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
        sA[a * TS + p] = X[a];
      FUS_WAVE_SYNC();
      T Tb[N], Uc[N];
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
        T acc = T(0);
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
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < N; ++k)
          acc += Dk.d[q * N + k] * Uc[k];
        sA[b * TS + c * N + q] = acc;  // d/dX2 at point (b, c, q)
      }
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
        F2[a] = sA[a * TS + p];
      // stiffness::transform (spectral_op.hpp:113-130)
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
        T acc = T(0);
#pragma unroll
        for (int q = 0; q < N; ++q)
          acc += Dk.d[q * N + j] * Tb[q];
        sA[b * TS + j * N + c] = acc;
      }
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
      {
        T acc = sA[a * TS + p];
#pragma unroll
        for (int q = 0; q < N; ++q)
          acc += Dk.d[q * N + a] * F0[q];
        Y[a] = acc;
      }
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
        sA[a * TS + p] = F2[a];
      FUS_WAVE_SYNC();
#pragma unroll
      for (int k = 0; k < N; ++k)
        Uc[k] = sA[b * TS + c * N + k];
      FUS_WAVE_SYNC();
#pragma unroll
      for (int j = 0; j < N; ++j)
      {
        T acc = T(0);
#pragma unroll
        for (int q = 0; q < N; ++q)
          acc += Dk.d[q * N + j] * Uc[q];
        sA[b * TS + c * N + j] = acc;
      }
      FUS_WAVE_SYNC();
#pragma unroll
      for (int a = 0; a < N; ++a)
        Y[a] += sA[a * TS + p];
    }
    else
    {
    // derivative along tensor index 0: registers only (spectral_op.hpp:194-196)
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
      T acc = T(0);
#pragma unroll
      for (int q = 0; q < N; ++q)
        acc += Dk.d[q * N + a] * F0[q];
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
#pragma unroll
  for (int a = 0; a < N; ++a)
  {
    if (ATOMIC)
      __hip_atomic_fetch_add(&y_l[li[a]], Y[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      y_l[li[a]] += Y[a];
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
    T acc = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i)
      acc += Dk.d[q * N + i] * X[i];
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

template <typename T, int N, int DL = 0>
__device__ __forceinline__ void elem_stiff_bwd(const DTab<T, N>& Dk, const T (&Dcb_)[N], const T (&Dcc_)[N],
                                               const T* __restrict__ D_l,
                                               T* __restrict__ y_l, T* __restrict__ sA, int p, int b,
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
    T acc = T(0);
#pragma unroll
    for (int q = 0; q < N; ++q)
      acc += Dk.d[q * N + a] * F0[q];
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
    __hip_atomic_fetch_add(&y_l[li[a]], Y[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Single geometry register set (see elem_stiff_fwd): fp64, degrees 5 and 6, streamed geometry,
// LDS-atomic accumulation.  With two sets those kernels need 290-330 registers, i.e. one wave per
// SIMD and one block per CU; with one set they fit 256: two waves per SIMD, two blocks per CU.
#ifdef FUS_NO_PF1  // experiment switch
#define FUS_PF1(T, P, OP, ATOMIC, GEOM, TD) 0
#else
#define FUS_PF1(T, P, OP, ATOMIC, GEOM, TD)                                                         \
  ((sizeof(T) == 8 && (P) >= 5 && (P) <= FUS_PF1_MAXP && (OP) == OP_STIFFNESS && (ATOMIC) && (GEOM) == GEOM_STREAM && (TD) == 3) ? 1 : 0)
#ifndef FUS_PF1_MAXP
#define FUS_PF1_MAXP 7
#endif
#endif

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
template <typename T, int P, int OP, int ATOMIC, int STAGE, int NF, int GEOM, int TD = 3>
__global__ void __launch_bounds__((P <= 4) ? 512 : 256, (P <= 4 && GEOM == GEOM_AFFINE)
                                                            ? 4
                                                            : ((GEOM == GEOM_TRILINEAR || (GEOM == GEOM_AFFINE && P <= 6)) ? FUS_TRI_WAVES(P) : (FUS_PF1(T, P, OP, ATOMIC, GEOM, TD) ? 2 : 1)))
k_block_op(const BlockArgs A, const DTab<T, P + 1> Dk, const T* __restrict__ Dg,
           const T* __restrict__ geo, const T* __restrict__ coef, const T* __restrict__ x,
           T* __restrict__ bvec, T* __restrict__ partial, const StageArgs<T> S)
{
  static_assert(TD == 3 || GEOM == GEOM_STREAM, "quadrilaterals use the streamed geometry");
  constexpr int N = P + 1, N2 = N * N, Nd = (TD == 3) ? N * N * N : N * N;
  constexpr int EPW = (64 / N2) > 0 ? (64 / N2) : 1;

  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* x_l = reinterpret_cast<T*>(smem_raw);
  T* y_l = x_l + A.lds_nloc;
  T* x2_l = y_l + A.lds_nloc;                               // second input (NF == 2 only)
  T* scratch = x2_l + (NF == 2 ? A.lds_nloc : 0);
  T* D_l = scratch + (size_t)A.waves * EPW * (Nd + N);     // derivative table (tiles: Nd + N each, see elem_compute REMAP)
  T* cf_l = D_l + N2;                                       // per-element coefficient(s)
  T* cf2_l = cf_l + A.lds_nelem;
  constexpr int GCS = geom_cell_stride(GEOM);               // per-cell geometry numbers (7 / 21 / 0)
  T* gc_l = cf2_l + (NF == 2 ? A.lds_nelem : 0);            // affine: 6 G + 1 detJ; trilinear: 21 map coefficients
  T* w_l = gc_l + GCS * A.lds_nelem;                        // 1-D weights (8 slots), 1-D points (8 slots)
  T* pt_l = w_l + 8;
  uint16_t* ldm_l = reinterpret_cast<uint16_t*>(w_l + (GCS ? 16 : 0));  // 16-B aligned
  int16_t* rt_l = reinterpret_cast<int16_t*>(ldm_l + (size_t)A.lds_nelem * Nd);

  const int blk = blockIdx.x + A.blk_begin;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const ShapeDev sh = A.shapes[A.blk_shape[blk]];
  const int elem_off = A.blk_elem_off[blk];
  const int int_off = A.blk_int_off[blk];
  const int64_t sh_off = A.blk_sh_off[blk];

  const int lane = tid & 63, wave = tid >> 6;
  const int s = lane / N2, p = lane - s * N2;
  const int b = p / N, c = p - b * N;
  const bool active = s < EPW;
  const int slots = A.waves * EPW;
  const int myslot = wave * EPW + (active ? s : 0);
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

  FUS_STAMP(blk, 0);
  // first trip's geometry is requested before the block's dof values are staged
  ElemIn<T, N, OP, GEOM, TD> inA, inB;
  {
    int e0 = -1;
    if (active && ntrips > 0)
      e0 = ATOMIC ? (myslot < sh.nelem ? myslot : -1) : (int)A.rounds[sh.rounds_off + myslot];
    elem_fetch<T, N, OP, GEOM, TD>(inA, e0, geo, elem_off, p);
  }

  // ---- prologue: stage the block's dof values, local dofmaps and coefficients in LDS, clear the
  // accumulator.  Measured (phase timestamps, tools/gpu_trace.py) the prologue was the longest phase
  // of a block when each of its copy loops waited for its own memory round trip, so every load that
  // does not depend on another is issued first (one round trip), the gather of the shared dofs
  // (it needs their indices) second, and only then the LDS stores.  Blocks larger than the first
  // batches finish in plain loops. ----
  {
    typedef T V2 __attribute__((ext_vector_type(2)));
    typedef uint32_t U4 __attribute__((ext_vector_type(4)));
    constexpr int UI = 4, US = 5, UL = 4;  // per thread: interior 16-B vectors, shared dofs, dofmap 16-B vectors
    const V2* xg = reinterpret_cast<const V2*>(x + int_off);  // int_off is a multiple of 16
    const V2* xg2 = reinterpret_cast<const V2*>((NF == 2 ? S.x2 : x) + int_off);
    const int nvec = sh.nint >> 1;
    const int nsh = sh.nloc - sh.nint;
    const int32_t* gix = A.sh_gidx + sh_off;
    const int n16 = (sh.nelem * Nd * 2 + 15) >> 4;  // ldm_off is a multiple of 8 entries
    const U4* lsrc = reinterpret_cast<const U4*>(A.ldm + sh.ldm_off);
    const int ngc = sh.nelem * GCS;

    // round trip 1
    V2 xi[UI], xi2[UI];
    int gi[US];
    U4 lq[UL];
#pragma unroll
    for (int u = 0; u < UI; ++u)
      if (tid + u * nthr < nvec)
      {
        xi[u] = xg[tid + u * nthr];
        if (NF == 2)
          xi2[u] = xg2[tid + u * nthr];
      }
#pragma unroll
    for (int u = 0; u < US; ++u)
      gi[u] = (tid + u * nthr < nsh) ? gix[tid + u * nthr] : 0;
#pragma unroll
    for (int u = 0; u < UL; ++u)
      if (tid + u * nthr < n16)
        lq[u] = lsrc[tid + u * nthr];
    const T cfv = (tid < sh.nelem) ? coef[elem_off + tid] : T(0);
    const T cf2v = (NF == 2 && tid < sh.nelem) ? S.coef2[elem_off + tid] : T(0);
    const T gcv = (tid < ngc) ? geo[(int64_t)elem_off * GCS + tid] : T(0);
    const T gcv2 = (GCS > 7 && tid + nthr < ngc) ? geo[(int64_t)elem_off * GCS + tid + nthr] : T(0);
    const T dgv = (tid < N2 + 2 * N) ? Dg[tid] : T(0);  // derivative table, 1-D weights, 1-D points
    const T xtail = (tid == 0 && (sh.nint & 1)) ? x[int_off + sh.nint - 1] : T(0);
    const T xtail2 = (NF == 2 && tid == 0 && (sh.nint & 1)) ? S.x2[int_off + sh.nint - 1] : T(0);
    // round trip 2: the shared dofs' values
    T xs[US], xs2[US];
#pragma unroll
    for (int u = 0; u < US; ++u)
    {
      xs[u] = x[gi[u]];
      if (NF == 2)
        xs2[u] = S.x2[gi[u]];
    }
    // LDS stores
#pragma unroll
    for (int u = 0; u < UI; ++u)
      if (tid + u * nthr < nvec)
      {
        reinterpret_cast<V2*>(x_l)[tid + u * nthr] = xi[u];
        if (NF == 2)
          reinterpret_cast<V2*>(x2_l)[tid + u * nthr] = xi2[u];
        reinterpret_cast<V2*>(y_l)[tid + u * nthr] = V2(T(0));
      }
#pragma unroll
    for (int u = 0; u < UL; ++u)
      if (tid + u * nthr < n16)
        reinterpret_cast<U4*>(ldm_l)[tid + u * nthr] = lq[u];
    if (tid < sh.nelem)
    {
      cf_l[tid] = cfv;
      if (NF == 2)
        cf2_l[tid] = cf2v;
    }
    if (tid < ngc)
      gc_l[tid] = gcv;
    if (GCS > 7 && tid + nthr < ngc)
      gc_l[tid + nthr] = gcv2;
    if (tid < N2)
      D_l[tid] = dgv;
    if (GCS && tid >= N2 && tid < N2 + N)
      w_l[tid - N2] = dgv;
    if (GCS && tid >= N2 + N && tid < N2 + 2 * N)
      pt_l[tid - N2 - N] = dgv;
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
    if (tid == 0 && (sh.nint & 1))
    {
      x_l[sh.nint - 1] = xtail;
      if (NF == 2)
        x2_l[sh.nint - 1] = xtail2;
      y_l[sh.nint - 1] = T(0);
    }
#pragma unroll
    for (int u = 0; u < US; ++u)
      if (tid + u * nthr < nsh)
      {
        x_l[sh.nint + tid + u * nthr] = xs[u];
        if (NF == 2)
          x2_l[sh.nint + tid + u * nthr] = xs2[u];
        y_l[sh.nint + tid + u * nthr] = T(0);
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
          reinterpret_cast<V2*>(y_l)[base + u * nthr] = V2(T(0));
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
          v2[u] = S.x2[g[u]];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (base + u * nthr < nsh)
        {
          x_l[sh.nint + base + u * nthr] = v[u];
          if (NF == 2)
            x2_l[sh.nint + base + u * nthr] = v2[u];
          y_l[sh.nint + base + u * nthr] = T(0);
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
      cf_l[k] = coef[elem_off + k];
      if (NF == 2)
        cf2_l[k] = S.coef2[elem_off + k];
    }
    for (int k = tid + (GCS > 7 ? 2 : 1) * nthr; k < ngc; k += nthr)
      gc_l[k] = geo[(int64_t)elem_off * GCS + k];
    // the round table (deterministic mode only) -> LDS, so the per-round element lookup is not a
    // global load that would drain the geometry prefetch queue (vmcnt retires in order)
    if (!ATOMIC)
      for (int k = tid; k < sh.nrounds * slots; k += nthr)
        rt_l[k] = A.rounds[sh.rounds_off + k];
  }

  T* sA = scratch + (size_t)(wave * EPW + (active ? s : 0)) * (Nd + N);
  T* sB = sA;  // single exchange tile per element slot

  // lane-dependent rows/columns of the derivative table (tiny, cache resident)
  __syncthreads();
  FUS_STAMP(blk, 1);
  T Drb[N], Drc[N], Dcb[N], Dcc[N];
#pragma unroll
  for (int j = 0; j < N; ++j)
  {
    constexpr bool inreg = (OP == OP_STIFFNESS) && !(FUS_PF1(T, P, OP, ATOMIC, GEOM, TD) && P >= 6)
                           && GEOM != GEOM_TRILINEAR && !(GEOM == GEOM_AFFINE && (P == 5 || P == 6));
    Drb[j] = inreg ? D_l[b * N + j] : T(0);
    Drc[j] = inreg ? D_l[c * N + j] : T(0);
    Dcb[j] = inreg ? D_l[j * N + b] : T(0);
    Dcc[j] = inreg ? D_l[j * N + c] : T(0);
  }
  T w3[N];  // w_q = w_a w_b w_c of this lane's points (affine geometry only)
#pragma unroll
  for (int a = 0; a < N; ++a)
    w3[a] = (GEOM == GEOM_AFFINE) ? w_l[a] * w_l[b] * w_l[c] : T(0);
  const T wbc = (GEOM == GEOM_TRILINEAR) ? w_l[b] * w_l[c] : T(0);
  const T pb = (GEOM == GEOM_TRILINEAR) ? pt_l[b] : T(0), pc = (GEOM == GEOM_TRILINEAR) ? pt_l[c] : T(0);

  // ---- trips, two per iteration: while one register set is consumed the other is in flight ----
#define FUS_ELEM_COMPUTE(in)                                                                       \
  do                                                                                               \
  {                                                                                                \
    if constexpr (TD == 3)                                                                         \
      elem_compute<T, N, OP, ATOMIC, NF, GEOM>(in, Dk, Drb, Drc, Dcb, Dcc, x_l, y_l, sA, sB,       \
                                               ldm_l, cf_l, x2_l, cf2_l, gc_l, w3, D_l, wbc, pb,   \
                                               pc, p, b, c);                                       \
    else                                                                                           \
      elem_compute2d<T, N, OP, ATOMIC, NF>(in, Drb, Drc, Dcb, Dcc, x_l, y_l, sA, ldm_l, cf_l,      \
                                           x2_l, cf2_l, p, b, c);                                  \
  } while (0)
  if constexpr (FUS_PF1(T, P, OP, ATOMIC, GEOM, TD))
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
        elem_stiff_bwd<T, N, DL>(Dk, Dcb, Dcc, D_l, y_l, sA, p, b, c, li, F0, F1, F2);
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
    if (!ATOMIC && A.waves > 1)
      __syncthreads();
    elem_fetch<T, N, OP, GEOM, TD>(inA, elem_of(r + 2), geo, elem_off, p);
    if (has1)
    {
      FUS_ELEM_COMPUTE(inB);
      if (r == 0)
        FUS_STAMP(blk, 6);
      if (!ATOMIC && A.waves > 1)
        __syncthreads();
    }
  }
#undef FUS_ELEM_COMPUTE
  __syncthreads();
  FUS_STAMP(blk, 2);

  // ---- epilogue: each dof written once ----
  typedef T V2 __attribute__((ext_vector_type(2)));
  const int nvec = sh.nint >> 1;
  if (STAGE == STAGE_NONE)
  {
    V2* bg = reinterpret_cast<V2*>(bvec + int_off);
    for (int i = tid; i < nvec; i += nthr)
      bg[i] = reinterpret_cast<const V2*>(y_l)[i];
    if (tid == 0 && (sh.nint & 1))
      bvec[int_off + sh.nint - 1] = y_l[sh.nint - 1];
  }
  else
  {
    // boundary terms of block-interior dofs (Linear.hpp:205; forms.py:38-39 collocated):
    // b += g(t) src - abs * v_stage ; every entry is a distinct dof
    const int k0 = S.blk_bnd_off[blk], k1 = S.blk_bnd_off[blk + 1];
    if (k1 > k0)
    {
      const T* vstage = (STAGE == 0) ? S.v0 : S.vn;
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
    // fused stage update on the contiguous interior range (16-byte accesses); the scalar tail
    // element of an odd range is handled by thread 0 with the same formulas
    const int ntot = nvec + (sh.nint & 1);
    // degrees <= 4, one operator input: two interior ranges per pass, the loads of both in flight before
    // the first store (+1-3 % on the per-cell geometry paths; with the second input's extra operands
    // -- Westervelt -- it measured 5 % slower, and the higher degrees were not measured)
    constexpr bool EPI2 = (P <= 4) && (NF == 1);
    if constexpr (EPI2)
    {
    struct Epi
    {
      bool on, tail;
      int i, o;
      V2 bv, mi, w, a0, b0, au, av, m1, m0v, us;
    };
    auto epi_load = [&](int i, Epi& E) __attribute__((always_inline))
    {
      E.on = i < ntot;
      if (!E.on)
        return;
      E.i = i, E.tail = i >= nvec, E.o = int_off + 2 * i;
      auto ld = [&](const T* ptr) -> V2 {
        V2 r;
        if (!E.tail)
          r = __builtin_nontemporal_load(reinterpret_cast<const V2*>(ptr + E.o));
        else
          r[0] = ptr[E.o], r[1] = T(0);
        return r;
      };
      if (!E.tail)
        E.bv = reinterpret_cast<const V2*>(y_l)[i];
      else
        E.bv[0] = y_l[2 * i], E.bv[1] = T(0);
      if (STAGE == 0)
        E.a0 = ld(S.u0), E.b0 = ld(S.v0);
      else
        E.w = ld(S.vn);
      if (NF == 2 && S.mn1)
      {
        if (!E.tail)
          E.us = reinterpret_cast<const V2*>(x_l)[i];
        else
          E.us[0] = x_l[2 * i], E.us[1] = T(0);
        E.m1 = ld(S.mn1), E.m0v = ld(S.m0);
      }
      else
        E.mi = ld(S.minv);
      if (STAGE == 3)
        E.au = ld(S.u_), E.av = ld(S.v_);
      else if (STAGE != 0)
        E.au = ld(S.u_), E.av = ld(S.v_), E.a0 = ld(S.u0), E.b0 = ld(S.v0);
    };
    auto epi_store = [&](const Epi& E) __attribute__((always_inline))
    {
      if (!E.on)
        return;
      auto st = [&](T* ptr, V2 val) {
        if (!E.tail)
          __builtin_nontemporal_store(val, reinterpret_cast<V2*>(ptr + E.o));
        else
          ptr[E.o] = val[0];
      };
      V2 kv;
      if (NF == 2 && S.mn1)
      {
        const V2 vs = (STAGE == 0) ? E.b0 : E.w;
        V2 den = E.m0v + E.m1 * E.us;
        if (E.tail)
          den[1] = T(1);
        kv = (E.bv - E.m1 * vs * vs) / den;
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
      else if (STAGE == 3)
      {
        st(S.u0, E.w * S.bdt + E.au);
        st(S.v0, kv * S.bdt + E.av);
      }
      else
      {
        st(S.u_, E.w * S.bdt + E.au);
        st(S.v_, kv * S.bdt + E.av);
        st(S.un, E.w * S.adt + E.a0);
        st(S.vn, kv * S.adt + E.b0);
      }
    };
    for (int i = tid; i < ntot; i += 2 * nthr)
    {
      Epi E0, E1;
      epi_load(i, E0);
      epi_load(i + nthr, E1);
      epi_store(E0);
      epi_store(E1);
    }
    }
    else
    for (int i = tid; i < ntot; i += nthr)
    {
      const bool tail = i >= nvec;
      const int o = int_off + 2 * i;
      V2 bv, mi, w, a0, b0, au, av;
      auto ld = [&](const T* ptr) -> V2 {
        V2 r;
        if (!tail)  // read once per stage: keep these streams out of L2 / MALL
          r = __builtin_nontemporal_load(reinterpret_cast<const V2*>(ptr + o));
        else
          r[0] = ptr[o], r[1] = T(0);
        return r;
      };
      auto st = [&](T* ptr, V2 val) {
        if (!tail)
          __builtin_nontemporal_store(val, reinterpret_cast<V2*>(ptr + o));
        else
          ptr[o] = val[0];
      };
      if (!tail)
        bv = reinterpret_cast<const V2*>(y_l)[i];
      else
        bv[0] = y_l[2 * i], bv[1] = T(0);
      // stage inputs v_n (= v0 at stage 0) are needed by every variant below; load once
      if (STAGE == 0)
        a0 = ld(S.u0), b0 = ld(S.v0);
      else
        w = ld(S.vn);
      V2 kv;
      if (NF == 2 && S.mn1)
      {
        // Westervelt: kv = (b - mn1 v_n^2) / (m0 + mn1 u_n); u_n of interior dofs is still in x_l
        const V2 vs = (STAGE == 0) ? b0 : w;
        V2 us;
        if (!tail)
          us = reinterpret_cast<const V2*>(x_l)[i];
        else
          us[0] = x_l[2 * i], us[1] = T(0);
        const V2 m1 = ld(S.mn1);
        V2 den = ld(S.m0) + m1 * us;
        if (tail)
          den[1] = T(1);
        kv = (bv - m1 * vs * vs) / den;
      }
      else
      {
        mi = ld(S.minv);
        kv = bv * mi;
      }
      if (STAGE == 0)
      {
        st(S.u_, b0 * S.bdt + a0);
        st(S.v_, kv * S.bdt + b0);
        st(S.un, b0 * S.adt + a0);
        st(S.vn, kv * S.adt + b0);
      }
      else if (STAGE == 3)
      {
        au = ld(S.u_), av = ld(S.v_);
        st(S.u0, w * S.bdt + au);
        st(S.v0, kv * S.bdt + av);
      }
      else
      {
        au = ld(S.u_), av = ld(S.v_), a0 = ld(S.u0), b0 = ld(S.v0);
        st(S.u_, w * S.bdt + au);
        st(S.v_, kv * S.bdt + av);
        st(S.un, w * S.adt + a0);
        st(S.vn, kv * S.adt + b0);
      }
    }
  }
  for (int l = sh.nint + tid; l < sh.nloc; l += nthr)
    partial[sh_off + (l - sh.nint)] = y_l[l];
  FUS_TRACE_END(blk);
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
template <typename T, int STAGE>
__device__ __forceinline__ T stage_update_dof(int64_t s, T acc, const T* __restrict__ minv,
                                                 T* __restrict__ vn, T* __restrict__ un,
                                                 T* __restrict__ u0, T* __restrict__ v0,
                                                 T* __restrict__ u_, T* __restrict__ v_, T adt, T bdt,
                                                 const T* __restrict__ m0, const T* __restrict__ mn1)
{
#define FUS_LD(p) __builtin_nontemporal_load(&(p)[s])
#define FUS_ST(p, val) __builtin_nontemporal_store((val), &(p)[s])
  T kv;
  if (mn1)  // Westervelt (see StageArgs): stage inputs u_n, v_n are u0, v0 at stage 0
  {
    const T us = (STAGE == 0) ? u0[s] : un[s], vs = (STAGE == 0) ? v0[s] : vn[s];
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
  else if (STAGE == 3)
  {
    FUS_ST(u0, FUS_LD(vn) * bdt + FUS_LD(u_));
    vnext = kv * bdt + FUS_LD(v_);
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



// Shared dofs of one rank: fixed-order sum of the partials fused with the RK4 stage update of
// stage_update_dof; vectors are passed offset to the shared range.
template <typename T, int STAGE>
__global__ void __launch_bounds__(256)
k_shared_stage(int64_t n, const int32_t* __restrict__ sh_ptr, const int32_t* __restrict__ sh_pairs,
               const T* __restrict__ partial, const T* __restrict__ minv, T* __restrict__ vn,
               T* __restrict__ un, T* __restrict__ u0, T* __restrict__ v0, T* __restrict__ u_,
               T* __restrict__ v_, T adt, T bdt, const T* __restrict__ m0,
               const T* __restrict__ mn1, const BndNext<T> B)
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
  const T vnext = stage_update_dof<T, STAGE>(s, acc, minv, vn, un, u0, v0, u_, v_, adt, bdt, m0, mn1);
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
                  const int32_t* __restrict__ sh_pairs, const BndNext<T> B)
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
  const T vnext = stage_update_dof<T, STAGE>(u, acc, minv, vn, un, u0, v0, u_, v_, adt, bdt, m0, mn1);
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

// Streaming triad y = x + a z with 16-byte accesses: the measured device bandwidth the roofline
// fractions are also quoted against (fus_measure_bandwidth).
template <typename V>  // V = 2 doubles as an ext_vector (a template only so that every unit may include it)
__global__ void __launch_bounds__(256) k_triad(int64_t nvec, const V* __restrict__ x,
                                               const V* __restrict__ z, V* __restrict__ y, double a)
{
  constexpr int U = 4;  // independent 16-byte loads per array in flight per thread
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < nvec; i += U * stride)
  {
    V xv[U], zv[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      xv[u] = __builtin_nontemporal_load(x + i + u * stride);
      zv[u] = __builtin_nontemporal_load(z + i + u * stride);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      __builtin_nontemporal_store(xv[u] + a * zv[u], y + i + u * stride);
    }
  }
  for (; i < nvec; i += stride)
  {
    y[i] = x[i] + a * z[i];
  }
}

// halo helpers
// pack: sendbuf[k] = vec[idx[k]] over the concatenated neighbour lists
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
