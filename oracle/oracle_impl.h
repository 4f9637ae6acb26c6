/*
 * oracle_impl.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE).
 * Included twice by oracle.c with REAL/SUF = double/f64 and float/f32.
 * See oracle.h for scope, parity status and the reference citations.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* sum_factorisation.hpp:43-49 -- strided 3-D copy B[offa a + offb b + offc c] = A[a,b,c] */
void FN(orc_transpose3)(int Na, int Nb, int Nc, int offa, int offb, int offc, const REAL* A,
                        REAL* B)
{
  for (int a = 0; a < Na; a++)
    for (int b = 0; b < Nb; b++)
      for (int c = 0; c < Nc; c++)
        B[offa * a + offb * b + offc * c] = A[(a * Nb + b) * Nc + c];
}

/* sum_factorisation.hpp:70-86 -- C[a,d] += A[a,k] B[k,d] (transpose!=0) or A[k,a] B[k,d];
 * loop nest k (outer), a, d as in the reference so the summation order over k matches. */
void FN(orc_contract)(int Nk, int Na, int Nb, int Nc, int transpose, const REAL* A, const REAL* B,
                      REAL* C)
{
  const int Nd = Nb * Nc;
  for (int k = 0; k < Nk; k++)
    for (int a = 0; a < Na; a++)
    {
      const REAL s = transpose ? A[a * Nk + k] : A[k * Na + a];
      for (int d = 0; d < Nd; d++)
        C[a * Nd + d] += s * B[k * Nd + d];
    }
}

/* naive sum_factorisation.hpp:10-18 */
void FN(orc_transpose2)(int Na, int Nb, int offa, int offb, const REAL* A, REAL* B)
{
  for (int a = 0; a < Na; ++a)
    for (int b = 0; b < Nb; ++b)
      B[a * offa + b * offb] = A[a * Nb + b];
}

/* naive sum_factorisation.hpp:27-37 -- C[a,b] += sum_k A[a,k] B[b,k] */
void FN(orc_contract2)(int Na, int Nb, int Nk, const REAL* A, const REAL* B, REAL* C)
{
  for (int a = 0; a < Na; ++a)
    for (int b = 0; b < Nb; ++b)
      for (int k = 0; k < Nk; ++k)
        C[a * Nb + b] += A[a * Nk + k] * B[b * Nk + k];
}

/* ---- geometry: precompute.hpp:33-94, 101-213 ------------------------------------------------
 * DOLFINx's CoordinateElement (third party) is restated for the degree-1 tensor cell:
 * phi_v(X) = prod_d (v_d ? X_d : 1 - X_d), J_ij = sum_v x_v,i dphi_v/dX_j (compute_jacobian),
 * K = J^-1 by cofactors, G = K K^T |det J| w (precompute.hpp:191-208).                        */
static void FN(jac3)(const REAL cd[8][3], double X0, double X1, double X2, REAL J[3][3])
{
  const double X[3] = {X0, X1, X2};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      J[i][j] = 0;
  for (int v = 0; v < 8; ++v)
  {
    const int bit[3] = {v & 1, (v >> 1) & 1, v >> 2};
    REAL f[3], df[3];
    for (int d = 0; d < 3; ++d)
    {
      f[d] = (REAL)(bit[d] ? X[d] : 1.0 - X[d]);
      df[d] = (REAL)(bit[d] ? 1.0 : -1.0);
    }
    const REAL g[3] = {df[0] * f[1] * f[2], f[0] * df[1] * f[2], f[0] * f[1] * df[2]};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        J[i][j] += cd[v][i] * g[j];
  }
}

static void FN(jac2)(const REAL cd[4][3], double X0, double X1, REAL J[2][2])
{
  const double X[2] = {X0, X1};
  J[0][0] = J[0][1] = J[1][0] = J[1][1] = 0;
  for (int v = 0; v < 4; ++v)
  {
    const int bit[2] = {v & 1, (v >> 1) & 1};
    REAL f[2], df[2];
    for (int d = 0; d < 2; ++d)
    {
      f[d] = (REAL)(bit[d] ? X[d] : 1.0 - X[d]);
      df[d] = (REAL)(bit[d] ? 1.0 : -1.0);
    }
    const REAL g[2] = {df[0] * f[1], f[0] * df[1]};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        J[i][j] += cd[v][i] * g[j];
  }
}

void FN(orc_geometry)(int tdim, int64_t ncells, const REAL* xg, const int32_t* xdofmap, int N,
                      const double* pts, const double* wts, REAL* G, REAL* detJ)
{
  if (tdim == 3)
  {
    const int Nd = N * N * N;
    for (int64_t c = 0; c < ncells; ++c)
    {
      REAL cd[8][3];
      for (int v = 0; v < 8; ++v)
        for (int j = 0; j < 3; ++j)
          cd[v][j] = xg[3 * (int64_t)xdofmap[c * 8 + v] + j];
      for (int q0 = 0; q0 < N; ++q0)
        for (int q1 = 0; q1 < N; ++q1)
          for (int q2 = 0; q2 < N; ++q2)
          {
            const int q = (q0 * N + q1) * N + q2;
            const REAL w = (REAL)(wts[q0] * wts[q1] * wts[q2]);
            REAL J[3][3], K[3][3];
            FN(jac3)(cd, pts[q0], pts[q1], pts[q2], J);
            const REAL c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
            const REAL c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
            const REAL c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
            const REAL det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
            K[0][0] = c00 / det;
            K[1][0] = c01 / det;
            K[2][0] = c02 / det;
            K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
            K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
            K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
            K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
            K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
            K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
            const REAL dw = (REAL)fabs((double)det) * w;
            if (detJ)
              detJ[c * Nd + q] = dw;
            if (G)
            {
              REAL* g = G + (c * Nd + q) * 6;
              int n = 0;
              for (int i = 0; i < 3; ++i)
                for (int j = i; j < 3; ++j)
                  g[n++] = dw * (K[i][0] * K[j][0] + K[i][1] * K[j][1] + K[i][2] * K[j][2]);
            }
          }
    }
  }
  else
  {
    const int Nd = N * N;
    for (int64_t c = 0; c < ncells; ++c)
    {
      REAL cd[4][3];
      for (int v = 0; v < 4; ++v)
        for (int j = 0; j < 3; ++j)
          cd[v][j] = xg[3 * (int64_t)xdofmap[c * 4 + v] + j];
      for (int q0 = 0; q0 < N; ++q0)
        for (int q1 = 0; q1 < N; ++q1)
        {
          const int q = q0 * N + q1;
          const REAL w = (REAL)(wts[q0] * wts[q1]);
          REAL J[2][2], K[2][2];
          FN(jac2)(cd, pts[q0], pts[q1], J);
          const REAL det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
          K[0][0] = J[1][1] / det;
          K[0][1] = -J[0][1] / det;
          K[1][0] = -J[1][0] / det;
          K[1][1] = J[0][0] / det;
          const REAL dw = (REAL)fabs((double)det) * w;
          if (detJ)
            detJ[c * Nd + q] = dw;
          if (G)
          {
            REAL* g = G + (c * Nd + q) * 3;
            g[0] = dw * (K[0][0] * K[0][0] + K[0][1] * K[0][1]);
            g[1] = dw * (K[0][0] * K[1][0] + K[0][1] * K[1][1]);
            g[2] = dw * (K[1][0] * K[1][0] + K[1][1] * K[1][1]);
          }
        }
    }
  }
}


/* ---- second-order (27-node) hexahedral geometry ---------------------------------------------------
 * precompute.hpp:52-55 tabulates the mesh's coordinate element whatever its degree; for degree 2
 * (the reference's Gmsh `mesh_2` fixtures, cpp/fenicsx-sf-naive/tests/test_operators3d) the map is
 * triquadratic.  Nodes in tensor order n = nx + 3 ny + 9 nz, n_d in {0,1,2} <-> X_d in {0,1/2,1}. */
static void FN(jac3_q2)(const REAL cd[27][3], double X0, double X1, double X2, REAL J[3][3])
{
  const double X[3] = {X0, X1, X2};
  REAL l[3][3], dl[3][3];
  for (int d = 0; d < 3; ++d)
  {
    const double x = X[d];
    l[d][0] = (REAL)((2.0 * x - 1.0) * (x - 1.0)), dl[d][0] = (REAL)(4.0 * x - 3.0);
    l[d][1] = (REAL)(4.0 * x * (1.0 - x)), dl[d][1] = (REAL)(4.0 - 8.0 * x);
    l[d][2] = (REAL)(x * (2.0 * x - 1.0)), dl[d][2] = (REAL)(4.0 * x - 1.0);
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      J[i][j] = 0;
  for (int nz = 0; nz < 3; ++nz)
    for (int ny = 0; ny < 3; ++ny)
      for (int nx = 0; nx < 3; ++nx)
      {
        const int n = nx + 3 * ny + 9 * nz;
        const REAL g[3] = {dl[0][nx] * l[1][ny] * l[2][nz], l[0][nx] * dl[1][ny] * l[2][nz],
                           l[0][nx] * l[1][ny] * dl[2][nz]};
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j)
            J[i][j] += cd[n][i] * g[j];
      }
}

void FN(orc_geometry_q2)(int64_t ncells, const REAL* xg, const int32_t* xdofmap, int N,
                         const double* pts, const double* wts, REAL* G, REAL* detJ)
{
  const int Nd = N * N * N;
  for (int64_t c = 0; c < ncells; ++c)
  {
    REAL cd[27][3];
    for (int v = 0; v < 27; ++v)
      for (int j = 0; j < 3; ++j)
        cd[v][j] = xg[3 * (int64_t)xdofmap[c * 27 + v] + j];
    for (int q0 = 0; q0 < N; ++q0)
      for (int q1 = 0; q1 < N; ++q1)
        for (int q2 = 0; q2 < N; ++q2)
        {
          const int q = (q0 * N + q1) * N + q2;
          const REAL w = (REAL)(wts[q0] * wts[q1] * wts[q2]);
          REAL J[3][3], K[3][3];
          FN(jac3_q2)(cd, pts[q0], pts[q1], pts[q2], J);
          const REAL c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
          const REAL c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
          const REAL c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
          const REAL det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
          K[0][0] = c00 / det, K[1][0] = c01 / det, K[2][0] = c02 / det;
          K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
          K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
          K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
          K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
          K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
          K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
          const REAL dw = (REAL)fabs((double)det) * w;
          if (detJ)
            detJ[c * Nd + q] = dw;
          if (G)
          {
            REAL* g = G + (c * Nd + q) * 6;
            int n = 0;
            for (int i = 0; i < 3; ++i)
              for (int j = i; j < 3; ++j)
                g[n++] = dw * (K[i][0] * K[j][0] + K[i][1] * K[j][1] + K[i][2] * K[j][2]);
          }
        }
  }
}

void FN(orc_facet_diag_q2)(int64_t nfacets, const int32_t* facet_cell, const int32_t* facet_local,
                           const REAL* cellcoef, const REAL* xg, const int32_t* xdofmap, int N,
                           const double* pts, const double* wts, const int32_t* tensor_dofmap,
                           REAL* out)
{
  static const int axis3[6] = {2, 1, 0, 0, 1, 2}, side3[6] = {0, 0, 0, 1, 1, 1};
  const int Nd = N * N * N;
  int i_lo = 0, i_hi = 0;
  for (int i = 0; i < N; ++i)
  {
    if (pts[i] < pts[i_lo])
      i_lo = i;
    if (pts[i] > pts[i_hi])
      i_hi = i;
  }
  for (int64_t f = 0; f < nfacets; ++f)
  {
    const int64_t c = facet_cell[f];
    const int ax = axis3[facet_local[f]], sd = side3[facet_local[f]];
    const int d1 = (ax + 1) % 3, d2 = (ax + 2) % 3;
    REAL cd[27][3];
    for (int v = 0; v < 27; ++v)
      for (int j = 0; j < 3; ++j)
        cd[v][j] = xg[3 * (int64_t)xdofmap[c * 27 + v] + j];
    for (int a = 0; a < N; ++a)
      for (int b = 0; b < N; ++b)
      {
        int idx[3];
        idx[ax] = sd ? i_hi : i_lo, idx[d1] = a, idx[d2] = b;
        REAL J[3][3];
        FN(jac3_q2)(cd, pts[idx[0]], pts[idx[1]], pts[idx[2]], J);
        const REAL t1[3] = {J[0][d1], J[1][d1], J[2][d1]}, t2[3] = {J[0][d2], J[1][d2], J[2][d2]};
        const REAL n0 = t1[1] * t2[2] - t1[2] * t2[1], n1 = t1[2] * t2[0] - t1[0] * t2[2],
                   n2 = t1[0] * t2[1] - t1[1] * t2[0];
        const REAL area = (REAL)sqrt((double)(n0 * n0 + n1 * n1 + n2 * n2));
        const int li = (idx[0] * N + idx[1]) * N + idx[2];
        out[tensor_dofmap[c * Nd + li]] += cellcoef[c] * area * (REAL)(wts[a] * wts[b]);
      }
  }
}

/* Second-order (9-node, biquadratic) quadrilaterals: the 2-D counterpart of jac3_q2 for the
 * reference's `mesh_2` fixture of cpp/fenicsx-sf-naive/tests/test_operators2d (main.cpp:31, G = 2).
 * Nodes in tensor order n = nx + 3 ny, n_d in {0,1,2} <-> X_d in {0,1/2,1}. */
static void FN(jac2_q2)(const REAL cd[9][3], double X0, double X1, REAL J[2][2])
{
  const double X[2] = {X0, X1};
  REAL l[2][3], dl[2][3];
  for (int d = 0; d < 2; ++d)
  {
    const double x = X[d];
    l[d][0] = (REAL)((2.0 * x - 1.0) * (x - 1.0)), dl[d][0] = (REAL)(4.0 * x - 3.0);
    l[d][1] = (REAL)(4.0 * x * (1.0 - x)), dl[d][1] = (REAL)(4.0 - 8.0 * x);
    l[d][2] = (REAL)(x * (2.0 * x - 1.0)), dl[d][2] = (REAL)(4.0 * x - 1.0);
  }
  J[0][0] = J[0][1] = J[1][0] = J[1][1] = 0;
  for (int ny = 0; ny < 3; ++ny)
    for (int nx = 0; nx < 3; ++nx)
    {
      const int n = nx + 3 * ny;
      const REAL g[2] = {dl[0][nx] * l[1][ny], l[0][nx] * dl[1][ny]};
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          J[i][j] += cd[n][i] * g[j];
    }
}

void FN(orc_geometry_q2_2d)(int64_t ncells, const REAL* xg, const int32_t* xdofmap, int N,
                            const double* pts, const double* wts, REAL* G, REAL* detJ)
{
  const int Nd = N * N;
  for (int64_t c = 0; c < ncells; ++c)
  {
    REAL cd[9][3];
    for (int v = 0; v < 9; ++v)
      for (int j = 0; j < 3; ++j)
        cd[v][j] = xg[3 * (int64_t)xdofmap[c * 9 + v] + j];
    for (int q0 = 0; q0 < N; ++q0)
      for (int q1 = 0; q1 < N; ++q1)
      {
        const int q = q0 * N + q1;
        const REAL w = (REAL)(wts[q0] * wts[q1]);
        REAL J[2][2], K[2][2];
        FN(jac2_q2)(cd, pts[q0], pts[q1], J);
        const REAL det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        K[0][0] = J[1][1] / det, K[0][1] = -J[0][1] / det, K[1][0] = -J[1][0] / det, K[1][1] = J[0][0] / det;
        const REAL dw = (REAL)fabs((double)det) * w;
        if (detJ)
          detJ[c * Nd + q] = dw;
        if (G)
        {
          REAL* g = G + (c * Nd + q) * 3;
          g[0] = dw * (K[0][0] * K[0][0] + K[0][1] * K[0][1]);
          g[1] = dw * (K[0][0] * K[1][0] + K[0][1] * K[1][1]);
          g[2] = dw * (K[1][0] * K[1][0] + K[1][1] * K[1][1]);
        }
      }
  }
}

void FN(orc_facet_diag_q2_2d)(int64_t nfacets, const int32_t* facet_cell, const int32_t* facet_local,
                              const REAL* cellcoef, const REAL* xg, const int32_t* xdofmap, int N,
                              const double* pts, const double* wts, const int32_t* tensor_dofmap,
                              REAL* out)
{
  static const int axis2[4] = {1, 0, 0, 1}, side2[4] = {0, 0, 1, 1};
  const int Nd = N * N;
  int i_lo = 0, i_hi = 0;
  for (int i = 0; i < N; ++i)
  {
    if (pts[i] < pts[i_lo])
      i_lo = i;
    if (pts[i] > pts[i_hi])
      i_hi = i;
  }
  for (int64_t f = 0; f < nfacets; ++f)
  {
    const int64_t c = facet_cell[f];
    const int ax = axis2[facet_local[f]], sd = side2[facet_local[f]], d1 = 1 - ax;
    REAL cd[9][3];
    for (int v = 0; v < 9; ++v)
      for (int j = 0; j < 3; ++j)
        cd[v][j] = xg[3 * (int64_t)xdofmap[c * 9 + v] + j];
    for (int a = 0; a < N; ++a)
    {
      int idx[2];
      idx[ax] = sd ? i_hi : i_lo, idx[d1] = a;
      REAL J[2][2];
      FN(jac2_q2)(cd, pts[idx[0]], pts[idx[1]], J);
      const REAL len = (REAL)sqrt((double)(J[0][d1] * J[0][d1] + J[1][d1] * J[1][d1]));
      out[tensor_dofmap[c * Nd + idx[0] * N + idx[1]]] += cellcoef[c] * len * (REAL)wts[a];
    }
  }
}

/* spectral_op.hpp:69-86 with mass::transform :19-26 (identical in the naive 2-D class) */
void FN(orc_mass)(int tdim, int64_t ncells, int N, const int32_t* tensor_dofmap, const REAL* detJ,
                  const REAL* coeffs, const REAL* x, REAL* y)
{
  const int Nd = (tdim == 3) ? N * N * N : N * N;
  REAL* x_ = (REAL*)malloc(sizeof(REAL) * Nd);
  for (int64_t c = 0; c < ncells; ++c)
  {
    const int32_t* dm = tensor_dofmap + c * Nd;
    for (int i = 0; i < Nd; ++i)
      x_[i] = x[dm[i]];
    const REAL* sdetJ = detJ + c * Nd;
    const REAL coeff = coeffs[c];
    for (int iq = 0; iq < Nd; ++iq)
      x_[iq] = coeff * x_[iq] * sdetJ[iq];
    for (int i = 0; i < Nd; ++i)
      y[dm[i]] += x_[i];
  }
  free(x_);
}

/* spectral_op.hpp:173-243: same call sequence (zero-fill, contract, transposes) per cell */
void FN(orc_stiffness3d)(int64_t ncells, int N, const int32_t* tensor_dofmap, const REAL* G,
                         const REAL* dphi, const REAL* coeffs, const REAL* x, REAL* y)
{
  const int Nd = N * N * N;
  const size_t nb = sizeof(REAL) * Nd;
  REAL* buf = (REAL*)malloc(nb * 11);
  REAL *x_ = buf, *fw0 = buf + Nd, *fw1 = buf + 2 * Nd, *fw2 = buf + 3 * Nd, *y0 = buf + 4 * Nd,
       *y1 = buf + 5 * Nd, *y2 = buf + 6 * Nd, *T1 = buf + 7 * Nd, *T2 = buf + 8 * Nd,
       *T3 = buf + 9 * Nd, *T4 = buf + 10 * Nd;
  for (int64_t c = 0; c < ncells; ++c)
  {
    const int32_t* dm = tensor_dofmap + c * Nd;
    for (int i = 0; i < Nd; ++i)
      x_[i] = x[dm[i]];

    memset(T1, 0, nb), memset(T2, 0, nb), memset(T3, 0, nb), memset(T4, 0, nb);

    /* derivative along tensor index 0 (:194-196) */
    memset(fw0, 0, nb);
    FN(orc_contract)(N, N, N, N, 1, dphi, x_, fw0);
    /* along index 1 (:199-203) */
    memset(fw1, 0, nb);
    FN(orc_transpose3)(N, N, N, N, N * N, 1, x_, T1);
    FN(orc_contract)(N, N, N, N, 1, dphi, T1, T2);
    FN(orc_transpose3)(N, N, N, N, N * N, 1, T2, fw1);
    /* along index 2 (:206-210) */
    memset(fw2, 0, nb);
    FN(orc_transpose3)(N, N, N, 1, N, N * N, x_, T3);
    FN(orc_contract)(N, N, N, N, 1, dphi, T3, T4);
    FN(orc_transpose3)(N, N, N, 1, N, N * N, T4, fw2);

    /* stiffness::transform (:113-130) */
    const REAL* Gc = G + c * Nd * 6;
    const REAL coeff = coeffs[c];
    for (int iq = 0; iq < Nd; ++iq)
    {
      const REAL* _G = Gc + iq * 6;
      const REAL w0 = fw0[iq], w1 = fw1[iq], w2 = fw2[iq];
      fw0[iq] = coeff * (_G[0] * w0 + _G[1] * w1 + _G[2] * w2);
      fw1[iq] = coeff * (_G[1] * w0 + _G[3] * w1 + _G[4] * w2);
      fw2[iq] = coeff * (_G[2] * w0 + _G[4] * w1 + _G[5] * w2);
    }

    memset(T1, 0, nb), memset(T2, 0, nb), memset(T3, 0, nb), memset(T4, 0, nb);

    /* transposed contractions (:222-238) */
    memset(y0, 0, nb);
    FN(orc_contract)(N, N, N, N, 0, dphi, fw0, y0);
    memset(y1, 0, nb);
    FN(orc_transpose3)(N, N, N, N, N * N, 1, fw1, T1);
    FN(orc_contract)(N, N, N, N, 0, dphi, T1, T2);
    FN(orc_transpose3)(N, N, N, N, N * N, 1, T2, y1);
    memset(y2, 0, nb);
    FN(orc_transpose3)(N, N, N, 1, N, N * N, fw2, T3);
    FN(orc_contract)(N, N, N, N, 0, dphi, T3, T4);
    FN(orc_transpose3)(N, N, N, 1, N, N * N, T4, y2);

    for (int i = 0; i < Nd; ++i)
      y[dm[i]] += y0[i] + y1[i] + y2[i];
  }
  free(buf);
}

/* naive spectral_op.hpp:273-323 with the 2-D transform :195-207 (G index reversed:
 * the derivative along the LAST tensor index pairs with G[2]) */
void FN(orc_stiffness2d)(int64_t ncells, int N, const int32_t* tensor_dofmap, const REAL* G,
                         const REAL* dphi, const REAL* coeffs, const REAL* x, REAL* y)
{
  const int Nd = N * N;
  const size_t nb = sizeof(REAL) * Nd;
  REAL* buf = (REAL*)malloc(nb * 8);
  REAL *x_ = buf, *fw0 = buf + Nd, *fw1 = buf + 2 * Nd, *y0 = buf + 3 * Nd, *y1 = buf + 4 * Nd,
       *T1 = buf + 5 * Nd, *T2 = buf + 6 * Nd, *dphiT = buf + 7 * Nd;
  FN(orc_transpose2)(N, N, 1, N, dphi, dphiT); /* ctor :268-270 */
  for (int64_t c = 0; c < ncells; ++c)
  {
    const int32_t* dm = tensor_dofmap + c * Nd;
    for (int i = 0; i < Nd; ++i)
      x_[i] = x[dm[i]];
    memset(T1, 0, nb), memset(T2, 0, nb);
    memset(fw0, 0, nb);
    FN(orc_contract2)(N, N, N, x_, dphi, fw0);
    memset(fw1, 0, nb);
    FN(orc_transpose2)(N, N, 1, N, x_, T1);
    FN(orc_contract2)(N, N, N, T1, dphi, T2);
    FN(orc_transpose2)(N, N, 1, N, T2, fw1);

    const REAL* Gc = G + c * Nd * 3;
    const REAL coeff = coeffs[c];
    for (int iq = 0; iq < Nd; ++iq)
    {
      const REAL* _G = Gc + iq * 3;
      const REAL w0 = fw0[iq], w1 = fw1[iq];
      fw0[iq] = coeff * (_G[2] * w0 + _G[1] * w1);
      fw1[iq] = coeff * (_G[1] * w0 + _G[0] * w1);
    }

    memset(T1, 0, nb), memset(T2, 0, nb);
    memset(y0, 0, nb);
    FN(orc_contract2)(N, N, N, fw0, dphiT, y0);
    memset(y1, 0, nb);
    FN(orc_transpose2)(N, N, 1, N, fw1, T1);
    FN(orc_contract2)(N, N, N, T1, dphiT, T2);
    FN(orc_transpose2)(N, N, 1, N, T2, y1);
    for (int i = 0; i < Nd; ++i)
      y[dm[i]] += y0[i] + y1[i];
  }
  free(buf);
}

/* Dense-table cross-check (fenicsx-pc formulation): y_i += sum_q grad phi_i(q) . c G_q grad u(q)
 * with the full 3-D derivative table built from the 1-D one; O(N^6) per cell. */
void FN(orc_stiffness3d_dense)(int64_t ncells, int N, const int32_t* tensor_dofmap, const REAL* G,
                               const REAL* dphi, const REAL* coeffs, const REAL* x, REAL* y)
{
  const int Nd = N * N * N;
  /* dtab[d][q][i] = d phi_i / dX_d at point q; phi_i = l_i0 l_i1 l_i2, l_a(x_b) = delta_ab */
  REAL* dtab = (REAL*)calloc((size_t)3 * Nd * Nd, sizeof(REAL));
  for (int q0 = 0; q0 < N; ++q0)
    for (int q1 = 0; q1 < N; ++q1)
      for (int q2 = 0; q2 < N; ++q2)
        for (int i0 = 0; i0 < N; ++i0)
          for (int i1 = 0; i1 < N; ++i1)
            for (int i2 = 0; i2 < N; ++i2)
            {
              const size_t q = (q0 * N + q1) * N + q2, i = (i0 * N + i1) * N + i2;
              dtab[(0 * Nd + q) * Nd + i] = dphi[q0 * N + i0] * (i1 == q1) * (i2 == q2);
              dtab[(1 * Nd + q) * Nd + i] = (i0 == q0) * dphi[q1 * N + i1] * (i2 == q2);
              dtab[(2 * Nd + q) * Nd + i] = (i0 == q0) * (i1 == q1) * dphi[q2 * N + i2];
            }
  REAL* xe = (REAL*)malloc(sizeof(REAL) * Nd);
  REAL* ye = (REAL*)malloc(sizeof(REAL) * Nd);
  for (int64_t c = 0; c < ncells; ++c)
  {
    const int32_t* dm = tensor_dofmap + c * Nd;
    for (int i = 0; i < Nd; ++i)
      xe[i] = x[dm[i]], ye[i] = 0;
    for (int q = 0; q < Nd; ++q)
    {
      REAL gu[3] = {0, 0, 0};
      for (int d = 0; d < 3; ++d)
        for (int i = 0; i < Nd; ++i)
          gu[d] += dtab[((size_t)d * Nd + q) * Nd + i] * xe[i];
      const REAL* g = G + (c * Nd + q) * 6;
      const REAL w[3] = {coeffs[c] * (g[0] * gu[0] + g[1] * gu[1] + g[2] * gu[2]),
                         coeffs[c] * (g[1] * gu[0] + g[3] * gu[1] + g[4] * gu[2]),
                         coeffs[c] * (g[2] * gu[0] + g[4] * gu[1] + g[5] * gu[2])};
      for (int d = 0; d < 3; ++d)
        for (int i = 0; i < Nd; ++i)
          ye[i] += dtab[((size_t)d * Nd + q) * Nd + i] * w[d];
    }
    for (int i = 0; i < Nd; ++i)
      y[dm[i]] += ye[i];
  }
  free(dtab), free(xe), free(ye);
}

/* Facet diagonal weights.  Local facet -> (fixed axis, side). */
void FN(orc_facet_diag)(int tdim, int64_t nfacets, const int32_t* facet_cell,
                        const int32_t* facet_local, const REAL* cellcoef, const REAL* xg,
                        const int32_t* xdofmap, int N, const double* pts, const double* wts,
                        const int32_t* tensor_dofmap, REAL* out)
{
  /* endpoint node indices in the given 1-D node order */
  int i_lo = 0, i_hi = 0;
  for (int i = 0; i < N; ++i)
  {
    if (pts[i] < pts[i_lo])
      i_lo = i;
    if (pts[i] > pts[i_hi])
      i_hi = i;
  }
  if (tdim == 3)
  {
    static const int axis3[6] = {2, 1, 0, 0, 1, 2}, side3[6] = {0, 0, 0, 1, 1, 1};
    const int Nd = N * N * N;
    for (int64_t f = 0; f < nfacets; ++f)
    {
      const int64_t c = facet_cell[f];
      const int ax = axis3[facet_local[f]], sd = side3[facet_local[f]];
      const int d1 = (ax + 1) % 3, d2 = (ax + 2) % 3;
      REAL cd[8][3];
      for (int v = 0; v < 8; ++v)
        for (int j = 0; j < 3; ++j)
          cd[v][j] = xg[3 * (int64_t)xdofmap[c * 8 + v] + j];
      for (int a = 0; a < N; ++a)
        for (int b = 0; b < N; ++b)
        {
          int idx[3];
          double X[3];
          idx[ax] = sd ? i_hi : i_lo;
          idx[d1] = a;
          idx[d2] = b;
          for (int d = 0; d < 3; ++d)
            X[d] = pts[idx[d]];
          REAL J[3][3];
          FN(jac3)(cd, X[0], X[1], X[2], J);
          const REAL t1[3] = {J[0][d1], J[1][d1], J[2][d1]}, t2[3] = {J[0][d2], J[1][d2], J[2][d2]};
          const REAL n0 = t1[1] * t2[2] - t1[2] * t2[1], n1 = t1[2] * t2[0] - t1[0] * t2[2],
                     n2 = t1[0] * t2[1] - t1[1] * t2[0];
          const REAL area = (REAL)sqrt((double)(n0 * n0 + n1 * n1 + n2 * n2));
          const int li = (idx[0] * N + idx[1]) * N + idx[2];
          out[tensor_dofmap[c * Nd + li]] += cellcoef[c] * area * (REAL)(wts[a] * wts[b]);
        }
    }
  }
  else
  {
    static const int axis2[4] = {1, 0, 0, 1}, side2[4] = {0, 0, 1, 1};
    const int Nd = N * N;
    for (int64_t f = 0; f < nfacets; ++f)
    {
      const int64_t c = facet_cell[f];
      const int ax = axis2[facet_local[f]], sd = side2[facet_local[f]];
      const int d1 = 1 - ax;
      REAL cd[4][3];
      for (int v = 0; v < 4; ++v)
        for (int j = 0; j < 3; ++j)
          cd[v][j] = xg[3 * (int64_t)xdofmap[c * 4 + v] + j];
      for (int a = 0; a < N; ++a)
      {
        int idx[2];
        idx[ax] = sd ? i_hi : i_lo;
        idx[d1] = a;
        REAL J[2][2];
        FN(jac2)(cd, pts[idx[0]], pts[idx[1]], J);
        const REAL len = (REAL)sqrt((double)(J[0][d1] * J[0][d1] + J[1][d1] * J[1][d1]));
        out[tensor_dofmap[c * Nd + idx[0] * N + idx[1]]] += cellcoef[c] * len * (REAL)wts[a];
      }
    }
  }
}

/* Linear.hpp:21-38 */
static void FN(k_copy)(int64_t n, const REAL* in, REAL* out) { memcpy(out, in, sizeof(REAL) * n); }
static void FN(k_axpy)(int64_t n, REAL* r, REAL alpha, const REAL* x, const REAL* y)
{
  for (int64_t i = 0; i < n; ++i)
    r[i] = x[i] * alpha + y[i];
}

/* Linear.hpp:161-314.  Single process: scatter_fwd/scatter_rev (:196,199,206) are no-ops.
 * The FFCx facet assembly (:205) is the diagonal form  b += g(t) src - absb .* v_n. */
/* order: 4 = classical RK4 (Linear.hpp:263-265); 1, 2, 3 = forward Euler / Ralston tables of the
 * Python reference's rk() (python/src/fenicsxfus/_linear.py:286-311, loop :461-499). */
int64_t FN(orc_linear_rk)(int order, int tdim, int64_t ncells, int64_t ndofs, int N,
                          const int32_t* tensor_dofmap, const REAL* G, const REAL* dphi,
                          const REAL* coeff, const REAL* m, const REAL* src, const REAL* absb,
                          double freq_, double p0_, double s0_, double t0, double tf_, double dt_,
                          REAL* u_n, REAL* v_n)
{
  const REAL freq = (REAL)freq_, p0 = (REAL)p0_, s0 = (REAL)s0_;
  const REAL w0 = (REAL)(2 * M_PI * freq_);
  const REAL period = (REAL)(1.0 / freq_), window_length = (REAL)4.0;
  const size_t nb = sizeof(REAL) * ndofs;
  REAL *u_ = (REAL*)malloc(nb), *v_ = (REAL*)malloc(nb), *un = (REAL*)malloc(nb),
       *vn = (REAL*)malloc(nb), *u0 = (REAL*)malloc(nb), *v0 = (REAL*)malloc(nb),
       *ku = (REAL*)malloc(nb), *kv = (REAL*)malloc(nb), *b = (REAL*)malloc(nb),
       *g = (REAL*)malloc(nb), *uw = (REAL*)malloc(nb), *vw = (REAL*)malloc(nb);
  REAL t = (REAL)t0, tf = (REAL)tf_, dt = (REAL)dt_;
  int64_t step = 0;
  FN(k_copy)(ndofs, u_n, u_), FN(k_copy)(ndofs, v_n, v_);
  FN(k_copy)(ndofs, u_, ku), FN(k_copy)(ndofs, v_, kv);
  REAL a_runge[4] = {0.0, 0.5, 0.5, 1.0};
  REAL b_runge[4] = {(REAL)(1.0 / 6.0), (REAL)(1.0 / 3.0), (REAL)(1.0 / 3.0), (REAL)(1.0 / 6.0)};
  REAL c_runge[4] = {0.0, 0.5, 0.5, 1.0};
  if (order == 1)
    a_runge[0] = 0, b_runge[0] = 1, c_runge[0] = 0;
  else if (order == 2)
  {
    a_runge[1] = (REAL)(2.0 / 3.0), b_runge[0] = (REAL)(1.0 / 4.0), b_runge[1] = (REAL)(3.0 / 4.0);
    c_runge[1] = (REAL)(2.0 / 3.0);
  }
  else if (order == 3)
  {
    a_runge[1] = (REAL)(1.0 / 2.0), a_runge[2] = (REAL)(3.0 / 4.0);
    b_runge[0] = (REAL)(2.0 / 9.0), b_runge[1] = (REAL)(1.0 / 3.0), b_runge[2] = (REAL)(4.0 / 9.0);
    c_runge[1] = (REAL)(1.0 / 2.0), c_runge[2] = (REAL)(3.0 / 4.0);
  }
  while (t < tf)
  {
    dt = (dt < tf - t) ? dt : tf - t;
    FN(k_copy)(ndofs, u_, u0), FN(k_copy)(ndofs, v_, v0);
    for (int i = 0; i < order; i++)
    {
      FN(k_copy)(ndofs, u0, un), FN(k_copy)(ndofs, v0, vn);
      FN(k_axpy)(ndofs, un, dt * a_runge[i], ku, un);
      FN(k_axpy)(ndofs, vn, dt * a_runge[i], kv, vn);
      const REAL tn = t + c_runge[i] * dt;
      /* f0 (:171-174) */
      FN(k_copy)(ndofs, vn, ku);
      /* f1 (:181-222) */
      {
        REAL window;
        if (tn < period * window_length)
          window = (REAL)(0.5 * (1.0 - cos((double)(freq * (REAL)M_PI * tn / window_length))));
        else
          window = 1.0;
        const REAL gval = window * p0 * w0 / s0 * (REAL)cos((double)(w0 * tn));
        for (int64_t k = 0; k < ndofs; ++k)
          g[k] = gval;
        FN(k_copy)(ndofs, un, uw);
        FN(k_copy)(ndofs, vn, vw);
        for (int64_t k = 0; k < ndofs; ++k)
          b[k] = 0;
        if (tdim == 3)
          FN(orc_stiffness3d)(ncells, N, tensor_dofmap, G, dphi, coeff, uw, b);
        else
          FN(orc_stiffness2d)(ncells, N, tensor_dofmap, G, dphi, coeff, uw, b);
        for (int64_t k = 0; k < ndofs; ++k)
          b[k] += g[k] * src[k] - absb[k] * vw[k];
        for (int64_t k = 0; k < ndofs; ++k)
          kv[k] = b[k] / m[k];
      }
      FN(k_axpy)(ndofs, u_, dt * b_runge[i], ku, u_);
      FN(k_axpy)(ndofs, v_, dt * b_runge[i], kv, v_);
    }
    t += dt;
    step += 1;
  }
  FN(k_copy)(ndofs, u_, u_n), FN(k_copy)(ndofs, v_, v_n);
  free(u_), free(v_), free(un), free(vn), free(u0), free(v0), free(ku), free(kv), free(b), free(g),
      free(uw), free(vw);
  return step;
}


int64_t FN(orc_linear_rk4)(int tdim, int64_t ncells, int64_t ndofs, int N,
                           const int32_t* tensor_dofmap, const REAL* G, const REAL* dphi,
                           const REAL* coeff, const REAL* m, const REAL* src, const REAL* absb,
                           double freq_, double p0_, double s0_, double t0, double tf_, double dt_,
                           REAL* u_n, REAL* v_n)
{
  return FN(orc_linear_rk)(4, tdim, ncells, ndofs, N, tensor_dofmap, G, dphi, coeff, m, src, absb,
                           freq_, p0_, s0_, t0, tf_, dt_, u_n, v_n);
}

/* Lossy.hpp:176-342: init + rk4 of the lossy (viscoelastic) model.  f1 (:196-250) applies TWO
 * stiffness actions per stage, lin_op on u_n with -1/rho and att_op on v_n with -delta/(rho c^2)
 * (:231-232, coefficients :166-169), and uses the heterogeneous-domain source scaling
 * g = 2 W p0 w0/s0 cos(w0 t) with its time derivative dg (:216-220).  The FFCx facet assembly
 * (:233, forms BM7-SC1/forms.py:40-42) is the diagonal form
 *   b += g src - absb .* v_n + dg src2,
 * src = (1/rho) w_f on tag 1, absb = (1/(rho c)) w_f on every boundary facet,
 * src2 = (delta/(rho c^2)) w_f on tag 1; m includes the (delta/(rho c^3)) w_f boundary term. */
/* source_scale: 2 = the live "heterogenous domain" branch of Lossy.hpp:216-220; 1 = the Python
 * package (python/src/fenicsxfus/_lossy.py:186-189), which also keeps the absorbing and delta-mass
 * terms on tag 2 only -- that choice is in the vectors the caller passes (absb, m). */
int64_t FN(orc_lossy_rk4_s)(int tdim, int64_t ncells, int64_t ndofs, int N,
                            const int32_t* tensor_dofmap, const REAL* G, const REAL* dphi,
                            const REAL* lin_coeff, const REAL* att_coeff, const REAL* m,
                            const REAL* src, const REAL* absb, const REAL* src2, double freq_,
                            double p0_, double s0_, double t0, double tf_, double dt_, REAL* u_n,
                            REAL* v_n, double source_scale)
{
  const REAL two = (REAL)source_scale;
  const REAL freq = (REAL)freq_, p0 = (REAL)p0_, s0 = (REAL)s0_;
  const REAL w0 = (REAL)(2 * M_PI * freq_);
  const REAL period = (REAL)(1.0 / freq_), window_length = (REAL)4.0;
  const size_t nb = sizeof(REAL) * ndofs;
  REAL *u_ = (REAL*)malloc(nb), *v_ = (REAL*)malloc(nb), *un = (REAL*)malloc(nb),
       *vn = (REAL*)malloc(nb), *u0 = (REAL*)malloc(nb), *v0 = (REAL*)malloc(nb),
       *ku = (REAL*)malloc(nb), *kv = (REAL*)malloc(nb), *b = (REAL*)malloc(nb),
       *uw = (REAL*)malloc(nb), *vw = (REAL*)malloc(nb);
  REAL t = (REAL)t0, tf = (REAL)tf_, dt = (REAL)dt_;
  int64_t step = 0;
  FN(k_copy)(ndofs, u_n, u_), FN(k_copy)(ndofs, v_n, v_);
  FN(k_copy)(ndofs, u_, ku), FN(k_copy)(ndofs, v_, kv);
  const REAL a_runge[4] = {0.0, 0.5, 0.5, 1.0};
  const REAL b_runge[4] = {(REAL)(1.0 / 6.0), (REAL)(1.0 / 3.0), (REAL)(1.0 / 3.0),
                           (REAL)(1.0 / 6.0)};
  const REAL c_runge[4] = {0.0, 0.5, 0.5, 1.0};
  while (t < tf)
  {
    dt = (dt < tf - t) ? dt : tf - t;
    FN(k_copy)(ndofs, u_, u0), FN(k_copy)(ndofs, v_, v0);
    for (int i = 0; i < 4; i++)
    {
      FN(k_copy)(ndofs, u0, un), FN(k_copy)(ndofs, v0, vn);
      FN(k_axpy)(ndofs, un, dt * a_runge[i], ku, un);
      FN(k_axpy)(ndofs, vn, dt * a_runge[i], kv, vn);
      const REAL tn = t + c_runge[i] * dt;
      FN(k_copy)(ndofs, vn, ku);
      {
        REAL window, dwindow;
        if (tn < period * window_length)
        {
          window = (REAL)(0.5 * (1.0 - cos((double)(freq * (REAL)M_PI * tn / window_length))));
          dwindow = (REAL)(0.5 * M_PI) * freq / window_length
                    * (REAL)sin((double)(freq * (REAL)M_PI * tn / window_length));
        }
        else
          window = 1.0, dwindow = 0.0;
        const REAL gval = window * two * p0 * w0 / s0 * (REAL)cos((double)(w0 * tn));
        const REAL dgval = dwindow * two * p0 * w0 / s0 * (REAL)cos((double)(w0 * tn))
                           - window * two * p0 * w0 * w0 / s0 * (REAL)sin((double)(w0 * tn));
        FN(k_copy)(ndofs, un, uw);
        FN(k_copy)(ndofs, vn, vw);
        for (int64_t k = 0; k < ndofs; ++k)
          b[k] = 0;
        if (tdim == 3)
        {
          FN(orc_stiffness3d)(ncells, N, tensor_dofmap, G, dphi, lin_coeff, uw, b);
          FN(orc_stiffness3d)(ncells, N, tensor_dofmap, G, dphi, att_coeff, vw, b);
        }
        else
        {
          FN(orc_stiffness2d)(ncells, N, tensor_dofmap, G, dphi, lin_coeff, uw, b);
          FN(orc_stiffness2d)(ncells, N, tensor_dofmap, G, dphi, att_coeff, vw, b);
        }
        for (int64_t k = 0; k < ndofs; ++k)
          b[k] += gval * src[k] - absb[k] * vw[k] + dgval * src2[k];
        for (int64_t k = 0; k < ndofs; ++k)
          kv[k] = b[k] / m[k];
      }
      FN(k_axpy)(ndofs, u_, dt * b_runge[i], ku, u_);
      FN(k_axpy)(ndofs, v_, dt * b_runge[i], kv, v_);
    }
    t += dt;
    step += 1;
  }
  FN(k_copy)(ndofs, u_, u_n), FN(k_copy)(ndofs, v_, v_n);
  free(u_), free(v_), free(un), free(vn), free(u0), free(v0), free(ku), free(kv), free(b), free(uw),
      free(vw);
  return step;
}


int64_t FN(orc_lossy_rk4)(int tdim, int64_t ncells, int64_t ndofs, int N,
                          const int32_t* tensor_dofmap, const REAL* G, const REAL* dphi,
                          const REAL* lin_coeff, const REAL* att_coeff, const REAL* m,
                          const REAL* src, const REAL* absb, const REAL* src2, double freq_,
                          double p0_, double s0_, double t0, double tf_, double dt_, REAL* u_n,
                          REAL* v_n)
{
  return FN(orc_lossy_rk4_s)(tdim, ncells, ndofs, N, tensor_dofmap, G, dphi, lin_coeff, att_coeff, m,
                             src, absb, src2, freq_, p0_, s0_, t0, tf_, dt_, u_n, v_n, 2.0);
}

/* Westervelt.hpp:196-373: init + rk4 of the nonlinear (Westervelt) model.  f1 (:216-281) is the
 * lossy f1 plus two diagonal mass actions per stage: the LHS is re-assembled as
 * m = m0 + M(nlin1) u_n  (:249-257, nlin1 = -2 beta/(rho^2 c^4), :185) and the RHS gains
 * M(nlin2) (v_n .* v_n)  (:246-247, :263, nlin2 = +2 beta/(rho^2 c^4), :186).  detJ is the scaled
 * Jacobian determinant the mass operator uses (spectral_op.hpp:80-81).  Other arguments as
 * orc_lossy_rk4 (m0 = its m). */
int64_t FN(orc_westervelt_rk4_s)(int tdim, int64_t ncells, int64_t ndofs, int N,
                                 const int32_t* tensor_dofmap, const REAL* G, const REAL* detJ,
                                 const REAL* dphi, const REAL* lin_coeff, const REAL* att_coeff,
                                 const REAL* nlin1_coeff, const REAL* nlin2_coeff, const REAL* m0,
                                 const REAL* src, const REAL* absb, const REAL* src2, double freq_,
                                 double p0_, double s0_, double t0, double tf_, double dt_,
                                 REAL* u_n, REAL* v_n, double source_scale)
{
  const REAL two = (REAL)source_scale;
  const REAL freq = (REAL)freq_, p0 = (REAL)p0_, s0 = (REAL)s0_;
  const REAL w0 = (REAL)(2 * M_PI * freq_);
  const REAL period = (REAL)(1.0 / freq_), window_length = (REAL)4.0;
  const size_t nb = sizeof(REAL) * ndofs;
  REAL *u_ = (REAL*)malloc(nb), *v_ = (REAL*)malloc(nb), *un = (REAL*)malloc(nb),
       *vn = (REAL*)malloc(nb), *u0 = (REAL*)malloc(nb), *v0 = (REAL*)malloc(nb),
       *ku = (REAL*)malloc(nb), *kv = (REAL*)malloc(nb), *b = (REAL*)malloc(nb),
       *uw = (REAL*)malloc(nb), *vw = (REAL*)malloc(nb), *ww = (REAL*)malloc(nb),
       *m = (REAL*)malloc(nb);
  REAL t = (REAL)t0, tf = (REAL)tf_, dt = (REAL)dt_;
  int64_t step = 0;
  FN(k_copy)(ndofs, u_n, u_), FN(k_copy)(ndofs, v_n, v_);
  FN(k_copy)(ndofs, u_, ku), FN(k_copy)(ndofs, v_, kv);
  const REAL a_runge[4] = {0.0, 0.5, 0.5, 1.0};
  const REAL b_runge[4] = {(REAL)(1.0 / 6.0), (REAL)(1.0 / 3.0), (REAL)(1.0 / 3.0),
                           (REAL)(1.0 / 6.0)};
  const REAL c_runge[4] = {0.0, 0.5, 0.5, 1.0};
  while (t < tf)
  {
    dt = (dt < tf - t) ? dt : tf - t;
    FN(k_copy)(ndofs, u_, u0), FN(k_copy)(ndofs, v_, v0);
    for (int i = 0; i < 4; i++)
    {
      FN(k_copy)(ndofs, u0, un), FN(k_copy)(ndofs, v0, vn);
      FN(k_axpy)(ndofs, un, dt * a_runge[i], ku, un);
      FN(k_axpy)(ndofs, vn, dt * a_runge[i], kv, vn);
      const REAL tn = t + c_runge[i] * dt;
      FN(k_copy)(ndofs, vn, ku);
      {
        REAL window, dwindow;
        if (tn < period * window_length)
        {
          window = (REAL)(0.5 * (1.0 - cos((double)(freq * (REAL)M_PI * tn / window_length))));
          dwindow = (REAL)(0.5 * M_PI) * freq / window_length
                    * (REAL)sin((double)(freq * (REAL)M_PI * tn / window_length));
        }
        else
          window = 1.0, dwindow = 0.0;
        const REAL gval = window * two * p0 * w0 / s0 * (REAL)cos((double)(w0 * tn));
        const REAL dgval = dwindow * two * p0 * w0 / s0 * (REAL)cos((double)(w0 * tn))
                           - window * two * p0 * w0 * w0 / s0 * (REAL)sin((double)(w0 * tn));
        FN(k_copy)(ndofs, un, uw);
        FN(k_copy)(ndofs, vn, vw);
        for (int64_t k = 0; k < ndofs; ++k)
          ww[k] = vw[k] * vw[k];
        /* LHS (:249-257) */
        for (int64_t k = 0; k < ndofs; ++k)
          m[k] = 0;
        FN(orc_mass)(tdim, ncells, N, tensor_dofmap, detJ, nlin1_coeff, uw, m);
        for (int64_t k = 0; k < ndofs; ++k)
          m[k] = m0[k] + m[k];
        /* RHS (:260-265) */
        for (int64_t k = 0; k < ndofs; ++k)
          b[k] = 0;
        if (tdim == 3)
        {
          FN(orc_stiffness3d)(ncells, N, tensor_dofmap, G, dphi, lin_coeff, uw, b);
          FN(orc_stiffness3d)(ncells, N, tensor_dofmap, G, dphi, att_coeff, vw, b);
        }
        else
        {
          FN(orc_stiffness2d)(ncells, N, tensor_dofmap, G, dphi, lin_coeff, uw, b);
          FN(orc_stiffness2d)(ncells, N, tensor_dofmap, G, dphi, att_coeff, vw, b);
        }
        FN(orc_mass)(tdim, ncells, N, tensor_dofmap, detJ, nlin2_coeff, ww, b);
        for (int64_t k = 0; k < ndofs; ++k)
          b[k] += gval * src[k] - absb[k] * vw[k] + dgval * src2[k];
        for (int64_t k = 0; k < ndofs; ++k)
          kv[k] = b[k] / m[k];
      }
      FN(k_axpy)(ndofs, u_, dt * b_runge[i], ku, u_);
      FN(k_axpy)(ndofs, v_, dt * b_runge[i], kv, v_);
    }
    t += dt;
    step += 1;
  }
  FN(k_copy)(ndofs, u_, u_n), FN(k_copy)(ndofs, v_, v_n);
  free(u_), free(v_), free(un), free(vn), free(u0), free(v0), free(ku), free(kv), free(b), free(uw),
      free(vw), free(ww), free(m);
  return step;
}

int64_t FN(orc_westervelt_rk4)(int tdim, int64_t ncells, int64_t ndofs, int N,
                               const int32_t* tensor_dofmap, const REAL* G, const REAL* detJ,
                               const REAL* dphi, const REAL* lin_coeff, const REAL* att_coeff,
                               const REAL* nlin1_coeff, const REAL* nlin2_coeff, const REAL* m0,
                               const REAL* src, const REAL* absb, const REAL* src2, double freq_,
                               double p0_, double s0_, double t0, double tf_, double dt_,
                               REAL* u_n, REAL* v_n)
{
  return FN(orc_westervelt_rk4_s)(tdim, ncells, ndofs, N, tensor_dofmap, G, detJ, dphi, lin_coeff,
                                  att_coeff, nlin1_coeff, nlin2_coeff, m0, src, absb, src2, freq_, p0_,
                                  s0_, t0, tf_, dt_, u_n, v_n, 2.0);
}


/* Threaded variant of orc_linear_rk4 for the CPU baseline: the reference runs
 * one MPI rank per core on an element-wise partition; here the cells (ordered slowest-axis-major, as
 * BoxMesh emits them) are cut into `nslabs` contiguous slabs, one OpenMP thread per slab with
 * private scratch.  Slabs of equal parity touch disjoint DOFs, so the operator runs as two
 * barrier-separated passes (even slabs, odd slabs) -- the role of the interface scatter_rev.  The
 * vector passes are split statically over the threads.  Same arithmetic per cell as the serial
 * loop; only the order in which interface-plane contributions are added differs. */
int64_t FN(orc_linear_rk4_mt)(int64_t ncells, int64_t ndofs, int N, const int32_t* tensor_dofmap,
                              const REAL* G, const REAL* dphi, const REAL* coeff, const REAL* m,
                              const REAL* src, const REAL* absb, double freq_, double p0_,
                              double s0_, double t0, double tf_, double dt_, REAL* u_n, REAL* v_n,
                              int nslabs, const int64_t* slab_cell_off)
{
  const REAL freq = (REAL)freq_, p0 = (REAL)p0_, s0 = (REAL)s0_;
  const REAL w0 = (REAL)(2 * M_PI * freq_);
  const REAL period = (REAL)(1.0 / freq_), window_length = (REAL)4.0;
  const size_t nb = sizeof(REAL) * ndofs;
  const int Nd = N * N * N;
  REAL *u_ = (REAL*)malloc(nb), *v_ = (REAL*)malloc(nb), *un = (REAL*)malloc(nb),
       *vn = (REAL*)malloc(nb), *u0 = (REAL*)malloc(nb), *v0 = (REAL*)malloc(nb),
       *kv = (REAL*)malloc(nb), *b = (REAL*)malloc(nb);
  REAL t = (REAL)t0, tf = (REAL)tf_, dt = (REAL)dt_;
  int64_t step = 0;
  memcpy(u_, u_n, nb), memcpy(v_, v_n, nb);
  memcpy(kv, v_, nb);
  const REAL a_runge[4] = {0.0, 0.5, 0.5, 1.0};
  const REAL b_runge[4] = {(REAL)(1.0 / 6.0), (REAL)(1.0 / 3.0), (REAL)(1.0 / 3.0),
                           (REAL)(1.0 / 6.0)};
  const REAL c_runge[4] = {0.0, 0.5, 0.5, 1.0};
  while (t < tf)
  {
    dt = (dt < tf - t) ? dt : tf - t;
#pragma omp parallel num_threads(nslabs)
    {
#pragma omp for schedule(static)
      for (int64_t k = 0; k < ndofs; ++k)
        u0[k] = u_[k], v0[k] = v_[k];
      for (int i = 0; i < 4; i++)
      {
        const REAL adt = dt * a_runge[i], bdt = dt * b_runge[i];
        const REAL tn = t + c_runge[i] * dt;
        REAL window;
        if (tn < period * window_length)
          window = (REAL)(0.5 * (1.0 - cos((double)(freq * (REAL)M_PI * tn / window_length))));
        else
          window = 1.0;
        const REAL gval = window * p0 * w0 / s0 * (REAL)cos((double)(w0 * tn));
        /* un = u0 + a dt ku (ku = previous vn), vn = v0 + a dt kv; b = 0  (Linear.hpp:279-283,203) */
#pragma omp for schedule(static)
        for (int64_t k = 0; k < ndofs; ++k)
        {
          const REAL ku = (i == 0) ? (REAL)0 : vn[k];
          un[k] = ku * adt + u0[k];
          vn[k] = kv[k] * adt + v0[k];
          b[k] = 0;
        }
        /* operator: even slabs, then odd slabs */
        for (int parity = 0; parity < 2; ++parity)
        {
#pragma omp for schedule(static, 1)
          for (int sl = 0; sl < nslabs; ++sl)
            if ((sl & 1) == parity && slab_cell_off[sl + 1] > slab_cell_off[sl])
              FN(orc_stiffness3d)(slab_cell_off[sl + 1] - slab_cell_off[sl], N,
                                  tensor_dofmap + slab_cell_off[sl] * Nd,
                                  G + slab_cell_off[sl] * Nd * 6, dphi, coeff + slab_cell_off[sl], un, b);
        }
        /* boundary terms, divide, accumulate (Linear.hpp:205,212-221,293-294; ku = vn) */
#pragma omp for schedule(static)
        for (int64_t k = 0; k < ndofs; ++k)
        {
          const REAL bk = b[k] + gval * src[k] - absb[k] * vn[k];
          kv[k] = bk / m[k];
          u_[k] = vn[k] * bdt + u_[k];
          v_[k] = kv[k] * bdt + v_[k];
        }
      }
    }
    t += dt;
    step += 1;
  }
  memcpy(u_n, u_, nb), memcpy(v_n, v_, nb);
  free(u_), free(v_), free(un), free(vn), free(u0), free(v0), free(kv), free(b);
  return step;
}

#undef FN
#undef CAT
#undef CAT_
