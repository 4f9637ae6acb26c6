// geom.hpp -- trilinear hexahedron geometry shared by the device setup kernel and the host
// boundary-weight builder.  Restates, for degree-1 geometry, what the reference obtains from
// DOLFINx's CoordinateElement in compute_scaled_geometrical_factor /
// compute_scaled_jacobian_determinant (cpp/fenicsx-sf/common/precompute.hpp:101-213, 33-94):
// J_ij = sum_v x_v,i dphi_v/dX_j,  K = J^-1,  G = K K^T |det J| w,  detJw = |det J| w.
#pragma once

#if defined(__HIPCC__)
#define FUS_HD __host__ __device__
#else
#define FUS_HD
#endif

namespace fus
{

// cd[v][i]: coordinates of vertex v = vx + 2 vy + 4 vz.  X: reference point in [0,1]^3.
template <typename T>
FUS_HD inline void jacobian3(const T cd[8][3], double X0, double X1, double X2, T J[3][3])
{
  const double X[3] = {X0, X1, X2};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      J[i][j] = 0;
  for (int v = 0; v < 8; ++v)
  {
    T f[3], df[3];
    for (int d = 0; d < 3; ++d)
    {
      const int bit = (v >> d) & 1;
      f[d] = (T)(bit ? X[d] : 1.0 - X[d]);
      df[d] = (T)(bit ? 1.0 : -1.0);
    }
    const T g[3] = {df[0] * f[1] * f[2], f[0] * df[1] * f[2], f[0] * f[1] * df[2]};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        J[i][j] += cd[v][i] * g[j];
  }
}

// Second-order (27-node, triquadratic) hexahedron: nodes in TENSOR order n = nx + 3 ny + 9 nz with
// n_d in {0,1,2} <-> reference coordinate {0, 1/2, 1}; phi_n = l_nx(X0) l_ny(X1) l_nz(X2), the 1-D
// quadratic Lagrange basis.  (DOLFINx stores these nodes vertices-edges-faces-interior; the adapter
// permutes.)  precompute.hpp:52-55 tabulates the coordinate element of whatever order the mesh has.
template <typename T>
FUS_HD inline void jacobian3_q2(const T cd[27][3], double X0, double X1, double X2, T J[3][3])
{
  const double X[3] = {X0, X1, X2};
  T l[3][3], dl[3][3];
  for (int d = 0; d < 3; ++d)
  {
    const double x = X[d];
    l[d][0] = (T)((2.0 * x - 1.0) * (x - 1.0)), dl[d][0] = (T)(4.0 * x - 3.0);
    l[d][1] = (T)(4.0 * x * (1.0 - x)), dl[d][1] = (T)(4.0 - 8.0 * x);
    l[d][2] = (T)(x * (2.0 * x - 1.0)), dl[d][2] = (T)(4.0 * x - 1.0);
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      J[i][j] = 0;
  for (int nz = 0; nz < 3; ++nz)
    for (int ny = 0; ny < 3; ++ny)
      for (int nx = 0; nx < 3; ++nx)
      {
        const int n = nx + 3 * ny + 9 * nz;
        const T g[3] = {dl[0][nx] * l[1][ny] * l[2][nz], l[0][nx] * dl[1][ny] * l[2][nz],
                        l[0][nx] * l[1][ny] * dl[2][nz]};
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j)
            J[i][j] += cd[n][i] * g[j];
      }
}

// G6 = (xx, xy, xz, yy, yz, zz) of K K^T |det J| w; returns |det J| w  (precompute.hpp:191-208)
template <typename T>
FUS_HD inline T geometric_factor3(const T J[3][3], T w, T G6[6])
{
  const T c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const T c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const T c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const T det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  T K[3][3];
  K[0][0] = c00 / det;
  K[1][0] = c01 / det;
  K[2][0] = c02 / det;
  K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
  K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
  K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
  K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
  K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
  K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
  const T dw = (det < 0 ? -det : det) * w;
  int n = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 3; ++j)
      G6[n++] = dw * (K[i][0] * K[j][0] + K[i][1] * K[j][1] + K[i][2] * K[j][2]);
  return dw;
}

// ---- quadrilaterals (cpp/fenicsx-sf-naive/common/precompute.hpp, the 2-D operators of SURVEY a-5) ----
// cd[v][i]: coordinates of vertex v = vx + 2 vy (the third component is carried but unused).
template <typename T>
FUS_HD inline void jacobian2(const T cd[4][3], double X0, double X1, T J[2][2])
{
  const double X[2] = {X0, X1};
  J[0][0] = J[0][1] = J[1][0] = J[1][1] = 0;
  for (int v = 0; v < 4; ++v)
  {
    T f[2], df[2];
    for (int d = 0; d < 2; ++d)
    {
      const int bit = (v >> d) & 1;
      f[d] = (T)(bit ? X[d] : 1.0 - X[d]);
      df[d] = (T)(bit ? 1.0 : -1.0);
    }
    const T g[2] = {df[0] * f[1], f[0] * df[1]};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        J[i][j] += cd[v][i] * g[j];
  }
}

// Second-order (9-node, biquadratic) quadrilateral: nodes in tensor order n = nx + 3 ny, n_d in
// {0,1,2} <-> reference coordinate {0, 1/2, 1} (the `mesh_2` fixture of the naive 2-D operator test).
template <typename T>
FUS_HD inline void jacobian2_q2(const T cd[9][3], double X0, double X1, T J[2][2])
{
  const double X[2] = {X0, X1};
  T l[2][3], dl[2][3];
  for (int d = 0; d < 2; ++d)
  {
    const double x = X[d];
    l[d][0] = (T)((2.0 * x - 1.0) * (x - 1.0)), dl[d][0] = (T)(4.0 * x - 3.0);
    l[d][1] = (T)(4.0 * x * (1.0 - x)), dl[d][1] = (T)(4.0 - 8.0 * x);
    l[d][2] = (T)(x * (2.0 * x - 1.0)), dl[d][2] = (T)(4.0 * x - 1.0);
  }
  J[0][0] = J[0][1] = J[1][0] = J[1][1] = 0;
  for (int ny = 0; ny < 3; ++ny)
    for (int nx = 0; nx < 3; ++nx)
    {
      const int n = nx + 3 * ny;
      const T g[2] = {dl[0][nx] * l[1][ny], l[0][nx] * dl[1][ny]};
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          J[i][j] += cd[n][i] * g[j];
    }
}

// G3 = (xx, xy, yy) of K K^T |det J| w; returns |det J| w
template <typename T>
FUS_HD inline T geometric_factor2(const T J[2][2], T w, T G3[3])
{
  const T det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
  const T K00 = J[1][1] / det, K01 = -J[0][1] / det, K10 = -J[1][0] / det, K11 = J[0][0] / det;
  const T dw = (det < 0 ? -det : det) * w;
  G3[0] = dw * (K00 * K00 + K01 * K01);
  G3[1] = dw * (K00 * K10 + K01 * K11);
  G3[2] = dw * (K10 * K10 + K11 * K11);
  return dw;
}

} // namespace fus
