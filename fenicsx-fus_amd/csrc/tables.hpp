// tables.hpp -- 1-D spectral-element tables on the caller's node order (host, double).
// Replaces Basix (third party, absent from the reference tree): make_quadrature(GLL) weights
// (spectral_op.hpp:160-162) and tabulate_1d's derivative block (precompute.hpp:217-234,
// spectral_op.hpp:168-170).  Definitions: SURVEY.md A.2.
#pragma once
#include <cmath>
#include <vector>

namespace fus
{
// Legendre P_n(x)
inline double legendre(int n, double x)
{
  double p0 = 1.0, p1 = x;
  if (n == 0)
    return 1.0;
  for (int k = 2; k <= n; ++k)
  {
    const double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
    p0 = p1, p1 = p2;
  }
  return p1;
}

// GLL weight on [0,1] of each node: w = 1 / (N (N-1) P_{N-1}(2x-1)^2)
inline std::vector<double> gll_weights_at(int N, const double* nodes)
{
  std::vector<double> w(N);
  for (int k = 0; k < N; ++k)
  {
    const double p = legendre(N - 1, 2.0 * nodes[k] - 1.0);
    w[k] = 1.0 / ((double)N * (N - 1) * p * p);
  }
  return w;
}

// D[q*N+i] = phi_i'(nodes[q]), Lagrange basis on `nodes`, barycentric form
inline std::vector<double> dphi_table(int N, const double* x)
{
  std::vector<double> lam(N), D((size_t)N * N);
  for (int i = 0; i < N; ++i)
  {
    double p = 1.0;
    for (int j = 0; j < N; ++j)
      if (j != i)
        p *= (x[i] - x[j]);
    lam[i] = 1.0 / p;
  }
  for (int q = 0; q < N; ++q)
  {
    double s = 0.0;
    for (int i = 0; i < N; ++i)
      if (i != q)
      {
        D[q * N + i] = (lam[i] / lam[q]) / (x[q] - x[i]);
        s += D[q * N + i];
      }
    D[q * N + q] = -s;
  }
  return D;
}

// true when `nodes` are the N GLL points of [0,1] in some order
inline bool is_gll_node_set(int N, const double* nodes)
{
  // x is a GLL node iff (1-xi^2) P'_{N-1}(xi) = 0  <=>  (N-1)(P_{N-2} - xi P_{N-1}) = 0
  for (int k = 0; k < N; ++k)
  {
    const double xi = 2.0 * nodes[k] - 1.0;
    const double r = legendre(N - 2, xi) - xi * legendre(N - 1, xi);
    if (std::fabs(r) > 1e-10)
      return false;
    for (int j = 0; j < k; ++j)
      if (std::fabs(nodes[j] - nodes[k]) < 1e-12)
        return false;
  }
  return true;
}
} // namespace fus
