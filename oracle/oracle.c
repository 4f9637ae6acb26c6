/*
 * oracle.c -- CPU restatement of the fenicsx-fus hot path (TEST INFRASTRUCTURE, see oracle.h).
 * Plain C; built by oracle/Makefile into oracle/liboracle.so.
 */
#define _USE_MATH_DEFINES
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Legendre P_n(x) and P_{n-1}(x) by the three-term recurrence. */
static void legendre(int n, double x, double* pn, double* pnm1)
{
  double p0 = 1.0, p1 = x;
  if (n == 0)
  {
    *pn = 1.0, *pnm1 = 0.0;
    return;
  }
  for (int k = 2; k <= n; ++k)
  {
    const double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
    p0 = p1, p1 = p2;
  }
  *pn = p1, *pnm1 = p0;
}

/* x_q = roots of (1 - xi^2) P'_{N-1}(xi) mapped to [0,1];
 * w_q = 1 / (N (N-1) P_{N-1}(xi_q)^2)  (already halved for [0,1]). */
void orc_gll(int N, double* pts, double* wts)
{
  const int n = N - 1;
  for (int k = 0; k < N; ++k)
  {
    double x = -cos(M_PI * k / n);
    if (k == 0)
      x = -1.0;
    else if (k == n)
      x = 1.0;
    else
    {
      for (int it = 0; it < 100; ++it)
      {
        double pn, pnm1;
        legendre(n, x, &pn, &pnm1);
        /* (1-x^2) P_n' = n (P_{n-1} - x P_n);  Newton on f = P_{n-1} - x P_n,
         * f' = P'_{n-1} - P_n - x P_n' = -(n+1) P_n */
        const double dx = (pnm1 - x * pn) / ((n + 1) * pn);
        x += dx;
        if (fabs(dx) < 1e-16)
          break;
      }
    }
    double pn, pnm1;
    legendre(n, x, &pn, &pnm1);
    pts[k] = 0.5 * (x + 1.0);
    wts[k] = 1.0 / ((double)N * n * pn * pn);
  }
  /* symmetrise */
  for (int k = 0; k < N / 2; ++k)
  {
    const double p = 0.5 * (pts[k] + (1.0 - pts[N - 1 - k]));
    const double w = 0.5 * (wts[k] + wts[N - 1 - k]);
    pts[k] = p, pts[N - 1 - k] = 1.0 - p;
    wts[k] = w, wts[N - 1 - k] = w;
  }
  if (N % 2)
    pts[N / 2] = 0.5;
}

void orc_gll_weights_at(int N, const double* nodes, double* wts)
{
  const int n = N - 1;
  for (int k = 0; k < N; ++k)
  {
    double pn, pnm1;
    legendre(n, 2.0 * nodes[k] - 1.0, &pn, &pnm1);
    wts[k] = 1.0 / ((double)N * n * pn * pn);
  }
}

/* Barycentric differentiation matrix on arbitrary distinct nodes. */
void orc_dphi(int N, const double* x, double* D)
{
  double* lam = (double*)malloc(sizeof(double) * N);
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
  free(lam);
}

#define REAL double
#define SUF f64
#include "oracle_impl.h"
#undef REAL
#undef SUF

#define REAL float
#define SUF f32
#include "oracle_impl.h"
#undef REAL
#undef SUF
