/* cabi_min.c -- smallest consumer of libfusmi's C ABI (plain C, no C++/Python/torch types).
 * Builds a 2x2x2 hex mesh of degree 2 by hand, runs the host-only layout check, and -- when a HIP
 * device is present -- creates the operator data and applies the stiffness operator to x = 1
 * (K 1 = 0).  Compile:  gcc -Iinclude examples/cabi_min.c -Lfenicsx-fus_amd/fenicsxfus_amd -lfusmi
 * (tests/test_abi_cpu.py builds and runs it). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fusmi.h"

int main(void)
{
  enum { n = 2, P = 2, N = P + 1, Nd = N * N * N, nc = n * n * n, nd1 = n * P + 1, nv1 = n + 1 };
  static int32_t dofmap[nc * Nd], gdm[nc * 8];
  static double x[nv1 * nv1 * nv1 * 3], cen[nc * 3];
  const double nodes[N] = {0.0, 0.5, 1.0}; /* GLL points of degree 2 on [0,1] */
  for (int i = 0; i < nv1; ++i)
    for (int j = 0; j < nv1; ++j)
      for (int k = 0; k < nv1; ++k)
      {
        const int v = (i * nv1 + j) * nv1 + k;
        x[3 * v] = i / (double)n, x[3 * v + 1] = j / (double)n, x[3 * v + 2] = k / (double)n;
      }
  for (int cx = 0; cx < n; ++cx)
    for (int cy = 0; cy < n; ++cy)
      for (int cz = 0; cz < n; ++cz)
      {
        const int c = (cx * n + cy) * n + cz;
        for (int v = 0; v < 8; ++v) /* vertex v = vx + 2 vy + 4 vz */
          gdm[c * 8 + v] = ((cx + (v & 1)) * nv1 + cy + ((v >> 1) & 1)) * nv1 + cz + (v >> 2);
        for (int a = 0; a < N; ++a)
          for (int b = 0; b < N; ++b)
            for (int d = 0; d < N; ++d)
              dofmap[c * Nd + (a * N + b) * N + d] = ((cx * P + a) * nd1 + cy * P + b) * nd1 + cz * P + d;
        cen[3 * c] = (cx + 0.5) / n, cen[3 * c + 1] = (cy + 0.5) / n, cen[3 * c + 2] = (cz + 0.5) / n;
      }
  const int64_t ndofs = (int64_t)nd1 * nd1 * nd1;
  int64_t info[8];
  if (fus_layout_check(P, nc, ndofs, dofmap, cen, 4, 1, info) != FUS_OK)
  {
    fprintf(stderr, "layout check failed: %s\n", fus_last_error());
    return 1;
  }
  printf("layout ok: %lld blocks, %lld interior + %lld shared dofs\n", (long long)info[0], (long long)info[1],
         (long long)info[2]);
  fus_ctx* ctx = NULL;
  if (fus_init(0, &ctx) != FUS_OK)
  {
    printf("no device: %s\n", fus_last_error()); /* expected on a CPU-only host */
    return 0;
  }
  fus_op* op = NULL;
  if (fus_op_create(ctx, 3, P, FUS_F64, nc, ndofs, dofmap, nodes, x, nv1 * nv1 * nv1, gdm, 1, &op) != FUS_OK)
  {
    fprintf(stderr, "op_create failed: %s\n", fus_last_error());
    return 1;
  }
  double *xv = malloc(sizeof(double) * ndofs), *yv = calloc(ndofs, sizeof(double)), coef[nc];
  for (int64_t i = 0; i < ndofs; ++i)
    xv[i] = 1.0;
  for (int c = 0; c < nc; ++c)
    coef[c] = -1.0;
  if (fus_stiffness_apply(op, xv, coef, yv, FUS_HOST) != FUS_OK)
  {
    fprintf(stderr, "apply failed: %s\n", fus_last_error());
    return 1;
  }
  double mx = 0;
  for (int64_t i = 0; i < ndofs; ++i)
    mx = fmax(mx, fabs(yv[i]));
  printf("max |K 1| = %.3e\n", mx);
  fus_op_destroy(op);
  fus_finalize(ctx);
  free(xv), free(yv);
  return mx < 1e-12 ? 0 : 1;
}
