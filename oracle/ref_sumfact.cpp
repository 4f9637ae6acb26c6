// ref_sumfact.cpp -- build recipe glue for oracle/_ref (TEST INFRASTRUCTURE).
//
// Compiles the reference's OWN dependency-free header
//   /root/reference/cpp/fenicsx-sf/common/sum_factorisation.hpp
// from where it lies (include path given by oracle/Makefile, never copied) and
// exposes fixed-size instantiations of its `contract` / `transpose` templates
// through a C ABI so tests can pin the oracle's primitives against the real
// reference code.  Only built where /root/reference exists.
#include <array>
#include <cstring>

#include "sum_factorisation.hpp"

namespace
{
template <int N>
int run_contract(int tr, const double* A, const double* B, double* C)
{
  if (tr)
    contract<double, N, N, N, N, true>(A, B, C);
  else
    contract<double, N, N, N, N, false>(A, B, C);
  return 0;
}
template <int N>
int run_transpose(int pattern, double* A, double* B)
{
  // the two strided copies spectral_op.hpp:200-210 uses
  if (pattern == 0)
    transpose<double, N, N, N, N, N * N, 1>(A, B);
  else
    transpose<double, N, N, N, 1, N, N * N>(A, B);
  return 0;
}
} // namespace

extern "C"
{
// C[a,{b,c}] += A[a,k] B[k,{b,c}] (tr=1) or A[k,a] B[k,{b,c}] (tr=0), all extents N
int ref_contract_f64(int N, int tr, const double* A, const double* B, double* C)
{
  switch (N)
  {
  case 2: return run_contract<2>(tr, A, B, C);
  case 3: return run_contract<3>(tr, A, B, C);
  case 4: return run_contract<4>(tr, A, B, C);
  case 5: return run_contract<5>(tr, A, B, C);
  case 6: return run_contract<6>(tr, A, B, C);
  case 7: return run_contract<7>(tr, A, B, C);
  case 8: return run_contract<8>(tr, A, B, C);
  default: return -1;
  }
}
int ref_transpose_f64(int N, int pattern, double* A, double* B)
{
  switch (N)
  {
  case 2: return run_transpose<2>(pattern, A, B);
  case 3: return run_transpose<3>(pattern, A, B);
  case 4: return run_transpose<4>(pattern, A, B);
  case 5: return run_transpose<5>(pattern, A, B);
  case 6: return run_transpose<6>(pattern, A, B);
  case 7: return run_transpose<7>(pattern, A, B);
  case 8: return run_transpose<8>(pattern, A, B);
  default: return -1;
  }
}
// the rectangular case of cpp/mwe/sum_factorisation/main.cpp:42-55 (M=3, N=2)
int ref_contract_mwe_f64(const double* A, const double* B, double* C, double* Ct)
{
  contract<double, 2, 3, 2, 2, true>(A, B, C);
  transpose<double, 3, 2, 2, 2, 1, 3 * 2>(C, Ct);
  return 0;
}
}
