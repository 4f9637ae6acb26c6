// Lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by brute force: wave (la, lb) runs the instruction with
// A = 1 at lane la only and B = 1 at lane lb only; out[(la * 64 + lb) * 64 + l] = D of lane l.  tools/mfma4_probe.py
// reads the dump and prints the bit fields of (block, i, k) in A, (block, k, j) in B and (block, i, j) in D.
// Also checks __builtin_amdgcn_permlane32_swap (lane l <-> lane l ^ 32).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma4_probe.hip -o tools/_bin/mfma4_probe && tools/_bin/mfma4_probe > dump.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_probe(double* out)
{
  const int la = blockIdx.x / 64, lb = blockIdx.x % 64, l = threadIdx.x;
  const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[(size_t)blockIdx.x * 64 + l] = d;
}

__global__ void k_swap(int* out)
{
  const int l = threadIdx.x;
  int x = l, y = 1000 + l;
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
  auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  out[l] = r[0], out[64 + l] = r[1];
#else
  out[l] = -1, out[64 + l] = -1;
#endif
  out[128 + l] = __shfl_xor(l, 32, 64);
}

int main()
{
  double* d;
  hipMalloc(&d, sizeof(double) * 4096 * 64);
  hipLaunchKernelGGL(k_probe, dim3(4096), dim3(64), 0, 0, d);
  std::vector<double> h(4096 * 64);
  hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0)
          printf("A %d B %d D %d\n", la, lb, l);
  int* s;
  hipMalloc(&s, sizeof(int) * 192);
  hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, s);
  int hs[192];
  hipMemcpy(hs, s, sizeof(hs), hipMemcpyDeviceToHost);
  printf("SWAP r0:");
  for (int l = 0; l < 64; ++l)
    printf(" %d", hs[l]);
  printf("\nSWAP r1:");
  for (int l = 0; l < 64; ++l)
    printf(" %d", hs[64 + l]);
  printf("\nXOR32:");
  for (int l = 0; l < 64; ++l)
    printf(" %d", hs[128 + l]);
  printf("\n");
  return 0;
}
