// bw_probe.hip -- which 16-byte streaming kernel reaches the device's copy bandwidth (the guide quotes
// 6.29 TB/s for a float4 copy on MI355X)?  Development tool: hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip
// -o gpurun_out/bw_probe && gpurun_out/bw_probe.  Prints GB/s per variant (best of 10).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float F4 __attribute__((ext_vector_type(4)));

#define CHK(e)                                                                                      \
  do                                                                                                \
  {                                                                                                 \
    hipError_t r_ = (e);                                                                            \
    if (r_ != hipSuccess)                                                                           \
    {                                                                                               \
      printf("%s: %s\n", #e, hipGetErrorString(r_));                                                \
      exit(1);                                                                                      \
    }                                                                                               \
  } while (0)

template <int NT>
__global__ void __launch_bounds__(256) k_copy_flat(size_t n, const F4* __restrict__ x, F4* __restrict__ y)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n)
  {
    if (NT)
      __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i);
    else
      y[i] = x[i];
  }
}

template <int NT, int U>
__global__ void __launch_bounds__(256) k_copy_gs(size_t n, const F4* __restrict__ x, F4* __restrict__ y)
{
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride)
  {
    F4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      if (NT)
        __builtin_nontemporal_store(v[u], y + i + u * stride);
      else
        y[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride)
    y[i] = x[i];
}

// block-contiguous chunks: block b streams [b * chunk, (b + 1) * chunk) with U loads in flight per thread
template <int NT, int U>
__global__ void __launch_bounds__(256) k_copy_chunk(size_t n, const F4* __restrict__ x, F4* __restrict__ y)
{
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
  size_t i = b0 + threadIdx.x;
  for (; i + (U - 1) * 256 < b1; i += U * 256)
  {
    F4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = NT ? __builtin_nontemporal_load(x + i + u * 256) : x[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      if (NT)
        __builtin_nontemporal_store(v[u], y + i + u * 256);
      else
        y[i + u * 256] = v[u];
    }
  }
  for (; i < b1; i += 256)
    y[i] = x[i];
}

template <int NT, int U>
__global__ void __launch_bounds__(256) k_triad_gs(size_t n, const F4* __restrict__ x, const F4* __restrict__ z,
                                                  F4* __restrict__ y)
{
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride)
  {
    F4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      a[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
      b[u] = NT ? __builtin_nontemporal_load(z + i + u * stride) : z[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const F4 r = a[u] + 0.5f * b[u];
      if (NT)
        __builtin_nontemporal_store(r, y + i + u * stride);
      else
        y[i + u * stride] = r;
    }
  }
}

template <int NT, int U>
__global__ void __launch_bounds__(256) k_read_gs(size_t n, const F4* __restrict__ x, F4* __restrict__ y)
{
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  F4 acc = F4(0.f);
  for (; i + (U - 1) * stride < n; i += U * stride)
  {
#pragma unroll
    for (int u = 0; u < U; ++u)
      acc += NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
  }
  if (acc[0] == 12345.f)
    y[0] = acc;
}

int main(int argc, char** argv)
{
  const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 512) << 20;
  const size_t n = bytes / 16;
  F4 *x, *z, *y;
  CHK(hipMalloc(&x, bytes));
  CHK(hipMalloc(&z, bytes));
  CHK(hipMalloc(&y, bytes));
  CHK(hipMemset(x, 1, bytes));
  CHK(hipMemset(z, 1, bytes));
  CHK(hipMemset(y, 0, bytes));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  auto bench = [&](const char* name, int arrays, auto&& launch)
  {
    float best = 1e30f;
    for (int r = 0; r < 11; ++r)
    {
      CHK(hipEventRecord(e0, 0));
      launch();
      CHK(hipEventRecord(e1, 0));
      CHK(hipEventSynchronize(e1));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms < best)
        best = ms;
    }
    CHK(hipGetLastError());
    printf("%-34s %8.1f GB/s  (%.3f ms)\n", name, arrays * (double)bytes / (best * 1e-3) / 1e9, best);
  };
  const unsigned flat = (unsigned)((n + 255) / 256);
  bench("copy flat plain", 2, [&] { hipLaunchKernelGGL((k_copy_flat<0>), dim3(flat), dim3(256), 0, 0, n, x, y); });
  bench("copy flat nt", 2, [&] { hipLaunchKernelGGL((k_copy_flat<1>), dim3(flat), dim3(256), 0, 0, n, x, y); });
  for (unsigned g : {1024u, 2048u, 4096u, 8192u})
  {
    char nm[64];
    snprintf(nm, 64, "copy gs U4 plain grid %u", g);
    bench(nm, 2, [&] { hipLaunchKernelGGL((k_copy_gs<0, 4>), dim3(g), dim3(256), 0, 0, n, x, y); });
    snprintf(nm, 64, "copy gs U4 nt grid %u", g);
    bench(nm, 2, [&] { hipLaunchKernelGGL((k_copy_gs<1, 4>), dim3(g), dim3(256), 0, 0, n, x, y); });
    snprintf(nm, 64, "copy gs U8 plain grid %u", g);
    bench(nm, 2, [&] { hipLaunchKernelGGL((k_copy_gs<0, 8>), dim3(g), dim3(256), 0, 0, n, x, y); });
    snprintf(nm, 64, "copy gs U8 nt grid %u", g);
    bench(nm, 2, [&] { hipLaunchKernelGGL((k_copy_gs<1, 8>), dim3(g), dim3(256), 0, 0, n, x, y); });
    snprintf(nm, 64, "copy chunk U8 plain grid %u", g);
    bench(nm, 2, [&] { hipLaunchKernelGGL((k_copy_chunk<0, 8>), dim3(g), dim3(256), 0, 0, n, x, y); });
    snprintf(nm, 64, "copy chunk U8 nt grid %u", g);
    bench(nm, 2, [&] { hipLaunchKernelGGL((k_copy_chunk<1, 8>), dim3(g), dim3(256), 0, 0, n, x, y); });
    snprintf(nm, 64, "triad gs U4 plain grid %u", g);
    bench(nm, 3, [&] { hipLaunchKernelGGL((k_triad_gs<0, 4>), dim3(g), dim3(256), 0, 0, n, x, z, y); });
    snprintf(nm, 64, "triad gs U4 nt grid %u", g);
    bench(nm, 3, [&] { hipLaunchKernelGGL((k_triad_gs<1, 4>), dim3(g), dim3(256), 0, 0, n, x, z, y); });
    snprintf(nm, 64, "read gs U8 plain grid %u", g);
    bench(nm, 1, [&] { hipLaunchKernelGGL((k_read_gs<0, 8>), dim3(g), dim3(256), 0, 0, n, x, y); });
    snprintf(nm, 64, "read gs U8 nt grid %u", g);
    bench(nm, 1, [&] { hipLaunchKernelGGL((k_read_gs<1, 8>), dim3(g), dim3(256), 0, 0, n, x, y); });
  }
  return 0;
}
