// dev_trace.hpp -- development builds only (build.py --dev -DFUS_TRACE; never part of libfusmi.so).
// Phase timestamps of k_block_op: per block 8 slots of the 100 MHz wall clock -- 0 start, 1 prologue
// done, 2 trips done, 3 epilogue done; 4 = CU id; 5-7 = first trips.  Read back by tools/gpu_trace.py
// through fus_debug_trace.
#pragma once
__device__ unsigned long long g_fus_trace[65536 * 8];
#define FUS_STAMP(blk, k)                                                                          \
  do                                                                                               \
  {                                                                                                \
    if (threadIdx.x == 0 && (blk) < 65536)                                                         \
      g_fus_trace[(size_t)(blk) * 8 + (k)] = wall_clock64();                                       \
  } while (0)
#define FUS_TRACE_END(blk)                                                                         \
  do                                                                                               \
  {                                                                                                \
    __syncthreads();                                                                               \
    FUS_STAMP(blk, 3);                                                                             \
    if (threadIdx.x == 0 && (blk) < 65536)                                                         \
      g_fus_trace[(size_t)(blk) * 8 + 4] = __smid();                                               \
  } while (0)
