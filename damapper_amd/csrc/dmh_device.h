// dmh_device.h -- small device-side helpers shared by several kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmh {

// First index of the run of equal codes containing i (sorted array): gallop + bisect.
__device__ __forceinline__ int64_t run_start(const uint64_t *__restrict__ k, int64_t i)
{ const uint64_t c = k[i];
  int64_t hi = i, lo, step = 1;
  while (true)
    { int64_t t = hi - step;
      if (t < 0) { lo = -1; break; }
      if (k[t] == c) { hi = t; step <<= 1; }
      else { lo = t; break; }
    }
  while (hi - lo > 1)
    { int64_t m = (lo + hi) >> 1;
      if (k[m] == c) hi = m; else lo = m;
    }
  return hi;
}

// One past the last index of the run of equal codes containing i.
__device__ __forceinline__ int64_t run_end(const uint64_t *__restrict__ k, int64_t n, int64_t i)
{ const uint64_t c = k[i];
  int64_t lo = i, hi, step = 1;
  while (true)
    { int64_t t = lo + step;
      if (t >= n) { hi = n; break; }
      if (k[t] == c) { lo = t; step <<= 1; }
      else { hi = t; break; }
    }
  while (hi - lo > 1)
    { int64_t m = (lo + hi) >> 1;
      if (k[m] == c) lo = m; else hi = m;
    }
  return hi;
}

// Inclusive scan across a wavefront.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, int lane)
{
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
    { uint32_t y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
  return x;
}

// The same on the vector ALU alone (data-parallel-primitive moves: four shifts inside the rows of 16 lanes, then lane 15
// of row 0 / 2 added to row 1 / 3 and lane 31 to rows 2 and 3) -- a __shfl_up goes through the LDS crossbar
// (ds_bpermute), which the kernels that use this one are bound by.
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t x)
{ x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x111, 0xf, 0xf, true);      // row_shr:1 (0 shifted in)
  x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x112, 0xf, 0xf, true);      // row_shr:2
  x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x114, 0xf, 0xf, true);      // row_shr:4
  x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x118, 0xf, 0xf, true);      // row_shr:8
  x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
  x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
  return x;
}

// the 64-bit value of the lane below / above (lane 0 / lane 63: 0)
__device__ __forceinline__ uint64_t wave_from_below(uint64_t v)
{ const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) v, 0x138, 0xf, 0xf, true);            // wave_shr:1
  const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (v >> 32), 0x138, 0xf, 0xf, true);
  return ((uint64_t) hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_from_above(uint64_t v)
{ const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) v, 0x130, 0xf, 0xf, true);            // wave_shl:1
  const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (v >> 32), 0x130, 0xf, 0xf, true);
  return ((uint64_t) hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_lane_u64(uint64_t v, int l)
{ return ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), l) << 32)
         | (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, l);
}

// Exclusive scan across a 256-thread block; *total receives the block sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *ws, uint32_t *total)
{ int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t x = wave_incl_scan(v, lane);
  if (lane == 63) ws[w] = x;
  __syncthreads();
  uint32_t wbase = 0;
  for (int i = 0; i < w; i++) wbase += ws[i];
  *total = ws[0] + ws[1] + ws[2] + ws[3];
  __syncthreads();
  return wbase + x - v;
}

}  // namespace dmh
