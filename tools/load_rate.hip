// Calibration: how many bytes per clock can ONE CU pull into LDS with the GEMM staging pattern
// (256 threads, 8 x global_load_dwordx4 per thread per step -> ds_write_b128 -> barrier), no MFMA?
//   pattern 0: k-contiguous rows of a big [N x 1024] matrix (128-B row segments, HBM stream)   -- the x operand
//   pattern 1: a 1 MB matrix re-read by every workgroup (L2-resident)                          -- the W1 operand
//   pattern 2: m-contiguous [32 rows x 512 B] slabs of the big matrix                          -- the TN operands
// Build: hipcc --offload-arch=gfx950 -O3 tools/load_rate.hip -o gpurun_out/load_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k(const float* __restrict__ big, const float* __restrict__ small, float* out,
                                         int pattern, int steps, int nrows) {
  __shared__ __align__(16) float lds[2][9216];
  const int tid = threadIdx.x;
  float4 r[8];
  float acc = 0.f;
  const int tile = blockIdx.x;
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int f = tid + i * 256;
      const float* p;
      if (pattern == 0) {        // 256 rows x 32 floats, rows 4 KB apart
        int row = (tile * 256 + (f >> 3)) % nrows;
        p = big + (size_t)row * 1024 + ((s * 32) & 1023) + 4 * (f & 7);
      } else if (pattern == 1) { // 256 rows of the 1 MB matrix
        int row = f >> 3;
        p = small + (size_t)row * 1024 + ((s * 32) & 1023) + 4 * (f & 7);
      } else {                   // 64 k-rows x 128 floats
        int krow = ((tile * steps + s) * 64 + (f >> 5)) % nrows;
        p = big + (size_t)krow * 1024 + 4 * (f & 31);
      }
      r[i] = *reinterpret_cast<const float4*>(p);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(&lds[s & 1][(tid + i * 256) * 4]) = r[i];
    __syncthreads();
    acc += lds[s & 1][(tid * 7) & 8191];
  }
  out[blockIdx.x * 256 + tid] = acc;
}
int main() {
  const int nrows = 50000;
  float *big, *small, *out;
  hipMalloc(&big, (size_t)nrows * 1024 * 4); hipMalloc(&small, 1 << 20); hipMalloc(&out, 4096 * 256 * 4);
  hipMemset(big, 0, (size_t)nrows * 1024 * 4); hipMemset(small, 0, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pattern = 0; pattern < 3; ++pattern)
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
      int blocks = 256 * wg_per_cu, steps = 256;
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, big, small, out, pattern, steps, nrows);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      double bytes = (double)blocks * steps * 32768.0;
      printf("pattern %d  %d WG/CU: %.3f ms  %.2f TB/s chip  = %.1f B/clk/CU @2.4GHz\n", pattern, wg_per_cu, ms,
             bytes / ms / 1e9, bytes / ms / 1e3 / 256 / 2.4e3 / 1e0 * 1e-3 * 1e3 / 1e3);
    }
  return 0;
}
