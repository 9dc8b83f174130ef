// Calibration: fp32 MFMA (v_mfma_f32_32x32x2_f32) issue rate from registers, W waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
  int nb = 256 * blocks_per_cu;
  float* out; hipMalloc(&out, nb * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(nb), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)nb * 4 * iters * NACC * 4096.0;
    if (rep == 2) printf("acc=%d blocks/CU=%d iters=%d : %.3f ms  %.1f TFLOP/s  (%.2f GHz-equivalent at 64 cyc/MFMA/SIMD)\n",
                         NACC, blocks_per_cu, iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 2.4);
  }
  hipFree(out);
}
// The GEMM's shape: 8-wave workgroups (2 waves per SIMD), one per CU on `nb` CUs, NACC accumulators in rotation,
// `iters` rounds = the MFMA count of one wide tile (224 x 256 x 1024: 7 accumulators x 512 rounds).  Timed like
// bench.py times a kernel (events around ONE launch, after warm launches): launch + ramp-down included.
template <int NACC>
__global__ __launch_bounds__(512) void k8(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run_shape(int nb, int iters, int lds_bytes) {
  float* out; hipMalloc(&out, nb * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f, sum = 0.f;
  for (int rep = 0; rep < 24; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k8<NACC>, dim3(nb), dim3(512), lds_bytes, 0, out, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 4) { sum += ms; if (ms < best) best = ms; }
  }
  const double ideal_us = (double)iters * NACC * 2 * 64 / 2.4e9 * 1e6;   // 2 waves per SIMD, 64 cycles per MFMA, 2.4 GHz
  printf("8-wave WGs: blocks=%d acc=%d iters=%d lds=%d : avg %.1f us  best %.1f us  (ideal at 2.4 GHz %.1f us => %.1f %%)\n",
         nb, NACC, iters, lds_bytes, sum / 20 * 1e3, best * 1e3, ideal_us, ideal_us / (sum / 20 * 1e3) * 100);
  hipFree(out);
}
// The GEMM's MFMA order without anything else: 32 chunks x 4 fragment groups x (4 + 3 row blocks) x 4 k-pairs, distinct
// operand registers per (block, k-pair), optional sched_barrier between the steps and s_barrier per chunk.
template <bool SCHED, bool BAR>
__global__ __launch_bounds__(512) void k8g(float* out, int chunks, float a0, float b0) {
  f32x16 acc[7];
  for (int i = 0; i < 7; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a[7][4], b[4];
  for (int i = 0; i < 7; ++i) for (int j = 0; j < 4; ++j) a[i][j] = a0 + threadIdx.x * 1e-3f + i * 0.1f + j;
  for (int j = 0; j < 4; ++j) b[j] = b0 + threadIdx.x * 2e-3f + j;
  for (int c = 0; c < chunks; ++c) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        if (SCHED) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mb = (part ? 4 : 0); mb < (part ? 7 : 4); ++mb)
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][j], b[j], acc[mb], 0, 0, 0);
        if (SCHED) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 7; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <bool SCHED, bool BAR>
void run_gemm_order(int nb, int chunks) {
  float* out; hipMalloc(&out, nb * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float sum = 0.f;
  for (int rep = 0; rep < 24; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k8g<SCHED, BAR>), dim3(nb), dim3(512), 0, 0, out, chunks, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 4) sum += ms;
  }
  printf("GEMM MFMA order: blocks=%d chunks=%d sched_barrier=%d s_barrier=%d : avg %.1f us\n", nb, chunks, (int)SCHED, (int)BAR, sum / 20 * 1e3);
  hipFree(out);
}
int main() {
  run<4>(1, 20000); run<4>(2, 20000); run<1>(1, 40000); run<2>(2, 20000); run<4>(1, 200000);
  run_shape<7>(224, 512, 0); run_shape<7>(256, 512, 0); run_shape<7>(224, 512, 64 * 1024);
  run_shape<4>(224, 896, 0); run_shape<7>(224, 5120, 0);
  run_gemm_order<false, false>(224, 32); run_gemm_order<true, false>(224, 32);
  run_gemm_order<false, true>(224, 32); run_gemm_order<true, true>(224, 32);
  return 0;
}
