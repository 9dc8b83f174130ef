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
int main() {
  run<4>(1, 20000); run<4>(2, 20000); run<1>(1, 40000); run<2>(2, 20000); run<4>(1, 200000);
  return 0;
}
