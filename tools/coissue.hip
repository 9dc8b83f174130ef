// Do a matrix-core wave and a vector-ALU wave on the SAME SIMD overlap?  One 512-thread workgroup per CU: waves 0-3 run
// MFMAs (one wave per SIMD), waves 4-7 run a VALU stream on the same SIMDs.  Each role is timed alone and together.
//   hipcc --offload-arch=gfx950 -O2 tools/coissue.hip -o multimodalfusion_amd/_diag/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ACC_AGPR>
__device__ inline void mfma_loop(int iters, float* out) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  f32x4 a = {1.f, 2.f, 3.f, 4.f}, b = {0.5f, 0.25f, 0.125f, 1.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ACC_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  if (s == 1.2345f) out[0] = s;
}

template <int KIND>   // 0: v_fma_f32, 1: v_exp_f32, 2: v_mul_lo_u32, 3: v_cvt_pk_bf16_f32-like mix
__device__ inline void valu_loop(int iters, float* out) {
  float v[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { v[i] = 1.0f + i + threadIdx.x * 1e-3f; u[i] = threadIdx.x * 2654435761u + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(0.999f));
      if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      if (KIND == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(0x9E3779B1u));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i];
  if (s == 1.2345f) out[1] = s;
}

template <int KIND, int ACC_AGPR>
__global__ __launch_bounds__(512) void k(int mfma_iters, int valu_iters, float* out, unsigned long long* cyc, int prio) {
  const int wave = threadIdx.x >> 6;
  if (prio == 1 && wave >= 4) __builtin_amdgcn_s_setprio(3);
  if (prio == 2 && wave < 4) __builtin_amdgcn_s_setprio(3);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) { if (mfma_iters) mfma_loop<ACC_AGPR>(mfma_iters, out); }
  else { if (valu_iters) valu_loop<KIND>(valu_iters, out); }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int KIND, int ACC_AGPR>
void run(const char* name, int mi, int vi, float* out, unsigned long long* cyc, int prio = 0) {
  unsigned long long h[8];
  int cfg[3][2] = {{mi, 0}, {0, vi}, {mi, vi}};
  printf("%-28s", name);
  for (auto& c : cfg) {
    hipLaunchKernelGGL((k<KIND, ACC_AGPR>), dim3(256), dim3(512), 0, 0, c[0], c[1], out, cyc, prio);
    hipLaunchKernelGGL((k<KIND, ACC_AGPR>), dim3(256), dim3(512), 0, 0, c[0], c[1], out, cyc, prio);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("  [mfma %d valu %d] mfma-wave %7llu valu-wave %7llu", c[0], c[1], h[0], h[4]);
  }
  printf("\n");
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64); hipMalloc(&cyc, 64);
  const int MI = 2000;                 // 8000 MFMAs = 256k cycles
  run<0, 0>("fma, acc in VGPR", MI, 8000, out, cyc);   // 64000 v_fma
  run<0, 1>("fma, acc in AGPR", MI, 8000, out, cyc);
  run<1, 0>("exp, acc in VGPR", MI, 4000, out, cyc);
  run<1, 1>("exp, acc in AGPR", MI, 4000, out, cyc);
  run<2, 0>("mul_lo_u32, acc in VGPR", MI, 2000, out, cyc);
  run<2, 1>("mul_lo_u32, acc in AGPR", MI, 2000, out, cyc);
  run<0, 0>("fma, VALU wave prio 3", MI, 8000, out, cyc, 1);
  run<1, 0>("exp, VALU wave prio 3", MI, 4000, out, cyc, 1);
  run<2, 0>("mul_lo, VALU wave prio 3", MI, 2000, out, cyc, 1);
  run<0, 0>("fma, MFMA wave prio 3", MI, 8000, out, cyc, 2);
  return 0;
}
