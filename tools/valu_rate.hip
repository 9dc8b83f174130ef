// Calibration: issue rate of the VALU instructions the operand split uses, per SIMD, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o multimodalfusion_amd/_diag/valu_rate tools/valu_rate.hip && multimodalfusion_amd/_diag/valu_rate
// One wave per SIMD (256 threads, one workgroup per CU), ITER x 32 independent instructions of one kind per wave;
// cycles per instruction = s_memtime ticks (100 MHz) x clock ratio, reported as ns per instruction and per-SIMD rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
  float a[16], b[16];
  uint32_t u[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.0f + 0.0001f * i; u[i] = threadIdx.x + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      if (KIND == 0) { asm volatile("v_add_f32 %0, %1, %0\n\tv_add_f32 %2, %3, %2" : "+v"(a[i]), "+v"(b[i]), "+v"(a[i + 1]), "+v"(b[i + 1])); }
      if (KIND == 1) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_cvt_pk_bf16_f32 %3, %2, %1" : "=v"(u[i]), "+v"(a[i]), "+v"(b[i]), "=v"(u[i + 1])); }
      if (KIND == 2) { asm volatile("v_and_b32 %0, %1, %0\n\tv_lshlrev_b32 %2, 16, %2" : "+v"(u[i]), "+v"(u[i + 1]), "+v"(u[i + 1])); }
      if (KIND == 3) { asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(*(f32x2*)&a[i]), "+v"(*(f32x2*)&b[i])); }
      if (KIND == 4) { asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]), "v"(u[i + 1])); asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i + 1]) : "v"(b[i]), "v"(a[i]), "v"(u[i])); }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a[i] + b[i] + (float)u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
double run(const char* name, int per_iter) {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<KIND><<<256, 256>>>(out, 100);
  hipEventRecord(e0);
  rate_kernel<KIND><<<256, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)iters * per_iter);
  printf("%-22s %.3f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, ns, ns * 2.4);
  hipFree(out);
  return ns;
}
int main() {
  run<0>("v_add_f32", 16);
  run<1>("v_cvt_pk_bf16_f32", 16);
  run<2>("v_and/v_lshlrev", 16);
  run<3>("v_pk_add_f32", 8);
  run<4>("v_perm_b32", 16);
  return 0;
}
