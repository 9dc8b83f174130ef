// Which workgroups share a CU?  512 workgroups of 256 threads with 76 KB of LDS each (two fit per CU): every workgroup records
// HW_ID / XCC_ID and its start time.   hipcc --offload-arch=gfx950 -O2 tools/cu_census.hip -o gpurun_out/cu_census && gpurun_out/cu_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void census(unsigned* out, int spin) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    out[blockIdx.x * 4 + 0] = hw;
    out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = (unsigned)__builtin_amdgcn_s_memrealtime();
  }
  lds[threadIdx.x] = 1;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
  if (threadIdx.x == 0) out[blockIdx.x * 4 + 3] = (unsigned)__builtin_amdgcn_s_memrealtime();
}
int main() {
  const int G = 782;
  unsigned* d;
  hipMalloc(&d, G * 16);
  hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 77000);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(census, dim3(G), dim3(256), 77000, 0, d, 300);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(G * 4);
  hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  unsigned t0 = ~0u;
  for (int b = 0; b < G; ++b) t0 = h[b * 4 + 2] < t0 ? h[b * 4 + 2] : t0;
  for (int b = 0; b < G; ++b) cu[(h[b * 4 + 1] << 16) | ((h[b * 4] >> 8) & 0xFF)].push_back(b);
  printf("distinct (xcc, se/sh/cu) keys: %zu\n", cu.size());
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ < 24) {
      printf("xcc %u cu-key 0x%02x:", kv.first >> 16, kv.first & 0xFF);
      for (int b : kv.second) printf(" %d(t=%u..%u)", b, (h[b * 4 + 2] - t0), (h[b * 4 + 3] - t0));
      printf("\n");
    }
  }
  std::map<size_t, int> hist;
  for (auto& kv : cu) hist[kv.second.size()]++;
  for (auto& kv : hist) printf("CUs with %zu workgroups: %d\n", kv.first, kv.second);
  printf("hw_id samples: ");
  for (int b = 0; b < 8; ++b) printf("%08x ", h[b * 4]);
  printf("\n");
  return 0;
}
