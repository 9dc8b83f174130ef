// Do LDS-DMA writes (buffer_load ... lds, M0-addressed) of two workgroups that share a CU stay inside their own LDS allocation?
// Each workgroup (256 threads, LDS bytes given on the command line) fills its LDS with a canary, DMAs its own pattern into
// [0, 48 KB), waits, and checks every dword of its allocation, many rounds.   usage: dma_two_wg <lds_bytes> <rounds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ inline void dma16(i32x4 rs, unsigned lds_dst, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__global__ __launch_bounds__(256, 2) void k(const unsigned* src, unsigned* errs, int lds_bytes, int rounds) {
  extern __shared__ __align__(16) unsigned lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long a = (unsigned long long)(src + (size_t)blockIdx.x * 12288);
  i32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xFFFF));
  rs.z = 49152; rs.w = 0x00020000;
  const int nd = lds_bytes / 4;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int i = tid; i < nd; i += 256) lds[i] = 0xC0000000u | (unsigned)i;
    __syncthreads();
    for (int i = 0; i < 12; ++i) {                      // 48 instructions of 1 KB: wave w issues pieces w + 4 i
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)((wave + 4 * i) * 1024));
      dma16(rs, dst, (unsigned)lane * 16u, dst);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < nd; i += 256) {
      const unsigned v = lds[i];
      const unsigned want = i < 12288 ? (blockIdx.x << 16 | (unsigned)i) : (0xC0000000u | (unsigned)i);
      if (v != want) { ++bad; if (bad == 1) { errs[blockIdx.x * 4 + 1] = (unsigned)i; errs[blockIdx.x * 4 + 2] = v; errs[blockIdx.x * 4 + 3] = r; } }
    }
    __syncthreads();
  }
  if (bad) atomicAdd(&errs[blockIdx.x * 4], bad);
}
int main(int argc, char** argv) {
  const int lds_bytes = argc > 1 ? atoi(argv[1]) : 79424, rounds = argc > 2 ? atoi(argv[2]) : 50, G = 512;
  unsigned *src, *errs;
  hipMalloc(&src, (size_t)G * 49152); hipMalloc(&errs, G * 16);
  unsigned* h = (unsigned*)malloc((size_t)G * 49152);
  for (int b = 0; b < G; ++b) for (int i = 0; i < 12288; ++i) h[(size_t)b * 12288 + i] = (unsigned)b << 16 | (unsigned)i;
  hipMemcpy(src, h, (size_t)G * 49152, hipMemcpyHostToDevice);
  hipMemset(errs, 0, G * 16);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(k, dim3(G), dim3(256), lds_bytes, 0, src, errs, lds_bytes, rounds);
  hipError_t e = hipDeviceSynchronize();
  unsigned he[G * 4];
  hipMemcpy(he, errs, sizeof he, hipMemcpyDeviceToHost);
  int nb = 0;
  for (int b = 0; b < G; ++b) if (he[b * 4]) { if (nb++ < 8) printf("block %d: %u bad dwords, first at dword %u (byte 0x%x) value 0x%08x round %u\n", b, he[b * 4], he[b * 4 + 1], he[b * 4 + 1] * 4, he[b * 4 + 2], he[b * 4 + 3]); }
  printf("lds %d bytes, %d rounds: %d of %d workgroups saw corruption (%s)\n", lds_bytes, rounds, nb, G, hipGetErrorString(e));
  return 0;
}
