// bf16 NT GEMM main loop with LDS-DMA staging (buffer_load_dwordx4 ... lds) and three LDS stages, for the
// HBM-bound bf16 kernels (mmf_amil_bf16.hip).
//
// Why not the register-staged core (mmf_gemm_core.h) at bf16 rates -- measured on the first bf16 kernels
// (profiles/r01/README.md, "bf16 path"): a 64-k chunk is only 2048 MFMA cycles per CU, so (a) a prefetch distance
// of one chunk no longer covers the HBM latency (the mid-chunk `s_waitcnt vmcnt` before the LDS write stalled every
// wave), and (b) the ds_write_b128 pass (~13 cycles per KB-instruction, ~79 B/clk/CU) takes a large share of the LDS
// pipe that the fragment reads need.  LDS-DMA removes the staging registers, the ds_write pass and the mid-chunk
// wait; with three stages the loads of chunk t+2 are in flight while chunk t computes.
//
// LDS image of an operand chunk: [rows][128 bytes] (64 bf16 of one row), UNPADDED (a DMA instruction writes
// base + 16 * lane, so a wave-instruction covers 8 whole rows), 16-byte pieces XOR-swizzled within the row:
//     slot(row, piece) = piece ^ ((row >> 1) & 7)
// applied on the SOURCE side of the DMA (lane l of an instruction fetches piece (l & 7) ^ ((row >> 1) & 7) of row
// l >> 3) and on the fragment READ side (same involution).  A ds_read_b128 lane group holds 16 rows with 16
// distinct values of row mod 16 (MI355X_MICROARCH.md, LDS table), i.e. 16 distinct (row & 1, slot) pairs = all 64
// banks once: conflict-free.
//
// Ordering (cdna_hip_programming.md, "Pipelining across barriers"): a DMA is a pending LDS write on the VM counter;
// each wave waits `vmcnt(N)` for its own DMAs of the chunk that will be read NEXT, then `lgkmcnt(0)` (its fragment
// reads of the chunk just finished) and a raw s_barrier -- never __syncthreads(), whose fence would drain the
// prefetch.  A stage is re-filled one barrier after its last read.
#pragma once
#include "mmf_gemm_core.h"

namespace mmf {

template <int BM_, int BN_, int WM_, int WN_, int STAGES_ = 3>
struct TileS {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int NT = WM * WN * 64, NW = WM * WN;
  static constexpr int MB = BM / WM / 32, NB = BN / WN / 32;
  static constexpr int ROWB = 128;                                  // bytes per chunk row (64 bf16)
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int STAGES = STAGES_;                             // DMA prefetch distance = STAGES - 1 chunks
  static_assert(STAGES_ == 2 || STAGES_ == 3, "two or three LDS stages");
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be a multiple of 32x32");
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "every wave issues whole 8-row DMA instructions");
};

typedef __attribute__((address_space(3))) char lds_char;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// The DMA is issued from inline asm so that hipcc does not see an LDS write: with the builtin it put
// `s_waitcnt vmcnt(0)` in front of the next ds_read of ANY stage, draining the prefetch it had just issued.
// Hidden, the instruction is absent from the compiler's wait bookkeeping (cdna_hip_programming.md, "What hipcc does
// not do"): completion is counted by hand (wait_vmcnt<N>() + barrier before the stage is read).
struct DmaRsrc {           // raw buffer resource: base, num_records (bytes), flags -- all wave-uniform
  i32x4 w;
  __device__ inline void set(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    w.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    w.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xFFFFu));
    w.z = __builtin_amdgcn_readfirstlane((int)bytes);
    w.w = 0x00020000;
  }
};
__device__ inline unsigned lds_addr(const void* p) {       // LDS byte address of a pointer into shared memory
  return (unsigned)(unsigned long long)(lds_char*)(p);
}
__device__ inline void dma16(const DmaRsrc& rs, unsigned lds_dst, unsigned voff, unsigned soff) {
#ifdef MMF_DMA_M0NOP
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_nop 4"
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
#endif
               :: "s"(lds_dst), "v"(voff), "s"(rs.w), "s"(soff) : "memory");   // M0 is written and consumed inside this one statement
}

__device__ inline int swz_slot(int row, int piece) { return piece ^ ((row >> 1) & 7); }

// plain k-contiguous bf16 operand S[row][k] (leading dimension ld elements); rows >= nrows read as zero
template <int ROWS, int NT>
struct DmaK {
  static constexpr bool DMA = true;
  static constexpr int NW = NT / 64, NI = ROWS / 8 / NW;            // DMA instructions per wave and chunk
  DmaRsrc rs;
  int wave;
  unsigned voff[NI];
  __device__ inline void init(const void* p, int ld, int row0, int nrows) {
    const int lane = threadIdx.x & 63;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    rs.set(p, (unsigned)nrows * (unsigned)ld * 2u);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rl = (wave + i * NW) * 8 + (lane >> 3);
      const int rr = row0 + rl;
      voff[i] = rr < nrows ? (unsigned)rr * (unsigned)ld * 2u + 16u * (unsigned)swz_slot(rl, lane & 7) : OOB;
    }
  }
  // kill = 0x80000000 turns the request into an out-of-range one (zeros, no memory traffic): a branch-free loop's requests past
  // the last chunk would otherwise read the first bytes of the NEXT row (k offset 2 L lands there), i.e. real traffic
  __device__ inline void issue(int kt, char* stage, unsigned kill = 0u) const {
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(stage)) + (unsigned)wave * 1024u;
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)kt * 128u);
#pragma unroll
    for (int i = 0; i < NI; ++i) dma16(rs, base + (unsigned)(i * NW) * 1024u, voff[i] | kill, soff);
  }
};

template <int N>
__device__ inline void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MFMAs of one staged chunk.  hook(q), q = 0..3, runs in front of the MFMA block of k-step q (register-staged
// operands use it to spread their loads / LDS writes through the chunk).
template <class T, class Hook>
__device__ inline void compute_chunk_swz(const char* As, const char* Bs, f32x16 (&acc)[T::MB][T::NB], int wm, int wn, int lane,
                                         Hook&& hook) {
  const int r = lane & 31, hh = lane >> 5, sw = (r >> 1) & 7;
  const char* a0 = As + (wm * T::MB * 32 + r) * 128;
  const char* b0 = Bs + (wn * T::NB * 32 + r) * 128;
  float4 fa[2][T::MB], fb[2][T::NB];
  auto rd = [&](int q, int buf) {
    const int o = 16 * ((2 * q + hh) ^ sw);
#pragma unroll
    for (int mb = 0; mb < T::MB; ++mb) fa[buf][mb] = *reinterpret_cast<const float4*>(a0 + mb * 32 * 128 + o);
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb) fb[buf][nb] = *reinterpret_cast<const float4*>(b0 + nb * 32 * 128 + o);
  };
  rd(0, 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q + 1 < 4) rd(q + 1, (q + 1) & 1);
    hook(q);
    __builtin_amdgcn_sched_barrier(0);
#ifndef MMF_DIAG_NOMFMA
#pragma unroll
    for (int mb = 0; mb < T::MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < T::NB; ++nb) {
        const float av[4] = {fa[q & 1][mb].x, fa[q & 1][mb].y, fa[q & 1][mb].z, fa[q & 1][mb].w};
        const float bv[4] = {fb[q & 1][nb].x, fb[q & 1][nb].y, fb[q & 1][nb].z, fb[q & 1][nb].w};
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_bf16(av), frag_bf16(bv), acc[mb][nb], 0, 0, 0);
      }
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
}

// LA / LB: DMA loaders (`issue(kt, stage)`, `NI` instructions per wave and chunk, prefetch distance 2) or register
// loaders (`DMA == false`: `load(kt)` into registers during chunk kt-1, `store(stage)` half a chunk later; they write
// the swizzled image themselves).  Issue order inside an iteration: register loads first, then the DMAs, so that the
// in-order VM counter retires the register loads with `vmcnt(#DMA of this iteration)`.
template <class T, class LA, class LB>
__device__ inline void gemm_mainloop_dma(LA& la, LB& lb, int nk, char* lds, f32x16 (&acc)[T::MB][T::NB]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
  constexpr int NDMA = (LA::DMA ? LA::NI : 0) + (LB::DMA ? LB::NI : 0);
#pragma unroll
  for (int mb = 0; mb < T::MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  if (nk <= 0) return;
  auto stage = [&](int kt) { return lds + (kt % T::STAGES) * T::STAGE_BYTES; };
  constexpr int DIST = T::STAGES - 1;            // DMA prefetch distance in chunks
  // ---- prologue: chunks 0 .. DIST-1 on their way, chunk 0 landed ------------------------------------------
  if constexpr (!LA::DMA) { la.load(0); la.store(stage(0)); }
  if constexpr (!LB::DMA) { lb.load(0); lb.store(stage(0) + T::A_BYTES); }
  if constexpr (LA::DMA) la.issue(0, stage(0));
  if constexpr (LB::DMA) lb.issue(0, stage(0) + T::A_BYTES);
  if (DIST == 2 && nk > 1) {
    if constexpr (LA::DMA) la.issue(1, stage(1));
    if constexpr (LB::DMA) lb.issue(1, stage(1) + T::A_BYTES);
    wait_vmcnt<NDMA>();
  } else {
    wait_vmcnt<0>();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = stage(kt);
    const bool more = kt + 1 < nk, more_d = kt + DIST < nk;
#ifdef MMF_DIAG_NOLOAD
    const bool st1 = false, std_ = false;
#else
    const bool st1 = more, std_ = more_d;
#endif
    if (st1) {
      if constexpr (!LA::DMA) la.load(kt + 1);
      if constexpr (!LB::DMA) lb.load(kt + 1);
    }
    if (std_) {      // the stage being refilled was last read in iteration kt - 1 (DIST 2) / is chunk kt+1's own (DIST 1)
      if constexpr (LA::DMA) la.issue(kt + DIST, stage(kt + DIST));
      if constexpr (LB::DMA) lb.issue(kt + DIST, stage(kt + DIST) + T::A_BYTES);
    }
    compute_chunk_swz<T>(cur, cur + T::A_BYTES, acc, wm, wn, lane, [&](int q) {
      if (!st1) return;
      if (q == 2) { if constexpr (!LA::DMA) la.store(stage(kt + 1)); }
      if (q == 3) { if constexpr (!LB::DMA) lb.store(stage(kt + 1) + T::A_BYTES); }
    });
    // chunk kt+1 must have landed before the barrier that lets everyone read it; with three stages chunk kt+2
    // may stay in flight
    if (DIST == 2 && more_d) wait_vmcnt<NDMA>(); else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

}  // namespace mmf
