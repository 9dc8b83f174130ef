// fp32 GEMM on the bf16 matrix cores: 3-way operand split, 6 products, fp32 accumulation (gfx950).
//
// The exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 157 TFLOP/s; v_mfma_f32_32x32x16_bf16 at 2.5 PFLOP/s, 16 x as
// fast.  An fp32 value is the exact sum of three bf16 values,
//     a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)        (8 + 8 + 8 significant bits)
// and a product a.b = sum_{i,j} a_i b_j.  Every a_i b_j is exact in fp32 (8 x 8 bits); the three terms with i + j >= 3
// are below 2^-24 |a b| and are dropped, the other six are accumulated in fp32 by the matrix core:
//     a.b ~= a0 b2 + a2 b0 + a1 b1 + a0 b1 + a1 b0 + a0 b0        (smallest first)
// Six bf16 MFMAs replace one fp32 MFMA's worth of k: 2.67 x the exact instruction's rate.  Error against an fp64
// product: the same as the exact-fp32 instruction's (measured on the path's shapes: tests/test_gpu_split.py,
// DESIGN.md 7d) -- the accumulation is fp32 either way and dominates.
// Not representable: |a| within one bf16 ulp of FLT_MAX (bf16(a) rounds to inf); inf inputs give NaN instead of inf;
// below 2^-110 the low planes underflow (absolute error < 2^-126).  tests/test_split_math_cpu.py restates the arithmetic.
//
// Layout: operands are staged through LDS in chunks of SKC = 16 k-values.  An LDS row (one tile row / column) holds
// the chunk's three planes back to back, [a0: 16 bf16 | a1 | a2] = 96 bytes, padded to 112 (7 16-byte slots, odd, so
// the 16-lane groups of a ds_read_b128 hit distinct slots).  One ds_read_b128 at +32 p + 16 hh is lane (r, hh)'s
// operand of plane p for v_mfma_f32_32x32x16_bf16 (k = 8 hh .. 8 hh + 7).  The split happens in the staging path, on
// the way from the loaders' registers into LDS, so HBM still holds (and the kernels still read) plain fp32.
#pragma once
#include <utility>

#include "mmf_gemm_core.h"

namespace mmf {

constexpr int SKC = 16;           // fp32 k-values per staged chunk
constexpr int SROW = 112;         // bytes per LDS row
constexpr int SROW_F = SROW / 4;

template <int BM_, int BN_, int WM_, int WN_, int GM_ = 1>
struct TileSp {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, GM = GM_;
  static constexpr int NT = WM * WN * 64;
  static constexpr int MB = BM / WM / 32, NB = BN / WN / 32;
  static constexpr bool HALF = false, SPLIT = true, PERM = false;
  static constexpr int A_FLOATS = BM * SROW_F, B_FLOATS = BN * SROW_F;
  static constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
  static constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be a multiple of 32x32");
};

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ inline uint32_t cvt_pk_bf16(float lo, float hi) {       // v_cvt_pk_bf16_f32, round to nearest even
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// two fp32 values -> three packed bf16 pairs.  The first plane is taken from the value CLAMPED to the largest finite bf16
// (3.3895e38): rounding to nearest turns a finite fp32 value beyond it (up to FLT_MAX) into an infinite plane, and
// inf * 0 = NaN where the exact-fp32 path stays finite.  The remainder is taken from the unclamped value, so the three
// planes still sum to it exactly (the clamp moves the first plane by less than one of its ulps).  One v_med3_f32 per value.
// (A truncated first plane would do it for free but biases the dropped cross terms: 3x the error, tests/test_split_math_cpu.py.)
__device__ inline void split_pair(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  constexpr float BF16_MAX = 3.3895313892515355e38f;      // 0x7F7F0000
  p0 = cvt_pk_bf16(__builtin_amdgcn_fmed3f(x0, -BF16_MAX, BF16_MAX), __builtin_amdgcn_fmed3f(x1, -BF16_MAX, BF16_MAX));
  float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xFFFF0000u);
  p1 = cvt_pk_bf16(r0, r1);
  r0 -= __uint_as_float(p1 << 16); r1 -= __uint_as_float(p1 & 0xFFFF0000u);
  p2 = cvt_pk_bf16(r0, r1);
}
#ifdef MMF_SDIAG_NOLDSW        /* diagnostic build: the split is computed, nothing is written to LDS (results are wrong) */
__device__ inline void st_u2(float* p, uint32_t a, uint32_t b) { asm volatile("" :: "v"(a), "v"(b), "v"(p)); }
__device__ inline void st_u4(float* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(p)); }
#else
__device__ inline void st_u2(float* p, uint32_t a, uint32_t b) { *reinterpret_cast<uint2*>(p) = make_uint2(a, b); }
__device__ inline void st_u4(float* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
}
#endif
// Vector slot idx of a k-contiguous chunk image (4 float4 per row) -> tile row.  Not idx / 4: the 16 lanes that share
// a ds_write_b64 pass must land on 32 distinct banks, and 4 CONSECUTIVE rows do not (row stride 112 bytes = 28 banks:
// rows r and r + 1 overlap in 4 banks, a 2-way conflict on half of every pass).  Rows r, r + 2, r + 4, r + 6 do
// (0, 24, 16, 8 mod 32), so lane group g of a wave takes the even or odd rows of one half of the wave's 16 rows.
// The column piece is idx & 3 either way, so global loads stay 64 contiguous bytes per 4 lanes.
__device__ inline int krow(int idx) {
  const int l = idx & 63, g = l >> 4, rsel = (l & 15) >> 2;
  return (idx >> 6) * 16 + 2 * rsel + (g & 1) + 8 * (g >> 1);
}
// 4 consecutive k of one row -> the row's three planes (8 bytes each) at k offset 4 c4
__device__ inline void split_store4(float* row, int c4, const float4& v) {
  uint32_t a0, a1, a2, b0, b1, b2;
  split_pair(v.x, v.y, a0, a1, a2);
  split_pair(v.z, v.w, b0, b1, b2);
  st_u2(row + 2 * c4, a0, b0);
  st_u2(row + 8 + 2 * c4, a1, b1);
  st_u2(row + 16 + 2 * c4, a2, b2);
}
// 8 consecutive k of one row -> three 16-byte plane halves at k offset 8 kh
__device__ inline void split_store8(float* row, int kh, const float (&v)[8]) {
  uint32_t p[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], p[0][j], p[1][j], p[2][j]);
#pragma unroll
  for (int q = 0; q < 3; ++q) st_u4(row + 8 * q + 4 * kh, p[q][0], p[q][1], p[q][2], p[q][3]);
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int N, class F, int I = 0>
__device__ inline void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

// ---- loaders ------------------------------------------------------------------------------------------------
// k-contiguous fp32 source S[row][k] (leading dimension ld): a chunk row is 64 bytes = 4 float4.  A vector slot
// beyond the tile (ROWS * 4 not a multiple of NT) repeats the thread's previous slot -- same address, same data, same
// LDS destination -- so that load() and store() stay branch-free (see split_mainloop).
template <int ROWS, int NT>
struct SplitK {
  static constexpr int TOTAL = ROWS * 4, NV = (TOTAL + NT - 1) / NT;
  static_assert(TOTAL >= NT, "tile too small for this thread count");
  rsrc_t rs;
  int tid;
  unsigned voff[NV];
  float4 r[NV];
  __device__ static inline int slot(int tid, int i) { const int idx = tid + i * NT; return idx < TOTAL ? idx : idx - NT; }
  __device__ inline void init(const float* p0, int ld, int row0, int nrows) {
    rs = make_rsrc(p0, (unsigned)nrows * (unsigned)ld * 4u);
    tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = slot(tid, i), rr = row0 + krow(idx);
      voff[i] = rr < nrows ? ((unsigned)rr * (unsigned)ld + 4u * (idx & 3)) * 4u : OOB;
    }
  }
  __device__ inline void load(int kt) {
#ifdef MMF_SDIAG_NOGLOAD
    if (kt >= 4) return;
#endif
    const unsigned soff = (unsigned)(kt * SKC) * 4u;
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = bld4(rs, voff[i], soff);
  }
  static constexpr int PIECES = NV;
  __device__ inline void store_piece(float* lds, int i) const {
    const int idx = slot(tid, i);
    split_store4(lds + krow(idx) * SROW_F, idx & 3, r[i]);
  }
  __device__ inline void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = slot(tid, i);
#ifdef MMF_SDIAG_NOSPLIT       /* LDS writes of unsplit data: the staging path without its VALU work */
      st_u2(lds + krow(idx) * SROW_F + 2 * (idx & 3), __float_as_uint(r[i].x), __float_as_uint(r[i].y));
      st_u2(lds + krow(idx) * SROW_F + 8 + 2 * (idx & 3), __float_as_uint(r[i].z), __float_as_uint(r[i].w));
      st_u2(lds + krow(idx) * SROW_F + 16 + 2 * (idx & 3), __float_as_uint(r[i].x), __float_as_uint(r[i].w));
      continue;
#endif
      split_store4(lds + krow(idx) * SROW_F, idx & 3, r[i]);
    }
  }
};

// K-gate's B operand (see LoadGateW): tile row j -> row of Wa or Wb.  !HALVES: 32-row blocks alternate (a, b) for the
// same 32 attention dims.  A wave-instruction covers 16 consecutive tile rows, so the source is wave-uniform.
template <int ROWS, int NT, bool GATED>
struct SplitGateW {
  static constexpr int TOTAL = ROWS * 4, NV = TOTAL / NT;
  static_assert(TOTAL % NT == 0, "whole vector slots only");
  rsrc_t ra, rb;
  int tid;
  int which[NV];
  unsigned voff[NV];
  float4 r[NV];
  __device__ inline void init(const float* wa, const float* wb, int H, int D, int d0) {
    tid = threadIdx.x;
    ra = make_rsrc(wa, (unsigned)D * (unsigned)H * 4u);
    rb = make_rsrc(GATED ? wb : wa, (unsigned)D * (unsigned)H * 4u);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT, j = krow(idx);
      int w, d;
      if (!GATED) { w = 0; d = d0 + j; }
      else { w = (j >> 5) & 1; d = d0 + (j >> 6) * 32 + (j & 31); }
      which[i] = __builtin_amdgcn_readfirstlane(w);
      voff[i] = d < D ? ((unsigned)d * (unsigned)H + 4u * (idx & 3)) * 4u : OOB;
    }
  }
  __device__ inline void load(int kt) {
    const unsigned soff = (unsigned)(kt * SKC) * 4u;
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = bld4(which[i] ? rb : ra, voff[i], soff);
  }
  static constexpr int PIECES = NV;
  __device__ inline void store_piece(float* lds, int i) const {
    const int idx = tid + i * NT;
    split_store4(lds + krow(idx) * SROW_F, idx & 3, r[i]);
  }
  __device__ inline void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) store_piece(lds, i);
  }
};

// k-major fp32 source S[k][col] (leading dimension ld): tile row = source column.  A thread takes 8 consecutive k of
// ONE column (8 dword loads; a wave-instruction covers 64 consecutive columns of a k row, 256 contiguous bytes) and so
// holds a whole 16-byte plane half per plane: the transpose costs nothing beyond the split.  k rows outside
// [kbase, kmax) and columns >= ncols read as zero.
template <int ROWS, int NT>
struct SplitM {
  static constexpr int TOTAL = ROWS * 2, NV = (TOTAL + NT - 1) / NT;
  static_assert(TOTAL >= NT, "tile too small for this thread count");
  rsrc_t rs;
  unsigned ldb, kbase_b;
  int tid;
  unsigned voff[NV];
  float r[NV][8];
  __device__ static inline int slot(int tid, int i) { const int idx = tid + i * NT; return idx < TOTAL ? idx : idx - NT; }
  __device__ inline void init(const float* s, int ld, int col0, int ncols, int kbase, int kmax) {
    tid = threadIdx.x;
    rs = make_rsrc(s, (unsigned)(kmax > 0 ? kmax : 0) * (unsigned)ld * 4u);
    ldb = (unsigned)ld * 4u;
    kbase_b = (unsigned)kbase * ldb;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = slot(tid, i), c = col0 + idx % ROWS, kh = idx / ROWS;
      voff[i] = c < ncols ? (unsigned)(8 * kh) * ldb + (unsigned)c * 4u : OOB;
    }
  }
  __device__ inline void load(int kt) {
#ifdef MMF_SDIAG_NOGLOAD
    if (kt >= 4) return;
#endif
    const unsigned soff = kbase_b + (unsigned)(kt * SKC) * ldb;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) r[i][j] = bld1(rs, voff[i], soff + (unsigned)j * ldb);
  }
  static constexpr int PIECES = NV;
  __device__ inline void store_piece(float* lds, int i) const {
    const int idx = slot(tid, i);
    split_store8(lds + (idx % ROWS) * SROW_F, idx / ROWS, r[i]);
  }
  __device__ inline void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) store_piece(lds, i);
  }
};

// ---- MFMA over one staged chunk -------------------------------------------------------------------------------
struct FragS { f32x4 p[3]; };        // a row's three plane halves (8 bf16 each) for one lane

__device__ inline void read_frag(const float* row, FragS& f) {
#ifdef MMF_SDIAG_NOFRAG        /* diagnostic build: fragments come from registers, not LDS (results are wrong) */
#pragma unroll
  for (int q = 0; q < 3; ++q) f.p[q] = f32x4{(float)(size_t)row, 1.f, 2.f, (float)q};
  return;
#endif
#pragma unroll
  for (int q = 0; q < 3; ++q) f.p[q] = *reinterpret_cast<const f32x4*>(row + 8 * q);
}
__device__ inline f32x16 mfma_bf(const f32x4& a, const f32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Row blocks are taken one at a time (GM = 1: the 6 MFMAs of a block chain on one accumulator, which the matrix core
// forwards at its issue rate; two blocks at a time cost 24 more fragment registers and spilled in K-dh); the fragments
// of group s+1 are read while the 6 GM NB MFMAs of group s issue.  hook(s) runs in
// front of group s's MFMAs: the main loop spreads the staging of later chunks over it.
template <class T>
constexpr int split_gm() { return T::GM; }
template <class T>
constexpr int split_steps() { return (T::MB + split_gm<T>() - 1) / split_gm<T>(); }

template <class T, class Hook>
__device__ inline void compute_chunk_split(const float* __restrict__ As, const float* __restrict__ Bs,
                                           f32x16 (&acc)[T::MB][T::NB], int wm, int wn, int lane, Hook&& hook) {
  constexpr int GM = split_gm<T>(), NS = split_steps<T>();
  const int r = lane & 31, hh = lane >> 5;
  const float* a0 = As + (wm * T::MB * 32 + r) * SROW_F + 4 * hh;
  const float* b0 = Bs + (wn * T::NB * 32 + r) * SROW_F + 4 * hh;
  FragS fb[T::NB], fa[2][GM];
#pragma unroll
  for (int nb = 0; nb < T::NB; ++nb) read_frag(b0 + nb * 32 * SROW_F, fb[nb]);
#pragma unroll
  for (int m = 0; m < GM; ++m)
    if (m < T::MB) read_frag(a0 + m * 32 * SROW_F, fa[0][m]);
  constexpr int TI[6] = {0, 2, 1, 0, 1, 0}, TJ[6] = {2, 0, 1, 1, 0, 0};   // smallest products first
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (s + 1 < NS) {
#pragma unroll
      for (int m = 0; m < GM; ++m)
        if ((s + 1) * GM + m < T::MB) read_frag(a0 + ((s + 1) * GM + m) * 32 * SROW_F, fa[(s + 1) & 1][m]);
    }
    __builtin_amdgcn_sched_barrier(0);       // the reads stay up here (left free, hipcc sank them to the end of the step)
    // The step's staging work (hook) and its MFMAs are ONE scheduling region: the split's VALU work has no
    // dependence on the MFMAs and hipcc interleaves the two streams (a few VALU instructions behind every MFMA).
    // Fenced off from each other, both waves of a SIMD did their MFMAs and then their staging, in phase, and the
    // two streams added up (phase split, 50k projection: 92 us of MFMA + 64 us of staging = 137 measured).
    hook(s);
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int m = 0; m < GM; ++m)
#pragma unroll
        for (int nb = 0; nb < T::NB; ++nb) {
          const int mb = s * GM + m;
#ifndef MMF_SDIAG_NOMFMA      /* diagnostic builds (tools/diag_build.py): timing only, results are wrong */
          if (mb < T::MB) acc[mb][nb] = mfma_bf(fa[s & 1][m].p[TI[t]], fb[nb].p[TJ[t]], acc[mb][nb]);
#endif
        }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Main loop: D loader copies keep D chunks in flight (a chunk computes in ~1.3-1.5 thousand cycles per wave, a load
// takes 2-4 thousand); chunk kt+D is requested while chunk kt computes, and split + written to LDS D-1 chunks later.
// LDS is double-buffered, one barrier per chunk.  Branch-free on purpose, and nk % D == 0: see gemm_mainloop_deep.
// Loaders that accumulate something while staging (the TN kernel's column sums) define absorb(copy): the main loop
// works on D copies of the caller's loader and hands their state back at the end.
template <class L, class = void>
struct has_absorb : std::false_type {};
template <class L>
struct has_absorb<L, std::void_t<decltype(std::declval<L&>().absorb(std::declval<const L&>()))>> : std::true_type {};

// STAMP (diagnostic -DMMF_STAMPS builds only): per wave, s_memtime ticks inside the chunks' work and at their barriers,
// summed into g_stamps[0 / 1] (waves 0-3) and [2 / 3] (waves 4-7), chunk count in [4]
template <class T, int D, class LA, class LB, bool STAMP = false>
__device__ inline void split_mainloop(LA& la0, const LB& lb0, int nk, float* lds, f32x16 (&acc)[T::MB][T::NB]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
#pragma unroll
  for (int mb = 0; mb < T::MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  if (nk <= 0) return;
  LA la[D];
  LB lb[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    la[j] = la0; lb[j] = lb0;
    la[j].load(j); lb[j].load(j);
  }
  la[0].store(lds);
  lb[0].store(lds + T::A_FLOATS);
  __syncthreads();
  constexpr int NS = split_steps<T>();
  static_assert((D & 1) == 0, "an even D keeps the LDS stage of copy j fixed");
#ifdef MMF_STAMPS
  unsigned long long st_work = 0, st_bar = 0;
#endif
  for (int kt0 = 0; kt0 < nk; kt0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int kt = kt0 + j;
      float* cur = lds + (j & 1) * T::STAGE_FLOATS;
      float* nxt = lds + ((j + 1) & 1) * T::STAGE_FLOATS;
#ifdef MMF_STAMPS
      unsigned long long st0 = 0;
      if constexpr (STAMP) st0 = stamp_now();
#endif
      compute_chunk_split<T>(cur, cur + T::A_FLOATS, acc, wm, wn, lane, [&](int s) {
#ifdef MMF_SDIAG_NOSTAGE
        return;
#endif
        // requests in the first steps (copy j has been written out: reuse it); the split + LDS writes of chunk kt+1 in
        // PIECES spread evenly over the chunk's steps
        if (s == 0) la[j].load(kt + D);
        if (s == (NS >= 2 ? 1 : 0)) lb[j].load(kt + D);
        constexpr int PA = LA::PIECES, PT = LA::PIECES + LB::PIECES;
        // D = 2 (the TN kernel: no registers for a deeper ring): the pieces go to the LAST steps, so that a request has
        // 1.5+ chunks to land instead of 1 -- with the pieces up front the whole workgroup waited at the barrier for
        // its slowest load (no-barrier diagnostic build: 207 -> 180 us)
        constexpr bool LATE = D == 2 && PT <= NS;
#pragma unroll
        for (int q = 0; q < PT; ++q)
          if (s == (LATE ? NS - PT + q : q * NS / PT)) {
            if (q < PA) la[(j + 1) % D].store_piece(nxt, q);
            else lb[(j + 1) % D].store_piece(nxt + T::A_FLOATS, q - PA);
          }
      });
#ifdef MMF_STAMPS
      unsigned long long st1 = 0;
      if constexpr (STAMP) st1 = stamp_now();
#endif
#ifndef MMF_SDIAG_NOBAR
      __syncthreads();
#endif
#ifdef MMF_STAMPS
      if constexpr (STAMP) { st_work += st1 - st0; st_bar += stamp_now() - st1; }
#endif
    }
  }
#ifdef MMF_STAMPS
  if constexpr (STAMP) {
    if (lane == 0) {
      const int grp = wave >= 4 ? 2 : 0;
      atomicAdd(&g_stamps[grp], st_work); atomicAdd(&g_stamps[grp + 1], st_bar);
      if (wave == 0) atomicAdd(&g_stamps[4], (unsigned long long)nk);
    }
  }
#endif
  if constexpr (has_absorb<LA>::value) {
#pragma unroll
    for (int j = 0; j < D; ++j) la0.absorb(la[j]);
  }
}

}  // namespace mmf
