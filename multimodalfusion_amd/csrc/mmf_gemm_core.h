// LDS-tiled exact-fp32 MFMA GEMM core for gfx950 (v_mfma_f32_32x32x2_f32), shared by every
// contraction on the attention-MIL path:
//
//   NT  C[m][n] = sum_k A[m][k] * B[n][k]     both operands k-contiguous   (x.W1^T, h.Wab^T)
//   NN  C[m][n] = sum_k A[m][k] * B[k][n]     A k-contiguous, B n-contiguous (dP.Wab)
//   TN  C[m][n] = sum_k A[k][m] * B[k][n]     both operands k-major          (du^T.x split-K)
//
// Design (wave64, one workgroup = WM x WN waves, each wave owns MB x NB accumulator blocks of
// 32x32):  operands are streamed through LDS in k-chunks of KC = 32, double-buffered, one
// barrier per chunk; the global loads of chunk t+1 are issued before the MFMAs of chunk t and
// written to LDS after them (register staging, issue-early / write-late).
//
// An operand's LDS image keeps the layout it has in HBM, so staging is a straight 16-byte
// copy (coalesced global_load_dwordx4 -> ds_write_b128):
//   k-contiguous  : [rows][KC+4]  fragment = ONE ds_read_b128 per 4 MFMAs; the +4 pad makes
//                   the 16-lane read groups hit 16 distinct 16-B slots (stride 9 slots, odd).
//   m-contiguous  : [KC][rows]    fragment = 4 ds_read_b32, lanes 0..31 on consecutive banks.
// The 32x32x2 MFMA takes A[i = lane&31][k = lane>>5] / B[k = lane>>5][j = lane&31].  Inside a
// group of 8 k-values lane-half hh holds k = 8q + 4hh + j (j = 0..3), i.e. MFMA j contracts
// k in {8q+j, 8q+4+j}; A and B use the same mapping, and a sum does not care about k order.
//
// C/D layout (guide: cdna_hip_programming.md section 3): col = lane & 31,
// row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
#pragma once
#include <type_traits>

#include "mmf_common.h"

namespace mmf {

constexpr int KC = 32;         // k-chunk staged per pipeline step
constexpr int KSTR = KC + 4;   // padded LDS row stride (floats) of a k-contiguous image

// G = k-pairs per fragment group (one MFMA consumes one k-pair).  4 matches a ds_read_b128 of a k-contiguous
// image; all-m-contiguous tiles may use 2, which halves the (double-buffered) fragment registers.
//
// BF16 = true: the operands are bf16 and both k-contiguous.  The byte geometry does not change -- a chunk row is
// still 128 bytes (64 bf16 instead of 32 floats), staged by the same 16-byte copies into the same padded image --
// so loaders address a bf16 matrix as a float matrix of half the width, and only the MFMA differs: one
// ds_read_b128 fragment (8 bf16 of one row) feeds ONE v_mfma_f32_32x32x16_bf16 instead of four 32x32x2 fp32 MFMAs.
template <int BM_, int BN_, int WM_, int WN_, bool A_KCONTIG_, bool B_KCONTIG_, int G_ = 4, bool BF16_ = false>
struct Tile {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, G = G_;
  static constexpr bool BF16 = BF16_;
  static constexpr bool SPLIT = false;     // true: TileSp (mmf_gemm_split.h)
  static_assert(G_ == 4 || (G_ == 2 && !A_KCONTIG_ && !B_KCONTIG_), "G = 2 only for all-m-contiguous tiles");
  static_assert(!BF16_ || (A_KCONTIG_ && B_KCONTIG_ && G_ == 4), "bf16 tiles are k-contiguous on both sides");
  static constexpr bool A_KCONTIG = A_KCONTIG_, B_KCONTIG = B_KCONTIG_;
  static constexpr int NT = WM * WN * 64;
  static constexpr int MB = BM / WM / 32, NB = BN / WN / 32;
  // HALF: the wave's rows end with a 16-row HALF block (BM / WM = 32 MB + 16), multiplied on v_mfma_f32_16x16x4_f32
  // (same FLOP per cycle as the 32x32x2 instruction).  208-row tiles = 6.5 blocks put a 50k bag on 241 of 256 CUs
  // instead of 224 (195 rows per CU would be ideal; 192-row tiles need 261 workgroups = two rounds).
  static constexpr bool HALF = (BM / WM) % 32 == 16;
  static_assert(!HALF || (WM_ == 1 && A_KCONTIG_ && !BF16_ && G_ == 4), "half blocks: one wave row, k-contiguous A, fp32");
  static constexpr int A_STRIDE = A_KCONTIG ? KSTR : BM;
  static constexpr int B_STRIDE = B_KCONTIG ? KSTR : BN;
  static constexpr int A_FLOATS = A_KCONTIG ? BM * KSTR : KC * BM;
  static constexpr int B_FLOATS = B_KCONTIG ? BN * KSTR : KC * BN;
  static constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
  static constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;
  // PERM (all-m-contiguous 4 x 2-block wave tiles, i.e. the 256x256 TN tile): lane r of a wave feeds the MFMAs of
  // its 4 row blocks with tile rows 4r .. 4r+3 and those of its 2 column blocks with tile columns 2r, 2r+1, so ONE
  // ds_read_b128 / ds_read_b64 per k replaces 4 / 2 ds_read_b32 (the TN loop issued 192 LDS reads per 128 MFMAs).
  // Block mb / nb then holds rows 4 i + mb / columns 2 j + nb of the wave's 128 x 64 patch: see tn_store().
  static constexpr bool PERM = !A_KCONTIG_ && !B_KCONTIG_ && BM_ / WM_ == 128 && BN_ / WN_ == 64;
  static_assert((BM % (WM * 32) == 0 || HALF) && BN % (WN * 32) == 0, "wave tile must be a multiple of 32x32 (+ a half block)");
};

// thread -> (row, 16-byte column) maps of one staged chunk --------------------------------
template <int ROWS, int NT>
struct KMap {   // k-contiguous image [ROWS][KC]: 8 float4 per row
  static constexpr int NV = (ROWS * (KC / 4) + NT - 1) / NT;
  static constexpr bool EXACT = (ROWS * (KC / 4)) % NT == 0;
  __device__ static inline int idx(int tid, int i) { return tid + i * NT; }
  __device__ static inline bool valid(int tid, int i) { return EXACT || idx(tid, i) < ROWS * (KC / 4); }
  __device__ static inline int row(int tid, int i) { return idx(tid, i) >> 3; }
  __device__ static inline int c4(int tid, int i) { return idx(tid, i) & 7; }
  __device__ static inline int lds(int tid, int i) { return row(tid, i) * KSTR + 4 * c4(tid, i); }
};
template <int ROWS, int NT>
struct MMap {   // m-contiguous image [KC][ROWS]: ROWS/4 float4 per k-row
  static constexpr int VPR = ROWS / 4;
  static constexpr int NV = (KC * VPR + NT - 1) / NT;
  static constexpr bool EXACT = (KC * VPR) % NT == 0;
  __device__ static inline int idx(int tid, int i) { return tid + i * NT; }
  __device__ static inline bool valid(int tid, int i) { return EXACT || idx(tid, i) < KC * VPR; }
  __device__ static inline int krow(int tid, int i) { return idx(tid, i) / VPR; }
  __device__ static inline int c4(int tid, int i) { return idx(tid, i) % VPR; }
  __device__ static inline int lds(int tid, int i) { return krow(tid, i) * ROWS + 4 * c4(tid, i); }
};

// ---- buffer loads -----------------------------------------------------------------------------
// Every operand is read with `buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen`: a 4-SGPR resource
// (base, num_records), a per-thread byte offset computed ONCE, and a scalar offset that advances per
// chunk.  Two reasons, both measured:
//   * no per-chunk 64-bit VGPR address arithmetic.  With global_load, hipcc re-derived the addresses each
//     chunk into VGPRs it believed still had loads in flight and put s_waitcnt vmcnt(0) in front of the
//     second operand's loads, serialising them;
//   * the hardware range check returns 0 for offsets >= num_records, which is exactly the zero-fill a
//     ragged last tile needs (rows beyond the bag, k beyond the split) -- no clamps, no selects.
// Limits: an operand must be < 4 GiB (checked by the launchers).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;   // a byte offset no operand reaches: reads as zero

__device__ inline rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ inline float4 bld4(rsrc_t r, unsigned voff, unsigned soff) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  float4 f;
  f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
  return f;
}
// The same with DEVICE scope (sc1): what one workgroup stores this way, a workgroup on ANOTHER XCD reads back this way
// (the XCDs' L2s are not coherent with each other for plain accesses; LLVM's gfx942 memory model spells an agent-scope
// atomic load / store as exactly this bit).  Used for the K-split partial tiles, nothing else.
__device__ inline float4 bld4_dev(rsrc_t r, unsigned voff, unsigned soff) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 16);
  float4 f;
  f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
  return f;
}
__device__ inline void bst4_dev(rsrc_t r, unsigned voff, unsigned soff, const float4& f) {
  u32x4 v = {__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, 16);
}
__device__ inline float bld1(rsrc_t r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// ---- generic loaders ----------------------------------------------------------------------
// k-contiguous source S[row][k] (leading dimension ld); rows >= nrows read as zero.  The k range
// may be a concatenation of `nseg` equal segments held in separate buffers (radio: the four
// modality bags are never concatenated in memory, models/model_attention_mil_radio.py:80-82).
template <int ROWS, int NT>
struct LoadK {
  using Map = KMap<ROWS, NT>;
  rsrc_t r0, r1, r2, r3;
  int kseg, tid;
  int kt0 = 0;                 // first chunk of this workgroup's K range (K-split launches; 0 otherwise)
  unsigned voff[Map::NV];
  float4 r[Map::NV];
  __device__ inline void set_offsets(int ld, int row0, int nrows) {
    tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      int rr = row0 + Map::row(tid, i);
      voff[i] = (Map::valid(tid, i) && rr < nrows) ? ((unsigned)rr * (unsigned)ld + 4u * Map::c4(tid, i)) * 4u : OOB;
    }
  }
  __device__ inline void init(const float* p0, int ld, int row0, int nrows) {
    r0 = r1 = r2 = r3 = make_rsrc(p0, (unsigned)nrows * (unsigned)ld * 4u);
    kseg = 1 << 30;
    set_offsets(ld, row0, nrows);
  }
  __device__ inline void init_segments(const float* const* p, int nseg, int kseg_, int ld, int row0, int nrows) {
    const unsigned bytes = (unsigned)nrows * (unsigned)ld * 4u;
    r0 = make_rsrc(p[0], bytes);
    r1 = make_rsrc(p[nseg > 1 ? 1 : 0], bytes);
    r2 = make_rsrc(p[nseg > 2 ? 2 : 0], bytes);
    r3 = make_rsrc(p[nseg > 3 ? 3 : 0], bytes);
    kseg = kseg_;
    set_offsets(ld, row0, nrows);
  }
  __device__ inline void load(int kt) {
    const int k0 = (kt + kt0) * KC;
    const int sidx = k0 / kseg;
    const rsrc_t rs = sidx == 0 ? r0 : (sidx == 1 ? r1 : (sidx == 2 ? r2 : r3));
    const unsigned soff = (unsigned)(k0 - sidx * kseg) * 4u;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) r[i] = bld4(rs, voff[i], soff);
  }
  __device__ inline void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < Map::NV; ++i)
      if (Map::valid(tid, i)) st4(lds + Map::lds(tid, i), r[i]);
  }
  // the same in two halves (vector slots [0, NV/2) and [NV/2, NV)): see the 8-slot staging schedule of gemm_mainloop
  static constexpr bool kHalves = true;
  static constexpr int HV = Map::NV / 2;
  __device__ inline void load_half(int kt, int h) {
    const int k0 = (kt + kt0) * KC;
    const int sidx = k0 / kseg;
    const rsrc_t rs = sidx == 0 ? r0 : (sidx == 1 ? r1 : (sidx == 2 ? r2 : r3));
    const unsigned soff = (unsigned)(k0 - sidx * kseg) * 4u;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i)
      if ((i < HV) == (h == 0)) r[i] = bld4(rs, voff[i], soff);
  }
  __device__ inline void store_half(float* lds, int h) const {
#pragma unroll
    for (int i = 0; i < Map::NV; ++i)
      if ((i < HV) == (h == 0) && Map::valid(tid, i)) st4(lds + Map::lds(tid, i), r[i]);
  }
};

// m-contiguous source S[k][m]; k rows outside [kbase, kmax) and columns >= ncols read as zero
// (the resource ends at row kmax, so the range check does the k tail).
template <int ROWS, int NT>
struct LoadM {
  using Map = MMap<ROWS, NT>;
  rsrc_t rs;
  unsigned ldb, kbase_b;
  int tid;
  unsigned voff[Map::NV];
  float4 r[Map::NV];
  __device__ inline void init(const float* s, int ld, int col0, int ncols, int kbase, int kmax) {
    tid = threadIdx.x;
    rs = make_rsrc(s, (unsigned)(kmax > 0 ? kmax : 0) * (unsigned)ld * 4u);
    ldb = (unsigned)ld * 4u;
    kbase_b = (unsigned)kbase * ldb;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      int c = col0 + 4 * Map::c4(tid, i);
      voff[i] = (Map::valid(tid, i) && c < ncols) ? (unsigned)Map::krow(tid, i) * ldb + (unsigned)c * 4u : OOB;
    }
  }
  __device__ inline void load(int kt) {
    const unsigned soff = kbase_b + (unsigned)(kt * KC) * ldb;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) r[i] = bld4(rs, voff[i], soff);
  }
  __device__ inline void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < Map::NV; ++i)
      if (Map::valid(tid, i)) st4(lds + Map::lds(tid, i), r[i]);
  }
  static constexpr bool kHalves = true;
  static constexpr int HV = Map::NV / 2;
  __device__ inline void load_half(int kt, int h) {
    const unsigned soff = kbase_b + (unsigned)(kt * KC) * ldb;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i)
      if ((i < HV) == (h == 0)) r[i] = bld4(rs, voff[i], soff);
  }
  __device__ inline void store_half(float* lds, int h) const {
#pragma unroll
    for (int i = 0; i < Map::NV; ++i)
      if ((i < HV) == (h == 0) && Map::valid(tid, i)) st4(lds + Map::lds(tid, i), r[i]);
  }
};

template <class L, class = void>
struct has_halves : std::false_type {};
template <class L>
struct has_halves<L, std::void_t<decltype(L::kHalves)>> : std::true_type {};

// ---- MFMA over one staged chunk ---------------------------------------------------------------
// Fragments of k-group q+1 are read from LDS into a second register set BEFORE the 4*MB*NB MFMAs of
// group q are issued, so an LDS read has a whole MFMA block (>= 1024 cycles) to land.  (Left to
// itself hipcc emitted read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs, exposing the LDS latency every 256
// cycles: 55 % MFMA utilisation on the TN kernel.)  Registers are free here: LDS already limits the
// kernels to 2 waves per SIMD.
// Fragment pipeline.  A chunk is a sequence of STEPS = (fragment group g) x (row-block part): tall tiles (MB > 4)
// are cut in two row-block parts so that only half of the A fragments are live at a time (the 224-row tiles
// would otherwise need 2 x 28 registers for double-buffered A fragments alone).  The fragments of step s+1 are
// read from LDS while the MFMAs of step s issue, so an LDS read has >= 12 MFMAs (768 cycles) to land.
template <class T>
struct FragA { float v[T::MB > 4 ? (T::MB + 1) / 2 : T::MB][T::G]; };
template <class T>
struct FragB { float v[T::NB][T::G]; };

template <class T>
__device__ inline void read_a(const float* __restrict__ As, int g, int lo, int hi, int arow, int hh, FragA<T>& f) {
  constexpr int G = T::G;
#pragma unroll
  for (int mb = lo; mb < hi; ++mb) {
    if constexpr (T::A_KCONTIG) {
      float4 t = ld4(As + (arow + mb * 32) * KSTR + 8 * g + 4 * hh);
      f.v[mb - lo][0] = t.x; f.v[mb - lo][1] = t.y; f.v[mb - lo][2] = t.z; f.v[mb - lo][3] = t.w;
    } else if constexpr (!T::PERM) {
#pragma unroll
      for (int j = 0; j < G; ++j) f.v[mb - lo][j] = As[(2 * G * g + G * hh + j) * T::BM + arow + mb * 32];
    }
  }
  if constexpr (T::PERM) {        // arow = patch base + 4 r; lo = 0, hi = 4
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const float4 t = ld4(As + (2 * G * g + G * hh + j) * T::BM + arow);
      f.v[0][j] = t.x; f.v[1][j] = t.y; f.v[2][j] = t.z; f.v[3][j] = t.w;
    }
  }
}
template <class T>
__device__ inline void read_b(const float* __restrict__ Bs, int g, int brow, int hh, FragB<T>& f) {
  constexpr int G = T::G;
#pragma unroll
  for (int nb = 0; nb < T::NB; ++nb) {
    if constexpr (T::B_KCONTIG) {
      float4 t = ld4(Bs + (brow + nb * 32) * KSTR + 8 * g + 4 * hh);
      f.v[nb][0] = t.x; f.v[nb][1] = t.y; f.v[nb][2] = t.z; f.v[nb][3] = t.w;
    } else if constexpr (!T::PERM) {
#pragma unroll
      for (int j = 0; j < G; ++j) f.v[nb][j] = Bs[(2 * G * g + G * hh + j) * T::BN + brow + nb * 32];
    }
  }
  if constexpr (T::PERM) {        // brow = patch base + 2 r
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const float2 t = ld2(Bs + (2 * G * g + G * hh + j) * T::BN + brow);
      f.v[0][j] = t.x; f.v[1][j] = t.y;
    }
  }
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ inline bf16x8 frag_bf16(const float (&v)[4]) {
  f32x4 t = {v[0], v[1], v[2], v[3]};
  return __builtin_bit_cast(bf16x8, t);
}

template <class T>
__device__ inline void mfma_part(const FragA<T>& fa, const FragB<T>& fb, int lo, int hi, f32x16 (&acc)[T::MB][T::NB]) {
  if constexpr (T::BF16) {
#pragma unroll
    for (int mb = lo; mb < hi; ++mb)
#pragma unroll
      for (int nb = 0; nb < T::NB; ++nb)
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_bf16(fa.v[mb - lo]), frag_bf16(fb.v[nb]), acc[mb][nb], 0, 0, 0);
    return;
  }
#pragma unroll
  for (int j = 0; j < T::G; ++j)
#pragma unroll
    for (int mb = lo; mb < hi; ++mb)
#pragma unroll
      for (int nb = 0; nb < T::NB; ++nb)
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.v[mb - lo][j], fb.v[nb][j], acc[mb][nb], 0, 0, 0);
}

// ---- half block (Tile::HALF): rows 32 MB .. 32 MB + 15 of the wave's rows on v_mfma_f32_16x16x4_f32 ---------------
// Operand layout of that instruction: lane (i = lane & 15, kq = lane >> 4) supplies A[i][k_kq] and B[k_kq][j = i];
// D[i][j]: lane holds column j = lane & 15, rows 4 (lane >> 4) + reg.  Inside a fragment group of 8 k-values MFMA t
// (t = 0, 1) contracts k = 8 g + 2 kq + t, so a lane's two A values are adjacent in the k-contiguous image (one
// ds_read_b64, conflict-free with the KC + 4 pad: 16 rows x stride 36 words + 2 kq words hit 64 distinct banks).
typedef float f32x4acc __attribute__((ext_vector_type(4)));
template <class T>
struct FragH { float a[2]; float b[T::NB][2][2]; };     // b[nb][column half c][t]
template <class T>
__device__ inline void read_half(const float* __restrict__ As, const float* __restrict__ Bs, int g, int arow_h, int brow0,
                                 int lane, FragH<T>& f) {
  const int i16 = lane & 15, kq = lane >> 4;
  const float2 ta = ld2(As + (arow_h + i16) * KSTR + 8 * g + 2 * kq);
  f.a[0] = ta.x; f.a[1] = ta.y;
#pragma unroll
  for (int nb = 0; nb < T::NB; ++nb)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if constexpr (T::B_KCONTIG) {
        const float2 tb = ld2(Bs + (brow0 + nb * 32 + 16 * c + i16) * KSTR + 8 * g + 2 * kq);
        f.b[nb][c][0] = tb.x; f.b[nb][c][1] = tb.y;
      } else {
        f.b[nb][c][0] = Bs[(8 * g + 2 * kq) * T::BN + brow0 + nb * 32 + 16 * c + i16];
        f.b[nb][c][1] = Bs[(8 * g + 2 * kq + 1) * T::BN + brow0 + nb * 32 + 16 * c + i16];
      }
    }
}
template <class T>
__device__ inline void mfma_half(const FragH<T>& f, f32x4acc (&acch)[T::NB][2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        acch[nb][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[t], f.b[nb][c][t], acch[nb][c], 0, 0, 0);
}

// `hook(s)` (s = 0..NS-1, chunk_steps<T>()) runs in front of the MFMA block of step s: the main loop uses it
// to spread the staging of the NEXT chunk through this chunk's MFMA stream (global loads in the first half, LDS
// writes in the second) instead of bursting it between two MFMA blocks, where the VMEM issue (throttled by the
// address path: ~2000 cycles for a CU's 60 KB) and the LDS writes stalled every wave at once.
template <class T>
constexpr int chunk_steps() { return (KC / (2 * T::G)) * (T::MB > 4 ? 2 : 1); }

template <class T, class Hook>
__device__ inline void compute_chunk(const float* __restrict__ As, const float* __restrict__ Bs,
                                     f32x16 (&acc)[T::MB][T::NB], int wm, int wn, int lane, Hook&& hook,
                                     f32x4acc (*acch)[2] = nullptr, int ng_valid = 1 << 30) {
  const int r = lane & 31, hh = lane >> 5;
  const int arow = wm * T::MB * 32 + (T::PERM ? 4 * r : r);
  const int brow = wn * T::NB * 32 + (T::PERM ? 2 * r : r);
  constexpr int NG = KC / (2 * T::G);            // fragment groups per chunk: 4 (G = 4) or 8 (G = 2)
  constexpr int NP = T::MB > 4 ? 2 : 1;          // row-block parts
  constexpr int MBH = T::MB > 4 ? (T::MB + 1) / 2 : T::MB;
  constexpr int NS = NG * NP;                    // steps per chunk
  constexpr int PER_Q = NS / 4;
  FragA<T> fa[2];
  FragB<T> fb[2];
  FragH<T> fh;
  const int arow_h = wm * (T::BM / T::WM) + T::MB * 32, brow0 = wn * T::NB * 32;
#ifdef MMF_DIAG_NOFRAG       /* diagnostic build: fragments read once per chunk only (results are wrong) */
  read_b<T>(Bs, 0, brow, hh, fb[0]); read_b<T>(Bs, 1, brow, hh, fb[1]);
  read_a<T>(As, 0, 0, MBH, arow, hh, fa[0]); read_a<T>(As, 1, 0, MBH, arow, hh, fa[1]);
#else
  read_b<T>(Bs, 0, brow, hh, fb[0]);
  read_a<T>(As, 0, 0, MBH, arow, hh, fa[0]);
#endif
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int g = s / NP, part = s % NP;
    const int lo = part == 0 ? 0 : MBH, hi = part == 0 ? MBH : T::MB;
#ifndef MMF_DIAG_NOFRAG
    if (s + 1 < NS) {                            // prefetch the fragments of step s + 1
      const int g1 = (s + 1) / NP, part1 = (s + 1) % NP;
      if (part1 == 0) read_b<T>(Bs, g1, brow, hh, fb[g1 & 1]);
      read_a<T>(As, g1, part1 == 0 ? 0 : MBH, part1 == 0 ? MBH : T::MB, arow, hh, fa[(s + 1) & 1]);
    }
#endif
    if constexpr (T::HALF) {                     // the half block's fragments: read in the first part of group g,
      if (part == 0) read_half<T>(As, Bs, g, arow_h, brow0, lane, fh);   // multiplied behind its last part
    }
    hook(s);
#ifndef MMF_DIAG_NOSCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef MMF_DIAG_NOMFMA      /* diagnostic builds (tools/diag_build.py): timing only, results are wrong */
    // ng_valid: fragment groups of this chunk that hold data (split-K launches cut K at multiples of 4 instances, so
    // every workgroup's LAST chunk is partly zero fill); one scalar branch per step, same registers as without it
    if (g < ng_valid) mfma_part<T>(fa[s & 1], fb[g & 1], lo, hi, acc);
    if constexpr (T::HALF) {
      if (part == NP - 1) mfma_half<T>(fh, *reinterpret_cast<f32x4acc (*)[T::NB][2]>(acch));
    }
#endif
#ifndef MMF_DIAG_NOSCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

// Diagnostic build only (-DMMF_STAMPS): s_memtime phase stamps of the main loop, summed per wave into a
// per-translation-unit device array that no kernel reads (guide: "In-kernel stamps").  The shipped
// library is built without the macro and contains no stamp.
#ifdef MMF_STAMPS
static __device__ unsigned long long g_stamps[8];
__device__ inline unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#ifdef MMF_STAMPS_LIGHT        /* kernel-level stamps only: nothing inside the main loop */
#define MMF_STAMP(var)
#else
#define MMF_STAMP(var) unsigned long long var = stamp_now()
#endif
#define MMF_KSTAMP(var) unsigned long long var = stamp_now()
#else
#define MMF_STAMP(var)
#define MMF_KSTAMP(var)
#endif

// DEPHASE (8-wave tiles only): the two waves of a SIMD (w and w + 4) run the same code in phase, so they stall in the
// same unit at the same moment.  With DEPHASE the second group stages half a chunk out of phase: it writes chunk
// kt+1 (requested one chunk earlier) in quarters 0 / 1 and requests chunk kt+2 in quarters 2 / 3.  Pays only where
// the staging path holds real VALU work (K-dh builds its A operand there: -3 us); plain copies got slower.
template <class T, class LA, class LB, bool DEPHASE = false>
__device__ inline void gemm_mainloop(LA& la, LB& lb, int nk, float* lds, f32x16 (&acc)[T::MB][T::NB],
                                     f32x4acc (*acch)[2] = nullptr, int last_groups = 1 << 30) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
#pragma unroll
  for (int mb = 0; mb < T::MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  if constexpr (T::HALF) {
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) acch[nb][c][i] = 0.f;
  }
  if (nk <= 0) return;
#if defined(MMF_STAMPS) && !defined(MMF_STAMPS_LIGHT)
  unsigned long long s_load = 0, s_mfma = 0, s_store = 0, s_bar = 0;
#endif
#if defined(MMF_STAMPS) && defined(MMF_STAMP_FIRST_STAGE)
  const unsigned long long t_enter = stamp_now();
#endif
  la.load(0);
  lb.load(0);
  la.store(lds);
  lb.store(lds + T::A_FLOATS);
  const bool late = DEPHASE && T::NT == 512 && wave >= 4;
#ifndef MMF_DIAG_NOLOAD
  if (late && nk > 1) { la.load(1); lb.load(1); }
#endif
  __syncthreads();
#if defined(MMF_STAMPS) && defined(MMF_STAMP_FIRST_STAGE)   /* [2]: first stage (load + LDS write + barrier), [3]: count */
  if (lane == 0) { atomicAdd(&g_stamps[2], stamp_now() - t_enter); atomicAdd(&g_stamps[3], 1ull); }
#endif
  // Static priority for the second-dispatched half of an 8-wave workgroup (MI355X_MICROARCH.md, two waves per SIMD, item 4):
  // the younger wave of every SIMD loses each issue arbitration otherwise.  Measured on the 50k step, same box, two alternating
  // pairs: K-dh -2.3 us, the projection -0.8, the gate -0.5; the TN tiles (m-contiguous operands) +1.0, so they keep priority 0.
#ifndef MMF_GEMM_PRIO
#define MMF_GEMM_PRIO 1
#endif
  constexpr bool half_prio = MMF_GEMM_PRIO && T::NT == 512 && T::A_KCONTIG;
  if (half_prio && wave >= 4) __builtin_amdgcn_s_setprio(1);
  for (int kt = 0; kt < nk; ++kt) {
    float* cur = lds + (kt & 1) * T::STAGE_FLOATS;
    float* nxt = lds + ((kt + 1) & 1) * T::STAGE_FLOATS;
    const bool more = kt + 1 < nk;
    const bool more2 = kt + 2 < nk;
    MMF_STAMP(t0);
    MMF_STAMP(t1);
    // a K range that ends inside its last chunk: only the first `last_groups` fragment groups of that chunk hold data,
    // the rest is zero fill -- and multiplying zeros costs what multiplying data costs (see compute_chunk: ng_valid)
    const int ng_valid = more ? (1 << 30) : last_groups;
    compute_chunk<T>(cur, cur + T::A_FLOATS, acc, wm, wn, lane, [&](int s) {
      if (!more) return;
#ifdef MMF_DIAG_NOLOAD
      return;
#endif
      constexpr int NS = chunk_steps<T>();
      if constexpr (NS == 8 && !DEPHASE && has_halves<LA>::value && has_halves<LB>::value) {
        // plain copies on both sides: one half-operand per step, so that no step issues more than 2-4 VMEM or LDS
        // instructions per lane (the 4-slot schedule wrote a whole operand, 30 KB per CU, in one burst)
#ifdef MMF_DIAG_NOGLOAD
        if (s < 4) return;
#endif
        switch (s) {
          case 0: la.load_half(kt + 1, 0); break;
          case 1: la.load_half(kt + 1, 1); break;
          case 2: lb.load_half(kt + 1, 0); break;
          case 3: lb.load_half(kt + 1, 1); break;
          case 4: la.store_half(nxt, 0); break;
          case 5: la.store_half(nxt, 1); break;
          case 6: lb.store_half(nxt + T::A_FLOATS, 0); break;
          default: lb.store_half(nxt + T::A_FLOATS, 1); break;
        }
        return;
      }
      if (s % (NS / 4) != 0) return;
      const int q = s / (NS / 4);
#ifdef MMF_DIAG_NOGLOAD        /* diagnostic build: LDS writes of stale registers, no global loads (results are wrong) */
      if (q == 2) la.store(nxt);
      else if (q == 3) lb.store(nxt + T::A_FLOATS);
      return;
#endif
      if (late) {
        if (q == 0) la.store(nxt);
        else if (q == 1) lb.store(nxt + T::A_FLOATS);
        else if (q == 2) { if (more2) la.load(kt + 2); }
        else { if (more2) lb.load(kt + 2); }
        return;
      }
      if (q == 0) la.load(kt + 1);
      else if (q == 1) lb.load(kt + 1);
      else if (q == 2) la.store(nxt);
      else lb.store(nxt + T::A_FLOATS);
    }, acch, ng_valid);
    MMF_STAMP(t2);
    MMF_STAMP(t3);
#ifndef MMF_DIAG_NOBAR        /* diagnostic build: no barrier between chunks (results are wrong) */
    __syncthreads();
#endif
    MMF_STAMP(t4);
#if defined(MMF_STAMPS) && !defined(MMF_STAMPS_LIGHT)
    s_load += t1 - t0; s_mfma += t2 - t1; s_store += t3 - t2; s_bar += t4 - t3;
#endif
  }
  if (half_prio) __builtin_amdgcn_s_setprio(0);
#if defined(MMF_STAMPS) && !defined(MMF_STAMPS_LIGHT)
  if (lane == 0) {
    atomicAdd(&g_stamps[0], s_load); atomicAdd(&g_stamps[1], s_mfma);
    atomicAdd(&g_stamps[2], s_store); atomicAdd(&g_stamps[3], s_bar);
    atomicAdd(&g_stamps[4], (unsigned long long)nk);
  }
#endif
}

// Deep register prefetch for SHORT grids (small bags: a few dozen workgroups, one per CU, nothing else on the CU to
// hide a memory round trip).  The plain loop above requests chunk kt+1 while chunk kt computes; a 64x64 tile computes
// a chunk in ~1000 cycles, a load takes 2-4 thousand, so every chunk waited for its operands (N = 1k: 32 chunks x 0.9 us
// for 0.4 us of MFMA each).  Here D copies of the loaders hold D chunks in flight: chunk kt+D is requested while chunk
// kt computes and written to LDS D-1 chunks later.  LDS stays double-buffered.  The kt loop is unrolled by D so that
// the loader copies are addressed at compile time (they live in registers).  Loaders must be plain copies (their
// load(kt) / store(lds) pair may be split across copies): LoadK, LoadM, LoadGateW.  nk % D == 0 (checked by the
// launchers), and the body is branch-free on purpose: the last D chunks request operands past the end (a buffer load
// beyond the range returns zeros, within it stale columns) that nobody consumes, and the last store fills a stage
// nobody reads.  With a branch around a load or a store hipcc can no longer count the loads in flight and falls back
// to s_waitcnt vmcnt(0) in front of every LDS write -- which drains exactly the prefetch this loop exists for.
template <class T, int D, class LA, class LB>
__device__ inline void gemm_mainloop_deep(const LA& la0, const LB& lb0, int nk, float* lds, f32x16 (&acc)[T::MB][T::NB]) {
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
  constexpr int NS = chunk_steps<T>();
  static_assert((D & 1) == 0, "an even D keeps the LDS stage of copy j fixed");
  for (int kt0 = 0; kt0 < nk; kt0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int kt = kt0 + j;
      float* cur = lds + (j & 1) * T::STAGE_FLOATS;
      float* nxt = lds + ((j + 1) & 1) * T::STAGE_FLOATS;
      compute_chunk<T>(cur, cur + T::A_FLOATS, acc, wm, wn, lane, [&](int s) {
        if (s % (NS / 4) != 0) return;
        const int q = s / (NS / 4);
        if (q == 0) la[j].load(kt + D);                       // copy j has been written out: reuse it
        else if (q == 1) lb[j].load(kt + D);
        else if (q == 2) la[(j + 1) % D].store(nxt);
        else lb[(j + 1) % D].store(nxt + T::A_FLOATS);
      });
      __syncthreads();
    }
  }
}

// visit every accumulator element this lane owns: f(row_in_tile, col_in_tile, value)
template <class T, class F>
__device__ inline void for_each_c(f32x16 (&acc)[T::MB][T::NB], F&& f) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int mb = 0; mb < T::MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int row = (wm * T::MB + mb) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        int col = (wn * T::NB + nb) * 32 + r;
        f(row, col, acc[mb][nb][i]);
      }
}

// Row-major epilogue.  The MFMA accumulator layout gives a lane ONE column and 16 rows per 32x32 block, so a
// direct store is 16 scalar dword stores per block and lane -- store-issue bound (measured: the epilogue of the
// 224x256 tile took half as long as its 32-chunk main loop).  Here every wave transposes one block at a time
// through a private 32x36-float LDS scratch and hands the functor float4s that are 4 CONSECUTIVE COLUMNS of one
// row: f(row_in_tile, col_in_tile, float4) is called 4 times per block and lane, and a functor's
// global_store_dwordx4 covers 8 rows x 128 contiguous bytes per wave-instruction.
// Call only after the main loop (its last barrier has retired every LDS read); needs NT/64 * 4608 bytes of LDS.
constexpr int EPI_STRIDE = 36;
// f(mb, nb, row_in_tile, col_in_tile, v[4]): v[t] is row (row_in_tile + 8 t), columns col_in_tile .. +3.
// mb / nb are compile-time after unrolling, so a functor can index register arrays it preloaded per column
// strip (bias, ...).  Functors must issue ALL their global loads before their first store: loads and stores share
// the in-order vmcnt counter, so a load behind a store waits for that store to drain (this alone made the
// first row-major epilogue as slow as the scalar one).
template <class T, class F>
__device__ inline void epilogue_rows(f32x16 (&acc)[T::MB][T::NB], float* lds, F&& f) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int r = lane & 31, hh = lane >> 5;
  float* blk = lds + wave * (32 * EPI_STRIDE);
  const int rr = lane >> 3, c4 = lane & 7;
#ifdef MMF_DIAG_EPI1          /* diagnostic build: only the first row block is written out (results are wrong) */
  constexpr int MB_OUT = 1;
  {
    float t = 0.f;
    for (int mb = 1; mb < T::MB; ++mb)
      for (int nb = 0; nb < T::NB; ++nb)
        for (int i = 0; i < 16; ++i) t += acc[mb][nb][i];
    if (t == 1.2345e30f) blk[0] = t;
  }
#else
  constexpr int MB_OUT = T::MB;
#endif
#pragma unroll
  for (int mb = 0; mb < MB_OUT; ++mb)
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) blk[((i & 3) + 8 * (i >> 2) + 4 * hh) * EPI_STRIDE + r] = acc[mb][nb][i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave, in-order LDS queue: writes land before the reads
      float4 v[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = ld4(blk + (rr + 8 * t) * EPI_STRIDE + 4 * c4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next block overwrites the scratch
      f(mb, nb, (wm * T::MB + mb) * 32 + rr, (wn * T::NB + nb) * 32 + 4 * c4, v);
    }
}
// column (within the tile) of this lane's float4 in block-column nb of its wave: for preloading per-column data
template <class T>
__device__ inline int epilogue_col(int nb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  return ((wave % T::WN) * T::NB + nb) * 32 + 4 * (lane & 7);
}

// one 32x32 accumulator block -> 4 float4 per lane (row rr + 8t, columns 4*c4 .. 4*c4+3), via the wave's scratch
__device__ inline void transpose_block(const f32x16& a, float* blk, int lane, float4 (&out)[4]) {
  const int r = lane & 31, hh = lane >> 5, rr = lane >> 3, c4 = lane & 7;
#pragma unroll
  for (int i = 0; i < 16; ++i) blk[((i & 3) + 8 * (i >> 2) + 4 * hh) * EPI_STRIDE + r] = a[i];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 0; t < 4; ++t) out[t] = ld4(blk + (rr + 8 * t) * EPI_STRIDE + 4 * c4);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// the half block of one column block: two 16x16 accumulators (columns 16 c .. 16 c + 15) -> 2 float4 per lane
// (row rr + 8 t of the half block's 16 rows, columns 4*c4 .. 4*c4+3), via the wave's scratch
__device__ inline void transpose_half(const f32x4acc& c0, const f32x4acc& c1, float* blk, int lane, float4 (&out)[2]) {
  const int j = lane & 15, q = lane >> 4, rr = lane >> 3, c4 = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    blk[(4 * q + i) * EPI_STRIDE + j] = c0[i];
    blk[(4 * q + i) * EPI_STRIDE + 16 + j] = c1[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 0; t < 2; ++t) out[t] = ld4(blk + (rr + 8 * t) * EPI_STRIDE + 4 * c4);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <class T>
constexpr int epilogue_lds_bytes() { return (T::NT / 64) * 32 * EPI_STRIDE * 4; }

// XCD-aware block -> (m-tile, n-tile) map.  Blocks b and b+8 land on the same XCD (round-robin
// dispatch; speed only, never correctness), so the n-tiles of one m-tile are put 8 apart: they
// share the streamed A rows through that XCD's L2.  Returns false for the padding blocks of a
// grid rounded up to a multiple of 8*ntn.
__device__ inline bool tile_of_block(int b, int mt_count, int ntn, int& mt, int& nt) {
  int group = b / (8 * ntn);
  int within = b - group * 8 * ntn;
  nt = within / 8;
  mt = group * 8 + (within & 7);
  return mt < mt_count;
}
inline int grid_for_tiles(int mt_count, int ntn) { return ((mt_count + 7) / 8) * 8 * ntn; }

}  // namespace mmf
