// Backward kernels of the attention-MIL stack on MI355X (gfx950).  The reference has no
// hand-written backward: these kernels are what autograd derives for
// models/model_attention_mil_path.py:52-56 + models/model_modules.py:105-110
// (math: SURVEY.md Appendix A.4).
//
//   K-prep  p_i = softmax weight, ds_i = p_i (dM.h_i - dM.M) + gA_i
//   K-dh    du = (dP.Wab + p dM) . relu'(h) . mask-scale      NN GEMM, dP built on the fly
//   K-tn    dW1 = du^T.x, dWab = dP^T.h (+ bias / Wc column sums)   split-K TN GEMM -> slabs
//   K-nn    plain NN GEMM (radio: d(reduce_dim output) = du.W1)
//   K-red   deterministic slab reduction
//
// dP[i][k]:  gated  k <  D : ds_i Wc[k] b_d (1 - a^2) m_a        (d pre-tanh)
//                   k >= D : ds_i Wc[k'] a_d b (1 - b) m_b       (d pre-sigmoid), k' = k - D
//            ungated       : ds_i Wc[k] (1 - a^2) m_a
// with a_d = a m_a, b_d = b m_b the dropped activations (m = keep/(1-p); m = 1 in eval).
#include <cstdlib>

#include <type_traits>

#include "mmf_gemm_core.h"
#include "mmf_gemm_split.h"
#include "mmf_kernels.h"

namespace mmf {

// =============================================================================================
// K-prep
// =============================================================================================
__global__ __launch_bounds__(256) void bwd_prep_kernel(BwdPrepParams p) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float m = p.stats[0], inv = 1.0f / p.stats[1];
  // dM.M, one value per wave (H <= 1024)
  float dmm = 0.f;
  for (int c = lane; c < p.H; c += 64) dmm += p.dM[c] * p.M[c];
  dmm = wave_sum(dmm);
  float dbc = 0.f;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < p.N; i += nw) {
    const float* hr = p.h + (size_t)i * p.H;
    float g = 0.f;
    for (int c = 4 * lane; c < p.H; c += 256) {
      float4 hv = ld4(hr + c), dv = ld4(p.dM + c);
      g += hv.x * dv.x + hv.y * dv.y + hv.z * dv.z + hv.w * dv.w;
    }
    g = wave_sum(g);
    if (lane == 0) {
      float pi = __expf(p.A_raw[i] - m) * inv;
      float d = pi * (g - dmm) + (p.gA ? p.gA[i] : 0.f);
      p.p[i] = pi;
      p.ds[i] = d;
      dbc += d;
    }
  }
  if (lane == 0) red[wave] = dbc;
  __syncthreads();
  if (tid == 0) p.dbc_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// =============================================================================================
// K-dh : NN GEMM with on-the-fly A operand
// =============================================================================================
// MODE < 0: gate / dropout switches read at run time; MODE = 2 * gated + dropout: compiled in (wide tiles)
template <int ROWS, int NT, int MODE = -1>
struct LoadP_K {   // A[i][k] = dP, k-contiguous image
  using Map = KMap<ROWS, NT>;
  GateBwdCtx g;
  rsrc_t ra, rb, rwc;
  int row0, tid, part, d0;
  uint32_t thr;
  float dscale;
  unsigned voff[Map::NV];
  float dsr[Map::NV];
  float4 ra4[Map::NV], rb4[Map::NV], wc4;
  __device__ inline void init(const GateBwdCtx& g_, int row0_, int nrows) {
    g = g_; g.resolve_seed(); row0 = row0_; tid = threadIdx.x;
    thr = drop_threshold(g.drop_p);
    dscale = g.drop_p > 0.f ? 1.0f / (1.0f - g.drop_p) : 1.0f;
    const unsigned bytes = (unsigned)nrows * (unsigned)g.D * 4u;
    ra = make_rsrc(g.a, bytes);
    rb = make_rsrc(g.gated ? g.b : g.a, bytes);
    rwc = make_rsrc(g.Wc, (unsigned)g.D * 4u);
    rsrc_t rds = make_rsrc(g.ds, (unsigned)nrows * 4u);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      int rr = row0 + Map::row(tid, i);
      bool ok = Map::valid(tid, i) && rr < nrows;
      voff[i] = ok ? ((unsigned)rr * (unsigned)g.D + 4u * Map::c4(tid, i)) * 4u : OOB;
      dsr[i] = bld1(rds, ok ? (unsigned)rr * 4u : OOB, 0);     // rows beyond the bag: ds = 0 => dP = 0
    }
  }
  // ds of the tile's rows already sits in LDS (fused prep); otherwise identical to init()
  __device__ inline void init_lds(const GateBwdCtx& g_, int row0_, int nrows, const float* ds_lds) {
    g = g_; g.resolve_seed(); row0 = row0_; tid = threadIdx.x;
    thr = drop_threshold(g.drop_p);
    dscale = g.drop_p > 0.f ? 1.0f / (1.0f - g.drop_p) : 1.0f;
    const unsigned bytes = (unsigned)nrows * (unsigned)g.D * 4u;
    ra = make_rsrc(g.a, bytes);
    rb = make_rsrc(g.gated ? g.b : g.a, bytes);
    rwc = make_rsrc(g.Wc, (unsigned)g.D * 4u);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      int rl = Map::row(tid, i), rr = row0 + rl;
      bool ok = Map::valid(tid, i) && rr < nrows;
      voff[i] = ok ? ((unsigned)rr * (unsigned)g.D + 4u * Map::c4(tid, i)) * 4u : OOB;
      dsr[i] = ok ? ds_lds[rl] : 0.f;
    }
  }
  // k order of the gated contraction: chunk 2j = d pre-tanh, chunk 2j + 1 = d pre-sigmoid of the SAME 32 attention
  // dims (LoadWab_M walks [Wa ; Wb] in the same order; a sum does not care).  Both halves are built from the same
  // a, b, Wc values, so the odd chunk issues no loads and reuses the even chunk's registers: a and b cross HBM once
  // per tile instead of twice (PMC, 50k bag: 369 MB per launch with the [all of Wa | all of Wb] order).
  __device__ inline void load_meta(int kt) {
    part = g.gated ? (kt & 1) : 0;
    d0 = (g.gated ? (kt >> 1) : kt) * KC;
  }
  __device__ inline void load(int kt) {
    load_meta(kt);
    if (part) return;
    const unsigned soff = (unsigned)d0 * 4u;
    wc4 = bld4(rwc, 16u * (tid & 7), soff);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      ra4[i] = bld4(ra, voff[i], soff);
      rb4[i] = bld4(rb, voff[i], soff);
    }
  }
  template <bool GATED, bool DROP, int PART>
  __device__ inline void store_t(float* lds) const {
    const int c = d0 + 4 * (tid & 7);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      if (!Map::valid(tid, i)) continue;
      int rr = row0 + Map::row(tid, i);
      uint32_t idx = (uint32_t)rr * (uint32_t)g.D + (uint32_t)c;
      float dummy;
      float4 o;
      o.x = gate_dp_t<GATED, DROP, PART>(g, ra4[i].x, rb4[i].x, wc4.x, dsr[i], idx + 0, thr, dscale, dummy);
      o.y = gate_dp_t<GATED, DROP, PART>(g, ra4[i].y, rb4[i].y, wc4.y, dsr[i], idx + 1, thr, dscale, dummy);
      o.z = gate_dp_t<GATED, DROP, PART>(g, ra4[i].z, rb4[i].z, wc4.z, dsr[i], idx + 2, thr, dscale, dummy);
      o.w = gate_dp_t<GATED, DROP, PART>(g, ra4[i].w, rb4[i].w, wc4.w, dsr[i], idx + 3, thr, dscale, dummy);
      st4(lds + Map::lds(tid, i), o);
    }
  }
  __device__ inline void store(float* lds) const {
    if constexpr (MODE >= 0) {
      constexpr bool GATED = (MODE & 2) != 0, DROP = (MODE & 1) != 0;
      if (GATED && part) store_t<GATED, DROP, 1>(lds);      // one wave-uniform branch per chunk
      else store_t<GATED, DROP, 0>(lds);
      return;
    }
    const int c = d0 + 4 * (tid & 7);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      if (!Map::valid(tid, i)) continue;
      int rr = row0 + Map::row(tid, i);
      uint32_t idx = (uint32_t)rr * (uint32_t)g.D + (uint32_t)c;
      float dummy;
      float4 o;
      o.x = gate_dp(g, part, ra4[i].x, rb4[i].x, wc4.x, dsr[i], idx + 0, thr, dscale, dummy);
      o.y = gate_dp(g, part, ra4[i].y, rb4[i].y, wc4.y, dsr[i], idx + 1, thr, dscale, dummy);
      o.z = gate_dp(g, part, ra4[i].z, rb4[i].z, wc4.z, dsr[i], idx + 2, thr, dscale, dummy);
      o.w = gate_dp(g, part, ra4[i].w, rb4[i].w, wc4.w, dsr[i], idx + 3, thr, dscale, dummy);
      st4(lds + Map::lds(tid, i), o);
    }
  }
};

// B[k][n] = rows of Wa / Wb in LoadP_K's k order (gated: chunk 2j = Wa rows 32j.., chunk 2j + 1 = Wb rows 32j..);
// m-contiguous image
template <int ROWS, int NT>
struct LoadWab_M {
  using Map = MMap<ROWS, NT>;
  rsrc_t ra, rb;
  bool gated;
  int D, tid;
  unsigned hb;
  unsigned voff[Map::NV];
  float4 r[Map::NV];
  __device__ inline void init(const float* wa, const float* wb, int H, int D_, int col0, bool gated_) {
    D = D_; tid = threadIdx.x; hb = (unsigned)H * 4u; gated = gated_ && wb != nullptr;
    ra = make_rsrc(wa, (unsigned)D * hb);
    rb = make_rsrc(wb ? wb : wa, (unsigned)D * hb);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      int c = col0 + 4 * Map::c4(tid, i);
      voff[i] = (Map::valid(tid, i) && c < H) ? (unsigned)Map::krow(tid, i) * hb + (unsigned)c * 4u : OOB;
    }
  }
  __device__ inline void load(int kt) {
    const bool second = gated && (kt & 1);
    const unsigned soff = (unsigned)((gated ? (kt >> 1) : kt) * KC) * hb;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) r[i] = bld4(second ? rb : ra, voff[i], soff);
  }
  __device__ inline void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < Map::NV; ++i)
      if (Map::valid(tid, i)) st4(lds + Map::lds(tid, i), r[i]);
  }
};

// Deep-prefetch main loop of K-dh for short grids (see gemm_mainloop_deep): gated stacks only.  The A loader works in
// PAIRS of chunks (chunk 2j loads a, b; chunk 2j + 1 reuses the registers), so it gets two copies, each holding one pair,
// reloaded as soon as the pair's second chunk has been written to LDS; the B loader gets four plain copies.
// nk % 4 == 0; branch-free for the same reason as gemm_mainloop_deep.
template <class T, class LA, class LB>
__device__ inline void dh_mainloop_deep(const LA& la0, const LB& lb0, int nk, float* lds, f32x16 (&acc)[T::MB][T::NB]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
#pragma unroll
  for (int mb = 0; mb < T::MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  LA la[2] = {la0, la0};
  LB lb[4] = {lb0, lb0, lb0, lb0};
  la[0].load(0); la[1].load(2);
#pragma unroll
  for (int j = 0; j < 4; ++j) lb[j].load(j);
  la[0].store(lds);
  lb[0].store(lds + T::A_FLOATS);
  __syncthreads();
  constexpr int NS = chunk_steps<T>();
  for (int kt0 = 0; kt0 < nk; kt0 += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kt = kt0 + j;
      float* cur = lds + (j & 1) * T::STAGE_FLOATS;
      float* nxt = lds + ((j + 1) & 1) * T::STAGE_FLOATS;
      compute_chunk<T>(cur, cur + T::A_FLOATS, acc, wm, wn, lane, [&](int s) {
        if (s % (NS / 4) != 0) return;
        const int q = s / (NS / 4);
        if (q == 0) {
          // copy c's pair is fully written once its odd chunk has been stored, i.e. after step 2c of this round
          if (j == 1) la[0].load(kt0 + 4);
          else if (j == 3) la[1].load(kt0 + 6);
        } else if (q == 1) {
          lb[j].load(kt + 4);
        } else if (q == 2) {
          LA& l = la[((j + 1) & 3) >> 1];
          l.load_meta(kt + 1);                               // part / d0 of the chunk being written (no memory access)
          l.store(nxt);
        } else {
          lb[(j + 1) & 3].store(nxt + T::A_FLOATS);
        }
      });
      __syncthreads();
    }
  }
}

// ---- the same operands for the split-operand core (mmf_gemm_split.h): chunks of 16 attention dims ------------------
// Gated stacks with compiled-in switches only (MODE = 2 | dropout).  k order as LoadP_K: chunk 2j = d pre-tanh, chunk
// 2j + 1 = d pre-sigmoid of dims 16 j .. 16 j + 15; the odd chunk reuses the even chunk's a, b, Wc registers.
template <int ROWS, int NT, int MODE>
struct SplitP_K {
  static_assert(MODE >= 2, "gated, switches compiled in");
  static constexpr bool DROP = (MODE & 1) != 0;
  static constexpr int TOTAL = ROWS * 4, NV = (TOTAL + NT - 1) / NT;
  static_assert(TOTAL >= NT, "tile too small for this thread count");
  GateBwdCtx g;
  rsrc_t ra, rb, rwc;
  int row0, tid, d0;
  uint32_t thr;
  float dscale;
  unsigned voff[NV];
  float dsr[NV];
  float4 ra4[NV], rb4[NV], wc4;
  __device__ static inline int slot(int tid, int i) { const int idx = tid + i * NT; return idx < TOTAL ? idx : idx - NT; }
  __device__ inline void init_common(const GateBwdCtx& g_, int row0_, int nrows) {
    g = g_; g.resolve_seed(); row0 = row0_; tid = threadIdx.x;
    thr = drop_threshold(g.drop_p);
    dscale = g.drop_p > 0.f ? 1.0f / (1.0f - g.drop_p) : 1.0f;
    const unsigned bytes = (unsigned)nrows * (unsigned)g.D * 4u;
    ra = make_rsrc(g.a, bytes);
    rb = make_rsrc(g.b, bytes);
    rwc = make_rsrc(g.Wc, (unsigned)g.D * 4u);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = slot(tid, i), rr = row0 + krow(idx);
      voff[i] = rr < nrows ? ((unsigned)rr * (unsigned)g.D + 4u * (idx & 3)) * 4u : OOB;
    }
  }
  __device__ inline void init(const GateBwdCtx& g_, int row0_, int nrows) {
    init_common(g_, row0_, nrows);
    rsrc_t rds = make_rsrc(g.ds, (unsigned)nrows * 4u);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int rr = row0 + (krow(slot(tid, i)));
      dsr[i] = bld1(rds, rr < nrows ? (unsigned)rr * 4u : OOB, 0);     // rows beyond the bag: ds = 0 => dP = 0
    }
  }
  __device__ inline void init_lds(const GateBwdCtx& g_, int row0_, int nrows, const float* ds_lds) {
    init_common(g_, row0_, nrows);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int rl = krow(slot(tid, i));
      dsr[i] = row0 + rl < nrows ? ds_lds[rl] : 0.f;
    }
  }
  __device__ inline void load_pair(int kt) {          // kt even: a, b, Wc of dims 8 kt .. 8 kt + 15
    d0 = (kt >> 1) * SKC;
    const unsigned soff = (unsigned)d0 * 4u;
    wc4 = bld4(rwc, 16u * (tid & 3), soff);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      ra4[i] = bld4(ra, voff[i], soff);
      rb4[i] = bld4(rb, voff[i], soff);
    }
  }
  template <int PART>
  __device__ inline void store_part(float* lds) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) store_part_piece<PART>(lds, i);
  }
  template <int PART>
  __device__ inline void store_part_piece(float* lds, int i) const {
    const int c = d0 + 4 * (tid & 3);
    {
      const int idx = slot(tid, i), rr = row0 + krow(idx);
      const uint32_t e0 = (uint32_t)rr * (uint32_t)g.D + (uint32_t)c;
      float dummy;
      float4 o;
      o.x = gate_dp_t<true, DROP, PART>(g, ra4[i].x, rb4[i].x, wc4.x, dsr[i], e0 + 0, thr, dscale, dummy);
      o.y = gate_dp_t<true, DROP, PART>(g, ra4[i].y, rb4[i].y, wc4.y, dsr[i], e0 + 1, thr, dscale, dummy);
      o.z = gate_dp_t<true, DROP, PART>(g, ra4[i].z, rb4[i].z, wc4.z, dsr[i], e0 + 2, thr, dscale, dummy);
      o.w = gate_dp_t<true, DROP, PART>(g, ra4[i].w, rb4[i].w, wc4.w, dsr[i], e0 + 3, thr, dscale, dummy);
      split_store4(lds + krow(idx) * SROW_F, idx & 3, o);
    }
  }
};

// Ungated stacks (MODE = 0 | dropout): one part, chunk kt = attention dims 16 kt .. 16 kt + 15; a plain split_mainloop
// loader (load / store_piece).
template <int ROWS, int NT, int MODE>
struct SplitP_U {
  static_assert(MODE == 0 || MODE == 1, "ungated, switches compiled in");
  static constexpr bool DROP = (MODE & 1) != 0;
  static constexpr int TOTAL = ROWS * 4, NV = (TOTAL + NT - 1) / NT, PIECES = NV;
  static_assert(TOTAL >= NT, "tile too small for this thread count");
  GateBwdCtx g;
  rsrc_t ra, rwc;
  int row0, tid, d0;
  uint32_t thr;
  float dscale;
  unsigned voff[NV];
  float dsr[NV];
  float4 ra4[NV], wc4;
  __device__ static inline int slot(int tid, int i) { const int idx = tid + i * NT; return idx < TOTAL ? idx : idx - NT; }
  __device__ inline void init_common(const GateBwdCtx& g_, int row0_, int nrows) {
    g = g_; g.resolve_seed(); row0 = row0_; tid = threadIdx.x;
    thr = drop_threshold(g.drop_p);
    dscale = g.drop_p > 0.f ? 1.0f / (1.0f - g.drop_p) : 1.0f;
    ra = make_rsrc(g.a, (unsigned)nrows * (unsigned)g.D * 4u);
    rwc = make_rsrc(g.Wc, (unsigned)g.D * 4u);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = slot(tid, i), rr = row0 + krow(idx);
      voff[i] = rr < nrows ? ((unsigned)rr * (unsigned)g.D + 4u * (idx & 3)) * 4u : OOB;
    }
  }
  __device__ inline void init(const GateBwdCtx& g_, int row0_, int nrows) {
    init_common(g_, row0_, nrows);
    rsrc_t rds = make_rsrc(g.ds, (unsigned)nrows * 4u);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int rr = row0 + krow(slot(tid, i));
      dsr[i] = bld1(rds, rr < nrows ? (unsigned)rr * 4u : OOB, 0);
    }
  }
  __device__ inline void init_lds(const GateBwdCtx& g_, int row0_, int nrows, const float* ds_lds) {
    init_common(g_, row0_, nrows);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int rl = krow(slot(tid, i));
      dsr[i] = row0 + rl < nrows ? ds_lds[rl] : 0.f;
    }
  }
  __device__ inline void load(int kt) {
    d0 = kt * SKC;
    const unsigned soff = (unsigned)d0 * 4u;
    wc4 = bld4(rwc, 16u * (tid & 3), soff);
#pragma unroll
    for (int i = 0; i < NV; ++i) ra4[i] = bld4(ra, voff[i], soff);
  }
  __device__ inline void store_piece(float* lds, int i) const {
    const int c = d0 + 4 * (tid & 3);
    const int idx = slot(tid, i), rr = row0 + krow(idx);
    const uint32_t e0 = (uint32_t)rr * (uint32_t)g.D + (uint32_t)c;
    float dummy;
    float4 o;
    o.x = gate_dp_t<false, DROP, 0>(g, ra4[i].x, 0.f, wc4.x, dsr[i], e0 + 0, thr, dscale, dummy);
    o.y = gate_dp_t<false, DROP, 0>(g, ra4[i].y, 0.f, wc4.y, dsr[i], e0 + 1, thr, dscale, dummy);
    o.z = gate_dp_t<false, DROP, 0>(g, ra4[i].z, 0.f, wc4.z, dsr[i], e0 + 2, thr, dscale, dummy);
    o.w = gate_dp_t<false, DROP, 0>(g, ra4[i].w, 0.f, wc4.w, dsr[i], e0 + 3, thr, dscale, dummy);
    split_store4(lds + krow(idx) * SROW_F, idx & 3, o);
  }
  __device__ inline void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) store_piece(lds, i);
  }
};

// B[k][n] = rows of Wa / Wb in SplitP_K's k order; the source is k-major, so a thread takes 8 k of one column (SplitM)
template <int ROWS, int NT, bool GATED>
struct SplitWab_M {
  static constexpr int TOTAL = ROWS * 2, NV = (TOTAL + NT - 1) / NT, PIECES = NV;
  static_assert(TOTAL % NT == 0 || TOTAL < NT, "whole vector slots only");
  rsrc_t ra, rb;
  int tid;
  unsigned hb;
  unsigned voff[NV];
  float r[NV][8];
  __device__ inline void init(const float* wa, const float* wb, int H, int D, int col0, bool) {
    // fewer (column, k half) items than threads (64-column tiles): the upper threads repeat the lower ones' work --
    // same addresses, same data, same LDS destinations, no branch in load() / store()
    tid = TOTAL < NT ? (int)threadIdx.x % TOTAL : (int)threadIdx.x; hb = (unsigned)H * 4u;
    ra = make_rsrc(wa, (unsigned)D * hb);
    rb = make_rsrc(GATED ? wb : wa, (unsigned)D * hb);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT, c = col0 + idx % ROWS, kh = idx / ROWS;
      voff[i] = c < H ? (unsigned)(8 * kh) * hb + (unsigned)c * 4u : OOB;
    }
  }
  __device__ inline void load(int kt) {
    const bool second = GATED && (kt & 1);
    const unsigned soff = (unsigned)((GATED ? (kt >> 1) : kt) * SKC) * hb;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) r[i][j] = bld1(second ? rb : ra, voff[i], soff + (unsigned)j * hb);
  }
  __device__ inline void store_piece(float* lds, int i) const {
    const int idx = tid + i * NT;
    split_store8(lds + (idx % ROWS) * SROW_F, idx / ROWS, r[i]);
  }
  __device__ inline void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) store_piece(lds, i);
  }
};

// main loop of the split K-dh: dh_mainloop_deep's pair ring (two A copies of one chunk pair each, four B copies) over
// compute_chunk_split.  nk % 4 == 0; branch-free (the chunk's part is the unrolled position's parity).
template <class T, class LA, class LB>
__device__ inline void dh_split_mainloop(const LA& la0, const LB& lb0, int nk, float* lds, f32x16 (&acc)[T::MB][T::NB]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
#pragma unroll
  for (int mb = 0; mb < T::MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  LA la[2] = {la0, la0};
  LB lb[4] = {lb0, lb0, lb0, lb0};
  la[0].load_pair(0); la[1].load_pair(2);
#pragma unroll
  for (int j = 0; j < 4; ++j) lb[j].load(j);
  la[0].template store_part<0>(lds);
  lb[0].store(lds + T::A_FLOATS);
  __syncthreads();
  constexpr int NS = split_steps<T>();
  for (int kt0 = 0; kt0 < nk; kt0 += 4) {
    static_for<4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int kt = kt0 + j;
      float* cur = lds + (j & 1) * T::STAGE_FLOATS;
      float* nxt = lds + ((j + 1) & 1) * T::STAGE_FLOATS;
      compute_chunk_split<T>(cur, cur + T::A_FLOATS, acc, wm, wn, lane, [&](int s) {
        if (s == 0) {
          // copy c's pair is fully written once its odd chunk has been stored, i.e. after position 2c of this round
          if constexpr (j == 1) la[0].load_pair(kt0 + 4);
          else if constexpr (j == 3) la[1].load_pair(kt0 + 6);
        }
        if (s == (NS >= 4 ? 1 : 0)) lb[j].load(kt + 4);
        // split + LDS writes of chunk kt+1 spread over the steps: A's vector slots first, B last
        constexpr int PA = LA::NV;
#pragma unroll
        for (int q = 0; q < PA; ++q)
          if (s == q * (NS - 1) / PA) la[((j + 1) & 3) >> 1].template store_part_piece<(j + 1) & 1>(nxt, q);
        if (s == NS - 1) lb[(j + 1) & 3].store(nxt + T::A_FLOATS);
      });
      __syncthreads();
    });
  }
}

template <class T, bool FUSED, int MODE = -1>
__global__ __launch_bounds__(T::NT) void bwd_dh_kernel(BwdDhParams p) {
  extern __shared__ __align__(16) float lds[];
  int mt, nt;
  if (!tile_of_block(blockIdx.x, p.mt_count, p.nt_count, mt, nt)) return;
  const int row0 = mt * T::BM, col0 = nt * T::BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* ds_l = lds + 2 * T::STAGE_FLOATS;     // [BM] ds, [BM] p, behind the staging buffers (FUSED only)
  float* p_l = ds_l + T::BM;
  MMF_KSTAMP(k0);
  constexpr bool SGATED = T::SPLIT && MODE >= 2;
  std::conditional_t<T::SPLIT,
                     std::conditional_t<SGATED, SplitP_K<T::BM, T::NT, (SGATED ? MODE : 2)>, SplitP_U<T::BM, T::NT, (T::SPLIT && !SGATED ? MODE : 0)>>,
                     LoadP_K<T::BM, T::NT, MODE>> la;
  if constexpr (FUSED) {
    // ---- K-prep for this tile's rows: p_i = softmax weight, ds_i = p_i (dM.h_i - dM.M) + gA_i ----------
    // g_i = dM.h_i: every wave takes a contiguous share of the rows; lanes cover float4 pieces of h with 8
    // independent loads in flight (the first version loaded 4 rows, reduced, repeated: latency-bound)
    constexpr int NW = T::NT / 64;
    constexpr int RPW = (T::BM + NW - 1) / NW;       // rows per wave (the last wave may own fewer)
    float* g_l = p_l + T::BM + 16;                   // [BM] behind ds, p and the reduction scratch
    const int LPR = p.H / 4;                         // float4 pieces per row: 64, 128 or 256
    const int PPL = LPR / 64;                        // pieces per lane and row: 1, 2 or 4
    float4 dm_l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) dm_l[q] = q < PPL ? ld4(p.dM + 4 * (lane + 64 * q)) : zero4();
    // everything that does not depend on g is requested up front, so its latency hides behind the h stream:
    // dM.M (dmm), the softmax statistics and this thread's row of A_raw / gA (BM <= NT for the fused tiles)
    float dmm = 0.f;
    for (int c = lane; c < p.H; c += 64) dmm += p.dM[c] * p.Mpool[c];
    const float smax = p.stats[0], inv = 1.0f / p.stats[1];
    static_assert(T::BM <= T::NT, "one row per thread in K-prep");
    const int prow = row0 + tid;
    const bool pok = tid < T::BM && prow < p.N;
    const float araw = pok ? p.A_raw[prow] : 0.f;
    const float gav = (pok && p.gA) ? p.gA[prow] : 0.f;
    const int wr0 = wave * RPW, wr1 = wr0 + RPW < T::BM ? wr0 + RPW : T::BM;
    auto g_rows = [&](auto nrows_c, int rb) {          // NR rows at a time: NR x PPL independent loads per lane
      constexpr int NR = decltype(nrows_c)::value;
      float part[NR];
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        const int row = row0 + rb + u;
        const int rc = row < p.N ? row : (int)p.N - 1;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < PPL) {
            const float4 hv = ld4(p.h + (size_t)rc * p.H + 4 * (lane + 64 * q));
            acc += hv.x * dm_l[q].x + hv.y * dm_l[q].y + hv.z * dm_l[q].z + hv.w * dm_l[q].w;
          }
        part[u] = acc;
      }
#pragma unroll
      for (int u = 0; u < NR; ++u) part[u] = wave_sum(part[u]);
      if (lane < NR && rb + lane < wr1) {
        float gv = part[0];
#pragma unroll
        for (int u = 1; u < NR; ++u) gv = lane == u ? part[u] : gv;
        g_l[rb + lane] = gv;
      }
    };
    if (PPL == 1) {       // H = 256: a row is one float4 per lane -- 16 rows in flight (the stream is latency-bound)
      for (int rb = wr0; rb < wr1; rb += 16) g_rows(std::integral_constant<int, 16>{}, rb);
    } else {
      for (int rb = wr0; rb < wr1; rb += 8) g_rows(std::integral_constant<int, 8>{}, rb);
    }
    dmm = wave_sum(dmm);
    __syncthreads();
    float dbc = 0.f;
    if (tid < T::BM) {
      float pi = 0.f, d = 0.f;
      if (pok) {
        pi = __expf(araw - smax) * inv;
        d = pi * (g_l[tid] - dmm) + gav;
        if (nt == 0) { p.p_out[prow] = pi; p.ds_out[prow] = d; }
      }
      ds_l[tid] = d;
      p_l[tid] = pi;
      dbc = d;
    }
    dbc = wave_sum(dbc);
    float* red = p_l + T::BM;
    if (lane == 0) red[wave] = dbc;
    __syncthreads();
    if (tid == 0 && nt == 0) {
      float s = 0.f;
      for (int w = 0; w < NW; ++w) s += red[w];
      p.dbc_part[mt] = s;
    }
    la.init_lds(p.g, row0, (int)p.N, ds_l);
  } else {
    la.init(p.g, row0, (int)p.N);
  }
  std::conditional_t<T::SPLIT, SplitWab_M<T::BN, T::NT, SGATED>, LoadWab_M<T::BN, T::NT>> lb;
  lb.init(p.Wa, p.Wb, p.H, p.g.D, col0, p.g.gated != 0);
  f32x16 acc[T::MB][T::NB];
  f32x4acc acch[T::NB][2];                   // the half block's accumulators (Tile::HALF; unused otherwise)
  const int nk = (p.g.gated ? 2 : 1) * p.g.D / (T::SPLIT ? SKC : KC);
  MMF_KSTAMP(k1);
  if constexpr (SGATED) {
    dh_split_mainloop<T>(la, lb, nk, lds, acc);
  } else if constexpr (T::SPLIT) {
    split_mainloop<T, 4>(la, lb, nk, lds, acc);
  } else if constexpr (T::NT == 256 && T::BM <= 64) {
    if (p.deep && p.g.gated) dh_mainloop_deep<T>(la, lb, nk, lds, acc);       // short grid: see dh_mainloop_deep
    else gemm_mainloop<T, decltype(la), decltype(lb), false>(la, lb, nk, lds, acc);
  } else {
    gemm_mainloop<T, decltype(la), decltype(lb), FUSED && MODE >= 0>(la, lb, nk, lds, acc, acch);
  }
  MMF_KSTAMP(k2);
  // ---- epilogue: du = (acc + p dM) relu'(h) scale_h, row-major.  The h values (and p) of block b+1 are requested
  // before block b is transposed and stored: with the reload inside the block (first version) every one of the
  // wave's blocks waited a full memory latency for its own h.  A load only waits for the stores issued BEFORE it
  // (in-order vmcnt), i.e. those of block b-1, which have had a whole block to drain.
  {
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int rr = lane >> 3, c4 = lane & 7;
    float* blk = lds + wave * (32 * EPI_STRIDE);
    constexpr int NBLK = T::MB * T::NB;
    float4 dm4[T::NB];                         // dM of this lane's columns, loaded once
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb) {
      const int col = col0 + epilogue_col<T>(nb);
      dm4[nb] = (p.dM && col < p.H) ? ld4(p.dM + col) : zero4();
    }
    if (p.relu_bits) {
      // relu'(u) . keep comes as one bit per element from the forward (LinearParams::relu_bits): 16 ballot words per
      // 32x32 block -- lane i < 16 loads word i (one 128-byte request per block), v_readlane moves a word into an
      // SGPR pair and that pair IS the lane mask of a v_cndmask.  h is not read again (51 MB per 50k bag, in the one
      // phase of the kernel where every CU sits on HBM at the same time).
      const int cbn = p.H >> 5;
      typedef unsigned long long u64;
      u64 mv[2];
      // bit blocks are 16 rows x 32 columns, 8 words each (LinearParams::relu_bits): a 32-row block is two of them
      auto fetch_bits = [&](int b, int s) {
        const int mb = b / T::NB, nb = b % T::NB;       // mb == T::MB: the half block (lanes 0..7 only)
        const int64_t rb16 = (((int64_t)row0 + wm * (T::BM / T::WM) + mb * 32) >> 4) + (lane >> 3);
        const int cb = (col0 >> 5) + wn * T::NB + nb;
        const bool ok = rb16 * 16 < p.N && cb < cbn && lane < (mb == T::MB ? 8 : 16);
        mv[s] = ok ? p.relu_bits[((size_t)rb16 * cbn + cb) * 8 + (lane & 7)] : 0ull;
      };
      fetch_bits(0, 0);
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        const int mb = b / T::NB, nb = b % T::NB, s = b & 1;
        if (b + 1 < NBLK) fetch_bits(b + 1, s ^ 1);
        float4 v[4];
        transpose_block(acc[mb][nb], blk, lane, v);
        const int r = wm * (T::BM / T::WM) + mb * 32 + rr, col = col0 + (wn * T::NB + nb) * 32 + 4 * c4;
        if (col >= p.H) continue;
        const float4 dm = dm4[nb];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = row0 + r + 8 * t;
          const float pi = FUSED ? p_l[r + 8 * t] : p.p[row < p.N ? row : (int)p.N - 1];
          const float raw[4] = {(v[t].x + pi * dm.x) * p.scale_h, (v[t].y + pi * dm.y) * p.scale_h,
                                (v[t].z + pi * dm.z) * p.scale_h, (v[t].w + pi * dm.w) * p.scale_h};
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)mv[s], 4 * t + e);
            const unsigned hi = __builtin_amdgcn_readlane((unsigned)(mv[s] >> 32), 4 * t + e);
            const u64 m = ((u64)hi << 32) | lo;
            asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(o[e]) : "v"(raw[e]), "s"(m));
          }
          if (row < p.N) st4(p.du + (size_t)row * p.H + col, make_float4(o[0], o[1], o[2], o[3]));
        }
      }
      if constexpr (T::HALF) {                  // the 16-row half block of every column block
#pragma unroll
        for (int nb = 0; nb < T::NB; ++nb) {
          fetch_bits(T::MB * T::NB + nb, 0);
          float4 v[2];
          transpose_half(acch[nb][0], acch[nb][1], blk, lane, v);
          const int r = wm * (T::BM / T::WM) + T::MB * 32 + rr, col = col0 + (wn * T::NB + nb) * 32 + 4 * c4;
          if (col >= p.H) continue;
          const float4 dm = dm4[nb];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int row = row0 + r + 8 * t;
            const float pi = FUSED ? p_l[r + 8 * t] : p.p[row < p.N ? row : (int)p.N - 1];
            const float raw[4] = {(v[t].x + pi * dm.x) * p.scale_h, (v[t].y + pi * dm.y) * p.scale_h,
                                  (v[t].z + pi * dm.z) * p.scale_h, (v[t].w + pi * dm.w) * p.scale_h};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned lo = __builtin_amdgcn_readlane((unsigned)mv[0], 4 * t + e);
              const unsigned hi = __builtin_amdgcn_readlane((unsigned)(mv[0] >> 32), 4 * t + e);
              const u64 m = ((u64)hi << 32) | lo;
              asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(o[e]) : "v"(raw[e]), "s"(m));
            }
            if (row < p.N) st4(p.du + (size_t)row * p.H + col, make_float4(o[0], o[1], o[2], o[3]));
          }
        }
      }
    } else {
    float4 hv[2][4];
    float pv[2][4];
    auto fetch = [&](int b, int s) {
      const int mb = b / T::NB, nb = b % T::NB;
      const int r = (wm * T::MB + mb) * 32 + rr, col = col0 + (wn * T::NB + nb) * 32 + 4 * c4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = row0 + r + 8 * t;
        const int rc = row < p.N ? row : (int)p.N - 1;
        // p.h == null: the standalone attention scorer (mmf_attn_net_backward) -- its input is not a ReLU output, so
        // there is no relu' mask and no pooling term: du = dP . Wab
        hv[s][t] = !p.h ? make_float4(1.f, 1.f, 1.f, 1.f) : (col < p.H ? ld4(p.h + (size_t)rc * p.H + col) : zero4());
        pv[s][t] = FUSED ? p_l[r + 8 * t] : (p.p ? p.p[rc] : 0.f);
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const int mb = b / T::NB, nb = b % T::NB, s = b & 1;
      if (b + 1 < NBLK) fetch(b + 1, s ^ 1);
      float4 v[4];
      transpose_block(acc[mb][nb], blk, lane, v);
      const int r = (wm * T::MB + mb) * 32 + rr, col = col0 + (wn * T::NB + nb) * 32 + 4 * c4;
      if (col >= p.H) continue;
      const float4 dm = dm4[nb];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = row0 + r + 8 * t;
        if (row >= p.N) continue;
        const float4 h4 = hv[s][t];
        const float pi = pv[s][t];
        float4 du;
        du.x = h4.x > 0.f ? (v[t].x + pi * dm.x) * p.scale_h : 0.f;
        du.y = h4.y > 0.f ? (v[t].y + pi * dm.y) * p.scale_h : 0.f;
        du.z = h4.z > 0.f ? (v[t].z + pi * dm.z) * p.scale_h : 0.f;
        du.w = h4.w > 0.f ? (v[t].w + pi * dm.w) * p.scale_h : 0.f;
        st4(p.du + (size_t)row * p.H + col, du);
      }
    }
    }
  }
#ifdef MMF_STAMPS
  MMF_KSTAMP(k3);
  if ((threadIdx.x & 63) == 0) {     // K-dh: prologue (K-prep), main loop, epilogue, waves
    atomicAdd(&g_stamps[4], k1 - k0); atomicAdd(&g_stamps[5], k2 - k1); atomicAdd(&g_stamps[6], k3 - k2); atomicAdd(&g_stamps[7], 1ull);
  }
#endif
}

// =============================================================================================
// K-nn : plain NN GEMM
// =============================================================================================
template <class T>
__global__ __launch_bounds__(T::NT) void gemm_nn_kernel(NnParams p) {
  extern __shared__ __align__(16) float lds[];
  int mt, nt;
  if (!tile_of_block(blockIdx.x, p.mt_count, p.nt_count, mt, nt)) return;
  const int row0 = mt * T::BM, col0 = nt * T::BN;
  LoadK<T::BM, T::NT> la;
  la.init(p.A, p.lda, row0, (int)p.M);
  LoadM<T::BN, T::NT> lb;
  lb.init(p.B, p.ldb, col0, p.N, 0, p.K);
  f32x16 acc[T::MB][T::NB];
  gemm_mainloop<T>(la, lb, p.K / KC, lds, acc);
  epilogue_rows<T>(acc, lds, [&](int mb, int nb, int r, int c, const float4 (&v)[4]) {
    const int col = col0 + c;
    if (col >= p.N) return;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = row0 + r + 8 * t;
      if (row < p.M) st4(p.C + (size_t)row * p.ldc + col, v[t]);
    }
  });
}

// =============================================================================================
// K-tn : grouped split-K TN GEMM
// =============================================================================================
// A-operand loaders accumulate per-thread column sums of what they stage (every thread owns 4
// fixed columns and KC*VPR/NT k-rows per chunk), which yields the bias gradients for free.
template <int ROWS, int NT>
struct LoadA_M_Plain {
  using Map = MMap<ROWS, NT>;
  rsrc_t rs;
  unsigned ldb, kbase_b;
  int tid;
  bool do_sum;
  unsigned voff[Map::NV];
  float4 r[Map::NV];
  float4 csum;
  __device__ inline void init(const float* s, int ld, int col0, int ncols, int kbase, int kmax, bool do_sum_) {
    tid = threadIdx.x; do_sum = do_sum_; csum = zero4();
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
  __device__ inline void store(float* lds) {
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      if (!Map::valid(tid, i)) continue;
      st4(lds + Map::lds(tid, i), r[i]);
      if (do_sum) { csum.x += r[i].x; csum.y += r[i].y; csum.z += r[i].z; csum.w += r[i].w; }   // out-of-range reads are 0
    }
  }
  static constexpr bool kHalves = true;
  static constexpr int HV = Map::NV / 2;
  __device__ inline void load_half(int kt, int h) {
    const unsigned soff = kbase_b + (unsigned)(kt * KC) * ldb;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i)
      if ((i < HV) == (h == 0)) r[i] = bld4(rs, voff[i], soff);
  }
  __device__ inline void store_half(float* lds, int h) {
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      if ((i < HV) != (h == 0) || !Map::valid(tid, i)) continue;
      st4(lds + Map::lds(tid, i), r[i]);
      if (do_sum) { csum.x += r[i].x; csum.y += r[i].y; csum.z += r[i].z; csum.w += r[i].w; }
    }
  }
};

// A[k = instance][m] = dP.  One tile covers DT attention dims d0 .. d0+DT-1:
//   gated  : DT = ROWS/2; image columns [0, DT) = d pre-tanh, [DT, ROWS) = d pre-sigmoid of the SAME dims, so a
//            thread loads a, b (and ds) once and emits both halves -- half the loads and loader registers of a
//            layout that puts the two halves in different tiles;
//   ungated: DT = ROWS, columns = d pre-tanh.
// DROP: 0 / 1 = attention dropout off / on, compiled in (the large-bag tile: the kernel is instantiated per value);
//       -1 = tested per element at run time (small tile)
template <int ROWS, int NT, bool GATED, int DROP = -1>
struct LoadA_M_Gate {
  static constexpr int DT = GATED ? ROWS / 2 : ROWS;
  using Map = MMap<DT, NT>;
  static_assert(NT % Map::VPR == 0 && Map::EXACT, "a thread must own the same columns in every vector slot");
  GateBwdCtx g;
  rsrc_t ra, rb, rds;
  int d0, kbase, tid, kt_loaded;
  bool do_sum;
  uint32_t thr;
  float dscale;
  unsigned db;
  unsigned voff[Map::NV], voff_ds[Map::NV];
  float4 ra4[Map::NV], rb4[Map::NV], wc4;
  float dsr[Map::NV];
  float4 csum_a, csum_b, csum2;   // column sums: d pre-tanh, d pre-sigmoid (bias grads), ds.a_d.b_d (dWc)
  __device__ inline void init(const GateBwdCtx& g_, int d0_, int kbase_, int kmax, bool do_sum_) {
    g = g_; g.resolve_seed(); d0 = d0_; kbase = kbase_; tid = threadIdx.x; do_sum = do_sum_;
    thr = drop_threshold(g.drop_p);
    dscale = g.drop_p > 0.f ? 1.0f / (1.0f - g.drop_p) : 1.0f;
    csum_a = zero4(); csum_b = zero4(); csum2 = zero4();
    db = (unsigned)g.D * 4u;
    const unsigned rows = (unsigned)(kmax > 0 ? kmax : 0);
    ra = make_rsrc(g.a, rows * db);
    rb = make_rsrc(GATED ? g.b : g.a, rows * db);
    rds = make_rsrc(g.ds, rows * 4u);
    const int c = d0 + 4 * Map::c4(tid, 0);
    wc4 = bld4(make_rsrc(g.Wc, db), c < g.D ? (unsigned)c * 4u : OOB, 0);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      const bool ok = c < g.D;
      voff[i] = ok ? (unsigned)Map::krow(tid, i) * db + (unsigned)c * 4u : OOB;
      voff_ds[i] = ok ? (unsigned)Map::krow(tid, i) * 4u : OOB;
    }
    kt_loaded = 0;
  }
  __device__ inline void load(int kt) {
    kt_loaded = kt;
    const unsigned k0 = (unsigned)(kbase + kt * KC);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      ra4[i] = bld4(ra, voff[i], k0 * db);
      if (GATED) rb4[i] = bld4(rb, voff[i], k0 * db);
      dsr[i] = bld1(rds, voff_ds[i], k0 * 4u);    // 0 beyond the split's last instance => dP = 0 there
    }
  }
  __device__ inline void store(float* lds) {
    const int c = d0 + 4 * Map::c4(tid, 0);
    GateBwdCtx gg = g;
    gg.gated = GATED ? 1 : 0;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      const int k = kbase + kt_loaded * KC + Map::krow(tid, i);
      const uint32_t idx = (uint32_t)k * (uint32_t)g.D + (uint32_t)c;
      const float dsv = dsr[i];
      const float av[4] = {ra4[i].x, ra4[i].y, ra4[i].z, ra4[i].w};
      const float bv[4] = {rb4[i].x, rb4[i].y, rb4[i].z, rb4[i].w};
      const float wc[4] = {wc4.x, wc4.y, wc4.z, wc4.w};
      float oa[4], ob[4], w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (DROP >= 0) {
          oa[e] = gate_dp_t<GATED, DROP != 0, 0>(gg, av[e], GATED ? bv[e] : 0.f, wc[e], dsv, idx + e, thr, dscale, w[e]);
          ob[e] = GATED ? gate_dp_t<GATED, DROP != 0, 1>(gg, av[e], bv[e], wc[e], dsv, idx + e, thr, dscale, w[e]) : 0.f;
        } else {
          oa[e] = gate_dp(gg, 0, av[e], GATED ? bv[e] : 0.f, wc[e], dsv, idx + e, thr, dscale, w[e]);
          ob[e] = GATED ? gate_dp(gg, 1, av[e], bv[e], wc[e], dsv, idx + e, thr, dscale, w[e]) : 0.f;
        }
      }
      float* dst = lds + Map::krow(tid, i) * ROWS + 4 * Map::c4(tid, i);
      st4(dst, make_float4(oa[0], oa[1], oa[2], oa[3]));
      if (GATED) st4(dst + DT, make_float4(ob[0], ob[1], ob[2], ob[3]));
      if (do_sum) {
        csum_a.x += oa[0]; csum_a.y += oa[1]; csum_a.z += oa[2]; csum_a.w += oa[3];
        csum_b.x += ob[0]; csum_b.y += ob[1]; csum_b.z += ob[2]; csum_b.w += ob[3];
        csum2.x += dsv * w[0]; csum2.y += dsv * w[1]; csum2.z += dsv * w[2]; csum2.w += dsv * w[3];
      }
    }
  }
};

// ---- A operands of the split-operand TN kernel (mmf_gemm_split.h): chunks of 16 instances ---------------------------
// Plain A[k][m]: SplitM's map (a thread owns ONE column and 8 consecutive instances) + the column sums.
template <int ROWS, int NT>
struct SplitA_M_Plain {
  static constexpr int TOTAL = ROWS * 2;
  static_assert(TOTAL == NT, "one (column, k half) per thread");
  rsrc_t rs;
  unsigned ldb, kbase_b, voff;
  int tid;
  bool do_sum;
  float r[8];
  float csum;
  __device__ inline void init(const float* s, int ld, int col0, int ncols, int kbase, int kmax, bool do_sum_) {
    tid = threadIdx.x; do_sum = do_sum_; csum = 0.f;
    rs = make_rsrc(s, (unsigned)(kmax > 0 ? kmax : 0) * (unsigned)ld * 4u);
    ldb = (unsigned)ld * 4u;
    kbase_b = (unsigned)kbase * ldb;
    const int c = col0 + tid % ROWS, kh = tid / ROWS;
    voff = c < ncols ? (unsigned)(8 * kh) * ldb + (unsigned)c * 4u : OOB;
  }
  __device__ inline void load(int kt) {
#ifdef MMF_SDIAG_NOGLOAD
    if (kt >= 4) return;
#endif
    const unsigned soff = kbase_b + (unsigned)(kt * SKC) * ldb;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = bld1(rs, voff, soff + (unsigned)j * ldb);
  }
  __device__ inline void store(float* lds) {
    split_store8(lds + (tid % ROWS) * SROW_F, tid / ROWS, r);
    if (do_sum) csum += ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));   // out-of-range reads are 0
  }
  static constexpr int PIECES = 1;
  __device__ inline void store_piece(float* lds, int) { store(lds); }
  __device__ inline void absorb(const SplitA_M_Plain& o) { csum += o.csum; }     // the copies start from a zero sum
};

// A[k = instance][m] = dP for DT = ROWS / 2 attention dims d0 .. d0 + DT - 1 (gated: image rows [0, DT) = d pre-tanh,
// [DT, ROWS) = d pre-sigmoid of the same dims; ungated: DT = ROWS).  A thread owns one dim and 16 DT / NT consecutive
// instances: it loads a, b, ds once and emits both halves.
template <int ROWS, int NT, bool GATED, int DROP>
struct SplitA_M_Gate {
  static constexpr int DT = GATED ? ROWS / 2 : ROWS;
  static constexpr int KPT = SKC * DT / NT;            // instances per thread: 4 (gated 256-row tile, 512 threads) or 8
  static_assert(KPT == 4 || KPT == 8, "plane pieces of 8 or 16 bytes");
  GateBwdCtx g;
  rsrc_t ra, rb, rds;
  int d0, kbase, tid, kt_loaded;
  bool do_sum;
  uint32_t thr;
  float dscale, wc;
  unsigned db, voff, voff_ds;
  float ra1[KPT], rb1[KPT], dsr[KPT];
  float csum_a, csum_b, csum2;   // column sums: d pre-tanh, d pre-sigmoid (bias grads), ds.a_d.b_d (dWc)
  __device__ inline void init(const GateBwdCtx& g_, int d0_, int kbase_, int kmax, bool do_sum_) {
    g = g_; g.resolve_seed(); d0 = d0_; kbase = kbase_; tid = threadIdx.x; do_sum = do_sum_;
    thr = drop_threshold(g.drop_p);
    dscale = g.drop_p > 0.f ? 1.0f / (1.0f - g.drop_p) : 1.0f;
    csum_a = csum_b = csum2 = 0.f;
    db = (unsigned)g.D * 4u;
    const unsigned rows = (unsigned)(kmax > 0 ? kmax : 0);
    ra = make_rsrc(g.a, rows * db);
    rb = make_rsrc(GATED ? g.b : g.a, rows * db);
    rds = make_rsrc(g.ds, rows * 4u);
    const int c = d0 + tid % DT, kq = tid / DT;
    const bool ok = c < g.D;
    wc = bld1(make_rsrc(g.Wc, db), ok ? (unsigned)c * 4u : OOB, 0);
    voff = ok ? (unsigned)(KPT * kq) * db + (unsigned)c * 4u : OOB;
    voff_ds = ok ? (unsigned)(KPT * kq) * 4u : OOB;
    kt_loaded = 0;
  }
  __device__ inline void load(int kt) {
    kt_loaded = kt;
    const unsigned k0 = (unsigned)(kbase + kt * SKC);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      ra1[j] = bld1(ra, voff, (k0 + j) * db);
      if (GATED) rb1[j] = bld1(rb, voff, (k0 + j) * db);
      dsr[j] = bld1(rds, voff_ds, (k0 + j) * 4u);    // 0 beyond the split's last instance => dP = 0 there
    }
  }
  __device__ inline void store(float* lds) {
    const int dl = tid % DT, kq = tid / DT, c = d0 + dl;
    GateBwdCtx gg = g;
    gg.gated = GATED ? 1 : 0;
    float oa[KPT], ob[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int k = kbase + kt_loaded * SKC + KPT * kq + j;
      const uint32_t idx = (uint32_t)k * (uint32_t)g.D + (uint32_t)c;
      float w;
      oa[j] = gate_dp_t<GATED, DROP != 0, 0>(gg, ra1[j], GATED ? rb1[j] : 0.f, wc, dsr[j], idx, thr, dscale, w);
      ob[j] = GATED ? gate_dp_t<GATED, DROP != 0, 1>(gg, ra1[j], rb1[j], wc, dsr[j], idx, thr, dscale, w) : 0.f;
      if (do_sum) { csum_a += oa[j]; csum_b += ob[j]; csum2 += dsr[j] * w; }
    }
    if constexpr (KPT == 4) {
      split_store4(lds + dl * SROW_F, kq, make_float4(oa[0], oa[1], oa[2], oa[3]));
      if (GATED) split_store4(lds + (DT + dl) * SROW_F, kq, make_float4(ob[0], ob[1], ob[2], ob[3]));
    } else {
      split_store8(lds + dl * SROW_F, kq, oa);
      if (GATED) split_store8(lds + (DT + dl) * SROW_F, kq, ob);
    }
  }
  static constexpr int PIECES = 1;
  __device__ inline void store_piece(float* lds, int) { store(lds); }
  __device__ inline void absorb(const SplitA_M_Gate& o) { csum_a += o.csum_a; csum_b += o.csum_b; csum2 += o.csum2; }
};

// per-thread scalar column sums: the NT / COLS threads that own column c are tid = c + q COLS
template <int COLS, int NT>
__device__ inline void colsum1_reduce_store(float* lds, float v, float* dst, int col0, int ncols) {
  const int tid = threadIdx.x;
  __syncthreads();
  lds[tid] = v;
  __syncthreads();
  if (tid < COLS) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NT / COLS; ++q) s += lds[q * COLS + tid];
    if (col0 + tid < ncols) dst[col0 + tid] = s;
  }
}

// reduce a per-thread float4 column sum over the NT/(COLS/4) threads that own the same 4 of COLS columns
template <int COLS, int NT>
__device__ inline void colsum_reduce_store(float* lds, float4 v, float* dst, int col0, int ncols) {
  constexpr int VPR = COLS / 4, GROUPS = NT / VPR;
  const int tid = threadIdx.x;
  __syncthreads();
  st4(lds + (tid / VPR) * COLS + 4 * (tid % VPR), v);
  __syncthreads();
  if (tid < COLS) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < GROUPS; ++q) s += lds[q * COLS + tid];
    if (col0 + tid < ncols) dst[col0 + tid] = s;
  }
}

// rowmap(tile_row) -> output row, or -1 to drop it
template <class T, class RowMap>
__device__ inline void tn_store(const TnProblem& q, int split, int tn, f32x16 (&acc)[T::MB][T::NB], float* lds,
                                RowMap&& rowmap) {
  float* out = q.out + (size_t)split * q.split_stride;
  if constexpr (T::PERM) {
    // permuted fragment layout (mmf_gemm_core.h): block (mb, nb) of a wave holds rows 4 i + mb and columns 2 j + nb
    // of its 128 x 64 patch.  Transposing both column blocks of a row block gives every lane 8 CONSECUTIVE columns
    // (2 (4 c4 + e) + nb, e = 0..3) of rows 4 (rr + 8 t) + mb: two float4 stores, 256 contiguous bytes per row.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int rr = lane >> 3, c4 = lane & 7;
    float* blk = lds + wave * (32 * EPI_STRIDE);
    const int col = tn * T::BN + wn * 64 + 8 * c4;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      float4 v0[4], v1[4];
      transpose_block(acc[mb][0], blk, lane, v0);
      transpose_block(acc[mb][1], blk, lane, v1);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = rowmap(wm * 128 + 4 * (rr + 8 * t) + mb);
        if (row < 0) continue;
        float* o = out + (size_t)row * q.ldc + col;
        if (col < q.Ncols) st4(o, make_float4(v0[t].x, v1[t].x, v0[t].y, v1[t].y));
        if (col + 4 < q.Ncols) st4(o + 4, make_float4(v0[t].z, v1[t].z, v0[t].w, v1[t].w));
      }
    }
    return;
  }
  epilogue_rows<T>(acc, lds, [&](int mb, int nb, int r, int c, const float4 (&v)[4]) {
    const int col = tn * T::BN + c;
    if (col >= q.Ncols) return;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = rowmap(r + 8 * t);
      if (row >= 0) st4(out + (size_t)row * q.ldc + col, v[t]);
    }
  });
}

template <class T, bool GATED, int DROP>
__device__ inline void tn_gate_tile(const TnParams& p, const TnProblem& q,
                                    std::conditional_t<T::SPLIT, SplitM<T::BN, T::NT>, LoadM<T::BN, T::NT>>& lb, int split, int tm,
                                    int tn, int kbase, int kmax, int nk, int last_groups, bool do_sum, float* lds) {
  using LA = std::conditional_t<T::SPLIT, SplitA_M_Gate<T::BM, T::NT, GATED, (DROP > 0 ? 1 : 0)>, LoadA_M_Gate<T::BM, T::NT, GATED, DROP>>;
  constexpr int DT = LA::DT;
  const int D = p.g.D, d0 = tm * DT;
  LA la;
  la.init(p.g, d0, kbase, kmax, do_sum);
  f32x16 acc[T::MB][T::NB];
  if constexpr (T::SPLIT) split_mainloop<T, 2>(la, lb, nk, lds, acc);
  else gemm_mainloop<T>(la, lb, nk, lds, acc, nullptr, last_groups);
  tn_store<T>(q, split, tn, acc, lds, [&](int r) {      // tile row -> row of the stacked [dWa ; dWb] slab
    const int half = r / DT, d = d0 + r - half * DT;
    return d < D ? half * D + d : -1;
  });
  if constexpr (T::SPLIT) {
    if (do_sum) {
      float* cs = q.colsum + (size_t)split * q.colsum_stride;
      colsum1_reduce_store<DT, T::NT>(lds, la.csum_a, cs, d0, D);
      if (GATED) colsum1_reduce_store<DT, T::NT>(lds, la.csum_b, cs + D, d0, D);
      if (q.colsum2) colsum1_reduce_store<DT, T::NT>(lds, la.csum2, q.colsum2 + (size_t)split * q.colsum2_stride, d0, D);
    }
  } else
  if (do_sum) {
    float* cs = q.colsum + (size_t)split * q.colsum_stride;
    colsum_reduce_store<DT, T::NT>(lds, la.csum_a, cs, d0, D);
    if (GATED) colsum_reduce_store<DT, T::NT>(lds, la.csum_b, cs + D, d0, D);
    if (q.colsum2) colsum_reduce_store<DT, T::NT>(lds, la.csum2, q.colsum2 + (size_t)split * q.colsum2_stride, d0, D);
  }
}

template <class T, int DROP = -1>
__global__ __launch_bounds__(T::NT) void tn_kernel(TnParams p) {
  extern __shared__ __align__(16) float lds[];
  const int b = blockIdx.x;
  int split, pi = 0, t;
  if (p.xcd_map == 2) {  // host-built table: the tiles that share an operand panel sit on one XCD
    const unsigned e = p.map[b];
    if (e == 0xFFFFu) return;
    split = (int)(e >> 5);
    const int tg = (int)(e & 31u);
    for (int i = 1; i < p.nprob; ++i)
      if (tg >= p.prob[i].tile_begin) pi = i;
    t = tg - p.prob[pi].tile_begin;
  } else {               // plain order: problem by problem, split by split
    for (int i = 1; i < p.nprob; ++i)
      if (b >= p.prob[i].block_begin) pi = i;
    const int rel = b - p.prob[pi].block_begin, nt = p.prob[pi].tiles_m * p.prob[pi].tiles_n;
    split = rel / nt;
    t = rel - split * nt;
  }
  const TnProblem& q = p.prob[pi];
  if (split >= q.splits) return;
  const int tm = t / q.tiles_n, tn = t - tm * q.tiles_n;
  const int64_t kb64 = (int64_t)split * q.k_per_split;
  const int kbase = (int)(kb64 < p.K ? kb64 : p.K);
  const int kmax = (int)((kb64 + q.k_per_split) < p.K ? (kb64 + q.k_per_split) : p.K);
  constexpr int CH = T::SPLIT ? SKC : KC;        // instances per staged chunk
  // split-operand tiles: an even number of chunks (split_mainloop<T, 2>); the padding chunk reads zeros past kmax
  const int nk = T::SPLIT ? 2 * ((kmax - kbase + 2 * CH - 1) / (2 * CH)) : (kmax - kbase + CH - 1) / CH;
  // fragment groups (2 G instances each) of the last chunk that hold data: splits are cut at multiples of 4 instances,
  // not of whole chunks, so that all of them have the same length (50k bag: 42 x 1192 = 37.25 chunks each, where whole
  // chunks gave 41 x 38 + one split of 4.5)
  int last_groups = 0;
  if constexpr (!T::SPLIT) last_groups = nk > 0 ? ((kmax - kbase) - (nk - 1) * KC + 2 * T::G - 1) / (2 * T::G) : 0;
  const bool do_sum = tn == 0 && q.colsum != nullptr;
#ifdef MMF_STAMPS             /* workgroup life time by tile kind: [0] sum plain, [1] sum gate, [2] count plain, [3] count gate */
  struct TnLife {
    unsigned long long t0; int kind;
    __device__ ~TnLife() {
      const unsigned long long t1 = stamp_now();
      if (threadIdx.x == 0) { atomicAdd(&g_stamps[kind], t1 - t0); atomicAdd(&g_stamps[2 + kind], 1ull); }
    }
  } life{stamp_now(), q.kind == TN_A_PLAIN ? 0 : 1};
#endif

  std::conditional_t<T::SPLIT, SplitM<T::BN, T::NT>, LoadM<T::BN, T::NT>> lb;
  lb.init(q.B, q.ldb, tn * T::BN, q.Ncols, kbase, kmax);
  if (q.kind == TN_A_PLAIN) {
    std::conditional_t<T::SPLIT, SplitA_M_Plain<T::BM, T::NT>, LoadA_M_Plain<T::BM, T::NT>> la;
    la.init(q.A, q.lda, tm * T::BM, q.M, kbase, kmax, do_sum);
    f32x16 acc[T::MB][T::NB];
    if constexpr (T::SPLIT) split_mainloop<T, 2, decltype(la), decltype(lb), true>(la, lb, nk, lds, acc);
    else gemm_mainloop<T>(la, lb, nk, lds, acc, nullptr, last_groups);
    tn_store<T>(q, split, tn, acc, lds, [&](int r) { const int row = tm * T::BM + r; return row < q.M ? row : -1; });
    if (do_sum) {
      if constexpr (T::SPLIT) colsum1_reduce_store<T::BM, T::NT>(lds, la.csum, q.colsum + (size_t)split * q.colsum_stride, tm * T::BM, q.M);
      else colsum_reduce_store<T::BM, T::NT>(lds, la.csum, q.colsum + (size_t)split * q.colsum_stride, tm * T::BM, q.M);
    }
  } else if (p.g.gated) {
    tn_gate_tile<T, true, DROP>(p, q, lb, split, tm, tn, kbase, kmax, nk, last_groups, do_sum, lds);
  } else {
    tn_gate_tile<T, false, DROP>(p, q, lb, split, tm, tn, kbase, kmax, nk, last_groups, do_sum, lds);
  }
}

// =============================================================================================
// K-red : out[j] = sum_s in[s*stride + j]  for a short list of segments (deterministic order)
// =============================================================================================
__global__ __launch_bounds__(256) void reduce_kernel(ReduceParams p) {
  const int b = blockIdx.x;
  int si = 0;
  for (int i = 1; i < p.nseg; ++i)
    if (b >= p.seg[i].block_begin) si = i;
  const ReduceSeg& s = p.seg[si];
  if (s.len < 4) {   // tiny segment (dbc): all 256 threads stride over the splits, fixed-order block reduce
    __shared__ float red[4];
    for (int e = 0; e < s.len; ++e) {
      float acc = 0.f;
      for (int k = threadIdx.x; k < s.nsplit; k += 256) acc += s.in[(size_t)k * s.stride + e];
      acc = wave_sum(acc);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
      __syncthreads();
      if (threadIdx.x == 0) s.out[e] = (p.accumulate ? s.out[e] : 0.f) + (red[0] + red[1] + red[2] + red[3]);
      __syncthreads();
    }
    return;
  }
  if (s.tall) {      // many splits, short rows (per-row-tile partials): 16 column quads x 16 split groups per block
    __shared__ __align__(16) float part[16][64];
    const int cq = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int j = (b - s.block_begin) * 64 + 4 * cq;
    float4 acc = zero4();
    if (j < s.len) {
      const float* in = s.in + j;
      int k = grp;
      for (; k + 7 * 16 < s.nsplit; k += 8 * 16) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld4(in + (size_t)(k + 16 * u) * s.stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      for (; k < s.nsplit; k += 16) {
        float4 v = ld4(in + (size_t)k * s.stride);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    st4(&part[grp][4 * cq], acc);
    __syncthreads();
    if (threadIdx.x < 64) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += part[g][threadIdx.x];
      const int c = (b - s.block_begin) * 64 + threadIdx.x;
      if (c < s.len) s.out[c] = (p.accumulate ? s.out[c] : 0.f) + t;
    }
    return;
  }
  const int j = ((b - s.block_begin) * 256 + threadIdx.x) * 4;
  if (j >= s.len) return;
  if ((s.len & 3) == 0 && (s.stride & 3) == 0) {
    // 8 independent loads in flight per thread: the serial version was latency-bound (73 us for 34 MB)
    float4 acc = zero4();
    const float* in = s.in + j;
    int k = 0;
    for (; k + 8 <= s.nsplit; k += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld4(in + (size_t)(k + u) * s.stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; k < s.nsplit; ++k) {
      float4 v = ld4(in + (size_t)k * s.stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (p.accumulate) { const float4 o = ld4(s.out + j); acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    st4(s.out + j, acc);
  } else {
    for (int e = j; e < j + 4 && e < s.len; ++e) {
      float acc = 0.f;
      for (int k = 0; k < s.nsplit; ++k) acc += s.in[(size_t)k * s.stride + e];
      s.out[e] = (p.accumulate ? s.out[e] : 0.f) + acc;
    }
  }
}

// =============================================================================================
// host launchers
// =============================================================================================
template <class T, class P>
static int launch_tiled(const char* name, void (*kern)(P), const P& p, int grid, hipStream_t st) {
  if (int e = set_dyn_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES)) return e;
  ProfScope ps(name, st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(T::NT), T::LDS_BYTES, st, p);
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

int launch_bwd_prep(BwdPrepParams p, hipStream_t st) {
  if (p.H % 4 != 0 || p.H > 1024) return MMF_ERR_SHAPE;
  { ProfScope ps("bwd_prep_kernel", st); hipLaunchKernelGGL(bwd_prep_kernel, dim3(p.n_groups), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

// launch with extra dynamic LDS behind the staging buffers (fused prep scratch)
template <class T, class P>
static int launch_tiled_extra(const char* name, void (*kern)(P), const P& p, int grid, int extra_bytes, hipStream_t st) {
  const int bytes = T::LDS_BYTES + extra_bytes;
  if (int e = set_dyn_lds(reinterpret_cast<const void*>(kern), bytes)) return e;
  ProfScope ps(name, st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(T::NT), bytes, st, p);
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

template <int ROWS>
static int launch_bwd_dh_wide(BwdDhParams p, hipStream_t st) {
  using T = Tile<ROWS, 256, 1, 8, true, false>;
  p.mt_count = (int)((p.N + T::BM - 1) / T::BM); p.nt_count = p.H / 256;
  const int grid = grid_for_tiles(p.mt_count, p.nt_count);
  if (p.fused_prep) {
    constexpr int extra = (3 * T::BM + 16) * 4;
    switch ((p.g.gated ? 2 : 0) + (p.g.drop_p > 0.f ? 1 : 0)) {      // gate / dropout switches compiled in
      case 0: return launch_tiled_extra<T>("bwd_dh_kernel", bwd_dh_kernel<T, true, 0>, p, grid, extra, st);
      case 1: return launch_tiled_extra<T>("bwd_dh_kernel", bwd_dh_kernel<T, true, 1>, p, grid, extra, st);
      case 2: return launch_tiled_extra<T>("bwd_dh_kernel", bwd_dh_kernel<T, true, 2>, p, grid, extra, st);
      // gated + attention dropout: two hashes per element inlined 16 times overflow the 256 VGPRs (48 spilled
      // inside the loop) -- that combination keeps the run-time switches
      default: return launch_tiled_extra<T>("bwd_dh_kernel", bwd_dh_kernel<T, true>, p, grid, extra, st);
    }
  }
  return launch_tiled<T>("bwd_dh_kernel", bwd_dh_kernel<T, false>, p, grid, st);
}

// > 0: the wide path will be taken and the kernel does K-prep itself, writing that many dbc partials
static inline bool dh_short_grid(int64_t N, int H, int split = 0) {       // the 64x64 tiles on a grid of at most 512 workgroups
  static const int cap = tune_int("MMF_DH_SHORT_MAX", 1024);   // tuning override (round 4, 10k-14k bags: K-prep inside the 64 x 64 tiles, +5.4 us, against its own launch, 9.7 us)
  return !use_wide_tiles(N, H, split) && (N / 128) * ((H + 127) / 128) < 256 && ((N + 63) / 64) * ((H + 63) / 64) <= cap;
}
// the split-operand mode's 64 x 64 K-dh tiles: every bag below its wide tiles
static inline bool dh_split_small_ok(int64_t N, int H, int D, int gated, int split) {
  return split && !use_wide_tiles(N, H, 1) && N >= split_min_rows() && ((gated ? 2 : 1) * D / SKC) % 4 == 0;
}
// the split-operand K-dh: the training step's shape only (fused K-prep, wide tiles)
bool bwd_dh_split_ok(int64_t N, int H, int D, int gated, int split) {
  return split && use_wide_tiles(N, H, 1) && ((gated ? 2 : 1) * D / SKC) % 4 == 0;
}
int bwd_dh_split_rows(int64_t N) { (void)N; return 224; }

// K-dh's tallest tile: the 240-row tile of the training variant (gated, relu bits) spills 28 registers (112 B per lane)
constexpr int DH_MAX_ROWS = 224;
int bwd_dh_fused_groups(int64_t N, int H, int allow_half, int D, int gated, int split, int concurrent) {
  static const int env = tune_int("MMF_FUSED_PREP", 1);
  if (!env) return 0;
  // short grids: every column tile redoes K-prep for its 64 rows (64 KB of h) -- cheaper than a launch of its own
  const int sm = dh_split_small_ok(N, H, D, gated, split);
  if (dh_short_grid(N, H, sm)) return (int)((N + 63) / 64);
  if (sm || !use_wide_tiles(N, H, split)) return 0;
  if (bwd_dh_split_ok(N, H, D, gated, split)) return (int)((N + bwd_dh_split_rows(N) - 1) / bwd_dh_split_rows(N));
  const int rows = pick_wide_rows(N, H / 256, allow_half != 0, concurrent != 0, DH_MAX_ROWS);   // the fused launch always has the forward's relu bits
  return (int)((N + rows - 1) / rows);
}

template <int ROWS>
static int launch_bwd_dh_split(BwdDhParams p, hipStream_t st) {
  using T = TileSp<ROWS, 256, 1, 8>;
  p.mt_count = (int)((p.N + T::BM - 1) / T::BM); p.nt_count = p.H / 256;
  const int grid = grid_for_tiles(p.mt_count, p.nt_count);
  constexpr int extra = (3 * T::BM + 16) * 4;
  switch ((p.g.gated ? 2 : 0) + (p.g.drop_p > 0.f ? 1 : 0)) {
    case 0: return launch_tiled_extra<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, true, 0>, p, grid, extra, st);
    case 1: return launch_tiled_extra<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, true, 1>, p, grid, extra, st);
    case 2: return launch_tiled_extra<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, true, 2>, p, grid, extra, st);
    default: return launch_tiled_extra<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, true, 3>, p, grid, extra, st);
  }
}

// 64 x 64 split tiles for bags below the wide tiles (with or without the fused K-prep)
static int launch_bwd_dh_split_small(BwdDhParams p, hipStream_t st) {
  using T = TileSp<64, 64, 2, 2>;
  p.mt_count = (int)((p.N + 63) / 64); p.nt_count = (p.H + 63) / 64;
  const int grid = grid_for_tiles(p.mt_count, p.nt_count);
  constexpr int extra = (3 * T::BM + 16) * 4;
  const int mode = (p.g.gated ? 2 : 0) + (p.g.drop_p > 0.f ? 1 : 0);
  if (p.fused_prep) {
    switch (mode) {
      case 0: return launch_tiled_extra<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, true, 0>, p, grid, extra, st);
      case 1: return launch_tiled_extra<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, true, 1>, p, grid, extra, st);
      case 2: return launch_tiled_extra<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, true, 2>, p, grid, extra, st);
      default: return launch_tiled_extra<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, true, 3>, p, grid, extra, st);
    }
  }
  switch (mode) {
    case 0: return launch_tiled<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, false, 0>, p, grid, st);
    case 1: return launch_tiled<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, false, 1>, p, grid, st);
    case 2: return launch_tiled<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, false, 2>, p, grid, st);
    default: return launch_tiled<T>("bwd_dh_split_kernel", bwd_dh_kernel<T, false, 3>, p, grid, st);
  }
}

int launch_bwd_dh(BwdDhParams p, hipStream_t st) {
  if (p.g.D % KC != 0 || p.H % 4 != 0) return MMF_ERR_SHAPE;
  if (p.N <= 0) return MMF_OK;
  if (p.fused_prep && bwd_dh_split_ok(p.N, p.H, p.g.D, p.g.gated, p.split)) return launch_bwd_dh_split<224>(p, st);
  if (dh_split_small_ok(p.N, p.H, p.g.D, p.g.gated, p.split) && (!p.fused_prep || dh_short_grid(p.N, p.H, 1)))
    return launch_bwd_dh_split_small(p, st);
  if (use_wide_tiles(p.N, p.H, p.split)) {
    // the half-block tile's epilogue exists for the relu-bits path only (every stack backward; not the standalone scorer)
    switch (pick_wide_rows(p.N, p.H / 256, p.allow_half && p.fused_prep && p.relu_bits, p.concurrent != 0, DH_MAX_ROWS)) {
#define MMF_WIDE_CASE(R) case R: return launch_bwd_dh_wide<R>(p, st);
      MMF_WIDE_CASE(64) MMF_WIDE_CASE(80) MMF_WIDE_CASE(96) MMF_WIDE_CASE(112) MMF_WIDE_CASE(128)
      MMF_WIDE_CASE(144) MMF_WIDE_CASE(160) MMF_WIDE_CASE(176) MMF_WIDE_CASE(192) MMF_WIDE_CASE(208)
#undef MMF_WIDE_CASE
      default: return launch_bwd_dh_wide<224>(p, st);
    }
  }
  if (p.fused_prep && !dh_short_grid(p.N, p.H)) return MMF_ERR_ARG;
  const int ntn = (p.H + 127) / 128;
  if ((p.N / 128) * ntn >= 256) {
    using T = Tile<128, 128, 2, 2, true, false>;
    p.mt_count = (int)((p.N + 127) / 128); p.nt_count = ntn;
    return launch_tiled<T>("bwd_dh_kernel", bwd_dh_kernel<T, false>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
  }
  {
    // 64 x 128 tiles (two column tiles of a row tile instead of four: the dP operand and K-prep are built twice, not four
    // times, and a workgroup's chunk holds 32 MFMAs per wave instead of 16) where they fill the 512 slots once: measured
    // (round 4, same call, K-dh us 64 x 64 -> 64 x 128): 14,000 rows 56.7 -> 50.0, 16,000 57.1 -> 50.1; 12,288 43.2 -> 49.1,
    // 10,000 42.6 -> 49.0, 4,096 23.8 -> 33.1 -- a round of these tiles is 50 us whatever fills it, so only from 13,000 rows
    static const int wide_min = tune_int("MMF_DH_64X128", 13000);
    if (wide_min > 0 && p.N >= wide_min && p.fused_prep && p.g.gated && p.H % 128 == 0 && (2 * p.g.D / KC) % 4 == 0) {
      using T2 = Tile<64, 128, 2, 2, true, false>;
      p.mt_count = (int)((p.N + 63) / 64); p.nt_count = p.H / 128;
      p.deep = 1;
      return launch_tiled_extra<T2>("bwd_dh_kernel", bwd_dh_kernel<T2, true>, p, grid_for_tiles(p.mt_count, p.nt_count),
                                    (3 * T2::BM + 16) * 4, st);
    }
  }
  using T = Tile<64, 64, 2, 2, true, false>;
  p.mt_count = (int)((p.N + 63) / 64); p.nt_count = (p.H + 63) / 64;
  static const int env_deep = tune_int("MMF_DEEP", 1);     // A/B switch
  static const int deep_cap = tune_int("MMF_DEEP_MAX", 1024);   // tuning override (10k bag, 628 workgroups: 252 -> 242 us per step)
  p.deep = env_deep && (int64_t)p.mt_count * p.nt_count <= deep_cap && p.g.gated && (2 * p.g.D / KC) % 4 == 0 ? 1 : 0;
  if (p.fused_prep)
    return launch_tiled_extra<T>("bwd_dh_kernel", bwd_dh_kernel<T, true>, p, grid_for_tiles(p.mt_count, p.nt_count),
                                 (3 * T::BM + 16) * 4, st);
  return launch_tiled<T>("bwd_dh_kernel", bwd_dh_kernel<T, false>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
}

int launch_nn(NnParams p, hipStream_t st) {
  if (p.K % KC != 0 || p.lda % 4 != 0 || p.ldb % 4 != 0 || p.N % 4 != 0) return MMF_ERR_SHAPE;
  if (p.M <= 0) return MMF_OK;
  const int ntn = (p.N + 127) / 128;
  if ((p.M / 128) * ntn >= 256) {
    using T = Tile<128, 128, 2, 2, true, false>;
    p.mt_count = (int)((p.M + 127) / 128); p.nt_count = ntn;
    return launch_tiled<T>("gemm_nn_kernel", gemm_nn_kernel<T>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
  }
  using T = Tile<64, 64, 2, 2, true, false>;
  p.mt_count = (int)((p.M + 63) / 64); p.nt_count = (p.N + 63) / 64;
  return launch_tiled<T>("gemm_nn_kernel", gemm_nn_kernel<T>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
}

// Tile / split plan of the TN GEMM.  256x256 tiles (8 waves, one workgroup per CU): 65 FLOP per operand byte
// instead of 32, which is what the per-CU load rate needs (profiles/r01/load_rate.txt); the price is twice the
// splits (slab traffic) for the same number of workgroups, so it is used only for long K.
int tn_tile_dim(int64_t K, int D_gate) {
  static const int env = tune_int("MMF_TN_WIDE", 1);
  static const int kmin = tune_int("MMF_TN_WIDE_MIN", 12288);   // tuning override
  if (!env || K < kmin) return 128;       // measured crossover: 10k bags 236 vs 243 us per step, 14k 294 vs 289
  (void)D_gate;
  return 256;
}
int tn_splits(int64_t K, int total_tiles, int tile) {
  static const int env_splits = tune_int("MMF_TN_SPLITS", 0);   // tuning override
  int splits = (tile == 256 ? 256 : 512) / (total_tiles > 0 ? total_tiles : 1);
  if (env_splits > 0) splits = env_splits;
  const int64_t max_splits = (K + 127) / 128;
  if (splits > max_splits) splits = (int)max_splits;
  return splits < 1 ? 1 : splits;
}

// the large-bag tile is instantiated per attention-dropout state (no per-element test in the gate tiles' staging path)
template <class T>
static int launch_tn_grid(const TnParams& p, int grid, hipStream_t st) {
  if constexpr (T::BM == 256 || T::SPLIT) {
    const char* name = T::SPLIT ? "tn_split_kernel" : "tn_kernel";
    if (p.g.drop_p > 0.f) return launch_tiled<T>(name, tn_kernel<T, 1>, p, grid, st);
    return launch_tiled<T>(name, tn_kernel<T, 0>, p, grid, st);
  }
  return launch_tiled<T>("tn_kernel", tn_kernel<T>, p, grid, st);
}

template <class T>
static int launch_tn_t(TnParams p, hipStream_t st) {
  if (p.k_per_split % 4 != 0 || p.splits < 1) return MMF_ERR_ARG;
  int tiles = 0, blocks = 0;
  for (int i = 0; i < p.nprob; ++i) {
    TnProblem& q = p.prob[i];
    if (q.Ncols % 4 != 0 || q.ldb % 4 != 0 || q.M % 4 != 0) return MMF_ERR_SHAPE;
    if (q.kind == TN_A_PLAIN && q.lda % 4 != 0) return MMF_ERR_SHAPE;
    if (q.splits <= 0) { q.splits = p.splits; q.k_per_split = p.k_per_split; }
    if (q.k_per_split % 4 != 0 || q.k_per_split < 4) return MMF_ERR_ARG;
    if (q.kind == TN_A_GATE) {
      const int dt = p.g.gated ? T::BM / 2 : T::BM;      // attention dims per tile (both halves together when gated)
      q.tiles_m = (p.g.D + dt - 1) / dt;
    } else {
      q.tiles_m = (q.M + T::BM - 1) / T::BM;
    }
    q.tiles_n = (q.Ncols + T::BN - 1) / T::BN;
    q.tile_begin = tiles;
    q.block_begin = blocks;
    tiles += q.tiles_m * q.tiles_n;
    blocks += q.tiles_m * q.tiles_n * q.splits;
  }
  if (blocks == 0) return MMF_OK;
  p.total_tiles = tiles;
  static const int env_xcd = tune_int("MMF_TN_XCD", -1);
  p.xcd_map = env_xcd == 0 ? 0 : 2;      // default: the table (same speed, a third less HBM traffic: PMC 682 -> 477 MB)
  if (p.xcd_map == 2) {
    // Pack groups (= the tiles of one problem in one split: they share the A or the B panel) into 8 bins, one per
    // XCD, largest groups first, each into the least loaded bin; bin x, slot j is workgroup x + 8 j.
    int bin_load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool ok = tiles <= 32;
    for (int i = 0; i < 512; ++i) p.map[i] = 0xFFFFu;
    int order[6], np = p.nprob;
    for (int i = 0; i < np; ++i) order[i] = i;
    for (int i = 0; i < np; ++i)                       // problems by tile count, descending
      for (int j = i + 1; j < np; ++j)
        if (p.prob[order[j]].tiles_m * p.prob[order[j]].tiles_n > p.prob[order[i]].tiles_m * p.prob[order[i]].tiles_n) {
          const int t = order[i]; order[i] = order[j]; order[j] = t;
        }
    for (int oi = 0; oi < np && ok; ++oi) {
      const TnProblem& q = p.prob[order[oi]];
      const int n = q.tiles_m * q.tiles_n;
      if (n == 0) continue;
      if (q.splits >= 2048) { ok = false; break; }
      for (int s = 0; s < q.splits && ok; ++s) {
        int x = 0;
        for (int k = 1; k < 8; ++k) if (bin_load[k] < bin_load[x]) x = k;
        if (8 * (bin_load[x] + n) > 512) { ok = false; break; }
        for (int t = 0; t < n; ++t) p.map[x + 8 * (bin_load[x] + t)] = (uint16_t)((s << 5) | (q.tile_begin + t));
        bin_load[x] += n;
      }
    }
    if (ok) {
      int mx = 0;
      for (int k = 0; k < 8; ++k) mx = bin_load[k] > mx ? bin_load[k] : mx;
      return launch_tn_grid<T>(p, 8 * mx, st);
    }
    p.xcd_map = 0;       // too many tiles for the table: plain order
  }
  return launch_tn_grid<T>(p, blocks, st);
}

int launch_tn(TnParams p, hipStream_t st) {
  if (p.tile == 256 && p.split) return launch_tn_t<TileSp<256, 256, 2, 4>>(p, st);
  if (p.tile == 256) return launch_tn_t<Tile<256, 256, 2, 4, false, false, 2>>(p, st);
  if (p.split && p.K >= split_min_rows()) return launch_tn_t<TileSp<128, 128, 2, 2>>(p, st);
  return launch_tn_t<Tile<128, 128, 2, 2, false, false, 2>>(p, st);
}

int launch_reduce(ReduceParams p, hipStream_t st) {
  int blocks = 0;
  for (int i = 0; i < p.nseg; ++i) {
    ReduceSeg& s = p.seg[i];
    s.tall = s.nsplit >= 64 && s.len >= 4 && (s.len & 3) == 0 && (s.stride & 3) == 0;
    s.block_begin = blocks;
    blocks += s.tall ? (s.len + 63) / 64 : (s.len + 1023) / 1024;
  }
  if (blocks == 0) return MMF_OK;
  { ProfScope ps("reduce_kernel", st); hipLaunchKernelGGL(reduce_kernel, dim3(blocks), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

// diagnostic: read and clear this translation unit's phase stamps (zeros unless built with -DMMF_STAMPS)
void debug_stamps_bwd(unsigned long long* out8) {
#ifdef MMF_STAMPS
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), 8 * sizeof(unsigned long long));
  unsigned long long z[8] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z);
#else
  for (int i = 0; i < 8; ++i) out8[i] = 0;
#endif
}

}  // namespace mmf
