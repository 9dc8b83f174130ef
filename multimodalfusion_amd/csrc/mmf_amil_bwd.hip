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

#include "mmf_gemm_core.h"
#include "mmf_kernels.h"

namespace mmf {

__device__ inline float gate_dp(const GateBwdCtx& g, int part, float av, float bv, float wc, float dsv,
                                uint32_t idx, uint32_t thr, float dscale, float& a_d_b_d) {
  float ma = 1.f, mb = 1.f;
  if (g.drop_p > 0.f) {
    ma = keep(g.key_a, idx, thr) ? dscale : 0.f;
    if (g.gated) mb = keep(g.key_b, idx, thr) ? dscale : 0.f;
  }
  if (g.gated) {
    a_d_b_d = (av * ma) * (bv * mb);
    return part == 0 ? dsv * wc * (bv * mb) * ma * (1.f - av * av)
                     : dsv * wc * (av * ma) * mb * bv * (1.f - bv);
  }
  a_d_b_d = av * ma;
  return dsv * wc * ma * (1.f - av * av);
}

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
template <int ROWS, int NT>
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
    g = g_; row0 = row0_; tid = threadIdx.x;
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
  __device__ inline void load(int kt) {
    const int nka = g.D / KC;
    part = kt >= nka ? 1 : 0;
    d0 = (kt - part * nka) * KC;
    const unsigned soff = (unsigned)d0 * 4u;
    wc4 = bld4(rwc, 16u * (tid & 7), soff);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      ra4[i] = bld4(ra, voff[i], soff);
      rb4[i] = bld4(rb, voff[i], soff);
    }
  }
  __device__ inline void store(float* lds) const {
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

// B[k][n] = stacked [Wa ; Wb] rows (k < D -> Wa[k], else Wb[k-D]); m-contiguous image
template <int ROWS, int NT>
struct LoadWab_M {
  using Map = MMap<ROWS, NT>;
  rsrc_t ra, rb;
  int D, tid;
  unsigned hb;
  unsigned voff[Map::NV];
  float4 r[Map::NV];
  __device__ inline void init(const float* wa, const float* wb, int H, int D_, int col0) {
    D = D_; tid = threadIdx.x; hb = (unsigned)H * 4u;
    ra = make_rsrc(wa, (unsigned)D * hb);
    rb = make_rsrc(wb ? wb : wa, (unsigned)D * hb);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      int c = col0 + 4 * Map::c4(tid, i);
      voff[i] = (Map::valid(tid, i) && c < H) ? (unsigned)Map::krow(tid, i) * hb + (unsigned)c * 4u : OOB;
    }
  }
  __device__ inline void load(int kt) {     // a chunk never straddles the Wa | Wb boundary (D % KC == 0)
    const int k0 = kt * KC;
    const bool second = k0 >= D;
    const unsigned soff = (unsigned)(k0 - (second ? D : 0)) * hb;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) r[i] = bld4(second ? rb : ra, voff[i], soff);
  }
  __device__ inline void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < Map::NV; ++i)
      if (Map::valid(tid, i)) st4(lds + Map::lds(tid, i), r[i]);
  }
};

template <class T>
__global__ __launch_bounds__(T::NT) void bwd_dh_kernel(BwdDhParams p) {
  extern __shared__ __align__(16) float lds[];
  int mt, nt;
  if (!tile_of_block(blockIdx.x, p.mt_count, p.nt_count, mt, nt)) return;
  const int row0 = mt * T::BM, col0 = nt * T::BN;
  LoadP_K<T::BM, T::NT> la;
  la.init(p.g, row0, (int)p.N);
  LoadWab_M<T::BN, T::NT> lb;
  lb.init(p.Wa, p.Wb, p.H, p.g.D, col0);
  f32x16 acc[T::MB][T::NB];
  const int nk = (p.g.gated ? 2 : 1) * p.g.D / KC;
  gemm_mainloop<T>(la, lb, nk, lds, acc);
  for_each_c<T>(acc, [&](int r, int c, float v) {
    int row = row0 + r, col = col0 + c;
    if (row < p.N && col < p.H) {
      size_t o = (size_t)row * p.H + col;
      float dh = v + p.p[row] * p.dM[col];
      p.du[o] = p.h[o] > 0.f ? dh * p.scale_h : 0.f;
    }
  });
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
  for_each_c<T>(acc, [&](int r, int c, float v) {
    int row = row0 + r, col = col0 + c;
    if (row < p.M && col < p.N) p.C[(size_t)row * p.ldc + col] = v;
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
};

template <int ROWS, int NT>
struct LoadA_M_Gate {   // A[k = instance][m] = dP over the stacked (d pre-tanh | d pre-sigmoid) columns
  using Map = MMap<ROWS, NT>;
  static_assert(NT % Map::VPR == 0, "a thread must own the same columns in every vector slot");
  GateBwdCtx g;
  rsrc_t ra, rb, rds;
  int part, d0, kbase, tid, kt_loaded;
  bool do_sum;
  uint32_t thr;
  float dscale;
  unsigned db;
  unsigned voff[Map::NV], voff_ds[Map::NV];
  float4 ra4[Map::NV], rb4[Map::NV], wc4;
  float dsr[Map::NV];
  float4 csum, csum2;   // column sums of dP (bias grads) and of ds.a_d.b_d (dWc)
  __device__ inline void init(const GateBwdCtx& g_, int col0, int kbase_, int kmax, bool do_sum_) {
    g = g_; kbase = kbase_; tid = threadIdx.x; do_sum = do_sum_;
    part = col0 >= g.D ? 1 : 0;
    d0 = col0 - part * g.D;
    thr = drop_threshold(g.drop_p);
    dscale = g.drop_p > 0.f ? 1.0f / (1.0f - g.drop_p) : 1.0f;
    csum = zero4(); csum2 = zero4();
    db = (unsigned)g.D * 4u;
    const unsigned rows = (unsigned)(kmax > 0 ? kmax : 0);
    ra = make_rsrc(g.a, rows * db);
    rb = make_rsrc(g.gated ? g.b : g.a, rows * db);
    rds = make_rsrc(g.ds, rows * 4u);
    const int c = d0 + 4 * Map::c4(tid, 0);
    wc4 = bld4(make_rsrc(g.Wc, db), c < g.D ? (unsigned)c * 4u : OOB, 0);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      bool ok = Map::valid(tid, i) && c < g.D;
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
      rb4[i] = bld4(rb, voff[i], k0 * db);
      dsr[i] = bld1(rds, voff_ds[i], k0 * 4u);    // 0 beyond the split's last instance => dP = 0 there
    }
  }
  __device__ inline void store(float* lds) {
    const int c = d0 + 4 * Map::c4(tid, 0);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      if (!Map::valid(tid, i)) continue;
      int k = kbase + kt_loaded * KC + Map::krow(tid, i);
      uint32_t idx = (uint32_t)k * (uint32_t)g.D + (uint32_t)c;
      const float dsv = dsr[i];
      float4 o, w;
      o.x = gate_dp(g, part, ra4[i].x, rb4[i].x, wc4.x, dsv, idx + 0, thr, dscale, w.x);
      o.y = gate_dp(g, part, ra4[i].y, rb4[i].y, wc4.y, dsv, idx + 1, thr, dscale, w.y);
      o.z = gate_dp(g, part, ra4[i].z, rb4[i].z, wc4.z, dsv, idx + 2, thr, dscale, w.z);
      o.w = gate_dp(g, part, ra4[i].w, rb4[i].w, wc4.w, dsv, idx + 3, thr, dscale, w.w);
      st4(lds + Map::lds(tid, i), o);
      if (do_sum) {
        csum.x += o.x; csum.y += o.y; csum.z += o.z; csum.w += o.w;
        csum2.x += dsv * w.x; csum2.y += dsv * w.y; csum2.z += dsv * w.z; csum2.w += dsv * w.w;
      }
    }
  }
};

// reduce a per-thread float4 column sum over the NT/VPR threads that own the same columns
template <class T>
__device__ inline void colsum_reduce_store(float* lds, float4 v, float* dst, int col0, int ncols) {
  constexpr int VPR = T::BM / 4, GROUPS = T::NT / VPR;
  const int tid = threadIdx.x;
  __syncthreads();
  st4(lds + (tid / VPR) * T::BM + 4 * (tid % VPR), v);
  __syncthreads();
  if (tid < T::BM) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < GROUPS; ++q) s += lds[q * T::BM + tid];
    if (col0 + tid < ncols) dst[col0 + tid] = s;
  }
}

template <class T>
__device__ inline void tn_store(const TnProblem& q, int split, int tm, int tn, f32x16 (&acc)[T::MB][T::NB]) {
  float* out = q.out + (size_t)split * q.split_stride;
  for_each_c<T>(acc, [&](int r, int c, float v) {
    int row = tm * T::BM + r, col = tn * T::BN + c;
    if (row < q.M && col < q.Ncols) out[(size_t)row * q.ldc + col] = v;
  });
}

template <class T>
__global__ __launch_bounds__(T::NT) void tn_kernel(TnParams p) {
  extern __shared__ __align__(16) float lds[];
  // XCD-aware block -> (split, tile) map.  Blocks b, b+8, ... share an XCD (round-robin dispatch; speed only):
  // every tile of one K-split is put on ONE XCD, so the operand rows of that split are fetched into that
  // XCD's L2 once and shared by all its tiles, instead of once per tile through eight different L2s
  // (the first version moved ~1.3 GB per launch through the fabric and ran at 55 % MFMA utilisation).
  const int b = blockIdx.x;
  int split, tg;
  if (p.xcd_map) {
    const int xcd = b & 7, idx = b >> 3;
    split = xcd + 8 * (idx / p.total_tiles);
    tg = idx % p.total_tiles;
  } else {
    split = b / p.total_tiles;
    tg = b - split * p.total_tiles;
  }
  if (split >= p.splits) return;
  int pi = 0;
  for (int i = 1; i < p.nprob; ++i)
    if (tg >= p.prob[i].block_begin) pi = i;
  const TnProblem& q = p.prob[pi];
  const int t = tg - q.block_begin;
  const int tm = t / q.tiles_n, tn = t - tm * q.tiles_n;
  const int64_t kb64 = (int64_t)split * p.k_per_split;
  const int kbase = (int)(kb64 < p.K ? kb64 : p.K);
  const int kmax = (int)((kb64 + p.k_per_split) < p.K ? (kb64 + p.k_per_split) : p.K);
  const int nk = (kmax - kbase + KC - 1) / KC;
  const bool do_sum = tn == 0 && q.colsum != nullptr;

  LoadM<T::BN, T::NT> lb;
  lb.init(q.B, q.ldb, tn * T::BN, q.Ncols, kbase, kmax);
  f32x16 acc[T::MB][T::NB];
  if (q.kind == TN_A_PLAIN) {
    LoadA_M_Plain<T::BM, T::NT> la;
    la.init(q.A, q.lda, tm * T::BM, q.M, kbase, kmax, do_sum);
    gemm_mainloop<T>(la, lb, nk, lds, acc);
    tn_store<T>(q, split, tm, tn, acc);
    if (do_sum) colsum_reduce_store<T>(lds, la.csum, q.colsum + (size_t)split * q.colsum_stride, tm * T::BM, q.M);
  } else {
    LoadA_M_Gate<T::BM, T::NT> la;
    la.init(p.g, tm * T::BM, kbase, kmax, do_sum);
    gemm_mainloop<T>(la, lb, nk, lds, acc);
    tn_store<T>(q, split, tm, tn, acc);
    if (do_sum) {
      colsum_reduce_store<T>(lds, la.csum, q.colsum + (size_t)split * q.colsum_stride, tm * T::BM, q.M);
      if (la.part == 0 && q.colsum2)
        colsum_reduce_store<T>(lds, la.csum2, q.colsum2 + (size_t)split * q.colsum2_stride, la.d0, p.g.D);
    }
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
      if (threadIdx.x == 0) s.out[e] = red[0] + red[1] + red[2] + red[3];
      __syncthreads();
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
    st4(s.out + j, acc);
  } else {
    for (int e = j; e < j + 4 && e < s.len; ++e) {
      float acc = 0.f;
      for (int k = 0; k < s.nsplit; ++k) acc += s.in[(size_t)k * s.stride + e];
      s.out[e] = acc;
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

int launch_bwd_dh(BwdDhParams p, hipStream_t st) {
  if (p.g.D % KC != 0 || p.H % 4 != 0) return MMF_ERR_SHAPE;
  if (p.N <= 0) return MMF_OK;
  const int ntn = (p.H + 127) / 128;
  if ((p.N / 128) * ntn >= 256) {
    using T = Tile<128, 128, 2, 2, true, false>;
    p.mt_count = (int)((p.N + 127) / 128); p.nt_count = ntn;
    return launch_tiled<T>("bwd_dh_kernel", bwd_dh_kernel<T>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
  }
  using T = Tile<64, 64, 2, 2, true, false>;
  p.mt_count = (int)((p.N + 63) / 64); p.nt_count = (p.H + 63) / 64;
  return launch_tiled<T>("bwd_dh_kernel", bwd_dh_kernel<T>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
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

int launch_tn(TnParams p, hipStream_t st) {
  using T = Tile<128, 128, 2, 2, false, false>;
  if (p.k_per_split % KC != 0 || p.splits < 1) return MMF_ERR_ARG;
  int blocks = 0;
  for (int i = 0; i < p.nprob; ++i) {
    TnProblem& q = p.prob[i];
    if (q.Ncols % 4 != 0 || q.ldb % 4 != 0 || q.M % 4 != 0) return MMF_ERR_SHAPE;
    if (q.kind == TN_A_PLAIN && q.lda % 4 != 0) return MMF_ERR_SHAPE;
    if (q.kind == TN_A_GATE && (p.g.D % T::BM != 0)) return MMF_ERR_SHAPE;   // a tile never straddles the a|b halves
    q.tiles_m = (q.M + T::BM - 1) / T::BM;
    q.tiles_n = (q.Ncols + T::BN - 1) / T::BN;
    q.block_begin = blocks;          // first global tile index of this problem
    blocks += q.tiles_m * q.tiles_n;
  }
  if (blocks == 0) return MMF_OK;
  p.total_tiles = blocks;
  static const int env_xcd = getenv("MMF_TN_XCD") ? atoi(getenv("MMF_TN_XCD")) : -1;
  p.xcd_map = env_xcd >= 0 ? env_xcd : 0;
  const int grid = (p.xcd_map ? 8 * ((p.splits + 7) / 8) : p.splits) * blocks;
  return launch_tiled<T>("tn_kernel", tn_kernel<T>, p, grid, st);
}

int launch_reduce(ReduceParams p, hipStream_t st) {
  int blocks = 0;
  for (int i = 0; i < p.nseg; ++i) {
    p.seg[i].block_begin = blocks;
    blocks += (p.seg[i].len + 1023) / 1024;
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
