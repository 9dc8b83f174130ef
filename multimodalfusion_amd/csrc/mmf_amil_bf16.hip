// bf16-storage attention-MIL kernels for MI355X (gfx950): v_mfma_f32_32x32x16_bf16, fp32 accumulate.
// Same decomposition as the fp32 path (mmf_amil_fwd.hip / mmf_amil_bwd.hip) -- K-lin, K-gate, K-pool, K-dh (K-prep
// fused), K-tn, K-red -- and the same formulas; what changes is the storage type of the bag
// and of every saved activation (mmf_bf16.h lists the rounding points), which halves the HBM traffic of a path that,
// at bf16 MFMA rates, is bound by HBM and by the per-CU operand delivery rather than by the matrix cores.
//
//   NT kernels (K-lin, K-gate, K-dh) reuse the fp32 GEMM core: a 64-bf16 chunk row is the same 128 bytes as a
//   32-float row, so the loaders, the padded LDS image and the pipeline are shared and only the MFMA differs
//   (Tile<..., BF16 = true>).
//   K-tn contracts over the instance index, which is the strided dimension of both operands.  The chunk is staged
//   as it lies in HBM ([instance][column], 16-byte copies) and the MFMA fragments are read with
//   ds_read_b64_tr_b16, gfx950's transposing LDS read: per 16-lane group it takes a 4-instance x 16-column block
//   and hands every lane ONE column's 4 instances, i.e. 4 consecutive k of that lane's output row.
#include <type_traits>
#include <cstdlib>

#include "mmf_gemm_core.h"
#include "mmf_gemm_dma.h"
#include "mmf_bf16.h"

namespace mmf {

// Diagnostic build only (-DMMF_STAMPS): per-kernel phase cycles, kernel k in slots [8k, 8k+8):
// {prologue, main loop, epilogue, -, -, -, -, waves}.  The shipped library contains none of this.
#ifdef MMF_STAMPS
static __device__ unsigned long long g_bst[32];
__device__ inline unsigned long long real_now() {      // constant 100 MHz counter: calibrates the shader clock
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define BST_BEGIN() unsigned long long bst_real = real_now(), bst_prev = stamp_now(), bst_t
#define BST_MARK(k, slot) do { bst_t = stamp_now(); if ((threadIdx.x & 63) == 0) atomicAdd(&g_bst[8 * (k) + (slot)], bst_t - bst_prev); bst_prev = bst_t; } while (0)
#define BST_COUNT(k) do { if ((threadIdx.x & 63) == 0) { atomicAdd(&g_bst[8 * (k) + 7], 1ull); atomicAdd(&g_bst[8 * (k) + 6], real_now() - bst_real); } } while (0)
#else
#define BST_BEGIN()
#define BST_MARK(k, slot)
#define BST_COUNT(k)
#endif
enum { BST_LIN = 0, BST_GATE = 1, BST_DH = 2, BST_TN = 3 };   // the fused forward uses BST_LIN: {projection loop, its epilogue, gate phase, pooling}

void debug_stamps_fwd2(unsigned long long* out8);   // mmf_amil_bf16_fwd2.hip
void debug_stamps_bf16(unsigned long long* out32) {
#ifdef MMF_STAMPS
  hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_bst), 32 * sizeof(unsigned long long));
  unsigned long long z[32] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_bst), z, sizeof z);
  debug_stamps_fwd2(out32);
#else
  for (int i = 0; i < 32; ++i) out32[i] = 0;
#endif
}

template <class T, class P>
static int launch_tiled_b(const char* name, void (*kern)(P), const P& p, int grid, int lds_bytes, hipStream_t st) {
  if (int e = set_dyn_lds(reinterpret_cast<const void*>(kern), lds_bytes)) return e;
  ProfScope ps(name, st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(T::NT), lds_bytes, st, p);
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

// =============================================================================================
// K-cvt : fp32 parameters -> bf16 copies (optionally transposed), a few hundred thousand elements
// =============================================================================================
__global__ __launch_bounds__(256) void cvt_bf16_kernel(CvtParams p) {
  const int b = blockIdx.x;
  int si = 0;
  for (int i = 1; i < p.nseg; ++i)
    if (b >= p.seg[i].block_begin) si = i;
  const CvtSeg& s = p.seg[si];
  const int e = (b - s.block_begin) * 256 + threadIdx.x;
  if (e >= s.rows * s.cols) return;
  const int r = e / s.cols, c = e - r * s.cols;
  const bf16_t v = f2bf(s.src[e]);
  if (s.transpose == 3) {
    const int kt = c >> 6, q = (c >> 4) & 3, hh = (c >> 3) & 1, w = r >> 6, fb = (r >> 5) & 1;
    s.dst[((size_t)((((kt * 4 + w) * 4 + q) * 2 + fb) * 64 + hh * 32 + (r & 31))) * 8 + (c & 7)] = v;
  } else if (s.transpose == 5) {             // r = attention dim d, c = hidden feature f
    const int kt = r >> 5, kk = s.c0 + (r & 31), q = kk >> 4, hh = (kk >> 3) & 1, w = c >> 6, fb = (c >> 5) & 1;
    s.dst[((size_t)((((kt * 4 + w) * 4 + q) * 2 + fb) * 64 + hh * 32 + (c & 31))) * 8 + (kk & 7)] = v;
  } else if (s.transpose == 4) {
    const int ps = r >> 6, w = (r >> 4) & 3, sk = c >> 4, hh = (c >> 3) & 1;
    s.dst[((size_t)(((ps * 4 + w) * 16 + sk) * 64 + hh * 32 + s.c0 + (r & 15))) * 8 + (c & 7)] = v;
  } else if (s.transpose == 2) s.dst[(size_t)c * s.dst_ld + s.c0 + (r >> 5) * 64 + (r & 31)] = v;   // 32-row blocks, 64 apart
  else if (s.transpose) s.dst[(size_t)c * s.dst_ld + s.c0 + r] = v;
  else s.dst[(size_t)r * s.dst_ld + s.c0 + c] = v;
}

int launch_cvt_bf16(CvtParams p, hipStream_t st) {
  int blocks = 0;
  for (int i = 0; i < p.nseg; ++i) {
    p.seg[i].block_begin = blocks;
    blocks += (p.seg[i].rows * p.seg[i].cols + 255) / 256;
  }
  if (blocks == 0) return MMF_OK;
  { ProfScope ps("cvt_bf16_kernel", st); hipLaunchKernelGGL(cvt_bf16_kernel, dim3(blocks), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

// row tiles: 256 x 256 (8 waves as 2 x 4, wave tile 128 x 64) or 128 x 256; one workgroup per CU either way
using TileB256 = Tile<256, 256, 2, 4, true, true, 4, true>;
using TileB128 = Tile<128, 256, 2, 4, true, true, 4, true>;

// 256-row tiles unless 128-row tiles fill the 256 CUs in fewer (half-length) rounds
static inline int pick_bm(int64_t rows, int ntn) {
  static const int env = tune_int("MMF_BF16_BM", 0);   // tuning override
  if (env == 128 || env == 256) return env;
  const int64_t t256 = ((rows + 255) / 256) * ntn, t128 = ((rows + 127) / 128) * ntn;
  const double c256 = (double)((t256 + 255) / 256) * 2.15, c128 = (double)((t128 + 255) / 256) * 1.15;
  return c128 < c256 ? 128 : 256;
}

// =============================================================================================
// K-lin : h = bf16(drop(relu(x.W1^T + b1)))
// =============================================================================================
// LDS-DMA main loop (mmf_gemm_dma.h).  A register-staged version of this kernel on the fp32 core measured the same
// (91 vs 94 us at 100k rows): the kernel is bound by the per-CU operand delivery (x from HBM plus the W1 tile re-read
// from L2 by every row tile) and by its un-overlapped epilogue, not by the staging method (profiles/r01/README.md).
using TileS128 = TileS<128, 256, 2, 4>;
using TileS256 = TileS<256, 256, 2, 4, 2>;      // two stages of 64 KB: half the L2 re-reads of the weight tile

template <class T>
__global__ __launch_bounds__(T::NT) void linear_bf16_dma_kernel(LinearBfParams p) {
  extern __shared__ __align__(16) char ldsc[];
  int mt, nt;
  if (!tile_of_block(blockIdx.x, p.mt_count, p.nt_count, mt, nt)) return;
  const int row0 = mt * T::BM, col0 = nt * T::BN;
  BST_BEGIN();
  DmaK<T::BM, T::NT> la;
  la.init(p.x, p.K, row0, (int)p.M);
  DmaK<T::BN, T::NT> lb;
  lb.init(p.w, p.K, col0, p.N);
  f32x16 acc[T::MB][T::NB];
  BST_MARK(BST_LIN, 0);
  gemm_mainloop_dma<T>(la, lb, p.K / 64, ldsc, acc);
  BST_MARK(BST_LIN, 1);
  float* lds = reinterpret_cast<float*>(ldsc);

  const uint32_t thr = drop_threshold(p.drop_p);
  const float scale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  const uint32_t dkey = p.drop_key + (p.seed_dev ? *p.seed_dev : 0u);
  float4 bias4[T::NB];
#pragma unroll
  for (int nb = 0; nb < T::NB; ++nb) {
    const int col = col0 + epilogue_col<T>(nb);
    bias4[nb] = (p.bias && col < p.N) ? ld4(p.bias + col) : zero4();
  }
  epilogue_rows<T>(acc, lds, [&](int mb, int nb, int r, int c, const float4 (&v)[4]) {
    const int col = col0 + c;
    if (col >= p.N) return;
    const float4 b4 = bias4[nb];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = row0 + r + 8 * t;
      if (row >= p.M) continue;
      float y[4] = {v[t].x + b4.x, v[t].y + b4.y, v[t].z + b4.z, v[t].w + b4.w};
      const uint32_t idx = (uint32_t)row * (uint32_t)p.N + (uint32_t)col;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = fmaxf(y[e], 0.f);
        if (p.drop_p > 0.f) y[e] = keep(dkey, idx + e, thr) ? y[e] * scale : 0.f;
      }
      *reinterpret_cast<uint2*>(p.y + (size_t)row * p.N + col) = pack4(y[0], y[1], y[2], y[3]);
    }
  });
  BST_MARK(BST_LIN, 2);
  BST_COUNT(BST_LIN);
}

int launch_linear_bf16(LinearBfParams p, hipStream_t st) {
  if (p.K % 64 != 0 || p.N % 256 != 0) return MMF_ERR_SHAPE;
  if (p.M <= 0) return MMF_OK;
  p.nt_count = p.N / 256;
  if (pick_bm(p.M, p.nt_count) == 128) {
    using T = TileS128;
    p.mt_count = (int)((p.M + T::BM - 1) / T::BM);
    return launch_tiled_b<T>("linear_bf16_kernel", linear_bf16_dma_kernel<T>, p, grid_for_tiles(p.mt_count, p.nt_count), T::LDS_BYTES, st);
  }
  using T = TileS256;
  p.mt_count = (int)((p.M + T::BM - 1) / T::BM);
  return launch_tiled_b<T>("linear_bf16_kernel", linear_bf16_dma_kernel<T>, p, grid_for_tiles(p.mt_count, p.nt_count), T::LDS_BYTES, st);
}

// =============================================================================================
// K-gate : a, b saved as bf16; per-row partial scores from those same (rounded) values
// =============================================================================================
// B-operand rows alternate 32-row blocks (a, b) of the same 32 attention dims (the fp32 kernel's layout): a wave's
// nb = 2t, 2t+1 accumulators are the tanh and the sigmoid pre-activation of the same (instance, d).
template <int ROWS, int NT, bool GATED>
struct LoadGateWB {
  using Map = KMap<ROWS, NT>;
  rsrc_t ra, rb;
  int tid;
  int which[Map::NV];
  unsigned voff[Map::NV];
  float4 r[Map::NV];
  __device__ inline void init(const bf16_t* wa, const bf16_t* wb, int H, int D, int d0) {
    tid = threadIdx.x;
    ra = make_rsrc(wa, (unsigned)D * (unsigned)H * 2u);
    rb = make_rsrc(GATED ? wb : wa, (unsigned)D * (unsigned)H * 2u);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      const int j = Map::row(tid, i);
      int w, d;
      if (!GATED) { w = 0; d = d0 + j; }
      else { w = (j >> 5) & 1; d = d0 + (j >> 6) * 32 + (j & 31); }
      which[i] = __builtin_amdgcn_readfirstlane(w);
      voff[i] = (Map::valid(tid, i) && d < D) ? ((unsigned)d * (unsigned)H + 8u * Map::c4(tid, i)) * 2u : OOB;
    }
  }
  __device__ inline void load(int kt) {
    const unsigned soff = (unsigned)kt * 128u;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) r[i] = bld4(which[i] ? rb : ra, voff[i], soff);
  }
  __device__ inline void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < Map::NV; ++i)
      if (Map::valid(tid, i)) st4(lds + Map::lds(tid, i), r[i]);
  }
};

template <class T, bool GATED>
__global__ __launch_bounds__(T::NT) void gate_bf16_kernel(GateBfParams p) {
  extern __shared__ __align__(16) float lds[];
  constexpr int DT = GATED ? T::BN / 2 : T::BN;
  int mt, nt;
  if (!tile_of_block(blockIdx.x, p.mt_count, p.nt_count, mt, nt)) return;
  const int row0 = mt * T::BM, d0 = nt * DT;
  BST_BEGIN();
  LoadK<T::BM, T::NT> la;
  la.init(reinterpret_cast<const float*>(p.h), p.H / 2, row0, (int)p.N);
  LoadGateWB<T::BN, T::NT, GATED> lb;
  lb.init(p.Wa, p.Wb, p.H, p.D, d0);
  f32x16 acc[T::MB][T::NB];
  BST_MARK(BST_GATE, 0);
  gemm_mainloop<T>(la, lb, p.H / 64, lds, acc);
  BST_MARK(BST_GATE, 1);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int rr = lane >> 3, c4 = lane & 7;
  const uint32_t thr = drop_threshold(p.drop_p);
  const bool drop = p.drop_p > 0.f;
  const float dscale = drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  const uint32_t sdev = p.seed_dev ? *p.seed_dev : 0u;
  const uint32_t key_a = p.key_a + sdev, key_b = p.key_b + sdev;
  float* blk = lds + wave * (32 * EPI_STRIDE);
  float* sred = lds + (T::NT / 64) * (32 * EPI_STRIDE);   // [WN][BM] row partials

  constexpr int NPAIR = GATED ? T::NB / 2 : T::NB;
  float4 ba_r[NPAIR], bb_r[NPAIR], wc_r[NPAIR];
#pragma unroll
  for (int t = 0; t < NPAIR; ++t) {
    const int d = d0 + (wn * NPAIR + t) * 32 + 4 * c4;
    const bool dok = d < p.D;
    ba_r[t] = dok ? ld4(p.ba + d) : zero4();
    bb_r[t] = (GATED && dok) ? ld4(p.bb + d) : zero4();
    wc_r[t] = dok ? ld4(p.Wc + d) : zero4();
  }
#pragma unroll
  for (int mb = 0; mb < T::MB; ++mb) {
    float rowsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NPAIR; ++t) {
      const int d = d0 + (wn * NPAIR + t) * 32 + 4 * c4;
      const bool dok = d < p.D;
      float4 va[4], vb[4];
      transpose_block(acc[mb][GATED ? 2 * t : t], blk, lane, va);
      if constexpr (GATED) transpose_block(acc[mb][2 * t + 1], blk, lane, vb);
      const float4 ba4 = ba_r[t], bb4 = bb_r[t], wc4 = wc_r[t];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = row0 + (wm * T::MB + mb) * 32 + rr + 8 * q;
        float av[4] = {fast_tanh(va[q].x + ba4.x), fast_tanh(va[q].y + ba4.y), fast_tanh(va[q].z + ba4.z), fast_tanh(va[q].w + ba4.w)};
        float bv[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (GATED) {
          bv[0] = fast_sigmoid(vb[q].x + bb4.x); bv[1] = fast_sigmoid(vb[q].y + bb4.y);
          bv[2] = fast_sigmoid(vb[q].z + bb4.z); bv[3] = fast_sigmoid(vb[q].w + bb4.w);
        }
        // the scores use a, b AS SAVED (bf16): forward and backward see the same activations
        const uint2 pa = pack4(av[0], av[1], av[2], av[3]), pb = pack4(bv[0], bv[1], bv[2], bv[3]);
        unpack2(pa.x, av[0], av[1]); unpack2(pa.y, av[2], av[3]);
        if constexpr (GATED) { unpack2(pb.x, bv[0], bv[1]); unpack2(pb.y, bv[2], bv[3]); }
        if (row < p.N && dok) {
          const size_t o = (size_t)row * p.D + d;
          if (p.a) {     // null in forward-only (inference) calls
            *reinterpret_cast<uint2*>(p.a + o) = pa;
            if constexpr (GATED) *reinterpret_cast<uint2*>(p.b + o) = pb;
          }
          const float wc[4] = {wc4.x, wc4.y, wc4.z, wc4.w};
          const uint32_t idx = (uint32_t)row * (uint32_t)p.D + (uint32_t)d;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float ad = av[e], bd = bv[e];
            if (drop) {
              ad = keep(key_a, idx + e, thr) ? ad * dscale : 0.f;
              if constexpr (GATED) bd = keep(key_b, idx + e, thr) ? bd * dscale : 0.f;
            }
            rowsum[q] += ad * bd * wc[e];
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float s = rowsum[q];
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      if (c4 == 0) sred[wn * T::BM + (wm * T::MB + mb) * 32 + rr + 8 * q] = s;
    }
  }
  __syncthreads();
  for (int i = tid; i < T::BM; i += T::NT) {
    const int row = row0 + i;
    if (row < p.N) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < T::WN; ++w) s += sred[w * T::BM + i];
      p.s_part[(size_t)nt * p.N + row] = s;
    }
  }
  BST_MARK(BST_GATE, 2);
  BST_COUNT(BST_GATE);
}

int gate_parts_bf16(int D, int gated) { return gated ? (D + 127) / 128 : (D + 255) / 256; }

template <class T>
static int launch_gate_bf16_t(GateBfParams p, hipStream_t st) {
  p.mt_count = (int)((p.N + T::BM - 1) / T::BM);
  const int grid = grid_for_tiles(p.mt_count, p.nt_count);
  return p.gated ? launch_tiled_b<T>("gate_bf16_kernel", gate_bf16_kernel<T, true>, p, grid, T::LDS_BYTES, st)
                 : launch_tiled_b<T>("gate_bf16_kernel", gate_bf16_kernel<T, false>, p, grid, T::LDS_BYTES, st);
}

int launch_gate_bf16(GateBfParams p, hipStream_t st) {
  if (p.H % 64 != 0 || p.D % 32 != 0) return MMF_ERR_SHAPE;
  if (p.N <= 0) return MMF_OK;
  p.nt_count = gate_parts_bf16(p.D, p.gated);
  return pick_bm(p.N, p.nt_count) == 128 ? launch_gate_bf16_t<TileB128>(p, st) : launch_gate_bf16_t<TileB256>(p, st);
}

// =============================================================================================
// K-pool : scores + per-group online-softmax partials over the bf16 h
// =============================================================================================
constexpr int POOLB_MAX_ROWS = 8192;

__global__ __launch_bounds__(256) void pool_partial_bf16_kernel(PoolBfParams pb) {
  __shared__ float s_lds[POOLB_MAX_ROWS];
  __shared__ float red[256];
  __shared__ __align__(16) float vred[2048];   // RG * VPR == 256 slots of 8 floats
  const PoolParams& p = pb.base;
  const int tid = threadIdx.x, g = blockIdx.x;
  const int64_t r0 = (int64_t)g * p.rows_per_group;
  const int64_t r1 = r0 + p.rows_per_group < p.N ? r0 + p.rows_per_group : p.N;
  const int nrows = r1 > r0 ? (int)(r1 - r0) : 0;
  const float bc = p.bc ? p.bc[0] : 0.f;

  float lmax = -INFINITY;
  for (int i = tid; i < nrows; i += 256) {
    float s = bc;
    for (int t = 0; t < p.n_parts; ++t) s += p.s_part[(size_t)t * p.N + r0 + i];
    p.A_raw[r0 + i] = s;
    s_lds[i] = s;
    lmax = fmaxf(lmax, s);
  }
  lmax = wave_max(lmax);
  if ((tid & 63) == 0) red[tid >> 6] = lmax;
  __syncthreads();
  const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();

  const int VPR = p.H / 8;          // 16-byte vectors per row: 32 (H=256), 64 (H=512), 128 (H=1024)
  const int RG = 256 / VPR;
  const int cv = tid % VPR, rg = tid / VPR;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float lsum = 0.f;
  for (int i = rg; i < nrows; i += RG) {
    const float e = __expf(s_lds[i] - m);
    const float4 raw = ld4(reinterpret_cast<const float*>(pb.h + (size_t)(r0 + i) * p.H + 8 * cv));
    float hv[8];
    unpack8(raw, hv);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += e * hv[k];
    if (cv == 0) lsum += e;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) vred[(rg * VPR + cv) * 8 + k] = v[k];
  if (cv == 0) red[rg] = lsum;
  __syncthreads();
  float* out = p.partials + (size_t)g * (2 + p.H);
  for (int c = tid; c < p.H; c += 256) {
    float s = 0.f;
    for (int q = 0; q < RG; ++q) s += vred[q * p.H + c];      // slot (q, cv) covers columns 8cv .. 8cv+7
    out[2 + c] = s;
  }
  if (tid == 0) {
    float l = 0.f;
    for (int q = 0; q < RG; ++q) l += red[q];
    out[0] = nrows > 0 ? m : -INFINITY;
    out[1] = l;
  }
}

static int launch_pool_partial_bf16(PoolBfParams& pb, hipStream_t st) {
  PoolParams& p = pb.base;
  if (p.H != 256 && p.H != 512 && p.H != 1024) return MMF_ERR_SHAPE;
  p.n_groups = pool_groups(p.N);
  p.rows_per_group = (int)((p.N + p.n_groups - 1) / p.n_groups);
  if (p.rows_per_group > POOLB_MAX_ROWS) return MMF_ERR_SHAPE;
  { ProfScope ps("pool_partial_bf16_kernel", st); hipLaunchKernelGGL(pool_partial_bf16_kernel, dim3(p.n_groups), dim3(256), 0, st, pb); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

int launch_pool_bf16(PoolBfParams pb, hipStream_t st) {
  if (int e = launch_pool_partial_bf16(pb, st)) return e;
  return launch_pool_merge(pb.base, st);
}

// =============================================================================================
// K-fwd (fused): instance projection + gated attention scoring + pooling partials in ONE kernel (H = D = 256, gated).
//   "LDS-tiled attention": the [128 x 256] bf16 h tile a workgroup has just produced stays in LDS as the A operand
//   of the gate GEMM and as the pooled operand, so h is written once (for the backward) and never read back in the
//   forward; the tile's scores are complete inside the workgroup, so the online-softmax partial (max, sum e,
//   sum e.h) is taken in place.  Replaces K-lin + K-gate + K-pool-partial: -2 launches, -2 reads of h.
//   Phase 2 keeps the WEIGHTS stationary in registers: wave w owns gate columns (a, b) of attention dims
//   32 w .. 32 w + 31 for all 128 rows, i.e. 64 rows of [Wa ; Wb] x K = 256 as 32 MFMA B-fragments (128 VGPRs),
//   fetched once per workgroup straight from L2 while the phase-1 epilogue runs.  No staging, no barrier and no
//   memory wait inside the gate GEMM (a two-stage DMA version of this phase spent 2/3 of its time waiting for the
//   next 32 KB weight chunk: 1024 MFMA cycles per chunk cannot cover an L2 round trip).
// LDS map (bytes):  phase 1: three 48 KB stages from 0, its epilogue's fp32 transpose scratch at 86016
//                   [0, 20480) per-wave bf16 transpose scratch of phase 2
//                   [20480, 86016) h tile: 4 k-chunks x [128 rows][128 B], XOR-swizzled like a staged chunk
//                   [151552, ...) score partials [8][128], e[128], reduction scratch
// =============================================================================================
constexpr int FF_SCR2 = 0, FF_HIMG = 20480, FF_SCR1 = 86016, FF_MISC = 151552;
constexpr int FF_LDS_BYTES = FF_MISC + (8 * 128 + 128 + 4 * 256 + 16) * 4;

__global__ __launch_bounds__(512) void amil_fwd_fused_bf16_kernel(FusedFwdParams p) {
  using T = TileS128;
  extern __shared__ __align__(16) char ldsf[];
  char* lds = ldsf;
  const int mt = blockIdx.x;
  const int row0 = mt * T::BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const uint32_t sdev = p.seed_dev ? *p.seed_dev : 0u;
  float* sred = reinterpret_cast<float*>(lds + FF_MISC);        // [8 waves][128 rows]
  float* e_l = sred + 8 * 128;                                   // [128]
  float* vred = e_l + 128;                                       // [4][256]
  float* red = vred + 4 * 256;                                   // [16]
  BST_BEGIN();

  // ---------------- phase 1: u = x.W1^T (K = L), three-stage LDS-DMA pipeline -------------------------------
  f32x16 acc[T::MB][T::NB];
  {
    DmaK<T::BM, T::NT> la;
    la.init(p.x, p.L, row0, (int)p.N);
    DmaK<T::BN, T::NT> lb;
    lb.init(p.w1, p.L, 0, 256);
    gemm_mainloop_dma<T>(la, lb, p.L / 64, lds, acc);
  }
  BST_MARK(BST_LIN, 0);
  // the wave's stationary gate weights: B-fragments of rows 32 w + r of Wa (nb 0) and Wb (nb 1), all of K = 256;
  // issued now, consumed after the epilogue below
  float4 wfr[2][16];
  {
    const unsigned wbytes = 256u * 256u * 2u;
    const rsrc_t ra = make_rsrc(p.Wa, wbytes), rb = make_rsrc(p.Wb, wbytes);
    const unsigned voff = (unsigned)(32 * wave + r) * 512u + (unsigned)hh * 16u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {                       // j = 4 kc + q: k bytes = 128 kc + 32 q (+ 16 hh)
      wfr[0][j] = bld4(ra, voff, (unsigned)((j >> 2) * 128 + (j & 3) * 32));
      wfr[1][j] = bld4(rb, voff, (unsigned)((j >> 2) * 128 + (j & 3) * 32));
    }
  }
  // epilogue: h = bf16(drop(relu(u + b1))) -> HBM (for backward) and -> the LDS h tile (A operand of phase 2)
  {
    const uint32_t thr = drop_threshold(p.p_h);
    const float scale = p.p_h > 0.f ? 1.0f / (1.0f - p.p_h) : 1.0f;
    const uint32_t dkey = p.key_h + sdev;
    float4 bias4[T::NB];
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb) bias4[nb] = ld4(p.b1 + epilogue_col<T>(nb));
    auto epi = [&](auto drop_c) {              // dropout on / off decided once, not per element
      constexpr bool DROP = decltype(drop_c)::value;
      epilogue_rows<T>(acc, reinterpret_cast<float*>(lds + FF_SCR1), [&](int mb, int nb, int rr, int c, const float4 (&v)[4]) {
        const float4 b4 = bias4[nb];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int R = rr + 8 * t, row = row0 + R;
          float y[4] = {v[t].x + b4.x, v[t].y + b4.y, v[t].z + b4.z, v[t].w + b4.w};
          const uint32_t idx = (uint32_t)row * 256u + (uint32_t)c;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[e] = fmaxf(y[e], 0.f);
            if constexpr (DROP) y[e] = keep(dkey, idx + e, thr) ? y[e] * scale : 0.f;
          }
          const uint2 pk = pack4(y[0], y[1], y[2], y[3]);
          if (p.h && row < p.N) *reinterpret_cast<uint2*>(p.h + (size_t)row * 256 + c) = pk;
          const int cc = c & 63;
          *reinterpret_cast<uint2*>(lds + FF_HIMG + (c >> 6) * 16384 + R * 128 + 16 * ((cc >> 3) ^ ((R >> 1) & 7)) + (cc & 7) * 2) = pk;
        }
      });
    };
    if (p.p_h > 0.f) epi(std::true_type{}); else epi(std::false_type{});
  }
  // per-wave constants of the phase-2 epilogue
  const int dj = 32 * wave + r;                            // this lane's dim in the accumulator layout
  const float bia = p.ba[dj], bib = p.bb[dj];
  const int rowl = lane >> 1, half = lane & 1;             // after the transpose: 16 dims of one row per lane
  const int dq = 32 * wave + half * 16;
  float wc[16];
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const float4 t = ld4(p.Wc + dq + 4 * q4);
    wc[4 * q4] = t.x; wc[4 * q4 + 1] = t.y; wc[4 * q4 + 2] = t.z; wc[4 * q4 + 3] = t.w;
  }
  __syncthreads();                                         // h tile complete
  BST_MARK(BST_LIN, 1);

  // ---------------- phase 2: [128 x 256] h tile . [64 weight rows of this wave]^T, weights in registers ----------
  const uint32_t thr_a = drop_threshold(p.p_att);
  const bool drop = p.p_att > 0.f;
  const float dscale = drop ? 1.0f / (1.0f - p.p_att) : 1.0f;
  const uint32_t key_a = p.key_a + sdev, key_b = p.key_b + sdev;
  char* scr = lds + FF_SCR2 + wave * 2560;                 // [32 rows][80 B] bf16 transpose scratch of this wave
  const int sw = (r >> 1) & 7;
#pragma unroll 1
  for (int hf = 0; hf < 2; ++hf) {                         // two 64-row halves: 2 x 2 accumulator blocks at a time
    f32x16 ag[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) ag[mb][nb][i] = 0.f;
    const char* a0 = lds + FF_HIMG + (hf * 64 + r) * 128;
    float4 fa[2][2];
    fa[0][0] = *reinterpret_cast<const float4*>(a0 + 16 * (hh ^ sw));
    fa[0][1] = *reinterpret_cast<const float4*>(a0 + 32 * 128 + 16 * (hh ^ sw));
#pragma unroll
    for (int j = 0; j < 16; ++j) {                         // j = 4 kc + q
      if (j + 1 < 16) {
        const int kc = (j + 1) >> 2, q = (j + 1) & 3;
        const int o = kc * 16384 + 16 * ((2 * q + hh) ^ sw);
        fa[(j + 1) & 1][0] = *reinterpret_cast<const float4*>(a0 + o);
        fa[(j + 1) & 1][1] = *reinterpret_cast<const float4*>(a0 + 32 * 128 + o);
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float av[4] = {fa[j & 1][mb].x, fa[j & 1][mb].y, fa[j & 1][mb].z, fa[j & 1][mb].w};
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const float bv[4] = {wfr[nb][j].x, wfr[nb][j].y, wfr[nb][j].z, wfr[nb][j].w};
          ag[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_bf16(av), frag_bf16(bv), ag[mb][nb], 0, 0, 0);
        }
      }
    }
    // ---- activations in the accumulator layout, bf16 transpose through LDS, stores of a / b, score partials ----
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int R = hf * 64 + mb * 32 + rowl, row = row0 + R;
      float4 qa[2], qb[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * hh;
        *reinterpret_cast<bf16_t*>(scr + rr * 80 + r * 2) = f2bf(fast_tanh(ag[mb][0][i] + bia));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      qa[0] = *reinterpret_cast<const float4*>(scr + rowl * 80 + half * 32);
      qa[1] = *reinterpret_cast<const float4*>(scr + rowl * 80 + half * 32 + 16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * hh;
        *reinterpret_cast<bf16_t*>(scr + rr * 80 + r * 2) = f2bf(fast_sigmoid(ag[mb][1][i] + bib));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      qb[0] = *reinterpret_cast<const float4*>(scr + rowl * 80 + half * 32);
      qb[1] = *reinterpret_cast<const float4*>(scr + rowl * 80 + half * 32 + 16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (p.a && row < p.N) {
        float4* oa = reinterpret_cast<float4*>(p.a + (size_t)row * 256 + dq);
        float4* ob = reinterpret_cast<float4*>(p.b + (size_t)row * 256 + dq);
        oa[0] = qa[0]; oa[1] = qa[1];
        ob[0] = qb[0]; ob[1] = qb[1];
      }
      float av[16], bv[16];
      unpack8(qa[0], *reinterpret_cast<float(*)[8]>(av));
      unpack8(qa[1], *reinterpret_cast<float(*)[8]>(av + 8));
      unpack8(qb[0], *reinterpret_cast<float(*)[8]>(bv));
      unpack8(qb[1], *reinterpret_cast<float(*)[8]>(bv + 8));
      const uint32_t idx = (uint32_t)row * 256u + (uint32_t)dq;
      float s = 0.f;
      if (drop) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float ad = keep(key_a, idx + e, thr_a) ? av[e] * dscale : 0.f;
          const float bd = keep(key_b, idx + e, thr_a) ? bv[e] * dscale : 0.f;
          s += ad * bd * wc[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) s += av[e] * bv[e] * wc[e];
      }
      s += __shfl_xor(s, 1, 64);
      if (half == 0) sred[wave * 128 + R] = s;
    }
  }
  __syncthreads();
  BST_MARK(BST_LIN, 2);

  // ---------------- scores of the tile, online-softmax partial ------------------------------------------------
  const float bc = p.bc[0];
  float sv = -INFINITY;
  if (tid < 128) {
    const int row = row0 + tid;
    float s = bc;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += sred[w * 128 + tid];
    if (row < p.N) { p.A_raw[row] = s; sv = s; }
  }
  float m = wave_max(sv);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(red[0], red[1]);                               // the rows live in threads 0..127 = waves 0, 1
  const float ev = sv > -INFINITY ? __expf(sv - m) : 0.f;
  if (tid < 128) e_l[tid] = ev;
  const float lsum = wave_sum(ev);
  if (lane == 0) red[8 + wave] = lsum;
  __syncthreads();
  {
    const int cp = tid & 127, rg = tid >> 7, c = 2 * cp;   // columns c, c+1; rows rg*32 .. +31
    const int cc = c & 63;
    const char* hb = lds + FF_HIMG + (c >> 6) * 16384 + (cc & 7) * 2;
    float v0 = 0.f, v1 = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const int R = rg * 32 + i;
      const uint32_t w = *reinterpret_cast<const uint32_t*>(hb + R * 128 + 16 * ((cc >> 3) ^ ((R >> 1) & 7)));
      float h0, h1;
      unpack2(w, h0, h1);
      const float e = e_l[R];
      v0 += e * h0; v1 += e * h1;
    }
    vred[rg * 256 + c] = v0;
    vred[rg * 256 + c + 1] = v1;
  }
  __syncthreads();
  float* out = p.partials + (size_t)mt * (2 + 256);
  if (tid < 256) out[2 + tid] = vred[tid] + vred[256 + tid] + vred[512 + tid] + vred[768 + tid];
  if (tid == 0) { out[0] = m; out[1] = red[8] + red[9]; }
  BST_MARK(BST_LIN, 3);
  BST_COUNT(BST_LIN);
}

int fused_fwd_tiles(int64_t N) { return (int)((N + 127) / 128); }
bool fused_fwd_ok(int64_t N, int L, int H, int D) {     // callers also require gated && D == 256 (8 waves x 32 dims)
  static const int env = tune_int("MMF_BF16_FUSED", 1);   // A/B switch
  (void)D;
  return env && H == 256 && L % 64 == 0 && fused_fwd_tiles(N) <= 4096;
}

int launch_fused_fwd_bf16(FusedFwdParams p, int gated, hipStream_t st) {
  if (!gated || p.D != 256) return MMF_ERR_SHAPE;
  p.mt_count = fused_fwd_tiles(p.N);
  auto kern = amil_fwd_fused_bf16_kernel;
  if (int e = set_dyn_lds(reinterpret_cast<const void*>(kern), FF_LDS_BYTES)) return e;
  ProfScope ps("amil_fwd_fused_bf16_kernel", st);
  hipLaunchKernelGGL(kern, dim3(p.mt_count), dim3(512), FF_LDS_BYTES, st, p);
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

// =============================================================================================
// K-dh : du = bf16((dP.Wab + p dM) . relu'(h) . scale_h).  K-prep is fused in front; the loader builds dP from
//        a, b, ds, writes it to LDS (A operand) AND to HBM (the TN kernel's operand for dWa/dWb/dba/dbb), and
//        takes the dWc column sums on the way.
// =============================================================================================
// k order of a gated chunk (64 k): [0, 32) = d pre-tanh of attention dims 32 kt .. 32 kt + 31, [32, 64) = d
// pre-sigmoid of the SAME dims, so a thread loads the 8 a and 8 b of its (instance, dims) once and emits both;
// WabT is stored in the same interleaved order (launch side: CvtSeg::transpose == 2).  Ungated: 64 plain dims.
__device__ inline GateBwdCtx gate_ctx(const GateBwdBf& g) {
  GateBwdCtx c{};
  c.D = g.D; c.gated = g.gated; c.drop_p = g.drop_p;
  const uint32_t sd = g.seed_dev ? *g.seed_dev : 0u;
  c.key_a = g.key_a + sd; c.key_b = g.key_b + sd; c.seed_dev = nullptr;
  return c;
}

template <int ROWS, int NT, bool GATED>
struct LoadPB {
  static constexpr int PPR = GATED ? 4 : 8;                 // 8-dim pieces per row and chunk
  static constexpr int DPC = GATED ? 32 : 64;               // attention dims per chunk
  static constexpr int NV = ROWS * PPR / NT;
  static_assert((ROWS * PPR) % NT == 0 && NT % PPR == 0, "slot ownership must be exact");
  GateBwdCtx gc;
  rsrc_t ra, rb, rwc;
  int row0, nrows, tid, d0, D, mstk;
  bool write_dp;
  uint32_t thr;
  float dscale;
  bf16_t* dP;          // [N x mstk]
  float* dwc_l;        // LDS [waves][D]
  unsigned voff[NV];
  float dsr[NV];
  float4 ra4[NV], rb4[NV], wc_lo, wc_hi;
  __device__ static inline int row(int tid, int i) { return (tid + i * NT) / PPR; }
  __device__ static inline int piece(int tid) { return tid % PPR; }
  __device__ inline void init_lds(const GateBwdBf& g, int row0_, int nrows_, const float* ds_lds, bf16_t* dP_, bool write_dp_,
                                  float* dwc_l_) {
    gc = gate_ctx(g); gc.gated = GATED ? 1 : 0;
    D = g.D; mstk = GATED ? 2 * D : D; row0 = row0_; nrows = nrows_; tid = threadIdx.x;
    dP = dP_; write_dp = write_dp_; dwc_l = dwc_l_;
    thr = drop_threshold(g.drop_p);
    dscale = g.drop_p > 0.f ? 1.0f / (1.0f - g.drop_p) : 1.0f;
    const unsigned bytes = (unsigned)nrows * (unsigned)D * 2u;
    ra = make_rsrc(g.a, bytes);
    rb = make_rsrc(GATED ? g.b : g.a, bytes);
    rwc = make_rsrc(g.Wc, (unsigned)D * 4u);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int rl = row(tid, i), rr = row0 + rl;
      const bool ok = rr < nrows;
      voff[i] = ok ? ((unsigned)rr * (unsigned)D + 8u * piece(tid)) * 2u : OOB;
      dsr[i] = ok ? ds_lds[rl] : 0.f;
    }
  }
  __device__ inline void load(int kt) {
    d0 = kt * DPC;
    wc_lo = bld4(rwc, 32u * piece(tid), (unsigned)d0 * 4u);
    wc_hi = bld4(rwc, 32u * piece(tid) + 16u, (unsigned)d0 * 4u);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      ra4[i] = bld4(ra, voff[i], (unsigned)d0 * 2u);
      if (GATED) rb4[i] = bld4(rb, voff[i], (unsigned)d0 * 2u);
    }
  }
  __device__ inline void store(float* lds) {
    if (gc.drop_p > 0.f) store_t<true>(lds); else store_t<false>(lds);   // one branch per chunk, none per element
  }
  template <bool DROP>
  __device__ inline void store_t(float* lds) {
    const int c = d0 + 8 * piece(tid);
    const float wc[8] = {wc_lo.x, wc_lo.y, wc_lo.z, wc_lo.w, wc_hi.x, wc_hi.y, wc_hi.z, wc_hi.w};
    float wsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int rl = row(tid, i), rr = row0 + rl;
      const uint32_t idx = (uint32_t)rr * (uint32_t)D + (uint32_t)c;
      const float dsv = dsr[i];
      // one packed dword (2 dims) at a time: keeps the live set small (the 256-row tile is at the register limit)
      const uint32_t aw[4] = {__float_as_uint(ra4[i].x), __float_as_uint(ra4[i].y), __float_as_uint(ra4[i].z), __float_as_uint(ra4[i].w)};
      const uint32_t bw[4] = {__float_as_uint(rb4[i].x), __float_as_uint(rb4[i].y), __float_as_uint(rb4[i].z), __float_as_uint(rb4[i].w)};
      uint32_t pw[4], qw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a0, a1, b0 = 0.f, b1 = 0.f, w0, w1;
        unpack2(aw[q], a0, a1);
        if (GATED) unpack2(bw[q], b0, b1);
        const float oa0 = gate_dp_t<GATED, DROP, 0>(gc, a0, b0, wc[2 * q], dsv, idx + 2 * q, thr, dscale, w0);
        const float oa1 = gate_dp_t<GATED, DROP, 0>(gc, a1, b1, wc[2 * q + 1], dsv, idx + 2 * q + 1, thr, dscale, w1);
        pw[q] = pack2(oa0, oa1);
        if (GATED) {
          const float ob0 = gate_dp_t<GATED, DROP, 1>(gc, a0, b0, wc[2 * q], dsv, idx + 2 * q, thr, dscale, w0);
          const float ob1 = gate_dp_t<GATED, DROP, 1>(gc, a1, b1, wc[2 * q + 1], dsv, idx + 2 * q + 1, thr, dscale, w1);
          qw[q] = pack2(ob0, ob1);
        }
        wsum[2 * q] += dsv * w0;
        wsum[2 * q + 1] += dsv * w1;
      }
      const float4 pa = make_float4(__uint_as_float(pw[0]), __uint_as_float(pw[1]), __uint_as_float(pw[2]), __uint_as_float(pw[3]));
      float* dst = lds + rl * KSTR + 4 * piece(tid);
      st4(dst, pa);
      float4 pb4 = zero4();
      if (GATED) {
        pb4 = make_float4(__uint_as_float(qw[0]), __uint_as_float(qw[1]), __uint_as_float(qw[2]), __uint_as_float(qw[3]));
        st4(dst + 16, pb4);
      }
      if (write_dp && rr < nrows) {
        bf16_t* o = dP + (size_t)rr * mstk + c;
        *reinterpret_cast<float4*>(o) = pa;
        if (GATED) *reinterpret_cast<float4*>(o + D) = pb4;
      }
    }
    // dWc partial of this chunk's dims: sum over the wave's rows (lanes that share `piece`), one LDS row per wave
    constexpr int FIRST = PPR;          // lane bits above the piece bits select the row
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = wsum[e];
#pragma unroll
      for (int o = FIRST; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      wsum[e] = v;
    }
    if ((tid & 63) < PPR && c < D) {
      float* o = dwc_l + (tid >> 6) * D + c;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = wsum[e];
    }
  }
};

template <class T, bool GATED>
__global__ __launch_bounds__(T::NT) void dh_bf16_kernel(DhBfParams p) {
  extern __shared__ __align__(16) float lds[];
  int mt, nt;
  if (!tile_of_block(blockIdx.x, p.mt_count, p.nt_count, mt, nt)) return;
  const int row0 = mt * T::BM, col0 = nt * T::BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = T::NT / 64;
  float* ds_l = lds + 2 * T::STAGE_FLOATS;     // [BM] ds, [BM] p, [BM] g, [16] scratch, [NW][D] dWc -- behind the staging buffers
  float* p_l = ds_l + T::BM;
  float* g_l = p_l + T::BM;
  float* red = g_l + T::BM;
  float* dwc_l = red + 16;
  BST_BEGIN();
  {
    // ---- K-prep for this tile's rows: p_i = softmax weight, ds_i = p_i (dM.h_i - dM.M) + gA_i ----------
    // g_i = dM.h_i: every wave takes BM/NW rows; lanes cover 16-byte pieces of h, 8 independent loads in flight
    const int LPR = p.H / 8;                       // 16-byte pieces per row: 32, 64 or 128
    constexpr int RPW = T::BM / NW;                // rows per wave
    const int cl = LPR >= 64 ? lane : (lane & (LPR - 1));
    float dm_l[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = 8 * (cl + 64 * q);
      const float4 lo = c < p.H ? ld4(p.dM + c) : zero4(), hi = c < p.H ? ld4(p.dM + c + 4) : zero4();
      dm_l[q][0] = lo.x; dm_l[q][1] = lo.y; dm_l[q][2] = lo.z; dm_l[q][3] = lo.w;
      dm_l[q][4] = hi.x; dm_l[q][5] = hi.y; dm_l[q][6] = hi.z; dm_l[q][7] = hi.w;
    }
    const int total_j = RPW * LPR / 64;            // wave-wide loads for this wave's rows
    for (int jb = 0; jb < total_j; jb += 8) {
      float4 raw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pidx = lane + 64 * (jb + u);
        const int row = row0 + wave * RPW + pidx / LPR;
        const int rc = row < p.N ? row : (int)p.N - 1;
        raw[u] = ld4(reinterpret_cast<const float*>(p.h + (size_t)rc * p.H + 8 * (pidx % LPR)));
      }
      float part[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float hv[8];
        unpack8(raw[u], hv);
        const int q = LPR > 64 ? ((jb + u) & 1) : 0;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += hv[k] * (q ? dm_l[1][k] : dm_l[0][k]);
        part[u] = acc;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float v = part[u];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
        if (LPR >= 64) v += __shfl_xor(v, 32, 64);
        part[u] = v;
      }
      if (LPR == 32) {
        if ((lane & 31) == 0) {
#pragma unroll
          for (int u = 0; u < 8; ++u) g_l[wave * RPW + 2 * (jb + u) + (lane >> 5)] = part[u];
        }
      } else if (LPR == 64) {
        if (lane == 0) {
#pragma unroll
          for (int u = 0; u < 8; ++u) g_l[wave * RPW + jb + u] = part[u];
        }
      } else {
        if (lane == 0) {
#pragma unroll
          for (int u = 0; u < 8; u += 2) g_l[wave * RPW + (jb + u) / 2] = part[u] + part[u + 1];
        }
      }
    }
    float dmm = 0.f;
    for (int c = lane; c < p.H; c += 64) dmm += p.dM[c] * p.Mpool[c];
    dmm = wave_sum(dmm);
    __syncthreads();
    const float smax = p.stats[0], inv = 1.0f / p.stats[1];
    float dbc = 0.f;
    for (int r = tid; r < T::BM; r += T::NT) {
      const int row = row0 + r;
      float pi = 0.f, d = 0.f;
      if (row < p.N) {
        pi = __expf(p.A_raw[row] - smax) * inv;
        d = pi * (g_l[r] - dmm) + (p.gA ? p.gA[row] : 0.f);
        if (nt == 0) { p.p_out[row] = pi; p.ds_out[row] = d; }
      }
      ds_l[r] = d;
      p_l[r] = pi;
      dbc += d;
    }
    dbc = wave_sum(dbc);
    if (lane == 0) red[wave] = dbc;
    __syncthreads();
    if (tid == 0 && nt == 0) {
      float s = 0.f;
      for (int w = 0; w < NW; ++w) s += red[w];
      p.dbc_part[mt] = s;
    }
  }
  const int mstk = GATED ? 2 * p.g.D : p.g.D;
  LoadPB<T::BM, T::NT, GATED> la;
  la.init_lds(p.g, row0, (int)p.N, ds_l, p.dP, nt == 0, dwc_l);
  LoadK<T::BN, T::NT> lb;
  lb.init(reinterpret_cast<const float*>(p.WabT), mstk / 2, col0, p.H);
  f32x16 acc[T::MB][T::NB];
  BST_MARK(BST_DH, 0);
  gemm_mainloop<T>(la, lb, mstk / 64, lds, acc);
  BST_MARK(BST_DH, 1);
  if (nt == 0) {                               // per-tile dWc partial: sum the waves' rows (the loop's last barrier has passed)
    for (int d = tid; d < p.g.D; d += T::NT) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += dwc_l[w * p.g.D + d];
      p.dwc_part[(size_t)mt * p.g.D + d] = s;
    }
  }
  float4 dm4[T::NB];
#pragma unroll
  for (int nb = 0; nb < T::NB; ++nb) {
    const int col = col0 + epilogue_col<T>(nb);
    dm4[nb] = col < p.H ? ld4(p.dM + col) : zero4();
  }
  epilogue_rows<T>(acc, lds, [&](int mb, int nb, int r, int c, const float4 (&v)[4]) {
    const int col = col0 + c;
    if (col >= p.H) return;
    uint2 hraw[4];
    float pi[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {             // all loads first ...
      const int row = row0 + r + 8 * t;
      const int rc = row < p.N ? row : (int)p.N - 1;
      hraw[t] = *reinterpret_cast<const uint2*>(p.h + (size_t)rc * p.H + col);
      pi[t] = p_l[r + 8 * t];
    }
    const float4 dm = dm4[nb];
#pragma unroll
    for (int t = 0; t < 4; ++t) {             // ... then the stores
      const int row = row0 + r + 8 * t;
      if (row >= p.N) continue;
      float h0, h1, h2, h3;
      unpack2(hraw[t].x, h0, h1);
      unpack2(hraw[t].y, h2, h3);
      const float d0 = h0 > 0.f ? (v[t].x + pi[t] * dm.x) * p.scale_h : 0.f;
      const float d1 = h1 > 0.f ? (v[t].y + pi[t] * dm.y) * p.scale_h : 0.f;
      const float d2 = h2 > 0.f ? (v[t].z + pi[t] * dm.z) * p.scale_h : 0.f;
      const float d3 = h3 > 0.f ? (v[t].w + pi[t] * dm.w) * p.scale_h : 0.f;
      *reinterpret_cast<uint2*>(p.du + (size_t)row * p.H + col) = pack4(d0, d1, d2, d3);
    }
  });
  BST_MARK(BST_DH, 2);
  BST_COUNT(BST_DH);
}

int dh_bf16_row_tiles(int64_t N) { return (int)((N + 127) / 128); }   // upper bound of mt_count (128-row tiles)
int dh_bf16_tiles_used(int64_t N, int ntn) { (void)ntn; return (int)((N + 127) / 128); }

template <class T>
static int launch_dh_bf16_t(DhBfParams p, hipStream_t st) {
  p.mt_count = (int)((p.N + T::BM - 1) / T::BM);
  const int bytes = T::LDS_BYTES + (3 * T::BM + 16 + (T::NT / 64) * p.g.D) * 4;
  if (bytes > 160 * 1024) return MMF_ERR_SHAPE;
  const int grid = grid_for_tiles(p.mt_count, p.nt_count);
  return p.g.gated ? launch_tiled_b<T>("dh_bf16_kernel", dh_bf16_kernel<T, true>, p, grid, bytes, st)
                   : launch_tiled_b<T>("dh_bf16_kernel", dh_bf16_kernel<T, false>, p, grid, bytes, st);
}

int launch_dh_bf16(DhBfParams p, hipStream_t st) {
  if (p.g.D % 64 != 0 || p.H % 256 != 0 || p.H > 1024) return MMF_ERR_SHAPE;
  if (p.N <= 0) return MMF_OK;
  p.nt_count = p.H / 256;
  // 128-row tiles only: with the on-the-fly dP loader the 256-row tile needs > 256 VGPRs and spills inside the loop,
  // where a scratch reload waits on vmcnt behind the prefetch and exposes the whole HBM latency (measured: 219 vs 162 us)
  return launch_dh_bf16_t<TileB128>(p, st);
}

// =============================================================================================
// K-tn : split-K TN GEMM over bf16 operands, fragments by transposing LDS reads
// =============================================================================================
// LDS image of one operand chunk: [TNB_KCH instances][COLS bf16], row pitch COLS*2 + 64 bytes.  A 32-lane half of a
// ds_read_b64_tr_b16 covers 4 instance rows x 64 contiguous bytes; with pitch = 64 (mod 256) the 4 rows land on the
// 4 disjoint quarters of the 64 banks, so the read is conflict-free.
template <int BM_, int BN_>
struct TileT {
  static constexpr int BM = BM_, BN = BN_, WM = 2, WN = 4, NT = 512;
  static constexpr int MB = BM / WM / 32, NB = BN / WN / 32;
  static constexpr int A_PITCH = BM * 2 + 64, B_PITCH = BN * 2 + 64;
  static constexpr int A_BYTES = TNB_KCH * A_PITCH, B_BYTES = TNB_KCH * B_PITCH;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
  static_assert(A_PITCH % 256 == 64 && B_PITCH % 256 == 64, "pitch must be 64 mod 256 for conflict-free transposed reads");
};

template <int COLS, int NT>
struct TMap {    // [TNB_KCH][COLS] image in 16-byte pieces; a thread owns the same piece (8 columns) in every slot
  static constexpr int PPR = COLS / 8;
  static constexpr int NV = TNB_KCH * PPR / NT;
  static_assert(NT % PPR == 0 && (TNB_KCH * PPR) % NT == 0, "piece ownership must be slot-invariant");
  __device__ static inline int krow(int tid, int i) { return (tid + i * NT) / PPR; }
  __device__ static inline int piece(int tid) { return tid % PPR; }
};

template <int COLS, int NT, int PITCH>
struct LoadTPlain {
  using Map = TMap<COLS, NT>;
  rsrc_t rs;
  unsigned ldb, kbase_b;
  int tid;
  bool do_sum;
  unsigned voff[Map::NV];
  float4 r[Map::NV];
  float cs[8];
  __device__ inline void init(const bf16_t* s, int ld, int col0, int ncols, int kbase, int kmax, bool do_sum_) {
    tid = threadIdx.x; do_sum = do_sum_;
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    rs = make_rsrc(s, (unsigned)(kmax > 0 ? kmax : 0) * (unsigned)ld * 2u);
    ldb = (unsigned)ld * 2u;
    kbase_b = (unsigned)kbase * ldb;
    const int c = col0 + 8 * Map::piece(tid);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i)
      voff[i] = c < ncols ? (unsigned)Map::krow(tid, i) * ldb + (unsigned)c * 2u : OOB;
  }
  __device__ inline void load(int kt) {
    const unsigned soff = kbase_b + (unsigned)(kt * TNB_KCH) * ldb;
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) r[i] = bld4(rs, voff[i], soff);
  }
  __device__ inline void store(char* lds) {
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      *reinterpret_cast<float4*>(lds + Map::krow(tid, i) * PITCH + 16 * Map::piece(tid)) = r[i];
      if (do_sum) {
        float v[8];
        unpack8(r[i], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += v[e];
      }
    }
  }
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// one 32x32x16 operand (this lane: output row/column = block column 16 (g&1) + i, k = 8 (g>>1) .. +7) from two
// transposed reads 4 instance rows apart; `p` already holds the lane part of the address
template <int PITCH>
__device__ inline bf16x8 tr_frag(const char* p) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * PITCH));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <class T>
__device__ inline void tn_compute_chunk(const char* As, const char* Bs, f32x16 (&acc)[T::MB][T::NB], int wm, int wn, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int kq = 8 * (g >> 1) + (i >> 2);            // instance row (within a 16-instance step) this lane addresses
  const int cq = 16 * (g & 1) + 4 * (i & 3);         // first of the 4 columns this lane addresses
  const char* a0 = As + kq * T::A_PITCH + (wm * T::MB * 32 + cq) * 2;
  const char* b0 = Bs + kq * T::B_PITCH + (wn * T::NB * 32 + cq) * 2;
#pragma unroll
  for (int q = 0; q < TNB_KCH / 16; ++q) {
    bf16x8 fa[T::MB], fb[T::NB];
#pragma unroll
    for (int mb = 0; mb < T::MB; ++mb) fa[mb] = tr_frag<T::A_PITCH>(a0 + 16 * q * T::A_PITCH + mb * 64);
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb) fb[nb] = tr_frag<T::B_PITCH>(b0 + 16 * q * T::B_PITCH + nb * 64);
#pragma unroll
    for (int mb = 0; mb < T::MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < T::NB; ++nb)
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mb], fb[nb], acc[mb][nb], 0, 0, 0);
  }
}

template <class T, class LA, class LB>
__device__ inline void tn_mainloop(LA& la, LB& lb, int nk, char* lds, f32x16 (&acc)[T::MB][T::NB]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
#pragma unroll
  for (int mb = 0; mb < T::MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  if (nk <= 0) return;
  la.load(0);
  lb.load(0);
  la.store(lds);
  lb.store(lds + T::A_BYTES);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = lds + (kt & 1) * T::STAGE_BYTES;
    char* nxt = lds + ((kt + 1) & 1) * T::STAGE_BYTES;
    const bool more = kt + 1 < nk;
#ifdef MMF_DIAG_NOLOAD        /* diagnostic builds (tools/diag_build.py): timing only, results are wrong */
    const bool stage = false;
#else
    const bool stage = more;
#endif
    if (stage) { la.load(kt + 1); lb.load(kt + 1); }
#ifndef MMF_DIAG_NOMFMA
    tn_compute_chunk<T>(cur, cur + T::A_BYTES, acc, wm, wn, lane);
#endif
    if (stage) { la.store(nxt); lb.store(nxt + T::A_BYTES); }
    __syncthreads();
  }
}

// reduce per-thread 8-column sums over the NT/(COLS/8) threads that own the same 8 of COLS columns
template <int COLS, int NT>
__device__ inline void colsum8_reduce_store(float* lds, const float (&v)[8], float* dst, int col0, int ncols) {
  constexpr int PPR = COLS / 8, GROUPS = NT / PPR;
  const int tid = threadIdx.x;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) lds[(tid / PPR) * COLS + 8 * (tid % PPR) + e] = v[e];
  __syncthreads();
  if (tid < COLS) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < GROUPS; ++q) s += lds[q * COLS + tid];
    if (col0 + tid < ncols) dst[col0 + tid] = s;
  }
}

template <class T, class RowMap>
__device__ inline void tnb_store(const TnBfProblem& q, int split, int tn, f32x16 (&acc)[T::MB][T::NB], float* lds, RowMap&& rowmap) {
  float* out = q.out + (size_t)split * q.split_stride;
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

template <class T>
__global__ __launch_bounds__(T::NT) void tn_bf16_kernel(TnBfParams p) {
  extern __shared__ __align__(16) char ldsb[];
  const int b = blockIdx.x;
  int split, tg;
  if (p.xcd_map) {       // blocks b, b+8, ... share an XCD (round-robin dispatch): all tiles of one K split on one XCD,
    const int xcd = b & 7, idx = b >> 3;   // so du / h / dP rows are fetched from HBM once and re-read through that L2
    split = xcd + 8 * (idx / p.total_tiles);
    tg = idx % p.total_tiles;
  } else {
    split = b / p.total_tiles;
    tg = b - split * p.total_tiles;
  }
  int pi = 0;
  for (int i = 1; i < p.nprob; ++i)
    if (tg >= p.prob[i].block_begin) pi = i;
  const TnBfProblem& q = p.prob[pi];
  const int t = tg - q.block_begin;
  const int tm = t / q.tiles_n, tn = t - tm * q.tiles_n;
  const int64_t kb64 = (int64_t)split * p.k_per_split;
  const int kbase = (int)(kb64 < p.K ? kb64 : p.K);
  const int kmax = (int)((kb64 + p.k_per_split) < p.K ? (kb64 + p.k_per_split) : p.K);
  const int nk = (kmax - kbase + TNB_KCH - 1) / TNB_KCH;
  const bool do_sum = tn == 0 && q.colsum != nullptr;
  BST_BEGIN();

  LoadTPlain<T::BN, T::NT, T::B_PITCH> lb;
  lb.init(q.B, q.ldb, tn * T::BN, q.Ncols, kbase, kmax, false);
  {
    LoadTPlain<T::BM, T::NT, T::A_PITCH> la;
    la.init(q.A, q.lda, tm * T::BM, q.M, kbase, kmax, do_sum);
    f32x16 acc[T::MB][T::NB];
    tn_mainloop<T>(la, lb, nk, ldsb, acc);
    BST_MARK(BST_TN, 1);
    float* fl = reinterpret_cast<float*>(ldsb);
    tnb_store<T>(q, split, tn, acc, fl, [&](int r) { const int row = tm * T::BM + r; return row < q.M ? row : -1; });
    if (do_sum) colsum8_reduce_store<T::BM, T::NT>(fl, la.cs, q.colsum + (size_t)split * q.colsum_stride, tm * T::BM, q.M);
  }
  BST_MARK(BST_TN, 2);
  BST_COUNT(BST_TN);
}

int tn_bf16_splits(int64_t K, int total_tiles) {
  static const int env = tune_int("MMF_BF16_TN_SPLITS", 0);   // tuning override
  int splits = 256 / (total_tiles > 0 ? total_tiles : 1);
  if (splits >= 8) splits &= ~7;           // whole splits per XCD (launch_tn_bf16's block map)
  if (env > 0) splits = env;
  const int64_t max_splits = (K + 255) / 256;
  if (splits > max_splits) splits = (int)max_splits;
  return splits < 1 ? 1 : splits;
}

int launch_tn_bf16(TnBfParams p, hipStream_t st) {
  using T = TileT<TNB_TILE, TNB_TILE>;
  if (p.k_per_split % TNB_KCH != 0 || p.splits < 1) return MMF_ERR_ARG;
  int blocks = 0;
  for (int i = 0; i < p.nprob; ++i) {
    TnBfProblem& q = p.prob[i];
    if (q.Ncols % 8 != 0 || q.ldb % 8 != 0 || q.M % 8 != 0 || q.lda % 8 != 0) return MMF_ERR_SHAPE;
    q.tiles_m = (q.M + T::BM - 1) / T::BM;
    q.tiles_n = (q.Ncols + T::BN - 1) / T::BN;
    q.block_begin = blocks;
    blocks += q.tiles_m * q.tiles_n;
  }
  if (blocks == 0) return MMF_OK;
  p.total_tiles = blocks;
  static const int env_xcd = tune_int("MMF_BF16_TN_XCD", 1);   // tuning override
  p.xcd_map = env_xcd && p.splits % 8 == 0;
  return launch_tiled_b<T>("tn_bf16_kernel", tn_bf16_kernel<T>, p, p.splits * blocks, T::LDS_BYTES, st);
}

}  // namespace mmf
