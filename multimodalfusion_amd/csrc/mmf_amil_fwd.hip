// Forward kernels of the attention-MIL stack on MI355X (gfx950).
//
//   K-lin   y = drop(act(x.W^T + b))                 models/model_attention_mil_path.py:20-21 (Linear+ReLU+Dropout)
//                                                    models/model_attention_mil_radio.py:80-82 (cat + reduce_dim)
//   K-gate  a = tanh(h.Wa^T+ba), b = sigmoid(h.Wb^T+bb), s_part = sum_d a.b.Wc   models/model_modules.py:105-110 (gated)
//           a = tanh(h.Wa^T+ba),                          s_part = sum_d a.Wc     models/model_modules.py:73-85  (ungated)
//   K-pool  A_raw = sum(s_part)+bc ; per-group online-softmax partials (max, sum e, sum e.h)
//   K-merge M = softmax(A_raw).h, (max, denom)        models/model_attention_mil_path.py:53-56
#include <cstdlib>

#define MMF_STAMP_FIRST_STAGE      /* stamps builds: this unit owns g_stamps[2..3] for the main loops' first stage */
#include "mmf_gemm_core.h"
#include "mmf_gemm_split.h"
#include "mmf_kernels.h"

namespace mmf {

// =============================================================================================
// K-lin : NT GEMM + bias + activation + dropout
// =============================================================================================
__device__ inline float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_TANH: return fast_tanh(v);
    case ACT_SIGMOID: return fast_sigmoid(v);
    case ACT_SELU: {
      const float alpha = 1.6732632423543772f, scale = 1.0507009873554805f;
      return scale * (v > 0.f ? v : alpha * (__expf(v) - 1.0f));
    }
    default: return v;
  }
}

// Epilogue of K-lin.  The activation and the dropout switch are compile-time: with run-time `switch (p.act)` /
// `if (p.drop_p > 0)` inside the element loop the tile's 448 elements per lane each went through three scalar
// branches (every taken branch refills the instruction buffer); the kernel dispatches ONCE per workgroup instead.
// ACT = -1: any activation, decided per element (the rarely used tanh / sigmoid / SELU projections).
// K-split launches: where the last-arriving workgroup of a tile finds the tile's partial sums (LinearParams::kpart)
struct PartSrc {
  unsigned bytes;       // size of the whole kpart buffer (the buffer resource is rebuilt where it is used: kept in the
                        // struct it lived in scratch)
  unsigned base;        // byte offset of this tile's first partial
  unsigned slab;        // bytes per partial tile (BM x BN floats)
  int S;                // partials per tile
};

template <class T, int ACT, bool DROP>
__device__ inline void linear_epilogue(const LinearParams& p, f32x16 (&acc)[T::MB][T::NB], f32x4acc (*acch)[2], float* lds,
                                       int row0, int col0, const PartSrc ps = PartSrc{}) {
#ifdef MMF_DIAG_NOEPI         /* diagnostic build: main loop only (results are wrong) */
  {
    float t = 0.f;
    for_each_c<T>(acc, [&](int, int, float v) { t += v; });
    if (t == 1.2345e30f) p.y[0] = t;
    return;
  }
#endif
  const uint32_t thr = drop_threshold(p.drop_p);
  const float scale = DROP ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  const uint32_t dkey = p.drop_key + (p.seed_dev ? *p.seed_dev : 0u);   // device-resident part of the seed, if any
  float4 bias4[T::NB];                       // per column strip, loaded once, before any store
#pragma unroll
  for (int nb = 0; nb < T::NB; ++nb) {
    const int col = col0 + epilogue_col<T>(nb);
    bias4[nb] = (p.bias && col < p.N) ? ld4(p.bias + col) : zero4();
  }
  const int lane = threadIdx.x & 63;
  // one transposed block: NT row groups (rows r + 8 t) of 4 consecutive columns per lane; NT = 4 for a 32-row block,
  // 2 for the 16-row half block
  auto rows_op = [&](auto nt_c, int nb, int r, int c, const float4* v) {
    constexpr int NTR = decltype(nt_c)::value;
    const int col = col0 + c;
    const bool col_ok = col < p.N;           // N % 4 == 0: a float4 never straddles the edge
    const float4 b4 = bias4[nb];
    unsigned long long mine = 0ull;          // lane 4 t + e keeps ballot (t, e) of this block (LinearParams::relu_bits)
#pragma unroll
    for (int t = 0; t < NTR; ++t) {
      const int row = row0 + r + 8 * t;
      const bool ok = col_ok && row < p.M;
      float y[4] = {v[t].x + b4.x, v[t].y + b4.y, v[t].z + b4.z, v[t].w + b4.w};
      const uint32_t idx = (uint32_t)row * (uint32_t)p.N + (uint32_t)col;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (ACT == ACT_RELU) y[e] = fmaxf(y[e], 0.f);
        else if constexpr (ACT < 0) y[e] = apply_act(y[e], p.act);
        if constexpr (DROP) y[e] = keep(dkey, idx + e, thr) ? y[e] * scale : 0.f;
      }
      // (Tried: taking these ballots in K-gate's A loader instead, where h streams through staging registers anyway.
      // The projection got 2 us back, K-gate lost 11: its first-column tiles became the launch's critical path.)
      if (p.relu_bits) {                     // wave-uniform; the ballots are taken by every lane, valid or not
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned long long bal = __ballot(ok && y[e] > 0.f);
          if (lane == 4 * t + e) mine = bal;
        }
      }
#ifdef MMF_DIAG_NOSTORE       /* diagnostic build: everything but the global stores (results are wrong) */
      if (y[0] == 1.2345e30f)
#endif
      if (ok) st4(p.y + (size_t)row * p.N + col, make_float4(y[0], y[1], y[2], y[3]));
    }
    if (p.relu_bits && lane < 4 * NTR) {
      // 16-row granularity: rows r0 + rr + 8 t with t = 0, 1 are bit block r0 / 16, t = 2, 3 bit block r0 / 16 + 1
      const size_t rb16 = ((size_t)(row0 + r - (lane >> 3)) >> 4) + (lane >> 3), cb = (size_t)(col0 + c - 4 * (lane & 7)) >> 5;
      if ((int64_t)rb16 * 16 < p.M && (int)(cb * 32) < p.N) p.relu_bits[(rb16 * (size_t)(p.N >> 5) + cb) * 8 + (lane & 7)] = mine;
    }
  };
  if (ps.S > 0) {
    // the tile's sum comes from memory, already row-major: partial s of the tile, rows r + 8 t, columns c .. c + 3 of this
    // lane -- added in split order (bit-reproducible); the loads of block b + 1 are issued before block b's stores
    const int wave = threadIdx.x >> 6, wm = wave / T::WN, wn = wave % T::WN;
    const int rr = lane >> 3, c4 = lane & 7;
    constexpr int NBLK = T::MB * T::NB;
    const rsrc_t prs = make_rsrc(p.kpart, ps.bytes);
    float4 buf[2][4];
    auto fetch = [&](int b, int rows4, float4 (&v)[4]) {
      const int mb = b / T::NB, nb = b % T::NB;
      const int r = wm * (T::BM / T::WM) + mb * 32 + rr, c = (wn * T::NB + nb) * 32 + 4 * c4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t >= rows4) break;
        const unsigned o = ps.base + (unsigned)((r + 8 * t) * T::BN + c) * 4u;
        float4 a = bld4_dev(prs, o, 0);
        for (int k = 1; k < ps.S; ++k) {
          const float4 q = bld4_dev(prs, o, (unsigned)k * ps.slab);
          a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
        }
        v[t] = a;
      }
    };
    fetch(0, 4, buf[0]);
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const int mb = b / T::NB, nb = b % T::NB;
      if (b + 1 < NBLK) fetch(b + 1, 4, buf[(b + 1) & 1]);
      else if constexpr (T::HALF) fetch(T::MB * T::NB, 2, buf[(b + 1) & 1]);
      rows_op(std::integral_constant<int, 4>{}, nb, wm * (T::BM / T::WM) + mb * 32 + rr, (wn * T::NB + nb) * 32 + 4 * c4, buf[b & 1]);
    }
    if constexpr (T::HALF) {
#pragma unroll
      for (int nb = 0; nb < T::NB; ++nb) {
        if (nb + 1 < T::NB) fetch(T::MB * T::NB + nb + 1, 2, buf[(NBLK + nb + 1) & 1]);
        rows_op(std::integral_constant<int, 2>{}, nb, wm * (T::BM / T::WM) + T::MB * 32 + rr, (wn * T::NB + nb) * 32 + 4 * c4,
                buf[(NBLK + nb) & 1]);
      }
    }
    return;
  }
  epilogue_rows<T>(acc, lds, [&](int mb, int nb, int r, int c, const float4 (&v)[4]) {
    rows_op(std::integral_constant<int, 4>{}, nb, r, c, v);
  });
  if constexpr (T::HALF) {
    const int wave = threadIdx.x >> 6, wm = wave / T::WN, wn = wave % T::WN;
    float* blk = lds + wave * (32 * EPI_STRIDE);
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb) {
      float4 v[2];
      transpose_half(acch[nb][0], acch[nb][1], blk, lane, v);
      rows_op(std::integral_constant<int, 2>{}, nb, wm * (T::BM / T::WM) + T::MB * 32 + (lane >> 3),
              (wn * T::NB + nb) * 32 + 4 * (lane & 7), v);
    }
  }
}

// K-split: this workgroup's accumulators -> its partial tile (device-scope stores), then a ticket.  Returns true for the
// workgroup that arrived LAST at the tile (all partials are in memory then): it runs the epilogue from the partials.
template <class T>
__device__ inline bool ksplit_publish(const LinearParams& p, f32x16 (&acc)[T::MB][T::NB], f32x4acc (*acch)[2], float* lds,
                                      int tile, int ks, PartSrc& ps) {
  const int S = p.ksplit;
  ps.bytes = (unsigned)((size_t)p.mt_count * p.nt_count * S * T::BM * T::BN * 4u);
  const rsrc_t prs = make_rsrc(p.kpart, ps.bytes);
  ps.slab = (unsigned)(T::BM * T::BN * 4);
  ps.base = (unsigned)tile * (unsigned)S * ps.slab;
  ps.S = S;
  const unsigned mine = ps.base + (unsigned)ks * ps.slab;
  epilogue_rows<T>(acc, lds, [&](int mb, int nb, int r, int c, const float4 (&v)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) bst4_dev(prs, mine + (unsigned)((r + 8 * t) * T::BN + c) * 4u, 0, v[t]);
  });
  if constexpr (T::HALF) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / T::WN, wn = wave % T::WN;
    float* blk = lds + wave * (32 * EPI_STRIDE);
#pragma unroll
    for (int nb = 0; nb < T::NB; ++nb) {
      float4 v[2];
      transpose_half(acch[nb][0], acch[nb][1], blk, lane, v);
      const int r = wm * (T::BM / T::WM) + T::MB * 32 + (lane >> 3), c = (wn * T::NB + nb) * 32 + 4 * (lane & 7);
#pragma unroll
      for (int t = 0; t < 2; ++t) bst4_dev(prs, mine + (unsigned)((r + 8 * t) * T::BN + c) * 4u, 0, v[t]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's partial rows have reached device scope
  __syncthreads();                                        // ... and every wave's
  unsigned* flag = reinterpret_cast<unsigned*>(lds + (T::NT / 64) * 32 * EPI_STRIDE);     // behind the transpose scratch
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(p.ktick + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)(S - 1)) __hip_atomic_store(p.ktick + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left zero for the next launch
    flag[0] = old;
  }
  __syncthreads();
  return flag[0] == (unsigned)(S - 1);
}

template <class T>
__global__ __launch_bounds__(T::NT) void linear_nt_kernel(LinearParams p) {
  extern __shared__ __align__(16) float lds[];
  MMF_KSTAMP(kernel_t0);
  const int S = p.ksplit > 1 ? p.ksplit : 1;
  int mt, ntk;
  // the (column tile, K split) pairs of one row tile are 8 workgroups apart: one XCD, whose L2 serves the shared x rows
  if (!tile_of_block(blockIdx.x, p.mt_count, p.nt_count * S, mt, ntk)) return;
  const int nt = ntk % p.nt_count, ks = ntk / p.nt_count;
  const int row0 = mt * T::BM, col0 = nt * T::BN;
  const int nk = p.K / KC / S;

  LoadK<T::BM, T::NT> la;
  la.init_segments(p.x, p.nseg, p.kseg, p.ldx, row0, (int)p.M);
  LoadK<T::BN, T::NT> lb;
  lb.init(p.w, p.K, col0, p.N);
  la.kt0 = lb.kt0 = ks * nk;

  f32x16 acc[T::MB][T::NB];
  f32x4acc acch[T::NB][2];                   // the half block's accumulators (Tile::HALF; unused otherwise)
  MMF_KSTAMP(k0);
  if constexpr (T::NT == 256 && T::BM <= 64) {
    if (p.deep) gemm_mainloop_deep<T, 4>(la, lb, nk, lds, acc);      // short grid: see gemm_mainloop_deep
    else gemm_mainloop<T>(la, lb, nk, lds, acc);
  } else {
    gemm_mainloop<T>(la, lb, nk, lds, acc, acch);
  }
  MMF_KSTAMP(k1);
#ifdef MMF_STAMPS
  if ((threadIdx.x & 63) == 0) atomicAdd(&g_stamps[4], k0 - kernel_t0);     // entry -> loaders initialised
#endif
  PartSrc ps{};                              // S = 0: the epilogue takes the tile from the accumulators
  if (S > 1) {
    if (!ksplit_publish<T>(p, acc, acch, lds, mt * p.nt_count + nt, ks, ps)) return;
  }
  const bool drop = p.drop_p > 0.f;
  if (p.act == ACT_RELU) {
    if (drop) linear_epilogue<T, ACT_RELU, true>(p, acc, acch, lds, row0, col0, ps);
    else linear_epilogue<T, ACT_RELU, false>(p, acc, acch, lds, row0, col0, ps);
  } else if (p.act == ACT_NONE && !drop) {
    linear_epilogue<T, ACT_NONE, false>(p, acc, acch, lds, row0, col0, ps);
  } else if (drop) {
    linear_epilogue<T, -1, true>(p, acc, acch, lds, row0, col0, ps);
  } else {
    linear_epilogue<T, -1, false>(p, acc, acch, lds, row0, col0, ps);
  }
#ifdef MMF_STAMPS
  MMF_KSTAMP(k2);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&g_stamps[5], k1 - k0); atomicAdd(&g_stamps[6], k2 - k1); atomicAdd(&g_stamps[7], 1ull);
  }
#endif
}

// The same projection on the bf16 matrix cores (mmf_gemm_split.h: 3-way operand split, fp32-equivalent accuracy).
template <class T>
__global__ __launch_bounds__(T::NT) void linear_nt_split_kernel(LinearParams p) {
  extern __shared__ __align__(16) float lds[];
  int mt, nt;
  if (!tile_of_block(blockIdx.x, p.mt_count, p.nt_count, mt, nt)) return;
  const int row0 = mt * T::BM, col0 = nt * T::BN;
  SplitK<T::BM, T::NT> la;
  la.init(p.x[0], p.ldx, row0, (int)p.M);
  SplitK<T::BN, T::NT> lb;
  lb.init(p.w, p.K, col0, p.N);
  f32x16 acc[T::MB][T::NB];
  split_mainloop<T, 4, decltype(la), decltype(lb), true>(la, lb, p.K / SKC, lds, acc);
  const bool drop = p.drop_p > 0.f;
  if (p.act == ACT_RELU) {
    if (drop) linear_epilogue<T, ACT_RELU, true>(p, acc, nullptr, lds, row0, col0);
    else linear_epilogue<T, ACT_RELU, false>(p, acc, nullptr, lds, row0, col0);
  } else if (p.act == ACT_NONE && !drop) {
    linear_epilogue<T, ACT_NONE, false>(p, acc, nullptr, lds, row0, col0);
  } else if (drop) {
    linear_epilogue<T, -1, true>(p, acc, nullptr, lds, row0, col0);
  } else {
    linear_epilogue<T, -1, false>(p, acc, nullptr, lds, row0, col0);
  }
}

// =============================================================================================
// K-gate : NT GEMM of h against an interleaved [Wa-block | Wb-block] tile, fused gate epilogue
// =============================================================================================
// B-operand loader: tile row j -> 32-row block jb = j/32.  Gated: blocks alternate a, b for the
// same 32 attention dims, so a wave's (nb = 2t, 2t+1) accumulators hold the tanh- and the
// sigmoid-branch pre-activations of the SAME (instance, d) in the SAME lane and register.
// HALVES = false: 32-row blocks alternate (a, b) for the same 32 attention dims (2x2-wave tiles: a wave's
//                 nb = 2t, 2t+1 accumulators are the tanh and sigmoid branch of the same (instance, d));
// HALVES = true : rows [0, ROWS/2) are the a-branch, [ROWS/2, ROWS) the b-branch of dims d0 .. d0+ROWS/2-1
//                 (1x8-wave tiles: waves 0-3 hold a, waves 4-7 hold b; they meet through LDS in the epilogue).
template <int ROWS, int NT, bool GATED, bool HALVES>
struct LoadGateW {
  using Map = KMap<ROWS, NT>;
  rsrc_t ra, rb;
  int tid;
  int which[Map::NV];          // wave-uniform (a wave-instruction covers 8 consecutive tile rows)
  unsigned voff[Map::NV];
  float4 r[Map::NV];
  __device__ inline void init(const float* wa, const float* wb, int H, int D, int d0) {
    tid = threadIdx.x;
    ra = make_rsrc(wa, (unsigned)D * (unsigned)H * 4u);
    rb = make_rsrc(GATED ? wb : wa, (unsigned)D * (unsigned)H * 4u);
#pragma unroll
    for (int i = 0; i < Map::NV; ++i) {
      const int j = Map::row(tid, i);
      int w, d;
      if (!GATED) { w = 0; d = d0 + j; }
      else if (HALVES) { w = j >= ROWS / 2 ? 1 : 0; d = d0 + j - w * (ROWS / 2); }
      else { w = (j >> 5) & 1; d = d0 + (j >> 6) * 32 + (j & 31); }
      which[i] = __builtin_amdgcn_readfirstlane(w);
      voff[i] = (Map::valid(tid, i) && d < D) ? ((unsigned)d * (unsigned)H + 4u * Map::c4(tid, i)) * 4u : OOB;
    }
  }
  __device__ inline void load(int kt) {
    const unsigned soff = (unsigned)(kt * KC) * 4u;
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
__device__ inline void gate_fwd_tile(const GateFwdParams& p, float* lds, int row0, int nt);

template <class T, bool GATED>
__global__ __launch_bounds__(T::NT) void gate_fwd_kernel(GateFwdParams p) {
  extern __shared__ __align__(16) float lds[];
  int mt, nt;
  if (!tile_of_block(blockIdx.x, p.mt_count, p.nt_count, mt, nt)) return;
  gate_fwd_tile<T, GATED>(p, lds, (int)p.row_begin + mt * T::BM, nt);
}

// Two tile heights in ONE launch: the bag's rows are cut into regions of tall (TB) or short (TS) tiles, taken by
// consecutive ranges of workgroups (GateFwdParams::reg; launch_gate_fwd plans them).  TB and TS have the same thread
// count and column width.
template <class TB, class TS, bool GATED>
__global__ __launch_bounds__(TB::NT) void gate_fwd_mixed_kernel(GateFwdParams p) {
  static_assert(TB::NT == TS::NT && TB::BN == TS::BN, "mixed tiles share the launch shape");
  extern __shared__ __align__(16) float lds[];
  int r = 0;
  for (int i = 1; i < p.nreg; ++i)
    if ((int)blockIdx.x >= p.reg[i].grid_begin) r = i;
  int mt, nt;
  if (!tile_of_block(blockIdx.x - p.reg[r].grid_begin, p.reg[r].mt_count, p.nt_count, mt, nt)) return;
  if (p.reg[r].tall) gate_fwd_tile<TB, GATED>(p, lds, (int)p.reg[r].row0 + mt * TB::BM, nt);
  else gate_fwd_tile<TS, GATED>(p, lds, (int)p.reg[r].row0 + mt * TS::BM, nt);
}

template <class T, bool GATED>
__device__ inline void gate_fwd_tile(const GateFwdParams& p, float* lds, int row0, int nt) {
  constexpr int DT = GATED ? T::BN / 2 : T::BN;   // attention dims covered by one tile
  const int d0 = nt * DT;

  std::conditional_t<T::SPLIT, SplitK<T::BM, T::NT>, LoadK<T::BM, T::NT>> la;
  la.init(p.h, p.H, row0, (int)p.row_end);
  std::conditional_t<T::SPLIT, SplitGateW<T::BN, T::NT, GATED>, LoadGateW<T::BN, T::NT, GATED, false>> lb;
  lb.init(p.Wa, p.Wb, p.H, p.D, d0);

  f32x16 acc[T::MB][T::NB];
  if constexpr (T::SPLIT) {
    split_mainloop<T, 4>(la, lb, p.H / SKC, lds, acc);
  } else if constexpr (T::BM <= 64) {
    if (p.deep) gemm_mainloop_deep<T, 4>(la, lb, p.H / KC, lds, acc);      // short grid: see gemm_mainloop_deep
    else gemm_mainloop<T>(la, lb, p.H / KC, lds, acc);
  } else {
    gemm_mainloop<T>(la, lb, p.H / KC, lds, acc);
  }

#ifdef MMF_DIAG_NOEPI         /* diagnostic build: main loop only (results are wrong) */
  {
    float t = 0.f;
    for_each_c<T>(acc, [&](int, int, float v) { t += v; });
    if (t == 1.2345e30f) p.s_part[0] = t;
    return;
  }
#endif
  // ---- epilogue (row-major, float4): activations, stores of a / b, per-row partial score ----------------
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int rr = lane >> 3, c4 = lane & 7;
  const uint32_t thr = drop_threshold(p.drop_p);
  const bool drop = p.drop_p > 0.f;
  const float dscale = drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  const uint32_t sdev = p.seed_dev ? *p.seed_dev : 0u;
  const uint32_t key_a = p.key_a + sdev, key_b = p.key_b + sdev;
  float* blk = lds + wave * (32 * EPI_STRIDE);
  float* sred = lds + (T::NT / 64) * (32 * EPI_STRIDE);   // [WN][BM] row partials, behind the transpose scratch

  constexpr int NPAIR = GATED ? T::NB / 2 : T::NB;
  float4 ba_r[NPAIR], bb_r[NPAIR], wc_r[NPAIR];   // per column strip, loaded once, before any store
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
    float rowsum[4] = {0.f, 0.f, 0.f, 0.f};     // rows rr + 8t of this 32-row block, this lane's columns
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
        if (row < p.row_end && dok) {
          const size_t o = (size_t)row * p.D + d;
          if (p.a) {     // null in forward-only (inference) calls: nothing is saved for a backward
            st4(p.a + o, make_float4(av[0], av[1], av[2], av[3]));
            if constexpr (GATED) st4(p.b + o, make_float4(bv[0], bv[1], bv[2], bv[3]));
          }
          const float wc[4] = {wc4.x, wc4.y, wc4.z, wc4.w};
          const uint32_t idx = (uint32_t)row * (uint32_t)p.D + (uint32_t)d;
          if (drop) {       // decided once per row, not per element
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float ad = keep(key_a, idx + e, thr) ? av[e] * dscale : 0.f;
              float bd = bv[e];
              if constexpr (GATED) bd = keep(key_b, idx + e, thr) ? bd * dscale : 0.f;
              rowsum[q] += ad * bd * wc[e];
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) rowsum[q] += av[e] * bv[e] * wc[e];
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {       // the 8 lanes that share a row hold 4 columns each
      float s = rowsum[q];
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      if (c4 == 0) sred[wn * T::BM + (wm * T::MB + mb) * 32 + rr + 8 * q] = s;
    }
  }
  __syncthreads();
  for (int i = tid; i < T::BM; i += T::NT) {
    int row = row0 + i;
    if (row < p.row_end) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < T::WN; ++w) s += sred[w * T::BM + i];
      p.s_part[(size_t)nt * p.N + row] = s;
    }
  }
}

// =============================================================================================
// K-pool : scores + per-group online-softmax partials.  HBM-bound (reads h once).
// =============================================================================================
constexpr int POOL_MAX_ROWS = 8192;

__global__ __launch_bounds__(256) void pool_partial_kernel(PoolParams p) {
  __shared__ float s_lds[POOL_MAX_ROWS];
  __shared__ float red[256];
  __shared__ __align__(16) float vred[1024];   // RG * VPR == 256 float4 slots
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

  const int VPR = p.H / 4;          // float4 per row: 64 (H=256), 128 (H=512), 256 (H=1024)
  const int RG = 256 / VPR;         // row groups processed concurrently
  const int cv = tid % VPR, rg = tid / VPR;
  float4 v = zero4();
  float lsum = 0.f;
  int i = rg;
  for (; i + 7 * RG < nrows; i += 8 * RG) {        // 8 independent row loads in flight (the loop is latency-bound)
    float4 hv[8];
    float e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) hv[u] = ld4(p.h + (size_t)(r0 + i + u * RG) * p.H + 4 * cv);
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = __expf(s_lds[i + u * RG] - m);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v.x += e[u] * hv[u].x; v.y += e[u] * hv[u].y; v.z += e[u] * hv[u].z; v.w += e[u] * hv[u].w;
      if (cv == 0) lsum += e[u];
    }
  }
  for (; i < nrows; i += RG) {
    float e = __expf(s_lds[i] - m);
    float4 hv = ld4(p.h + (size_t)(r0 + i) * p.H + 4 * cv);
    v.x += e * hv.x; v.y += e * hv.y; v.z += e * hv.z; v.w += e * hv.w;
    if (cv == 0) lsum += e;
  }
  st4(vred + (rg * VPR + cv) * 4, v);
  if (cv == 0) red[rg] = lsum;
  __syncthreads();
  float* out = p.partials + (size_t)g * (2 + p.H);
  if (tid < VPR) {
    float4 s = zero4();
    for (int q = 0; q < RG; ++q) {
      float4 t = ld4(vred + (q * VPR + tid) * 4);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    out[2 + 4 * tid + 0] = s.x; out[2 + 4 * tid + 1] = s.y;
    out[2 + 4 * tid + 2] = s.z; out[2 + 4 * tid + 3] = s.w;
  }
  if (tid == 0) {
    float l = 0.f;
    for (int q = 0; q < RG; ++q) l += red[q];
    out[0] = nrows > 0 ? m : -INFINITY;
    out[1] = l;
  }
}

// Head tail (HeadTail; head_tail_kernel below): the classifier, the hazards and -- when a label is given -- nll_surv
// with its backward down to dM, on 1024 threads.  Everything it needs that does not depend on M (classifier rows /
// columns, bias, label, censorship) is requested at kernel entry (TailPre): the tail is a chain of dependent steps,
// and each global load left inside it costs a memory round trip.
struct TailPre {
  float wk_row[16];   // wave k < K: Wk[k][lane + 64 j]        (H <= 1024)
  float wk_col[8];    // thread c < H: Wk[k][c], k < min(K, 8)
  float bk;           // lane 0 of wave k
  float c;            // thread 0
  long long y;        // thread 0
};
__device__ inline void tail_preload(const PoolParams& p, TailPre& r) {
  const HeadTail& t = p.tail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < 16; ++j) r.wk_row[j] = (wave < t.K && lane + 64 * j < p.H) ? t.Wk[(size_t)wave * p.H + lane + 64 * j] : 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) r.wk_col[k] = (k < t.K && tid < p.H) ? t.Wk[(size_t)k * p.H + tid] : 0.f;
  r.bk = (wave < t.K && lane == 0) ? t.bk[wave] : 0.f;
  r.c = (tid == 0 && t.c) ? t.c[0] : 0.f;
  r.y = (tid == 0 && t.Y) ? (long long)t.Y[0] : 0;
}
__device__ inline void head_tail(const PoolParams& p, const TailPre& r, float* sm /* 1024 + 160 floats */) {
  const HeadTail& t = p.tail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, K = t.K;
  float* Ml = sm;                 // [H]
  float* z = sm + 1024;           // [K] logits, then dz
  float *hz = z + 32, *S = z + 64, *gH = z + 96, *gS = z + 128;   // K <= 32; in LDS: indexed arrays in registers would go to scratch
  if (!p.merge_in_tail)           // M comes from the K-merge launch in front (else the caller has filled Ml)
    for (int c = tid; c < H; c += 1024) Ml[c] = p.M[c];
  __syncthreads();
  if (wave < K) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (lane + 64 * j < H) acc += Ml[lane + 64 * j] * r.wk_row[j];
    acc = wave_sum(acc);
    if (lane == 0) z[wave] = acc + r.bk;
  }
  for (int k = 16 + wave; k < K; k += 16) {          // K > 16: the rows that were not preloaded
    float acc = 0.f;
    for (int c = lane; c < H; c += 64) acc += Ml[c] * t.Wk[(size_t)k * H + c];
    acc = wave_sum(acc);
    if (lane == 0) z[k] = acc + t.bk[k];
  }
  __syncthreads();
  if (tid == 0) {
    // forward (surv_head_fwd_kernel) ...
    float run = 1.f, best = -INFINITY, ssum = 0.f;
    int arg = 0;
    for (int k = 0; k < K; ++k) {
      const float zz = z[k];
      hz[k] = 1.0f / (1.0f + expf(-zz));
      run *= (1.0f - hz[k]);
      S[k] = run;
      ssum += run;
      t.logits[k] = zz; t.hazards[k] = hz[k]; t.S[k] = run;
      if (zz > best) { best = zz; arg = k; }
    }
    t.Y_hat[0] = arg;
    if (t.risk) t.risk[0] = -ssum;
    if (t.Y) {
      // ... nll_surv for the one sample (nll_surv_kernel) ...
      for (int k = 0; k < K; ++k) { gH[k] = 0.f; gS[k] = 0.f; }
      const long long y64 = r.y;
      float l;
      if (y64 < 0 || y64 >= K) {
        l = __builtin_nanf("");
      } else {
        const int y = (int)y64;
        const float c = r.c;
        const float sp_y = y == 0 ? 1.0f : S[y - 1];
        const float hy = hz[y];
        const float unc = -(1.f - c) * (logf(fmaxf(sp_y, t.eps)) + logf(fmaxf(hy, t.eps)));
        if (y > 0 && sp_y >= t.eps) gS[y - 1] += -(1.f - c) / sp_y;
        if (hy >= t.eps) gH[y] += -(1.f - c) / hy;
        const float sp_y1 = S[y];
        const float cen = -c * logf(fmaxf(sp_y1, t.eps));
        if (sp_y1 >= t.eps) gS[y] += -(1.f - t.alpha) * c / sp_y1;
        l = (1.f - t.alpha) * (cen + unc) + t.alpha * unc;
      }
      t.loss[0] = l;
      // ... and the head's backward (surv_head_bwd_kernel): dz_t = (gH_t - sum_{j>=t} gS_j prod_{u<=j,u!=t}(1-h_u)) h_t (1-h_t)
      for (int k = 0; k < K; ++k) {
        float g = gH[k];
        for (int j = k; j < K; ++j) {
          float prod = 1.f;
          for (int u = 0; u <= j; ++u)
            if (u != k) prod *= (1.0f - hz[u]);
          g -= gS[j] * prod;
        }
        z[k] = g * hz[k] * (1.0f - hz[k]) * t.loss_scale;
      }
    }
  }
  if (!t.Y) return;
  __syncthreads();
  for (int c = tid; c < H; c += 1024) {
    float acc = 0.f;
    if (c == tid) {                   // the preloaded columns (every c when H <= 1024)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < K) acc += z[k] * r.wk_col[k];
      for (int k = 8; k < K; ++k) acc += z[k] * t.Wk[(size_t)k * H + c];
    } else {
      for (int k = 0; k < K; ++k) acc += z[k] * t.Wk[(size_t)k * H + c];
    }
    t.dM[c] = acc;
    for (int k = 0; k < K; ++k) {
      float* o = t.dWk + (size_t)k * H + c;
      const float v = z[k] * Ml[c];
      *o = t.accumulate ? *o + v : v;
    }
  }
  if (tid < K) t.dbk[tid] = t.accumulate ? t.dbk[tid] + z[tid] : z[tid];
}

// H/32 workgroups of 1024 threads: merge the per-group partials.
// Group weights exp(m_g - m) are computed once into LDS; the column sums then run as independent, unrolled
// loads (the first version's serial dependent loop over groups cost 115 us).
constexpr int MERGE_MAX_GROUPS = 4096;
__global__ __launch_bounds__(1024) void pool_merge_kernel(PoolParams p) {
  __shared__ float wl[MERGE_MAX_GROUPS];
  __shared__ float red[32];
  __shared__ float colred[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int stride = 2 + p.H;
  float m = -INFINITY;
  for (int g = tid; g < p.n_groups; g += 1024) {
    float mg = p.partials[(size_t)g * stride];
    wl[g] = mg;
    m = fmaxf(m, mg);
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
  float l = 0.f;
  for (int g = tid; g < p.n_groups; g += 1024) {
    float mg = wl[g];
    float w = mg > -INFINITY ? __expf(mg - m) : 0.f;
    wl[g] = w;
    l += p.partials[(size_t)g * stride + 1] * w;
  }
  l = wave_sum(l);
  if (lane == 0) red[16 + wave] = l;
  __syncthreads();
  l = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) l += red[16 + i];
  // Column sums: workgroup b owns columns 32 b .. 32 b + 31 (a single workgroup reading all n_groups x H partials
  // was bound by what one CU can pull: 264 KB = 10 us at 256 groups); its 1024 threads are 32 columns x 32 group
  // slices of independent loads.  m and l above are recomputed by every workgroup (2 floats per group).
  const int cl = tid & 31, sl = tid >> 5;
  const int c = blockIdx.x * 32 + cl;
  float acc = 0.f;
  const float* q = p.partials + 2 + c;
#pragma unroll 8
  for (int g = sl; g < p.n_groups; g += 32) acc += q[(size_t)g * stride] * wl[g];
  colred[tid] = acc;
  __syncthreads();
  if (tid < 32) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += colred[i * 32 + tid];
    p.M[c] = s / l;
  }
  if (blockIdx.x == 0 && tid == 0) { p.stats[0] = m; p.stats[1] = l; }
}

// The head tail as its own single-workgroup launch behind K-merge (HeadTail).  It can also merge up to TAIL_MERGE_MAX
// partials itself (thread c sums column c over the groups: one launch less) -- off by default, see launch_pool_merge.
// (First version: tail run by the LAST workgroup of K-merge, found with a ticket counter -- the fences, the atomic and
// the re-read of M through L2 cost 12 us, more than the 5 us of this launch.)
constexpr int TAIL_MERGE_MAX = 64;
__global__ __launch_bounds__(1024) void head_tail_kernel(PoolParams p) {
  __shared__ float tail_sm[1024 + 160];
  __shared__ float wl[TAIL_MERGE_MAX];
  __shared__ float ml[2];
  TailPre pre;
  tail_preload(p, pre);
  const int tid = threadIdx.x;
  if (p.merge_in_tail) {
    const int stride = 2 + p.H;
    if (tid < 64) {                                   // one wave: group maxima, weights exp(m_g - m), denominator
      const float mg = tid < p.n_groups ? p.partials[(size_t)tid * stride] : -INFINITY;
      const float m = wave_max(mg);
      const float w = mg > -INFINITY ? __expf(mg - m) : 0.f;
      const float l = wave_sum(tid < p.n_groups ? p.partials[(size_t)tid * stride + 1] * w : 0.f);
      wl[tid] = w;
      if (tid == 0) { ml[0] = m; ml[1] = l; p.stats[0] = m; p.stats[1] = l; }
    }
    __syncthreads();
    if (tid < p.H) {
      // the same order of additions as pool_merge_kernel (32 interleaved group slices, then the slices in order): the
      // forward-only path merges there, and the two must agree to the bit (tests/test_gpu_infer.py)
      const float* q = p.partials + 2 + tid;
      float acc = 0.f;
      for (int sl = 0; sl < 32; ++sl) {
        float a = 0.f;
        for (int g = sl; g < p.n_groups; g += 32) a += q[(size_t)g * stride] * wl[g];
        acc += a;
      }
      const float mv = acc / ml[1];
      p.M[tid] = mv;
      tail_sm[tid] = mv;
    }
  }
  head_tail(p, pre, tail_sm);
}

// A[i] = sum_t s_part[t][i] + bc : the scores alone (standalone Attn_Net / Attn_Net_Gated forward, no pooling)
__global__ __launch_bounds__(256) void score_sum_kernel(const float* s_part, int n_parts, const float* bc, float* A, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float s = bc ? bc[0] : 0.f;
  for (int t = 0; t < n_parts; ++t) s += s_part[(size_t)t * N + i];
  A[i] = s;
}
int launch_score_sum(const float* s_part, int n_parts, const float* bc, float* A, int64_t N, hipStream_t st) {
  { ProfScope ps("score_sum_kernel", st);
    hipLaunchKernelGGL(score_sum_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, s_part, n_parts, bc, A, N); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
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

// a grid this short leaves every workgroup alone on its CU: nothing hides a memory round trip but deeper prefetch
static inline bool short_grid(int64_t workgroups) {
  static const int env = tune_int("MMF_DEEP", 1);     // A/B switch
  static const int cap = tune_int("MMF_DEEP_MAX", 1024);   // tuning override (10k bag, 628 workgroups: 252 -> 242 us per step)
  return env && workgroups <= cap;
}

using TileNT128 = Tile<128, 128, 2, 2, true, true>;
using TileNT64 = Tile<64, 64, 2, 2, true, true>;
using TileNT32 = Tile<32, 64, 1, 2, true, true>;     // two waves: fills the chip where 64 x 64 tiles make <= 128 workgroups

// rows-per-tile choice: big tiles once there is enough work to fill 256 CUs twice over
static inline bool use_big_tiles(int64_t M, int N) { return (M / 128) * ((N + 127) / 128) >= 256; }

// Wide tiles: (32*MB) x 256, ONE 8-wave workgroup per CU (wave w owns the 32-column strip w of every row).
// Measured (profiles/r01/load_rate.txt): a CU pulls ~11.5 B/clk of streamed and ~29 B/clk of L2-hot operand
// into LDS; two 128x128 workgroups per CU need 64 KB per 8192 MFMA cycles and stall on it.  A 224x256 tile
// needs 60 KB per 14336 MFMA cycles (60 FLOP/B instead of 32) and, at N = 50k, gives exactly 224 tiles.
template <int ROWS>
using TileW = Tile<ROWS, 256, 1, 8, true, true>;

// tile height that minimises rounds x blocks over the 256 CUs (0.35: per-tile prologue / epilogue, in 32-row blocks).
// Heights are the multiples of 16 from 64 to 240 rows; an odd multiple ends with a 16-row half block (Tile::HALF):
// 208 rows put a 50k bag on 241 CUs instead of 224, 96 rows a 24k bag on 250 instead of 188 (128 rows), 160 rows a 40k
// bag on 250 instead of 209 (192), 240 rows a 60k bag on 250 in ONE round instead of 289 tiles of 208 rows in two.
// 256 rows need > 256 VGPRs with double-buffered fragments (spills).
// concurrent (mmf_amil_desc::concurrent: other bags' kernels run beside this one): plan for 224 of the 256 CUs, which
// keeps the measured choice at 50k (224 tiles of 224 rows; 1376 vs 1337 bags/s with 208-row tiles and bags in flight)
// and, unlike the former "no half blocks when concurrent", does not push a 60k bag into two rounds of 224-row tiles.
int pick_wide_rows(int64_t M, int ntn, bool allow_half, bool concurrent, int max_rows) {
  static const int env = tune_int("MMF_WIDE_ROWS", 0);   // tuning override
  if (env > 0 && (env % 32 == 0 || allow_half)) return env;
  const int cus = concurrent ? 224 : 256;
  int best = 224;
  double bestc = 1e30;
  for (int rows = max_rows; rows >= 64; rows -= 16) {
    if (rows % 32 != 0 && !allow_half) continue;
    int64_t tiles = ((M + rows - 1) / rows) * ntn;
    int64_t rounds = (tiles + cus - 1) / cus;
    double c = (double)rounds * (rows / 32.0 + 0.35);
    if (c < bestc) { bestc = c; best = rows; }
  }
  return best;
}
bool use_wide_tiles(int64_t M, int N, int split) {
  static const int env = tune_int("MMF_WIDE", 1);
  static const int min_env = tune_int("MMF_WIDE_MIN", 0);   // tuning override
  // crossover against the 64-row tiles, one bag per step: exact fp32 16,384 rows (0.300 vs 0.329 ms there); the
  // split-operand tiles cross later -- 16,384: 0.307 wide vs 0.284 small, 24,000: 0.355 vs 0.412
  const int min_rows = min_env > 0 ? min_env : (split ? 80 * 256 : 64 * 256);
  return env && N % 256 == 0 && M * (int64_t)(N / 256) >= min_rows;
}

template <int ROWS>
static int launch_linear_wide(LinearParams p, hipStream_t st) {
  using T = TileW<ROWS>;
  p.mt_count = (int)((p.M + T::BM - 1) / T::BM); p.nt_count = p.N / 256;
  return launch_tiled<T>("linear_nt_kernel", linear_nt_kernel<T>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
}

int split_min_rows() {
  // measured one bag per step, exact fp32 -> bf16x3: 1k 0.101 -> 0.096 ms, 2k 0.115 -> 0.106, 4,096 0.137 -> 0.127,
  // 6k 0.165 -> 0.140, 10k 0.235 -> 0.190, 14k 0.285 -> 0.235 (profiles/r02/d_split_sizes.txt): never slower.  Default 1:
  // every bag takes the split tiles, so that an instance's score does not depend on the size of the bag it is scored
  // in, bit for bit (heat-map batches of 512 patches vs the whole slide: tests/test_gpu_infer.py) -- the accumulation
  // order is the same on every split tile shape, but not between a split tile and an exact-fp32 one
  static const int env = tune_int("MMF_SPLIT_MIN", 1);   // tuning override
  return env;
}
template <int ROWS, int WM, int WN, int GM = 1>
static int launch_linear_split(LinearParams p, hipStream_t st) {
  using T = TileSp<ROWS, 256, WM, WN, GM>;
  p.mt_count = (int)((p.M + T::BM - 1) / T::BM); p.nt_count = p.N / 256;
  return launch_tiled<T>("linear_nt_split_kernel", linear_nt_split_kernel<T>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
}

// K split of the 64 x 64 projection tiles on short grids (LinearParams::ksplit).  A bag of 1,000 instances is 64 tiles, each
// a serial loop over 32 k-chunks (~25 us whatever the tile does: one MFMA block per wave and chunk), on 256 CUs; the radio
// head's reduce_dim (M = 512, K = 4 x 1024) is 128 tiles of 128 chunks.  Split S ways they are S times as many workgroups with
// loops 1 / S as long; the partial tiles (S x M x N floats) cross L2 / Infinity Cache once each way.
int linear_ksplit(int64_t M, int N, int K, int nseg, int kseg) {
  static const int max_wg = tune_int("MMF_KSPLIT_MAX_WG", 512);     // tuning override: 0 = never split
  if (M <= 0 || N % 4 != 0 || K % KC != 0 || use_wide_tiles(M, N, 0) || use_big_tiles(M, N)) return 1;
  const int64_t tiles = ((M + 63) / 64) * ((N + 63) / 64);
  const int nk = K / KC;
  if (nseg > 1 && (kseg % (4 * KC) != 0)) return 1;
  // measured (round 4, path bags, projection kernel us unsplit -> split): 512 rows 25.4 -> 23.2, 1,000 26.4 -> 20.3, 2,000
  // 26.5 -> 24.5, 4,096 (256 tiles, two ways) 26.4 -> 29.3: from one tile per CU on, splitting only adds the partial traffic
  static const int max_tiles = tune_int("MMF_KSPLIT_MAX_TILES", 128);
  if (tiles > max_tiles) return 1;
  for (int S = 4; S >= 2; S -= 2)
    if (tiles * S <= max_wg && nk % (4 * S) == 0 && nk / S >= 8) return S;
  return 1;
}
size_t linear_ksplit_floats(int64_t M, int N, int K, int nseg, int kseg) {
  const int S = linear_ksplit(M, N, K, nseg, kseg);
  if (S <= 1) return 0;
  return (size_t)S * (size_t)((M + 63) / 64) * (size_t)((N + 63) / 64) * 64 * 64;
}

int launch_linear(LinearParams p, hipStream_t st) {
  if (p.K % KC != 0 || (p.nseg > 1 && p.kseg % KC != 0)) return MMF_ERR_SHAPE;
  if (p.ldx % 4 != 0) return MMF_ERR_ALIGN;
  if (p.M <= 0) return MMF_OK;
  const bool can_split = p.split && p.K % (4 * SKC) == 0 && p.nseg == 1;
  if (can_split && use_wide_tiles(p.M, p.N, 1)) {
    static const int rows = tune_int("MMF_SPLIT_ROWS", 224);
    static const int gm = tune_int("MMF_SPLIT_GM", 1);      // A/B switch
    if (gm == 2 && rows == 224) return launch_linear_split<224, 1, 8, 2>(p, st);
    if (rows == 256) return launch_linear_split<256, 2, 4>(p, st);
    if (rows == 192) return launch_linear_split<192, 1, 8>(p, st);
    return launch_linear_split<224, 1, 8>(p, st);
  }
  if (use_wide_tiles(p.M, p.N, can_split)) {
    switch (pick_wide_rows(p.M, p.N / 256, p.allow_half != 0, p.concurrent != 0, 240)) {
#define MMF_WIDE_CASE(R) case R: return launch_linear_wide<R>(p, st);
      MMF_WIDE_CASE(64) MMF_WIDE_CASE(80) MMF_WIDE_CASE(96) MMF_WIDE_CASE(112) MMF_WIDE_CASE(128)
      MMF_WIDE_CASE(144) MMF_WIDE_CASE(160) MMF_WIDE_CASE(176) MMF_WIDE_CASE(192) MMF_WIDE_CASE(208) MMF_WIDE_CASE(240)
#undef MMF_WIDE_CASE
      default: return launch_linear_wide<224>(p, st);
    }
  }
  if (!can_split && use_big_tiles(p.M, p.N)) {
    p.mt_count = (int)((p.M + 127) / 128); p.nt_count = (p.N + 127) / 128;
    return launch_tiled<TileNT128>("linear_nt_kernel", linear_nt_kernel<TileNT128>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
  }
  p.mt_count = (int)((p.M + 63) / 64); p.nt_count = (p.N + 63) / 64;
  if (can_split && p.M >= split_min_rows()) {
    using T = TileSp<64, 64, 2, 2>;
    return launch_tiled<T>("linear_nt_split_kernel", linear_nt_split_kernel<T>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
  }
  {
    const int S = (p.kpart && p.ktick) ? linear_ksplit(p.M, p.N, p.K, p.nseg, p.kseg) : 1;
    if (S > 1 && p.mt_count * p.nt_count <= p.ktick_words) {
      p.ksplit = S;
      p.deep = short_grid((int64_t)p.mt_count * p.nt_count * S) ? 1 : 0;        // (K / KC / S) % 4 == 0 by linear_ksplit
      return launch_tiled<TileNT64>("linear_nt_kernel", linear_nt_kernel<TileNT64>, p, grid_for_tiles(p.mt_count, p.nt_count * S), st);
    }
  }
  // a segmented (radio: M = 512, K = 4 x 1024) or otherwise long-K projection on <= 128 workgroups leaves half the CUs
  // idle for the whole K loop: half-height tiles (two waves) put it on twice as many
  static const int half_tiles = tune_int("MMF_LINEAR_HALF_TILES", 0);   // A/B switch (off: measured slower or equal)
  static const int deep_seg = tune_int("MMF_DEEP_SEG", 1);      // A/B switch: deep prefetch for segmented inputs too
  p.deep = short_grid(p.mt_count * p.nt_count) && (p.K / KC) % 4 == 0 && (p.nseg == 1 || (deep_seg && p.kseg % (4 * KC) == 0)) ? 1 : 0;
  if (half_tiles && !p.deep && p.mt_count * p.nt_count <= 128 && p.M > 32 && p.K >= 1024) {   // (short plain grids: deep prefetch instead)
    p.mt_count = (int)((p.M + 31) / 32);
    return launch_tiled<TileNT32>("linear_nt_kernel", linear_nt_kernel<TileNT32>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
  }
  return launch_tiled<TileNT64>("linear_nt_kernel", linear_nt_kernel<TileNT64>, p, grid_for_tiles(p.mt_count, p.nt_count), st);
}

int gate_parts(int D, int gated, int64_t N) {
  (void)N;
  // one s_part row per attention-dim tile; tile width must match launch_gate_fwd()
  return gated ? (D + 63) / 64 : (D + 127) / 128;
}

int launch_gate_fwd(GateFwdParams p, hipStream_t st) {
  if (p.H % KC != 0 || p.D % 32 != 0) return MMF_ERR_SHAPE;
  if (p.N <= 0) return MMF_OK;
  // BN = 128 always (64 gated dims, or 128 ungated dims, per tile) so that gate_parts() is size independent
  p.nt_count = gate_parts(p.D, p.gated, p.N);
  p.row_begin = 0; p.row_end = p.N;
  using TS = Tile<64, 128, 2, 2, true, true>;
  using SB = TileSp<128, 128, 2, 2>;
  using SS = TileSp<64, 128, 2, 2>;
  const bool split = p.split && p.H % (4 * SKC) == 0;
  auto small = [&](GateFwdParams q) {
    q.mt_count = (int)((q.row_end - q.row_begin + 63) / 64);
    if (split && q.N >= split_min_rows()) {
      const int grid = grid_for_tiles(q.mt_count, q.nt_count);
      return q.gated ? launch_tiled<SS>("gate_fwd_split_kernel", gate_fwd_kernel<SS, true>, q, grid, st)
                     : launch_tiled<SS>("gate_fwd_split_kernel", gate_fwd_kernel<SS, false>, q, grid, st);
    }
    q.deep = short_grid(q.mt_count * q.nt_count) && (q.H / KC) % 4 == 0 ? 1 : 0;
    const int grid = grid_for_tiles(q.mt_count, q.nt_count);
    return q.gated ? launch_tiled<TS>("gate_fwd_kernel", gate_fwd_kernel<TS, true>, q, grid, st)
                   : launch_tiled<TS>("gate_fwd_kernel", gate_fwd_kernel<TS, false>, q, grid, st);
  };
  // 128-row tiles only once they fill most of the 512 slots (two workgroups per CU): a 10k bag is 316 tall tiles -- every
  // CU busy for a tall tile's time, 60 of them twice -- or 628 short ones in 1.2 rounds: measured 40.6 vs 35.1 us (round 4,
  // tools/r4_mid_try.sh; 12,288 rows: 40.3 vs 35.3; from 12,800 rows the tall tiles win, 14k: 41.0 vs 44.0)
  static const int big_min = tune_int("MMF_GATE_BIG_MIN", 400);
  const bool big = (p.N / 128) * p.nt_count >= big_min;
  if (!big) return small(p);
  // 128x128 tiles run two per CU: 512 slots.  Two things cost this launch time: a sparse last round (a 50k bag is
  // 1564 tiles = 3 rounds + 28 tiles, which cost most of a 4th round: 132 us against 116 us for the 1536 tiles of
  // 49,152 rows) and the two workgroups of a CU running IN PHASE -- both in their main loops, then both in their
  // epilogues (tanh / sigmoid pairs, 64 KB of a, b stores each) with the MFMA pipe idle.  Both are answered by the
  // ORDER and the HEIGHT of the tiles in one launch (workgroups start in block order as slots free up; the first 512
  // start at once, the dispatcher placing one per CU before the second):
  //   A  256 tall tiles          -> slot 1 of every CU, finishing at T, 2T, 3T ...
  //   B  256 short (64-row) ones -> slot 2, finishing at T/2: from here on slot 2 runs half a tile out of phase
  //   C  256 (2R - 2) tall tiles -> taken alternately by slot 2 (at T/2, 3T/2, ...) and slot 1 (at T, 2T, ...)
  //   D  256 short tiles         -> slot 2's last, so that both slots end at R T
  //   E  the remaining rows as short tiles (twice as many CUs work on them, each for half the time)
  // R = whole rounds of 512 tall tiles in the bag.  MMF_GATE_MIXED=2: A + E only (the first version of this: 133 -> 128 us).
  // split-operand tiles: their main loop is short against their epilogue, and the out-of-phase plan's extra short tiles
  // cost more than the phase offset wins (A + E only: 91.2 us; dephased: 96.4; uniform tiles: 94.5)
  static const int env_gate = tune_int("MMF_GATE_MIXED", -1);   // A/B switch
  const int env_mixed = env_gate >= 0 ? env_gate : (split ? 2 : 1);
  int64_t mt = (p.N + 127) / 128;
  const int64_t slots = 512, total = mt * p.nt_count;
  const int R = (int)(total / slots);
  if (env_mixed && R >= 2 && slots % (2 * p.nt_count) == 0) {
    const int half_m = (int)(slots / 2 / p.nt_count);          // m-tiles in a wave of 256 workgroups
    int64_t row = 0;
    int grid = 0, n = 0;
    auto region = [&](int tall, int64_t m_tiles) {
      if (m_tiles <= 0) return;
      p.reg[n].row0 = row; p.reg[n].mt_count = (int)m_tiles; p.reg[n].grid_begin = grid; p.reg[n].tall = tall;
      grid += grid_for_tiles((int)m_tiles, p.nt_count);
      row += m_tiles * (tall ? 128 : 64);
      ++n;
    };
    if (env_mixed == 2) {
      const int64_t rem = total % slots;
      region(1, (total - rem) / p.nt_count);
    } else {
      region(1, half_m);
      region(0, half_m);
      region(1, (int64_t)half_m * (2 * R - 2));
      region(0, half_m);
    }
    region(0, (p.N - row + 63) / 64);
    p.nreg = n;
    if (split)
      return p.gated ? launch_tiled<SB>("gate_fwd_split_kernel", gate_fwd_mixed_kernel<SB, SS, true>, p, grid, st)
                     : launch_tiled<SB>("gate_fwd_split_kernel", gate_fwd_mixed_kernel<SB, SS, false>, p, grid, st);
    return p.gated ? launch_tiled<TileNT128>("gate_fwd_kernel", gate_fwd_mixed_kernel<TileNT128, TS, true>, p, grid, st)
                   : launch_tiled<TileNT128>("gate_fwd_kernel", gate_fwd_mixed_kernel<TileNT128, TS, false>, p, grid, st);
  }
  p.mt_count = (int)mt;
  const int grid = grid_for_tiles(p.mt_count, p.nt_count);
  if (split)
    return p.gated ? launch_tiled<SB>("gate_fwd_split_kernel", gate_fwd_kernel<SB, true>, p, grid, st)
                   : launch_tiled<SB>("gate_fwd_split_kernel", gate_fwd_kernel<SB, false>, p, grid, st);
  return p.gated ? launch_tiled<TileNT128>("gate_fwd_kernel", gate_fwd_kernel<TileNT128, true>, p, grid, st)
                 : launch_tiled<TileNT128>("gate_fwd_kernel", gate_fwd_kernel<TileNT128, false>, p, grid, st);
}

int pool_groups(int64_t N) {
  // one 4-wave workgroup per CU: the partial kernel streams h at ~5 TB/s with 256, 512 or 1024 groups alike
  // (measured), and the merge reads one partial per group
  static const int cap = tune_int("MMF_POOL_GROUPS", 256);   // tuning override
  int64_t g = (N + 63) / 64;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  while ((N + g - 1) / g > POOL_MAX_ROWS) g *= 2;
  return (int)g;
}

int launch_pool(PoolParams p, hipStream_t st) {
  if (p.H != 256 && p.H != 512 && p.H != 1024) return MMF_ERR_SHAPE;
  p.n_groups = pool_groups(p.N);
  p.rows_per_group = (int)((p.N + p.n_groups - 1) / p.n_groups);
  if (p.rows_per_group > POOL_MAX_ROWS) return MMF_ERR_SHAPE;
  { ProfScope ps("pool_partial_kernel", st); hipLaunchKernelGGL(pool_partial_kernel, dim3(p.n_groups), dim3(256), 0, st, p); }
  return launch_pool_merge(p, st);
}

int launch_pool_merge(PoolParams p, hipStream_t st) {
  if (p.n_groups < 1 || p.n_groups > MERGE_MAX_GROUPS || p.H > 1024 || p.H % 32 != 0) return MMF_ERR_SHAPE;
  if (p.tail.Wk && (p.tail.K < 1 || p.tail.K > 32)) return MMF_ERR_SHAPE;
  // Merging inside the single-workgroup tail kernel saves a launch and loses more than that: measured, one bag per
  // step, separate merge vs merge in the tail: 1k 0.0995 vs 0.1016 ms, 2k 0.1085 vs 0.1146, 4,096 (64 groups) 0.1236 vs
  // 0.1370.  Off by default; MMF_TAIL_MERGE=<max groups, <= 64> turns it back on.
  static const int tail_merge = tune_int("MMF_TAIL_MERGE", 0);
  p.merge_in_tail = p.tail.Wk && p.n_groups <= (tail_merge < TAIL_MERGE_MAX ? tail_merge : TAIL_MERGE_MAX) ? 1 : 0;
  if (!p.merge_in_tail) {
    ProfScope ps("pool_merge_kernel", st);
    hipLaunchKernelGGL(pool_merge_kernel, dim3(p.H / 32), dim3(1024), 0, st, p);
  }
  if (p.tail.Wk) {
    ProfScope ps("head_tail_kernel", st);
    hipLaunchKernelGGL(head_tail_kernel, dim3(1), dim3(1024), 0, st, p);
  }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

// The head tail on a feature vector that is already there (p.M [p.H <= 1024]; the multimodal step: the branches' embeddings
// side by side): classifier, hazards, nll_surv and its backward down to d(feature), one single-workgroup launch.
int launch_head_tail(PoolParams p, hipStream_t st) {
  if (!p.M || !p.tail.Wk || p.H < 1 || p.H > 1024 || p.tail.K < 1 || p.tail.K > 32) return MMF_ERR_SHAPE;
  p.merge_in_tail = 0;
  ProfScope ps("head_tail_kernel", st);
  hipLaunchKernelGGL(head_tail_kernel, dim3(1), dim3(1024), 0, st, p);
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

// diagnostic: read and clear this translation unit's phase stamps (zeros unless built with -DMMF_STAMPS)
void debug_stamps_fwd(unsigned long long* out8) {
#ifdef MMF_STAMPS
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), 8 * sizeof(unsigned long long));
  unsigned long long z[8] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z);
#else
  for (int i = 0; i < 8; ++i) out8[i] = 0;
#endif
}

}  // namespace mmf
