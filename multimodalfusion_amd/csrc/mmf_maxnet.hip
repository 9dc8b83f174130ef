// Omic head, one training step in ONE launch ("the goal is one launch").
//
//   MaxNet.forward      f = SNN(SNN(x)); risk = classifier(f).squeeze()      models/model_genomic.py:53-72
//     SNN_Block         Linear + SELU + AlphaDropout(0.25)                    models/model_modules.py:64-68
//   CoxSurvLoss         -mean((theta - log sum_j e^theta_j [t_j >= t_i]) (1 - c))   utils/loss_utils.py:124-139
//   and what autograd derives from them.
//
// The reference spends ~20 framework launches on this step (three addmm / selu / alpha-dropout forward, a Python double
// loop for the risk-set matrix, the same again backward); round 3 of this repo spent 9 (three dense forward, Cox, three
// dense backward, an H2D copy and a fill: 82 us of GPU time, 0.29 ms with the host's issue time).  Here: 32 workgroups (64 for batches of 129 - 256 rows).
// The step is latency, not throughput (52 MFLOP): what counts is the number of dependent global round trips, so
//   phase 1  rows:   workgroup j owns R batch rows -- both SNN blocks and the classifier for those rows, nothing leaves the CU
//                    but the saved layer outputs (y0, y1) and the risks;
//   ---- grid barrier 1 (every risk is needed by every row's Cox term)
//   phase 2  Cox:    every workgroup recomputes the B x B risk-set sums (16k terms) and keeps d loss / d risk of ITS rows;
//   phase 3  rows:   d pre-activations of both blocks for its rows (dy0 = dpre1 . W1 with W1 read as it lies in memory);
//   ---- grid barrier 2 (a weight gradient sums over ALL rows)
//   phase 4  slices: workgroup j owns 8 output features of every layer: dW[n][:] = sum_b dpre[b][n] in[b][:], db[n], and
//                    workgroup 0 the classifier's -- each gradient element is one thread's sum in batch order: deterministic.
// The two barriers are counters in the caller's tick words (mmf_amil_desc::sync's contract: zero before, zero after): 32
// workgroups of 256 threads are co-resident on any MI355X that is not wedged, and nothing else ever waits on them.
// Data that crosses workgroups (risks, y0, y1, dpre) crosses XCDs: a device-scope fence on both sides of each barrier.
#include "mmf_common.h"
#include "mmf_kernels.h"
#include "mmf_mlp.h"

namespace mmf {

constexpr float MX_SELU_ALPHA = 1.6732632423543772f;
constexpr float MX_SELU_SCALE = 1.0507009873554805f;
constexpr int MX_H = 256;            // both hidden widths of the `small` omic net (model_genomic.py:17)
constexpr int MX_R = 4;              // batch rows per workgroup in phases 1 - 3
constexpr int MX_NW_MAX = 64;        // workgroups: 32 for B <= 128, 64 for B <= 256 (kernel template parameter NW)


__device__ inline float mx_selu(float v) { return MX_SELU_SCALE * (v > 0.f ? v : MX_SELU_ALPHA * (expf(v) - 1.0f)); }
__device__ inline float mx_selu_grad_from_y(float y) { return y > 0.f ? MX_SELU_SCALE : y + MX_SELU_SCALE * MX_SELU_ALPHA; }

struct MxDrop { float a, b, alpha_p; uint32_t thr; bool on; };
__device__ inline MxDrop mx_drop(float p) {
  MxDrop d;
  d.on = p > 0.f;
  d.alpha_p = -MX_SELU_ALPHA * MX_SELU_SCALE;
  d.a = d.on ? 1.0f / sqrtf((d.alpha_p * d.alpha_p * p + 1.0f) * (1.0f - p)) : 1.f;
  d.b = d.on ? -d.a * d.alpha_p * p : 0.f;
  d.thr = drop_threshold(p);
  return d;
}

// FENCED = false: nothing but values written with device-scope stores and read with device-scope loads crosses this barrier
// (the risks at barrier 1), so the release / acquire fences -- an L2 write-back and an invalidate -- are left out: the
// __syncthreads in front has waited for the stores (workgroup-scope release = vmcnt(0); a device-scope store is complete when
// it is visible at the coherent level), the ticket is an agent-scope atomic.
// `mid` runs on every thread between the two workgroup barriers, i.e. while thread 0 takes its ticket and polls: loads that
// depend on nothing another workgroup writes are requested there and travel during the wait.
template <bool FENCED = true, class Mid>
__device__ inline void mx_grid_barrier(unsigned* cnt, int nw, Mid&& mid) {
  __syncthreads();
  mid();
  if (threadIdx.x == 0) {
#ifndef MMF_MX_NOFENCE               // (diagnostic build: the barrier without its fences -- results are wrong, the time is the point)
    if (FENCED) __threadfence();     // release: this workgroup's stores are visible device-wide
#endif
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nw) __builtin_amdgcn_s_sleep(1);
#ifndef MMF_MX_NOFENCE
    if (FENCED) __threadfence();     // acquire
#endif
  }
  __syncthreads();
}

// acc[r] = sum_k xs[k][r] W[n = tid][k], k < K: W (row-major [256][ldw]) is staged through LDS in chunks of 64 k so that the
// global reads are coalesced along k and thread n's reads walk its own padded row; the next chunk travels in registers
// while this one is multiplied.  xs: [K][R] in LDS (R batch values of one k contiguous: one broadcast read per k).
// VEC (ldw % 4 == 0, K % 64 == 0, 16-byte aligned W): 16-byte global loads, LDS writes and LDS reads; the row pitch of 68
// floats keeps both the b128 writes (a quarter wave writes 16 slots of one row) and the b128 reads (a quarter wave reads the
// same four k of 16 consecutive rows: 16 x 4 distinct banks) conflict-free.  `pre` runs once, right behind the first chunk's
// load requests: work that does not depend on W hides their latency.
constexpr int MX_WP = 68;
// One wave per SIMD: nothing hides an LDS round trip, and the compiler sinks every LDS read next to its use (three reads,
// a full wait, four MFMAs, 16 times per chunk = 2.6 k of a chunk's 2.9 k cycles).  A batch of reads written in front of this
// fence is issued before anything behind it (the memory clobber keeps the reads in front, the scheduling barrier keeps the
// MFMAs, which are not memory operations, behind).
#define MX_ISSUE_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
typedef float mx_v4f __attribute__((ext_vector_type(4)));
// The risks, all that crosses barrier 1, are written and read at DEVICE scope (the sc1 bit: through to / from the level the
// XCDs' L2s are coherent at), so that barrier needs no L2 write-back / invalidate around its ticket.
__device__ inline void mx_st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float mx_ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// v_mfma_f32_4x4x1 with CBSZ = 4: the A values of block ABID (lanes 4 ABID .. 4 ABID + 3) serve all 16 blocks.  A register
// whose lane l holds A[row l % 4][k0 + l / 4] therefore feeds 16 consecutive k -- one 4-byte LDS read per lane and 16 k
// instead of one per k.
#define MX_MFMA_K(acc, areg, bval, q) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(areg, bval, acc, 4, q, 0)
template <int R, bool VEC, class Pre>
__device__ inline void mx_rows_gemm(const float* __restrict__ xs, const float* __restrict__ W, int ldw, int K, float* wl,
                                    float (&acc)[R], Pre&& pre, unsigned long long* dbg = nullptr) {
#ifdef MMF_STAMPS
#define MX_GSTAMP(i) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[i] = __builtin_readcyclecounter(); } while (0)
#else
#define MX_GSTAMP(i)
  (void)dbg;
#endif
  const int tid = threadIdx.x;
  typedef mx_v4f v4f;
  // four accumulator sets, k mod 4: a chain of dependent 4x4x1 MFMAs advances one instruction per ~44 cycles (measured: 64 of
  // them 2.8 k cycles), four interleaved chains keep the unit busy; the sets are added at the end, (0 + 1) + (2 + 3)
  v4f accq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) accq[q] = v4f{0.f, 0.f, 0.f, 0.f};
  // two chunks travel in registers: with the products on the matrix unit a chunk is multiplied in ~1 k cycles, less than
  // the round trip of the next one's loads
  float4 stage0[16], stage1[16];
  auto fetch = [&](int k0, float4 (&stage)[16]) {
    if constexpr (VEC) {             // thread t: float4 e = t + 256 i of the [256][16 float4] chunk: row e / 16, float4 e % 16
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = tid + 256 * i;
        stage[i] = ld4(W + (size_t)(e >> 4) * ldw + k0 + 4 * (e & 15));
      }
    } else {                         // any ldw / K: scalars, element e = t + 256 i: row e / 64, column e % 64
      // (tried for layer 0, G = 36, 9.4 k cycles: only the chunk's 36 columns with row / column advanced incrementally -- 13 k,
      // the address chain serialises the loads; W0 as one contiguous block of 16-byte loads with a division per float4 --
      // 8.6 k, but 36 - 44 B of scratch per lane from the second staging path; neither kept)
      float* sf = reinterpret_cast<float*>(stage);
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        const int e = tid + 256 * i, n = e >> 6, kk = e & 63;
        sf[i] = (k0 + kk < K) ? W[(size_t)n * ldw + k0 + kk] : 0.f;
      }
    }
  };
  auto put = [&](const float4 (&stage)[16]) {
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = tid + 256 * i;
        st4(wl + (e >> 4) * MX_WP + 4 * (e & 15), stage[i]);
      }
    } else {
      const float* sf = reinterpret_cast<const float*>(stage);
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        const int e = tid + 256 * i;
        wl[(e >> 6) * MX_WP + (e & 63)] = sf[i];
      }
    }
  };
  // The products run on v_mfma_f32_4x4x1_16B_f32: 16 independent 4 x 4 blocks per instruction -- block q of a wave is the
  // wave's columns 4 q .. 4 q + 3 against the workgroup's four rows.  Lane l supplies A = x[row l % 4][k] (one 4-byte LDS
  // read per k) and B = W[column l][k]; its four accumulator registers are rows 0 .. 3 of column l -- the same ownership as
  // a scalar loop over k, one k per instruction.
  static_assert(R == 4, "the 4 x 4 x 1 blocks hold four rows");
  auto chunk = [&](int k0) {
    const int kn = K - k0 < 64 ? K - k0 : 64;
    const float* wr = wl + tid * MX_WP;
    const float* xa = xs + (size_t)k0 * R + (tid & 3);
    if (kn == 64) {                  // a full chunk: all of its LDS reads (16 x 16 B of W, 64 x 4 B of x) in one batch
      float4 w4[16];
      float a16[4];                  // lane l: x[row l % 4][k0 + 16 g + l / 4]  (xs is [k][4]: element k0 * 4 + 64 g + l)
#pragma unroll
      for (int u = 0; u < 16; ++u) w4[u] = ld4(wr + 4 * u);
#pragma unroll
      for (int g = 0; g < 4; ++g) a16[g] = xs[(size_t)k0 * R + 64 * g + (tid & 63)];
      MX_ISSUE_FENCE();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#define MX_Q4(j) MX_MFMA_K(accq[0], a16[g], w4[4 * g + j].x, 4 * j); MX_MFMA_K(accq[1], a16[g], w4[4 * g + j].y, 4 * j + 1); \
                 MX_MFMA_K(accq[2], a16[g], w4[4 * g + j].z, 4 * j + 2); MX_MFMA_K(accq[3], a16[g], w4[4 * g + j].w, 4 * j + 3)
        MX_Q4(0); MX_Q4(1); MX_Q4(2); MX_Q4(3);
#undef MX_Q4
      }
    } else {
      int kk = 0;
      for (; kk + 3 < kn; kk += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          accq[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[(size_t)(kk + u) * R], wr[kk + u], accq[u], 0, 0, 0);
      }
      for (; kk < kn; ++kk) accq[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[(size_t)kk * R], wr[kk], accq[0], 0, 0, 0);
    }
  };
  fetch(0, stage0);
  if (K > 64) fetch(64, stage1);
  pre();
  MX_GSTAMP(0);
#pragma unroll 1
  for (int k0 = 0; k0 < K; k0 += 128) {
    __syncthreads();                 // the previous chunk has been read by everybody
    put(stage0);
    __syncthreads();
    if (k0 == 0) MX_GSTAMP(1);
    if (k0 + 128 < K) fetch(k0 + 128, stage0);
    chunk(k0);
    if (k0 == 0) MX_GSTAMP(2);
    if (k0 + 64 < K) {
      __syncthreads();
      put(stage1);
      __syncthreads();
      if (k0 + 192 < K) fetch(k0 + 192, stage1);
      chunk(k0 + 64);
    }
  }
  MX_GSTAMP(3);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = (accq[0][r] + accq[1][r]) + (accq[2][r] + accq[3][r]);
}

// The same product for a narrow first layer (K <= 64, K % 4 == 0, W = one contiguous 16-byte aligned [256][K] block: G = 36):
// thread n keeps ITS row of W in registers -- K / 4 16-byte loads per lane, 144 bytes apart at K = 36, so a wave's K / 4
// instructions together read one contiguous 9 KB block -- and nothing goes through LDS but x.  (The chunked path stages a
// 64-wide chunk of a 36-wide matrix with 64 predicated scalar loads and LDS stores per thread: 9.4 k cycles.)
template <int R, class Pre>
__device__ inline void mx_rows_narrow(const float* __restrict__ xs, const float* __restrict__ W, int K, float (&acc)[R], Pre&& pre) {
  static_assert(R == 4, "the 4 x 4 x 1 blocks hold four rows");
  const int tid = threadIdx.x;
  float4 w4[16];
  const float* wrow = W + (size_t)tid * K;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (4 * i < K) w4[i] = ld4(wrow + 4 * i);
  pre();
  __syncthreads();                   // x is in LDS
  mx_v4f accq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) accq[q] = mx_v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (16 * g + 15 < K) {           // a full group of 16 k on one broadcast A register
      const float a16 = xs[64 * g + (tid & 63)];
#define MX_Q4(j) MX_MFMA_K(accq[0], a16, w4[4 * g + j].x, 4 * j); MX_MFMA_K(accq[1], a16, w4[4 * g + j].y, 4 * j + 1); \
                 MX_MFMA_K(accq[2], a16, w4[4 * g + j].z, 4 * j + 2); MX_MFMA_K(accq[3], a16, w4[4 * g + j].w, 4 * j + 3)
      MX_Q4(0); MX_Q4(1); MX_Q4(2); MX_Q4(3);
#undef MX_Q4
    } else {                         // the last, partial group: four k per 16-byte register
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 16 * g + 4 * j;
        if (k < K) {
          const float* xa = xs + (size_t)k * R + (tid & 3);
          accq[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[0], w4[4 * g + j].x, accq[0], 0, 0, 0);
          accq[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[R], w4[4 * g + j].y, accq[1], 0, 0, 0);
          accq[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[2 * R], w4[4 * g + j].z, accq[2], 0, 0, 0);
          accq[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[3 * R], w4[4 * g + j].w, accq[3], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = (accq[0][r] + accq[1][r]) + (accq[2][r] + accq[3][r]);
}

// dst[e] = src(e) for e < n, every thread's loads of a batch of 8 issued before its first LDS store (a plain loop is one
// memory round trip per element: the compiler keeps a load in front of the store that might alias it)
template <class Src>
__device__ inline void mx_stage(float* dst, int n, Src&& src) {
  const int tid = threadIdx.x;
  for (int e0 = tid; e0 < n; e0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = e0 + 256 * u < n ? src(e0 + 256 * u) : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + 256 * u < n) dst[e0 + 256 * u] = v[u];
  }
}

#ifdef MMF_STAMPS
#define MX_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && p.stamps) p.stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define MX_STAMP(i)
#endif

template <int NW>
__global__ __launch_bounds__(256) void maxnet_cox_step_kernel(MaxnetStepParams p) {
  constexpr int R = MX_R, MX_NW = NW, MX_NS = MX_H / NW;     // output features per workgroup in phase 4
  constexpr int BP = NW * MX_R;                              // row pitch of the transposed dpre arrays: every workgroup's rows
  extern __shared__ __align__(16) float sm[];
  MX_STAMP(0);
  float* xs = sm;                               // [256][R] layer input of this workgroup's rows
  float* wl = xs + 256 * R;                     // [256][68] staged weights; phases 2-4: scratch
  float* red = wl + 256 * MX_WP;                // [4][R] + misc
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, G = p.G;
  const int r0 = blockIdx.x * R;
  const uint32_t sdev = p.seed_dev ? *p.seed_dev : 0u;
  const MxDrop dr = mx_drop(p.p);
  const uint32_t key0 = p.key0 + sdev, key1 = p.key1 + sdev;

  // the Cox phase's operands that do not depend on the net are requested now: their round trip hides behind phase 1
  const double t_pre = tid < B ? p.times[tid] : 0.0;
  const float c_pre = tid < B ? p.c[tid] : 0.f;

  // ---------------- phase 1: this workgroup's rows through the net ----------------------------------------------------
  float acc[R], y0d[R], y1d[R];
  auto stage_x = [&]() {
    for (int e = tid; e < G * R; e += 256) {    // the rows' inputs, behind W0's load requests
      const int k = e / R, r = e % R;
      xs[e] = (r0 + r < B) ? p.x[(size_t)(r0 + r) * G + k] : 0.f;
    }
  };
  if (G <= 64 && (G & 3) == 0 && (reinterpret_cast<uintptr_t>(p.W0) & 15) == 0) mx_rows_narrow<R>(xs, p.W0, G, acc, stage_x);
  else mx_rows_gemm<R, false>(xs, p.W0, G, G, wl, acc, stage_x);
  MX_STAMP(1);
  {
    const float b = p.b0[tid];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float y = mx_selu(acc[r] + b);
      const uint32_t idx = (uint32_t)(r0 + r) * MX_H + (uint32_t)tid;
      y0d[r] = dr.on ? dr.a * (keep(key0, idx, dr.thr) ? y : dr.alpha_p) + dr.b : y;
      if (r0 + r < B) p.y0[(size_t)(r0 + r) * MX_H + tid] = y0d[r];
    }
  }
  const float b1v = p.b1[tid], wcv = p.Wc[tid], bcv = p.bc[0];
  mx_rows_gemm<R, true>(xs, p.W1, MX_H, MX_H, wl, acc, [&]() {
    __syncthreads();                            // every thread is done with xs as layer 0's input
#pragma unroll
    for (int r = 0; r < R; ++r) xs[tid * R + r] = y0d[r];
  }, p.stamps ? p.stamps + 12 : nullptr);
  float part[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float y = mx_selu(acc[r] + b1v);
    const uint32_t idx = (uint32_t)(r0 + r) * MX_H + (uint32_t)tid;
    y1d[r] = dr.on ? dr.a * (keep(key1, idx, dr.thr) ? y : dr.alpha_p) + dr.b : y;
    part[r] = wave_sum(y1d[r] * wcv);
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) red[wave * R + r] = part[r];
  }
  __syncthreads();
  if (tid < R && r0 + tid < B)       // device-scope store: every workgroup reads every risk right behind barrier 1
    mx_st_dev(p.risk + r0 + tid, red[tid] + red[R + tid] + red[2 * R + tid] + red[3 * R + tid] + bcv);
  MX_STAMP(2);
  constexpr int SG3 = 32;
  float w3[2][SG3];                             // dy0's first two stages of W1 rows (phase 3), requested while the barrier waits
  mx_grid_barrier<false>(p.bar, MX_NW, [&]() {
#pragma unroll
    for (int u = 0; u < SG3; ++u) w3[0][u] = p.W1[(size_t)u * MX_H + tid];
#pragma unroll
    for (int u = 0; u < SG3; ++u) w3[1][u] = p.W1[(size_t)(SG3 + u) * MX_H + tid];
  });
  MX_STAMP(3);

  // ---------------- phase 2: Cox over the whole batch, gradient of this workgroup's rows ---------------------------------
  float* th = wl;                               // [256] theta
  float* et = wl + 256;                         // [256] e^theta
  float* uc = wl + 512;                         // [256] 1 - c
  float* wq = wl + 768;                         // [256] (1 - c_i) / D_i
  float* lp = wl + 1024;                        // [4] loss terms per wave
  double* tl = reinterpret_cast<double*>(wl + 1280);    // [256] event times
  if (tid < B) {
    const float t = mx_ld_dev(p.risk + tid);
    th[tid] = t; et[tid] = expf(t); uc[tid] = 1.f - c_pre; tl[tid] = t_pre;
  }
  __syncthreads();
  float lterm = 0.f;
  if (tid < B) {                                // B <= 256: one risk set per thread
    const double ti = tl[tid];
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    int j = 0;
    // 16 risk-set members per pass, their twelve 16-byte LDS reads issued together (four at a time, each read waited for,
    // was one LDS round trip per member: 16 k of the phase's 19 k cycles)
    for (; j + 15 < B; j += 16) {
      double2 t2[8];
      float4 e4[4];
#pragma unroll
      for (int u = 0; u < 8; ++u) t2[u] = *reinterpret_cast<const double2*>(tl + j + 2 * u);
#pragma unroll
      for (int u = 0; u < 4; ++u) e4[u] = *reinterpret_cast<const float4*>(et + j + 4 * u);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d0 += (t2[2 * u].x >= ti) ? e4[u].x : 0.f; d1 += (t2[2 * u].y >= ti) ? e4[u].y : 0.f;
        d2 += (t2[2 * u + 1].x >= ti) ? e4[u].z : 0.f; d3 += (t2[2 * u + 1].y >= ti) ? e4[u].w : 0.f;
      }
    }
    for (; j + 3 < B; j += 4) {
      d0 += (tl[j] >= ti) ? et[j] : 0.f; d1 += (tl[j + 1] >= ti) ? et[j + 1] : 0.f;
      d2 += (tl[j + 2] >= ti) ? et[j + 2] : 0.f; d3 += (tl[j + 3] >= ti) ? et[j + 3] : 0.f;
    }
    for (; j < B; ++j) d0 += (tl[j] >= ti) ? et[j] : 0.f;
    const float Di = (d0 + d1) + (d2 + d3);
    lterm = (th[tid] - logf(Di)) * uc[tid];
    wq[tid] = uc[tid] / Di;
  }
  lterm = wave_sum(lterm);
  if (lane == 0) lp[wave] = lterm;
  __syncthreads();
  const float invB = 1.0f / (float)B;
  {
    // row r of this workgroup: sum_i [t_k >= t_i] w_i over the batch, R rows x (256 / R) threads each
    constexpr int TPR = 256 / R;
    const int r = tid / TPR, q = tid % TPR, k = r0 + r;
    float a = 0.f;
    if (k < B) {
      const double tk = tl[k];
      for (int i = q; i < B; i += TPR) a += (tk >= tl[i]) ? wq[i] : 0.f;
    }
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);       // TPR <= 64: within one wave
    if (q == 0) {
      float g = 0.f;
      if (k < B) {
        g = -invB * (uc[k] - et[k] * a) * p.loss_scale;
        p.dr[k] = g;
      }
      red[r] = g;
    }
  }
  if (blockIdx.x == 0 && tid == 0) p.loss[0] = -((lp[0] + lp[1]) + (lp[2] + lp[3])) * invB;
  __syncthreads();

  MX_STAMP(4);
  // ---------------- phase 3: d pre-activations of this workgroup's rows --------------------------------------------------
  float* dps = xs;                              // [256][R] dpre1 of these rows (layer 1's outputs n)
  {
    float wpart = 0.f;                          // this workgroup's share of dWc[k = tid] = sum_b dr[b] y1[b][k]
    float dd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t idx = (uint32_t)(r0 + r) * MX_H + (uint32_t)tid;
      float dydy = 1.f, y = y1d[r];
      if (dr.on) {
        const bool kp = keep(key1, idx, dr.thr);
        dydy = kp ? dr.a : 0.f;
        y = kp ? (y1d[r] - dr.b) / dr.a : 0.f;
      }
      const float d = red[r] * wcv * dydy * mx_selu_grad_from_y(y);
      dps[tid * R + r] = d;
      dd[r] = d;                                // rows beyond the batch: red[r] = 0, so d = 0
      if (r0 + r < B) wpart += red[r] * y1d[r];
    }
    // dpre1 / dpre0 leave transposed ([feature][row], pitch BP): phase 4 wants a feature's whole batch column, and this
    // way it is one contiguous row there instead of a gather of 32-byte pieces
    static_assert(R == 4, "one float4 per thread");
    st4(p.dp1 + (size_t)tid * BP + r0, float4{dd[0], dd[1], dd[2], dd[3]});
    p.dwc_part[(size_t)blockIdx.x * MX_H + tid] = wpart;
  }
  __syncthreads();
  {
    float a[R];
    mx_v4f aq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) aq[q] = mx_v4f{0.f, 0.f, 0.f, 0.f};
    // dy0[r][k = tid] = sum_n dpre1[r][n] W1[n][k]: W1 is read as it lies (row n, coalesced along k), 16 rows per stage and the
    // next stage's 16 loads in flight while this one multiplies (the loop is a chain of memory round trips otherwise)
    // (64 rows per stage, two stages requested before the first is used: with 16-row stages every stage waited out a full
    // memory round trip, 16 of them)
    constexpr int SG = SG3;
    float (&w)[2][SG3] = w3;
#pragma unroll 1
    for (int nb0 = 0; nb0 < MX_H; nb0 += 2 * SG) {       // not unrolled: the compiler would request all 256 rows at once
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int nb = nb0 + h * SG;
        // the same 4 x 4 x 1 blocks as the forward layers: lane = column k, A = dpre1[row l % 4][n] from LDS (the stage's 32
        // reads in one batch), B = W1[n][k] from the registers the stage travelled in
        static_assert(SG == 32, "two 16-row groups per stage");
        float dv16[2];                          // lane l: dpre1[row l % 4][n = nb + 16 g + l / 4]
#pragma unroll
        for (int g = 0; g < 2; ++g) dv16[g] = dps[(nb + 16 * g) * R + (tid & 63)];
        MX_ISSUE_FENCE();
#define MX_D4(g, j) MX_MFMA_K(aq[0], dv16[g], w[h][16 * g + 4 * j], 4 * j); MX_MFMA_K(aq[1], dv16[g], w[h][16 * g + 4 * j + 1], 4 * j + 1); \
                    MX_MFMA_K(aq[2], dv16[g], w[h][16 * g + 4 * j + 2], 4 * j + 2); MX_MFMA_K(aq[3], dv16[g], w[h][16 * g + 4 * j + 3], 4 * j + 3)
        MX_D4(0, 0); MX_D4(0, 1); MX_D4(0, 2); MX_D4(0, 3); MX_D4(1, 0); MX_D4(1, 1); MX_D4(1, 2); MX_D4(1, 3);
#undef MX_D4
        if (nb + 2 * SG < MX_H) {
#pragma unroll
          for (int u = 0; u < SG; ++u) w[h][u] = p.W1[(size_t)(nb + 2 * SG + u) * MX_H + tid];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = (aq[0][r] + aq[1][r]) + (aq[2][r] + aq[3][r]);
    float d0[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t idx = (uint32_t)(r0 + r) * MX_H + (uint32_t)tid;
      float dydy = 1.f, y = y0d[r];
      if (dr.on) {
        const bool kp = keep(key0, idx, dr.thr);
        dydy = kp ? dr.a : 0.f;
        y = kp ? (y0d[r] - dr.b) / dr.a : 0.f;
      }
      d0[r] = a[r] * dydy * mx_selu_grad_from_y(y);         // rows beyond the batch: a[r] = 0
    }
    st4(p.dp0 + (size_t)tid * BP + r0, float4{d0[0], d0[1], d0[2], d0[3]});
  }
  MX_STAMP(5);
  // (tried: y0, dpre1, dpre0, dr and the dWc shares at device scope and this barrier without its fences too -- 32.3 -> 34.3 us:
  // eight write-through scalar stores per thread instead of two 16-byte ones cost phase 3 more than the fences cost here)
  // phase 4's LDS map (wl is free from here on); the first block of x rows -- an input, nothing another workgroup writes --
  // is staged while the barrier waits
  float* d1s = wl;                              // [B][8] dpre1[:, n0 .. n0 + 7]
  float* d0s = wl + 8 * 256;                    // [B][8] dpre0[:, n0 .. n0 + 7]   (B <= 256)
  float* rb = wl + 16 * 256;                    // [parts][G][NS] dW0 partial sums (parts * G <= 256)
  float* xl = wl + 24 * 256;                    // x rows in blocks of XB rows (dW0)
  float* cw = wl + 256 * MX_WP - 256;           // [NS][NW] the dWc shares of this workgroup's columns
  const int XB = (256 * MX_WP - 24 * 256 - 256) / G;   // rows of x that fit behind d1s / d0s / rb and in front of cw
  const int xb0 = B < XB ? B : XB;
  mx_grid_barrier(p.bar + 1, MX_NW, [&]() { mx_stage(xl, xb0 * G, [&](int e) { return p.x[e]; }); });
  MX_STAMP(6);

  // ---------------- phase 4: weight gradients, 8 output features per workgroup ------------------------------------------
  const int n0 = blockIdx.x * MX_NS;
  // everything this phase reads from memory is requested up front, where the addresses do not depend on anything computed here
  // classifier: dWc[k] = sum over the workgroups' shares, in workgroup order -- every workgroup does ITS columns (one load per
  // thread: share tid / NS of column n0 + tid % NS) instead of workgroup 0 all 256 with 32 loads per thread
  static_assert(MX_NS * MX_NW == 256, "one share element per thread");
  const float cshare = p.dwc_part[(size_t)(tid / MX_NS) * MX_H + n0 + tid % MX_NS];
  // (the first two stages of dW1's y0 rows are requested here too: one round trip for all of it)
  constexpr int SG4 = 32;
  float yv[2][SG4];
  auto ld = [&](int b0, float (&v)[SG4]) {
#pragma unroll
    for (int u = 0; u < SG4; ++u) v[u] = b0 + u < B ? p.y0[(size_t)(b0 + u) * MX_H + tid] : 0.f;
  };
  ld(0, yv[0]);
  ld(SG4, yv[1]);
  {                                             // this workgroup's MX_NS feature columns of dpre1 / dpre0: contiguous rows of BP
    static_assert(MX_NS * BP == 4 * 256, "four elements per thread and array");
    float v1[4], v0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      v1[u] = p.dp1[(size_t)(n0 + e / BP) * BP + e % BP];
      v0[u] = p.dp0[(size_t)(n0 + e / BP) * BP + e % BP];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      d1s[(e % BP) * MX_NS + e / BP] = v1[u];   // LDS keeps [row][feature]: the product loops read a row's features together
      d0s[(e % BP) * MX_NS + e / BP] = v0[u];
    }
  }
  cw[(tid % MX_NS) * MX_NW + tid / MX_NS] = cshare;
  __syncthreads();
  MX_STAMP(8);
  {
    float a[MX_NS];
    // dW1[n0 + i][k = tid] = sum_b dpre1[b][n0 + i] y0[b][k]: 32 batch rows per stage, the next stage in flight.  4 x 4 x 1
    // blocks again: lane = column k, A = dpre1[b][feature l % 4 (+ 4)] from LDS (a stage's reads in one batch), B = y0[b][k]
    // from the stage's registers; two accumulator sets per half of the features (rows alternate)
    constexpr int SG = SG4;
    constexpr int NH = MX_NS / 4;                // feature quads of this workgroup: 2 (32 workgroups) or 1 (64)
    mx_v4f dq[NH][2];
#pragma unroll
    for (int f = 0; f < NH; ++f) { dq[f][0] = mx_v4f{0.f, 0.f, 0.f, 0.f}; dq[f][1] = mx_v4f{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int b0 = 0; b0 < B; b0 += 2 * SG) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int bb = b0 + SG * h;
        if (bb < B) {
          static_assert(SG == 32, "two 16-row groups per stage");
          float dv16[NH][2];                     // lane l: dpre1[b = bb + 16 g + l / 4][feature 4 f + l % 4]
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const int b = bb + 16 * g + ((tid & 63) >> 2);
            const float* src = d1s + (b < B ? b : 0) * MX_NS + (tid & 3);                 // rows beyond the batch: yv = 0
#pragma unroll
            for (int f = 0; f < NH; ++f) dv16[f][g] = src[4 * f];
          }
          MX_ISSUE_FENCE();
#pragma unroll
          for (int f = 0; f < NH; ++f) {
#define MX_W4(g, j) MX_MFMA_K(dq[f][0], dv16[f][g], yv[h][16 * g + 4 * j], 4 * j); MX_MFMA_K(dq[f][1], dv16[f][g], yv[h][16 * g + 4 * j + 1], 4 * j + 1); \
                    MX_MFMA_K(dq[f][0], dv16[f][g], yv[h][16 * g + 4 * j + 2], 4 * j + 2); MX_MFMA_K(dq[f][1], dv16[f][g], yv[h][16 * g + 4 * j + 3], 4 * j + 3)
            MX_W4(0, 0); MX_W4(0, 1); MX_W4(0, 2); MX_W4(0, 3); MX_W4(1, 0); MX_W4(1, 1); MX_W4(1, 2); MX_W4(1, 3);
#undef MX_W4
          }
          if (bb + 2 * SG < B) ld(bb + 2 * SG, yv[h]);
        }
      }
    }
#pragma unroll
    for (int f = 0; f < NH; ++f) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[4 * f + i] = dq[f][0][i] + dq[f][1][i];
    }
#pragma unroll
    for (int i = 0; i < MX_NS; ++i) {
      float* o = p.dW1 + (size_t)(n0 + i) * MX_H + tid;
      *o = p.accumulate ? *o + a[i] : a[i];
    }
  }
  MX_STAMP(9);
  {                                              // dW0[n0 + i][g] = sum_b dpre0[b][n0 + i] x[b][g]   (G <= 256)
    // (tried: the same 4 x 4 x 1 blocks as dW1, lane = column, the waves splitting the batch and meeting in LDS -- 4.8 -> 5.5 k
    // cycles at G = 36: two 16-row groups per wave do not pay for the extra reduction)
    // all 256 threads: thread = (batch part, input column g), `parts` = 256 / G interleaved row subsets (36 columns alone
    // would leave 220 threads idle over 128 serial rows); the parts meet in LDS and are added in part order
    const int parts = 256 / G, part = tid / G, g = tid - part * G;
    const bool live = part < parts;
    float a[MX_NS];
#pragma unroll
    for (int i = 0; i < MX_NS; ++i) a[i] = 0.f;
    for (int bs = 0; bs < B; bs += XB) {         // x through LDS, XB rows at a time (all of it at B = 128, G = 36)
      const int nb = B - bs < XB ? B - bs : XB;
      if (bs > 0) {
        __syncthreads();
        mx_stage(xl, nb * G, [&](int e) { return p.x[(size_t)bs * G + e]; });
        __syncthreads();
      }
      if (live) {
#pragma unroll 4
        for (int b = part; b < nb; b += parts) {
          const float xv = xl[b * G + g];
          const float* dv = d0s + (bs + b) * MX_NS;
#pragma unroll
          for (int i = 0; i < MX_NS; ++i) a[i] += dv[i] * xv;
        }
      }
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < MX_NS; ++i) rb[(part * G + g) * MX_NS + i] = a[i];
    }
    __syncthreads();
    for (int e = tid; e < G * MX_NS; e += 256) {   // element (g, i): the parts in order
      float sum = 0.f;
      for (int q = 0; q < parts; ++q) sum += rb[q * G * MX_NS + e];
      const int gg = e / MX_NS, i = e - gg * MX_NS;
      float* o = p.dW0 + (size_t)(n0 + i) * G + gg;
      *o = p.accumulate ? *o + sum : sum;
    }
  }
  MX_STAMP(10);
  {                                              // db1 / db0 of the slice: 16 lanes per (array, feature), rows interleaved
    constexpr int NSUM = 2 * MX_NS;              // 16 (or 8) sums
    const int sidx = tid >> 4, sub = tid & 15;   // 256 threads = 16 sums x 16 lanes
    float sacc = 0.f;
    if (sidx < NSUM) {
      const int i = sidx % MX_NS;
      const float* src = sidx < MX_NS ? d1s : d0s;
      for (int b = sub; b < B; b += 16) sacc += src[b * MX_NS + i];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
    if (sidx < NSUM && sub == 0) {
      const int i = sidx % MX_NS;
      float* o = (sidx < MX_NS ? p.db1 : p.db0) + n0 + i;
      *o = p.accumulate ? *o + sacc : sacc;
    }
  }
  MX_STAMP(11);
  if (tid < MX_NS) {                            // the shares of column n0 + tid in workgroup order, eight interleaved partial sums
    float cpart[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) cpart[u] = 0.f;
#pragma unroll
    for (int j = 0; j < MX_NW; ++j) cpart[j & 7] += cw[tid * MX_NW + j];
    const float s = ((cpart[0] + cpart[1]) + (cpart[2] + cpart[3])) + ((cpart[4] + cpart[5]) + (cpart[6] + cpart[7]));
    float* o = p.dWc + n0 + tid;
    *o = p.accumulate ? *o + s : s;
  }
  if (blockIdx.x == 0) {
    float sb = tid < B ? p.dr[tid] : 0.f;       // dbc = sum_b dr[b]
    sb = wave_sum(sb);
    if (lane == 0) red[8 + wave] = sb;
    __syncthreads();
    if (tid == 0) {
      const float t = (red[8] + red[9]) + (red[10] + red[11]);
      p.dbc[0] = p.accumulate ? p.dbc[0] + t : t;
    }
  }
  MX_STAMP(7);
  // the tick words go back to zero: the last workgroup to get here knows that everybody has passed both barriers
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(p.bar + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)(MX_NW - 1)) {
      __hip_atomic_store(p.bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.bar + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Workgroups of the launch: B / 4 rounded up to 32 or 64 (four rows each).  Measured: 64 workgroups at B = 128 -- phase 4, a
// third of the step, divides the 256 output features among however many there are, and the 32 workgroups without a batch row
// run phases 1 - 3 on zero rows beside the others -- 32.0 -> 34.4-35.3 us: twice the arrivals at both barriers and twice the
// workgroups streaming W1 cost more than half a phase 4 saves (MMF_MX_NW_MIN = 64 in the tuning build).
static int maxnet_step_workgroups(int B) {
  static const int nw_min = tune_int("MMF_MX_NW_MIN", 32);
  return (B > MX_R * 32 || nw_min > 32) ? 64 : 32;
}
int maxnet_step_dp_pitch(int B) { return maxnet_step_workgroups(B) * MX_R; }     // BP of the kernel the launcher picks

size_t maxnet_step_workspace_floats(int B) {
  return (size_t)2 * B * MX_H + (size_t)2 * MX_H * maxnet_step_dp_pitch(B) + (size_t)((B + 63) / 64 * 64) + 32 + (size_t)MX_NW_MAX * MX_H;
}

bool maxnet_step_ok(int B, int G, int H0, int H1) { return B >= 1 && B <= 256 && G >= 1 && G <= 256 && H0 == MX_H && H1 == MX_H; }

int launch_maxnet_cox_step(MaxnetStepParams p, hipStream_t st) {
  const int nw = maxnet_step_workgroups(p.B);
  const int lds = (256 * MX_R + 256 * MX_WP + 64) * (int)sizeof(float);
  auto kern = nw == 32 ? maxnet_cox_step_kernel<32> : maxnet_cox_step_kernel<64>;
  if (int e = set_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  ProfScope ps("maxnet_cox_step_kernel", st);
  hipLaunchKernelGGL(kern, dim3(nw), dim3(256), lds, st, p);
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

}  // namespace mmf

