// Omic head, one training step in ONE launch ("the goal is one launch").
//
//   MaxNet.forward      f = SNN(SNN(x)); risk = classifier(f).squeeze()      models/model_genomic.py:53-72
//     SNN_Block         Linear + SELU + AlphaDropout(0.25)                    models/model_modules.py:64-68
//   CoxSurvLoss         -mean((theta - log sum_j e^theta_j [t_j >= t_i]) (1 - c))   utils/loss_utils.py:124-139
//   and what autograd derives from them.
//
// The reference spends ~20 framework launches on this step (three addmm / selu / alpha-dropout forward, a Python double
// loop for the risk-set matrix, the same again backward); round 3 of this repo spent 9 (three dense forward, Cox, three
// dense backward, an H2D copy and a fill: 82 us of GPU time, 0.29 ms with the host's issue time).  Here: 32 workgroups.
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
constexpr int MX_NW = 32;            // workgroups
constexpr int MX_NS = MX_H / MX_NW;  // output features per workgroup in phase 4
constexpr int MX_WP = 65;            // LDS pitch of a staged weight chunk [256][64 + 1]


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

__device__ inline void mx_grid_barrier(unsigned* cnt) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                 // release: this workgroup's stores are visible device-wide
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)MX_NW) __builtin_amdgcn_s_sleep(1);
    __threadfence();                 // acquire
  }
  __syncthreads();
}

// acc[r] = sum_k xs[k][r] W[n = tid][k], k < K: W (row-major [256][ldw]) is staged through LDS in chunks of 64 k so that the
// global reads are coalesced along k and thread n's reads walk its own padded row; the next chunk travels in registers
// while this one is multiplied.  xs: [K][R] in LDS (R batch values of one k contiguous: one broadcast read per k).
template <int R>
__device__ inline void mx_rows_gemm(const float* __restrict__ xs, const float* __restrict__ W, int ldw, int K, float* wl,
                                    float (&acc)[R]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  float stage[64];
  auto fetch = [&](int k0) {         // thread t: elements e = t + 256 i of the [256][64] chunk, i.e. row e / 64, column e % 64
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const int e = tid + 256 * i, n = e >> 6, kk = e & 63;
      stage[i] = (k0 + kk < K) ? W[(size_t)n * ldw + k0 + kk] : 0.f;
    }
  };
  auto put = [&]() {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const int e = tid + 256 * i;
      wl[(e >> 6) * MX_WP + (e & 63)] = stage[i];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += 64) {
    __syncthreads();                 // the previous chunk has been read by everybody
    put();
    __syncthreads();
    if (k0 + 64 < K) fetch(k0 + 64);
    const int kn = K - k0 < 64 ? K - k0 : 64;
    const float* wr = wl + tid * MX_WP;
    if (kn == 64) {                  // a full chunk: 16 k at a time, their LDS reads issued together
#pragma unroll 1
      for (int kq = 0; kq < 64; kq += 16) {
        float w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = wr[kq + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float* xv = xs + (size_t)(k0 + kq + u) * R;
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r] += xv[r] * w[u];
        }
      }
    } else {
      for (int kk = 0; kk < kn; ++kk) {
        const float w = wr[kk];
        const float* xv = xs + (size_t)(k0 + kk) * R;
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] += xv[r] * w;
      }
    }
  }
}

#ifdef MMF_STAMPS
#define MX_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && p.stamps) p.stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define MX_STAMP(i)
#endif

template <int R>
__global__ __launch_bounds__(256) void maxnet_cox_step_kernel(MaxnetStepParams p) {
  extern __shared__ __align__(16) float sm[];
  MX_STAMP(0);
  float* xs = sm;                               // [256][R] layer input of this workgroup's rows
  float* wl = xs + 256 * R;                     // [256][65] staged weights; phases 2-4: scratch
  float* red = wl + 256 * MX_WP;                // [4][R] + misc
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, G = p.G;
  const int r0 = blockIdx.x * R;
  const uint32_t sdev = p.seed_dev ? *p.seed_dev : 0u;
  const MxDrop dr = mx_drop(p.p);
  const uint32_t key0 = p.key0 + sdev, key1 = p.key1 + sdev;

  // ---------------- phase 1: this workgroup's rows through the net ----------------------------------------------------
  for (int e = tid; e < G * R; e += 256) {
    const int k = e / R, r = e % R;
    xs[e] = (r0 + r < B) ? p.x[(size_t)(r0 + r) * G + k] : 0.f;
  }
  float acc[R], y0d[R], y1d[R];
  mx_rows_gemm<R>(xs, p.W0, G, G, wl, acc);
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
  __syncthreads();                              // every thread is done with xs as layer 0's input
#pragma unroll
  for (int r = 0; r < R; ++r) xs[tid * R + r] = y0d[r];
  mx_rows_gemm<R>(xs, p.W1, MX_H, MX_H, wl, acc);
  float part[R];
  {
    const float b = p.b1[tid], wc = p.Wc[tid];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float y = mx_selu(acc[r] + b);
      const uint32_t idx = (uint32_t)(r0 + r) * MX_H + (uint32_t)tid;
      y1d[r] = dr.on ? dr.a * (keep(key1, idx, dr.thr) ? y : dr.alpha_p) + dr.b : y;
      if (r0 + r < B) p.y1[(size_t)(r0 + r) * MX_H + tid] = y1d[r];
      part[r] = wave_sum(y1d[r] * wc);
    }
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) red[wave * R + r] = part[r];
  }
  __syncthreads();
  if (tid < R && r0 + tid < B) p.risk[r0 + tid] = red[tid] + red[R + tid] + red[2 * R + tid] + red[3 * R + tid] + p.bc[0];
  MX_STAMP(2);
  mx_grid_barrier(p.bar);
  MX_STAMP(3);

  // ---------------- phase 2: Cox over the whole batch, gradient of this workgroup's rows ---------------------------------
  float* et = wl;                               // [B] e^theta
  float* wq = wl + 256;                         // [B] (1 - c_i) / D_i
  float* lp = wl + 512;                         // [256] loss terms (workgroup 0)
  double* tl = reinterpret_cast<double*>(wl + 768);     // [B] event times
  for (int i = tid; i < B; i += 256) { et[i] = expf(p.risk[i]); tl[i] = p.times[i]; }
  __syncthreads();
  float lterm = 0.f;
  for (int i = tid; i < B; i += 256) {
    const double ti = tl[i];
    float Di = 0.f;
    for (int j = 0; j < B; ++j) Di += (tl[j] >= ti) ? et[j] : 0.f;
    const float unc = 1.f - p.c[i];
    lterm += (p.risk[i] - logf(Di)) * unc;
    wq[i] = unc / Di;
  }
  lp[tid] = lterm;
  __syncthreads();
  const float invB = 1.0f / (float)B;
  if (tid < R) {
    float g = 0.f;
    const int k = r0 + tid;
    if (k < B) {
      const double tk = tl[k];
      float a = 0.f;
      for (int i = 0; i < B; ++i) a += (tk >= tl[i]) ? wq[i] : 0.f;
      g = -invB * ((1.f - p.c[k]) - et[k] * a) * p.loss_scale;
      p.dr[k] = g;
    }
    red[tid] = g;
  }
  if (blockIdx.x == 0 && tid == 0) {
    float s = 0.f;
    for (int i = 0; i < 256; ++i) s += lp[i];
    p.loss[0] = -s * invB;
  }
  __syncthreads();

  MX_STAMP(4);
  // ---------------- phase 3: d pre-activations of this workgroup's rows --------------------------------------------------
  float* dps = xs;                              // [256][R] dpre1 of these rows (layer 1's outputs n)
  {
    const float wc = p.Wc[tid];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t idx = (uint32_t)(r0 + r) * MX_H + (uint32_t)tid;
      float dydy = 1.f, y = y1d[r];
      if (dr.on) {
        const bool kp = keep(key1, idx, dr.thr);
        dydy = kp ? dr.a : 0.f;
        y = kp ? (y1d[r] - dr.b) / dr.a : 0.f;
      }
      const float d = red[r] * wc * dydy * mx_selu_grad_from_y(y);
      dps[tid * R + r] = d;
      if (r0 + r < B) p.dp1[(size_t)(r0 + r) * MX_H + tid] = d;
    }
  }
  __syncthreads();
  {
    float a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = 0.f;
    // dy0[r][k = tid] = sum_n dpre1[r][n] W1[n][k]: W1 is read as it lies (row n, coalesced along k), 16 rows per stage and the
    // next stage's 16 loads in flight while this one multiplies (the loop is a chain of memory round trips otherwise)
    float w[2][16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w[0][u] = p.W1[(size_t)u * MX_H + tid];
#pragma unroll 1
    for (int n0 = 0; n0 < MX_H; n0 += 32) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int nb = n0 + 16 * h, nn = nb + 16;
        if (nn < MX_H) {
#pragma unroll
          for (int u = 0; u < 16; ++u) w[h ^ 1][u] = p.W1[(size_t)(nn + u) * MX_H + tid];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float* dv = dps + (nb + u) * R;
#pragma unroll
          for (int r = 0; r < R; ++r) a[r] += dv[r] * w[h][u];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t idx = (uint32_t)(r0 + r) * MX_H + (uint32_t)tid;
      float dydy = 1.f, y = y0d[r];
      if (dr.on) {
        const bool kp = keep(key0, idx, dr.thr);
        dydy = kp ? dr.a : 0.f;
        y = kp ? (y0d[r] - dr.b) / dr.a : 0.f;
      }
      if (r0 + r < B) p.dp0[(size_t)(r0 + r) * MX_H + tid] = a[r] * dydy * mx_selu_grad_from_y(y);
    }
  }
  MX_STAMP(5);
  mx_grid_barrier(p.bar + 1);
  MX_STAMP(6);

  // ---------------- phase 4: weight gradients, 8 output features per workgroup ------------------------------------------
  const int n0 = blockIdx.x * MX_NS;
  float* d1s = wl;                              // [B][8] dpre1[:, n0 .. n0 + 7]
  float* d0s = wl + 8 * 256;                    // [B][8] dpre0[:, n0 .. n0 + 7]   (B <= 256)
  for (int e = tid; e < B * MX_NS; e += 256) {
    const int b = e / MX_NS, i = e % MX_NS;
    d1s[e] = p.dp1[(size_t)b * MX_H + n0 + i];
    d0s[e] = p.dp0[(size_t)b * MX_H + n0 + i];
  }
  __syncthreads();
  {
    float a[MX_NS];
#pragma unroll
    for (int i = 0; i < MX_NS; ++i) a[i] = 0.f;
    // dW1[n0 + i][k = tid] = sum_b dpre1[b][n0 + i] y0[b][k]: 16 batch rows per stage, the next stage in flight; rows beyond
    // the batch multiply zeros of d1s' padding... they are simply not loaded (yv = 0)
    float yv[2][16];
    auto ld = [&](int b0, float (&v)[16]) {
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = b0 + u < B ? p.y0[(size_t)(b0 + u) * MX_H + tid] : 0.f;
    };
    ld(0, yv[0]);
#pragma unroll 1
    for (int b0 = 0; b0 < B; b0 += 32) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int bb = b0 + 16 * h;
        if (bb >= B) break;
        if (bb + 16 < B) ld(bb + 16, yv[h ^ 1]);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (bb + u >= B) break;
          const float* dv = d1s + (bb + u) * MX_NS;
#pragma unroll
          for (int i = 0; i < MX_NS; ++i) a[i] += dv[i] * yv[h][u];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MX_NS; ++i) {
      float* o = p.dW1 + (size_t)(n0 + i) * MX_H + tid;
      *o = p.accumulate ? *o + a[i] : a[i];
    }
  }
  if (tid < G) {                                 // dW0[n0 + i][g = tid] = sum_b dpre0[b][n0 + i] x[b][g]   (G <= 256)
    float a[MX_NS];
#pragma unroll
    for (int i = 0; i < MX_NS; ++i) a[i] = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) {
      const float xv = p.x[(size_t)b * G + tid];
      const float* dv = d0s + b * MX_NS;
#pragma unroll
      for (int i = 0; i < MX_NS; ++i) a[i] += dv[i] * xv;
    }
#pragma unroll
    for (int i = 0; i < MX_NS; ++i) {
      float* o = p.dW0 + (size_t)(n0 + i) * G + tid;
      *o = p.accumulate ? *o + a[i] : a[i];
    }
  }
  if (tid < 2 * MX_NS) {                         // db1 / db0 of the slice
    const int i = tid % MX_NS;
    const float* src = tid < MX_NS ? d1s : d0s;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += src[b * MX_NS + i];
    float* o = (tid < MX_NS ? p.db1 : p.db0) + n0 + i;
    *o = p.accumulate ? *o + s : s;
  }
  if (blockIdx.x == 0) {                         // classifier: dWc[k] = sum_b dr[b] y1[b][k], dbc = sum_b dr[b]
    float s = 0.f, sb = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) {
      const float g = p.dr[b];
      s += g * p.y1[(size_t)b * MX_H + tid];
      sb += g;
    }
    p.dWc[tid] = p.accumulate ? p.dWc[tid] + s : s;
    if (tid == 0) p.dbc[0] = p.accumulate ? p.dbc[0] + sb : sb;
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

size_t maxnet_step_workspace_floats(int B) { return (size_t)4 * B * MX_H + (size_t)((B + 63) / 64 * 64) + 16; }

bool maxnet_step_ok(int B, int G, int H0, int H1) { return B >= 1 && B <= 256 && G >= 1 && G <= 256 && H0 == MX_H && H1 == MX_H; }

int launch_maxnet_cox_step(MaxnetStepParams p, hipStream_t st) {
  const int R = p.B <= 4 * MX_NW ? 4 : 8;
  const int lds = (256 * R + 256 * MX_WP + 64) * (int)sizeof(float);
  auto kern = R == 4 ? maxnet_cox_step_kernel<4> : maxnet_cox_step_kernel<8>;
  if (int e = set_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return e;
  ProfScope ps("maxnet_cox_step_kernel", st);
  hipLaunchKernelGGL(kern, dim3(MX_NW), dim3(256), lds, st, p);
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

}  // namespace mmf

