// Small dense layers and the Kronecker fusion block: odd dimensions (omic G = 36/80/186, 17^m = 4913),
// batch 1..128.  Latency-bound work; no MFMA (no dimension is tile-shaped), every kernel is a single
// short launch spread over as many waves as there are outputs.
//
//   dense_fwd    y = drop(act(x.W^T + b))          models/model_modules.py:64-68 (SNN_Block), :133-152 (XlinearFusion
//                                                  Linear+ReLU+Dropout stacks), model_mm_attention_mil.py:91,95 (classifier)
//   dense_bwd    dpre = dy . drop' . act'(y) ; dx = dpre.W ; dW = dpre^T.x ; db = colsum(dpre)
//   gate_mul     o = sigmoid(z) * h                models/model_modules.py:163 (information gating)
//   kron         [o1,1] x [o2,1] (x [o3,1]) + Dropout    models/model_modules.py:164-171
#include "mmf_common.h"
#include "mmf_kernels.h"
#include "mmf_mlp.h"

namespace mmf {

constexpr float SELU_ALPHA = 1.6732632423543772f;
constexpr float SELU_SCALE = 1.0507009873554805f;

__device__ inline float act_fwd(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_TANH: return tanhf(v);
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case ACT_SELU: return SELU_SCALE * (v > 0.f ? v : SELU_ALPHA * (expf(v) - 1.0f));
    default: return v;
  }
}
// derivative expressed through the activation OUTPUT y (so only y is saved)
__device__ inline float act_grad_from_y(float y, int act) {
  switch (act) {
    case ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case ACT_TANH: return 1.f - y * y;
    case ACT_SIGMOID: return y * (1.f - y);
    case ACT_SELU: return y > 0.f ? SELU_SCALE : y + SELU_SCALE * SELU_ALPHA;
    default: return 1.f;
  }
}

// dropout forms: 1 = nn.Dropout (scale 1/(1-p)); 2 = nn.AlphaDropout (affine, SNN)
struct DropAffine { float a, b, alpha_p; };
__host__ __device__ inline DropAffine alpha_affine(float p) {
  DropAffine d;
  d.alpha_p = -SELU_ALPHA * SELU_SCALE;
  d.a = 1.0f / sqrtf((d.alpha_p * d.alpha_p * p + 1.0f) * (1.0f - p));
  d.b = -d.a * d.alpha_p * p;
  return d;
}
__device__ inline float drop_fwd(float y, const DropSpec& d, uint32_t idx) {
  if (d.kind == 0) return y;
  const bool k = keep(d.key + (d.dev ? *d.dev : 0u), idx, drop_threshold(d.p));
  if (d.kind == 1) return k ? y / (1.0f - d.p) : 0.f;
  DropAffine af = alpha_affine(d.p);
  return af.a * (k ? y : af.alpha_p) + af.b;
}
// (d out / d y, and y recovered from the dropped output) for backward
__device__ inline void drop_bwd(float yd, const DropSpec& d, uint32_t idx, float& dydy, float& y) {
  if (d.kind == 0) { dydy = 1.f; y = yd; return; }
  const bool k = keep(d.key + (d.dev ? *d.dev : 0u), idx, drop_threshold(d.p));
  if (d.kind == 1) { dydy = k ? 1.0f / (1.0f - d.p) : 0.f; y = k ? yd * (1.0f - d.p) : 0.f; return; }
  DropAffine af = alpha_affine(d.p);
  dydy = k ? af.a : 0.f;
  y = k ? (yd - af.b) / af.a : 0.f;
}

// one wave per output element (b, n); lanes stride over K
__global__ __launch_bounds__(256) void dense_fwd_kernel(DenseParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t total = (int64_t)p.B * p.N;
  for (int64_t o = (int64_t)blockIdx.x * 4 + wave; o < total; o += (int64_t)gridDim.x * 4) {
    const int b = (int)(o / p.N), n = (int)(o - (int64_t)b * p.N);
    const float* xr = p.x + (size_t)b * p.K;
    const float* wr = p.W + (size_t)n * p.K;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int k = lane;
    for (; k + 192 < p.K; k += 256) {
      a0 += xr[k] * wr[k]; a1 += xr[k + 64] * wr[k + 64];
      a2 += xr[k + 128] * wr[k + 128]; a3 += xr[k + 192] * wr[k + 192];
    }
    for (; k < p.K; k += 64) a0 += xr[k] * wr[k];
    float acc = wave_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) {
      float y = act_fwd(acc + (p.bias ? p.bias[n] : 0.f), p.act);
      p.y[o] = drop_fwd(y, p.drop, (uint32_t)o);
    }
  }
}

__global__ __launch_bounds__(256) void dense_dpre_kernel(DenseBwdParams p) {
  const int64_t total = (int64_t)p.B * p.N;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    float dydy, y;
    drop_bwd(p.y[o], p.drop, (uint32_t)o, dydy, y);
    p.dpre[o] = p.dy[o] * dydy * act_grad_from_y(y, p.act);
  }
}

// dx[b][k] = sum_n dpre[b][n] W[n][k]   (threads along k: coalesced W rows)
__global__ __launch_bounds__(256) void dense_dx_kernel(DenseBwdParams p) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (k >= p.K) return;
  const float* dp = p.dpre + (size_t)b * p.N;
  float a0 = 0.f, a1 = 0.f;
  int n = 0;
  for (; n + 1 < p.N; n += 2) {
    a0 += dp[n] * p.W[(size_t)n * p.K + k];
    a1 += dp[n + 1] * p.W[(size_t)(n + 1) * p.K + k];
  }
  if (n < p.N) a0 += dp[n] * p.W[(size_t)n * p.K + k];
  p.dx[(size_t)b * p.K + k] = a0 + a1;
}

// dW[n][k] = sum_b dpre[b][n] x[b][k] ; db[n] = sum_b dpre[b][n]
__global__ __launch_bounds__(256) void dense_dw_kernel(DenseBwdParams p) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (k < p.K) {
    float acc = 0.f;
    for (int b = 0; b < p.B; ++b) acc += p.dpre[(size_t)b * p.N + n] * p.x[(size_t)b * p.K + k];
    p.dW[(size_t)n * p.K + k] = acc;
  }
  if (p.db && blockIdx.x == 0 && threadIdx.x == 0) {
    float acc = 0.f;
    for (int b = 0; b < p.B; ++b) acc += p.dpre[(size_t)b * p.N + n];
    p.db[n] = acc;
  }
}

__global__ __launch_bounds__(256) void gate_mul_fwd_kernel(const float* z, const float* h, float* o, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) o[i] = (1.0f / (1.0f + expf(-z[i]))) * h[i];
}
__global__ __launch_bounds__(256) void gate_mul_bwd_kernel(const float* g, const float* z, const float* h,
                                                           float* dz, float* dh, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    float s = 1.0f / (1.0f + expf(-z[i]));
    dz[i] = g[i] * h[i] * s * (1.f - s);
    dh[i] = g[i] * s;
  }
}

// out[b][(i*S + j)*S + k] = o1'[i] o2'[j] o3'[k], o' = [o, 1], S = dim + 1; m = 2 or 3 operands
__global__ __launch_bounds__(256) void kron_fwd_kernel(KronParams p) {
  const int S = p.dim + 1;
  const int total = p.m == 3 ? S * S * S : S * S;
  const int b = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  int i, j, k = 0;
  if (p.m == 3) { i = e / (S * S); j = (e / S) % S; k = e % S; } else { i = e / S; j = e % S; }
  auto val = [&](const float* o, int t) { return t < p.dim ? o[(size_t)b * p.dim + t] : 1.f; };
  float v = val(p.o[0], i) * val(p.o[1], j);
  if (p.m == 3) v *= val(p.o[2], k);
  p.out[(size_t)b * total + e] = drop_fwd(v, p.drop, (uint32_t)(b * total + e));
}

// d o_t[i] = sum over the other indices of g . (product of the other operands) . drop'
__global__ __launch_bounds__(64) void kron_bwd_kernel(KronParams p) {
  const int S = p.dim + 1;
  const int total = p.m == 3 ? S * S * S : S * S;
  const int b = blockIdx.y, t = blockIdx.z, i = blockIdx.x;   // operand t, component i < dim
  const int lane = threadIdx.x;
  auto val = [&](const float* o, int u) { return u < p.dim ? o[(size_t)b * p.dim + u] : 1.f; };
  const int others = p.m == 3 ? S * S : S;
  float acc = 0.f;
  for (int q = lane; q < others; q += 64) {
    int e;
    float prod;
    if (p.m == 3) {
      const int q0 = q / S, q1 = q % S;
      int i0, i1, i2;
      if (t == 0) { i0 = i; i1 = q0; i2 = q1; }
      else if (t == 1) { i0 = q0; i1 = i; i2 = q1; }
      else { i0 = q0; i1 = q1; i2 = i; }
      e = (i0 * S + i1) * S + i2;
      const float v0 = val(p.o[0], i0), v1 = val(p.o[1], i1), v2 = val(p.o[2], i2);
      prod = t == 0 ? v1 * v2 : (t == 1 ? v0 * v2 : v0 * v1);
    } else {
      const int i0 = t == 0 ? i : q, i1 = t == 0 ? q : i;
      e = i0 * S + i1;
      prod = t == 0 ? val(p.o[1], i1) : val(p.o[0], i0);
    }
    float dydy, ydummy;
    drop_bwd(0.f, p.drop, (uint32_t)(b * total + e), dydy, ydummy);
    acc += p.g[(size_t)b * total + e] * dydy * prod;
  }
  acc = wave_sum(acc);
  if (lane == 0) p.d[t][(size_t)b * p.dim + i] = acc;
}

// ---------------------------------------------------------------------------------------------
// Fused gating stage (single workgroup; the whole stage is ~100 kFLOP: one launch instead of 4 per modality)
// ---------------------------------------------------------------------------------------------
constexpr int XR_MAX = 3 * 8 * 16;     // m * B * sdim values kept in LDS
__device__ inline DropSpec site_drop(const DropSpec& d, int i) {
  DropSpec r = d;
  r.key = d.key + 0x632BE5ABu * (uint32_t)i;
  return r;
}
// dot of two length-n vectors (n % 4 == 0, 16-byte aligned) across a wave; float4 loads, all issued before the adds
__device__ inline float wave_dot(const float* a, const float* b, int n, int lane) {
  float acc = 0.f;
#pragma unroll 4
  for (int k = 4 * lane; k < n; k += 256) {
    const float4 x = *reinterpret_cast<const float4*>(a + k), w = *reinterpret_cast<const float4*>(b + k);
    acc += x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w;
  }
  return acc;                         // caller reduces: several partial dots are reduced together
}

constexpr int XR_NT = 1024;           // one workgroup of 16 waves: the stage is latency-bound, so width = parallel loads

__global__ __launch_bounds__(XR_NT) void xreduce_fwd_kernel(XReduceParams p) {
  __shared__ float sh_h[XR_MAX], sh_z[XR_MAX], sh_gm[XR_MAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = XR_NT / 64;
  const int S = p.sdim, per = p.B * S, total = p.m * per;
  // h and z: one wave per output value, 4 outputs per wave in flight (independent loads, then 4 reductions)
  for (int o0 = wave; o0 < 2 * total; o0 += 4 * NW) {
    float acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = o0 + u * NW;
      acc[u] = 0.f;
      if (o < 2 * total) {
        const bool is_z = o >= total;
        const int q = is_z ? o - total : o;
        const int i = q / per, b = (q % per) / S, j = q % S;
        if (!is_z) {
          acc[u] = wave_dot(p.v[i] + (size_t)b * p.dim, p.Wh[i] + (size_t)j * p.dim, p.dim, lane);
        } else {
          for (int t = 0; t < p.m; ++t)      // v_cat = [v_0 | v_1 | ...]
            acc[u] += wave_dot(p.v[t] + (size_t)b * p.dim, p.Wz[i] + (size_t)j * p.m * p.dim + (size_t)t * p.dim, p.dim, lane);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = o0 + u * NW;
      const float r = wave_sum(acc[u]);
      if (o < 2 * total && lane == 0) {
        const bool is_z = o >= total;
        const int q = is_z ? o - total : o;
        const int i = q / per, j = q % S;
        if (is_z) sh_z[q] = r + p.bz[i][j];
        else sh_h[q] = fmaxf(r + p.bh[i][j], 0.f);
      }
    }
  }
  __syncthreads();
  for (int q = tid; q < total; q += XR_NT) {
    const int i = q / per, r = q % per;
    const float g = (1.0f / (1.0f + expf(-sh_z[q]))) * sh_h[q];
    sh_gm[q] = g;
    p.h[i][r] = sh_h[q]; p.z[i][r] = sh_z[q]; p.gm[i][r] = g;
  }
  __syncthreads();
  for (int q = tid; q < total; q += XR_NT) {
    const int i = q / per, r = q % per, b = r / S, j = r % S;
    float acc = p.bo[i][j];
#pragma unroll 16
    for (int t = 0; t < S; ++t) acc += sh_gm[i * per + b * S + t] * p.Wo[i][j * S + t];
    acc = fmaxf(acc, 0.f);
    p.o[i][r] = drop_fwd(acc, site_drop(p.drop, i), (uint32_t)r);
  }
}

__global__ __launch_bounds__(XR_NT) void xreduce_bwd_kernel(XReduceParams p) {
  __shared__ float sh_dpo[XR_MAX], sh_dgm[XR_MAX], sh_dz[XR_MAX], sh_dph[XR_MAX];
  const int tid = threadIdx.x;
  const int S = p.sdim, per = p.B * S, total = p.m * per, KZ = p.m * p.dim;
  for (int q = tid; q < total; q += XR_NT) {       // d(pre-activation of o)
    const int i = q / per, r = q % per;
    float dydy, y;
    drop_bwd(p.o[i][r], site_drop(p.drop, i), (uint32_t)r, dydy, y);
    sh_dpo[q] = p.d_o[i][r] * dydy * (y > 0.f ? 1.f : 0.f);
  }
  __syncthreads();
  for (int q = tid; q < total; q += XR_NT) {       // d gm = dpo . Wo ; then dz, dh
    const int i = q / per, r = q % per, b = r / S, t = r % S;
    float acc = 0.f;
#pragma unroll 16
    for (int j = 0; j < S; ++j) acc += sh_dpo[i * per + b * S + j] * p.Wo[i][j * S + t];
    sh_dgm[q] = acc;
    const float hv = p.h[i][r], sg = 1.0f / (1.0f + expf(-p.z[i][r]));
    sh_dz[q] = acc * hv * sg * (1.f - sg);
    sh_dph[q] = hv > 0.f ? acc * sg : 0.f;         // d(pre-activation of h)
  }
  __syncthreads();
  // small weight grads: dWo [S x S], dbo, dbh, dbz
  for (int q = tid; q < p.m * S * S; q += XR_NT) {
    const int i = q / (S * S), j = (q / S) % S, t = q % S;
    float acc = 0.f;
    for (int b = 0; b < p.B; ++b) acc += sh_dpo[i * per + b * S + j] * p.gm[i][b * S + t];
    p.dWo[i][j * S + t] = acc;
  }
  for (int q = tid; q < p.m * S; q += XR_NT) {
    const int i = q / S, j = q % S;
    float a = 0.f, bsum = 0.f, c = 0.f;
    for (int b = 0; b < p.B; ++b) { a += sh_dpo[i * per + b * S + j]; bsum += sh_dph[i * per + b * S + j]; c += sh_dz[i * per + b * S + j]; }
    p.dbo[i][j] = a; p.dbh[i][j] = bsum; p.dbz[i][j] = c;
  }
  // dWh [S x dim], dWz [S x m*dim]: outer products with v, threads along the long dimension (pure stores for B = 1)
  for (int q = tid; q < p.m * S * p.dim; q += XR_NT) {
    const int i = q / (S * p.dim), jk = q % (S * p.dim), j = jk / p.dim, k = jk % p.dim;
    float acc = 0.f;
    for (int b = 0; b < p.B; ++b) acc += sh_dph[i * per + b * S + j] * p.v[i][(size_t)b * p.dim + k];
    p.dWh[i][jk] = acc;
  }
  for (int q = tid; q < p.m * S * KZ; q += XR_NT) {
    const int i = q / (S * KZ), jk = q % (S * KZ), j = jk / KZ, k = jk % KZ, t = k / p.dim, kk = k % p.dim;
    float acc = 0.f;
    for (int b = 0; b < p.B; ++b) acc += sh_dz[i * per + b * S + j] * p.v[t][(size_t)b * p.dim + kk];
    p.dWz[i][jk] = acc;
  }
  // dv_t = dph_t . Wh_t + sum_i dz_i . Wz_i[:, t-th block]: (1 + m) * S independent, coalesced loads per output
  for (int q = tid; q < p.m * p.B * p.dim; q += XR_NT) {
    const int t = q / (p.B * p.dim), b = (q / p.dim) % p.B, k = q % p.dim;
    float acc = 0.f;
#pragma unroll 16
    for (int j = 0; j < S; ++j) acc += sh_dph[t * per + b * S + j] * p.Wh[t][(size_t)j * p.dim + k];
    for (int i = 0; i < p.m; ++i) {
      float a2 = 0.f;
#pragma unroll 16
      for (int j = 0; j < S; ++j) a2 += sh_dz[i * per + b * S + j] * p.Wz[i][(size_t)j * KZ + (size_t)t * p.dim + k];
      acc += a2;
    }
    p.dv[t][(size_t)b * p.dim + k] = acc;
  }
}

int launch_xreduce_fwd(XReduceParams p, hipStream_t st) {
  if (p.m * p.B * p.sdim > XR_MAX || p.m < 1 || p.m > 3) return MMF_ERR_SHAPE;
  if (p.dim % 4 != 0) return MMF_ERR_SHAPE;
  { ProfScope ps("xreduce_fwd_kernel", st); hipLaunchKernelGGL(xreduce_fwd_kernel, dim3(1), dim3(XR_NT), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}
int launch_xreduce_bwd(XReduceParams p, hipStream_t st) {
  if (p.m * p.B * p.sdim > XR_MAX || p.m < 1 || p.m > 3) return MMF_ERR_SHAPE;
  { ProfScope ps("xreduce_bwd_kernel", st); hipLaunchKernelGGL(xreduce_bwd_kernel, dim3(1), dim3(XR_NT), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

int launch_dense_fwd(DenseParams p, hipStream_t st) {
  const int64_t total = (int64_t)p.B * p.N;
  int blocks = cdiv(total, 4);
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  { ProfScope ps("dense_fwd_kernel", st); hipLaunchKernelGGL(dense_fwd_kernel, dim3(blocks), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

// One launch for the whole dense backward: blocks [0, nbx*B) compute dx rows, the rest compute dW rows (+ db);
// dpre = dy . drop' . act'(y) is rebuilt through LDS by whoever needs it instead of a separate pass
// (the layers are tiny: three launches of ~7 us each per layer were the cost, not the arithmetic).
constexpr int DENSE_MAX_N = 2048, DENSE_MAX_B = 256;
__device__ inline float dense_dpre_at(const DenseBwdParams& p, int b, int n) {
  const int64_t o = (int64_t)b * p.N + n;
  float dydy, y;
  drop_bwd(p.y[o], p.drop, (uint32_t)o, dydy, y);
  return p.dy[o] * dydy * act_grad_from_y(y, p.act);
}
__global__ __launch_bounds__(256) void dense_bwd_kernel(DenseBwdParams p, int nbx, int dx_blocks, int nbw) {
  __shared__ float sh[DENSE_MAX_N];
  const int tid = threadIdx.x;
  int id = blockIdx.x;
  if (id < dx_blocks) {                       // ---- dx[b][k] = sum_n dpre[b][n] W[n][k]
    // 64 k-columns x 4 n-slices per block: 4x more blocks and 4x shorter load chains than one thread per column
    // (the layer is latency-bound: B = 1, a handful of blocks, hundreds of dependent-address loads each)
    __shared__ float part[4][64];
    const int b = id / nbx, kl = tid & 63, sl = tid >> 6, k = (id % nbx) * 64 + kl;
    for (int n = tid; n < p.N; n += 256) sh[n] = dense_dpre_at(p, b, n);
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (k < p.K) {
      const int n0 = sl * ((p.N + 3) / 4), n1 = n0 + (p.N + 3) / 4 < p.N ? n0 + (p.N + 3) / 4 : p.N;
      int n = n0;
      for (; n + 3 < n1; n += 4) {
        a0 += sh[n] * p.W[(size_t)n * p.K + k];
        a1 += sh[n + 1] * p.W[(size_t)(n + 1) * p.K + k];
        a2 += sh[n + 2] * p.W[(size_t)(n + 2) * p.K + k];
        a3 += sh[n + 3] * p.W[(size_t)(n + 3) * p.K + k];
      }
      for (; n < n1; ++n) a0 += sh[n] * p.W[(size_t)n * p.K + k];
    }
    part[sl][kl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && k < p.K) p.dx[(size_t)b * p.K + k] = (part[0][kl] + part[1][kl]) + (part[2][kl] + part[3][kl]);
  } else {                                    // ---- dW[n][k] = sum_b dpre[b][n] x[b][k] ; db[n] = sum_b dpre[b][n]
    id -= dx_blocks;
    const int n = id / nbw, kb = id % nbw, k = kb * 256 + tid;
    if (tid < p.B) sh[tid] = dense_dpre_at(p, tid, n);
    __syncthreads();
    if (k < p.K) {
      float acc = 0.f;
      for (int b = 0; b < p.B; ++b) acc += sh[b] * p.x[(size_t)b * p.K + k];
      p.dW[(size_t)n * p.K + k] = acc;
    }
    if (p.db && kb == 0 && tid == 0) {
      float acc = 0.f;
      for (int b = 0; b < p.B; ++b) acc += sh[b];
      p.db[n] = acc;
    }
  }
}

int launch_dense_bwd(DenseBwdParams p, hipStream_t st) {
  const int64_t total = (int64_t)p.B * p.N;
  if (p.N <= DENSE_MAX_N && p.B <= DENSE_MAX_B) {
    const int nbx = cdiv(p.K, 64), nbw = cdiv(p.K, 256);        // dx: 64 columns per block; dW: 256 per block
    const int dx_blocks = p.dx ? nbx * p.B : 0;
    const int dw_blocks = p.dW ? nbw * p.N : 0;
    if (dx_blocks + dw_blocks == 0) return MMF_OK;
    { ProfScope ps("dense_bwd_kernel", st);
      hipLaunchKernelGGL(dense_bwd_kernel, dim3(dx_blocks + dw_blocks), dim3(256), 0, st, p, nbx, dx_blocks, nbw); }
    return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
  }
  { ProfScope ps("dense_dpre_kernel", st); hipLaunchKernelGGL(dense_dpre_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, p); }
  if (p.dx) { ProfScope ps("dense_dx_kernel", st); hipLaunchKernelGGL(dense_dx_kernel, dim3(cdiv(p.K, 256), p.B), dim3(256), 0, st, p); }
  if (p.dW) { ProfScope ps("dense_dw_kernel", st); hipLaunchKernelGGL(dense_dw_kernel, dim3(cdiv(p.K, 256), p.N), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

int launch_gate_mul(const float* z, const float* h, float* o, int n, hipStream_t st) {
  { ProfScope ps("gate_mul_fwd_kernel", st); hipLaunchKernelGGL(gate_mul_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, z, h, o, n); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}
int launch_gate_mul_bwd(const float* g, const float* z, const float* h, float* dz, float* dh, int n, hipStream_t st) {
  { ProfScope ps("gate_mul_bwd_kernel", st); hipLaunchKernelGGL(gate_mul_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, g, z, h, dz, dh, n); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}
int launch_kron_fwd(KronParams p, hipStream_t st) {
  const int S = p.dim + 1, total = p.m == 3 ? S * S * S : S * S;
  { ProfScope ps("kron_fwd_kernel", st); hipLaunchKernelGGL(kron_fwd_kernel, dim3(cdiv(total, 256), p.B), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}
int launch_kron_bwd(KronParams p, hipStream_t st) {
  { ProfScope ps("kron_bwd_kernel", st); hipLaunchKernelGGL(kron_bwd_kernel, dim3(p.dim, p.B, p.m), dim3(64), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}


// =============================================================================================
// Stage-2 building blocks: the embedding-level fusion models of
// models/nll_models_pretrained.py / models/coxranking_models_pretrained.py work on [B x 256] batches (B = 32): every
// op is latency-bound, so each is ONE small launch with its neighbours fused in.
// =============================================================================================
// BatchNorm1d (+ residual add, activation, dropout): one thread per feature, coalesced across features.
//   replaces nn.BatchNorm1d inside Highway (models/model_modules.py:13-14,18-19,26), the FCNN blocks
//   Linear-BatchNorm1d-ReLU-Dropout (models/nll_models_pretrained.py:82-90) and ResidualBlock (model_modules.py:29-49).
__global__ __launch_bounds__(256) void bn_fwd_kernel(BnParams p) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= p.F) return;
  float mean, invstd;
  if (p.training) {
    float s = 0.f;
    for (int b = 0; b < p.B; ++b) s += p.x[(size_t)b * p.F + f];
    mean = s / (float)p.B;
    float v = 0.f;
    for (int b = 0; b < p.B; ++b) { const float d = p.x[(size_t)b * p.F + f] - mean; v += d * d; }
    const float var = v / (float)p.B;                       // biased: what normalises the batch
    invstd = rsqrtf(var + p.eps);
    if (p.running_mean) {
      p.running_mean[f] = (1.f - p.momentum) * p.running_mean[f] + p.momentum * mean;
      p.running_var[f] = (1.f - p.momentum) * p.running_var[f] + p.momentum * (v / (float)(p.B - 1));   // unbiased
    }
  } else {
    mean = p.running_mean[f];
    invstd = rsqrtf(p.running_var[f] + p.eps);
  }
  p.save_mean[f] = mean;
  p.save_invstd[f] = invstd;
  const float g = p.gamma ? p.gamma[f] : 1.f, be = p.beta ? p.beta[f] : 0.f;
  for (int b = 0; b < p.B; ++b) {
    const size_t o = (size_t)b * p.F + f;
    float v = (p.x[o] - mean) * invstd * g + be;
    if (p.res) v += p.res[o];
    p.y[o] = drop_fwd(act_fwd(v, p.act), p.drop, (uint32_t)o);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_kernel(BnBwdParams p) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= p.F) return;
  const float mean = p.save_mean[f], invstd = p.save_invstd[f], g = p.gamma ? p.gamma[f] : 1.f;
  float sum_d = 0.f, sum_dx = 0.f;          // sum dpre, sum dpre * xhat
  for (int b = 0; b < p.B; ++b) {
    const size_t o = (size_t)b * p.F + f;
    float dydy, y;
    drop_bwd(p.y[o], p.drop, (uint32_t)o, dydy, y);
    const float dpre = p.dy[o] * dydy * act_grad_from_y(y, p.act);
    const float xhat = (p.x[o] - mean) * invstd;
    sum_d += dpre;
    sum_dx += dpre * xhat;
    if (p.dres) p.dres[o] = dpre;
  }
  if (p.dgamma) p.dgamma[f] = sum_dx;
  if (p.dbeta) p.dbeta[f] = sum_d;
  const float invB = 1.0f / (float)p.B;
  for (int b = 0; b < p.B; ++b) {
    const size_t o = (size_t)b * p.F + f;
    float dydy, y;
    drop_bwd(p.y[o], p.drop, (uint32_t)o, dydy, y);
    const float dpre = p.dy[o] * dydy * act_grad_from_y(y, p.act);
    const float xhat = (p.x[o] - mean) * invstd;
    p.dx[o] = p.training ? g * invstd * (dpre - invB * sum_d - xhat * invB * sum_dx) : g * invstd * dpre;
  }
}

int launch_bn_fwd(BnParams p, hipStream_t st) {
  if (p.B < 1 || p.F < 1 || (p.training && p.B < 2)) return MMF_ERR_SHAPE;    // torch: "Expected more than 1 value per channel"
  { ProfScope ps("bn_fwd_kernel", st); hipLaunchKernelGGL(bn_fwd_kernel, dim3(cdiv(p.F, 256)), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}
int launch_bn_bwd(BnBwdParams p, hipStream_t st) {
  if (p.B < 1 || p.F < 1) return MMF_ERR_SHAPE;
  { ProfScope ps("bn_bwd_kernel", st); hipLaunchKernelGGL(bn_bwd_kernel, dim3(cdiv(p.F, 256)), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

// Highway mix (models/model_modules.py:21-25): x = gate * f(nonlinear) + (1 - gate) * linear, gate = sigmoid, f = relu
__global__ __launch_bounds__(256) void highway_fwd_kernel(HighwayParams p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const float g = 1.0f / (1.0f + expf(-p.zg[i]));
  p.y[i] = g * fmaxf(p.zn[i], 0.f) + (1.f - g) * p.zl[i];
}
__global__ __launch_bounds__(256) void highway_bwd_kernel(HighwayParams p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const float g = 1.0f / (1.0f + expf(-p.zg[i])), zn = p.zn[i], dy = p.dy[i];
  p.dzg[i] = dy * (fmaxf(zn, 0.f) - p.zl[i]) * g * (1.f - g);
  p.dzn[i] = zn > 0.f ? dy * g : 0.f;
  p.dzl[i] = dy * (1.f - g);
}
int launch_highway_fwd(HighwayParams p, hipStream_t st) {
  { ProfScope ps("highway_fwd_kernel", st); hipLaunchKernelGGL(highway_fwd_kernel, dim3(cdiv(p.n, 256)), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}
int launch_highway_bwd(HighwayParams p, hipStream_t st) {
  { ProfScope ps("highway_bwd_kernel", st); hipLaunchKernelGGL(highway_bwd_kernel, dim3(cdiv(p.n, 256)), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

// Pairwise ranking loss (utils/loss_utils.py:58-101: a Python loop over all B(B-1)/2 pairs): one workgroup,
// thread i owns sample i and walks all j.  Pair {a < b}: a more risky if t_a < t_b and event_a; else b more risky if
// t_b < t_a and event_b; else not comparable.  loss = -(mean | sum) of phi(risk_more - risk_less); 0 if no pair.
__global__ __launch_bounds__(256) void rank_loss_kernel(RankParams p) {
  __shared__ float s_sum[256];
  __shared__ int s_cnt[256];
  const int tid = threadIdx.x;
  float sum = 0.f;
  int cnt = 0;
  for (int i = tid; i < p.B; i += 256) {
    float gi = 0.f;
    const double ti = p.times[i];
    const bool ei = (1.f - p.c[i]) != 0.f;
    const float ri = p.risks[i];
    for (int j = 0; j < p.B; ++j) {
      if (j == i) continue;
      const double tj = p.times[j];
      const bool ej = (1.f - p.c[j]) != 0.f;
      // index-ordered rule of the reference: a = min(i, j), b = max(i, j)
      const bool i_is_a = i < j;
      const double ta = i_is_a ? ti : tj, tb = i_is_a ? tj : ti;
      const bool ea = i_is_a ? ei : ej, eb = i_is_a ? ej : ei;
      int more;                                   // 0: a, 1: b, -1: not comparable
      if (ta < tb && ea) more = 0;
      else if (tb < ta && eb) more = 1;
      else continue;
      const bool i_more = (more == 0) == i_is_a;
      const float r = i_more ? ri - p.risks[j] : p.risks[j] - ri;
      float phi, dphi;
      if (p.phi == 0) { phi = 1.0f / (1.0f + expf(-r)); dphi = phi * (1.f - phi); }
      else { phi = fmaxf(r, 0.f); dphi = r > 0.f ? 1.f : 0.f; }
      gi += i_more ? dphi : -dphi;
      if (i_is_a) { sum += phi; ++cnt; }          // every unordered pair counted once
    }
    p.d_risks[i] = gi;                            // scaled below
  }
  s_sum[tid] = sum;
  s_cnt[tid] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { s_sum[tid] += s_sum[tid + o]; s_cnt[tid] += s_cnt[tid + o]; }
    __syncthreads();
  }
  const int n = s_cnt[0];
  const float scale = n == 0 ? 0.f : (p.reduction == 0 ? -1.0f / (float)n : -1.0f);
  if (tid == 0) p.loss[0] = n == 0 ? 0.f : s_sum[0] * scale;
  for (int i = tid; i < p.B; i += 256) p.d_risks[i] *= scale;
}
int launch_rank_loss(RankParams p, hipStream_t st) {
  if (p.B < 2) return MMF_ERR_SHAPE;              // the reference raises NotImplementedError for batch size 1
  { ProfScope ps("rank_loss_kernel", st); hipLaunchKernelGGL(rank_loss_kernel, dim3(1), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

// logits -> hazards = sigmoid, S = cumprod(1 - hazards), Y_hat = argmax, risk = -sum S
// (models/nll_models_pretrained.py:58-62,193-197); one thread per sample, K <= 32.
__global__ __launch_bounds__(256) void hazard_fwd_kernel(HazardParams p) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= p.B) return;
  float s = 1.f, rs = 0.f, best = -INFINITY;
  int arg = 0;
  for (int k = 0; k < p.K; ++k) {
    const float z = p.logits[(size_t)b * p.K + k];
    if (z > best) { best = z; arg = k; }
    const float h = 1.0f / (1.0f + expf(-z));
    s *= 1.f - h;
    p.hazards[(size_t)b * p.K + k] = h;
    p.S[(size_t)b * p.K + k] = s;
    rs += s;
  }
  if (p.Y_hat) p.Y_hat[b] = arg;
  if (p.risk) p.risk[b] = -rs;
}
__global__ __launch_bounds__(256) void hazard_bwd_kernel(HazardParams p) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= p.B) return;
  float h[32];
  for (int k = 0; k < p.K; ++k) h[k] = p.hazards[(size_t)b * p.K + k];
  const float gr = p.g_risk ? p.g_risk[b] : 0.f;
  for (int t = 0; t < p.K; ++t) {
    // d S_j / d h_t = -prod_{u <= j, u != t} (1 - h_u) for j >= t   (no division: 1 - h_t may underflow)
    float acc = 0.f, pre = 1.f;
    for (int u = 0; u < t; ++u) pre *= 1.f - h[u];
    float prod = pre;
    for (int j = t; j < p.K; ++j) {
      if (j > t) prod *= 1.f - h[j];
      const float gS = (p.g_S ? p.g_S[(size_t)b * p.K + j] : 0.f) - gr;
      acc -= gS * prod;
    }
    const float gh = (p.g_hazards ? p.g_hazards[(size_t)b * p.K + t] : 0.f) + acc;
    p.dlogits[(size_t)b * p.K + t] = gh * h[t] * (1.f - h[t]);
  }
}
int launch_hazard_fwd(HazardParams p, hipStream_t st) {
  if (p.B < 1 || p.K < 1 || p.K > 32) return MMF_ERR_SHAPE;
  { ProfScope ps("hazard_fwd_kernel", st); hipLaunchKernelGGL(hazard_fwd_kernel, dim3(cdiv(p.B, 256)), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}
int launch_hazard_bwd(HazardParams p, hipStream_t st) {
  if (p.B < 1 || p.K < 1 || p.K > 32) return MMF_ERR_SHAPE;
  { ProfScope ps("hazard_bwd_kernel", st); hipLaunchKernelGGL(hazard_bwd_kernel, dim3(cdiv(p.B, 256)), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

}  // namespace mmf
