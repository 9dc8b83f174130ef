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

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

int launch_dense_fwd(DenseParams p, hipStream_t st) {
  const int64_t total = (int64_t)p.B * p.N;
  int blocks = cdiv(total, 4);
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  { ProfScope ps("dense_fwd_kernel", st); hipLaunchKernelGGL(dense_fwd_kernel, dim3(blocks), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

int launch_dense_bwd(DenseBwdParams p, hipStream_t st) {
  const int64_t total = (int64_t)p.B * p.N;
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

}  // namespace mmf
