// Latency-bound single-workgroup kernels of the path: survival head, nll_surv and Cox losses.
// The goal here is ONE launch each, not FLOPs.
//
//   surv_head   logits = f.Wk^T + bk ; hazards = sigmoid ; S = cumprod(1 - hazards) ; Y_hat = argmax
//               models/model_attention_mil_path.py:58-61
//   nll_surv    utils/loss_utils.py:22-39   (loss + d/dhazards + d/dS in the same launch)
//   cox         utils/loss_utils.py:124-139 (risk-set mask built on device, no O(B^2) host loop)
#include "mmf_common.h"
#include "mmf_kernels.h"
#include "mmf_small.h"

namespace mmf {

constexpr int HEAD_MAX_BK = 256;   // B*K values kept in LDS

__global__ __launch_bounds__(256) void surv_head_fwd_kernel(HeadParams p) {
  __shared__ float z[HEAD_MAX_BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int BK = p.B * p.K;
  for (int o = wave; o < BK; o += 4) {
    int b = o / p.K, k = o - b * p.K;
    const float* f = p.feat + (size_t)b * p.F;
    const float* w = p.Wk + (size_t)k * p.F;
    float acc = 0.f;
    for (int c = lane; c < p.F; c += 64) acc += f[c] * w[c];
    acc = wave_sum(acc);
    if (lane == 0) z[o] = acc + p.bk[k];
  }
  __syncthreads();
  if (tid < p.B) {
    const int b = tid;
    float run = 1.f, best = -INFINITY;
    int arg = 0;
    for (int k = 0; k < p.K; ++k) {
      float zz = z[b * p.K + k];
      float hz = 1.0f / (1.0f + expf(-zz));
      run *= (1.0f - hz);
      p.logits[b * p.K + k] = zz;
      p.hazards[b * p.K + k] = hz;
      p.S[b * p.K + k] = run;
      if (zz > best) { best = zz; arg = k; }
    }
    p.Y_hat[b] = arg;
  }
}

// dL/dz from (gH = dL/dhazards, gS = dL/dS): S_j = prod_{u<=j}(1-h_u)
//   dL/dh_t = gH_t - sum_{j>=t} gS_j prod_{u<=j, u!=t}(1-h_u) ;  dz_t = dL/dh_t h_t (1-h_t)
__global__ __launch_bounds__(256) void surv_head_bwd_kernel(HeadBwdParams p) {
  __shared__ float dz[HEAD_MAX_BK];
  const int tid = threadIdx.x;
  const int BK = p.B * p.K;
  if (tid < BK) {
    int b = tid / p.K, t = tid - b * p.K;
    const float* hz = p.hazards + b * p.K;
    float g = p.gH ? p.gH[tid] : 0.f;
    if (p.gS) {
      for (int j = t; j < p.K; ++j) {
        float prod = 1.f;
        for (int u = 0; u <= j; ++u)
          if (u != t) prod *= (1.0f - hz[u]);
        g -= p.gS[b * p.K + j] * prod;
      }
    }
    dz[tid] = g * hz[t] * (1.0f - hz[t]);
  }
  __syncthreads();
  for (int c = tid; c < p.F; c += 256) {
    for (int b = 0; b < p.B; ++b) {
      float acc = 0.f;
      for (int k = 0; k < p.K; ++k) acc += dz[b * p.K + k] * p.Wk[(size_t)k * p.F + c];
      p.dfeat[(size_t)b * p.F + c] = acc;
    }
    for (int k = 0; k < p.K; ++k) {
      float acc = 0.f;
      for (int b = 0; b < p.B; ++b) acc += dz[b * p.K + k] * p.feat[(size_t)b * p.F + c];
      p.dWk[(size_t)k * p.F + c] = acc;
    }
  }
  if (tid < p.K) {
    float acc = 0.f;
    for (int b = 0; b < p.B; ++b) acc += dz[b * p.K + tid];
    p.dbk[tid] = acc;
  }
}

// utils/loss_utils.py:22-39.  One thread per sample; gradients w.r.t. BOTH inputs (hazards and S are
// separate autograd inputs of the loss in the reference, S being produced by the model).
__global__ __launch_bounds__(256) void nll_surv_kernel(NllParams p) {
  __shared__ float part[256];
  const int tid = threadIdx.x;
  float l = 0.f;
  const float invB = 1.0f / (float)p.B;
  for (int b = tid; b < p.B; b += 256) {
    const float* hz = p.hazards + b * p.K;
    const float* S = p.S + b * p.K;
    float* gH = p.gH + b * p.K;
    float* gS = p.gS + b * p.K;
    for (int k = 0; k < p.K; ++k) { gH[k] = 0.f; gS[k] = 0.f; }
    const int64_t y64 = p.Y[b];
    if (y64 < 0 || y64 >= p.K) {        // the reference's gather raises; here: NaN loss, zero gradients, no stray access
      l = __builtin_nanf("");
      continue;
    }
    const int y = (int)y64;
    const float c = p.c[b];
    const float sp_y = y == 0 ? 1.0f : S[y - 1];      // S_padded[y]
    const float hy = hz[y];
    float unc = -(1.f - c) * (logf(fmaxf(sp_y, p.eps)) + logf(fmaxf(hy, p.eps)));
    if (y > 0 && sp_y >= p.eps) gS[y - 1] += -(1.f - c) / sp_y * invB;
    if (hy >= p.eps) gH[y] += -(1.f - c) / hy * invB;
    float cen = 0.f;
    if (y + 1 <= p.K) {                                 // S_padded[y+1] = S[y]
      const float sp_y1 = S[y];
      cen = -c * logf(fmaxf(sp_y1, p.eps));
      if (sp_y1 >= p.eps) gS[y] += -(1.f - p.alpha) * c / sp_y1 * invB;
    }
    l += (1.f - p.alpha) * (cen + unc) + p.alpha * unc;
  }
  part[tid] = l;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < 256; ++i) s += part[i];
    p.loss[0] = s * invB;
  }
}

// utils/loss_utils.py:124-139 (Appendix A.5).  fp32, no max-subtraction, as the reference.
__global__ __launch_bounds__(256) void cox_kernel(CoxParams p) {
  extern __shared__ float sm[];          // exp_theta[B], w[B] = (1-c_i)/D_i
  float* et = sm;
  float* w = sm + p.B;
  __shared__ float part[256];
  const int tid = threadIdx.x;
  for (int i = tid; i < p.B; i += 256) et[i] = expf(p.risks[i]);
  __syncthreads();
  float l = 0.f;
  for (int i = tid; i < p.B; i += 256) {
    const double ti = p.times[i];
    float Di = 0.f;
    for (int j = 0; j < p.B; ++j) Di += (p.times[j] >= ti) ? et[j] : 0.f;
    const float unc = 1.f - p.c[i];
    l += (p.risks[i] - logf(Di)) * unc;
    w[i] = unc / Di;
  }
  part[tid] = l;
  __syncthreads();
  const float invB = 1.0f / (float)p.B;
  for (int k = tid; k < p.B; k += 256) {
    const double tk = p.times[k];
    float acc = 0.f;
    for (int i = 0; i < p.B; ++i) acc += (tk >= p.times[i]) ? w[i] : 0.f;
    p.drisks[k] = -invB * ((1.f - p.c[k]) - et[k] * acc);
  }
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < 256; ++i) s += part[i];
    p.loss[0] = -s * invB;
  }
}

// ---------------------------------------------------------------------------------------------
// Per-step tail: l1_reg_all's gradient + torch.optim.Adam(weight_decay) in ONE pass over the flat
// parameter / gradient buffers (utils/utils.py:144-146,249-257; utils/core_utils.py:216-219,242-247).
//   g' = g + l1 * sign(w) + wd * w ; m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2
//   w -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)          (torch's Adam, amsgrad off)
// `l1` = lambda_reg x (micro-batches accumulated): the reference adds lambda*|W|_1 un-divided to every micro-batch's
// loss, so autograd would have added lambda*sign(w) that many times.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_l1_kernel(AdamParams p) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= p.n) return;
  auto upd = [&](float w, float g, float& m, float& v, float l1) {
    const float sg = w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f);
    const float gg = g + l1 * sg + p.wd * w;
    m = p.b1 * m + (1.f - p.b1) * gg;
    v = p.b2 * v + (1.f - p.b2) * gg * gg;
    const float denom = sqrtf(v) / p.bc2_sqrt + p.eps;
    return w - p.step_size * (m / denom);
  };
  if (i4 + 3 < p.n) {
    float4 w = ld4(p.w + i4), g = ld4(p.g + i4), m = ld4(p.m + i4), v = ld4(p.v + i4);
    float4 k = make_float4(p.l1, p.l1, p.l1, p.l1);
    if (p.l1_mask) { const float4 q = ld4(p.l1_mask + i4); k.x *= q.x; k.y *= q.y; k.z *= q.z; k.w *= q.w; }
    w.x = upd(w.x, g.x, m.x, v.x, k.x); w.y = upd(w.y, g.y, m.y, v.y, k.y);
    w.z = upd(w.z, g.z, m.z, v.z, k.z); w.w = upd(w.w, g.w, m.w, v.w, k.w);
    st4(p.w + i4, w); st4(p.m + i4, m); st4(p.v + i4, v);
  } else {
    for (int64_t i = i4; i < p.n; ++i) {
      float m = p.m[i], v = p.v[i];
      p.w[i] = upd(p.w[i], p.g[i], m, v, p.l1_mask ? p.l1 * p.l1_mask[i] : p.l1);
      p.m[i] = m; p.v[i] = v;
    }
  }
}

// sum_i |w_i| (the value of l1_reg_all, for logging): per-block partials, then one block; fixed order
__global__ __launch_bounds__(256) void abs_sum_kernel(const float* w, int64_t n, float* partials, int nblocks, float* out) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float acc = 0.f;
  if (out == nullptr) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) acc += fabsf(w[i]);
  } else {
    for (int i = tid; i < nblocks; i += 256) acc += partials[i];
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    float s = red[0] + red[1] + red[2] + red[3];
    if (out) out[0] = s; else partials[blockIdx.x] = s;
  }
}

int launch_adam_l1(AdamParams p, hipStream_t st) {
  const int64_t blocks = (p.n + 1023) / 1024;
  { ProfScope ps("adam_l1_kernel", st); hipLaunchKernelGGL(adam_l1_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}
int launch_abs_sum(const float* w, int64_t n, float* partials, float* out, hipStream_t st) {
  const int nb = 512;
  { ProfScope ps("abs_sum_kernel", st);
    hipLaunchKernelGGL(abs_sum_kernel, dim3(nb), dim3(256), 0, st, w, n, partials, nb, (float*)nullptr);
    hipLaunchKernelGGL(abs_sum_kernel, dim3(1), dim3(256), 0, st, w, n, partials, nb, out); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

int launch_head_fwd(HeadParams p, hipStream_t st) {
  if (p.B * p.K > HEAD_MAX_BK || p.B > 256) return MMF_ERR_SHAPE;
  { ProfScope ps("surv_head_fwd_kernel", st); hipLaunchKernelGGL(surv_head_fwd_kernel, dim3(1), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}
int launch_head_bwd(HeadBwdParams p, hipStream_t st) {
  if (p.B * p.K > HEAD_MAX_BK) return MMF_ERR_SHAPE;
  { ProfScope ps("surv_head_bwd_kernel", st); hipLaunchKernelGGL(surv_head_bwd_kernel, dim3(1), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}
int launch_nll(NllParams p, hipStream_t st) {
  { ProfScope ps("nll_surv_kernel", st); hipLaunchKernelGGL(nll_surv_kernel, dim3(1), dim3(256), 0, st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}
int launch_cox(CoxParams p, hipStream_t st) {
  if (p.B > 8192) return MMF_ERR_SHAPE;
  { ProfScope ps("cox_kernel", st); hipLaunchKernelGGL(cox_kernel, dim3(1), dim3(256), 2 * p.B * sizeof(float), st, p); }
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

}  // namespace mmf
