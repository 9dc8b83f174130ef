// Parameter blocks of the single-workgroup kernels (mmf_small.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmf {

struct HeadParams {
  const float* feat;   // [B x F]
  const float* Wk;     // [K x F]
  const float* bk;     // [K]
  int B, F, K;
  float *logits, *hazards, *S;   // [B x K]
  int64_t* Y_hat;      // [B]
};
struct HeadBwdParams {
  const float *gH, *gS;          // [B x K], either may be null
  const float* hazards;          // [B x K]
  const float* feat;             // [B x F]
  const float* Wk;               // [K x F]
  int B, F, K;
  float *dfeat, *dWk, *dbk;
};
struct NllParams {
  const float *hazards, *S;      // [B x K]
  const int64_t* Y;              // [B]
  const float* c;                // [B]
  int B, K;
  float alpha, eps;
  float *loss, *gH, *gS;
};
struct CoxParams {
  const float* risks;            // [B]
  const double* times;           // [B]
  const float* c;                // [B]
  int B;
  float *loss, *drisks;
};

struct AdamParams {
  float *w, *m, *v;              // flat parameters and Adam moments [n]
  const float* g;                // flat gradients [n]
  const float* l1_mask;          // [n] in {0, 1}: where the L1 term applies (l1_reg_modules), or null = everywhere
  int64_t n;
  float b1, b2, eps, wd, l1;     // l1 = lambda_reg x accumulated micro-batches
  float step_size, bc2_sqrt;     // lr / (1 - b1^t), sqrt(1 - b2^t)
};

int launch_adam_l1(AdamParams p, hipStream_t st);
int launch_abs_sum(const float* w, int64_t n, float* partials, float* out, hipStream_t st);
int launch_head_fwd(HeadParams p, hipStream_t st);
int launch_head_bwd(HeadBwdParams p, hipStream_t st);
int launch_nll(NllParams p, hipStream_t st);
int launch_cox(CoxParams p, hipStream_t st);

}  // namespace mmf
