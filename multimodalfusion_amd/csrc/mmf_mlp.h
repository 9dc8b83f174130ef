// Parameter blocks of the small dense / fusion kernels (mmf_mlp.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmf {

struct DropSpec {            // kind: 0 none, 1 Dropout, 2 AlphaDropout
  int kind; float p; uint32_t key;
  const uint32_t* dev;       // optional device-resident seed added to the key (graph-replay-safe dropout)
};

struct DenseParams {
  const float *x, *W, *bias;
  float* y;
  int B, K, N, act;
  DropSpec drop;
};
struct DenseBwdParams {
  const float *dy, *y, *x, *W;
  float *dpre, *dx, *dW, *db;
  int B, K, N, act;
  DropSpec drop;
};
struct KronParams {
  const float* o[3];
  float* out;          // forward
  const float* g;      // backward: d out
  float* d[3];         // backward: d o_t
  int m, dim, B;
  DropSpec drop;
};

int launch_dense_fwd(DenseParams p, hipStream_t st);
int launch_dense_bwd(DenseBwdParams p, hipStream_t st);
int launch_gate_mul(const float* z, const float* h, float* o, int n, hipStream_t st);
int launch_gate_mul_bwd(const float* g, const float* z, const float* h, float* dz, float* dh, int n, hipStream_t st);
int launch_kron_fwd(KronParams p, hipStream_t st);
int launch_kron_bwd(KronParams p, hipStream_t st);

}  // namespace mmf

namespace mmf {
// Fused per-modality gating stage of XlinearFusion (models/model_modules.py:158-165), all m modalities in ONE launch:
//   h_i = relu(Wh_i v_i + bh_i) ; z_i = Wz_i v_cat + bz_i ; gm_i = sigmoid(z_i) * h_i ; o_i = drop(relu(Wo_i gm_i + bo_i))
struct XReduceParams {
  int m, B, dim, sdim;                  // modalities (2|3), batch, 256, 16
  const float* v[3];                    // [B x dim]
  const float *Wh[3], *bh[3];           // [sdim x dim]
  const float *Wz[3], *bz[3];           // [sdim x m*dim]
  const float *Wo[3], *bo[3];           // [sdim x sdim]
  float *h[3], *z[3], *gm[3], *o[3];    // [B x sdim] (forward outputs / backward inputs)
  // backward
  const float* d_o[3];                  // [B x sdim]
  float* dv[3];                         // [B x dim]  (overwritten: includes the v_cat contribution)
  float *dWh[3], *dbh[3], *dWz[3], *dbz[3], *dWo[3], *dbo[3];
  DropSpec drop;                        // key of site 0; site i uses key + i*0x632BE5AB (see drop_key)
};
int launch_xreduce_fwd(XReduceParams p, hipStream_t st);
int launch_xreduce_bwd(XReduceParams p, hipStream_t st);
}  // namespace mmf

// ---- stage-2 (embedding-level) building blocks: SURVEY.md 8f row N3 --------------------------------------------
namespace mmf {
struct BnParams {            // y = drop(act(BatchNorm1d(x) [+ res]))      x, y: [B x F]
  const float *x, *res, *gamma, *beta;
  float *running_mean, *running_var;     // updated in training mode (momentum, unbiased variance), read in eval
  float *y, *save_mean, *save_invstd;    // save_*: [F], what backward needs
  int B, F, training, act;
  float eps, momentum;
  DropSpec drop;
};
struct BnBwdParams {
  const float *dy, *y, *x, *gamma, *save_mean, *save_invstd;
  float *dx, *dres, *dgamma, *dbeta;     // dres may be null
  int B, F, training, act;
  DropSpec drop;
};
struct HighwayParams {       // y = sigmoid(zg) * relu(zn) + (1 - sigmoid(zg)) * zl        all [n]
  const float *zg, *zn, *zl;
  float* y;
  const float* dy;                       // backward
  float *dzg, *dzn, *dzl;
  int64_t n;
};
struct RankParams {          // loss = -mean|sum over comparable pairs of phi(risk_more - risk_less)
  const float* risks; const double* times; const float* c;
  int B, phi, reduction;                 // phi: 0 sigmoid, 1 relu; reduction: 0 mean, 1 sum
  float *loss, *d_risks;
};
struct HazardParams {        // logits [B x K] -> hazards, S, Y_hat, risk = -sum_k S
  const float* logits;
  float *hazards, *S, *risk; int64_t* Y_hat;
  const float *g_hazards, *g_S, *g_risk;  // backward (any may be null)
  float* dlogits;
  int B, K;
};
int launch_bn_fwd(BnParams p, hipStream_t st);
int launch_bn_bwd(BnBwdParams p, hipStream_t st);
int launch_highway_fwd(HighwayParams p, hipStream_t st);
int launch_highway_bwd(HighwayParams p, hipStream_t st);
int launch_rank_loss(RankParams p, hipStream_t st);
int launch_hazard_fwd(HazardParams p, hipStream_t st);
int launch_hazard_bwd(HazardParams p, hipStream_t st);
// ---- omic head, one training step in one launch (mmf_maxnet.hip) ------------------------------------------------------
struct MaxnetStepParams {
  int B, G;
  const float *x, *W0, *b0, *W1, *b1, *Wc, *bc;
  const double* times;
  const float* c;
  float p;                           // AlphaDropout probability of both blocks (0: eval)
  uint32_t key0, key1;
  const uint32_t* seed_dev;
  float loss_scale;
  float *y0, *y1, *dp1, *dp0, *dr;   // workspace: y0, y1 [B][256]; dp1, dp0 TRANSPOSED [256][maxnet_step_dp_pitch(B)]; dr [B]
  float* dwc_part;                   // workspace: [32 workgroups][256] shares of dWc
  unsigned long long* stamps;        // -DMMF_STAMPS builds: 8 words of wall-clock stamps written by workgroup 0, else null
  unsigned* bar;                     // 3 tick words
  float *risk, *loss;
  float *dW0, *db0, *dW1, *db1, *dWc, *dbc;
  int accumulate;
};
size_t maxnet_step_workspace_floats(int B);
int maxnet_step_dp_pitch(int B);
bool maxnet_step_ok(int B, int G, int H0, int H1);
int launch_maxnet_cox_step(MaxnetStepParams p, hipStream_t st);

}  // namespace mmf
