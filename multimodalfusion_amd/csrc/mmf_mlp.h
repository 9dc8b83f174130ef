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
