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
