// bf16-storage variant of the attention-MIL path (BASELINE config 5: 100k x 1024 bags, HBM-bound).
//
// What is bf16: the bag x, the per-call copies of W1 / Wa / Wb, and every saved or exchanged activation
// (h, a, b, du, and the on-the-fly dP operand).  What stays fp32: parameters (masters), biases, Wc, the MFMA
// accumulators, every epilogue (bias, ReLU, tanh, sigmoid, dropout scale, scores, softmax, pooling), the
// gradients handed back, and the split-K slabs.  Rounding is round-to-nearest-even at exactly these points:
//   x (given), bf16(W1), bf16(Wa), bf16(Wb);  h = bf16(drop(relu(u)));  a = bf16(tanh), b = bf16(sigmoid) (saved, and
//   what the scores are computed from);  dP = bf16(gate_dp(a, b, ...));  du = bf16(...).
// The oracle restates the same points (oracle/torch_port.py: bf16 mode).
#pragma once
#include "mmf_kernels.h"

namespace mmf {

typedef unsigned short bf16_t;   // storage type (raw bits)

__device__ inline float bf2f(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ inline bf16_t f2bf(float f) {            // round to nearest even (v_cvt_pk_bf16_f32)
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
typedef float f32pair __attribute__((ext_vector_type(2)));
typedef __bf16 bf16pair __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pack2(float lo, float hi) {   // ONE v_cvt_pk_bf16_f32 (the scalar form costs two and an OR)
  const f32pair v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16pair));
}
__device__ inline void unpack2(uint32_t w, float& lo, float& hi) {
  lo = __uint_as_float(w << 16);
  hi = __uint_as_float(w & 0xFFFF0000u);
}
__device__ inline void unpack8(const float4& v, float (&o)[8]) {
  unpack2(__float_as_uint(v.x), o[0], o[1]);
  unpack2(__float_as_uint(v.y), o[2], o[3]);
  unpack2(__float_as_uint(v.z), o[4], o[5]);
  unpack2(__float_as_uint(v.w), o[6], o[7]);
}
__device__ inline float4 pack8(const float (&o)[8]) {
  return make_float4(__uint_as_float(pack2(o[0], o[1])), __uint_as_float(pack2(o[2], o[3])),
                     __uint_as_float(pack2(o[4], o[5])), __uint_as_float(pack2(o[6], o[7])));
}
__device__ inline uint2 pack4(float a, float b, float c, float d) { return make_uint2(pack2(a, b), pack2(c, d)); }

struct CvtSeg {            // dst[r][c0 + c] (ld = dst_ld) = bf16(src[r][c]);  transpose 1: dst[c][c0 + r];
                           // transpose 2: dst[c][c0 + 64 (r / 32) + r % 32]  (K-dh's interleaved [Wa-block | Wb-block] k order)
                           // transpose 3: W1 [256 x L] in the fused forward's A-fragment order: 16-byte unit (8 k) of
                           //   lane 32 hh + i, k-step q, row block fb, wave w, chunk kt at ((((kt 4 + w) 4 + q) 2 + fb) 64 + lane)
                           // transpose 4: Wa (c0 = 0) / Wb (c0 = 16) [256 x 256], unit of lane 32 hh + c0 + (d & 15), k-step s,
                           //   wave w = (d >> 4) & 3, pass ps = d >> 6 at (((ps 4 + w) 16 + s) 64 + lane)
                           // transpose 5: Wa (c0 = 0) / Wb (c0 = 32) [256 x 256] as K-dh's A operand [feature][k'] in fragment order:
                           //   k' = 64 (d / 32) + c0 + d % 32 (K-dh's chunk order), then as transpose 3 with L = 512
  const float* src; bf16_t* dst;
  int rows, cols, dst_ld, c0, transpose, block_begin;
};
struct CvtParams { CvtSeg seg[6]; int nseg; };

struct LinearBfParams {    // y[M x N] = bf16(drop(relu(x[M x K] . W[N x K]^T + bias)))
  const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y;
  int64_t M; int N, K;
  float drop_p; uint32_t drop_key; const uint32_t* seed_dev;
  int mt_count, nt_count;
};

struct GateBfParams {
  const bf16_t* h;                       // [N x H]
  const bf16_t *Wa, *Wb;                 // bf16 copies, [D x H]
  const float *ba, *bb, *Wc;             // fp32
  bf16_t *a, *b;                         // [N x D] saved (un-dropped)
  float* s_part;                         // [nt_count x N]
  int64_t N; int H, D, gated;
  float drop_p; uint32_t key_a, key_b; const uint32_t* seed_dev;
  int mt_count, nt_count;
};

struct PoolBfParams { PoolParams base; const bf16_t* h; };

struct GateBwdBf {         // what the on-the-fly dP operand needs (bf16 a, b)
  const bf16_t *a, *b; const float *ds, *Wc;
  int D, gated; float drop_p; uint32_t key_a, key_b; const uint32_t* seed_dev;
};

struct DhBfParams {        // du = bf16((dP . Wab + p dM) . relu'(h) . scale_h), K-prep fused
  GateBwdBf g;
  const bf16_t* WabT;      // [H x mstk] bf16, row n = column n of Wa / Wb in K-dh's k order (gated: 32-dim blocks a, b, a, b ...)
  bf16_t* dP;              // [N x mstk] out: [d pre-tanh | d pre-sigmoid], the TN kernel's operand
  float* dwc_part;         // [mt_count x D] out: per-row-tile partial of dWc
  const float* dM; const bf16_t* h; bf16_t* du;
  int64_t N; int H; float scale_h;
  const float *A_raw, *stats, *Mpool, *gA;
  float *p_out, *ds_out, *dbc_part;      // dbc_part[mt_count]
  int mt_count, nt_count;
};

struct TnBfProblem {       // C[M x Ncols] = A^T . B over one K split; A, B are bf16 [K x *]
  const bf16_t* A; int lda; int M;
  const bf16_t* B; int ldb; int Ncols;
  float* out; size_t split_stride; int ldc;
  float* colsum; size_t colsum_stride;   // per split: column sums of A (bias grads), length M; null = skip
  int tiles_m, tiles_n, block_begin;
};
struct TnBfParams {
  TnBfProblem prob[2]; int nprob;
  int64_t K; int splits, k_per_split, total_tiles, xcd_map;
};

constexpr int TNB_KCH = 64;   // instances per staged chunk of the bf16 TN kernel
constexpr int TNB_TILE = 256;

int launch_cvt_bf16(CvtParams p, hipStream_t st);
int launch_linear_bf16(LinearBfParams p, hipStream_t st);
int gate_parts_bf16(int D, int gated);
int launch_gate_bf16(GateBfParams p, hipStream_t st);
int launch_pool_bf16(PoolBfParams p, hipStream_t st);
struct FusedFwdParams {     // fused forward (H = 256): instance projection + gate scoring + pooling partials
  const bf16_t* x; const bf16_t* w1; const float* b1;
  const bf16_t *Wa, *Wb; const float *ba, *bb, *Wc, *bc;
  const bf16_t *w1f, *wabf;     // second form only: W1 and [Wa ; Wb] in MFMA-fragment order (CvtSeg::transpose 3 / 4)
  bf16_t *h, *a, *b;            // saved for backward; all three null in forward-only calls
  float* A_raw; float* partials;   // partials: [fused_fwd_tiles(N)][2 + 256] = (max, sum e, sum e.h) per 128-row tile
  int64_t N; int L, D;
  float p_h, p_att; uint32_t key_h, key_a, key_b; const uint32_t* seed_dev;
  int mt_count;
  int hash_in_loop;             // second form: projection dropout bits hashed inside the main loop (L == 1024)
  int stagger;                  // diagnostic builds only (-DMMF_F2_DEBUG): mask of phases to leave out
};
int fused_fwd_tiles(int64_t N);
bool fused_fwd_ok(int64_t N, int L, int H, int D);
int launch_fused_fwd_bf16(FusedFwdParams p, int gated, hipStream_t st);
bool fused_fwd2_ok(int64_t N, int L, int H, int D);   // second form (mmf_amil_bf16_fwd2.hip): two 4-wave workgroups per CU
int launch_fused_fwd2_bf16(FusedFwdParams p, int gated, hipStream_t st);
int dh_bf16_row_tiles(int64_t N);              // capacity of dbc_part (upper bound over tile choices)
int dh_bf16_tiles_used(int64_t N, int ntn);     // dbc partials launch_dh_bf16 writes for this shape
int launch_dh_bf16(DhBfParams p, hipStream_t st);
bool dh2_bf16_ok(int64_t N, int H, int D, int gated);   // second form (mmf_amil_bf16_dh2.hip): WabT in A-fragment order (CvtSeg::transpose 5)
int launch_dh2_bf16(DhBfParams p, hipStream_t st);
int tn_bf16_splits(int64_t K, int total_tiles);
int launch_tn_bf16(TnBfParams p, hipStream_t st);
void debug_stamps_bf16(unsigned long long* out32);

}  // namespace mmf
