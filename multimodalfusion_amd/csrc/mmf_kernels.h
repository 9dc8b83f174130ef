// Internal parameter blocks and launcher prototypes shared by the .hip translation units.
// (The public C ABI is include/mmf_amil.h; nothing here is exported.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mmf_common.h"

namespace mmf {

enum : int { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3, ACT_SELU = 4 };

struct LinearParams {      // y[M x N] = drop(act(concat_k(x_s)[M x K] . W[N x K]^T + bias))
  const float* x[4];
  int nseg, kseg, ldx;
  const float* w;
  const float* bias;
  float* y;
  int64_t M;
  int N, K;
  int act;
  float drop_p;
  uint32_t drop_key;
  const uint32_t* seed_dev;   // optional device-resident seed added to every key (graph-replay-safe dropout), or null
  int mt_count, nt_count;
  // optional: one bit per output element, y > 0 (relu'(u) . keep), for K-dh's epilogue -- which then does not have to
  // read h a second time.  Layout: per 16x32 block (rb16 = row / 16, cb = col / 32) 8 words of 64 bits; a word is a
  // wave ballot over the row-major epilogue lanes (lane = 8 rr + c4 holds row rr + 8 t, columns 4 c4 + e of a 32- or
  // 16-row block): rows rr + 8 t with t = 0, 1 go to bit block rb16 as word 4 t + e, t = 2, 3 to rb16 + 1 as word
  // 4 (t - 2) + e.  bits[(rb16 * (N / 32) + cb) * 8 + word].  Needs N % 32 == 0 and tile rows that are multiples of 16.
  unsigned long long* relu_bits;
  int allow_half;          // 1: tile heights that end with a 16-row half block may be chosen (the stack's projection)
  int concurrent;          // mmf_amil_desc::concurrent: plan the wide tiles for 224 of the 256 CUs
  int deep;                // set by launch_linear: short grid, use the deep-prefetch main loop
  int split;               // 1: the split-operand core (mmf_gemm_split.h) where the shape has a tile for it
  // K-split (short grids: a few dozen tiles, each a long serial K loop, on 256 CUs): `ksplit` workgroups share an output tile,
  // each contracts 1 / ksplit of K and writes its partial tile to kpart[(tile * ksplit + s)][BM x BN] with device-scope
  // stores; the LAST to arrive (ktick[tile]: a device-scope counter, zero before the launch and left zero by it) sums the
  // partials in split order -- deterministic -- and runs the epilogue.  No workgroup ever waits for another.
  // launch_linear sets ksplit from linear_ksplit() when kpart / ktick are given; 1 otherwise.
  int ksplit;
  float* kpart;
  unsigned* ktick;
  int ktick_words;
};
// K split launch_linear will use for this shape when it is given partial-tile space and tick words; 1 = none
int linear_ksplit(int64_t M, int N, int K, int nseg, int kseg);
size_t linear_ksplit_floats(int64_t M, int N, int K, int nseg, int kseg);   // size of LinearParams::kpart (0: no split)

struct GateFwdParams {
  const float* h;          // [N x H] (post ReLU/dropout)
  const float *Wa, *ba, *Wb, *bb, *Wc;
  float *a, *b;            // [N x D] un-dropped activations, saved for backward
  float* s_part;           // [nt_count x N]
  int64_t N;
  int H, D, gated;
  float drop_p;            // dropout on a / b (model_modules.py:97-99), 0 = off
  uint32_t key_a, key_b;
  const uint32_t* seed_dev;
  int mt_count, nt_count;
  int64_t row_begin, row_end;   // this launch's rows of the bag (set by launch_gate_fwd)
  // mixed launch (gate_fwd_mixed_kernel): up to 5 row regions in workgroup order, each of tall (128-row) or short
  // (64-row) tiles
  struct Region { int64_t row0; int mt_count, grid_begin, tall; } reg[5];
  int nreg;
  int deep;                     // set by launch_gate_fwd: short grid, use the deep-prefetch main loop
  int split;                    // 1: the split-operand core (mmf_gemm_split.h) where the shape has a tile for it
};

// Optional tail behind K-merge, ONE single-workgroup launch (head_tail_kernel; for small bags it does the merge too):
// the classifier + hazard head of models/model_attention_mil_path.py:58-61 and, when Y is given, nll_surv
// (utils/loss_utils.py:22-39) with its backward down to dM -- instead of the three launches surv_head_fwd, nll_surv,
// surv_head_bwd, each ~5 us of pure launch latency.
struct HeadTail {
  const float *Wk, *bk;        // [K x H], [K]; Wk == null: no tail
  int K;
  float *logits, *hazards, *S; // [K] each
  int64_t* Y_hat;              // [1]
  float* risk;                 // [1] = -sum_k S_k, or null
  // nll_surv + backward (Y == null: head only)
  const int64_t* Y; const float* c;
  float alpha, eps, loss_scale;
  float *loss;                 // [1], unscaled
  float *dM;                   // [H]   d(loss * loss_scale)/dM
  float *dWk, *dbk;            // [K x H], [K]; overwritten, or added to when accumulate != 0
  int accumulate;
};

struct PoolParams {
  const float* s_part;
  int n_parts;
  const float* bc;         // device scalar (attention_c.bias)
  const float* h;
  int64_t N;
  int H;
  float* A_raw;            // [N]
  float* partials;         // [n_groups x (2 + H)]
  float* M;                // [H]
  float* stats;            // {max, denom}
  int n_groups, rows_per_group;
  HeadTail tail;
  int merge_in_tail;       // set by launch_pool_merge: the tail launch merges the (few) partials itself
};

struct BwdPrepParams {     // ds_i = p_i (dM.h_i - dM.M) + gA_i
  const float* h;
  const float* A_raw;
  const float* stats;
  const float* dM;         // [H]
  const float* M;          // [H]
  const float* gA;         // [N] or null
  int64_t N;
  int H;
  float* p;                // [N]
  float* ds;               // [N]
  float* dbc_part;         // [n_groups]
  int n_groups;
};

struct GateBwdCtx {        // what the on-the-fly dP operand needs
  const float *a, *b, *ds, *Wc;
  int D, gated;
  float drop_p;
  uint32_t key_a, key_b;
  const uint32_t* seed_dev;
  // loaders call this once on their private copy: fold the device-resident seed into the keys
  __device__ inline void resolve_seed() {
    if (seed_dev) { const uint32_t sd = *seed_dev; key_a += sd; key_b += sd; seed_dev = nullptr; }
  }
};

// dP[i][k] of one (instance, attention dim): part 0 = d pre-tanh, part 1 = d pre-sigmoid (formulas: mmf_amil_bwd.hip
// header); a_d_b_d returns the dropped a.b product (dWc needs it).  The gate / dropout / part switches are
// compile-time: callers sit in the staging path of a GEMM main loop, where a scalar branch per element costs more
// than the arithmetic it skips (the large-bag kernels are instantiated per (gated, dropout) and pick `part` per chunk).
template <bool GATED, bool DROP, int PART>
__device__ inline float gate_dp_t(const GateBwdCtx& g, float av, float bv, float wc, float dsv,
                                  uint32_t idx, uint32_t thr, float dscale, float& a_d_b_d) {
  float ma = 1.f, mb = 1.f;
  if constexpr (DROP) {
    ma = keep(g.key_a, idx, thr) ? dscale : 0.f;
    if constexpr (GATED) mb = keep(g.key_b, idx, thr) ? dscale : 0.f;
  }
  if constexpr (GATED) {
    a_d_b_d = (av * ma) * (bv * mb);
    return PART == 0 ? dsv * wc * (bv * mb) * ma * (1.f - av * av)
                     : dsv * wc * (av * ma) * mb * bv * (1.f - bv);
  }
  a_d_b_d = av * ma;
  return dsv * wc * ma * (1.f - av * av);
}
// the same with run-time switches (small-tile kernels, where the staging path is not the bottleneck)
__device__ inline float gate_dp(const GateBwdCtx& g, int part, float av, float bv, float wc, float dsv,
                                uint32_t idx, uint32_t thr, float dscale, float& a_d_b_d) {
  float ma = 1.f, mb = 1.f;
  if (g.drop_p > 0.f) {
    ma = keep(g.key_a, idx, thr) ? dscale : 0.f;
    if (g.gated) mb = keep(g.key_b, idx, thr) ? dscale : 0.f;
  }
  if (g.gated) {
    a_d_b_d = (av * ma) * (bv * mb);
    return part == 0 ? dsv * wc * (bv * mb) * ma * (1.f - av * av)
                     : dsv * wc * (av * ma) * mb * bv * (1.f - bv);
  }
  a_d_b_d = av * ma;
  return dsv * wc * ma * (1.f - av * av);
}

struct BwdDhParams {       // du = (dP.Wab + p dM) * relu'(h) * scale_h
  GateBwdCtx g;
  const float *Wa, *Wb;    // [D x H]
  const float* p;          // [N]  (read when !fused_prep)
  const float* dM;         // [H]
  const float* h;          // [N x H]
  const unsigned long long* relu_bits;   // LinearParams::relu_bits of the forward, or null (then h is re-read in the epilogue)
  float* du;               // [N x H]
  int64_t N;
  int H;
  float scale_h;           // 1/(1-p_h) in train mode, 1 in eval
  int mt_count, nt_count;
  int deep;                // set by launch_bwd_dh: short grid, deep-prefetch main loop (dh_mainloop_deep)
  int allow_half;          // 1: half-block tile heights may be chosen (needs fused_prep and relu_bits)
  int concurrent;          // mmf_amil_desc::concurrent
  int split;               // 1: the split-operand core (mmf_gemm_split.h): gated stacks with fused K-prep on wide tiles
  // fused prep (wide tiles own whole rows of h): the kernel computes p_i, ds_i itself (K-prep's job), keeps them
  // in LDS for its loader / epilogue and publishes them for the TN kernel
  int fused_prep;
  const float *A_raw, *stats, *Mpool, *gA;
  float *p_out, *ds_out, *dbc_part;
};

enum : int { TN_A_PLAIN = 0, TN_A_GATE = 1 };

struct TnProblem {         // C[M x Ncols] (+)= A^T . B over a slice of the K (instance) range
  int kind;
  const float* A; int lda; int M;          // TN_A_PLAIN: A[k][m]; TN_A_GATE: M = 2D (gated) or D
  const float* B; int ldb; int Ncols;      // B[k][n]
  float* out; size_t split_stride; int ldc;  // slab s at out + s*split_stride
  float* colsum; size_t colsum_stride;       // per split: column sums of A (bias grads), length M; null = skip
  float* colsum2; size_t colsum2_stride;     // TN_A_GATE only: dWc partials, length D
  int tiles_m, tiles_n, block_begin;       // block_begin: first workgroup of the problem in the plain block order
  int tile_begin;                          // first tile index of the problem (over all problems, set by launch_tn)
  int splits, k_per_split;                 // this problem's K split (0 = TnParams::splits / k_per_split)
};

struct TnParams {
  TnProblem prob[6];
  int nprob;
  int64_t K;               // number of instances (rows of A and B)
  int splits, k_per_split; // k_per_split is a multiple of KC
  int total_tiles;         // tiles over all problems (set by launch_tn)
  int xcd_map;             // 0: plain order (problem, split, tile); 2: block -> (split, tile) through `map`
  int tile;                // 128 or 256 (set by the caller from tn_tile_dim)
  int split;               // 1: split-operand 256 x 256 tiles (mmf_gemm_split.h) when tile == 256
  GateBwdCtx g;
  // xcd_map == 2: map[b] = split << 5 | tile (0xFFFF: no work).  Blocks b, b+8, b+16, ... run on the same XCD, and
  // launch_tn() packs the tiles of one (split, problem) -- which read the same A or B panel -- next to each other
  // there, so a panel is fetched from HBM once per XCD instead of once per tile.
  uint16_t map[512];
};

struct NnParams {          // C[M x N] = A[M x K] . B[K x N]   (plain; radio: dh0 = du.W1)
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;
  int64_t M;
  int N, K;
  int mt_count, nt_count;
};

struct ReduceSeg { const float* in; float* out; int len; int nsplit; size_t stride; int block_begin; int tall; };
struct ReduceParams { ReduceSeg seg[12]; int nseg; int accumulate; };   // accumulate: out += sum instead of out = sum

int launch_linear(LinearParams p, hipStream_t st);
int gate_parts(int D, int gated, int64_t N);
int launch_gate_fwd(GateFwdParams p, hipStream_t st);
int pool_groups(int64_t N);
int launch_pool(PoolParams p, hipStream_t st);
int launch_score_sum(const float* s_part, int n_parts, const float* bc, float* A, int64_t N, hipStream_t st);
int launch_head_tail(PoolParams p, hipStream_t st);    // head_tail_kernel alone on the feature vector p.M [p.H]
int launch_pool_merge(PoolParams p, hipStream_t st);   // single-workgroup merge of p.n_groups partials -> M, stats
int launch_bwd_prep(BwdPrepParams p, hipStream_t st);
int launch_bwd_dh(BwdDhParams p, hipStream_t st);
// split-operand mode on bags below the wide tiles: instances from which the small split tiles (64-row GEMM tiles, 128 x 128
// TN tile) are taken instead of the exact-fp32 ones (default 1: every bag, which keeps an instance's score independent of its bag's size)
int split_min_rows();
int bwd_dh_fused_groups(int64_t N, int H, int allow_half, int D, int gated, int split, int concurrent);   // > 0: launch_bwd_dh computes p/ds itself and writes that many dbc partials
int launch_tn(TnParams p, hipStream_t st);
// wide (32*MB x 256, one 8-wave workgroup per CU) tile selection, shared by the row-parallel GEMMs
int pick_wide_rows(int64_t M, int ntn, bool allow_half, bool concurrent, int max_rows);
bool use_wide_tiles(int64_t M, int N, int split = 0);   // split: the bf16x3 mode's (later) crossover
// split-K plan shared by the workspace carving and the launcher
int tn_tile_dim(int64_t K, int D_gate);                    // 256: one 8-wave 256x256 workgroup per CU; else 128
int tn_splits(int64_t K, int total_tiles, int tile);
int launch_nn(NnParams p, hipStream_t st);
int launch_reduce(ReduceParams p, hipStream_t st);
int set_dyn_lds(const void* kern, int bytes);
void debug_stamps_fwd(unsigned long long* out8);
void debug_stamps_bwd(unsigned long long* out8);

// Optional per-kernel timing with HIP events on the launch stream: records into the mmf_trace of the ABI call in
// progress (mmf_amil_desc::trace, thread-local while the call runs); no cost when the call carries none.
void prof_begin(const char* name, hipStream_t st);
void prof_end(hipStream_t st);
struct ProfScope {
  hipStream_t st;
  ProfScope(const char* name, hipStream_t s) : st(s) { prof_begin(name, s); }
  ~ProfScope() { prof_end(st); }
};

}  // namespace mmf
