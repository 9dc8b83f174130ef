/* mmf_amil.h -- C ABI of libmmf_amil.so: the MI355X (gfx950) attention-MIL + survival-loss hot path.
 *
 * The reference (MultimodalFusion/multimodalfusion) has no FFI / plugin API: its boundary is
 * the Python nn.Module surface.  This header is the C ABI that sits directly beneath that
 * surface -- the entry points a binding for this path binds (ctypes in this repo, see
 * INTEGRATION.md).  Each entry point names the reference lines it replaces
 * (paths relative to the reference repo root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (fp32, row-major, 16-byte aligned)
 *     unless the comment says "host";  the library allocates nothing and keeps no state (no globals: the optional
 *     device-resident dropout seed and the optional kernel trace are passed per call);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream), performs no host synchronisation, and is re-entrant / thread-safe (autograd
 *     calls backward from another thread);
 *   - return value: 0 on success, negative mmf error code otherwise (mmf_strerror()).
 *   - "workspace": a caller-owned scratch buffer; forward fills it with the saved activations
 *     (h, a, b, scores, softmax statistics) that backward reads, so the SAME buffer must be
 *     passed to the matching backward call, unmodified in between.
 */
#ifndef MMF_AMIL_H
#define MMF_AMIL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMF_ACT_NONE 0
#define MMF_ACT_RELU 1
#define MMF_ACT_TANH 2
#define MMF_ACT_SIGMOID 3
#define MMF_ACT_SELU 4

const char* mmf_strerror(int code);
/* ABI version; bumped on any signature change. */
int mmf_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Attention-MIL stack:  Sequential(Linear(L,H), ReLU, Dropout(0.25), Attn_Net[_Gated](H,D,1))
 * followed by softmax pooling over the instances.
 *   replaces models/model_attention_mil_path.py:20-29 (construction), :52-56 (forward),
 *            models/model_modules.py:70-85 (Attn_Net), :87-110 (Attn_Net_Gated),
 *            and the same stack in model_attention_mil_radio.py:88-99 / model_mm_attention_mil.py:146-160.
 * ------------------------------------------------------------------------------------------- */
struct mmf_trace;

typedef struct mmf_amil_desc {
  int64_t N;          /* instances in the bag */
  int32_t L, H, D;    /* feature dim, hidden dim, attention dim: small 1024/256/256, big 1024/512/384 */
  int32_t gated;      /* 1 = Attn_Net_Gated, 0 = Attn_Net */
  const float* W1;    /* [H x L]  attention_net.0.weight */
  const float* b1;    /* [H] */
  const float* Wa;    /* [D x H]  attention_a.0.weight (gated) / module.0.weight (ungated) */
  const float* ba;    /* [D] */
  const float* Wb;    /* [D x H]  attention_b.0.weight (gated only, else NULL) */
  const float* bb;    /* [D] */
  const float* Wc;    /* [1 x D]  attention_c.weight / module.{2|3}.weight */
  const float* bc;    /* [1] */
  float p_h;          /* dropout prob after the ReLU (0.25 in train mode, 0 in eval) */
  float p_att;        /* dropout prob on the tanh / sigmoid branches (0.25 iff dropout=True and training) */
  uint32_t seed;      /* dropout seed of this call; masks are regenerated, never stored */
  const uint32_t* seed_dev;  /* optional DEVICE word added to `seed` by every kernel of the call (uint32 wrap), or NULL.
                              * By-value seeds are frozen into a captured hipGraph; a graph whose first node bumps this
                              * word draws fresh masks on every replay.  Forward and backward must see the same value. */
  struct mmf_trace* trace;   /* optional kernel trace (mmf_trace_create), or NULL: see "Kernel trace" below */
  int32_t concurrent;        /* scheduling hint; results agree to fp32 rounding whatever it is (bit for bit unless the two
                              * tile plans put a row into a 16-row half block, v_mfma_f32_16x16x4_f32, in one and into a
                              * 32-row block in the other: same k order, rounded per instruction).  0: the call has the GPU to itself -- the
                              * row-parallel GEMMs take the tile height that finishes ONE bag soonest (208 rows: a 50k
                              * bag on 241 of 256 CUs, one bag per step 0.816 -> 0.801 ms).  1: other bags' kernels run
                              * beside it on other streams (pipeline.BagsInFlight) -- 224-row tiles, which leave 32 CUs
                              * to the neighbours, gave the higher aggregate rate in two of three same-run comparisons
                              * (1376 vs 1337 bags/s with three bags in flight) and the same rate in the third. */
  int32_t gemm;              /* how the stack's four large fp32 contractions are multiplied (fp32 calls only; inputs,
                              * outputs, saved activations and accumulation are fp32 either way):
                              *   MMF_GEMM_F32    (0) v_mfma_f32_32x32x2_f32, the exact-fp32 matrix instruction;
                              *   MMF_GEMM_BF16X3 (1) every fp32 operand as the exact sum of three bf16 values, the six
                              *                       leading products on v_mfma_f32_32x32x16_bf16, fp32 accumulation
                              *                       (csrc/mmf_gemm_split.h).  Same error against an fp64 product as
                              *                       mode 0 (tests/test_gpu_split.py), 2.67 x its instruction rate.
                              *                       Every bag size, gated or not (which tile a bag size takes is a
                              *                       tuning detail of the launcher, not a contract); the radio head's
                              *                       segmented projection and the stand-alone scorer run mode 0.
                              *                       Finite operands stay finite (values beyond the largest bf16 are
                              *                       split around a clamped first plane); infinite operands give NaN,
                              *                       as in mode 0. */
  uint32_t* sync;            /* optional DEVICE array of `sync_words` 32-bit words, or NULL.  Tick words of the launches in
                              * which several workgroups share an output tile (K-split projections of short grids: each
                              * writes a partial tile, the last to arrive -- found through one of these words -- sums them
                              * in a fixed order and finishes the tile; no workgroup waits for another).  Contract: zero
                              * before the first call that sees it; every call leaves it zero; calls that may run at the
                              * same time (different streams) need different arrays.  NULL / too few words: such launches
                              * fall back to one workgroup per tile (same results to fp32 rounding, slower small bags). */
  int32_t sync_words;        /* 1024 covers every shape */
} mmf_amil_desc;
#define MMF_GEMM_F32 0
#define MMF_GEMM_BF16X3 1

typedef struct mmf_amil_grads {
  float* dW1; float* db1;
  float* dWa; float* dba;
  float* dWb; float* dbb;   /* NULL when ungated */
  float* dWc; float* dbc;
  float* dx;                /* [N x L] or NULL (path bags need no input gradient; radio does) */
} mmf_amil_grads;

size_t mmf_amil_workspace_bytes(int64_t N, int32_t L, int32_t H, int32_t D, int32_t gated);

/* x [N x L] -> M [H] (pooled embedding), A_raw [N] (pre-softmax scores, what heat-maps consume). */
int mmf_amil_forward(const mmf_amil_desc* desc, const float* x, void* workspace, size_t workspace_bytes,
                     float* M, float* A_raw, void* stream);

/* dM [H], gA [N] (gradient w.r.t. A_raw, may be NULL) -> parameter grads (overwritten, not accumulated). */
int mmf_amil_backward(const mmf_amil_desc* desc, const float* x, void* workspace, size_t workspace_bytes,
                      const float* M, const float* A_raw, const float* dM, const float* gA,
                      const mmf_amil_grads* grads, void* stream);

/* ---------------------------------------------------------------------------------------------
 * bf16-storage variant of the same stack (BASELINE config 5: 100k x 1024 bags, HBM-bound).
 *   x is a bf16 bag [N x L] (raw bits, row-major).  Parameters and returned gradients stay fp32;
 *   per call the library makes bf16 copies of W1 / Wa / Wb in the workspace, keeps h, a, b, du in
 *   bf16, multiplies on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, and runs every epilogue
 *   (bias, ReLU, tanh, sigmoid, dropout, scores, softmax pooling) in fp32.  M and A_raw are fp32.
 *   Same desc, same dropout masks as the fp32 entry points.  Needs L % 64 == 0, H % 256 == 0.
 *   grads->dx must be NULL (the bag is a leaf).
 * ------------------------------------------------------------------------------------------- */
size_t mmf_amil_bf16_workspace_bytes(int64_t N, int32_t L, int32_t H, int32_t D, int32_t gated);
int mmf_amil_bf16_forward(const mmf_amil_desc* desc, const uint16_t* x, void* workspace, size_t workspace_bytes,
                          float* M, float* A_raw, void* stream);
int mmf_amil_bf16_backward(const mmf_amil_desc* desc, const uint16_t* x, void* workspace, size_t workspace_bytes,
                           const float* M, const float* A_raw, const float* dM, const float* gA,
                           const mmf_amil_grads* grads, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention stack + classifier / hazard head in one call, and the whole training step of one bag in one call.
 *   The path head's forward is stack -> classifier -> sigmoid / cumprod / argmax (models/model_attention_mil_path.py:
 *   52-61); the training loop then applies nll_surv and calls backward (utils/core_utils.py:200-243).  After the
 *   pooling kernel these are single-workgroup launches of a few microseconds each; here they run as the tail of the
 *   pooling merge (one single-workgroup launch behind it; for small bags the merge runs there too), so a bag costs 7-8 launches instead of 13 -- what matters for 1k-10k bags.
 *   x_bf16 != 0: x is a bf16 bag (uint16_t bits) and the bf16-storage kernels run (see above).
 *   workspace: mmf_amil_workspace_bytes / mmf_amil_bf16_workspace_bytes of the same shape.
 * mmf_amil_head_forward: stack + head; M [H], A_raw [N] and the head outputs are written; backward as usual
 *   (mmf_surv_head_backward, then mmf_amil[_bf16]_backward with the same workspace).
 * mmf_amil_nll_step: forward + head + nll_surv + backward.  Writes the head outputs, loss (unscaled) and A_raw, and the
 *   gradients of loss * loss_scale w.r.t. every parameter: grads (attention stack) and target->dWk / dbk
 *   (classifier), overwritten, or ADDED to what the buffers hold when target->accumulate != 0 (gradient accumulation
 *   over the `gc` bags of a window, utils/core_utils.py:242-247, with loss_scale = 1 / gc).  grads->dx must be NULL.
 * ------------------------------------------------------------------------------------------- */
typedef struct mmf_surv_head {
  const float* Wk;      /* [K x H] classifier.weight */
  const float* bk;      /* [K] */
  int32_t K;            /* <= 32 */
  float* logits;        /* [K] out */
  float* hazards;       /* [K] out */
  float* S;             /* [K] out */
  int64_t* Y_hat;       /* [1] out */
  float* risk;          /* [1] out = -sum_k S_k (what the loop logs, utils/core_utils.py:207), or NULL */
} mmf_surv_head;

typedef struct mmf_nll_target {
  const int64_t* Y;     /* [1] device: discrete time bin */
  const float* c;       /* [1] device: censorship */
  float alpha, eps;     /* NLLSurvLoss(alpha), eps = 1e-7 */
  float loss_scale;     /* gradients are those of loss * loss_scale */
  float* loss;          /* [1] out, unscaled */
  float* dWk;           /* [K x H] */
  float* dbk;           /* [K] */
  int32_t accumulate;   /* 0: every gradient buffer is overwritten; 1: added to */
} mmf_nll_target;

int mmf_amil_head_forward(const mmf_amil_desc* desc, const void* x, int32_t x_bf16, void* workspace, size_t workspace_bytes,
                          const mmf_surv_head* head, float* M, float* A_raw, void* stream);
int mmf_amil_nll_step(const mmf_amil_desc* desc, const void* x, int32_t x_bf16, void* workspace, size_t workspace_bytes,
                      const mmf_surv_head* head, const mmf_nll_target* target, float* A_raw,
                      const mmf_amil_grads* grads, void* stream);

/* The hazard head's training step on a feature vector that is already on the device: what
 *   `hazards, S, Y_hat = head(classifier(feat)); loss = NLLSurvLoss(alpha)(hazards, S, Y, c); (loss * loss_scale).backward()`
 * computes between the embedding and the loss (models/model_mm_attention_mil.py:190-191 with fusion = 'concat': feat is the
 * concatenation of the branch embeddings, which the caller lets the branches write side by side; utils/loss_utils.py:22-39)
 * in ONE single-workgroup launch: classifier, sigmoid, cumprod, argmax, the loss, dWk / dbk (target->accumulate as above) and
 * dfeat [F] = d(loss * loss_scale) / d feat, which the caller hands to the branches' backward calls.  F <= 1024, K <= 32. */
int mmf_surv_head_nll_step(const float* feat, int32_t F, const mmf_surv_head* head, const mmf_nll_target* target,
                           float* dfeat, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Forward-only variants for the inference consumers of the path -- embedding export
 * (pre_trained_feature.py:116-162: model(..., return_features=True) under no_grad), per-patient inference and
 * attention heat-map scoring (utils/heatmap_utils.py:111-150,249-275: A_raw per bag / per 512-patch batch).
 * Same results as mmf_amil[_bf16]_forward, but nothing is saved for a backward: the a / b activations are never
 * written and the workspace is the small one returned here.  desc->p_h / p_att should be 0 (eval mode).
 * ------------------------------------------------------------------------------------------- */
size_t mmf_amil_infer_workspace_bytes(int64_t N, int32_t L, int32_t H, int32_t D, int32_t gated);
int mmf_amil_infer(const mmf_amil_desc* desc, const float* x, void* workspace, size_t workspace_bytes,
                   float* M, float* A_raw, void* stream);
size_t mmf_amil_bf16_infer_workspace_bytes(int64_t N, int32_t L, int32_t H, int32_t D, int32_t gated);
int mmf_amil_bf16_infer(const mmf_amil_desc* desc, const uint16_t* x, void* workspace, size_t workspace_bytes,
                        float* M, float* A_raw, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The attention scorer on its own: Attn_Net(L, D).forward(x) / Attn_Net_Gated(L, D).forward(x) -> (A, x)
 *   (models/model_modules.py:84-85 and :105-110; n_classes = 1):  A[i] = (tanh(x_i Wa^T + ba) [. sigmoid(x_i Wb^T + bb)]) Wc^T + bc,
 *   with Dropout(0.25) on the branches when desc->p_att > 0.  Same kernels as inside the stack (K-gate, K-dh, K-tn).
 *   desc: N, H (= the scorer's input width L), D, gated, Wa..bc, p_att, seed[, seed_dev, trace]; L, W1, b1, p_h are ignored.
 *   H % 32 == 0, D % 32 == 0.  forward keeps a, b in the workspace for the matching backward.
 *   backward: gA [N] = dL/dA -> grads->dWa, dba, (dWb, dbb,) dWc, dbc and, when grads->dx != NULL, dx [N x H].
 * ------------------------------------------------------------------------------------------- */
size_t mmf_attn_net_workspace_bytes(int64_t N, int32_t H, int32_t D, int32_t gated);
int mmf_attn_net_forward(const mmf_amil_desc* desc, const float* x, void* workspace, size_t workspace_bytes, float* A,
                         void* stream);
int mmf_attn_net_backward(const mmf_amil_desc* desc, const float* x, void* workspace, size_t workspace_bytes,
                          const float* gA, const mmf_amil_grads* grads, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense layer on MFMA:  y = dropout(act(concat_k(x_0..x_{nseg-1}) . W^T + bias))
 *   replaces torch.cat + nn.Linear of model_attention_mil_radio.py:80-82 (reduce_dim; the modality
 *   bags are never concatenated in memory) and the instance projections generally.
 *   x_segs: HOST array of nseg (<= 4) device pointers, each [M x kseg]; K = nseg*kseg; K % 32 == 0.
 *   workspace / sync (both optional, may be NULL): scratch for the K-split plan of short grids (a 512-row radiology bag
 *   against the 4096-wide reduce_dim is 128 output tiles with a 128-chunk K loop each: split four ways -- one modality
 *   segment per workgroup -- it fills the chip) and the tick words it needs, under mmf_amil_desc::sync's contract.
 * ------------------------------------------------------------------------------------------- */
size_t mmf_linear_forward_workspace_bytes(int64_t M, int32_t N, int32_t nseg, int32_t kseg);   /* 0: the shape is not split */
int mmf_linear_forward(const float* const* x_segs, int32_t nseg, int32_t kseg, int64_t M,
                       const float* W, const float* bias, int32_t N, int32_t act,
                       float drop_p, uint32_t drop_seed, uint32_t drop_site, const uint32_t* seed_dev,
                       float* y, void* workspace, size_t workspace_bytes, uint32_t* sync, int32_t sync_words, void* stream);

size_t mmf_linear_backward_workspace_bytes(int64_t M, int32_t N, int32_t K);
/* dy [M x N] (gradient w.r.t. the pre-activation output) -> dW [N x K], db [N] (may be NULL),
 * dx [M x K] single buffer (may be NULL; only nseg == 1). */
int mmf_linear_backward(const float* dy, const float* const* x_segs, int32_t nseg, int32_t kseg, int64_t M,
                        const float* W, int32_t N, float* dW, float* db, float* dx,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Survival head:  logits = f.Wk^T + bk; hazards = sigmoid(logits); S = cumprod(1-hazards); Y_hat = argmax
 *   replaces models/model_attention_mil_path.py:58-61.
 * ------------------------------------------------------------------------------------------- */
int mmf_surv_head_forward(const float* feat, const float* Wk, const float* bk, int32_t B, int32_t F, int32_t K,
                          float* logits, float* hazards, float* S, int64_t* Y_hat, void* stream);
int mmf_surv_head_backward(const float* g_hazards, const float* g_S, const float* hazards, const float* feat,
                           const float* Wk, int32_t B, int32_t F, int32_t K,
                           float* dfeat, float* dWk, float* dbk, void* stream);

/* nll_surv loss (utils/loss_utils.py:22-39): loss [1], and its gradients g_hazards, g_S [B x K].
 * A label outside [0, K) (the reference's gather raises an index error) poisons the result instead of the memory:
 * loss = NaN, that sample's gradients = 0; nothing is read or written out of bounds. */
int mmf_nll_surv(const float* hazards, const float* S, const int64_t* Y, const float* c, int32_t B, int32_t K,
                 float alpha, float eps, float* loss, float* g_hazards, float* g_S, void* stream);

/* Cox partial-likelihood loss (utils/loss_utils.py:124-139): loss [1], d_risks [B].  times is float64. */
int mmf_cox_surv(const float* risks, const double* times, const float* c, int32_t B,
                 float* loss, float* d_risks, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Small dense layers (any dimensions, batch 1..128) and the Kronecker fusion block.
 *   replaces SNN_Block (models/model_modules.py:64-68: Linear+SELU+AlphaDropout), the Linear+ReLU+Dropout
 *   stacks and gating of XlinearFusion (models/model_modules.py:133-178), and the fusion classifiers
 *   (models/model_mm_attention_mil.py:91,95).
 *   drop_kind: 0 none, 1 nn.Dropout, 2 nn.AlphaDropout; the mask is the keep-hash of (seed, site, element).
 *   seed_dev (everywhere below): optional device word added to `seed`, or NULL -- see mmf_amil_desc.
 * ------------------------------------------------------------------------------------------- */
int mmf_dense_forward(const float* x, const float* W, const float* bias, int32_t B, int32_t K, int32_t N,
                      int32_t act, int32_t drop_kind, float drop_p, uint32_t seed, uint32_t site,
                      const uint32_t* seed_dev, float* y, void* stream);
/* dy, y (the forward OUTPUT) -> dx [B x K] (may be NULL), dW [N x K], db [N] (may be NULL);
 * dpre_scratch: [B x N] floats. */
int mmf_dense_backward(const float* dy, const float* y, const float* x, const float* W,
                       int32_t B, int32_t K, int32_t N, int32_t act,
                       int32_t drop_kind, float drop_p, uint32_t seed, uint32_t site, const uint32_t* seed_dev,
                       float* dpre_scratch, float* dx, float* dW, float* db, void* stream);
/* o = sigmoid(z) * h (n elements) and its backward. */
int mmf_gate_mul_forward(const float* z, const float* h, float* o, int32_t n, void* stream);
int mmf_gate_mul_backward(const float* g, const float* z, const float* h, float* dz, float* dh, int32_t n, void* stream);
/* out[b] = [o0,1] (x) [o1,1] ((x) [o2,1]) followed by Dropout(drop_p); o_t: [B x dim]; m = 2 or 3 (HOST array of ptrs). */
int mmf_kron_forward(const float* const* o, int32_t m, int32_t dim, int32_t B,
                     float drop_p, uint32_t seed, uint32_t site, const uint32_t* seed_dev, float* out, void* stream);
int mmf_kron_backward(const float* g, const float* const* o, int32_t m, int32_t dim, int32_t B,
                      float drop_p, uint32_t seed, uint32_t site, const uint32_t* seed_dev, float* const* d_o,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-step tail on flat fp32 buffers: the gradient of l1_reg_all + torch.optim.Adam(weight_decay) in one launch.
 *   replaces utils/utils.py:249-257 (l1_reg_all, through autograd) + utils/utils.py:144-146 (Adam) as used by
 *   utils/core_utils.py:216-219,242-247.  l1_coeff = lambda_reg x (micro-batches accumulated since the last step);
 *   step = 1-based optimizer step count.  w, m, v are updated in place.
 *   l1_mask: NULL = the L1 term covers every element (l1_reg_all); else [n] floats in {0, 1} selecting the elements it
 *   covers (l1_reg_modules, utils/utils.py:259-268: fc_omic and mm only).
 * mmf_abs_sum: out[0] = sum_i |w_i| (the value of l1_reg_all); partials = 512 floats of scratch.
 * ------------------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------------------
 * Omic head, ONE training step in ONE launch:  MaxNet forward (two SNN blocks + classifier -> risk), CoxSurvLoss, and
 * every parameter gradient.
 *   replaces models/model_genomic.py:53-72 (MaxNet.forward, bag_loss = cox_surv), models/model_modules.py:64-68 (SNN_Block:
 *   Linear + SELU + AlphaDropout), utils/loss_utils.py:124-139 (CoxSurvLoss) and the backward autograd derives from them --
 *   ~20 framework launches in the reference, 9 through the composable entry points above (mmf_dense_*, mmf_cox_surv).
 *   `small` net only (H0 = H1 = 256), B <= 256, G <= 256; other shapes: MMF_ERR_SHAPE (use the composable entry points).
 *   times: DEVICE float64 [B] (the reference compares event times in float64); loss: unscaled; gradients are those of
 *   loss * loss_scale, written or (accumulate) added.  Needs 3 tick words (mmf_amil_desc::sync's contract).
 * ------------------------------------------------------------------------------------------- */
typedef struct mmf_maxnet_desc {
  int32_t B, G, H0, H1;      /* batch, input genes, hidden widths */
  const float* x;            /* [B x G] */
  const float* W0;           /* [H0 x G]  fc_omic.0.0.weight */
  const float* b0;           /* [H0] */
  const float* W1;           /* [H1 x H0] fc_omic.1.0.weight */
  const float* b1;           /* [H1] */
  const float* Wc;           /* [1 x H1]  classifier.weight */
  const float* bc;           /* [1] */
  float p_drop;              /* AlphaDropout probability of both blocks (0.25 in train mode, 0 in eval) */
  uint32_t seed;             /* dropout seed: block i draws with site i, element index b * 256 + n (as mmf_dense_forward) */
  const uint32_t* seed_dev;  /* optional device word added to the seed (graph replays), or NULL */
  uint32_t* sync;            /* tick words, >= 3 */
  int32_t sync_words;
  struct mmf_trace* trace;
} mmf_maxnet_desc;
typedef struct mmf_maxnet_grads { float *dW0, *db0, *dW1, *db1, *dWc, *dbc; } mmf_maxnet_grads;
size_t mmf_maxnet_cox_step_workspace_bytes(int32_t B);
int mmf_maxnet_cox_step(const mmf_maxnet_desc* desc, const double* times, const float* c, float loss_scale,
                        void* workspace, size_t workspace_bytes, float* risk, float* loss,
                        const mmf_maxnet_grads* grads, int32_t accumulate, void* stream);

int mmf_adam_l1_step(float* w, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, float l1_coeff, const float* l1_mask, int32_t step, void* stream);
int mmf_abs_sum(const float* w, int64_t n, float* partials, float* out, void* stream);

/* Fused per-modality gating stage of XlinearFusion (models/model_modules.py:158-165), all m <= 3 modalities in ONE
 * single-workgroup launch:  h_i = relu(Wh_i v_i + bh_i); z_i = Wz_i [v_0|..|v_{m-1}] + bz_i;
 * gm_i = sigmoid(z_i) * h_i;  o_i = Dropout(relu(Wo_i gm_i + bo_i))  (dropout site i of `seed`).
 * forward writes h, z, gm, o ([B x sdim] each); backward reads them plus d_o and writes dv (incl. the v_cat path) and
 * every weight gradient.  All arrays are indexed by modality; entries >= m are ignored.  B * m * sdim <= 384. */
typedef struct mmf_xreduce_io {
  int32_t m, B, dim, sdim;
  const float* v[3];
  const float* Wh[3]; const float* bh[3];
  const float* Wz[3]; const float* bz[3];
  const float* Wo[3]; const float* bo[3];
  float* h[3]; float* z[3]; float* gm[3]; float* o[3];
  const float* d_o[3];
  float* dv[3];
  float* dWh[3]; float* dbh[3]; float* dWz[3]; float* dbz[3]; float* dWo[3]; float* dbo[3];
} mmf_xreduce_io;
int mmf_xreduce_forward(const mmf_xreduce_io* io, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);
int mmf_xreduce_backward(const mmf_xreduce_io* io, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage-2 building blocks: the embedding-level fusion models trained on the exported [B x 256] features
 * (models/nll_models_pretrained.py:13-197, models/coxranking_models_pretrained.py:14-200) and their batched losses.
 * ------------------------------------------------------------------------------------------- */
/* y = dropout(act(BatchNorm1d(x) [+ res])), x / y / res: [B x F].  training != 0: batch statistics (biased variance),
 * running_mean / running_var updated in place with `momentum` and the unbiased variance (torch semantics; B >= 2);
 * training == 0: running statistics.  save_mean / save_invstd [F] are what backward reads.
 * Replaces nn.BatchNorm1d (+ the ReLU / Dropout / residual add that follow it) in models/model_modules.py:5-49 and
 * models/nll_models_pretrained.py:82-90. */
int mmf_batchnorm_forward(const float* x, const float* res, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int32_t B, int32_t F, int32_t training,
                          float eps, float momentum, int32_t act, float drop_p, uint32_t seed, uint32_t site,
                          const uint32_t* seed_dev, float* y, float* save_mean, float* save_invstd, void* stream);
int mmf_batchnorm_backward(const float* dy, const float* y, const float* x, const float* gamma,
                           const float* save_mean, const float* save_invstd, int32_t B, int32_t F, int32_t training,
                           int32_t act, float drop_p, uint32_t seed, uint32_t site, const uint32_t* seed_dev,
                           float* dx, float* dres /* or NULL */, float* dgamma, float* dbeta, void* stream);
/* Highway mix, models/model_modules.py:21-25: y = sigmoid(zg) * relu(zn) + (1 - sigmoid(zg)) * zl, elementwise over n. */
int mmf_highway_mix_forward(const float* zg, const float* zn, const float* zl, int64_t n, float* y, void* stream);
int mmf_highway_mix_backward(const float* dy, const float* zg, const float* zn, const float* zl, int64_t n,
                             float* dzg, float* dzn, float* dzl, void* stream);
/* ranking_loss, utils/loss_utils.py:58-101 (a Python loop over all pairs): loss = -(mean | sum) over comparable pairs of
 * phi(risk_more - risk_less); phi 0 = sigmoid, 1 = relu; reduction 0 = mean, 1 = sum; 0 when no pair is comparable.
 * times: device double[B] (event times, or the bin labels for RankingNLLSurvLoss, loss_utils.py:160).  Writes the loss
 * and its gradient w.r.t. risks.  B >= 2. */
int mmf_ranking_loss(const float* risks, const double* times, const float* c, int32_t B, int32_t phi, int32_t reduction,
                     float* loss, float* d_risks, void* stream);
/* logits [B x K] -> hazards = sigmoid, S = cumprod(1 - hazards), Y_hat = argmax, risk = -sum_k S
 * (models/nll_models_pretrained.py:58-62,193-197).  K <= 32. */
int mmf_hazards_forward(const float* logits, int32_t B, int32_t K, float* hazards, float* S, int64_t* Y_hat, float* risk,
                        void* stream);
int mmf_hazards_backward(const float* g_hazards, const float* g_S, const float* g_risk /* each may be NULL */,
                         const float* hazards, int32_t B, int32_t K, float* dlogits, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Kernel trace: per-kernel device time of the attention-stack entry points, from HIP events recorded on the LAUNCH
 * stream around every kernel of a call whose desc->trace is set (bench.py's roofline leg).  A trace is a caller-owned
 * object (create / destroy); calls that carry the same trace must not run concurrently.  capacity = kernel launches
 * it can hold; further launches go unrecorded.  mmf_trace_dump synchronises on the recorded events, writes
 * "kernel_name launches total_ms" lines into buf, clears the records and returns the number of bytes written
 * (or needed when buf == NULL).
 * ------------------------------------------------------------------------------------------- */
typedef struct mmf_trace mmf_trace;
mmf_trace* mmf_trace_create(int32_t capacity);
void mmf_trace_destroy(mmf_trace* trace);
int mmf_trace_dump(mmf_trace* trace, char* buf, size_t buf_bytes);

/* Host-side restatement of the device dropout keep-hash (1 = kept).  For tests / mask inspection only. */
int mmf_dropout_keep_host(uint32_t seed, uint32_t site, uint32_t index, float p);

#ifdef __cplusplus
}
#endif
#endif /* MMF_AMIL_H */
