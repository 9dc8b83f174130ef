// C ABI (include/mmf_amil.h): argument checks, workspace carving, kernel sequencing.
// No allocation, no host synchronisation, no global mutable state except the (mutex-guarded)
// "dynamic LDS attribute already set" set.  The device-resident dropout seed and the kernel trace are per-call
// arguments; the trace of the call in progress is held in a thread_local for the launchers (ProfScope).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/mmf_amil.h"
#include "mmf_common.h"
#include "mmf_gemm_core.h"
#include "mmf_kernels.h"
#include "mmf_small.h"
#include "mmf_mlp.h"
#include "mmf_bf16.h"

namespace mmf {

int set_dyn_lds(const void* kern, int bytes) {
  if (bytes <= 48 * 1024) return MMF_OK;
  static std::mutex mu;
  static std::unordered_set<const void*> done;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count(kern)) return MMF_OK;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return MMF_ERR_LAUNCH;
  done.insert(kern);
  return MMF_OK;
}

}  // namespace mmf

// Caller-owned kernel trace (include/mmf_amil.h "Kernel trace"): HIP event pairs recorded on the launch stream.
struct mmf_trace {
  struct Rec { const char* name; hipEvent_t a, b; };
  std::mutex mu;
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;      // events are reused across dumps
  int cap = 0;
};

namespace mmf {

static thread_local mmf_trace* tls_trace = nullptr;
struct TraceScope {                  // the ABI entry that carries a trace makes it current for its launches
  mmf_trace* prev;
  explicit TraceScope(mmf_trace* t) : prev(tls_trace) { tls_trace = t; }
  ~TraceScope() { tls_trace = prev; }
};

void prof_begin(const char* name, hipStream_t st) {
  mmf_trace* t = tls_trace;
  if (!t) return;
  std::lock_guard<std::mutex> lock(t->mu);
  if ((int)t->recs.size() >= t->cap) return;
  auto take = [&](hipEvent_t& e) {
    if (!t->pool.empty()) { e = t->pool.back(); t->pool.pop_back(); return true; }
    return hipEventCreate(&e) == hipSuccess;
  };
  mmf_trace::Rec r{name, nullptr, nullptr};
  if (!take(r.a)) return;
  if (!take(r.b)) { t->pool.push_back(r.a); return; }     // the first event goes back to the pool, not lost
  hipEventRecord(r.a, st);
  t->recs.push_back(r);
}
void prof_end(hipStream_t st) {
  mmf_trace* t = tls_trace;
  if (!t) return;
  std::lock_guard<std::mutex> lock(t->mu);
  if (!t->recs.empty()) hipEventRecord(t->recs.back().b, st);
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int PREP_GROUPS = 512;

struct AmilWs {
  float *M_step, *dM_step;           // [H] each: pooled embedding and its gradient inside mmf_amil_nll_step
  unsigned long long* relu_bits;     // [ceil(N/16)][H/32][8]: h > 0 per element (LinearParams::relu_bits)
  float *h, *a, *b, *s_part, *partials, *stats, *p, *ds, *dbc_part, *du;
  float *slab_w1, *slab_wab, *cs_b1, *cs_bab, *cs_wc;
  float* kpart;                      // partial tiles of a K-split projection (LinearParams::kpart), or null
  int parts, groups, splits, k_per_split, mstk, tile;
  int splits_g, k_per_split_g;      // K split of the gate problem (d[Wa;Wb]): more, shorter splits than dW1
  size_t bytes;
};

static AmilWs carve(void* base, int64_t N, int L, int H, int D, int gated, bool infer = false) {
  AmilWs w{};
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* r = reinterpret_cast<float*>(p + off);
    off += align_up(nfloat * sizeof(float), 256);
    return r;
  };
  w.parts = gate_parts(D, gated, N);
  w.groups = pool_groups(N);
  w.mstk = gated ? 2 * D : D;
  const int td = tn_tile_dim(N, D);
  const int tiles = ((H + td - 1) / td) * ((L + td - 1) / td) + ((w.mstk + td - 1) / td) * ((H + td - 1) / td);
  int splits = tn_splits(N, tiles, td);
  w.tile = td;
  w.splits = splits;
  // splits are cut at multiples of 4 instances (one MFMA k-step of the TN tile), not of whole 32-instance chunks: all
  // splits then have the same length and end with a partial chunk whose empty fragment groups are skipped
  int64_t kps = (N + splits - 1) / splits;
  w.k_per_split = (int)((kps + 3) / 4 * 4);
  if (w.k_per_split < 4) w.k_per_split = 4;
  // A gate tile builds its A operand (dP from a, b, ds) in the staging path and lives ~8 % longer per K row than a
  // dW1 tile (stamps: 710 k vs 659 k cycles with equal splits), so the whole launch waited for the gate tiles.  The
  // workgroups the uniform split leaves over (256 - 6 x 42 = 4) go to the gate problem: 44 splits of 36 chunks beside
  // 42 of 38 at N = 50k.  Large-bag tile only, at most 12 % more splits.
  w.splits_g = splits; w.k_per_split_g = w.k_per_split;
  {
    static const int env = mmf::tune_int("MMF_TN_GATE_SPLITS", -1);   // tuning override
    const int t1 = ((H + td - 1) / td) * ((L + td - 1) / td), t2 = tiles - t1;
    int sg = splits;
    if (td == 256 && t2 > 0 && splits >= 8) {
      sg = (256 - t1 * splits) / t2;
      const int cap = splits + (splits * 12 + 99) / 100;
      if (sg > cap) sg = cap;
      if (sg < splits) sg = splits;
    }
    if (env > 0) sg = env;
    const int64_t kg = (N + sg - 1) / sg;
    int kpg = (int)((kg + 3) / 4 * 4);
    if (kpg < 4) kpg = 4;
    w.splits_g = sg; w.k_per_split_g = kpg;
  }
  w.M_step = take(H);
  w.dM_step = take(H);
  w.h = take((size_t)N * H);
  w.s_part = take((size_t)w.parts * N);
  w.partials = take((size_t)w.groups * (2 + H));
  w.stats = take(4);
  {
    const size_t kf = linear_ksplit_floats(N, H, L, 1, L);
    w.kpart = kf ? take(kf) : nullptr;
  }
  if (infer) {            // forward-only: nothing is kept for a backward (a, b stay null: K-gate skips their stores)
    w.bytes = off;
    return w;
  }
  w.relu_bits = reinterpret_cast<unsigned long long*>(take((size_t)((N + 15) / 16 + 2) * (H / 32) * 8 * 2));
  w.a = take((size_t)N * D);
  w.b = take(gated ? (size_t)N * D : 0);
  w.p = take((size_t)N);
  w.ds = take((size_t)N);
  w.dbc_part = take(PREP_GROUPS);
  w.du = take((size_t)N * H);
  w.slab_w1 = take((size_t)splits * H * L);
  w.slab_wab = take((size_t)w.splits_g * w.mstk * H);
  w.cs_b1 = take((size_t)splits * H);
  w.cs_bab = take((size_t)w.splits_g * w.mstk);
  w.cs_wc = take((size_t)w.splits_g * D);
  w.bytes = off;
  return w;
}

static int check_desc(const mmf_amil_desc* d, int elem_bytes = 4) {
  if (!d || !d->W1 || !d->b1 || !d->Wa || !d->ba || !d->Wc || !d->bc) return MMF_ERR_ARG;
  if (d->gated && (!d->Wb || !d->bb)) return MMF_ERR_ARG;
  if (d->N < 1) return MMF_ERR_SHAPE;
  if (d->L % KC != 0 || d->H % KC != 0) return MMF_ERR_SHAPE;
  if (d->H != 256 && d->H != 512 && d->H != 1024) return MMF_ERR_SHAPE;
  if (d->D % 128 != 0) return MMF_ERR_SHAPE;
  if (d->N * (int64_t)(d->H > d->D ? d->H : d->D) >= (int64_t)1 << 32) return MMF_ERR_SHAPE;  // 32-bit mask index
  // buffer loads: 32-bit byte offsets, and the "reads as zero" sentinel is 2^31 => every operand < 2 GiB
  const int64_t widest = d->L > 2 * d->D ? d->L : 2 * d->D;
  if (d->N * widest * elem_bytes >= (int64_t)1 << 31) return MMF_ERR_SHAPE;
  if (d->p_h < 0.f || d->p_h >= 1.f || d->p_att < 0.f || d->p_att >= 1.f) return MMF_ERR_ARG;
  if (d->gemm != MMF_GEMM_F32 && d->gemm != MMF_GEMM_BF16X3) return MMF_ERR_ARG;
  return MMF_OK;
}

// ---- bf16-storage path (mmf_bf16.h) ----------------------------------------------------------
struct AmilWsBf {
  float *M_step, *dM_step;
  bf16_t *w1, *wab, *wabT, *h, *a, *b, *du, *dP;
  float *s_part, *partials, *stats, *p, *ds, *dbc_part, *dwc_part;
  float *slab_w1, *slab_wab, *cs_b1, *cs_bab;
  int parts, groups, splits, k_per_split, mstk, dbc_cap;
  size_t bytes;
};

static AmilWsBf carve_bf16(void* base, int64_t N, int L, int H, int D, int gated, bool infer = false) {
  AmilWsBf w{};
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take_b = [&](size_t nbytes) {
    char* r = p + off;
    off += align_up(nbytes, 256);
    return r;
  };
  auto take16 = [&](size_t n) { return reinterpret_cast<bf16_t*>(take_b(n * 2)); };
  auto take32 = [&](size_t n) { return reinterpret_cast<float*>(take_b(n * 4)); };
  w.parts = gate_parts_bf16(D, gated);
  w.groups = pool_groups(N);
  w.mstk = gated ? 2 * D : D;
  const int td = TNB_TILE;
  const int tiles = ((H + td - 1) / td) * ((L + td - 1) / td) + ((w.mstk + td - 1) / td) * ((H + td - 1) / td);
  w.splits = tn_bf16_splits(N, tiles);
  const int64_t kps = (N + w.splits - 1) / w.splits;
  w.k_per_split = (int)((kps + TNB_KCH - 1) / TNB_KCH * TNB_KCH);
  w.dbc_cap = dh_bf16_row_tiles(N);
  w.M_step = take32(H);
  w.dM_step = take32(H);
  w.w1 = take16((size_t)H * L);
  w.wab = take16((size_t)w.mstk * H);
  w.wabT = take16((size_t)H * w.mstk);
  w.h = take16((size_t)N * H);
  w.s_part = take32((size_t)w.parts * N);
  {
    const int tiles = fused_fwd_tiles(N);          // the fused forward writes one pooling partial per 128-row tile
    w.partials = take32((size_t)(w.groups > tiles ? w.groups : tiles) * (2 + H));
  }
  w.stats = take32(4);
  if (infer) {
    w.bytes = off;
    return w;
  }
  w.a = take16((size_t)N * D);
  w.b = take16(gated ? (size_t)N * D : 0);
  w.du = take16((size_t)N * H);
  w.dP = take16((size_t)N * w.mstk);
  w.p = take32((size_t)N);
  w.ds = take32((size_t)N);
  w.dbc_part = take32((size_t)w.dbc_cap);
  w.dwc_part = take32((size_t)w.dbc_cap * D);
  w.slab_w1 = take32((size_t)w.splits * H * L);
  w.slab_wab = take32((size_t)w.splits * w.mstk * H);
  w.cs_b1 = take32((size_t)w.splits * H);
  w.cs_bab = take32((size_t)w.splits * w.mstk);
  w.bytes = off;
  return w;
}

static int check_desc_bf16(const mmf_amil_desc* d) {
  if (int e = check_desc(d, 2)) return e;        // the bag and every [N x *] activation are 2-byte here
  if (d->L % 64 != 0 || d->H % 256 != 0) return MMF_ERR_SHAPE;
  return MMF_OK;
}

}  // namespace mmf

using namespace mmf;

extern "C" {

int mmf_abi_version(void) { return 11; }

const char* mmf_strerror(int code) {
  switch (code) {
    case MMF_OK: return "ok";
    case MMF_ERR_ARG: return "invalid argument (null pointer or bad flag)";
    case MMF_ERR_SHAPE: return "unsupported shape (need L,H % 32 == 0, H in {256,512,1024}, D % 128 == 0, K % 32 == 0, every operand < 2 GiB)";
    case MMF_ERR_ALIGN: return "pointer or leading dimension not 16-byte aligned";
    case MMF_ERR_WORKSPACE: return "workspace too small (see mmf_*_workspace_bytes)";
    case MMF_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown mmf error";
  }
}

size_t mmf_amil_workspace_bytes(int64_t N, int32_t L, int32_t H, int32_t D, int32_t gated) {
  if (N < 1) N = 1;
  return carve(nullptr, N, L, H, D, gated).bytes;
}

static int amil_forward_impl(const mmf_amil_desc* d, const float* x, void* workspace, size_t workspace_bytes,
                             float* M, float* A_raw, void* stream, bool infer, const HeadTail* tail = nullptr) {
  if (int e = check_desc(d)) return e;
  if (!x || !workspace || !A_raw) return MMF_ERR_ARG;
  if (!aligned16(x) || !aligned16(workspace) || !aligned16(d->W1) || !aligned16(d->Wa) || (d->gated && !aligned16(d->Wb)))
    return MMF_ERR_ALIGN;
  AmilWs w = carve(workspace, d->N, d->L, d->H, d->D, d->gated, infer);
  if (w.bytes > workspace_bytes) return MMF_ERR_WORKSPACE;
  if (!M) M = w.M_step;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TraceScope ts(d->trace);
  const uint32_t* const seed_dev = d->seed_dev;

  LinearParams lp{};
  lp.x[0] = x; lp.nseg = 1; lp.kseg = d->L; lp.ldx = d->L;
  lp.w = d->W1; lp.bias = d->b1; lp.y = w.h;
  lp.M = d->N; lp.N = d->H; lp.K = d->L;
  lp.act = ACT_RELU; lp.drop_p = d->p_h; lp.drop_key = drop_key(d->seed, 0); lp.seed_dev = seed_dev;
  lp.relu_bits = infer ? nullptr : w.relu_bits;
  lp.allow_half = 1; lp.concurrent = d->concurrent ? 1 : 0;
  lp.split = d->gemm == MMF_GEMM_BF16X3;
  lp.kpart = w.kpart; lp.ktick = d->sync; lp.ktick_words = d->sync ? d->sync_words : 0;
  if (int e = launch_linear(lp, st)) return e;

  GateFwdParams gp{};
  gp.h = w.h; gp.Wa = d->Wa; gp.ba = d->ba; gp.Wb = d->Wb; gp.bb = d->bb; gp.Wc = d->Wc;
  gp.a = w.a; gp.b = w.b; gp.s_part = w.s_part;
  gp.N = d->N; gp.H = d->H; gp.D = d->D; gp.gated = d->gated;
  gp.drop_p = d->p_att; gp.key_a = drop_key(d->seed, 1); gp.key_b = drop_key(d->seed, 2); gp.seed_dev = seed_dev;
  gp.split = d->gemm == MMF_GEMM_BF16X3;
  if (int e = launch_gate_fwd(gp, st)) return e;

  PoolParams pp{};
  pp.s_part = w.s_part; pp.n_parts = w.parts; pp.bc = d->bc; pp.h = w.h; pp.N = d->N; pp.H = d->H;
  pp.A_raw = A_raw; pp.partials = w.partials; pp.M = M; pp.stats = w.stats;
  if (tail) { pp.tail = *tail; pp.tail.dM = w.dM_step; }
  return launch_pool(pp, st);
}

int mmf_amil_forward(const mmf_amil_desc* d, const float* x, void* workspace, size_t workspace_bytes,
                     float* M, float* A_raw, void* stream) {
  return amil_forward_impl(d, x, workspace, workspace_bytes, M, A_raw, stream, false);
}

size_t mmf_amil_infer_workspace_bytes(int64_t N, int32_t L, int32_t H, int32_t D, int32_t gated) {
  if (N < 1) N = 1;
  return carve(nullptr, N, L, H, D, gated, true).bytes;
}

int mmf_amil_infer(const mmf_amil_desc* d, const float* x, void* workspace, size_t workspace_bytes,
                   float* M, float* A_raw, void* stream) {
  return amil_forward_impl(d, x, workspace, workspace_bytes, M, A_raw, stream, true);
}

static int amil_backward_impl(const mmf_amil_desc* d, const float* x, void* workspace, size_t workspace_bytes,
                              const float* M, const float* A_raw, const float* dM, const float* gA,
                              const mmf_amil_grads* g, void* stream, int accumulate) {
  if (int e = check_desc(d)) return e;
  if (!x || !workspace || !A_raw || !g) return MMF_ERR_ARG;
  if (!g->dW1 || !g->db1 || !g->dWa || !g->dba || !g->dWc || !g->dbc) return MMF_ERR_ARG;
  if (d->gated && (!g->dWb || !g->dbb)) return MMF_ERR_ARG;
  if (!aligned16(g->dW1) || !aligned16(g->dWa) || (d->gated && !aligned16(g->dWb)) || (g->dx && !aligned16(g->dx)))
    return MMF_ERR_ALIGN;
  AmilWs w = carve(workspace, d->N, d->L, d->H, d->D, d->gated);
  if (w.bytes > workspace_bytes) return MMF_ERR_WORKSPACE;
  if (!M) M = w.M_step;          // inside mmf_amil_nll_step the pooled embedding and its gradient live in the workspace
  if (!dM) dM = w.dM_step;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TraceScope ts(d->trace);
  const uint32_t* const seed_dev = d->seed_dev;

  GateBwdCtx gc{};
  gc.a = w.a; gc.b = w.b; gc.ds = w.ds; gc.Wc = d->Wc; gc.D = d->D; gc.gated = d->gated;
  gc.drop_p = d->p_att; gc.key_a = drop_key(d->seed, 1); gc.key_b = drop_key(d->seed, 2); gc.seed_dev = seed_dev;

  BwdDhParams dp{};
  dp.g = gc; dp.Wa = d->Wa; dp.Wb = d->Wb; dp.p = w.p; dp.dM = dM; dp.h = w.h; dp.du = w.du;
  dp.relu_bits = w.relu_bits;
  dp.allow_half = 1; dp.concurrent = d->concurrent ? 1 : 0;
  dp.split = d->gemm == MMF_GEMM_BF16X3;
  dp.N = d->N; dp.H = d->H; dp.scale_h = d->p_h > 0.f ? 1.0f / (1.0f - d->p_h) : 1.0f;

  // K-prep (softmax weights, ds) either fused into the wide K-dh kernel or as its own launch
  int dbc_groups = bwd_dh_fused_groups(d->N, d->H, 1, d->D, d->gated, dp.split, d->concurrent ? 1 : 0);
  if (dbc_groups > 0 && dbc_groups <= PREP_GROUPS) {
    dp.fused_prep = 1;
    dp.A_raw = A_raw; dp.stats = w.stats; dp.Mpool = M; dp.gA = gA;
    dp.p_out = w.p; dp.ds_out = w.ds; dp.dbc_part = w.dbc_part;
  } else {
    BwdPrepParams bp{};
    bp.h = w.h; bp.A_raw = A_raw; bp.stats = w.stats; bp.dM = dM; bp.M = M; bp.gA = gA;
    bp.N = d->N; bp.H = d->H; bp.p = w.p; bp.ds = w.ds; bp.dbc_part = w.dbc_part;
    bp.n_groups = (int)((d->N + 3) / 4 < PREP_GROUPS ? (d->N + 3) / 4 : PREP_GROUPS);
    dbc_groups = bp.n_groups;
    if (int e = launch_bwd_prep(bp, st)) return e;
  }
  if (int e = launch_bwd_dh(dp, st)) return e;

  if (g->dx) {   // d(input) = du . W1   (radio: the input is reduce_dim's output)
    NnParams np{};
    np.A = w.du; np.lda = d->H; np.B = d->W1; np.ldb = d->L; np.C = g->dx; np.ldc = d->L;
    np.M = d->N; np.N = d->L; np.K = d->H;
    if (int e = launch_nn(np, st)) return e;
  }

  TnParams tp{};
  tp.nprob = 2; tp.K = d->N; tp.splits = w.splits; tp.k_per_split = w.k_per_split; tp.g = gc; tp.tile = w.tile;
  tp.split = dp.split;
  TnProblem& q1 = tp.prob[0];   // dW1[H x L] = du^T . x ; db1 = colsum(du)
  q1.kind = TN_A_PLAIN; q1.A = w.du; q1.lda = d->H; q1.M = d->H;
  q1.B = x; q1.ldb = d->L; q1.Ncols = d->L;
  q1.out = w.slab_w1; q1.split_stride = (size_t)d->H * d->L; q1.ldc = d->L;
  q1.colsum = w.cs_b1; q1.colsum_stride = d->H; q1.colsum2 = nullptr; q1.colsum2_stride = 0;
  TnProblem& q2 = tp.prob[1];   // dWab[(2)D x H] = dP^T . h ; (dba|dbb) = colsum(dP) ; dWc = colsum(ds.a_d.b_d)
  q2.kind = TN_A_GATE; q2.A = nullptr; q2.lda = 0; q2.M = w.mstk;
  q2.B = w.h; q2.ldb = d->H; q2.Ncols = d->H;
  q2.out = w.slab_wab; q2.split_stride = (size_t)w.mstk * d->H; q2.ldc = d->H;
  q2.colsum = w.cs_bab; q2.colsum_stride = w.mstk; q2.colsum2 = w.cs_wc; q2.colsum2_stride = d->D;
  q2.splits = w.splits_g; q2.k_per_split = w.k_per_split_g;
  if (int e = launch_tn(tp, st)) return e;

  ReduceParams rp{};
  int n = 0;
  auto seg = [&](const float* in, float* out, int len, int nsplit, size_t stride) {
    rp.seg[n].in = in; rp.seg[n].out = out; rp.seg[n].len = len; rp.seg[n].nsplit = nsplit; rp.seg[n].stride = stride; ++n;
  };
  seg(w.slab_w1, g->dW1, d->H * d->L, w.splits, (size_t)d->H * d->L);
  seg(w.slab_wab, g->dWa, d->D * d->H, w.splits_g, (size_t)w.mstk * d->H);
  if (d->gated) seg(w.slab_wab + (size_t)d->D * d->H, g->dWb, d->D * d->H, w.splits_g, (size_t)w.mstk * d->H);
  seg(w.cs_b1, g->db1, d->H, w.splits, d->H);
  seg(w.cs_bab, g->dba, d->D, w.splits_g, w.mstk);
  if (d->gated) seg(w.cs_bab + d->D, g->dbb, d->D, w.splits_g, w.mstk);
  seg(w.cs_wc, g->dWc, d->D, w.splits_g, d->D);
  seg(w.dbc_part, g->dbc, 1, dbc_groups, 1);
  rp.nseg = n;
  rp.accumulate = accumulate;
  return launch_reduce(rp, st);
}

int mmf_amil_backward(const mmf_amil_desc* d, const float* x, void* workspace, size_t workspace_bytes,
                      const float* M, const float* A_raw, const float* dM, const float* gA,
                      const mmf_amil_grads* g, void* stream) {
  if (!M || !dM) return MMF_ERR_ARG;
  return amil_backward_impl(d, x, workspace, workspace_bytes, M, A_raw, dM, gA, g, stream, 0);
}

size_t mmf_amil_bf16_workspace_bytes(int64_t N, int32_t L, int32_t H, int32_t D, int32_t gated) {
  if (N < 1) N = 1;
  return carve_bf16(nullptr, N, L, H, D, gated).bytes;
}

static int amil_bf16_forward_impl(const mmf_amil_desc* d, const uint16_t* x, void* workspace, size_t workspace_bytes,
                                  float* M, float* A_raw, void* stream, bool infer, const HeadTail* tail = nullptr) {
  if (int e = check_desc_bf16(d)) return e;
  if (!x || !workspace || !A_raw) return MMF_ERR_ARG;
  if (!aligned16(x) || !aligned16(workspace)) return MMF_ERR_ALIGN;
  AmilWsBf w = carve_bf16(workspace, d->N, d->L, d->H, d->D, d->gated, infer);
  if (w.bytes > workspace_bytes) return MMF_ERR_WORKSPACE;
  if (!M) M = w.M_step;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TraceScope ts(d->trace);
  const uint32_t* const seed_dev = d->seed_dev;
  HeadTail tl{};
  if (tail) { tl = *tail; tl.dM = w.dM_step; }

  CvtParams cp{};
  cp.nseg = 0;
  auto cvt = [&](const float* src, bf16_t* dst, int rows, int cols, int dst_ld, int c0, int transpose) {
    cp.seg[cp.nseg++] = CvtSeg{src, dst, rows, cols, dst_ld, c0, transpose, 0};
  };
  // the fused forward's second form reads W1 / [Wa ; Wb] in MFMA-fragment order (same bytes, same workspace slots)
  const bool fused2 = d->gated && fused_fwd2_ok(d->N, d->L, d->H, d->D);
  cvt(d->W1, w.w1, d->H, d->L, d->L, 0, fused2 ? 3 : 0);
  cvt(d->Wa, w.wab, d->D, d->H, d->H, 0, fused2 ? 4 : 0);
  const bool dh2 = !infer && dh2_bf16_ok(d->N, d->H, d->D, d->gated);         // K-dh's second form: [Wa ; Wb]^T in fragment order
  if (!infer) cvt(d->Wa, w.wabT, d->D, d->H, w.mstk, 0, dh2 ? 5 : (d->gated ? 2 : 1));     // K-dh's k order (mmf_amil_bf16.hip: LoadPB)
  if (d->gated) {
    if (fused2) cvt(d->Wb, w.wab, d->D, d->H, d->H, 16, 4);
    else cvt(d->Wb, w.wab + (size_t)d->D * d->H, d->D, d->H, d->H, 0, 0);
    if (!infer) cvt(d->Wb, w.wabT, d->D, d->H, w.mstk, 32, dh2 ? 5 : 2);
  }
  if (int e = launch_cvt_bf16(cp, st)) return e;

  if (fused2 || (d->gated && d->D == 256 && fused_fwd_ok(d->N, d->L, d->H, d->D))) {   // `small` gated stack: one kernel for projection + scoring + pooling partials
    FusedFwdParams fp{};
    fp.x = x; fp.w1 = w.w1; fp.b1 = d->b1;
    fp.Wa = w.wab; fp.Wb = d->gated ? w.wab + (size_t)d->D * d->H : nullptr;
    fp.w1f = w.w1; fp.wabf = w.wab;
    fp.ba = d->ba; fp.bb = d->bb; fp.Wc = d->Wc; fp.bc = d->bc;
    fp.h = infer ? nullptr : w.h; fp.a = w.a; fp.b = w.b;       // a / b are null when carved for inference
    fp.A_raw = A_raw; fp.partials = w.partials;
    fp.N = d->N; fp.L = d->L; fp.D = d->D;
    fp.p_h = d->p_h; fp.p_att = d->p_att;
    fp.key_h = drop_key(d->seed, 0); fp.key_a = drop_key(d->seed, 1); fp.key_b = drop_key(d->seed, 2);
    fp.seed_dev = seed_dev;
    if (fused2) {
      if (int e = launch_fused_fwd2_bf16(fp, d->gated, st)) return e;
    } else
    if (int e = launch_fused_fwd_bf16(fp, d->gated, st)) return e;
    // (Tried: sending the rows of a sparse last round -- 782 tiles at 100k = 3 rounds of 256 + 14 -- through the three
    // unfused kernels instead.  The fused kernel drops 173 -> 141 us, but the three small launches cost 48 us.)
    PoolParams pm{};
    pm.N = d->N; pm.H = d->H; pm.partials = w.partials; pm.M = M; pm.stats = w.stats;
    pm.n_groups = fused_fwd_tiles(d->N);
    pm.tail = tl;
    return launch_pool_merge(pm, st);
  }

  LinearBfParams lp{};
  lp.x = x; lp.w = w.w1; lp.bias = d->b1; lp.y = w.h;
  lp.M = d->N; lp.N = d->H; lp.K = d->L;
  lp.drop_p = d->p_h; lp.drop_key = drop_key(d->seed, 0); lp.seed_dev = seed_dev;
  if (int e = launch_linear_bf16(lp, st)) return e;

  GateBfParams gp{};
  gp.h = w.h; gp.Wa = w.wab; gp.Wb = d->gated ? w.wab + (size_t)d->D * d->H : nullptr;
  gp.ba = d->ba; gp.bb = d->bb; gp.Wc = d->Wc;
  gp.a = w.a; gp.b = w.b; gp.s_part = w.s_part;
  gp.N = d->N; gp.H = d->H; gp.D = d->D; gp.gated = d->gated;
  gp.drop_p = d->p_att; gp.key_a = drop_key(d->seed, 1); gp.key_b = drop_key(d->seed, 2); gp.seed_dev = seed_dev;
  if (int e = launch_gate_bf16(gp, st)) return e;

  PoolBfParams pb{};
  PoolParams& pp = pb.base;
  pp.s_part = w.s_part; pp.n_parts = w.parts; pp.bc = d->bc; pp.h = nullptr; pp.N = d->N; pp.H = d->H;
  pp.A_raw = A_raw; pp.partials = w.partials; pp.M = M; pp.stats = w.stats;
  pp.tail = tl;
  pb.h = w.h;
  return launch_pool_bf16(pb, st);
}

int mmf_amil_bf16_forward(const mmf_amil_desc* d, const uint16_t* x, void* workspace, size_t workspace_bytes,
                          float* M, float* A_raw, void* stream) {
  return amil_bf16_forward_impl(d, x, workspace, workspace_bytes, M, A_raw, stream, false);
}

size_t mmf_amil_bf16_infer_workspace_bytes(int64_t N, int32_t L, int32_t H, int32_t D, int32_t gated) {
  if (N < 1) N = 1;
  return carve_bf16(nullptr, N, L, H, D, gated, true).bytes;
}

int mmf_amil_bf16_infer(const mmf_amil_desc* d, const uint16_t* x, void* workspace, size_t workspace_bytes,
                        float* M, float* A_raw, void* stream) {
  return amil_bf16_forward_impl(d, x, workspace, workspace_bytes, M, A_raw, stream, true);
}

static int amil_bf16_backward_impl(const mmf_amil_desc* d, const uint16_t* x, void* workspace, size_t workspace_bytes,
                                   const float* M, const float* A_raw, const float* dM, const float* gA,
                                   const mmf_amil_grads* g, void* stream, int accumulate) {
  if (int e = check_desc_bf16(d)) return e;
  if (!x || !workspace || !A_raw || !g) return MMF_ERR_ARG;
  if (!g->dW1 || !g->db1 || !g->dWa || !g->dba || !g->dWc || !g->dbc) return MMF_ERR_ARG;
  if (d->gated && (!g->dWb || !g->dbb)) return MMF_ERR_ARG;
  if (g->dx) return MMF_ERR_ARG;     // the bf16 bag is a leaf: no input gradient on this path
  if (!aligned16(g->dW1) || !aligned16(g->dWa) || (d->gated && !aligned16(g->dWb))) return MMF_ERR_ALIGN;
  AmilWsBf w = carve_bf16(workspace, d->N, d->L, d->H, d->D, d->gated);
  if (w.bytes > workspace_bytes) return MMF_ERR_WORKSPACE;
  if (!M) M = w.M_step;
  if (!dM) dM = w.dM_step;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TraceScope ts(d->trace);
  const uint32_t* const seed_dev = d->seed_dev;

  GateBwdBf gc{};
  gc.a = w.a; gc.b = w.b; gc.ds = w.ds; gc.Wc = d->Wc; gc.D = d->D; gc.gated = d->gated;
  gc.drop_p = d->p_att; gc.key_a = drop_key(d->seed, 1); gc.key_b = drop_key(d->seed, 2); gc.seed_dev = seed_dev;

  DhBfParams dp{};
  dp.g = gc; dp.WabT = w.wabT; dp.dM = dM; dp.h = w.h; dp.du = w.du;
  dp.N = d->N; dp.H = d->H; dp.scale_h = d->p_h > 0.f ? 1.0f / (1.0f - d->p_h) : 1.0f;
  dp.A_raw = A_raw; dp.stats = w.stats; dp.Mpool = M; dp.gA = gA;
  dp.p_out = w.p; dp.ds_out = w.ds; dp.dbc_part = w.dbc_part; dp.dP = w.dP; dp.dwc_part = w.dwc_part;
  if (dh2_bf16_ok(d->N, d->H, d->D, d->gated)) {
    if (int e = launch_dh2_bf16(dp, st)) return e;
  } else if (int e = launch_dh_bf16(dp, st)) return e;
  const int ntn = d->H / 256;

  TnBfParams tp{};
  tp.nprob = 2; tp.K = d->N; tp.splits = w.splits; tp.k_per_split = w.k_per_split;
  TnBfProblem& q1 = tp.prob[0];   // dW1[H x L] = du^T . x ; db1 = colsum(du)
  q1.A = w.du; q1.lda = d->H; q1.M = d->H;
  q1.B = x; q1.ldb = d->L; q1.Ncols = d->L;
  q1.out = w.slab_w1; q1.split_stride = (size_t)d->H * d->L; q1.ldc = d->L;
  q1.colsum = w.cs_b1; q1.colsum_stride = d->H;
  TnBfProblem& q2 = tp.prob[1];   // dWab[(2)D x H] = dP^T . h ; (dba|dbb) = colsum(dP)
  q2.A = w.dP; q2.lda = w.mstk; q2.M = w.mstk;
  q2.B = w.h; q2.ldb = d->H; q2.Ncols = d->H;
  q2.out = w.slab_wab; q2.split_stride = (size_t)w.mstk * d->H; q2.ldc = d->H;
  q2.colsum = w.cs_bab; q2.colsum_stride = w.mstk;
  if (int e = launch_tn_bf16(tp, st)) return e;

  ReduceParams rp{};
  int n = 0;
  auto seg = [&](const float* in, float* out, int len, int nsplit, size_t stride) {
    rp.seg[n].in = in; rp.seg[n].out = out; rp.seg[n].len = len; rp.seg[n].nsplit = nsplit; rp.seg[n].stride = stride; ++n;
  };
  seg(w.slab_w1, g->dW1, d->H * d->L, w.splits, (size_t)d->H * d->L);
  seg(w.slab_wab, g->dWa, d->D * d->H, w.splits, (size_t)w.mstk * d->H);
  if (d->gated) seg(w.slab_wab + (size_t)d->D * d->H, g->dWb, d->D * d->H, w.splits, (size_t)w.mstk * d->H);
  seg(w.cs_b1, g->db1, d->H, w.splits, d->H);
  seg(w.cs_bab, g->dba, d->D, w.splits, w.mstk);
  if (d->gated) seg(w.cs_bab + d->D, g->dbb, d->D, w.splits, w.mstk);
  seg(w.dwc_part, g->dWc, d->D, dh_bf16_tiles_used(d->N, ntn), d->D);
  seg(w.dbc_part, g->dbc, 1, dh_bf16_tiles_used(d->N, ntn), 1);
  rp.nseg = n;
  rp.accumulate = accumulate;
  return launch_reduce(rp, st);
}

int mmf_amil_bf16_backward(const mmf_amil_desc* d, const uint16_t* x, void* workspace, size_t workspace_bytes,
                           const float* M, const float* A_raw, const float* dM, const float* gA,
                           const mmf_amil_grads* g, void* stream) {
  if (!M || !dM) return MMF_ERR_ARG;
  return amil_bf16_backward_impl(d, x, workspace, workspace_bytes, M, A_raw, dM, gA, g, stream, 0);
}

// ---- attention stack + hazard head [+ nll_surv + the whole backward] in one call ------------------------------
static int head_tail_of(const mmf_surv_head* h, const mmf_nll_target* t, int H, HeadTail& tl) {
  if (!h || !h->Wk || !h->bk || !h->logits || !h->hazards || !h->S || !h->Y_hat) return MMF_ERR_ARG;
  if (h->K < 1 || h->K > 32) return MMF_ERR_SHAPE;
  tl = HeadTail{};
  tl.Wk = h->Wk; tl.bk = h->bk; tl.K = h->K;
  tl.logits = h->logits; tl.hazards = h->hazards; tl.S = h->S; tl.Y_hat = h->Y_hat; tl.risk = h->risk;
  if (t) {
    if (!t->Y || !t->c || !t->loss || !t->dWk || !t->dbk) return MMF_ERR_ARG;
    tl.Y = t->Y; tl.c = t->c; tl.alpha = t->alpha; tl.eps = t->eps; tl.loss_scale = t->loss_scale;
    tl.loss = t->loss; tl.dWk = t->dWk; tl.dbk = t->dbk; tl.accumulate = t->accumulate;
  }
  (void)H;
  return MMF_OK;
}

int mmf_amil_head_forward(const mmf_amil_desc* d, const void* x, int32_t x_bf16, void* workspace, size_t workspace_bytes,
                          const mmf_surv_head* head, float* M, float* A_raw, void* stream) {
  if (!d || !M) return MMF_ERR_ARG;
  HeadTail tl;
  if (int e = head_tail_of(head, nullptr, d->H, tl)) return e;
  return x_bf16 ? amil_bf16_forward_impl(d, static_cast<const uint16_t*>(x), workspace, workspace_bytes, M, A_raw, stream, false, &tl)
                : amil_forward_impl(d, static_cast<const float*>(x), workspace, workspace_bytes, M, A_raw, stream, false, &tl);
}

int mmf_amil_nll_step(const mmf_amil_desc* d, const void* x, int32_t x_bf16, void* workspace, size_t workspace_bytes,
                      const mmf_surv_head* head, const mmf_nll_target* target, float* A_raw,
                      const mmf_amil_grads* grads, void* stream) {
  if (!d || !target || !grads) return MMF_ERR_ARG;
  HeadTail tl;
  if (int e = head_tail_of(head, target, d->H, tl)) return e;
  const int acc = target->accumulate ? 1 : 0;
  if (x_bf16) {
    const uint16_t* xb = static_cast<const uint16_t*>(x);
    if (int e = amil_bf16_forward_impl(d, xb, workspace, workspace_bytes, nullptr, A_raw, stream, false, &tl)) return e;
    return amil_bf16_backward_impl(d, xb, workspace, workspace_bytes, nullptr, A_raw, nullptr, nullptr, grads, stream, acc);
  }
  const float* xf = static_cast<const float*>(x);
  if (int e = amil_forward_impl(d, xf, workspace, workspace_bytes, nullptr, A_raw, stream, false, &tl)) return e;
  return amil_backward_impl(d, xf, workspace, workspace_bytes, nullptr, A_raw, nullptr, nullptr, grads, stream, acc);
}

int mmf_surv_head_nll_step(const float* feat, int32_t F, const mmf_surv_head* head, const mmf_nll_target* target,
                           float* dfeat, void* stream) {
  if (!feat || !target || !dfeat) return MMF_ERR_ARG;
  if (F < 1 || F > 1024) return MMF_ERR_SHAPE;
  PoolParams p{};
  if (int e = head_tail_of(head, target, F, p.tail)) return e;
  p.tail.dM = dfeat;
  p.M = const_cast<float*>(feat);        // read only: the launch neither merges nor stores M
  p.H = F;
  return launch_head_tail(p, static_cast<hipStream_t>(stream));
}

// ---- standalone attention scorer: Attn_Net / Attn_Net_Gated .forward(x) -> (A, x) ------------------------------
namespace mmf {
struct AttnWs {
  float *a, *b, *s_part, *slab, *cs_bab, *cs_wc;
  int parts, mstk, splits, k_per_split, tile;
  size_t bytes;
};
static AttnWs carve_attn(void* base, int64_t N, int H, int D, int gated) {
  AttnWs w{};
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* r = reinterpret_cast<float*>(p + off);
    off += align_up(nfloat * sizeof(float), 256);
    return r;
  };
  w.parts = gate_parts(D, gated, N);
  w.mstk = gated ? 2 * D : D;
  w.tile = tn_tile_dim(N, D);
  const int td = w.tile;
  const int dt = gated ? td / 2 : td;
  const int tiles = ((D + dt - 1) / dt) * ((H + td - 1) / td);
  w.splits = tn_splits(N, tiles, td);
  const int64_t kps = (N + w.splits - 1) / w.splits;
  w.k_per_split = (int)((kps + 3) / 4 * 4);
  if (w.k_per_split < 4) w.k_per_split = 4;
  w.a = take((size_t)N * D);
  w.b = take(gated ? (size_t)N * D : 0);
  w.s_part = take((size_t)w.parts * N);
  w.slab = take((size_t)w.splits * w.mstk * H);
  w.cs_bab = take((size_t)w.splits * w.mstk);
  w.cs_wc = take((size_t)w.splits * D);
  w.bytes = off;
  return w;
}
static int check_attn(const mmf_amil_desc* d) {
  if (!d || !d->Wa || !d->ba || !d->Wc || !d->bc) return MMF_ERR_ARG;
  if (d->gated && (!d->Wb || !d->bb)) return MMF_ERR_ARG;
  if (d->N < 1 || d->H % KC != 0 || d->D % 32 != 0) return MMF_ERR_SHAPE;
  const int64_t widest = d->H > 2 * d->D ? d->H : 2 * d->D;
  if (d->N * widest * 4 >= (int64_t)1 << 31) return MMF_ERR_SHAPE;
  if (d->p_att < 0.f || d->p_att >= 1.f) return MMF_ERR_ARG;
  return MMF_OK;
}
}  // namespace mmf

size_t mmf_attn_net_workspace_bytes(int64_t N, int32_t H, int32_t D, int32_t gated) {
  if (N < 1) N = 1;
  return carve_attn(nullptr, N, H, D, gated).bytes;
}

int mmf_attn_net_forward(const mmf_amil_desc* d, const float* x, void* workspace, size_t workspace_bytes, float* A,
                         void* stream) {
  if (int e = check_attn(d)) return e;
  if (!x || !workspace || !A) return MMF_ERR_ARG;
  if (!aligned16(x) || !aligned16(workspace) || !aligned16(d->Wa) || (d->gated && !aligned16(d->Wb))) return MMF_ERR_ALIGN;
  AttnWs w = carve_attn(workspace, d->N, d->H, d->D, d->gated);
  if (w.bytes > workspace_bytes) return MMF_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TraceScope ts(d->trace);
  GateFwdParams gp{};
  gp.h = x; gp.Wa = d->Wa; gp.ba = d->ba; gp.Wb = d->Wb; gp.bb = d->bb; gp.Wc = d->Wc;
  gp.a = w.a; gp.b = w.b; gp.s_part = w.s_part;
  gp.N = d->N; gp.H = d->H; gp.D = d->D; gp.gated = d->gated;
  gp.drop_p = d->p_att; gp.key_a = drop_key(d->seed, 1); gp.key_b = drop_key(d->seed, 2); gp.seed_dev = d->seed_dev;
  if (int e = launch_gate_fwd(gp, st)) return e;
  return launch_score_sum(w.s_part, w.parts, d->bc, A, d->N, st);
}

int mmf_attn_net_backward(const mmf_amil_desc* d, const float* x, void* workspace, size_t workspace_bytes,
                          const float* gA, const mmf_amil_grads* g, void* stream) {
  if (int e = check_attn(d)) return e;
  if (!x || !workspace || !gA || !g) return MMF_ERR_ARG;
  if (!g->dWa || !g->dba || !g->dWc || !g->dbc || (d->gated && (!g->dWb || !g->dbb))) return MMF_ERR_ARG;
  if (!aligned16(g->dWa) || (d->gated && !aligned16(g->dWb)) || (g->dx && !aligned16(g->dx))) return MMF_ERR_ALIGN;
  AttnWs w = carve_attn(workspace, d->N, d->H, d->D, d->gated);
  if (w.bytes > workspace_bytes) return MMF_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TraceScope ts(d->trace);
  GateBwdCtx gc{};      // ds_i = dL/dA_i: the scorer has no softmax behind it here
  gc.a = w.a; gc.b = w.b; gc.ds = gA; gc.Wc = d->Wc; gc.D = d->D; gc.gated = d->gated;
  gc.drop_p = d->p_att; gc.key_a = drop_key(d->seed, 1); gc.key_b = drop_key(d->seed, 2); gc.seed_dev = d->seed_dev;
  if (g->dx) {          // dx = dP . [Wa ; Wb]  (K-dh without relu' mask and pooling term)
    BwdDhParams dp{};
    dp.g = gc; dp.Wa = d->Wa; dp.Wb = d->Wb; dp.du = g->dx; dp.N = d->N; dp.H = d->H; dp.scale_h = 1.0f;
    if (int e = launch_bwd_dh(dp, st)) return e;
  }
  TnParams tp{};
  tp.nprob = 1; tp.K = d->N; tp.splits = w.splits; tp.k_per_split = w.k_per_split; tp.g = gc; tp.tile = w.tile;
  TnProblem& q = tp.prob[0];    // d[Wa ; Wb] = dP^T . x ; (dba | dbb) = colsum(dP) ; dWc = colsum(gA . a_d . b_d)
  q.kind = TN_A_GATE; q.A = nullptr; q.lda = 0; q.M = w.mstk;
  q.B = x; q.ldb = d->H; q.Ncols = d->H;
  q.out = w.slab; q.split_stride = (size_t)w.mstk * d->H; q.ldc = d->H;
  q.colsum = w.cs_bab; q.colsum_stride = w.mstk; q.colsum2 = w.cs_wc; q.colsum2_stride = d->D;
  if (int e = launch_tn(tp, st)) return e;
  ReduceParams rp{};
  int n = 0;
  auto seg = [&](const float* in, float* out, int len, int nsplit, size_t stride) {
    rp.seg[n].in = in; rp.seg[n].out = out; rp.seg[n].len = len; rp.seg[n].nsplit = nsplit; rp.seg[n].stride = stride; ++n;
  };
  seg(w.slab, g->dWa, d->D * d->H, w.splits, (size_t)w.mstk * d->H);
  if (d->gated) seg(w.slab + (size_t)d->D * d->H, g->dWb, d->D * d->H, w.splits, (size_t)w.mstk * d->H);
  seg(w.cs_bab, g->dba, d->D, w.splits, w.mstk);
  if (d->gated) seg(w.cs_bab + d->D, g->dbb, d->D, w.splits, w.mstk);
  seg(w.cs_wc, g->dWc, d->D, w.splits, d->D);
  seg(gA, g->dbc, 1, (int)d->N, 1);         // d(bc) = sum_i dL/dA_i
  rp.nseg = n;
  return launch_reduce(rp, st);
}

size_t mmf_linear_forward_workspace_bytes(int64_t M, int32_t N, int32_t nseg, int32_t kseg) {
  if (nseg < 1 || nseg > 4 || kseg < 1) return 0;
  return linear_ksplit_floats(M, N, nseg * kseg, nseg, kseg) * sizeof(float);
}

int mmf_linear_forward(const float* const* x_segs, int32_t nseg, int32_t kseg, int64_t M,
                       const float* W, const float* bias, int32_t N, int32_t act,
                       float drop_p, uint32_t drop_seed, uint32_t drop_site, const uint32_t* seed_dev,
                       float* y, void* workspace, size_t workspace_bytes, uint32_t* sync, int32_t sync_words, void* stream) {
  if (!x_segs || nseg < 1 || nseg > 4 || !W || !y) return MMF_ERR_ARG;
  if (act < 0 || act > ACT_SELU || drop_p < 0.f || drop_p >= 1.f) return MMF_ERR_ARG;
  if (M * (int64_t)kseg * 4 >= (int64_t)1 << 31 || (int64_t)N * nseg * kseg * 4 >= (int64_t)1 << 31) return MMF_ERR_SHAPE;
  LinearParams lp{};
  for (int i = 0; i < nseg; ++i) {
    if (!x_segs[i]) return MMF_ERR_ARG;
    if (!aligned16(x_segs[i])) return MMF_ERR_ALIGN;
    lp.x[i] = x_segs[i];
  }
  if (!aligned16(W)) return MMF_ERR_ALIGN;
  lp.nseg = nseg; lp.kseg = kseg; lp.ldx = kseg;
  lp.w = W; lp.bias = bias; lp.y = y; lp.M = M; lp.N = N; lp.K = nseg * kseg;
  lp.act = act; lp.drop_p = drop_p; lp.drop_key = drop_key(drop_seed, drop_site); lp.seed_dev = seed_dev;
  if (workspace && sync && sync_words > 0 && aligned16(workspace) &&
      workspace_bytes >= linear_ksplit_floats(M, N, lp.K, nseg, kseg) * sizeof(float) && workspace_bytes > 0) {
    lp.kpart = static_cast<float*>(workspace); lp.ktick = sync; lp.ktick_words = sync_words;
  }
  return launch_linear(lp, static_cast<hipStream_t>(stream));
}

static int linear_bwd_splits(int64_t M, int N, int K) {
  const int td = tn_tile_dim(M, 0);
  const int tiles = ((N + td - 1) / td) * ((K + td - 1) / td);
  return tn_splits(M, tiles, td);
}

size_t mmf_linear_backward_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  const int s = linear_bwd_splits(M, N, K);
  if (s == 1) return 256;
  return align_up((size_t)s * N * K * 4, 256) + align_up((size_t)s * N * 4, 256);
}

int mmf_linear_backward(const float* dy, const float* const* x_segs, int32_t nseg, int32_t kseg, int64_t M,
                        const float* W, int32_t N, float* dW, float* db, float* dx,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (!dy || !x_segs || nseg < 1 || nseg > 4 || !dW) return MMF_ERR_ARG;
  if (dx && (nseg != 1 || !W)) return MMF_ERR_ARG;
  const int K = nseg * kseg;
  if (N % 4 != 0 || kseg % 4 != 0) return MMF_ERR_SHAPE;
  if (M * (int64_t)(N > kseg ? N : kseg) * 4 >= (int64_t)1 << 31 || (int64_t)N * K * 4 >= (int64_t)1 << 31) return MMF_ERR_SHAPE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int splits = linear_bwd_splits(M, N, K);
  if (splits > 1 && (!workspace || workspace_bytes < mmf_linear_backward_workspace_bytes(M, N, K))) return MMF_ERR_WORKSPACE;
  float* slab = splits > 1 ? static_cast<float*>(workspace) : dW;
  float* cs = splits > 1 ? reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)splits * N * K * 4, 256)) : db;

  TnParams tp{};
  tp.nprob = nseg; tp.K = M; tp.splits = splits; tp.tile = tn_tile_dim(M, 0);
  int64_t kps = (M + splits - 1) / splits;
  tp.k_per_split = (int)((kps + 3) / 4 * 4);
  for (int i = 0; i < nseg; ++i) {
    if (!x_segs[i]) return MMF_ERR_ARG;
    TnProblem& q = tp.prob[i];
    q.kind = TN_A_PLAIN; q.A = dy; q.lda = N; q.M = N;
    q.B = x_segs[i]; q.ldb = kseg; q.Ncols = kseg;
    q.out = slab + (size_t)i * kseg; q.split_stride = (size_t)N * K; q.ldc = K;
    q.colsum = (i == 0 && db) ? cs : nullptr; q.colsum_stride = N;
  }
  if (int e = launch_tn(tp, st)) return e;
  if (splits > 1) {
    ReduceParams rp{};
    rp.seg[0] = ReduceSeg{slab, dW, N * K, splits, (size_t)N * K, 0, 0};
    rp.nseg = 1;
    if (db) { rp.seg[1] = ReduceSeg{cs, db, N, splits, (size_t)N, 0, 0}; rp.nseg = 2; }
    if (int e = launch_reduce(rp, st)) return e;
  }
  if (dx) {
    if (N % KC != 0) return MMF_ERR_SHAPE;
    NnParams np{};
    np.A = dy; np.lda = N; np.B = W; np.ldb = K; np.C = dx; np.ldc = K; np.M = M; np.N = K; np.K = N;
    return launch_nn(np, st);
  }
  return MMF_OK;
}

int mmf_surv_head_forward(const float* feat, const float* Wk, const float* bk, int32_t B, int32_t F, int32_t K,
                          float* logits, float* hazards, float* S, int64_t* Y_hat, void* stream) {
  if (!feat || !Wk || !bk || !logits || !hazards || !S || !Y_hat || B < 1 || K < 1) return MMF_ERR_ARG;
  HeadParams p{feat, Wk, bk, B, F, K, logits, hazards, S, Y_hat};
  return launch_head_fwd(p, static_cast<hipStream_t>(stream));
}

int mmf_surv_head_backward(const float* g_hazards, const float* g_S, const float* hazards, const float* feat,
                           const float* Wk, int32_t B, int32_t F, int32_t K,
                           float* dfeat, float* dWk, float* dbk, void* stream) {
  if (!hazards || !feat || !Wk || !dfeat || !dWk || !dbk || B < 1 || K < 1) return MMF_ERR_ARG;
  HeadBwdParams p{g_hazards, g_S, hazards, feat, Wk, B, F, K, dfeat, dWk, dbk};
  return launch_head_bwd(p, static_cast<hipStream_t>(stream));
}

int mmf_nll_surv(const float* hazards, const float* S, const int64_t* Y, const float* c, int32_t B, int32_t K,
                 float alpha, float eps, float* loss, float* g_hazards, float* g_S, void* stream) {
  if (!hazards || !S || !Y || !c || !loss || !g_hazards || !g_S || B < 1 || K < 1) return MMF_ERR_ARG;
  NllParams p{hazards, S, Y, c, B, K, alpha, eps, loss, g_hazards, g_S};
  return launch_nll(p, static_cast<hipStream_t>(stream));
}

int mmf_cox_surv(const float* risks, const double* times, const float* c, int32_t B,
                 float* loss, float* d_risks, void* stream) {
  if (!risks || !times || !c || !loss || !d_risks || B < 1) return MMF_ERR_ARG;
  CoxParams p{risks, times, c, B, loss, d_risks};
  return launch_cox(p, static_cast<hipStream_t>(stream));
}

size_t mmf_maxnet_cox_step_workspace_bytes(int32_t B) {
  return maxnet_step_workspace_floats(B < 1 ? 1 : B) * sizeof(float);
}

int mmf_maxnet_cox_step(const mmf_maxnet_desc* d, const double* times, const float* c, float loss_scale,
                        void* workspace, size_t workspace_bytes, float* risk, float* loss,
                        const mmf_maxnet_grads* g, int32_t accumulate, void* stream) {
  if (!d || !times || !c || !workspace || !risk || !loss || !g) return MMF_ERR_ARG;
  if (!d->x || !d->W0 || !d->b0 || !d->W1 || !d->b1 || !d->Wc || !d->bc) return MMF_ERR_ARG;
  if (!g->dW0 || !g->db0 || !g->dW1 || !g->db1 || !g->dWc || !g->dbc) return MMF_ERR_ARG;
  if (!maxnet_step_ok(d->B, d->G, d->H0, d->H1)) return MMF_ERR_SHAPE;
  if (!d->sync || d->sync_words < 3) return MMF_ERR_ARG;          // the two grid barriers live in the caller's tick words
  if (d->p_drop < 0.f || d->p_drop >= 1.f) return MMF_ERR_ARG;
  if (workspace_bytes < mmf_maxnet_cox_step_workspace_bytes(d->B)) return MMF_ERR_WORKSPACE;
  MaxnetStepParams p{};
  p.B = d->B; p.G = d->G;
  p.x = d->x; p.W0 = d->W0; p.b0 = d->b0; p.W1 = d->W1; p.b1 = d->b1; p.Wc = d->Wc; p.bc = d->bc;
  p.times = times; p.c = c;
  p.p = d->p_drop; p.key0 = drop_key(d->seed, 0); p.key1 = drop_key(d->seed, 1); p.seed_dev = d->seed_dev;
  p.loss_scale = loss_scale;
  float* w = static_cast<float*>(workspace);
  const size_t n = (size_t)d->B * 256;
  const size_t nt = (size_t)256 * maxnet_step_dp_pitch(d->B);
  p.y0 = w; p.y1 = w + n; p.dp1 = w + 2 * n; p.dp0 = p.dp1 + nt; p.dr = p.dp0 + nt;
  float* after = p.dr + (size_t)((d->B + 63) / 64 * 64);
  p.stamps = reinterpret_cast<unsigned long long*>(after);      // read by tools/stamps_maxnet.py only
  p.dwc_part = after + 32;
  p.bar = d->sync;
  p.risk = risk; p.loss = loss;
  p.dW0 = g->dW0; p.db0 = g->db0; p.dW1 = g->dW1; p.db1 = g->db1; p.dWc = g->dWc; p.dbc = g->dbc;
  p.accumulate = accumulate ? 1 : 0;
  TraceScope ts(d->trace);
  return launch_maxnet_cox_step(p, static_cast<hipStream_t>(stream));
}


int mmf_adam_l1_step(float* w, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, float l1_coeff, const float* l1_mask, int32_t step, void* stream) {
  if (!w || !g || !m || !v || n < 1 || step < 1) return MMF_ERR_ARG;
  if (l1_mask && !aligned16(l1_mask)) return MMF_ERR_ALIGN;
  if (!aligned16(w) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return MMF_ERR_ALIGN;
  AdamParams p{};
  p.w = w; p.g = g; p.m = m; p.v = v; p.n = n; p.l1_mask = l1_mask;
  p.b1 = beta1; p.b2 = beta2; p.eps = eps; p.wd = weight_decay; p.l1 = l1_coeff;
  // bias corrections in double, as torch computes them on the host
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  p.step_size = (float)((double)lr / bc1);
  p.bc2_sqrt = (float)std::sqrt(bc2);
  return launch_adam_l1(p, static_cast<hipStream_t>(stream));
}

int mmf_abs_sum(const float* w, int64_t n, float* partials, float* out, void* stream) {
  if (!w || !partials || !out || n < 1) return MMF_ERR_ARG;
  return launch_abs_sum(w, n, partials, out, static_cast<hipStream_t>(stream));
}

static DropSpec make_drop(int kind, float p, uint32_t seed, uint32_t site, const uint32_t* seed_dev) {
  DropSpec d;
  d.kind = p > 0.f ? kind : 0;
  d.p = p;
  d.key = drop_key(seed, site);
  d.dev = seed_dev;
  return d;
}

int mmf_dense_forward(const float* x, const float* W, const float* bias, int32_t B, int32_t K, int32_t N,
                      int32_t act, int32_t drop_kind, float drop_p, uint32_t seed, uint32_t site,
                      const uint32_t* seed_dev, float* y, void* stream) {
  if (!x || !W || !y || B < 1 || K < 1 || N < 1) return MMF_ERR_ARG;
  if (act < 0 || act > ACT_SELU || drop_kind < 0 || drop_kind > 2 || drop_p < 0.f || drop_p >= 1.f) return MMF_ERR_ARG;
  DenseParams p{x, W, bias, y, B, K, N, act, make_drop(drop_kind, drop_p, seed, site, seed_dev)};
  return launch_dense_fwd(p, static_cast<hipStream_t>(stream));
}

int mmf_dense_backward(const float* dy, const float* y, const float* x, const float* W,
                       int32_t B, int32_t K, int32_t N, int32_t act,
                       int32_t drop_kind, float drop_p, uint32_t seed, uint32_t site, const uint32_t* seed_dev,
                       float* dpre_scratch, float* dx, float* dW, float* db, void* stream) {
  if (!dy || !y || !x || !W || !dpre_scratch || B < 1 || K < 1 || N < 1) return MMF_ERR_ARG;
  if (act < 0 || act > ACT_SELU || drop_kind < 0 || drop_kind > 2) return MMF_ERR_ARG;
  DenseBwdParams p{dy, y, x, W, dpre_scratch, dx, dW, db, B, K, N, act, make_drop(drop_kind, drop_p, seed, site, seed_dev)};
  return launch_dense_bwd(p, static_cast<hipStream_t>(stream));
}

int mmf_gate_mul_forward(const float* z, const float* h, float* o, int32_t n, void* stream) {
  if (!z || !h || !o || n < 1) return MMF_ERR_ARG;
  return launch_gate_mul(z, h, o, n, static_cast<hipStream_t>(stream));
}
int mmf_gate_mul_backward(const float* g, const float* z, const float* h, float* dz, float* dh, int32_t n, void* stream) {
  if (!g || !z || !h || !dz || !dh || n < 1) return MMF_ERR_ARG;
  return launch_gate_mul_bwd(g, z, h, dz, dh, n, static_cast<hipStream_t>(stream));
}

int mmf_kron_forward(const float* const* o, int32_t m, int32_t dim, int32_t B,
                     float drop_p, uint32_t seed, uint32_t site, const uint32_t* seed_dev, float* out, void* stream) {
  if (!o || (m != 2 && m != 3) || dim < 1 || B < 1 || !out) return MMF_ERR_ARG;
  KronParams p{};
  for (int i = 0; i < m; ++i) { if (!o[i]) return MMF_ERR_ARG; p.o[i] = o[i]; }
  p.out = out; p.m = m; p.dim = dim; p.B = B; p.drop = make_drop(1, drop_p, seed, site, seed_dev);
  return launch_kron_fwd(p, static_cast<hipStream_t>(stream));
}
int mmf_kron_backward(const float* g, const float* const* o, int32_t m, int32_t dim, int32_t B,
                      float drop_p, uint32_t seed, uint32_t site, const uint32_t* seed_dev, float* const* d_o,
                      void* stream) {
  if (!g || !o || !d_o || (m != 2 && m != 3) || dim < 1 || B < 1) return MMF_ERR_ARG;
  KronParams p{};
  for (int i = 0; i < m; ++i) { if (!o[i] || !d_o[i]) return MMF_ERR_ARG; p.o[i] = o[i]; p.d[i] = d_o[i]; }
  p.g = g; p.m = m; p.dim = dim; p.B = B; p.drop = make_drop(1, drop_p, seed, site, seed_dev);
  return launch_kron_bwd(p, static_cast<hipStream_t>(stream));
}

/* diagnostic builds only (-DMMF_STAMPS): which = 0 forward TU, 1 backward TU; out8 = {load, mfma, store, barrier cycles, chunks};
 * which = 2: bf16 TU, writes 32 values (4 kernels x {prologue, main loop, epilogue, -, -, -, -, waves}) */
#ifdef MMF_STAMPS       /* exported by the diagnostic libraries only (tools/diag_build.py -> multimodalfusion_amd/_diag/) */
void mmf_debug_stamps(int which, unsigned long long* out8) {
  if (which == 0) debug_stamps_fwd(out8);
  else if (which == 1) debug_stamps_bwd(out8);
  else debug_stamps_bf16(out8);
}
#endif

static int xreduce_params(const mmf_xreduce_io* io, float drop_p, uint32_t seed, const uint32_t* seed_dev, bool bwd,
                          XReduceParams& p) {
  if (!io || io->m < 1 || io->m > 3 || io->B < 1 || io->dim < 1 || io->sdim < 1) return MMF_ERR_ARG;
  if (drop_p < 0.f || drop_p >= 1.f) return MMF_ERR_ARG;
  p = XReduceParams{};
  p.m = io->m; p.B = io->B; p.dim = io->dim; p.sdim = io->sdim;
  for (int i = 0; i < io->m; ++i) {
    if (!io->v[i] || !io->Wh[i] || !io->bh[i] || !io->Wz[i] || !io->bz[i] || !io->Wo[i] || !io->bo[i]) return MMF_ERR_ARG;
    if (!io->h[i] || !io->z[i] || !io->gm[i] || !io->o[i]) return MMF_ERR_ARG;
    p.v[i] = io->v[i]; p.Wh[i] = io->Wh[i]; p.bh[i] = io->bh[i]; p.Wz[i] = io->Wz[i]; p.bz[i] = io->bz[i];
    p.Wo[i] = io->Wo[i]; p.bo[i] = io->bo[i];
    p.h[i] = io->h[i]; p.z[i] = io->z[i]; p.gm[i] = io->gm[i]; p.o[i] = io->o[i];
    if (bwd) {
      if (!io->d_o[i] || !io->dv[i] || !io->dWh[i] || !io->dbh[i] || !io->dWz[i] || !io->dbz[i] || !io->dWo[i] || !io->dbo[i])
        return MMF_ERR_ARG;
      p.d_o[i] = io->d_o[i]; p.dv[i] = io->dv[i];
      p.dWh[i] = io->dWh[i]; p.dbh[i] = io->dbh[i]; p.dWz[i] = io->dWz[i]; p.dbz[i] = io->dbz[i];
      p.dWo[i] = io->dWo[i]; p.dbo[i] = io->dbo[i];
    }
  }
  p.drop = make_drop(1, drop_p, seed, 0, seed_dev);
  return MMF_OK;
}
int mmf_xreduce_forward(const mmf_xreduce_io* io, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream) {
  XReduceParams p;
  if (int e = xreduce_params(io, drop_p, seed, seed_dev, false, p)) return e;
  return launch_xreduce_fwd(p, static_cast<hipStream_t>(stream));
}
int mmf_xreduce_backward(const mmf_xreduce_io* io, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream) {
  XReduceParams p;
  if (int e = xreduce_params(io, drop_p, seed, seed_dev, true, p)) return e;
  return launch_xreduce_bwd(p, static_cast<hipStream_t>(stream));
}

int mmf_batchnorm_forward(const float* x, const float* res, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int32_t B, int32_t F, int32_t training,
                          float eps, float momentum, int32_t act, float drop_p, uint32_t seed, uint32_t site,
                          const uint32_t* seed_dev, float* y, float* save_mean, float* save_invstd, void* stream) {
  if (!x || !y || !save_mean || !save_invstd || B < 1 || F < 1) return MMF_ERR_ARG;
  if (!training && (!running_mean || !running_var)) return MMF_ERR_ARG;
  if (act < 0 || act > ACT_SELU || drop_p < 0.f || drop_p >= 1.f) return MMF_ERR_ARG;
  BnParams p{x, res, gamma, beta, running_mean, running_var, y, save_mean, save_invstd, B, F, training, act, eps, momentum,
             make_drop(1, drop_p, seed, site, seed_dev)};
  return launch_bn_fwd(p, static_cast<hipStream_t>(stream));
}
int mmf_batchnorm_backward(const float* dy, const float* y, const float* x, const float* gamma,
                           const float* save_mean, const float* save_invstd, int32_t B, int32_t F, int32_t training,
                           int32_t act, float drop_p, uint32_t seed, uint32_t site, const uint32_t* seed_dev,
                           float* dx, float* dres, float* dgamma, float* dbeta, void* stream) {
  if (!dy || !y || !x || !save_mean || !save_invstd || !dx || B < 1 || F < 1) return MMF_ERR_ARG;
  if (act < 0 || act > ACT_SELU || drop_p < 0.f || drop_p >= 1.f) return MMF_ERR_ARG;
  BnBwdParams p{dy, y, x, gamma, save_mean, save_invstd, dx, dres, dgamma, dbeta, B, F, training, act,
                make_drop(1, drop_p, seed, site, seed_dev)};
  return launch_bn_bwd(p, static_cast<hipStream_t>(stream));
}
int mmf_highway_mix_forward(const float* zg, const float* zn, const float* zl, int64_t n, float* y, void* stream) {
  if (!zg || !zn || !zl || !y || n < 1) return MMF_ERR_ARG;
  HighwayParams p{zg, zn, zl, y, nullptr, nullptr, nullptr, nullptr, n};
  return launch_highway_fwd(p, static_cast<hipStream_t>(stream));
}
int mmf_highway_mix_backward(const float* dy, const float* zg, const float* zn, const float* zl, int64_t n,
                             float* dzg, float* dzn, float* dzl, void* stream) {
  if (!dy || !zg || !zn || !zl || !dzg || !dzn || !dzl || n < 1) return MMF_ERR_ARG;
  HighwayParams p{zg, zn, zl, nullptr, dy, dzg, dzn, dzl, n};
  return launch_highway_bwd(p, static_cast<hipStream_t>(stream));
}
int mmf_ranking_loss(const float* risks, const double* times, const float* c, int32_t B, int32_t phi, int32_t reduction,
                     float* loss, float* d_risks, void* stream) {
  if (!risks || !times || !c || !loss || !d_risks) return MMF_ERR_ARG;
  if (phi < 0 || phi > 1 || reduction < 0 || reduction > 1) return MMF_ERR_ARG;
  RankParams p{risks, times, c, B, phi, reduction, loss, d_risks};
  return launch_rank_loss(p, static_cast<hipStream_t>(stream));
}
int mmf_hazards_forward(const float* logits, int32_t B, int32_t K, float* hazards, float* S, int64_t* Y_hat, float* risk,
                        void* stream) {
  if (!logits || !hazards || !S) return MMF_ERR_ARG;
  HazardParams p{logits, hazards, S, risk, Y_hat, nullptr, nullptr, nullptr, nullptr, B, K};
  return launch_hazard_fwd(p, static_cast<hipStream_t>(stream));
}
int mmf_hazards_backward(const float* g_hazards, const float* g_S, const float* g_risk, const float* hazards,
                         int32_t B, int32_t K, float* dlogits, void* stream) {
  if (!hazards || !dlogits) return MMF_ERR_ARG;
  HazardParams p{nullptr, const_cast<float*>(hazards), nullptr, nullptr, nullptr, g_hazards, g_S, g_risk, dlogits, B, K};
  return launch_hazard_bwd(p, static_cast<hipStream_t>(stream));
}

mmf_trace* mmf_trace_create(int32_t capacity) {
  if (capacity < 1) return nullptr;
  mmf_trace* t = new (std::nothrow) mmf_trace();
  if (t) { t->cap = capacity; t->recs.reserve(capacity); }
  return t;
}

void mmf_trace_destroy(mmf_trace* t) {
  if (!t) return;
  for (auto& r : t->recs) { if (r.a) hipEventDestroy(r.a); if (r.b) hipEventDestroy(r.b); }
  for (auto e : t->pool) hipEventDestroy(e);
  delete t;
}

// Synchronises on the recorded events, writes "name count total_ms\n" lines, clears the records.
int mmf_trace_dump(mmf_trace* t, char* buf, size_t buf_bytes) {
  if (!t) return MMF_ERR_ARG;
  std::lock_guard<std::mutex> lock(t->mu);
  std::map<std::string, std::pair<int, double>> agg;
  for (auto& r : t->recs) {
    float ms = 0.f;
    if (r.a && r.b && hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      auto& e = agg[r.name];
      e.first += 1;
      e.second += ms;
    }
    if (r.a) t->pool.push_back(r.a);
    if (r.b) t->pool.push_back(r.b);
  }
  t->recs.clear();
  std::string out;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof line, "%s %d %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  if (!buf || buf_bytes == 0) return (int)out.size();
  size_t n = out.size() < buf_bytes - 1 ? out.size() : buf_bytes - 1;
  memcpy(buf, out.data(), n);
  buf[n] = 0;
  return (int)n;
}

int mmf_dropout_keep_host(uint32_t seed, uint32_t site, uint32_t index, float p) {
  return keep(drop_key(seed, site), index, drop_threshold(p)) ? 1 : 0;
}

}  // extern "C"
