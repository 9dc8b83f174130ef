// K-fwd, second form (bf16 storage, H = D = 256, gated): instance projection + gated attention scoring + pooling
// partials of one 128-instance tile, built so that TWO workgroups share a CU (4 waves, <= 256 VGPRs, < 80 KB of LDS
// each).  The first form (mmf_amil_bf16.hip) runs one 8-wave workgroup per CU and its phases -- operand stream,
// MFMA, VALU-bound epilogues, gate phase, pooling -- strictly one after the other: the CU's load path idles for two
// thirds of a tile's life (profiles/r02: 0.26 of the HBM rate).  With two independent workgroups per CU the
// hardware overlaps one tile's epilogues / gate phase with the other's operand stream and MFMAs.
//
// What makes the halved budget fit:
//   * TRANSPOSED products.  Both GEMMs are computed as (weights) x (instances)^T: the MFMA's A operand is a block of
//     32 weight rows, its B operand 32 instances.  In the 32x32 accumulator a lane then holds ONE instance and groups
//     of 4 consecutive output features -- the contiguous direction of every output ([instance][feature] rows of h,
//     a, b) -- so the epilogues pack 4 values to 8 bytes and write them where they belong: no fp32 transpose through
//     LDS (37 KB of scratch and ~1/3 of the old epilogues' instructions), and the biases sit in the accumulators'
//     initial value instead of the epilogue.
//   * WEIGHTS NEVER TOUCH LDS.  The per-call bf16 copies of W1 and [Wa ; Wb] are written in MFMA-fragment order
//     (cvt_bf16_kernel, CvtSeg::transpose 3 / 4): one buffer_load_dwordx4 per lane IS an A fragment, a wave
//     instruction reads 1 KB of contiguous memory, and each of the 4 waves loads only the rows it multiplies
//     (wave w owns hidden features 64 w .. 64 w + 63; in the gate phase attention dims 64 p + 16 w .. + 15 of pass
//     p = 0..3).  LDS holds only the x chunks (three 16 KB stages, LDS-DMA) and then the tile's h image (66 KB,
//     aliasing the stages).
//   * the gate weights of a pass (32 rows = 16 tanh + 16 sigmoid rows of the same dims, K = 256: 64 VGPRs) sit in one of
//     two register sets; the other set is filled for the next pass, four k-steps per block.
//
// LDS map (bytes): [0, 49152) three x stages of [128 rows][128 B] (XOR-swizzled as mmf_gemm_dma.h), later
//                  [0, 67584) h image [128 rows][528 B] (512 B of features + 16 B of padding);
//                  [67584, ...) score partials [4 waves][128], e[128], scratch[16], pooling partials [8][256], ba / bb / Wc [3][256].
// Reference lines: models/model_attention_mil_path.py:52-56 (projection, attention net, softmax pooling),
// models/model_modules.py:105-110 (Attn_Net_Gated.forward).
#include <type_traits>
#include <cstdlib>

#include "mmf_gemm_dma.h"
#include "mmf_bf16.h"

namespace mmf {

constexpr int F2_BM = 128;
constexpr int F2_STAGE = F2_BM * 128;                 // one x chunk: 128 rows x 64 bf16
constexpr int F2_HROW = 528;                          // h image row pitch: 512 B + 16 (pitch / 4 = 4 mod 64 banks: the 16 lanes of a
                                                      // ds_read_b128 group, one row each, cover all 64 banks with NO address swizzle,
                                                      // so a k-step is an immediate offset of the read, not an instruction)
constexpr int F2_HIMG = 0, F2_MISC = F2_BM * F2_HROW;
constexpr int F2_LDS_BYTES = F2_MISC + (4 * 128 + 128 + 16 + 8 * 256 + 3 * 256) * 4;

typedef float f32x4v __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DMMF_STAMPS): phase cycles {main loop, epilogue 1, gate phase, pooling, -, -, real time, waves}
#ifdef MMF_STAMPS
static __device__ unsigned long long g_bst2[8];
__device__ inline unsigned long long real_now2() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define B2_BEGIN() unsigned long long b2_real = real_now2(), b2_prev = stamp_now(), b2_t
#define B2_MARK(slot) do { b2_t = stamp_now(); if ((threadIdx.x & 63) == 0) atomicAdd(&g_bst2[slot], b2_t - b2_prev); b2_prev = b2_t; } while (0)
#define B2_COUNT() do { if ((threadIdx.x & 63) == 0) { atomicAdd(&g_bst2[7], 1ull); atomicAdd(&g_bst2[6], real_now2() - b2_real); } } while (0)
#else
#define B2_BEGIN()
#define B2_MARK(slot)
#define B2_COUNT()
#endif
void debug_stamps_fwd2(unsigned long long* out8) {       // overwrites out8 when this kernel ran since the last call
#ifdef MMF_STAMPS
  unsigned long long v[8];
  hipMemcpyFromSymbol(v, HIP_SYMBOL(g_bst2), sizeof v);
  if (v[7] == 0) return;
  for (int i = 0; i < 8; ++i) out8[i] = v[i];
  unsigned long long z[8] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_bst2), z, sizeof z);
#else
  (void)out8;
#endif
}

// A-fragment load hidden from the compiler's wait bookkeeping (the x stream's LDS-DMAs share the in-order VM
// counter; completion is counted by hand in the main loop, cdna_hip_programming.md "mixing load KINDS in one k-loop")
template <int IMM>
__device__ inline void ldw_asm(f32x4v& dst, const DmaRsrc& rs, unsigned voff, unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=&v"(dst) : "v"(voff), "s"(rs.w), "s"(soff), "n"(IMM) : "memory");
}
__device__ inline void bst4(rsrc_t r, unsigned voff, const float4& v) {
  u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)voff, 0, 0);
}
__device__ inline bf16x8 frag_of(const f32x4v& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ inline bf16x8 frag_of(const float4& v) {
  f32x4v t = {v.x, v.y, v.z, v.w};
  return __builtin_bit_cast(bf16x8, t);
}

// MFMAs of one staged x chunk (64 k) against the wave's 64 weight rows held in `wf` (k-step q, row block fb -> wf[2q + fb])
template <class Hook>
__device__ inline void f2_chunk(const char* xs, const f32x4v (&wf)[8], f32x16 (&acc)[2][4], int r, int hh, Hook&& hook) {
  const int sw = (r >> 1) & 7;
  const char* b0 = xs + r * 128;
  float4 fx[2][4];
  auto rd = [&](int q, int buf) {
    const int o = 16 * ((2 * q + hh) ^ sw);
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) fx[buf][ib] = *reinterpret_cast<const float4*>(b0 + ib * 32 * 128 + o);
  };
  rd(0, 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q + 1 < 4) rd(q + 1, (q + 1) & 1);
    hook(q);                                               // vector-ALU work that rides in the MFMAs' shadow
#ifdef MMF_F2_REV_SCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
      for (int ib = 0; ib < 4; ++ib)
        acc[fb][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(wf[2 * q + fb]), frag_of(fx[q & 1][ib]), acc[fb][ib], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool HLOOP>      // projection dropout on and L == 1024: the keep-bits are hashed inside the main loop
__global__ __launch_bounds__(256, 2) void amil_fwd_fused2_bf16_kernel(FusedFwdParams p) {
  extern __shared__ __align__(16) char lds2[];
  char* lds = lds2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int mt = blockIdx.x, row0 = mt * F2_BM;
  const uint32_t sdev = p.seed_dev ? *p.seed_dev : 0u;
  float* sred = reinterpret_cast<float*>(lds + F2_MISC);         // [4 waves][128 rows]
  float* e_l = sred + 4 * 128;                                    // [128]
  float* red = e_l + 128;                                         // [16]
  float* vred = red + 16;                                         // [8][256]
  float* gpar = vred + 8 * 256;                                   // ba, bb, Wc: read per pass from LDS, not from memory (a load
                                                                  // issued behind a pass's a / b stores would wait for their acknowledgement)
  {
    const float va = p.ba[tid], vb = p.bb[tid], vc = p.Wc[tid];
    asm volatile("" :: "v"(va), "v"(vb), "v"(vc));
    gpar[tid] = va; gpar[256 + tid] = vb; gpar[512 + tid] = vc;
  }

#ifdef MMF_F2_DEBUG      /* timing experiments (wrong results): phases switched off by bits of MMF_F2_DEBUG_MASK */
  const int dbg = p.stagger;
#else
  constexpr int dbg = 0;
#endif
  B2_BEGIN();
  // ---------------- phase 1: u^T = W1 . x^T (K = L) ------------------------------------------------------------
  f32x16 acc[2][4];                                               // [feature block fb][instance block ib]
#pragma unroll
  for (int fb = 0; fb < 2; ++fb) {
    float bias[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t = ld4(p.b1 + 64 * wave + 32 * fb + 8 * g + 4 * hh);
      bias[4 * g] = t.x; bias[4 * g + 1] = t.y; bias[4 * g + 2] = t.z; bias[4 * g + 3] = t.w;
    }
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[fb][ib][i] = bias[i];
    asm volatile("" :: "v"(bias[0]), "v"(bias[4]), "v"(bias[8]), "v"(bias[12]));   // the compiler's own waits for these loads stay in front of the hidden queue
  }
  const uint32_t thr_h = drop_threshold(p.p_h);
  constexpr bool hloop = HLOOP;
  const uint32_t hbase = ((uint32_t)(row0 + r) * 256u + 64u * (uint32_t)wave + 4u * (uint32_t)hh) * 0x9E3779B1u + p.key_h + sdev;
  uint32_t km[4] = {0u, 0u, 0u, 0u};                              // bit e = ((fb 4 + ib) 4 + g) 4 + j of word e / 32: keep
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the hand-counted queue below starts empty
  __builtin_amdgcn_sched_barrier(0);
  {
    DmaK<F2_BM, 256> lx;
    lx.init(p.x, p.L, row0, (int)p.N);
    DmaRsrc rw;
    rw.set(p.w1f, 256u * (unsigned)p.L * 2u);
    const unsigned vw0 = (unsigned)lane * 16u + (unsigned)wave * 8192u, vw1 = vw0 + 4096u;
    f32x4v wf0[8], wf1[8];
    auto load_w = [&](int kt, f32x4v (&w)[8]) {
      const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)kt * 32768u);
      ldw_asm<0>(w[0], rw, vw0, so); ldw_asm<1024>(w[1], rw, vw0, so); ldw_asm<2048>(w[2], rw, vw0, so); ldw_asm<3072>(w[3], rw, vw0, so);
      ldw_asm<0>(w[4], rw, vw1, so); ldw_asm<1024>(w[5], rw, vw1, so); ldw_asm<2048>(w[6], rw, vw1, so); ldw_asm<3072>(w[7], rw, vw1, so);
    };
#ifndef MMF_F2_STAGE_BASE
#define MMF_F2_STAGE_BASE 0
#endif
    auto stage = [&](int kt) { return lds + MMF_F2_STAGE_BASE + (kt % 3) * F2_STAGE; };
    const int nk = p.L / 64;                                      // even (launcher)
    // Queue order per iteration: [W(kt+1) x 8] [x(kt+2) x 4]; before chunk kt+1 is read everything up to W(kt+1) has landed
    // (vmcnt(4)), x(kt+2) may still be in flight.  The loop body is BRANCH-FREE on purpose: the weight fragments are outputs
    // of asm statements whose data arrives later, and a conditional definition would let the compiler merge old and new
    // values with register copies placed right behind the (still unanswered) load.  The requests past the last chunk read
    // beyond num_records (weights: zeros, no traffic) or are turned into out-of-range requests (x: `kill`, or they would fetch
    // the first bytes of the next rows, 12 % more x traffic); all are drained behind the loop.
    // The streaming phase outranks the partner workgroup's vector-bound phases on the SIMD (issue is arbitrated by priority,
    // then age): its loads and MFMAs are what the memory pipe waits for.  Measured -4 us of 121; raising the vector-bound
    // phases instead changes nothing.
#ifndef MMF_F2_PRIO
#define MMF_F2_PRIO 1
#endif
    __builtin_amdgcn_s_setprio(MMF_F2_PRIO);
    load_w(0, wf0);
    lx.issue(0, stage(0));
    lx.issue(1, stage(1));
    wait_vmcnt<4>();
    asm volatile("" : "+v"(wf0[0]), "+v"(wf0[1]), "+v"(wf0[2]), "+v"(wf0[3]), "+v"(wf0[4]), "+v"(wf0[5]), "+v"(wf0[6]), "+v"(wf0[7]));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // The dropout keep-bits of the projection (model_attention_mil_path.py:21) depend on indices only: with 16 chunks they
    // are hashed here, 8 elements per chunk between the MFMAs (the loop waits for memory, its vector ALU is idle), and the
    // epilogue only applies them.  Element e = ((fb 4 + ib) 4 + g) 4 + j of this lane is hashed in chunk e / 8.
    {
      auto iter = [&](int kt, f32x4v (&wcur)[8], f32x4v (&wnext)[8]) {
#ifdef MMF_F2_COND_LOAD     /* diagnostic build (tools/diag_build.py f2cond*): the weight refill as a CONDITIONAL definition of the
                               asm-loaded registers, as an intermediate round-3 version had it -- what the branch-free loop replaced */
        if (kt + 1 < nk) load_w(kt + 1, wnext);
#else
        if (!(dbg & 2)) load_w(kt + 1, wnext);
#endif
        if (!(dbg & 1)) lx.issue(kt + 2, stage(kt + 2), kt + 2 < nk ? 0u : 0x80000000u);
        uint32_t bits = 0;
        const uint32_t sC = __builtin_amdgcn_readfirstlane((uint32_t)(((kt >> 1) & 3) * 8192 + (kt >> 3) * 32 + (kt & 1) * 16) * 0x9E3779B1u);
        __builtin_amdgcn_sched_barrier(0);
        if (!(dbg & 4)) f2_chunk(stage(kt), wcur, acc, r, hh, [&](int q) {
          if constexpr (HLOOP) {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
              const int i = 2 * q + i2;
              uint32_t h = hbase + sC + (uint32_t)((i >> 2) * 8 + (i & 3)) * 0x9E3779B1u;
              h ^= h >> 16; h *= 0x85EBCA6Bu;
              h ^= h >> 13; h *= 0xC2B2AE35u;
              h ^= h >> 16;
              bits |= ((h >> 8) >= thr_h ? 1u : 0u) << i;
            }
          }
        });
        if constexpr (HLOOP) {
          const uint32_t b = bits << ((kt & 3) * 8);
          const int w = kt >> 2;
          km[0] |= w == 0 ? b : 0u; km[1] |= w == 1 ? b : 0u; km[2] |= w == 2 ? b : 0u; km[3] |= w == 3 ? b : 0u;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (dbg & 3) wait_vmcnt<0>(); else wait_vmcnt<4>();
        // the fragments are defined HERE as far as the compiler is concerned: nothing may read or move them before the wait
        asm volatile("" : "+v"(wnext[0]), "+v"(wnext[1]), "+v"(wnext[2]), "+v"(wnext[3]), "+v"(wnext[4]), "+v"(wnext[5]), "+v"(wnext[6]), "+v"(wnext[7]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int kt = 0; kt < nk; kt += 2) {
        iter(kt, wf0, wf1);
        iter(kt + 1, wf1, wf0);
      }
    }
    wait_vmcnt<0>();                                             // the two refills past the last chunk
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }

  __builtin_amdgcn_s_setprio(0);
  B2_MARK(0);
  // Everything below derives its lane-dependent addresses from a thread index the compiler cannot see through: left alone it
  // computes the epilogues' LDS / store offsets at the top of the kernel and parks them in scratch across the main loop.
  int tid_late = threadIdx.x;
  asm volatile("" : "+v"(tid_late));
  {
  const int tid = tid_late, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  // ---------------- epilogue 1: h = bf16(drop(relu(u))) -> the LDS h image (B operand of phase 2, pooled operand) ----
  {
    const float scale = p.p_h > 0.f ? 1.0f / (1.0f - p.p_h) : 1.0f;
    const uint32_t dkey = p.key_h + sdev;
    auto epi = [&](auto mode_c) {                            // 0: no dropout, 1: keep-bits from the main loop, 2: hashed here
      constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
          const int R = 32 * ib + r;
          char* rowp = lds + F2_HIMG + R * F2_HROW + 8 * hh;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int f = 64 * wave + 32 * fb + 8 * g + 4 * hh;
            float y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              y[j] = fmaxf(acc[fb][ib][4 * g + j], 0.f);
              if constexpr (MODE == 1) y[j] = (km[2 * fb + (ib >> 1)] >> ((ib & 1) * 16 + 4 * g + j)) & 1u ? y[j] * scale : 0.f;
              if constexpr (MODE == 2) y[j] = keep(dkey, (uint32_t)(row0 + R) * 256u + (uint32_t)(f + j), thr_h) ? y[j] * scale : 0.f;
            }
            const int slot = 8 * wave + 4 * fb + g;
            *reinterpret_cast<uint2*>(rowp + 16 * slot) = pack4(y[0], y[1], y[2], y[3]);
          }
        }
    };
    if (p.p_h <= 0.f) epi(std::integral_constant<int, 0>{});
    else if (hloop) epi(std::integral_constant<int, 1>{});
    else epi(std::integral_constant<int, 2>{});
  }
  // the wave's gate weights of pass 0 (16 k-steps x one 32-row A block): in flight across the barrier and the h copy
  const rsrc_t rg = make_rsrc(p.wabf, 512u * 256u * 2u);
  const unsigned vg = (unsigned)lane * 16u + (unsigned)wave * 16384u;
  float4 wg0[16], wg1[16];                                 // weights of the even / odd passes
#pragma unroll
  for (int s = 0; s < 16; ++s) wg0[s] = bld4(rg, vg, (unsigned)s * 1024u);
  __syncthreads();                                         // h tile complete
  B2_MARK(1);
  // saved activations go out through buffer stores: rows beyond the bag fall outside num_records and are dropped
  // (forward-only calls: zero-size resources drop every store, so that no phase below needs a branch)
  const unsigned act_bytes = (unsigned)p.N * 512u;
  const bool keep_h = p.h && !(dbg & 128);
  const rsrc_t rh = make_rsrc(keep_h ? (const void*)p.h : (const void*)p.x, keep_h ? act_bytes : 0u);
  const rsrc_t rsa = make_rsrc(p.a ? (const void*)p.a : (const void*)p.x, p.a ? act_bytes : 0u),
               rsb = make_rsrc(p.a ? (const void*)p.b : (const void*)p.x, p.a ? act_bytes : 0u);

  // ---------------- phase 2: [a ; b]^T = (gate weights) . h^T as a software pipeline -------------------------------------
  // 16 blocks (pass ps = 0..3: attention dims 64 ps + 16 w .. + 15; instance block ib = 0..3).  A block is ONE 32 x 32
  // accumulator: 16 MFMAs along K = 256 (rows 0-15 of the A block are Wa, rows 16-31 Wb of the same dims: accumulator
  // i < 8 is pre-tanh, i >= 8 pre-sigmoid of dim dbase + 8 ((i >> 2) & 1) + 4 hh + (i & 3)), then its activations:
  // 32 quarter-rate instructions (exp2, rcp) and ~120 others -- tanh / sigmoid, bf16 rounding, a / b stores (16
  // contiguous bytes per lane after the half swap), score partial: about twice the block's MFMA time.  A SIMD does not run
  // one wave's vector work beside its partner's dense MFMA stream (tools/coissue.hip), but it does issue a wave's OWN
  // vector instructions in the shadow of that wave's MFMAs.  So block k's 16 (dependent) MFMAs are issued one by one
  // with ONE SLICE of block k - 1's activations behind each -- one activation of one accumulator element per slice, the
  // slices pinned in place (sched_barrier): left to itself the scheduler bunched the vector work into runs of 150-300
  // instructions with the matrix pipe idle.  The accumulators alternate; the weights of the next pass (second register
  // set) arrive four k-steps per block.  Block 0 has no predecessor: the h copy-out rides behind its MFMAs instead, and
  // a 17th, empty block (its weights read as zero beyond the buffer) carries block 15's activations.
  const uint32_t thr_a = drop_threshold(p.p_att);
  const bool drop = p.p_att > 0.f;
  const float dscale = drop ? 1.0f / (1.0f - p.p_att) : 1.0f;
  const uint32_t key_a = p.key_a + sdev, key_b = p.key_b + sdev;
  float sc[4] = {0.f, 0.f, 0.f, 0.f};                      // score partials of instance 32 ib + r over this lane's dims
  auto gate_phase = [&](auto drop_c) {                     // attention dropout on / off decided once, not per element
    constexpr bool DROP = decltype(drop_c)::value;
    f32x16 ag0, ag1;                                       // accumulators of the even / odd instance blocks
    // an accumulator starts as the biases of its block's dims (read from LDS straight into the accumulator)
    auto bias_init = [&](f32x16& acc, const float* gb) {    // gb = gpar + 4 hh + (first dim of the block)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float4 ta = ld4(gb + 8 * g), tb = ld4(gb + 256 + 8 * g);
        acc[4 * g] = ta.x; acc[4 * g + 1] = ta.y; acc[4 * g + 2] = ta.z; acc[4 * g + 3] = ta.w;
        acc[8 + 4 * g] = tb.x; acc[8 + 4 * g + 1] = tb.y; acc[8 + 4 * g + 2] = tb.z; acc[8 + 4 * g + 3] = tb.w;
      }
    };
    // One block.  MFMAs: block (dbM, ibM) into accM with the weights w.  Slices: the activations of block (dbA, ibA)
    // from accA (MODE 1) or the h copy-out (MODE 0).  Four k-steps of the next pass's weights (offset nw) go into
    // wn[4 slot ..]; at the end accA becomes the biases of the block it accumulates next (dims dbN ..).
    // The h fragments (B operands) come from LDS through a ring of four, three k-steps ahead and across block ends (ibN =
    // instance block of the next block): one step ahead, every MFMA waited out an LDS round trip.
    float4 fh[4];
    auto blk = [&](auto mode_c, const float4 (&w)[16], f32x16& accM, int ibM, int ibN, f32x16& accA, int dbA, int ibA,
                   float4 (&wn)[16], int slot, unsigned nw, int dbN) {
      constexpr int MODE = decltype(mode_c)::value;
      // lane-derived addresses are rebuilt per block from laundered lane coordinates: hoisted out of the round loop
      // they would be parked in scratch (64 fragment addresses alone)
      int rl = r, hl = hh;
      asm volatile("" : "+v"(rl), "+v"(hl));
      const char* hb = lds + F2_HIMG + (rl * F2_HROW + 16 * hl);                   // + (32 ib) rows + 32 s bytes: immediates
      const int row = row0 + 32 * ibA + rl;
      const float* gp = gpar + 4 * hl;
      float wc[8], te[16];
      uint32_t pa[4], pb[4];                               // [2 g + (0: dims 0-1, 1: dims 2-3 of the group)]
      bool kp[16];
      if constexpr (MODE == 0) {
#pragma unroll
        for (int s = 0; s < 3; ++s) fh[s] = *reinterpret_cast<const float4*>(hb + ibM * 32 * F2_HROW + 32 * s);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        fh[(s + 3) & 3] = *reinterpret_cast<const float4*>(hb + (s + 3 < 16 ? ibM : ibN) * 32 * F2_HROW + 32 * ((s + 3) & 15));
#ifndef MMF_F2_GATE_NOMM      /* diagnostic builds (wrong results): the gate phase without its MFMAs / without its activations */
        accM = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(w[s]), frag_of(fh[s & 3]), accM, 0, 0, 0);
#else
        asm volatile("" :: "v"(w[s].x), "v"(fh[s & 3].x));
#endif
#ifdef MMF_F2_GATE_NOACT
        if (MODE == 1) { if (s == 0) sc[ibA] += accA[0] + accA[15]; } else
#endif
        if constexpr (MODE == 0) {                         // whole 512-byte rows of h per wave instruction
          const int R = 8 * s + (tid >> 5), c = tid & 31;
          const float4 v = *reinterpret_cast<const float4*>(lds + F2_HIMG + R * F2_HROW + 16 * c);
#ifndef MMF_F2_NOHSTORE
          bst4(rh, (unsigned)(row0 + R) * 512u + 16u * (unsigned)c, v);
#else
          asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
#endif
        } else if (dbg & 16) {
          if (s == 0) sc[ibA] += accA[0] + accA[15];
        } else {
          // Element e = 0..15 of the block: g = e >> 3 (dims 8 g ..), j = (e >> 1) & 3, even e: tanh of accA[4 g + j], odd e:
          // sigmoid of accA[8 + 4 g + j].  An activation is a dependent chain (scale, exp2, 1 +, rcp, [2 x - 1]) of long-latency
          // instructions, so it runs as a three-stage pipeline ACROSS slices: slice s takes stage A of element s, stage B
          // of element s - 1 and stage C of element s - 2 -- three independent chains per slice; slice 15 drains.
          if (s == 0) {
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
              const float4 tw = ld4(gp + 512 + dbA + 8 * g2);
              wc[4 * g2] = tw.x; wc[4 * g2 + 1] = tw.y; wc[4 * g2 + 2] = tw.z; wc[4 * g2 + 3] = tw.w;
            }
          }
          auto stA = [&](int e) {                            // exp2 of the scaled pre-activation
            const int g = e >> 3, j = (e >> 1) & 3;
            te[e] = (e & 1) ? __builtin_amdgcn_exp2f(accA[8 + 4 * g + j] * -1.442695041f)
                            : __builtin_amdgcn_exp2f(accA[4 * g + j] * -2.885390082f);
            if constexpr (DROP) {
              const uint32_t idx = (uint32_t)row * 256u + (uint32_t)(dbA + 8 * g + 4 * hl + j);
              kp[e] = keep((e & 1) ? key_b : key_a, idx, thr_a);
            }
          };
          auto stB = [&](int e) { te[e] = __builtin_amdgcn_rcpf(1.0f + te[e]); };
          auto stC = [&](int e) {
            if (!(e & 1)) te[e] = __builtin_fmaf(2.0f, te[e], -1.0f);              // tanh = 2 sigmoid(2 x) - 1
            if ((e & 3) == 3) {                              // a pair of dims complete: a in te[e - 3], te[e - 1], b in te[e - 2], te[e]
              const int g = e >> 3, j = (e >> 1) & 3;        // j odd
              float a0 = te[e - 3], a1 = te[e - 1], b0 = te[e - 2], b1 = te[e];
              pa[2 * g + (j >> 1)] = pack2(a0, a1);
              pb[2 * g + (j >> 1)] = pack2(b0, b1);
              // the scores use a, b AS SAVED (bf16): forward and backward see the same activations
              unpack2(pa[2 * g + (j >> 1)], a0, a1);
              unpack2(pb[2 * g + (j >> 1)], b0, b1);
              if constexpr (DROP) {
                a0 = kp[e - 3] ? a0 * dscale : 0.f; a1 = kp[e - 1] ? a1 * dscale : 0.f;
                b0 = kp[e - 2] ? b0 * dscale : 0.f; b1 = kp[e] ? b1 * dscale : 0.f;
              }
              sc[ibA] = __builtin_fmaf(a0 * b0, wc[4 * g + j - 1], sc[ibA]);      // spelled out: the same two roundings for every block and lane
              sc[ibA] = __builtin_fmaf(a1 * b1, wc[4 * g + j], sc[ibA]);
            }
          };
          stA(s);
          if (s >= 1) stB(s - 1);
          if (s >= 2) stC(s - 2);
          if (s == 15) { stB(15); stC(14); stC(15); }
          if (s == 15) {
            // lane (r, hh) holds dims {4 hh + j} (g = 0) and {8 + 4 hh + j} (g = 1); after the swap lanes hh = 0 hold dims 0-7,
            // lanes hh = 1 dims 8-15 of instance r: one 16-byte store each
            uint32_t oa[4], ob[4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              auto sa = __builtin_amdgcn_permlane32_swap(pa[q], pa[2 + q], false, false);
              auto sb = __builtin_amdgcn_permlane32_swap(pb[q], pb[2 + q], false, false);
              oa[q] = sa[0]; oa[2 + q] = sa[1];
              ob[q] = sb[0]; ob[2 + q] = sb[1];
            }
            const unsigned o = (unsigned)row * 512u + (unsigned)(dbA + 8 * hl) * 2u;
#ifndef MMF_F2_NOSTORE
            bst4(rsa, o, make_float4(__uint_as_float(oa[0]), __uint_as_float(oa[1]), __uint_as_float(oa[2]), __uint_as_float(oa[3])));
            bst4(rsb, o, make_float4(__uint_as_float(ob[0]), __uint_as_float(ob[1]), __uint_as_float(ob[2]), __uint_as_float(ob[3])));
#else
            asm volatile("" :: "v"(oa[0]), "v"(oa[1]), "v"(oa[2]), "v"(oa[3]), "v"(ob[0]), "v"(ob[1]), "v"(ob[2]), "v"(ob[3]), "v"(o));
#endif
          }
        }
#ifndef MMF_F2_GATE_NOWN
        if (s >= 8 && s < 12) wn[4 * slot + s - 8] = bld4(rg, vg, nw + (unsigned)((4 * slot + s - 8) * 1024));
#endif
        if (s == 15) bias_init(accA, gp + dbN);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (!(dbg & 8)) {
      constexpr std::integral_constant<int, 0> copy_h{};
      constexpr std::integral_constant<int, 1> acts{};
      const int d0 = 16 * wave;
      bias_init(ag0, gpar + 4 * hh + d0);
      blk(copy_h, wg0, ag0, 0, 1, ag1, 0, 0, wg1, 0, 65536u, d0);          // block (0, 0); ag1 then accumulates block (0, 1)
#pragma unroll 1
      for (int p2 = 0; p2 < 2; ++p2) {
        asm volatile("" ::: "memory");                     // gate parameters and h fragments are re-read every round
        const int db = 128 * p2 + d0;                      // passes ps = 2 p2 (weights wg0) and ps + 1 (wg1)
        const unsigned n1 = __builtin_amdgcn_readfirstlane((unsigned)(2 * p2 + 1) * 65536u);   // passes 4, 5 do not exist: read as zero
        blk(acts, wg0, ag1, 1, 2, ag0, db, 0, wg1, 1, n1, db);
        blk(acts, wg0, ag0, 2, 3, ag1, db, 1, wg1, 2, n1, db);
        blk(acts, wg0, ag1, 3, 0, ag0, db, 2, wg1, 3, n1, db + 64);
        blk(acts, wg1, ag0, 0, 1, ag1, db, 3, wg0, 0, n1 + 65536u, db + 64);
        blk(acts, wg1, ag1, 1, 2, ag0, db + 64, 0, wg0, 1, n1 + 65536u, db + 64);
        blk(acts, wg1, ag0, 2, 3, ag1, db + 64, 1, wg0, 2, n1 + 65536u, db + 64);
        blk(acts, wg1, ag1, 3, 0, ag0, db + 64, 2, wg0, 3, n1 + 65536u, db + 128);
        blk(acts, wg0, ag0, 0, 1, ag1, db + 64, 3, wg1, 0, n1 + 131072u, db + 128);   // p2 = 1: the empty 17th block
      }
    }
  };
  if (drop) gate_phase(std::true_type{}); else gate_phase(std::false_type{});
#pragma unroll
  for (int ib = 0; ib < 4; ++ib) {
    const float s = sc[ib] + __shfl_xor(sc[ib], 32, 64);
    if (hh == 0) sred[wave * 128 + 32 * ib + r] = s;
  }
  __syncthreads();
  B2_MARK(2);

  // ---------------- scores of the tile, online-softmax partial, pooling partial ----------------------------------------
  const float bc = p.bc[0];
  float sv = -INFINITY;
  if (tid < 128) {
    const int row = row0 + tid;
    const float s = bc + sred[tid] + sred[128 + tid] + sred[256 + tid] + sred[384 + tid];
    if (row < p.N) { p.A_raw[row] = s; sv = s; }
  }
  float m = wave_max(sv);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(red[0], red[1]);                               // the rows live in threads 0..127 = waves 0, 1
  const float ev = sv > -INFINITY ? __expf(sv - m) : 0.f;
  if (tid < 128) e_l[tid] = ev;
  const float lsum = wave_sum(ev);
  if (lane == 0) red[8 + wave] = lsum;
  __syncthreads();
  if (!(dbg & 32)) {
    const int fg = tid & 31, ig = tid >> 5;                // 8 features x 16 instances per thread
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int R = 16 * ig + i;
      const float4 raw = *reinterpret_cast<const float4*>(lds + F2_HIMG + R * F2_HROW + 16 * fg);
      float hv[8];
      unpack8(raw, hv);
      const float e = e_l[R];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += e * hv[k];
    }
    float* o = vred + ig * 256 + 8 * fg;
    st4(o, make_float4(v[0], v[1], v[2], v[3]));
    st4(o + 4, make_float4(v[4], v[5], v[6], v[7]));
  }
  __syncthreads();
  float* out = p.partials + (size_t)mt * (2 + 256);
  {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += vred[q * 256 + tid];
    out[2 + tid] = s;
  }
  if (tid == 0) { out[0] = m; out[1] = red[8] + red[9]; }
  }
  B2_MARK(3);
  B2_COUNT();
}

bool fused_fwd2_ok(int64_t N, int L, int H, int D) {
  static const int env = tune_int("MMF_BF16_FUSED", 2);   // A/B switch: 1 = first form, 0 = unfused kernels
  return env == 2 && H == 256 && D == 256 && L % 128 == 0 && fused_fwd_tiles(N) <= 4096;
}

int launch_fused_fwd2_bf16(FusedFwdParams p, int gated, hipStream_t st) {
  if (!gated || p.D != 256 || p.L % 128 != 0 || !p.w1f || !p.wabf) return MMF_ERR_SHAPE;
  p.mt_count = fused_fwd_tiles(p.N);
  static const int dbg = tune_int("MMF_F2_DEBUG_MASK", 0);
  static const int hl = tune_int("MMF_F2_HLOOP", 1);       // A/B switch
  p.stagger = dbg;
  p.hash_in_loop = hl;
  const bool hloop = p.p_h > 0.f && p.L == 1024 && p.hash_in_loop;
  auto kern = hloop ? amil_fwd_fused2_bf16_kernel<true> : amil_fwd_fused2_bf16_kernel<false>;
  static const int lds_env = tune_int("MMF_F2_LDS", 0);     // experiment: > 80 KB forces one workgroup per CU
#ifdef MMF_F2_REV_GPAR
  const int lds_bytes = lds_env > 0 ? lds_env : F2_LDS_BYTES;       // experiment: any size (the gate parameters are not in LDS)
#else
  const int lds_bytes = lds_env > F2_LDS_BYTES ? lds_env : F2_LDS_BYTES;
#endif
  if (int e = set_dyn_lds(reinterpret_cast<const void*>(kern), lds_bytes)) return e;
  ProfScope ps("amil_fwd_fused_bf16_kernel", st);
  hipLaunchKernelGGL(kern, dim3(p.mt_count), dim3(256), lds_bytes, st, p);
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

}  // namespace mmf
