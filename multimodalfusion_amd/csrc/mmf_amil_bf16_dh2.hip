// K-dh, second form (bf16 storage, H = D = 256, gated): du = bf16((dP . [Wa ; Wb] + p dM) . relu'(h) . scale_h) with K-prep
// fused in front, built like the second form of the fused forward (mmf_amil_bf16_fwd2.hip): a 128-instance tile per
// 4-wave workgroup, TWO workgroups per CU, the product TRANSPOSED (A operand = 32 rows of [Wa ; Wb]^T, i.e. 32 hidden
// features; B operand = 32 instances), so that a lane of the accumulator holds one instance and 4 consecutive features.
//
//   prep      the tile's h rows go to LDS once (64 KB, swizzled as the forward's h image): g_i = dM . h_i, p_i, ds_i
//             (models/model_attention_mil_path.py:55-56 backwards: softmax, weighted sum), and the relu' bits of the 128
//             (instance, feature) pairs a lane will store -- 4 registers -- are taken from it; then the region is free.
//   loop      K = 512 in 8 chunks of 64 (32 attention dims: their d pre-tanh, then their d pre-sigmoid).  Chunk kt + 1 of the
//             dP operand is BUILT (a, b, ds, Wc -> 2 x 8 values per thread and instance, models/model_modules.py:105-110
//             backwards) into the other LDS stage -- and written to HBM for the TN kernel, with the dWc sums taken on the
//             way -- in the same scheduling region as the MFMAs of chunk kt: a SIMD issues a wave's own vector instructions
//             in the shadow of its MFMAs (tools/coissue.hip), and the build is the longer of the two.  The weights never
//             touch LDS: [Wa ; Wb]^T is stored in MFMA-fragment order (CvtSeg::transpose 5), a wave loads only the 64
//             features it multiplies, 1 KB of contiguous memory per instruction, each fragment re-filled for the next
//             chunk right behind the MFMAs that used it.
//   epilogue  the accumulators start as p_i dM[f]; du = relu' bit ? acc . scale_h : 0 is packed 4 features to 8 bytes into
//             the LDS du image and leaves in whole 512-byte rows.
//
// LDS map (bytes): [0, 65536) h image, then two dP stages of [128 rows][128 B] at 0 / 16384, then the du image;
//                  [65536, ...) ds[128], p[128], g[128], scratch[32], Wc[256], dWc partials [4 waves][256].
#include <type_traits>
#include <cstdlib>

#include "mmf_gemm_dma.h"
#include "mmf_bf16.h"

namespace mmf {

constexpr int D2_BM = 128;
constexpr int D2_STAGE = D2_BM * 128;
constexpr int D2_IMG = 0, D2_MISC = 65536;
constexpr int D2_LDS_BYTES = D2_MISC + (3 * 128 + 32 + 256 + 4 * 256) * 4;

__device__ inline bf16x8 d2_frag(const float4& v) {
  f32x4 t = {v.x, v.y, v.z, v.w};
  return __builtin_bit_cast(bf16x8, t);
}
__device__ inline void d2_bst4(rsrc_t r, unsigned voff, const float4& v) {
  u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)voff, 0, 0);
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void dh2_bf16_kernel(DhBfParams p) {
  extern __shared__ __align__(16) char ldsd[];
  char* lds = ldsd;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int mt = blockIdx.x, row0 = mt * D2_BM;
  float* ds_l = reinterpret_cast<float*>(lds + D2_MISC);          // [128]
  float* p_l = ds_l + 128;                                         // [128]
  float* g_l = p_l + 128;                                          // [128]
  float* red = g_l + 128;                                          // [32]
  float* wc_l = red + 32;                                          // [256]
  float* dwc_l = wc_l + 256;                                       // [4][256]
  const unsigned act_bytes = (unsigned)p.N * 512u;                 // h, a, b, du rows are 512 bytes

  // resources and per-thread offsets of the main loop
  const rsrc_t ra = make_rsrc(p.g.a, act_bytes), rbb = make_rsrc(p.g.b, act_bytes);
  const rsrc_t rdp = make_rsrc(p.dP, (unsigned)p.N * 1024u);
  const rsrc_t rw = make_rsrc(p.WabT, 256u * 512u * 2u);
  const unsigned vw = (unsigned)lane * 16u + (unsigned)wave * 8192u;
  const int piece = tid & 3, rl0 = tid >> 2;                       // builder: dims 8 piece .. + 7 of the chunk, rows rl0 and rl0 + 64
  const uint32_t sd = p.g.seed_dev ? *p.g.seed_dev : 0u;
  const uint32_t key_a = p.g.key_a + sd, key_b = p.g.key_b + sd;
  const uint32_t thr = drop_threshold(p.g.drop_p);
  const float dscale = p.g.drop_p > 0.f ? 1.0f / (1.0f - p.g.drop_p) : 1.0f;
  unsigned vab[2], vdp[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rl = rl0 + 64 * j;
    vab[j] = (unsigned)(row0 + rl) * 512u + 16u * (unsigned)piece;          // + 64 kt: the chunk's 32 dims
    vdp[j] = (unsigned)(row0 + rl) * 1024u + 16u * (unsigned)piece;         // d pre-tanh at + 64 kt, d pre-sigmoid at + 512 + 64 kt
  }
  float4 wfr[8];                                                   // A fragments of the chunk in flight: [2 q + fb]
  float4 la4[2][2], lb4[2][2];                                     // a, b of the next two chunks to be built: [chunk & 1][instance]
  const float my_A = (tid < 128 && row0 + tid < p.N) ? p.A_raw[row0 + tid] : 0.f;
  const float my_gA = (tid < 128 && row0 + tid < p.N && p.gA) ? p.gA[row0 + tid] : 0.f;

  // ---------------- prep: h tile -> LDS, g_i = dM . h_i, p_i, ds_i, relu' bits --------------------------------------------
  {
    const rsrc_t rh = make_rsrc(p.h, act_bytes);
    const int s = tid & 31;
    float4 hv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) hv[i] = bld4(rh, (unsigned)(row0 + 8 * i + (tid >> 5)) * 512u + 16u * (unsigned)s, 0);   // rows beyond the bag: zero
    wc_l[tid] = p.g.Wc[tid];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int R = 8 * i + (tid >> 5);
      *reinterpret_cast<float4*>(lds + D2_IMG + R * 512 + 16 * (s ^ (R & 15))) = hv[i];
    }
  }
  // the main loop's first operands: requested now (the 64 registers of the h tile are free again), landing beside the prep
#pragma unroll
  for (int j = 0; j < 8; ++j) wfr[j] = bld4(rw, vw, (unsigned)(j * 1024));
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    la4[0][j] = bld4(ra, vab[j], 0); lb4[0][j] = bld4(rbb, vab[j], 0);
    la4[1][j] = bld4(ra, vab[j], 64); lb4[1][j] = bld4(rbb, vab[j], 64);
  }

  float dmm = 0.f;                                                 // dM . M
  dmm = p.dM[lane] * p.Mpool[lane] + p.dM[lane + 64] * p.Mpool[lane + 64] + p.dM[lane + 128] * p.Mpool[lane + 128]
      + p.dM[lane + 192] * p.Mpool[lane + 192];
  dmm = wave_sum(dmm);
  __syncthreads();
  {
    // thread t: row t / 2, features 128 (t & 1) .. + 127
    const int R = tid >> 1, half = tid & 1;
    float acc = 0.f;
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const int slot = 16 * half + s;
      const float4 raw = *reinterpret_cast<const float4*>(lds + D2_IMG + R * 512 + 16 * (slot ^ (R & 15)));
      float hv[8];
      unpack8(raw, hv);
      const float4 d0 = ld4(p.dM + 8 * slot), d1 = ld4(p.dM + 8 * slot + 4);
      acc += hv[0] * d0.x + hv[1] * d0.y + hv[2] * d0.z + hv[3] * d0.w + hv[4] * d1.x + hv[5] * d1.y + hv[6] * d1.z + hv[7] * d1.w;
    }
    acc += __shfl_xor(acc, 1, 64);
    if (half == 0) g_l[R] = acc;
  }
  __syncthreads();
  {
    const float smax = p.stats[0], inv = 1.0f / p.stats[1];
    float dbc = 0.f;
    if (tid < 128) {
      const int row = row0 + tid;
      float pi = 0.f, d = 0.f;
      if (row < p.N) {
        pi = __expf(my_A - smax) * inv;
        d = pi * (g_l[tid] - dmm) + my_gA;
        p.p_out[row] = pi; p.ds_out[row] = d;
      }
      ds_l[tid] = d;
      p_l[tid] = pi;
      dbc = d;
    }
    dbc = wave_sum(dbc);
    if (lane == 0) red[wave] = dbc;
  }
  // relu' bits of this lane's 128 outputs: bit e = ((fb 4 + ib) 4 + g) 4 + j  <->  instance 32 ib + r, feature 64 w + 32 fb + 8 g + 4 hh + j
  uint32_t rb[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int fb = 0; fb < 2; ++fb)
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
      const int R = 32 * ib + r;
      uint32_t bits = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int slot = 8 * wave + 4 * fb + g;
        const uint2 w2 = *reinterpret_cast<const uint2*>(lds + D2_IMG + R * 512 + 16 * (slot ^ (R & 15)) + 8 * hh);
        // h >= 0 always (post-ReLU, dropped entries are +0): "h > 0" is "bits non-zero"
        bits |= ((w2.x & 0xFFFFu) ? 1u : 0u) << (4 * g) | ((w2.x >> 16) ? 1u : 0u) << (4 * g + 1)
              | ((w2.y & 0xFFFFu) ? 1u : 0u) << (4 * g + 2) | ((w2.y >> 16) ? 1u : 0u) << (4 * g + 3);
      }
      rb[2 * fb + (ib >> 1)] |= bits << (16 * (ib & 1));
      asm volatile("" ::: "memory");                               // one block's four reads at a time (all 32 at once spill 64 registers)
    }
  // the bits are wanted HERE, as four registers: left alone the compiler keeps the 64 raw dwords (in scratch: +200 MB of HBM
  // traffic per launch at 100k) and extracts the bits in the epilogue
  asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]));
  __syncthreads();                                                 // ds / p complete; every read of the h image done
  if (tid == 0) p.dbc_part[mt] = red[0] + red[1] + red[2] + red[3];

  // ---------------- accumulators start as p_i dM[f] ---------------------------------------------------------------------
  f32x16 acc[2][4];
  {
    float pi[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) pi[ib] = p_l[32 * ib + r];
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 dm = ld4(p.dM + 64 * wave + 32 * fb + 8 * g + 4 * hh);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
          acc[fb][ib][4 * g] = pi[ib] * dm.x; acc[fb][ib][4 * g + 1] = pi[ib] * dm.y;
          acc[fb][ib][4 * g + 2] = pi[ib] * dm.z; acc[fb][ib][4 * g + 3] = pi[ib] * dm.w;
        }
      }
  }

  // ---------------- main loop ---------------------------------------------------------------------------------------------
  float dsv[2];
  dsv[0] = ds_l[rl0]; dsv[1] = ds_l[rl0 + 64];
  // The build of one chunk of the dP operand (from la4 / lb4: 2 instances x 8 dims per thread), cut into 16 SLICES of one
  // (instance, dim pair, dim) each -- ~15 vector instructions -- so that the chunk loop can put one slice behind every second
  // MFMA: a SIMD issues a wave's own vector instructions in the shadow of its MFMA (32 cycles), but hipcc emits independent
  // MFMAs back to back when left alone (it fills latency gaps, and eight accumulators leave none).
  uint32_t pw[4], qw[4];                                           // packed d pre-tanh / d pre-sigmoid of the instance being built
  float wsum[8], wcv[8];
  auto slice = [&](int s, int kt, char* stage, const float4 (&la)[2], const float4 (&lb)[2]) {   // s = 8 j + 2 q + e
    const int j = s >> 3, q = (s >> 1) & 3, e = s & 1;
    const int d0 = 32 * kt + 8 * piece;
    const int rl = rl0 + 64 * j;
    const uint32_t aw = __float_as_uint(q == 0 ? la[j].x : q == 1 ? la[j].y : q == 2 ? la[j].z : la[j].w);
    const uint32_t bw = __float_as_uint(q == 0 ? lb[j].x : q == 1 ? lb[j].y : q == 2 ? lb[j].z : lb[j].w);
    const float av = e ? __uint_as_float(aw & 0xFFFF0000u) : __uint_as_float(aw << 16);
    const float bv = e ? __uint_as_float(bw & 0xFFFF0000u) : __uint_as_float(bw << 16);
    float ma = 1.f, mb = 1.f;
    if constexpr (DROP) {
      const uint32_t idx = (uint32_t)(row0 + rl) * 256u + (uint32_t)(d0 + 2 * q + e);
      ma = keep(key_a, idx, thr) ? dscale : 0.f;
      mb = keep(key_b, idx, thr) ? dscale : 0.f;
    }
    const float t = dsv[j] * wcv[2 * q + e];
    const float am = av * ma, bm = bv * mb;
    const float oa = t * bm * ma * (1.f - av * av);                // d pre-tanh    (mmf_kernels.h: gate_dp_t, PART 0)
    const float ob = t * am * mb * bv * (1.f - bv);                // d pre-sigmoid (PART 1)
    wsum[2 * q + e] = __builtin_fmaf(dsv[j], am * bm, wsum[2 * q + e]);
    // pack: the low half first, the high half completes the dword
    if (e == 0) { pw[q] = (uint32_t)f2bf(oa); qw[q] = (uint32_t)f2bf(ob); }
    else { pw[q] |= (uint32_t)f2bf(oa) << 16; qw[q] |= (uint32_t)f2bf(ob) << 16; }
    if (q == 3 && e == 1) {                                        // the instance's 2 x 16 bytes are complete
      const float4 pa = make_float4(__uint_as_float(pw[0]), __uint_as_float(pw[1]), __uint_as_float(pw[2]), __uint_as_float(pw[3]));
      const float4 pb = make_float4(__uint_as_float(qw[0]), __uint_as_float(qw[1]), __uint_as_float(qw[2]), __uint_as_float(qw[3]));
      const int sw2 = (rl >> 1) & 7;
      *reinterpret_cast<float4*>(stage + rl * 128 + 16 * (piece ^ sw2)) = pa;
      *reinterpret_cast<float4*>(stage + rl * 128 + 16 * ((4 + piece) ^ sw2)) = pb;
      d2_bst4(rdp, vdp[j] + 64u * (unsigned)kt, pa);               // rows beyond the bag are dropped by the range check
      d2_bst4(rdp, vdp[j] + 512u + 64u * (unsigned)kt, pb);
    }
  };
  auto begin_build = [&](int kt) {
    const int d0 = 32 * kt + 8 * piece;
    const float4 w0 = ld4(wc_l + d0), w1 = ld4(wc_l + d0 + 4);
    wcv[0] = w0.x; wcv[1] = w0.y; wcv[2] = w0.z; wcv[3] = w0.w; wcv[4] = w1.x; wcv[5] = w1.y; wcv[6] = w1.z; wcv[7] = w1.w;
#pragma unroll
    for (int e = 0; e < 8; ++e) wsum[e] = 0.f;
  };
  // dWc partial of the chunk's dims = sum over the wave's 32 rows, i.e. over the 16 lanes that share `piece` (lane bits 2-5).
  // Recursive halving: with the partner across lane bit 2 a lane trades four of its eight sums and keeps four, across bit 3
  // two, across bit 4 one, across bit 5 the two lanes add what is left -- 8 cross-lane moves instead of 32 -- and ends with
  // the total of dim 4 b2 + 2 b3 + b4 of the piece.  The moves' byte addresses are fixed per lane.
  const int xa4 = ((lane ^ 4) << 2), xa8 = ((lane ^ 8) << 2), xa16 = ((lane ^ 16) << 2), xa32 = ((lane ^ 32) << 2);
  const bool b2 = lane & 4, b3 = lane & 8, b4 = lane & 16;
  auto bperm = [&](int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); };
  auto end_build = [&](int kt) {
    float k4[4], k2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float keep_v = b2 ? wsum[4 + i] : wsum[i], send_v = b2 ? wsum[i] : wsum[4 + i];
      k4[i] = keep_v + bperm(xa4, send_v);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float keep_v = b3 ? k4[2 + i] : k4[i], send_v = b3 ? k4[i] : k4[2 + i];
      k2[i] = keep_v + bperm(xa8, send_v);
    }
    const float keep_v = b4 ? k2[1] : k2[0], send_v = b4 ? k2[0] : k2[1];
    float tot = keep_v + bperm(xa16, send_v);
    tot += bperm(xa32, tot);
    dwc_l[wave * 256 + 32 * kt + 8 * piece + (b2 ? 4 : 0) + (b3 ? 2 : 0) + (b4 ? 1 : 0)] = tot;   // both halves of the wave write the same value
  };
  auto load_ab = [&](int kt, float4 (&la)[2], float4 (&lb)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { la[j] = bld4(ra, vab[j], (unsigned)(64 * kt)); lb[j] = bld4(rbb, vab[j], (unsigned)(64 * kt)); }
  };

  begin_build(0);
#pragma unroll
  for (int s = 0; s < 16; ++s) slice(s, 0, lds + D2_IMG, la4[0], lb4[0]);
  end_build(0);
  __syncthreads();
  const int sw = (r >> 1) & 7;
  // One chunk: its 32 MFMAs with -- pinned behind every second one -- a slice of the next chunk's build.  Branch-free; the
  // last chunk has nothing to build and is its own instantiation.
  // (labuild, lbbuild): a, b of chunk kt + 1, loaded one chunk ago; (lanext, lbnext): where chunk kt + 2 is loaded to now
  auto chunk = [&](int kt, auto with_build_c, float4 (&labuild)[2], float4 (&lbbuild)[2], float4 (&lanext)[2], float4 (&lbnext)[2]) {
    constexpr bool WITH_BUILD = decltype(with_build_c)::value;
    const char* xs = lds + D2_IMG + (kt & 1) * D2_STAGE + r * 128;
    char* nxt = lds + D2_IMG + ((kt + 1) & 1) * D2_STAGE;
    const unsigned nw = __builtin_amdgcn_readfirstlane((unsigned)(kt + 1) * 32768u);   // chunk 8: beyond the buffer, reads zero
    float4 fx[4];                                                  // B fragments: re-filled for the next k-step behind their last use
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) fx[ib] = *reinterpret_cast<const float4*>(xs + ib * 32 * 128 + 16 * ((0 + hh) ^ sw));
    if constexpr (WITH_BUILD) {
      load_ab(kt + 2 < 8 ? kt + 2 : 7, lanext, lbnext);            // a whole chunk to land (the last re-read is not used)
      begin_build(kt + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 32; ++i) {                                 // MFMA i: k-step q = i / 8, feature block fb, instance block ib
      const int q = i >> 3, fb = (i >> 2) & 1, ib = i & 3;
      acc[fb][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d2_frag(wfr[2 * q + fb]), d2_frag(fx[ib]), acc[fb][ib], 0, 0, 0);
      if (fb == 1 && q + 1 < 4) fx[ib] = *reinterpret_cast<const float4*>(xs + ib * 32 * 128 + 16 * ((2 * (q + 1) + hh) ^ sw));
      if constexpr (WITH_BUILD) {
        if ((i & 3) == 3) wfr[2 * q + fb] = bld4(rw, vw, nw + (unsigned)((2 * q + fb) * 1024));   // this fragment's last use
        if (i & 1) slice(i >> 1, kt + 1, nxt, labuild, lbbuild);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (WITH_BUILD) end_build(kt + 1);
    __syncthreads();
  };
#pragma unroll 1
  for (int kt = 0; kt < 6; kt += 2) {
    chunk(kt, std::true_type{}, la4[1], lb4[1], la4[0], lb4[0]);
    chunk(kt + 1, std::true_type{}, la4[0], lb4[0], la4[1], lb4[1]);
  }
  chunk(6, std::true_type{}, la4[1], lb4[1], la4[0], lb4[0]);
  chunk(7, std::false_type{}, la4[0], lb4[0], la4[1], lb4[1]);

  // ---------------- epilogue: du = relu' ? acc . scale_h : 0 -> LDS du image -> whole rows to HBM -----------------------------
  // (addresses from a thread index the compiler cannot see through, or it computes them at the top and parks them in scratch)
  int tid_late = threadIdx.x;
  asm volatile("" : "+v"(tid_late));
  {
  const int tid = tid_late, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  {
    const float sh = p.scale_h;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) {
        const int R = 32 * ib + r;
        char* rowp = lds + D2_IMG + R * 512 + 8 * hh;
        const uint32_t bits = rb[2 * fb + (ib >> 1)] >> (16 * (ib & 1));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float y[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = (bits >> (4 * g + j)) & 1u ? acc[fb][ib][4 * g + j] * sh : 0.f;
          const int slot = 8 * wave + 4 * fb + g;
          *reinterpret_cast<uint2*>(rowp + 16 * (slot ^ (R & 15))) = pack4(y[0], y[1], y[2], y[3]);
        }
      }
  }
  __syncthreads();
  {
    const rsrc_t rdu = make_rsrc(p.du, act_bytes);
    const int s = tid & 31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int R = 8 * i + (tid >> 5);
      const float4 v = *reinterpret_cast<const float4*>(lds + D2_IMG + R * 512 + 16 * (s ^ (R & 15)));
      d2_bst4(rdu, (unsigned)(row0 + R) * 512u + 16u * (unsigned)s, v);
    }
  }
  // per-tile dWc partial: the four waves' row sums
  p.dwc_part[(size_t)mt * 256 + tid] = dwc_l[tid] + dwc_l[256 + tid] + dwc_l[512 + tid] + dwc_l[768 + tid];
  }
}

bool dh2_bf16_ok(int64_t N, int H, int D, int gated) {
  static const int env = tune_int("MMF_BF16_DH2", 1);   // A/B switch
  return env && gated && H == 256 && D == 256 && N > 0;
}

int launch_dh2_bf16(DhBfParams p, hipStream_t st) {
  if (!dh2_bf16_ok(p.N, p.H, p.g.D, p.g.gated)) return MMF_ERR_SHAPE;
  p.mt_count = (int)((p.N + D2_BM - 1) / D2_BM);
  p.nt_count = 1;
  const bool drop = p.g.drop_p > 0.f;
  auto kern = drop ? dh2_bf16_kernel<true> : dh2_bf16_kernel<false>;
  if (int e = set_dyn_lds(reinterpret_cast<const void*>(kern), D2_LDS_BYTES)) return e;
  ProfScope ps("dh_bf16_kernel", st);
  hipLaunchKernelGGL(kern, dim3(p.mt_count), dim3(256), D2_LDS_BYTES, st, p);
  return hipGetLastError() == hipSuccess ? MMF_OK : MMF_ERR_LAUNCH;
}

}  // namespace mmf
