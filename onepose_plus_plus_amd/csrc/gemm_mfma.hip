// MFMA GEMM / implicit-GEMM convolution for gfx950.
//
// Replaces, on the hot path, what the reference gets from cuDNN/cuBLAS through nn.Conv2d,
// nn.Linear and torch.einsum:
//   backbone convs      src/models/OnePosePlus/backbone/resnet.py:10-45, :101-124, :141-164
//   transformer Linears src/models/OnePosePlus/loftr_module/transformer.py:26-47, :76-92
//   coarse score einsum src/models/OnePosePlus/utils/coarse_matching.py:102-107
//
// Design (MI355X-first, not a translated CUDA tiling):
//  * Three operand arithmetics behind one template (PREC), always fp32 in / fp32 accumulate / fp32 out:
//      bf16x3 (default)  every operand carried EXACTLY as hi + mid + lo bf16, six v_mfma_f32_32x32x16_bf16 per product
//                        (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi): not narrower than fp32, 2.65x the fp32 MFMA rate;
//                        weights pre-split at pack time (48 B per 8 k), activations split on their way to LDS;
//      fp32              v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain (157 TFLOP/s);
//      fp16x2            narrower than fp32 (hi + lo fp16 pairs, three fp16 MFMAs per product): instantiated in the tuning library only (r06).
//    bf16x3 convolutions can also read their activations PRE-SPLIT (opp_gemm_asp_kernel, r06: the producing epilogue writes the triples once;
//    opt-in OPP_ASP=1, measured no faster end to end).
//  * Both operands are K-contiguous ("TN").  A 32-wide K chunk of a tile row is one 128 B line of fp32 activations
//    (192 B of pre-split weights): a wave loads 8 rows per instruction with raw buffer loads (out-of-range lanes -- the
//    padding halo of a convolution, rows past M -- read zeros from the buffer unit, so the K loop has no exec-masked control
//    flow).  LDS rows are padded (36 / 52 floats) so that the 16 rows of a ds_read_b128 lane group fall on 16 distinct
//    4-bank groups; within a chunk the k index is permuted so that lane half h owns one 8-k group per k16-step.
//  * Implicit im2col: the A row of an output pixel for chunk (tap, channel group) is one contiguous run of the NHWC input.
//    Inputs with 32 n + (1..4) channels (the 196-channel stages) pack their last channels 8 taps to a chunk.
//  * Software pipeline: register-staged prefetch two chunks ahead, global loads and the LDS hand-over issued ONE PER MFMA
//    SLOT, the barrier ahead of the last MFMA group so that its skew hides under MFMAs.
//  * 8-wave tiles (two waves per SIMD) 256x128 / 128x256 / 128x128 / 64x128 / 64x256, 4-wave 64x64; XCD-aware tile order.
//    128x192 (32x96 per wave) and 128x224 (32x128 | 32x96 per wave on FOUR rotating fragment sets, chunk_ring) for the 196-channel layers.
//  * Epilogue through the idle operand LDS, 16 B per lane: folded-BN bias, residual (direct or bilinear x2
//    align_corners=True), ReLU / LeakyReLU / elu+1, value scaling, LayerNorm of whole rows, dual-softmax statistics.
//  * Dense split-K (grid.y) for the weight gradients of the training step (linear_bwd.hip).
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include <vector>

#include "opp_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

// floats per LDS tile row of one 32-k chunk: fp32 / fp16x2 32 + 4 pad; bf16x3 48 + 4 pad (both keep 16 B
// alignment and make the 16 rows of a ds_read_b128 lane group land on 16 distinct 4-bank groups)
constexpr int lds_stride(int prec) { return prec == OPP_PREC_BF16X3 ? 52 : 36; }

// bf16x3 split of four consecutive k values: x = hi + mid + lo EXACTLY, hi = bf16_rne(x), mid = bf16_rne(x - hi), lo = bf16(x - hi - mid)
// (|mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|, the last residual has <= 8 significant bits).  ONE definition for the K loop (activations split on
// their way to LDS), for the epilogues that emit activations already split (ASP consumers, r06) and for opp_pack_b3: the same bits everywhere.
__device__ __forceinline__ void opp_split4_b3(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
  auto lvl = [](float a, float b, float& ra, float& rb) -> unsigned {
    const f32x2 t = {a, b};
    const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
    ra = a - __uint_as_float(p << 16);
    rb = b - __uint_as_float(p & 0xffff0000u);
    return p;
  };
  float r0, r1, r2, r3, s0, s1, s2, s3, u0, u1, u2, u3;
  const unsigned h01 = lvl(v.x, v.y, r0, r1), h23 = lvl(v.z, v.w, r2, r3);
  const unsigned m01 = lvl(r0, r1, s0, s1), m23 = lvl(r2, r3, s2, s3);
  const unsigned l01 = lvl(s0, s1, u0, u1), l23 = lvl(s2, s3, u2, u3);
  hi = make_uint2(h01, h23);
  mid = make_uint2(m01, m23);
  lo = make_uint2(l01, l23);
}
// four consecutive channels (col % 4 == 0) of a pixel row in the pre-split activation layout: every 8 channels = 48 B [hi x8 | mid x8 | lo x8]
__device__ __forceinline__ void opp_store_split4(void* base, size_t row_bytes_off, int col, const float4 v) {
  uint2 hi, mid, lo;
  opp_split4_b3(v, hi, mid, lo);
  char* p = static_cast<char*>(base) + row_bytes_off + (col >> 3) * 48 + (col & 7) * 2;
  *reinterpret_cast<uint2*>(p) = hi;
  *reinterpret_cast<uint2*>(p + 16) = mid;
  *reinterpret_cast<uint2*>(p + 32) = lo;
}

// ASP (r06, bf16x3 convolutions over Cin % 32 == 0): the activation operand arrives PRE-SPLIT in memory (the layout above, written by the
// producing layer's epilogue), is loaded in 16-byte pieces exactly like the weight rows and goes to LDS with one ds_write_b128 per piece --
// no split arithmetic in the K loop (it was redone for every tap of a 3 x 3 window and every column tile: ~120 of the ~190 VALU instructions
// per 32-k chunk and wave, profiles/r06_pmc_conv_*.txt) and no 8-byte LDS stores (a quarter of the LDS cycles were their bank conflicts).
// (The body is a device function with two kernel entries -- opp_gemm_kernel<..., PREC> and opp_gemm_asp_kernel<..., PREC> -- so that the
// symbol names of the default kernels, which every profile under profiles/ is keyed by, do not change with the opt-in variant.)
template <int BM, int BN, int WAVES_M, int WAVES_N, bool CONV, int ABL, int DEPTH, int PREC, bool ASP>
__device__ __forceinline__ void opp_gemm_body(const OppGemm& g) {
  constexpr bool H2 = PREC == OPP_PREC_FP16X2;   // operands as hi+lo fp16 pairs, 3 fp16 MFMA products
  constexpr bool H3 = PREC == OPP_PREC_BF16X3;   // operands as hi+mid+lo bf16 triples (exact), 6 bf16 MFMA products
  constexpr int kLdsStride = lds_stride(PREC);
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int NT32 = BN / 32;                               // 32-column sub-tiles per block
  constexpr int TN = (NT32 + WAVES_N - 1) / WAVES_N;          // per wave (last wave may own fewer)
  constexpr bool kRagged = (NT32 % WAVES_N) != 0;             // e.g. 224 columns on 2 waves = 4 + 3
  static_assert(!ASP || (CONV && PREC == OPP_PREC_BF16X3 && (ABL == 0 || ABL >= 200)), "pre-split activations: bf16x3 convolutions only");
  // tuning builds, timing only: 91..94 remove one part of the K loop; 200 + mask removes several (bit 0 the global prefetch, 1 the LDS hand-over,
  // 2 the barrier, 3 the fragment reads) -- 215 is the bare MFMA sequence, the parts come back one at a time (tools/r06_abl2.sh)
  constexpr int kAblMask = ABL >= 200 ? ABL - 200 : 0;
  constexpr bool kNoGload = ABL == 91 || (kAblMask & 1), kNoLdsSt = ABL == 92 || (kAblMask & 2), kNoBar = ABL == 93 || (kAblMask & 4),
                 kNoFrag = ABL == 94 || (kAblMask & 8);
  constexpr int A_LD = ASP ? (BM * 12 + NT - 1) / NT : BM * 8 / NT;   // ASP: 192 B per row and chunk, 16-byte pieces like the weights
  constexpr bool kAFrac = ASP && (BM * 12) % NT != 0;                 // (64-row tile on 8 waves: 1.5 pieces per thread)
  // 224-column ring tile: weight rows >= 208 are never real (the tile serves N <= 208, i.e. the 196-channel layers; the launcher checks) --
  // their LDS rows are zeroed once and the loader covers 208 rows = 4.9 pieces per thread instead of 5.25 (one prefetch register set less)
  constexpr int kBRowsLoad = (H3 && BN == 224) ? 208 : BN;
  constexpr int B_LD = H3 ? (kBRowsLoad * 12 + NT - 1) / NT : BN * 8 / NT;   // bf16x3 weights: 192 B per row and chunk
  // a 192-column tile (tuning builds: 196 = 192 + a tail) has 4.5 weight pieces per thread: the upper half of the workgroup loads zeros
  // for its fifth piece and does not store it
  constexpr bool kBFrac = H3 && (kBRowsLoad * 12) % NT != 0;
  // 224-column tile (r05: the 196(->224)-channel layers in ONE column tile, 7 sub-tiles of 32 columns on 2 x 4 waves): the waves with
  // wn = 0 own 4 sub-tiles, those with wn = 1 own 3, and the wave -> (wm, wn) map puts one of each on every SIMD (waves w and w + 4 share
  // SIMD w % 4), so every matrix pipe runs 7 sub-tiles per k-step where the 128 x 256 tile runs 8.  The 32 x 128 wave tile only fits the
  // 256 registers of a two-waves-per-SIMD kernel with FOUR operand fragment sets instead of six: kRing (chunk_ring below).
  constexpr bool kRing = H3 && kRagged;
  static_assert(BM % (WAVES_M * 32) == 0 && BN % 32 == 0, "tile shape");
  static_assert((BM * 8) % NT == 0 && (H3 ? (BN * 12) % NT == 0 || BN == 192 || BN == 224 : (BN * 8) % NT == 0), "load split");
  static_assert(!ASP || !((H3 && ((BN + 31) / 32 % WAVES_N) != 0)), "pre-split activations are not built for the ring tile");

  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
  if (ABL == 9 || ABL > 90) ts0 = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                         // [2][BM][36]
  float* Bs = smem + 2 * BM * kLdsStride;   // [2][BN][36]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = __builtin_amdgcn_readfirstlane(kRing ? wave % WAVES_M : wave / WAVES_N);
  const int wn = __builtin_amdgcn_readfirstlane(kRing ? wave / WAVES_M : wave % WAVES_N);
  bool tile_ok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) tile_ok[j] = kRagged ? (wn * TN + j < NT32) : true;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order; speed only,
  // never correctness), so hand each XCD a CONTIGUOUS range of tiles: vertically adjacent image
  // tiles (3x3 halo rows) and tiles sharing an A panel then hit the same 4 MiB L2.
  const int tiles_n = (g.n_store + BN - 1) / BN;
  int tile_lin = blockIdx.x;
  if (g.xcd_swizzle & 1) {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    tile_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tile_lin / tiles_n;
  const int tile_n = tile_lin - tile_m * tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const int kq = tid & 7;     // which float4 of the 32-float chunk this thread moves
  const int lrow = tid >> 3;  // tile row of the first load slot; slot i adds i*(NT/8)

  // ---- operand loaders: raw buffer loads, branch-free ------------------------------------
  // Out-of-range lanes (padding halo of the convolution, rows >= M, weight rows >= N) get a
  // byte offset beyond num_records: the buffer unit returns zeros, so there is no exec-masked
  // control flow in the K loop and the loads can be scheduled among the MFMAs.
  constexpr unsigned kOob = 0x80000000u;

  // byte offset of each load slot's row start (+ this thread's float4); kOob for rows >= M.
  // conv mode: a_mask bit t = tap t of the window falls OUTSIDE the image for this output pixel.
  unsigned a_base0[A_LD], a_base1[A_LD], a_mask[A_LD];
  int a_lds[ASP ? A_LD : 1];      // ASP: LDS float offset of the slot's piece inside the A tile
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    // ASP: slot u = tid + i NT -> (tile row u / 12, 16-byte piece u % 12); else (row lrow + i NT / 8, float4 kq)
    const int au = tid + i * NT;
    const int arow = ASP ? au / 12 : lrow + i * (NT / 8);
    const int apiece = ASP ? au - arow * 12 : 0;
    if constexpr (ASP) a_lds[i] = arow * kLdsStride + apiece * 4;
    const int r = m0 + arow;
    if (CONV) {
      const int ox = r % g.Wout;
      const int t = r / g.Wout;
      const int oy = t % g.Hout;
      const int b = t / g.Hout;
      const int iy0 = oy * g.stride - g.pad;
      const int ix0 = ox * g.stride - g.pad;
      if constexpr (ASP) a_base0[i] = (unsigned)(((b * g.Hin + iy0) * g.Win + ix0) * g.Cin) * 6u + (unsigned)apiece * 16u;
      else a_base0[i] = (unsigned)(((b * g.Hin + iy0) * g.Win + ix0) * g.Cin + kq * 4) * 4u;
      a_base1[i] = 0;
      unsigned m = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const bool ok = ky < g.ksize && kx < g.ksize && (unsigned)(iy0 + ky) < (unsigned)g.Hin &&
                          (unsigned)(ix0 + kx) < (unsigned)g.Win;
          m |= (ok ? 1u : 0u) << (ky * g.ksize + kx);
        }
      a_mask[i] = (r < g.M && (!kAFrac || au < BM * 12)) ? ~m : 0xffffffffu;   // bit t set = tap t must read zeros
    } else {
      a_base0[i] = r < g.M ? (unsigned)(r * g.lda0 + kq * 4) * 4u : kOob;
      a_base1[i] = r < g.M ? (unsigned)(r * g.lda1 + kq * 4) * 4u : kOob;
      a_mask[i] = 0;
    }
  }
  constexpr bool kRingB = H3 && (NT32 % WAVES_N) != 0;   // (= kRing) the ring tile recomputes these per use: 12 registers it does not have
  unsigned b_base[kRingB ? 1 : B_LD];
  int b_lds[kRingB ? 1 : B_LD];   // LDS float offset of the slot inside the B tile
  auto ring_b = [&](int i, unsigned& base, int& lds) {
    int t = tid;
    asm volatile("" : "+v"(t));                          // recompute at every use: hoisted out of the loops these are 12 live registers
    const int u = t + i * NT;
    const int row = u / 12, c = u - row * 12;
    const int n = n0 + row;
    base = (n < g.N && u < kBRowsLoad * 12) ? (unsigned)(n * g.ldw + c * 4) * 4u : kOob;
    lds = row * kLdsStride + c * 4;
  };
#pragma unroll
  for (int i = 0; i < (kRingB ? 0 : B_LD); ++i) {
    if (H3) {   // pre-split weights: a row's chunk is 12 x 16 B ([hi x8 | mid x8 | lo x8] per 8 k)
      const int u = tid + i * NT;
      const int row = u / 12, c = u - row * 12;
      const int n = n0 + row;
      if constexpr (kBFrac) b_base[i] = (n < g.N && u < kBRowsLoad * 12) ? (unsigned)(n * g.ldw + c * 4) * 4u : kOob;
      else b_base[i] = n < g.N ? (unsigned)(n * g.ldw + c * 4) * 4u : kOob;
      b_lds[i] = row * kLdsStride + c * 4;
    } else {
      const int n = n0 + lrow + i * (NT / 8);
      b_base[i] = n < g.N ? (unsigned)(n * g.ldw + kq * 4) * 4u : kOob;
      b_lds[i] = (lrow + i * (NT / 8)) * kLdsStride + kq * 4;
    }
  }

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  auto bload = [](const __amdgpu_buffer_rsrc_t r, unsigned off) -> float4 {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  };

  // wave-uniform K-chunk cursor of the global prefetch (no per-chunk integer division).
  // Every tile walks K in the same order, so equal operands give bit-equal results in every
  // tile (exact ties in the confidence matrix behave like the reference's).  A per-workgroup
  // rotation of the chunk order (to spread L2 channels) was measured and gave nothing.
  // dense split-K (g.k_splits > 1, plain epilogue only): blockIdx.y walks chunks [k_first, k_first + nk) of K and writes its
  // partial product to C + blockIdx.y * split_stride; opp_splitk_reduce sums the partials in split order
  // (conv mode as well: the short, long-K convolutions of the 1/8-resolution stage split K over grid.y so that they fill the chip;
  // opp_splitk_epilogue sums the partials in split order and applies bias / residual / activation)
  const int k_first = g.k_splits > 1 ? (int)blockIdx.y * g.k_chunks_per_split : 0;
  const int nk = g.k_splits > 1 ? max(0, min(g.k_chunks_per_split, g.K / 32 - k_first)) : g.K / 32;
  const int taps = CONV ? g.ksize * g.ksize : 1;
  const int ngrp = CONV ? g.Cin / 32 : nk;            // rotation period: channel groups / chunks
  int cur_i = 0;                                      // chunks issued so far
  int cur_grp = 0;                                    // channel group (conv) or chunk index (dense)
  int cur_tap = 0, cur_ky = 0, cur_kx = 0;
  int cur_k0 = k_first * 32;
  unsigned cur_past = nk > 0 ? 0u : kOob;
  // conv: byte offset of the chunk's (tap, channel group) from a_base0 and the shift that brings the
  // tap's invalid-bit to bit 31.  Per lane only in the K tail (g.tail_grp > 0: the last <= 4 real
  // channels of a Cin = 32 n + 4 input, e.g. 196, are packed 8 taps to a chunk - lane quarter kq
  // carries tap 8 t + kq - instead of one mostly-zero 32-channel chunk per tap: 9 chunks -> 2).
  unsigned eff_delta = 0, eff_sh = 31;
  int tail_i = -1;                                    // >= 0: index of the current tail chunk
  const int kq_lane = tid & 7;
  auto tail_lane = [&](int t, unsigned& delta, unsigned& sh) {
    const int tp = t * 8 + kq_lane;                   // taps >= ksize^2 have their invalid-bit set
    const int ty = tp / g.ksize, tx = tp - ty * g.ksize;
    delta = (unsigned)((ty * g.Win + tx) * g.Cin + g.tail_grp * 32 - kq_lane * 4) * 4u;
    sh = (unsigned)(31 - tp);
  };
  // conv: moves the (tap, channel group) cursor one chunk forward
  auto step_cursor = [&]() {
    if (CONV) {
      if (tail_i < 0) {
        ++cur_tap;
        if (++cur_kx == g.ksize) {
          cur_kx = 0;
          ++cur_ky;
        }
        if (cur_tap == taps) {
          cur_tap = 0;
          cur_ky = 0;
          cur_kx = 0;
          ++cur_grp;
          if (g.tail_grp > 0 && cur_grp == g.tail_grp) tail_i = 0;
          else if (cur_grp == ngrp) cur_grp = 0;
        }
      } else {
        ++tail_i;
      }
      if (tail_i < 0) {
        eff_delta = (unsigned)((cur_ky * g.Win + cur_kx) * g.Cin + cur_grp * 32) * (ASP ? 6u : 4u);
        eff_sh = (unsigned)(31 - cur_tap);
      } else {
        tail_lane(tail_i & 1, eff_delta, eff_sh);     // chunks past the end are killed by cur_past
      }
    }
  };
  auto advance = [&]() {
    ++cur_i;
    cur_past = cur_i < nk ? 0u : kOob;
    if (CONV) {
      cur_k0 = (k_first + cur_i) * 32;                // the packed K order is the chunk order
      step_cursor();
    } else {
      if (++cur_grp == ngrp) cur_grp = 0;
      cur_k0 = (k_first + cur_grp) * 32;
    }
  };
  if (CONV && k_first > 0) {                          // split-K slice: walk the cursor to its first chunk (wave-uniform scalar work)
    for (int t = 0; t < k_first; ++t) step_cursor();
  }

  // global -> registers for the chunk the cursor points at; item i in [0, A_LD + B_LD) is one
  // buffer_load_dwordx4 so that the loads can be issued one at a time between MFMAs.
  // Chunks past the end of K (the pipeline always runs an even number of chunks and prefetches
  // two ahead) read out of range -> zeros: they contribute exactly 0 to the accumulators.
  auto load_item = [&](int i, float4 (&a_reg)[A_LD], float4 (&b_reg)[B_LD]) {
    if (ABL == 6 && i >= A_LD) return;   // ablation: no weight loads
    if (ABL == 7 && i < A_LD) return;    // ablation: no activation loads
    // chunks past the end of K: a zero-sized buffer (one s_cselect on the wave-uniform descriptor)
    const bool live = cur_past == 0u;
    if (i < A_LD) {
      if (CONV) {
        // pure data flow (no exec masking): a_mask holds the INVALID taps, so shifting the current
        // tap's bit to bit 31 yields the out-of-range offset directly (shift, add, and-or per load)
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A0), 0, live ? g.a0_bytes : 0, 0x00020000);
        const unsigned bad = a_mask[i] << eff_sh;
        a_reg[i] = bload(r, (bad & kOob) | (a_base0[i] + eff_delta));
      } else if (cur_k0 < g.ksplit) {
        // (kOob + small) stays out of range, so invalid rows need no select
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A0), 0, live ? g.a0_bytes : 0, 0x00020000);
        a_reg[i] = bload(r, a_base0[i] + (unsigned)cur_k0 * 4u);
      } else {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A1 ? g.A1 : g.A0), 0, (live && g.A1) ? g.a1_bytes : 0, 0x00020000);
        a_reg[i] = bload(r, a_base1[i] + (unsigned)(cur_k0 - g.ksplit) * 4u);
      }
    } else {
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W), 0, live ? g.w_bytes : 0, 0x00020000);
      if constexpr (kRingB) {
        unsigned base;
        int lds;
        ring_b(i - A_LD, base, lds);
        b_reg[i - A_LD] = bload(r, base + (unsigned)cur_k0 * 6u);
      } else {
        b_reg[i - A_LD] = bload(r, b_base[i - A_LD] + (unsigned)cur_k0 * (H3 ? 6u : 4u));
      }
    }
  };
  auto load_global = [&](float4 (&a_reg)[A_LD], float4 (&b_reg)[B_LD]) {
#pragma unroll
    for (int i = 0; i < A_LD + B_LD; ++i) load_item(i, a_reg, b_reg);
  };

  // fp16x2 mode (H2): an fp32 value x is carried as hi = fp16_rtz(x), lo = fp16(x - hi) (22 significant
  // bits).  A 32-k chunk of a row stays 128 B: four groups of 8 k, each [hi x8 | lo x8]; the A operand is
  // split here, on its way from the prefetch registers to LDS; the weights are pre-split at pack time.
  // hi = fp16_rtz(x) (one v_cvt_pkrtz per pair); lo = fp16_rne(x - hi) straight from the packed hi with
  // v_fma_mix{lo,hi}_f16 (f16 source * -1 + f32 source, rounded once to f16): 6 VALU per float4, not 14
  auto split_h2 = [](const float4 v, uint2& hi, uint2& lo) {
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.x, v.y));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.z, v.w));
    unsigned l01, l23;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(h01), "v"(v.x));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h01), "v"(v.y));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(h23), "v"(v.z));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h23), "v"(v.w));
    hi = make_uint2(h01, h23);
    lo = make_uint2(l01, l23);
  };
  // bf16x3 mode (H3): x = hi + mid + lo EXACTLY, hi = bf16_rne(x), mid = bf16_rne(x - hi), lo = bf16(x - hi - mid)
  // (|mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|, the last residual has <= 8 significant bits).  A 32-k chunk of a row is
  // 192 B: four groups of 8 k, each [hi x8 | mid x8 | lo x8].  v_cvt_pk_bf16_f32 + shift/mask + subtract per level.
  auto split_b3 = [](const float4 v, uint2& hi, uint2& mid, uint2& lo) { opp_split4_b3(v, hi, mid, lo); };
  auto store_item = [&](int i, int buf, const float4 (&a_reg)[A_LD], const float4 (&b_reg)[B_LD]) {
    if (i < A_LD) {
      if constexpr (ASP) {           // already split: the piece goes to LDS as it is
        if (!kAFrac || tid + i * NT < BM * 12) *reinterpret_cast<float4*>(As + buf * BM * kLdsStride + a_lds[i]) = a_reg[i];
        return;
      }
      float* row = As + buf * BM * kLdsStride + (lrow + i * (NT / 8)) * kLdsStride;
      if (H3) {
        uint2 hi, mid, lo;
        if (ABL == 95) {   // ablation: no split arithmetic (wrong results, timing only)
          hi = make_uint2(__float_as_uint(a_reg[i].x), __float_as_uint(a_reg[i].y));
          mid = make_uint2(__float_as_uint(a_reg[i].z), __float_as_uint(a_reg[i].w));
          lo = hi;
        } else
        split_b3(a_reg[i], hi, mid, lo);
        float* dst = row + (kq >> 1) * 12 + (kq & 1) * 2;     // group kq/2, elements (kq&1)*4 .. +3 of each part
        *reinterpret_cast<uint2*>(dst) = hi;
        *reinterpret_cast<uint2*>(dst + 4) = mid;
        *reinterpret_cast<uint2*>(dst + 8) = lo;
      } else if (H2) {
        uint2 hi, lo;
        split_h2(a_reg[i], hi, lo);
        float* dst = row + (kq >> 1) * 8 + (kq & 1) * 2;      // group kq/2, elements (kq&1)*4 .. +3
        *reinterpret_cast<uint2*>(dst) = hi;
        *reinterpret_cast<uint2*>(dst + 4) = lo;
      } else {
        *reinterpret_cast<float4*>(row + kq * 4) = a_reg[i];
      }
    } else {
      if constexpr (kRingB) {
        unsigned base;
        int lds;
        ring_b(i - A_LD, base, lds);
        if (tid + (i - A_LD) * NT < kBRowsLoad * 12) *reinterpret_cast<float4*>(Bs + buf * BN * kLdsStride + lds) = b_reg[i - A_LD];
      } else if constexpr (kBFrac) {
        if (tid + (i - A_LD) * NT < kBRowsLoad * 12) *reinterpret_cast<float4*>(Bs + buf * BN * kLdsStride + b_lds[i - A_LD]) = b_reg[i - A_LD];
      } else {
        *reinterpret_cast<float4*>(Bs + buf * BN * kLdsStride + b_lds[i - A_LD]) = b_reg[i - A_LD];
      }
    }
  };
  auto store_lds = [&](int buf, const float4 (&a_reg)[A_LD], const float4 (&b_reg)[B_LD]) {
#pragma unroll
    for (int i = 0; i < A_LD + B_LD; ++i) store_item(i, buf, a_reg, b_reg);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fp32: lane half h owns k in [16h, 16h+16) of the chunk; fp16x2: it owns k-group 2s+h of k16-step s
  // bf16x3: as fp16x2 with 12-float groups
  const int a_frag = (wm * TM * 32 + l31) * kLdsStride + half * (H3 ? 12 : H2 ? 8 : 16);
  const int b_frag = (wn * TN * 32 + l31) * kLdsStride + half * (H3 ? 12 : H2 ? 8 : 16);

  // LDS -> MFMA operand fragments of ONE k-quarter (8 k values: 4 per lane half) of a chunk
  // (fp16x2: q = 2*step + part, part 0 = the 8 hi halves, 1 = the 8 lo halves of this lane's k-group)
  auto read_frags = [&](int buf, int q, float4 (&af)[TM], float4 (&bf)[TN]) {
    // bf16x3: q = 3*step + part (part 0/1/2 = the hi/mid/lo halves of this lane's k-group)
    const int qoff = H3 ? (q / 3) * 24 + (q % 3) * 4 : H2 ? (q >> 1) * 16 + (q & 1) * 4 : q * 4;
    const float* as = As + buf * BM * kLdsStride + a_frag + qoff;
    const float* bs = Bs + buf * BN * kLdsStride + b_frag + qoff;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(as + i * 32 * kLdsStride);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      if (tile_ok[j]) bf[j] = *reinterpret_cast<const float4*>(bs + j * 32 * kLdsStride);
  };
  // TM*TN*4 MFMAs on one k-quarter.  `between(n)` runs after the n-th MFMA was issued: the
  // pipeline uses it to issue ONE buffer load or ONE LDS write in the 64-cycle shadow of each
  // v_mfma_f32_32x32x2_f32 (a VMEM instruction costs ~60 issue cycles, measured as 15 % of the
  // kernel when all eight were issued back to back at the top of the chunk).
  auto mfma_quarter = [&](const float4 (&af)[TM], const float4 (&bf)[TN], auto&& between) {
    int n = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (tile_ok[j]) {
            const float av = e == 0 ? af[i].x : e == 1 ? af[i].y : e == 2 ? af[i].z : af[i].w;
            const float bv = e == 0 ? bf[j].x : e == 1 ? bf[j].y : e == 2 ? bf[j].z : bf[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
          between(n);
          ++n;
        }
      }
    }
  };
  auto nothing = [](int) {};
  constexpr int kSlots = TM * TN * 4;                     // MFMA issue slots per quarter
  constexpr int kItems = A_LD + B_LD;                     // loads (and LDS writes) per chunk
  constexpr int kStride = kSlots / kItems > 0 ? kSlots / kItems : 1;

  // ---- software-pipelined K loop ----------------------------------------------------------
  // Per chunk c (LDS buffer c&1; 4 k-quarters q0..q3 of TM*TN*4 MFMAs each; two operand
  // fragment register sets f0/f1 alternate per quarter):
  //   issue global loads of chunk c+2 (register set c&1)     -- two chunks of latency budget
  //   read q1 | MFMA q0
  //   read q2 | MFMA q1
  //   LDS-write chunk c+1 (register set (c+1)&1, loaded during chunk c-1) into buffer (c+1)&1
  //   read q3 | MFMA q2
  //   barrier   (every read of buffer c&1 and every write of buffer (c+1)&1 is complete)
  //   read q0 of chunk c+1 | MFMA q3                          -- covers barrier skew + LDS latency
  // so the matrix pipe never waits for a global load, an LDS write or the barrier.
  float4 ga[DEPTH][A_LD], gb[DEPTH][B_LD];   // DEPTH chunks of global prefetch in registers
  // fp32 uses sets 0,1 ; fp16x2 (0,1) = hi/lo of step 0, (2,3) of step 1 ; bf16x3 (0,1,2) = hi/mid/lo of step 0, (3,4,5) of step 1
  // (kRing: four sets -- hi of the even k16-step, hi of the odd one, mid, lo -- see chunk_ring)
  float4 fa[(H3 && !kRing) ? 6 : 4][TM], fb[(H3 && !kRing) ? 6 : 4][TN];
  float4 da[A_LD], db[B_LD];   // ablation 5 only: load sink that is never consumed in the loop

  if constexpr (kRing) {
    load_global(ga[0], gb[0]);                         // chunk 0 (its weight rows go to LDS below, ahead of the reload of the single set)
  } else {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (d) advance();
      load_global(ga[d], gb[d]);
    }
  }
  if constexpr (kBRowsLoad < BN) {   // weight rows the loader never touches: zero in both buffers
    for (int u = tid; u < 2 * (BN - kBRowsLoad) * kLdsStride; u += NT) {
      const int b = u / ((BN - kBRowsLoad) * kLdsStride), o = u - b * ((BN - kBRowsLoad) * kLdsStride);
      Bs[b * BN * kLdsStride + kBRowsLoad * kLdsStride + o] = 0.f;
    }
  }
  store_lds(0, ga[0], gb[0]);
  if constexpr (kRing) {
    __builtin_amdgcn_sched_barrier(0);
    advance();
    load_global(ga[1], gb[0]);                         // chunk 1: activations into set 1, weights into the single set
  }
  __syncthreads();
  read_frags(0, 0, fa[0], fb[0]);
  if constexpr (kRing) {
    read_frags(0, 1, fa[2], fb[2]);
    read_frags(0, 2, fa[3], fb[3]);
  } else {
    if (H2 || H3) read_frags(0, 1, fa[1], fb[1]);
    if constexpr (H3) read_frags(0, 2, fa[(H3 && !kRing) ? 2 : 0], fb[(H3 && !kRing) ? 2 : 0]);
  }

  // The chunk body has no branches: the waitcnt pass can prove that the chunk c+1 registers are
  // the OLDEST loads in flight and waits with a counted vmcnt instead of draining the freshly
  // issued prefetch.  The prefetch / LDS hand-over past the last chunk is harmless
  // (out-of-range loads return zeros, the extra LDS buffer is never consumed).
  // `set` = c % DEPTH is compile-time (register arrays), the LDS buffer c & 1 is run-time.
  auto chunk = [&](auto set, int lb) {
    constexpr int P = decltype(set)::value;
    constexpr int PN = (P + 1) % DEPTH;              // register set holding chunk c+1
    constexpr int A2 = ABL == 5 ? 1 : (ABL >= 6 ? 0 : ABL);   // ablation 5 behaves like 1 apart from the sink loads
    const int B0 = A2 >= 2 ? 0 : lb;
    const int B1 = A2 >= 2 ? 0 : lb ^ 1;
    if (ABL < 1 || ABL >= 5) advance();
    if (A2 < 3) read_frags(B0, 1, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_quarter(fa[0], fb[0], [&](int n) {          // q0 + prefetch of chunk c+DEPTH, one load per slot
      if ((ABL < 1 || ABL >= 5) && n % kStride == 0 && n / kStride < kItems) {
        if (ABL == 5) load_item(n / kStride, da, db);
        else load_item(n / kStride, ga[P], gb[P]);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    if (ABL < 1 || ABL >= 6) {                        // loads that did not fit into the slots
#pragma unroll
      for (int i = kSlots / kStride; i < kItems; ++i) load_item(i, ga[P], gb[P]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (A2 < 3) read_frags(B0, 2, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_quarter(fa[1], fb[1], nothing);
    __builtin_amdgcn_sched_barrier(0);
    if (A2 < 3) read_frags(B0, 3, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_quarter(fa[0], fb[0], [&](int n) {          // q2 + LDS hand-over of chunk c+1
      if (A2 < 2 && n % kStride == 0 && n / kStride < kItems) {
        store_item(n / kStride, B1, ga[PN], gb[PN]);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    if (A2 < 2) {
#pragma unroll
      for (int i = kSlots / kStride; i < kItems; ++i) store_item(i, B1, ga[PN], gb[PN]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (A2 < 2) __syncthreads();
    if (A2 < 3) read_frags(B1, 0, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_quarter(fa[1], fb[1], nothing);
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- fp16x2 variant of the chunk: 2 k16-steps of v_mfma_f32_32x32x16_f16, three products per
  // tile pair and step (hi*lo + lo*hi + hi*hi, fp32 accumulate): 3/16 of the fp32 MFMA cycles ------
  auto as_h8 = [](const float4& v) { return *reinterpret_cast<const f16x8*>(&v); };
  auto mfma_h2 = [&](const float4 (&ah)[TM], const float4 (&al)[TM], const float4 (&bh)[TN], const float4 (&bl)[TN],
                     bool cross, bool hh, auto&& between) {
    int n = 0;
    if (cross) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (tile_ok[j])
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(pr ? al[i] : ah[i]), as_h8(pr ? bh[j] : bl[j]), acc[i][j], 0, 0, 0);
            between(n);
            ++n;
          }
    }
    if (hh) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (tile_ok[j]) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[i]), as_h8(bh[j]), acc[i][j], 0, 0, 0);
          between(n);
          ++n;
        }
    }
  };
  constexpr int kSlotsL = 3 * TM * TN;                    // step 0: all three products
  constexpr int kStrideL = kSlotsL / kItems > 0 ? kSlotsL / kItems : 1;
  constexpr int kSlotsS = 2 * TM * TN;                    // step 1 cross terms
  constexpr int kStrideS = kSlotsS / kItems > 0 ? kSlotsS / kItems : 1;
  auto chunk_h2 = [&](auto set, int lb) {
    constexpr int P = decltype(set)::value;
    constexpr int PN = (P + 1) % DEPTH;
    const int B0 = lb, B1 = lb ^ 1;
    advance();
    read_frags(B0, 2, fa[2], fb[2]);
    read_frags(B0, 3, fa[3], fb[3]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_h2(fa[0], fa[1], fb[0], fb[1], true, true, [&](int n) {      // step 0 + prefetch of chunk c+DEPTH
      if (n % kStrideL == 0 && n / kStrideL < kItems) {
        if (!kNoGload) load_item(n / kStrideL, ga[P], gb[P]);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
#pragma unroll
    for (int i = kSlotsL / kStrideL; i < kItems; ++i)
      if (!kNoGload) load_item(i, ga[P], gb[P]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_h2(fa[2], fa[3], fb[2], fb[3], true, false, [&](int n) {     // step 1 cross terms + LDS hand-over
      if (n % kStrideS == 0 && n / kStrideS < kItems) {
        if (!kNoLdsSt) store_item(n / kStrideS, B1, ga[PN], gb[PN]);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
#pragma unroll
    for (int i = kSlotsS / kStrideS; i < kItems; ++i)
      if (!kNoLdsSt) store_item(i, B1, ga[PN], gb[PN]);
    __builtin_amdgcn_sched_barrier(0);
    if (!kNoBar) __syncthreads();
    read_frags(B1, 0, fa[0], fb[0]);
    read_frags(B1, 1, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_h2(fa[2], fa[3], fb[2], fb[3], false, true, nothing);        // step 1 hi*hi covers barrier + LDS latency
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- bf16x3 variant of the chunk: 2 k16-steps of v_mfma_f32_32x32x16_bf16, six products per tile pair and
  // step (lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi, fp32 accumulate; the dropped mid*lo, lo*mid, lo*lo
  // terms are <= 2^-26 |a||b|): 6/16 of the fp32 MFMA cycles, operands carried exactly ------
  auto as_b8 = [](const float4& v) { return *reinterpret_cast<const bf16x8*>(&v); };
  // products [p0, p1) of step `st` (fragment sets 3*st + {0 hi, 1 mid, 2 lo}), smallest terms first
  constexpr int kH6 = (H3 && !kRing) ? 3 : 0;        // offset of the second step's fragment sets (six-set scheme)
  auto mfma_b3 = [&](int st, int p0, int p1, auto&& between) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    int n = 0;
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {
      if (pr < p0 || pr >= p1) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (tile_ok[j])
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b8(fa[kH6 * st + PA[pr]][i]), as_b8(fb[kH6 * st + PB[pr]][j]), acc[i][j], 0, 0, 0);
          between(n);
          ++n;
        }
    }
  };
  constexpr int kSlots3L = 6 * TM * TN;                   // step 0: all six products
  constexpr int kStride3L = kSlots3L / kItems > 0 ? kSlots3L / kItems : 1;
  constexpr int kSlots3S = 4 * TM * TN;                   // step 1: first four products
  constexpr int kStride3S = kSlots3S / kItems > 0 ? kSlots3S / kItems : 1;
  auto chunk_h3 = [&](auto set, int lb) {
    constexpr int P = decltype(set)::value;
    constexpr int PN = (P + 1) % DEPTH;
    const int B0 = lb, B1 = lb ^ 1;
    advance();
    if (!kNoFrag) {
      read_frags(B0, 3, fa[kH6], fb[kH6]);
      read_frags(B0, 4, fa[kH6 ? 4 : 0], fb[kH6 ? 4 : 0]);
      read_frags(B0, 5, fa[kH6 ? 5 : 0], fb[kH6 ? 5 : 0]);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_b3(0, 0, 6, [&](int n) {                    // step 0 + prefetch of chunk c+DEPTH, one load per slot
      if (n % kStride3L == 0 && n / kStride3L < kItems) {
        if (!kNoGload) load_item(n / kStride3L, ga[P], gb[P]);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
#pragma unroll
    for (int i = kSlots3L / kStride3L; i < kItems; ++i)
      if (!kNoGload) load_item(i, ga[P], gb[P]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_b3(1, 0, 4, [&](int n) {                    // step 1, first four products + LDS hand-over of chunk c+1
      if (n % kStride3S == 0 && n / kStride3S < kItems) {
        if (!kNoLdsSt) store_item(n / kStride3S, B1, ga[PN], gb[PN]);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
#pragma unroll
    for (int i = kSlots3S / kStride3S; i < kItems; ++i)
      if (!kNoLdsSt) store_item(i, B1, ga[PN], gb[PN]);
    __builtin_amdgcn_sched_barrier(0);
    if (!kNoBar) __syncthreads();
    if (!kNoFrag) {
      read_frags(B1, 0, fa[0], fb[0]);
      read_frags(B1, 1, fa[1], fb[1]);
      read_frags(B1, 2, fa[kH6 ? 2 : 0], fb[kH6 ? 2 : 0]);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_b3(1, 4, 6, nothing);                       // step 1 hi*mid + hi*hi cover the barrier + LDS latency
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- bf16x3 chunk on FOUR fragment sets (kRing): set 0 = hi of k16-step 0, set 1 = hi of step 1, set 2 = mid, set 3 = lo of the
  // step being multiplied.  The product order of chunk_h3 (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi per step: the accumulation
  // sequence every tile shares) frees lo after the second product and mid after the fifth, so the NEXT step's lo / mid are read into
  // the same registers there, and the next step's hi into the other hi set:
  //   step 0:  read hi1 | P0 P1 | read lo1 | P2 P3 P4 | read mid1 | P5          (+ the global prefetch of chunk c+2, one load per slot)
  //   step 1:  P0 P1 P2 (+ the LDS hand-over of chunk c+1) | barrier | read hi0' lo0' of chunk c+1 | P3 P4 | read mid0' | P5
  // Every read is issued >= 12 MFMAs ahead of its first use; the barrier sits under the last three products as in chunk_h3.
  // The number of live column sub-tiles of the wave (4 for wn = 0, 3 for wn = 1) is a COMPILE-TIME parameter of the loop body: the K loop exists
  // twice and the wave picks its copy once, ahead of the loop.  (First version: one body with the wave-uniform `tile_ok[j]` test around every
  // MFMA and fragment read -- 3 x the scalar instructions and 7 x the s_nop / branch class of the 128 x 256 kernel in the counters,
  // profiles/r05_pmc_conv_224_vs_256.txt, and 14 % more wave cycles for 12.5 % fewer MFMAs.)
  auto read_frags_n = [&](int buf, int q, float4 (&af)[TM], float4 (&bf)[TN], auto tnw_c) {
    constexpr int TNW = decltype(tnw_c)::value;
    const int qoff = (q / 3) * 24 + (q % 3) * 4;
    const float* as = As + buf * BM * kLdsStride + a_frag + qoff;
    const float* bs = Bs + buf * BN * kLdsStride + b_frag + qoff;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(as + i * 32 * kLdsStride);
#pragma unroll
    for (int j = 0; j < TNW; ++j) bf[j] = *reinterpret_cast<const float4*>(bs + j * 32 * kLdsStride);
  };
  auto mfma_ring = [&](int hs, int p0, int p1, int n0, auto tnw_c, auto&& between) {
    constexpr int TNW = decltype(tnw_c)::value;
    int n = n0;
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {
      if (pr < p0 || pr >= p1) continue;
      // A part of product pr: lo hi mid mid hi hi ; B part: hi lo mid hi mid hi  (PA / PB of mfma_b3) -> ring sets: hi = hs, mid = 2, lo = 3
      const int sa = (pr == 0) ? 3 : (pr == 2 || pr == 3) ? 2 : hs;
      const int sb = (pr == 1) ? 3 : (pr == 2 || pr == 4) ? 2 : hs;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b8(fa[sa & 3][i]), as_b8(fb[sb & 3][j]), acc[i][j], 0, 0, 0);
          between(n);
          ++n;
        }
    }
  };
  // Global prefetch of the ring tile: the activation rows of chunk c+2 are loaded during step 0 of chunk c (two register sets, as everywhere);
  // the WEIGHT rows -- L2-resident, a few hundred cycles away -- have ONE register set: loaded under the last three products of chunk c
  // (right after the barrier behind which the set's previous content went to LDS), handed to LDS under the first three products of step 1 of
  // chunk c+1, a whole k16-step later.  20 registers less than two sets; the 32 x 128 wave tile does not fit without them.
  auto chunk_ring = [&](auto set, int lb, auto tnw_c) {
    constexpr int P = decltype(set)::value;
    constexpr int PN = (P + 1) % DEPTH;
    constexpr int TNW = decltype(tnw_c)::value;
    constexpr int kSlotsR0 = 6 * TM * TNW;                  // step 0: all six products carry the activation prefetch
    constexpr int kStrideR0 = kSlotsR0 / A_LD > 0 ? kSlotsR0 / A_LD : 1;
    constexpr int kSlotsR1 = 3 * TM * TNW;                  // step 1: the first three carry the LDS hand-over, the last three the weight prefetch
    constexpr int kStrideR1 = kSlotsR1 / kItems > 0 ? kSlotsR1 / kItems : 1;
    constexpr int kStrideRB = kSlotsR1 / B_LD > 0 ? kSlotsR1 / B_LD : 1;
    const int B0 = lb, B1 = lb ^ 1;
    auto pre = [&](int n) {
      if (n % kStrideR0 == 0 && n / kStrideR0 < A_LD) {
        load_item(n / kStrideR0, ga[P], gb[0]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto pre_b = [&](int n) {
      if (n % kStrideRB == 0 && n / kStrideRB < B_LD) {
        load_item(A_LD + n / kStrideRB, ga[P], gb[0]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    advance();
    read_frags_n(B0, 3, fa[1], fb[1], tnw_c);         // hi of step 1
    __builtin_amdgcn_sched_barrier(0);
    mfma_ring(0, 0, 2, 0, tnw_c, pre);
    __builtin_amdgcn_sched_barrier(0);
    read_frags_n(B0, 5, fa[3], fb[3], tnw_c);         // lo of step 1 over the (dead) lo of step 0
    __builtin_amdgcn_sched_barrier(0);
    mfma_ring(0, 2, 5, 2 * TM * TNW, tnw_c, pre);
    __builtin_amdgcn_sched_barrier(0);
    read_frags_n(B0, 4, fa[2], fb[2], tnw_c);         // mid of step 1 over the (dead) mid of step 0
    __builtin_amdgcn_sched_barrier(0);
    mfma_ring(0, 5, 6, 5 * TM * TNW, tnw_c, pre);
#pragma unroll
    for (int i = kSlotsR0 / kStrideR0; i < A_LD; ++i) load_item(i, ga[P], gb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_ring(1, 0, 3, 0, tnw_c, [&](int n) {         // step 1, first three products + LDS hand-over of chunk c+1
      if (n % kStrideR1 == 0 && n / kStrideR1 < kItems) {
        store_item(n / kStrideR1, B1, ga[PN], gb[0]);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
#pragma unroll
    for (int i = kSlotsR1 / kStrideR1; i < kItems; ++i) store_item(i, B1, ga[PN], gb[0]);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_frags_n(B1, 0, fa[0], fb[0], tnw_c);         // chunk c+1: hi of step 0 (set 0 is dead since step 0's last product)
    read_frags_n(B1, 2, fa[3], fb[3], tnw_c);         //            lo of step 0 (lo of step 1 is dead after its second product)
    __builtin_amdgcn_sched_barrier(0);
    mfma_ring(1, 3, 5, 0, tnw_c, pre_b);              // + the weight rows of chunk c+2 into the (just stored) single set
    __builtin_amdgcn_sched_barrier(0);
    read_frags_n(B1, 1, fa[2], fb[2], tnw_c);         //            mid of step 0 over the (dead) mid of step 1
    __builtin_amdgcn_sched_barrier(0);
    mfma_ring(1, 5, 6, 2 * TM * TNW, tnw_c, pre_b);
#pragma unroll
    for (int i = kSlotsR1 / kStrideRB; i < B_LD; ++i) load_item(A_LD + i, ga[P], gb[0]);
    __builtin_amdgcn_sched_barrier(0);
  };

  if (ABL == 9 || ABL > 90) ts1 = __builtin_readcyclecounter();
  // 8-wave tiles: the second-dispatched half of the workgroup (waves 4-7) is the arbitration loser on every SIMD it shares; one
  // static s_setprio for that half evens the pair out (MI355X_MICROARCH.md, "two waves per SIMD", item 4).  The guard must be
  // provably wave-uniform: s_setprio ignores EXEC.
  if (NT == 512 && (g.xcd_swizzle & 2) && __builtin_amdgcn_readfirstlane(tid) >= 256) __builtin_amdgcn_s_setprio(1);
  if constexpr (kRing) {
    constexpr int kFull = TN, kShort = NT32 - (WAVES_N - 1) * TN;     // live sub-tiles of the waves with wn < WAVES_N - 1 / of the last ones
    static_assert(DEPTH == 2 && kShort >= 1 && kShort <= TN, "ring tile");
    if (wn != WAVES_N - 1) {                                           // wave-uniform; both loops cross the same barriers
      for (int c = 0; c < nk; c += 2) {
        chunk_ring(std::integral_constant<int, 0>{}, c & 1, std::integral_constant<int, kFull>{});
        chunk_ring(std::integral_constant<int, 1>{}, (c + 1) & 1, std::integral_constant<int, kFull>{});
      }
    } else {
      for (int c = 0; c < nk; c += 2) {
        chunk_ring(std::integral_constant<int, 0>{}, c & 1, std::integral_constant<int, kShort>{});
        chunk_ring(std::integral_constant<int, 1>{}, (c + 1) & 1, std::integral_constant<int, kShort>{});
      }
    }
  }
  // nk rounded up to a multiple of DEPTH: the extra chunks are all-zero ones
  for (int c = 0; c < (kRing ? 0 : nk); c += DEPTH) {
    if constexpr (kRing) {
      // (handled by the two loops above)
    } else if constexpr (H3) {
      chunk_h3(std::integral_constant<int, 0>{}, c & 1);
      if (DEPTH > 1) chunk_h3(std::integral_constant<int, 1 % DEPTH>{}, (c + 1) & 1);
      if (DEPTH > 2) chunk_h3(std::integral_constant<int, 2 % DEPTH>{}, (c + 2) & 1);
      if (DEPTH > 3) chunk_h3(std::integral_constant<int, 3 % DEPTH>{}, (c + 3) & 1);
    } else if constexpr (H2) {
      chunk_h2(std::integral_constant<int, 0>{}, c & 1);
      if (DEPTH > 1) chunk_h2(std::integral_constant<int, 1 % DEPTH>{}, (c + 1) & 1);
      if (DEPTH > 2) chunk_h2(std::integral_constant<int, 2 % DEPTH>{}, (c + 2) & 1);
      if (DEPTH > 3) chunk_h2(std::integral_constant<int, 3 % DEPTH>{}, (c + 3) & 1);
    } else {
      chunk(std::integral_constant<int, 0>{}, c & 1);
      if (DEPTH > 1) chunk(std::integral_constant<int, 1 % DEPTH>{}, (c + 1) & 1);
      if (DEPTH > 2) chunk(std::integral_constant<int, 2 % DEPTH>{}, (c + 2) & 1);
      if (DEPTH > 3) chunk(std::integral_constant<int, 3 % DEPTH>{}, (c + 3) & 1);
    }
  }
  if (ABL == 9 || ABL > 90) ts2 = __builtin_readcyclecounter();
  if (ABL == 5) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) asm volatile("" ::"v"(da[i].x), "v"(da[i].w));
#pragma unroll
    for (int i = 0; i < B_LD; ++i) asm volatile("" ::"v"(db[i].x), "v"(db[i].w));
  }

  // ---- epilogue -----------------------------------------------------------------------
  // The MFMA accumulator layout puts 32 consecutive COLUMNS of one row on 32 lanes (4 B per
  // lane per store).  Stage the tile through the (now idle) LDS operand buffers and write it
  // out row-contiguously with 16 B per lane: full 128 B lines for the stores, the residual and
  // the bias, and the per-row work (bilinear taps, value scaling) is done once per 4 outputs.
  constexpr int kJPmax = TN < 4 ? TN : 4;
  constexpr int kOperandFloats = 2 * (BM + BN) * kLdsStride;
  // n-subtiles per wave staged per pass: as many as fit the operand buffers
  constexpr int JP = (BM * (WAVES_N * kJPmax * 32 + 4) <= kOperandFloats) ? kJPmax
                     : (BM * (WAVES_N * 2 * 32 + 4) <= kOperandFloats && kJPmax >= 2) ? 2 : 1;
  constexpr int NPASS = (TN + JP - 1) / JP;
  constexpr int WP = WAVES_N * JP * 32;                    // staged columns per pass
  constexpr int CS = WP + 4;                               // LDS row stride (floats)
  static_assert((size_t)BM * CS * 4 <= (size_t)2 * (BM + BN) * kLdsStride * 4, "C tile must fit the operand LDS");
  float* Cs = smem;
  const bool scale_on = (g.out_mul != 1.f) || (g.out_div != 1.f);
  const bool vec_ok = g.vec_epilogue != 0;
  // fp16x2: the weight matrix was pre-scaled by a power of two (its fp16 lo halves stay normal); undo, exactly
  const float h2_inv = (H2 && g.h2_inv != nullptr) ? *g.h2_inv : 1.f;
  if (H2 || scale_on) {   // output scaling once, in place (the statistics below reuse the scaled values)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float cv0 = H2 ? acc[i][j][r] * h2_inv : acc[i][j][r];
          acc[i][j][r] = scale_on ? (cv0 * g.out_mul) / g.out_div : cv0;
        }
  }
  if (g.col_mask != nullptr) {   // masked image cells: sim += -1e9 (coarse_matching.py:108-114); lane = column
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + l31;
      const float add = (col < g.n_store && g.col_mask[col] == 0.f) ? -1e9f : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += add;
    }
  }
  if constexpr (H2) {
    if (g.nonfinite != nullptr) {   // range guard: |x| beyond the fp16 range shows up as inf / NaN accumulators
      bool bad = false;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) bad |= !(fabsf(acc[i][j][r]) <= 3.4028235e38f);
      if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicOr(g.nonfinite, 1);
    }
  }
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    __syncthreads();   // operand buffers (or the previous pass) are no longer read
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jj = 0; jj < JP; ++jj) {
        const int j = ps * JP + jj;
        if (j < TN && tile_ok[j]) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lr = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            Cs[lr * CS + wn * JP * 32 + jj * 32 + l31] = acc[i][j][r];
          }
        }
      }
    __syncthreads();
    if constexpr (NPASS == 1 && !kRagged && (NT % BM) == 0 && kOperandFloats - BM * CS >= 2 * (NT / BM) * BM + 2 * WAVES_M * WP) {
      if (g.stat_rowmax != nullptr) {
        // fused dual-softmax statistics (coarse_matching.py:115): partial (max, sum exp) of this tile per
        // row (over its columns) and per column (over its rows); merged by tiny kernels.  Every partial is a
        // fixed-order function of the row's / column's values only, so duplicated rows or columns get
        // bit-equal statistics wherever they sit (exact ties then resolve like the reference's).
        constexpr int kParts = NT / BM;                 // threads per row
        constexpr int kQ = WP / 4 / kParts;             // float4 per thread
        float* sc_r = Cs + BM * CS;                     // [kParts][BM] scratch behind the staged tile
        float* sc_c = sc_r + 2 * kParts * BM;           // [WAVES_M][WP]
        const int ncols = min(WP, g.n_store - n0);
        const int nrows = min(BM, g.M - m0);
        {  // rows: thread = (row, part); 16 lanes of a ds_read_b128 group read 16 rows -> conflict-free (CS = 4 mod 64)
          const int lr = tid % BM, part = tid / BM;
          const float* rowp = Cs + lr * CS + part * kQ * 4;
          float mx = -INFINITY;
#pragma unroll 4
          for (int q = 0; q < kQ; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(rowp + q * 4);
            const int c = (part * kQ + q) * 4;
            mx = fmaxf(mx, c + 0 < ncols ? v.x : -INFINITY);
            mx = fmaxf(mx, c + 1 < ncols ? v.y : -INFINITY);
            mx = fmaxf(mx, c + 2 < ncols ? v.z : -INFINITY);
            mx = fmaxf(mx, c + 3 < ncols ? v.w : -INFINITY);
          }
          sc_r[part * BM + lr] = mx;
          __syncthreads();
          mx = sc_r[lr];
#pragma unroll
          for (int pp = 1; pp < kParts; ++pp) mx = fmaxf(mx, sc_r[pp * BM + lr]);
          float sm = 0.f;
#pragma unroll 4
          for (int q = 0; q < kQ; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(rowp + q * 4);
            const int c = (part * kQ + q) * 4;
            sm += c + 0 < ncols ? __expf(v.x - mx) : 0.f;
            sm += c + 1 < ncols ? __expf(v.y - mx) : 0.f;
            sm += c + 2 < ncols ? __expf(v.z - mx) : 0.f;
            sm += c + 3 < ncols ? __expf(v.w - mx) : 0.f;
          }
          sc_r[(kParts + part) * BM + lr] = sm;
          __syncthreads();
          if (part == 0 && lr < nrows) {
            float tot = sc_r[kParts * BM + lr];
#pragma unroll
            for (int pp = 1; pp < kParts; ++pp) tot += sc_r[(kParts + pp) * BM + lr];
            g.stat_rowmax[(size_t)(m0 + lr) * tiles_n + tile_n] = mx;
            g.stat_rowsum[(size_t)(m0 + lr) * tiles_n + tile_n] = tot;
          }
        }
        {  // columns: straight from the accumulators (lane = column), same scaled value as stored
          auto sval = [&](int i, int j, int r) -> float { return acc[i][j][r]; };
          auto rvalid = [&](int i, int r) -> bool {
            return wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < nrows;
          };
          float cmx[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int r = 0; r < 16; ++r) m = fmaxf(m, rvalid(i, r) ? sval(i, j, r) : -INFINITY);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            if (half == 0) sc_c[wm * WP + wn * TN * 32 + j * 32 + l31] = m;
          }
          __syncthreads();
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float m = sc_c[wn * TN * 32 + j * 32 + l31];
#pragma unroll
            for (int w = 1; w < WAVES_M; ++w) m = fmaxf(m, sc_c[w * WP + wn * TN * 32 + j * 32 + l31]);
            cmx[j] = m;
          }
          __syncthreads();
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int r = 0; r < 16; ++r) sm += rvalid(i, r) ? __expf(sval(i, j, r) - cmx[j]) : 0.f;
            sm += __shfl_xor(sm, 32, 64);
            if (half == 0) sc_c[wm * WP + wn * TN * 32 + j * 32 + l31] = sm;
          }
          __syncthreads();
          if (wm == 0 && half == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const int c = wn * TN * 32 + j * 32 + l31;
              if (c < ncols) {
                float tot = sc_c[c];
#pragma unroll
                for (int w = 1; w < WAVES_M; ++w) tot += sc_c[w * WP + c];
                g.stat_colmax[(size_t)tile_m * g.n_store + n0 + c] = cmx[j];
                g.stat_colsum[(size_t)tile_m * g.n_store + n0 + c] = tot;
              }
            }
          }
        }
      }
    }
    if constexpr (NPASS == 1 && !kRagged && (WP == 256 || WP == 128)) {
      if (g.ln_gamma != nullptr) {
        // LayerNorm of the finished rows (the tile spans the row; launcher checks n_store == BN): one wave per
        // row, four rows in lock step, same arithmetic as layernorm_kernel (attention.hip)
        constexpr int VPT = WP / 64, NW = WAVES_M * WAVES_N, RPW = 4;
        typedef float vec_t __attribute__((ext_vector_type(VPT)));
        static_assert(BM % (NW * RPW) == 0 || BM / NW < RPW, "rows per wave");
        const vec_t gmv = *reinterpret_cast<const vec_t*>(g.ln_gamma + lane * VPT);
        const vec_t btv = *reinterpret_cast<const vec_t*>(g.ln_beta + lane * VPT);
        constexpr int ROWS_W = BM / NW;                      // rows per wave
        constexpr int GRP = ROWS_W < RPW ? ROWS_W : RPW;
#pragma unroll
        for (int r0 = 0; r0 < ROWS_W; r0 += GRP) {
          float v[GRP][VPT], rv[GRP][VPT], sm[GRP], mean[GRP], rstd[GRP];
#pragma unroll
          for (int r = 0; r < GRP; ++r) {
            const int lr = wave * ROWS_W + r0 + r;
            const vec_t t = *reinterpret_cast<const vec_t*>(Cs + lr * CS + lane * VPT);
            const bool ok = m0 + lr < g.M;
            vec_t rt;
#pragma unroll
            for (int i = 0; i < VPT; ++i) rt[i] = 0.f;
            if (ok && g.ln_res) rt = *reinterpret_cast<const vec_t*>(g.ln_res + (size_t)(m0 + lr) * g.ln_ldres + lane * VPT);
            sm[r] = 0.f;
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
              v[r][i] = t[i];
              rv[r][i] = rt[i];
              sm[r] += t[i];
            }
          }
#pragma unroll
          for (int r = 0; r < GRP; ++r) sm[r] = opp_wave_sum_dpp(sm[r]);
#pragma unroll
          for (int r = 0; r < GRP; ++r) {
            mean[r] = sm[r] / (float)WP;
            sm[r] = 0.f;
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
              const float d = v[r][i] - mean[r];
              sm[r] = opp_ln_sq_acc(d, sm[r]);
            }
          }
#pragma unroll
          for (int r = 0; r < GRP; ++r) sm[r] = opp_wave_sum_dpp(sm[r]);
#pragma unroll
          for (int r = 0; r < GRP; ++r) rstd[r] = 1.0f / sqrtf(sm[r] / (float)WP + g.ln_eps);
#pragma unroll
          for (int r = 0; r < GRP; ++r) {
            const int lr = wave * ROWS_W + r0 + r;
            if (m0 + lr < g.M) {
              vec_t o;
#pragma unroll
              for (int i = 0; i < VPT; ++i) {
                float y = opp_ln_affine(v[r][i], mean[r], rstd[r], gmv[i], btv[i]);
                if (g.ln_res) y = rv[r][i] + y;
                o[i] = y;
              }
              *reinterpret_cast<vec_t*>(g.C + (size_t)(m0 + lr) * g.ldc + lane * VPT) = o;
            }
          }
        }
        continue;
      }
    }
    for (int u = tid; u < BM * (WP / 4); u += NT) {
      const int lr = u / (WP / 4);
      const int lc = (u - lr * (WP / 4)) * 4;
      const int wn2 = lc / (JP * 32);
      const int j = ps * JP + (lc - wn2 * JP * 32) / 32;
      const int row = m0 + lr;
      const int col = n0 + wn2 * TN * 32 + j * 32 + (lc & 31);
      if (j >= TN || (kRagged && wn2 * TN + j >= NT32) || row >= g.M || col >= g.n_store) continue;
      const float4 cv = *reinterpret_cast<const float4*>(Cs + lr * CS + lc);
      float v[4] = {cv.x, cv.y, cv.z, cv.w};
      const int nval = g.n_store - col < 4 ? g.n_store - col : 4;   // < 4 only on the scalar path
      if (g.bias) {
        if (vec_ok) {
          const float4 bv = *reinterpret_cast<const float4*>(g.bias + col);
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        } else {
          for (int e = 0; e < nval; ++e) v[e] += g.bias[col + e];
        }
      }
      if (g.res_mode == OPP_RES_DIRECT) {
        const float* rp = g.R + (size_t)row * g.ldr + col;
        if (vec_ok) {
          const float4 rv = *reinterpret_cast<const float4*>(rp);
          v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        } else {
          for (int e = 0; e < nval; ++e) v[e] += rp[e];
        }
      } else if (g.res_mode == OPP_RES_BILINEAR2X) {
        // bilinear x2 (align_corners=True) taps of the half-resolution residual (resnet.py:151,155).
        // No FMA contraction in this block: the compiler fuses differently in different tile instantiations, and
        // every tile shape must produce the same bits (tools/tile_invariance_check.py).
#pragma clang fp contract(off)
        const int ox = row % g.Wout;
        const int t = row / g.Wout;
        const int oy = t % g.Hout;
        const int b = t / g.Hout;
        const float sy = g.res_sy * (float)oy;
        const float sx = g.res_sx * (float)ox;
        int y0 = (int)sy;
        if (y0 > g.Hr - 1) y0 = g.Hr - 1;
        int x0 = (int)sx;
        if (x0 > g.Wr - 1) x0 = g.Wr - 1;
        const int y1 = y0 + (y0 < g.Hr - 1 ? 1 : 0);
        const int x1 = x0 + (x0 < g.Wr - 1 ? 1 : 0);
        const float wy1 = fminf(fmaxf(sy - (float)y0, 0.f), 1.f);
        const float wx1 = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
        const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
        const int pb = b * g.Hr * g.Wr;
        const float* r00 = g.R + (size_t)(pb + y0 * g.Wr + x0) * g.ldr + col;
        const float* r01 = g.R + (size_t)(pb + y0 * g.Wr + x1) * g.ldr + col;
        const float* r10 = g.R + (size_t)(pb + y1 * g.Wr + x0) * g.ldr + col;
        const float* r11 = g.R + (size_t)(pb + y1 * g.Wr + x1) * g.ldr + col;
        float a00[4], a01[4], a10[4], a11[4];
        if (vec_ok) {
          const float4 t0 = *reinterpret_cast<const float4*>(r00), t1 = *reinterpret_cast<const float4*>(r01);
          const float4 t2 = *reinterpret_cast<const float4*>(r10), t3 = *reinterpret_cast<const float4*>(r11);
          a00[0] = t0.x; a00[1] = t0.y; a00[2] = t0.z; a00[3] = t0.w;
          a01[0] = t1.x; a01[1] = t1.y; a01[2] = t1.z; a01[3] = t1.w;
          a10[0] = t2.x; a10[1] = t2.y; a10[2] = t2.z; a10[3] = t2.w;
          a11[0] = t3.x; a11[1] = t3.y; a11[2] = t3.z; a11[3] = t3.w;
        } else {
          for (int e = 0; e < 4; ++e) {
            const bool ok = e < nval;
            a00[e] = ok ? r00[e] : 0.f; a01[e] = ok ? r01[e] : 0.f;
            a10[e] = ok ? r10[e] : 0.f; a11[e] = ok ? r11[e] : 0.f;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float top = wx0 * a00[e] + wx1 * a01[e];
          const float bot = wx0 * a10[e] + wx1 * a11[e];
          v[e] += wy0 * top + wy1 * bot;
        }
      }
      if (g.act == OPP_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];   // NaN-propagating like torch.relu
      } else if (g.act == OPP_ACT_LEAKY) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
      } else if (g.act == OPP_ACT_QKV) {
        if (col < g.qk_cols) {   // elu(x) + 1 = x + 1 (x > 0) | exp(x) (x <= 0), linear_attention.py:10-11
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] + 1.f : __expf(v[e]);
        } else {                 // values / v_length, linear_attention.py:55-56
          const float vdiv = row < g.split_row ? g.s0 : g.s1;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] / vdiv;
        }
        if (g.row_mask != nullptr && row < g.row_mask_rows) {   // padded image tokens: Q, K, V rows -> 0 (linear_attention.py:49-53)
          const float rm = g.row_mask[row];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= rm;
        }
      }
      if constexpr (H3) {    // the same values once more, pre-split for the convolutions that consume this map (launcher: vec_ok, no K slices)
        if (g.C3 != nullptr) opp_store_split4(g.C3, (size_t)row * g.ld3, col, make_float4(v[0], v[1], v[2], v[3]));
      }
      if (g.C == nullptr) continue;       // split output only
      float* cp = g.C + (size_t)row * g.ldc + col + (g.k_splits > 1 ? (size_t)blockIdx.y * g.split_stride : 0);
      if (vec_ok) {
        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int e = 0; e < nval; ++e) cp[e] = v[e];
      }
    }
  }
  if ((ABL == 9 || ABL > 90) && g.dbg_ts != nullptr && lane == 0) {
    unsigned long long* o = g.dbg_ts + ((size_t)blockIdx.x * (WAVES_M * WAVES_N) + wave) * 4;
    o[0] = ts0;
    o[1] = ts1;
    o[2] = ts2;
    o[3] = __builtin_readcyclecounter();
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool CONV, int ABL = 0, int DEPTH = 2, int PREC = OPP_PREC_FP32>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void opp_gemm_kernel(const OppGemm g) {
  opp_gemm_body<BM, BN, WAVES_M, WAVES_N, CONV, ABL, DEPTH, PREC, false>(g);
}
template <int BM, int BN, int WAVES_M, int WAVES_N, int ABL = 0, int DEPTH = 2, int PREC = OPP_PREC_BF16X3>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void opp_gemm_asp_kernel(const OppGemm g) {
  opp_gemm_body<BM, BN, WAVES_M, WAVES_N, true, ABL, DEPTH, PREC, true>(g);
}

// out = act(sum_s part[s] + bias + R): the split-K slices of a convolution summed in slice order (deterministic), then the epilogue
// the unsplit kernel fuses (folded-BN bias, same-shape residual, ReLU / LeakyReLU); float4 granularity over [M][ld]
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float4* __restrict__ part, int splits, size_t stride4, size_t n4, int ld4,
                                                              const float4* __restrict__ bias, const float4* __restrict__ R, int act,
                                                              float4* __restrict__ out, void* __restrict__ out3, int ld3) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = part[i];
    for (int s = 1; s < splits; ++s) {
      const float4 v = part[(size_t)s * stride4 + i];
      a.x += v.x;
      a.y += v.y;
      a.z += v.z;
      a.w += v.w;
    }
    if (bias) {
      const float4 b = bias[i % (size_t)ld4];
      a.x += b.x;
      a.y += b.y;
      a.z += b.z;
      a.w += b.w;
    }
    if (R) {
      const float4 r = R[i];
      a.x += r.x;
      a.y += r.y;
      a.z += r.z;
      a.w += r.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    if (act == OPP_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
    } else if (act == OPP_ACT_LEAKY) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
    }
    if (out3 != nullptr) {           // pre-split copy for the consuming convolution (row = i / ld4, channels 4 (i % ld4) ..)
      const size_t row = i / (size_t)ld4;
      opp_store_split4(out3, row * (size_t)ld3, (int)(i - row * (size_t)ld4) * 4, make_float4(v[0], v[1], v[2], v[3]));
    }
    if (out != nullptr) out[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <typename K>
void set_lds_once(K k, size_t lds, OppLdsOnce& once) {   // per device: a process may drive several GPUs
  opp_lds_opt_in(reinterpret_cast<const void*>(k), lds, once);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int DEPTH, int PREC>
int launch_prec(const OppGemm& g, hipStream_t stream, size_t extra_lds) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  // split-operand variants are built for the 64/128/256-column tiles (bf16x3: 12 x 16 B weight slots per row
  // must divide over the workgroup; its 4-wave 128x128 tile (one wave per SIMD, 204 + 64 registers, no spills) was built and
  // measured in round 4: 5 .. 20 % SLOWER than the 8-wave tile on every layer shape (profiles/r04_conv_bench_4wave_128x128.txt))
  // split-operand variants are instantiated for the tiles their policies can pick only (depth 2: deeper prefetch measured
  // no better with the shorter MFMA phase; the 4-wave 128-column tiles are fp32 tiles)
  // (fp16x2 keeps the 4-wave 128x128 tile: its score GEMM with the fused statistics runs on it)
  // 128 x 192 on 8 waves (32 x 96 per wave, 219 registers, 133 KB of LDS): the 192-column body of the 196-channel layers, whose last 4
  // columns come from conv_tail.hip (round 5; config 24)
  constexpr bool kNarrowB3 = BM == 128 && (BN == 192 || BN == 224) && NT == 512 && PREC == OPP_PREC_BF16X3;   // (and the 224-column ring tile)
  constexpr bool ok = (PREC == OPP_PREC_FP32 && BN != 192 && BN != 224) ||
                      (DEPTH == 2 && (NT == 512 || (BM == 64 && BN == 64) || (PREC == OPP_PREC_FP16X2 && BM == 128 && BN == 128)) &&
                       (BN == 128 || BN == 64 || BN == 256 || kNarrowB3) &&
                       (PREC != OPP_PREC_BF16X3 || (BN * 12) % NT == 0 || kNarrowB3) && (PREC != OPP_PREC_BF16X3 || BM * BN / NT <= 64 || kNarrowB3));
  if constexpr (ok) {
    const size_t lds = (size_t)2 * (BM + BN) * lds_stride(PREC) * sizeof(float) + extra_lds;
    const int tiles = opp_cdiv(g.M, BM) * opp_cdiv(g.n_store, BN);
    constexpr bool kAspBuilt = PREC == OPP_PREC_BF16X3 && DEPTH == 2 && !kNarrowB3 && (NT == 512 || (BM == 64 && BN == 64));
    if (g.conv && g.a_split) {
      if constexpr (kAspBuilt) {
        auto k = opp_gemm_asp_kernel<BM, BN, WAVES_M, WAVES_N, 0, DEPTH, PREC>;
        static OppLdsOnce attr_done;
        set_lds_once(k, lds, attr_done);
        hipLaunchKernelGGL(k, dim3(tiles, g.k_splits > 1 ? g.k_splits : 1), dim3(NT), lds, stream, g);
      } else {
        opp_set_error("gemm: tile %dx%d is not built for pre-split activations", BM, BN);
        return OPP_ERR_UNSUPPORTED;
      }
    } else if (g.conv) {
      auto k = opp_gemm_kernel<BM, BN, WAVES_M, WAVES_N, true, 0, DEPTH, PREC>;
      static OppLdsOnce attr_done;
      set_lds_once(k, lds, attr_done);
      hipLaunchKernelGGL(k, dim3(tiles, g.k_splits > 1 ? g.k_splits : 1), dim3(NT), lds, stream, g);
    } else {
      auto k = opp_gemm_kernel<BM, BN, WAVES_M, WAVES_N, false, 0, DEPTH, PREC>;
      static OppLdsOnce attr_done;
      set_lds_once(k, lds, attr_done);
      hipLaunchKernelGGL(k, dim3(tiles, g.k_splits > 1 ? g.k_splits : 1), dim3(NT), lds, stream, g);
    }
    OPP_CHECK_LAUNCH("opp_gemm_kernel");
    return OPP_OK;
  } else {
    opp_set_error("gemm: tile %dx%d on %d waves is not built for operand precision %d", BM, BN, WAVES_M * WAVES_N, PREC);
    return OPP_ERR_UNSUPPORTED;
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int DEPTH = 2>
int launch_cfg(const OppGemm& g, hipStream_t stream) {
#ifdef OPP_TUNING
  static const size_t extra_lds = getenv("OPP_EXTRA_LDS") ? (size_t)atoi(getenv("OPP_EXTRA_LDS")) : 0;  // tuning knob
#else
  constexpr size_t extra_lds = 0;
#endif
  switch (g.prec) {
    case OPP_PREC_FP32: return launch_prec<BM, BN, WAVES_M, WAVES_N, DEPTH, OPP_PREC_FP32>(g, stream, extra_lds);
#ifdef OPP_TUNING            // narrower than fp32: instantiated in the tuning library only (round 6)
    case OPP_PREC_FP16X2: return launch_prec<BM, BN, WAVES_M, WAVES_N, DEPTH, OPP_PREC_FP16X2>(g, stream, extra_lds);
#else
    case OPP_PREC_FP16X2: opp_set_error("gemm: fp16x2 operands are built into the tuning library only"); return OPP_ERR_UNSUPPORTED;
#endif
    case OPP_PREC_BF16X3: return launch_prec<BM, BN, WAVES_M, WAVES_N, DEPTH, OPP_PREC_BF16X3>(g, stream, extra_lds);
    default: opp_set_error("gemm: unknown operand precision %d", g.prec); return OPP_ERR_INVALID;
  }
}

#ifdef OPP_TUNING
unsigned long long* g_dbg_ts = nullptr;   // opp_debug_timestamps()

template <int BM, int BN, int WM, int WN, int PREC, int ABL = 9, bool ASP = false>
int launch_timed(const OppGemm& g_in, hipStream_t stream) {   // conv kernel with phase time stamps
  OppGemm g = g_in;
  g.dbg_ts = g_dbg_ts;
  if (ASP != (g.a_split != 0)) {
    opp_set_error("gemm: tuning config and activation layout do not match");
    return OPP_ERR_INVALID;
  }
  const size_t lds = (size_t)2 * (BM + BN) * lds_stride(PREC) * sizeof(float);
  void (*k)(const OppGemm) = nullptr;
  if constexpr (ASP) k = opp_gemm_asp_kernel<BM, BN, WM, WN, ABL, 2, PREC>;
  else k = opp_gemm_kernel<BM, BN, WM, WN, true, ABL, 2, PREC>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(opp_cdiv(g.M, BM) * opp_cdiv(g.n_store, BN)), dim3(WM * WN * 64), lds, stream, g);
  OPP_CHECK_LAUNCH("opp_gemm_kernel(timed)");
  return OPP_OK;
}

template <int ABL>
int launch_ablate(const OppGemm& g, hipStream_t stream) {   // fp32 128x128 conv with parts removed
  const size_t lds = (size_t)2 * (128 + 128) * lds_stride(OPP_PREC_FP32) * sizeof(float);
  auto k = opp_gemm_kernel<128, 128, 2, 2, true, ABL>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(opp_cdiv(g.M, 128) * opp_cdiv(g.n_store, 128)), dim3(256), lds, stream, g);
  OPP_CHECK_LAUNCH("opp_gemm_kernel(ablate)");
  return OPP_OK;
}

// tuning-only tile configurations: phase-stamped convs (120/121/122), loop ablations (191-193, 101-107)
int launch_tuning_cfg(const OppGemm& g, int cfg, hipStream_t stream) {
  const bool h2 = g.prec == OPP_PREC_FP16X2, h3 = g.prec == OPP_PREC_BF16X3;
  switch (cfg) {
    case 120: return !g.conv ? OPP_ERR_INVALID : h3 ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3>(g, stream) : h2 ? launch_timed<256, 128, 4, 2, OPP_PREC_FP16X2>(g, stream) : launch_timed<256, 128, 4, 2, OPP_PREC_FP32>(g, stream);
    case 121: return !g.conv || h3 ? OPP_ERR_INVALID : h2 ? launch_timed<128, 128, 2, 2, OPP_PREC_FP16X2>(g, stream) : launch_timed<128, 128, 2, 2, OPP_PREC_FP32>(g, stream);
    case 122: return !g.conv ? OPP_ERR_INVALID : h3 ? launch_timed<128, 128, 4, 2, OPP_PREC_BF16X3>(g, stream) : h2 ? launch_timed<128, 128, 4, 2, OPP_PREC_FP16X2>(g, stream) : launch_timed<128, 128, 4, 2, OPP_PREC_FP32>(g, stream);
    case 291: return (g.conv && h3) ? launch_timed<128, 128, 4, 2, OPP_PREC_BF16X3, 91>(g, stream) : OPP_ERR_INVALID;   // no global loads
    case 292: return (g.conv && h3) ? launch_timed<128, 128, 4, 2, OPP_PREC_BF16X3, 92>(g, stream) : OPP_ERR_INVALID;   // no LDS stores / split
    case 293: return (g.conv && h3) ? launch_timed<128, 128, 4, 2, OPP_PREC_BF16X3, 93>(g, stream) : OPP_ERR_INVALID;   // no barrier
    case 294: return (g.conv && h3) ? launch_timed<128, 128, 4, 2, OPP_PREC_BF16X3, 94>(g, stream) : OPP_ERR_INVALID;   // no LDS fragment reads
    case 295: return (g.conv && h3) ? launch_timed<128, 128, 4, 2, OPP_PREC_BF16X3, 95>(g, stream) : OPP_ERR_INVALID;   // no split arithmetic
    case 391: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 91>(g, stream) : OPP_ERR_INVALID;   // 256x128: no global loads
    case 392: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 92>(g, stream) : OPP_ERR_INVALID;   // no LDS stores / split
    case 393: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 93>(g, stream) : OPP_ERR_INVALID;   // no barrier
    case 394: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 94>(g, stream) : OPP_ERR_INVALID;   // no LDS fragment reads
    case 395: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 95>(g, stream) : OPP_ERR_INVALID;   // no split arithmetic
    // 256 x 128, several parts removed at once (400 + mask: fp32 rows split in the K loop; 500 + mask: pre-split rows)
    case 400: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 200>(g, stream) : OPP_ERR_INVALID;
    case 401: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 201>(g, stream) : OPP_ERR_INVALID;
    case 403: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 203>(g, stream) : OPP_ERR_INVALID;
    case 407: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 207>(g, stream) : OPP_ERR_INVALID;
    case 415: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 215>(g, stream) : OPP_ERR_INVALID;
    case 411: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 211>(g, stream) : OPP_ERR_INVALID;
    case 500: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 200, true>(g, stream) : OPP_ERR_INVALID;
    case 501: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 201, true>(g, stream) : OPP_ERR_INVALID;
    case 503: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 203, true>(g, stream) : OPP_ERR_INVALID;
    case 504: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 204, true>(g, stream) : OPP_ERR_INVALID;
    case 507: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 207, true>(g, stream) : OPP_ERR_INVALID;
    case 508: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 208, true>(g, stream) : OPP_ERR_INVALID;
    case 511: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 211, true>(g, stream) : OPP_ERR_INVALID;
    case 515: return (g.conv && h3) ? launch_timed<256, 128, 4, 2, OPP_PREC_BF16X3, 215, true>(g, stream) : OPP_ERR_INVALID;
    case 191: return (g.conv && h2) ? launch_timed<256, 128, 4, 2, OPP_PREC_FP16X2, 91>(g, stream) : OPP_ERR_INVALID;   // no global loads
    case 192: return (g.conv && h2) ? launch_timed<256, 128, 4, 2, OPP_PREC_FP16X2, 92>(g, stream) : OPP_ERR_INVALID;   // no LDS stores
    case 193: return (g.conv && h2) ? launch_timed<256, 128, 4, 2, OPP_PREC_FP16X2, 93>(g, stream) : OPP_ERR_INVALID;   // no barrier
    // 192-column tile for the 196-channel layers as 192 + a 4-column tail: 32 x 96 per wave, 219 registers, no scratch, 133 KB of LDS
    // (the 256 x 192 tile, 64 x 96 per wave, spills 248 bytes: not built).  NOT measured yet -- compiled for the next conv_bench runs.
    case 140: return h3 ? launch_prec<128, 192, 4, 2, 2, OPP_PREC_BF16X3>(g, stream, 0) : OPP_ERR_INVALID;   // (= product config 24)
    case 101: return g.conv ? launch_ablate<1>(g, stream) : OPP_ERR_INVALID;
    case 102: return g.conv ? launch_ablate<2>(g, stream) : OPP_ERR_INVALID;
    case 103: return g.conv ? launch_ablate<3>(g, stream) : OPP_ERR_INVALID;
    case 105: return g.conv ? launch_ablate<5>(g, stream) : OPP_ERR_INVALID;
    case 106: return g.conv ? launch_ablate<6>(g, stream) : OPP_ERR_INVALID;
    case 107: return g.conv ? launch_ablate<7>(g, stream) : OPP_ERR_INVALID;
    default: opp_set_error("gemm: unknown tile config %d", cfg); return OPP_ERR_INVALID;
  }
}
#endif

}  // namespace

// few output tiles under a long K: the convolution runs as 4 K slices (shape-only, identical under both tile policies)
// (ONE predicate for the launcher and for the match-driven fine branch of api.hip, whose bit-identity with the dense map rests on both
// walking K in the same slices)
bool opp_conv_splitk_by_shape(long long M, int n_store, int K) {
  const long long tiles128 = ((M + 127) / 128) * (long long)opp_cdiv(n_store, 128);
  return tiles128 <= 64 && K / 32 >= 32;
}
static bool splitk_by_shape(const OppGemm& g) { return opp_conv_splitk_by_shape(g.M, g.n_store, g.K); }

// The launcher's own tile choice (tile_cfg < 0) as a pure function of the problem shape, the operand arithmetic and the tile policy:
// no device, no launch -- opp_gemm_tile_for (api.hip) exports it so that the policy is pinned by CPU tests (tests/test_host_logic.py).
static int choose_tile(const OppGemm& g) {
  const bool split = g.prec != OPP_PREC_FP32;
  int cfg = -1;
  // Tile choice from the MI355X micro-bench (tools/conv_bench.py; 256 CUs x 4 SIMDs).  The
  // 196(->224)-channel stages run as two column tiles (128 + 96 real columns): measured 4-12 %
  // faster than dedicated 128x224 / 64x224 tiles (removed).
  const int t0 = opp_cdiv(g.M, 128) * opp_cdiv(g.n_store, 128);
  const int t1 = opp_cdiv(g.M, 64) * opp_cdiv(g.n_store, 128);
  if (t0 >= 384) cfg = 0;
  else if (t1 >= 512) cfg = 1;
  else cfg = 2;
  // prefetch depth of the long-K convolutions on 128x128 tiles: 4 register sets when the grid
  // is at most two workgroups per CU, else 3 (measured +8 % / +4 % over depth 2)
  // (split operands: the MFMA phase is shorter, depth 2 measured best on every layer)
  if (g.conv && g.K >= 768 && cfg == 0 && !split) cfg = t0 <= 512 ? 11 : 10;
  // fp32, 128-column outputs of the 256x256-pixel layers: the 8-wave 128x128 tile (two waves per SIMD cover the
  // prologue / epilogue of each other) measured 4-11 % faster than the deep-prefetch 4-wave one
  if (!split && t0 >= 384 && g.n_store <= 128) cfg = 25;
  if (g.prec == OPP_PREC_FP16X2) {
    // fp16x2: 8-wave tiles (two waves per SIMD cover each other's LDS hand-over and barrier).  Policy:
    // 128x128 (73 KB LDS, 120 registers: a second workgroup -- of this or of another stream's kernel --
    // fits on the CU) wherever it still gives ~a workgroup per CU, else 64x128, else the 4-wave 64x64 tile.
    // The 256x128 / 128x256 tiles (110 KB LDS) are 2-5 % faster for a kernel running alone on the two
    // largest layers but cost 2.3 % of the throughput with three forwards in flight (no co-residency).
    if (t0 >= 200) cfg = 25;          // 128x128 on 8 waves (32x64 per wave)
    else if (t1 >= 512) cfg = 26;     // 64x128 on 8 waves
    else cfg = 2;
  } else if (g.prec == OPP_PREC_BF16X3) {
    // bf16x3: the operand tiles are 1.5x larger (208 B per row and chunk), so the 128x128 tile already owns a CU
    // (106 KB LDS) and the grid runs in whole rounds of <= 256 workgroups: pick the tile with the smallest
    // estimated time = rounds x (K chunks x cycles per chunk + prologue / epilogue), constants from the MI355X
    // micro-bench (tools/conv_bench.py --prec 2): the 256x128 / 128x256 8-wave tiles (160 KB LDS, 64x64 per wave)
    // win wherever they still fill the chip in one round (256x256-pixel layers, QKV), 128x128 on the 128x128-pixel
    // layers, 64x128 (80 KB, two workgroups per CU) / 64x64 below that.
    // wpc = workgroups of this tile that share a CU (by LDS); co-resident workgroups share the matrix pipes, so a
    // chunk then costs wpc x the stand-alone time.
    struct Cand { int cfg, bm, bn, wpc, chunk, fixed; };
    // (24: 128 x 192, for outputs that are whole 192-column tiles -- the body of a 196-channel layer: measured 228 us against 272 on the
    // 128 x 256 tile at 256 x 256 pixels, profiles/r05_conv_bench_192_columns.txt)
    // (27: 128 x 224 on four fragment sets, for outputs of exactly 224 stored columns -- the 196-channel layers in one column tile,
    // 7 sub-tiles per SIMD and k-step instead of the 8 of 128 x 256)
    static const Cand cands[] = {{24, 128, 192, 1, 4000, 16000}, {27, 128, 224, 1, 4250, 16000}, {22, 128, 256, 1, 4800, 16000}, {20, 256, 128, 1, 4800, 16000},
                                 {25, 128, 128, 1, 2600, 13000}, {26, 64, 128, 2, 1400, 11000},
                                 {2, 64, 64, 3, 1000, 9000}};
    const long long nk = g.K / 32, cus = 256;
    long long best = -1;
    static const int on224_env = getenv("OPP_TILE_224") ? atoi(getenv("OPP_TILE_224")) : 1;         // A/B switch of the tools
#ifdef OPP_TUNING
    static const int only_env = getenv("OPP_B3_ONLY_CFG") ? atoi(getenv("OPP_B3_ONLY_CFG")) : -1;   // tuning: force one tile
    static const int skip_env = getenv("OPP_B3_SKIP_BIG") ? atoi(getenv("OPP_B3_SKIP_BIG")) : 0;    // tuning: no 160 KB tiles
#endif
    for (const Cand& c : cands) {
#ifdef OPP_TUNING
      if (only_env >= 0 && c.cfg != only_env && !(c.cfg == 2 && only_env != 2 && (long long)opp_cdiv(g.M, 64) * opp_cdiv(g.n_store, 64) <= 256)) continue;
      if (skip_env && (c.cfg == 20 || c.cfg == 22)) continue;
#endif
      if (c.bn == 256 && g.n_store <= 128) continue;   // half the tile would be padding
      if (c.bn == 192 && (!g.conv || g.n_store % 192 != 0)) continue;      // measured on convolution bodies only (conv_tail path): never a silent choice for dense GEMMs
      // measured (profiles/r05_conv_bench_224_columns.txt): bit-identical to the other tiles; 10 % faster than 128 x 256 on a grid of two full
      // rounds (l1_out2a: 283 vs 315 us; one forward in flight +1.4 ... 2.8 % images/s), 5-8 % SLOWER per tile where the grid is half a round
      // (125 vs 118 us at 128 x 128 pixels: 32 x 128 per wave reads every B fragment for ONE row block), and with several forwards in flight
      // it LOSES 1-2 % even on the big grid -- the chip is power-limited there, the eighth sub-tile of 128 x 256 multiplies zeros (cheap in
      // energy, only costly in time) while the ring tile moves 25 % more fragment bytes per useful MFMA.  So: latency policy, grids of at
      // least one full round (OPP_TILE_224=0 never, =2 always: the A/B switch of the tools).
      if (c.bn == 224 && g.a_split) continue;          // the ring tile is not built for pre-split activations
      if (c.bn == 224 && (g.n_store != 224 || g.n_real <= 0 || g.n_real > 208 || on224_env == 0 ||
                          (on224_env != 2 && (g.tile_policy == OPP_TILES_THROUGHPUT || opp_cdiv(g.M, 128) < 256)))) continue;
      const long long tiles = (long long)opp_cdiv(g.M, c.bm) * opp_cdiv(g.n_store, c.bn);
      const long long slots = cus * c.wpc, full = tiles / slots, rem = tiles % slots;
      long long est = full * (nk * c.chunk * c.wpc + c.fixed);
      if (rem > 0) est += nk * c.chunk * ((rem + cus - 1) / cus) + c.fixed;
      if (g.tile_policy == OPP_TILES_THROUGHPUT) {
        // several forwards in flight (MatcherPool, bench --streams > 1): other streams' kernels fill the CUs a
        // grid leaves idle, so what counts is the CU time a launch occupies, not its own latency: the larger tiles
        // (64x64 per wave: fewer LDS bytes and barriers per MFMA) win even where they cover only part of the chip.
        // Measured with 3 forwards in flight: +3.5 ... 6.5 % images/s, at -9 % for a single forward on its own.
        if (tiles < 64 && c.cfg != 2) continue;
        est = tiles * (nk * c.chunk + c.fixed / c.wpc);
      }
      if (best < 0 || est < best) {
        best = est;
        cfg = c.cfg;
      }
    }
  }
  return cfg;
}

int opp_gemm_launch_cfg(const OppGemm& g_in, int cfg, hipStream_t stream) {
  OppGemm g = g_in;
  auto al16p = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    bool v = g.n_store % 4 == 0 && (g.C == nullptr || (g.ldc % 4 == 0 && al16(g.C))) && (!g.bias || al16(g.bias)) && g.qk_cols % 4 == 0;
    if (g.res_mode != OPP_RES_NONE) v = v && g.ldr % 4 == 0 && al16(g.R);
    g.vec_epilogue = v ? 1 : 0;
  }
#ifdef OPP_TUNING
  static const int xcd_env = getenv("OPP_XCD_SWIZZLE") ? atoi(getenv("OPP_XCD_SWIZZLE")) : 1;
  g.xcd_swizzle = xcd_env;
#else
  g.xcd_swizzle = 1;
#endif
  {  // bit 1: static wave priority for the second half of an 8-wave workgroup (A/B switch OPP_WAVE_PRIO; measured)
    static const int prio_env = getenv("OPP_WAVE_PRIO") ? atoi(getenv("OPP_WAVE_PRIO")) : 0;
    if (prio_env) g.xcd_swizzle |= 2;
  }
  OPP_CHECK_ARG(g.M > 0 && g.N > 0 && g.K > 0 && g.K % 32 == 0, "gemm: bad M/N/K (%d,%d,%d)", g.M, g.N, g.K);
  OPP_CHECK_ARG(g.n_store >= g.N && (g.C || g.C3) && g.W && g.A0, "gemm: bad output/operands");
  OPP_CHECK_ARG((size_t)g.M * (size_t)g.ldc < (1ull << 31), "gemm: output too large for 32-bit indexing");
  if (g.C3 != nullptr || g.a_split) {
    OPP_CHECK_ARG(g.conv && g.prec == OPP_PREC_BF16X3, "gemm: pre-split activations are a bf16x3 convolution feature");
    OPP_CHECK_ARG(g.C3 == nullptr || (g.vec_epilogue && g.n_store % 8 == 0 && g.ld3 >= g.n_store * 6 && g.ld3 % 16 == 0 && al16p(g.C3) && !g.ln_gamma && !g.stat_rowmax &&
                                      g.k_splits <= 1), "gemm: split output needs the 16-byte epilogue, whole 8-channel groups and a plain launch");
    OPP_CHECK_ARG(!g.a_split || (g.tail_grp == 0 && al16p(g.A0)), "gemm: pre-split input needs Cin %% 32 == 0 real channels");
    OPP_CHECK_ARG(g.C != nullptr || g.C3 != nullptr, "gemm: no output");
  }
  g.w_bytes = (unsigned)((size_t)g.N * g.ldw * 4);
  OPP_CHECK_ARG((size_t)g.N * g.ldw * 4 < (1ull << 31), "gemm: weight operand too large for buffer addressing");
  OPP_CHECK_ARG(g.prec != OPP_PREC_BF16X3 || g.ldw * 2 >= g.K * 3, "gemm: bf16x3 weights need a row stride of 1.5 K floats");
  if (g.conv) {
    const int k_expect = g.tail_grp > 0 ? (g.tail_grp * g.ksize * g.ksize + (g.ksize * g.ksize + 7) / 8) * 32 : g.ksize * g.ksize * g.Cin;
    OPP_CHECK_ARG(g.Cin % 32 == 0 && g.K == k_expect, "conv: Cin %% 32 / K mismatch");
    OPP_CHECK_ARG(g.tail_grp == 0 || (g.ksize == 3 && g.tail_grp == g.Cin / 32 - 1), "conv: K tail packing needs a 3x3 kernel and one partial channel group");
    OPP_CHECK_ARG(g.M == g.Bn * g.Hout * g.Wout, "conv: M != B*Hout*Wout");
    const size_t ab = (size_t)g.Bn * g.Hin * g.Win * g.Cin * (g.a_split ? 6 : 4);
    OPP_CHECK_ARG(ab < (1ull << 31), "conv: input too large for buffer addressing");
    g.a0_bytes = (unsigned)ab;
  } else {
    const size_t a0b = ((size_t)(g.M - 1) * g.lda0 + (g.ksplit < g.K ? g.ksplit : g.K)) * 4;
    const size_t a1b = g.A1 ? ((size_t)(g.M - 1) * g.lda1 + (g.K - g.ksplit)) * 4 : 0;
    OPP_CHECK_ARG(a0b < (1ull << 31) && a1b < (1ull << 31), "gemm: A operand too large for buffer addressing");
    g.a0_bytes = (unsigned)a0b;
    g.a1_bytes = (unsigned)a1b;
    OPP_CHECK_ARG(g.ksplit % 32 == 0 && (g.ksplit >= g.K || g.A1), "gemm: bad ksplit");
    OPP_CHECK_ARG(g.res_mode != OPP_RES_BILINEAR2X, "gemm: bilinear residual needs conv mode");
  }
  if (g.k_splits > 1) {
    OPP_CHECK_ARG(g.k_chunks_per_split > 0 && (long long)g.k_splits * g.k_chunks_per_split * 32 >= g.K && g.split_stride >= (size_t)g.M * g.ldc,
                  "gemm: bad split-K description (%d splits x %d chunks for K %d)", g.k_splits, g.k_chunks_per_split, g.K);
    OPP_CHECK_ARG(!g.bias && g.res_mode == OPP_RES_NONE && g.act == OPP_ACT_NONE && !g.ln_gamma && !g.stat_rowmax && !g.col_mask &&
                      g.out_mul == 1.f && g.out_div == 1.f, "gemm: split-K partial products take a plain epilogue");
    OPP_CHECK_ARG((size_t)g.k_splits * g.split_stride < (1ull << 31), "gemm: split-K partials too large for 32-bit indexing");
  }
  if (g.conv && cfg < 0 && g.prec == OPP_PREC_BF16X3 && g.k_splits <= 1 && g.splitk_ws != nullptr && !g.ln_gamma && !g.stat_rowmax && !g.col_mask &&
      (g.res_mode == OPP_RES_NONE || (g.res_mode == OPP_RES_DIRECT && g.ldr == g.ldc)) && (g.act == OPP_ACT_NONE || g.act == OPP_ACT_RELU || g.act == OPP_ACT_LEAKY) &&
      g.out_mul == 1.f && g.out_div == 1.f && g.vec_epilogue && g.n_store == g.ldc) {
    // few output tiles under a long K (shape-only decision, identical under both tile policies): 4 K slices on the 8-wave 128 x 128 tile
    constexpr int kSplits = 4;
    const int nkc = g.K / 32;
    const size_t need = (size_t)kSplits * g.M * g.ldc;
    const bool by_shape = splitk_by_shape(g);
    if ((g.splitk_force > 0 || (g.splitk_force < 0 && by_shape)) && nkc >= kSplits && g.splitk_ws_floats >= need && need < (1ull << 31)) {
      OppGemm gs = g;
      gs.C = g.splitk_ws;
      gs.C3 = nullptr;              // the K slices are fp32 partials; the fixed-order epilogue below emits the split copy
      gs.k_splits = kSplits;
      gs.k_chunks_per_split = opp_cdiv(nkc, kSplits);
      gs.split_stride = (size_t)g.M * g.ldc;
      gs.bias = nullptr;
      gs.res_mode = OPP_RES_NONE;
      gs.R = nullptr;
      gs.act = OPP_ACT_NONE;
      gs.splitk_ws = nullptr;
      OPP_TRY(opp_gemm_launch_cfg(gs, 25, stream));
      const size_t n4 = (size_t)g.M * g.ldc / 4;
      const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
      // algorithmic bytes of the reduction: the partial slices read, the residual read, the output written
      OppProfScope prof_epi(OPP_PROF_SPLITK_EPILOGUE, stream, (double)(kSplits + 1 + (g.res_mode == OPP_RES_DIRECT ? 1 : 0)) * (double)g.M * g.ldc * 4.0);
      hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(g.splitk_ws), kSplits, n4, n4, g.ldc / 4,
                         reinterpret_cast<const float4*>(g.bias), g.res_mode == OPP_RES_DIRECT ? reinterpret_cast<const float4*>(g.R) : nullptr, g.act,
                         reinterpret_cast<float4*>(g.C), g.C3, g.ld3);
      OPP_CHECK_LAUNCH("splitk_epilogue_kernel");
      return OPP_OK;
    }
  }
  const bool auto_cfg = cfg < 0;
  if (cfg < 0) cfg = choose_tile(g);
  // (the statistics scratch sits behind the staged C tile in the operand LDS: 4-wave tile for fp32 / fp16x2,
  // 8-wave tile for bf16x3, whose operand buffers are larger)
  OPP_CHECK_ARG(g.stat_rowmax == nullptr || (g.prec == OPP_PREC_BF16X3 ? (cfg == 25 || cfg == 20) : cfg == 0),
                "gemm: fused softmax statistics need the 128x128 tile (config 0; bf16x3: config 25, or 20 = 256x128)");
  if (g.ln_gamma != nullptr) {
    if (auto_cfg || (cfg != 30 && cfg != 26)) cfg = g.n_store == 256 ? 30 : 26;   // the tile must span the row
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    OPP_CHECK_ARG(g.ln_beta && ((cfg == 30 && g.n_store == 256) || (cfg == 26 && g.n_store == 128)) && g.N == g.n_store && !g.bias &&
                      g.act == OPP_ACT_NONE && g.res_mode == OPP_RES_NONE && g.ldc % 4 == 0 && al(g.C) && al(g.ln_gamma) &&
                      al(g.ln_beta) && (!g.ln_res || (al(g.ln_res) && g.ln_ldres % 4 == 0)),
                  "gemm: fused LayerNorm needs a 256- or 128-column output, no bias/activation/residual mode, aligned operands");
  }
  // live timing of this symbol when armed (bench.py roofline leg)
  // (the K slices of a split convolution have their own symbol: they are a different kernel shape than the same tile unsplit)
  OppProfScope prof_scope(g.conv && g.k_splits > 1 ? (int)OPP_PROF_CONV_SPLITK : opp_prof_gemm_symbol(cfg, g.conv ? 1 : (g.stat_rowmax ? 2 : 0)), stream,
                          g.alg_flops > 0.0 ? g.alg_flops : 2.0 * (double)g.M * (double)g.N * (double)g.K);
  int rc;
  switch (cfg) {
    case 0: rc = launch_cfg<128, 128, 2, 2>(g, stream); break;
    case 1: rc = launch_cfg<64, 128, 2, 2>(g, stream); break;
    case 2: rc = launch_cfg<64, 64, 2, 2>(g, stream); break;
    case 10: rc = launch_cfg<128, 128, 2, 2, 3>(g, stream); break;  // deeper global prefetch (long-K fp32 convs)
    case 11: rc = launch_cfg<128, 128, 2, 2, 4>(g, stream); break;
    case 20: rc = launch_cfg<256, 128, 4, 2>(g, stream); break;     // 8 waves: two per SIMD
    case 22: rc = launch_cfg<128, 256, 2, 4>(g, stream); break;
    case 24: rc = launch_cfg<128, 192, 4, 2>(g, stream); break;     // 8 waves, 32x96 per wave: 192-column bodies of the 196-channel layers
    case 25: rc = launch_cfg<128, 128, 4, 2>(g, stream); break;     // 8 waves, 32x64 per wave (M ~ 16k layers)
    case 27:
      // (explicit requests -- tests, tools -- vouch for zero weight rows >= 208 themselves: the C ABI of the single convolution carries the padded count only)
      if (g.prec != OPP_PREC_BF16X3 || g.n_store > 224 || (g.n_real > 208 && g.n_real != g.n_store)) {
        opp_set_error("gemm: tile config 27 (128 x 224) is a bf16x3 tile for outputs of <= 208 real and <= 224 stored columns");
        rc = OPP_ERR_UNSUPPORTED;
      } else {
        rc = launch_cfg<128, 224, 4, 2>(g, stream);
      }
      break;   // 8 waves, 32 x 128 | 32 x 96 per wave, four fragment sets
    case 26: rc = launch_cfg<64, 128, 2, 4>(g, stream); break;      // 8 waves, 32x32 per wave
    case 30: rc = launch_cfg<64, 256, 2, 4>(g, stream); break;      // 8 waves, full 256-column rows (fused LayerNorm)
    default:
#ifdef OPP_TUNING
      rc = launch_tuning_cfg(g, cfg, stream);
#else
      opp_set_error("gemm: unknown tile config %d", cfg);
      rc = OPP_ERR_INVALID;
#endif
      break;
  }
  return rc;
}

// tuning builds (-DOPP_TUNING): device buffer receiving 4 shader-clock stamps (entry, loop start, loop end,
// exit) per wave of the timed conv variants (tile configs 120-122); a no-op error otherwise
extern "C" int opp_debug_timestamps(void* buf) {
#ifdef OPP_TUNING
  g_dbg_ts = static_cast<unsigned long long*>(buf);
  opp_gemm_ss_debug_timestamps(buf, getenv("OPP_SS_TS_MODE") ? atoi(getenv("OPP_SS_TS_MODE")) : 0);
  return OPP_OK;
#else
  (void)buf;
  opp_set_error("opp_debug_timestamps: library built without -DOPP_TUNING");
  return OPP_ERR_UNSUPPORTED;
#endif
}

int opp_gemm_launch(const OppGemm& g, hipStream_t stream) { return opp_gemm_launch_cfg(g, -1, stream); }

// tile configuration the launcher would pick by itself (host only): >= 0 a tile config, + 1000 when a bf16x3 convolution of this shape runs as 4 K slices
int opp_gemm_choose_tile(const OppGemm& g) {
  int cfg = choose_tile(g);
  if (g.conv && g.prec == OPP_PREC_BF16X3 && splitk_by_shape(g) && g.K / 32 >= 4) cfg = 1000 + 25;
  return cfg;
}
