// "split-split" GEMM for gfx950: BOTH operands arrive pre-split in memory in the bf16x3 form
// (x = hi + mid + lo bf16 exactly; every 8 consecutive k = 48 B [hi x8 | mid x8 | lo x8], opp_pack_b3), so the K loop has no
// conversion arithmetic, no VGPR staging and no LDS stores: the operand tiles go global -> LDS with
// buffer_load_dwordx4 ... lds (LDS-DMA).
//
// Structure (MI355X-first): a workgroup is FOUR waves (one per SIMD) on a 128 x 128 tile, 64 x 64 per wave; a stage is one
// k16-step (96 B per tile row), three LDS slots of 24 KB = 72 KB per workgroup, so TWO workgroups share a CU: while one is in
// its prologue / epilogue (global latency, VALU-heavy statistics, stores) the other's MFMAs own the matrix pipes -- the
// overlap a single 8-wave workgroup per CU cannot have.  Per stage and wave: 24 MFMAs (six bf16 products per 32 x 32 x 16
// block, fp32 accumulate, the accumulation sequence of opp_gemm_kernel<bf16x3>), 12 ds_read_b128 of the next stage's
// fragments, 6 LDS-DMA instructions of the stage three ahead issued one per four MFMAs, ONE raw s_barrier with explicit
// vmcnt / lgkmcnt counts (a __syncthreads() would drain the DMA queue).
//
// LDS image: a tile row of one stage is 96 B = 6 pieces of 16 B (2 k-groups x {hi, mid, lo}).  LDS-DMA writes lane-linearly
// (wave-uniform base + lane * 16), so the image is plain row-major [row][6 pieces], no padding; the bank-conflict-free
// fragment reads come from a rotation applied on the SOURCE side: position pos of row r holds global piece
// (pos + 3 * ((r >> 3) & 1)) mod 6.  The 16 rows of a ds_read_b128 lane group then fall on 16 distinct 4-bank groups
// ((6 r + pos) mod 16 runs over the 8 even residues for the rows with bit 3 clear and the 8 odd ones for the others).
//
// First user: the coarse score matrix and its dual softmax (utils/coarse_matching.py:99-115, :145-172) in TWO SWEEPS of the
// same GEMM instead of GEMM + in-place softmax passes over the materialised N x L matrix:
//   sweep 1 (OPP_SS_STATS): the score tile lives only in the accumulators; its (max, sum exp) per tile row and tile column
//                           leave as partials (merged by the small kernels of coarse_match.hip);
//   sweep 2 (OPP_SS_CONF):  the tile is recomputed (bit-identical: same instruction sequence), turned into
//                           conf = softmax_col * softmax_row with the merged statistics and written ONCE; the maxima the
//                           mutual-nearest-neighbour test needs leave as per-tile partials.
// The kernel's M dimension is the IMAGE CELL and its N dimension (MFMA lanes) the 3D POINT: a lane then holds four
// consecutive cells of one point per accumulator quad, i.e. 16 contiguous bytes of conf[point][cell] -> dwordx4 stores
// straight from the accumulators, no LDS transposition.
// HBM traffic of the whole stage: one 82 MB write (N = 5000, L = 4096) instead of write + read + write (246 MB).
#include <stdlib.h>

#include <type_traits>

#include "opp_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned kOob = 0x80000000u;
constexpr int BM = 128, BN = 128, WM = 2, WN = 2, NT = 256, TM = 2, TN = 2;
constexpr int ROWB = 96;                              // bytes per tile row and stage (one k16-step)
constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, SLOT = A_BYTES + B_BYTES, NS = 3;
constexpr int A_LD = BM * 6 / NT, B_LD = BN * 6 / NT; // LDS-DMA instructions per wave and stage
constexpr int LPS = A_LD + B_LD;

template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  const int t = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false);
  return fmaxf(v, __int_as_float(t));
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
  return v + __int_as_float(t);
}
// reductions over the 32 lanes of one wave half (lanes 0-31 / 32-63) on the DPP path only (no LDS crossbar, no waits):
// fixed order; the result is valid in the LAST 16 lanes of the half (lanes 16-31 / 48-63)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_rm(float v) {
  const int t = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return fmaxf(v, __int_as_float(t));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_rm(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float half_max(float v) {
  v = dpp_max<0xB1>(v);           // quad_perm [1,0,3,2]
  v = dpp_max<0x4E>(v);           // quad_perm [2,3,0,1]
  v = dpp_max<0x141>(v);          // row_half_mirror
  v = dpp_max<0x140>(v);          // row_mirror: every lane of a 16-lane row holds the row max
  return dpp_max_rm<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3: they now hold the max of their half
}
__device__ __forceinline__ float half_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  return dpp_add_rm<0x142, 0xA>(v);
}

// x / d for a loop-invariant divisor d with rd = RN(1 / d): q = RN(x rd), then one exact-remainder correction -- the
// correctly rounded quotient (Markstein), three FMAs instead of the ten-instruction IEEE division sequence
__device__ __forceinline__ float div_invariant(float x, float d, float rd) {
  const float q = x * rd;
  const float rem = fmaf(-q, d, x);
  return fmaf(rem, rd, q);
}

template <int MODE>
__global__ __launch_bounds__(NT, 2) void gemm_ss_kernel(const OppGemmSS g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef OPP_TUNING
  const unsigned long long ts0 = __builtin_readcyclecounter();
  unsigned long long ts1 = 0, ts2 = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  // XCD-aware tile order (workgroup b runs on XCD b % 8; speed only -- every output is indexed by tile coordinates): an XCD gets
  // a contiguous range of a linear order that walks STRIPS of RS row panels column by column (row fastest).  The ~64 tiles an
  // XCD has in flight then cover RS row panels x 8 column panels = 16 operand panels of 192 KB (3 MB of its 4 MB L2), every
  // column panel is fetched once per strip and XCD instead of once per row panel: 227 -> MB of fetches per
  // launch at 4096 x 5000 (row-major order: each XCD streamed all 40 column panels four times).
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  int tile_lin = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    tile_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int RS = 8;
  const int strip = tile_lin / (RS * tiles_n);
  const int within = tile_lin - strip * (RS * tiles_n);
  const int strip_rows = min(RS, tiles_m - strip * RS);
  const int tile_n = within / strip_rows;
  const int tile_m = strip * RS + (within - tile_n * strip_rows);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- LDS-DMA slots: instruction n of a tile covers linear pieces [64 n, 64 n + 64) of the [rows][6] image --------
  unsigned a_voff[A_LD], b_voff[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int P = (wave * A_LD + i) * 64 + lane;
    const int row = P / 6, pos = P - row * 6;
    const int q = (pos + 3 * ((row >> 3) & 1)) % 6;
    a_voff[i] = m0 + row < g.M ? (unsigned)(row * g.lda + q * 16) : kOob;
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int P = (wave * B_LD + i) * 64 + lane;
    const int row = P / 6, pos = P - row * 6;
    const int q = (pos + 3 * ((row >> 3) & 1)) % 6;
    b_voff[i] = n0 + row < g.N ? (unsigned)(row * g.ldb + q * 16) : kOob;
  }
  const int soffA0 = m0 * g.lda, soffB0 = n0 * g.ldb;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  // item k of stage s (k < LPS): one buffer_load_dwordx4 ... lds.  live = false (stages past the end of K): a zero-sized buffer
  // (one s_cselect on the wave-uniform descriptor), the instruction still counts in vmcnt, so the loop has no branches
  auto dma_item = [&](int s, int slot, int k, bool live) {
    char* base = smem + slot * SLOT;
    if (k < A_LD) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, live ? g.a_bytes : 0, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(base + (wave * A_LD + k) * 1024), 16, (int)a_voff[k], soffA0 + s * ROWB, 0, 0);
    } else {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, live ? g.b_bytes : 0, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(base + A_BYTES + (wave * B_LD + (k - A_LD)) * 1024), 16, (int)b_voff[k - A_LD],
                                               soffB0 + s * ROWB, 0, 0);
    }
  };

  // ---- fragment addressing: lane (row l31, k-group = half) reads parts hi / mid / lo = global pieces 3 half + p --------
  int foff[3];
  {
    const int b3 = (l31 >> 3) & 1;
#pragma unroll
    for (int p = 0; p < 3; ++p) foff[p] = 16 * ((3 * half + p + 3 * b3) % 6);
  }
  const int a_row = (wm * TM * 32 + l31) * ROWB;
  const int b_row = A_BYTES + (wn * TN * 32 + l31) * ROWB;
  u32x4 fa[2][TM][3], fb[2][TN][3];
  auto read_frags = [&](int slot, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const char* base = smem + slot * SLOT;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[set][i][p] = *reinterpret_cast<const u32x4*>(base + a_row + i * 32 * ROWB + foff[p]);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[set][j][p] = *reinterpret_cast<const u32x4*>(base + b_row + j * 32 * ROWB + foff[p]);
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int ns = g.K / 16;     // stages (even: K % 32 == 0)
  // one stage: barrier (stage s + 1 landed everywhere, slot s % 3 free), 24 MFMAs of stage s; behind the first two the
  // fragment reads of stage s + 1, then the six DMA instructions of stage s + 3 (into slot s % 3), one per four MFMAs
  auto stage = [&](int s, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS) : "memory");     // stage s + 1 landed; s + 2 may be in flight
    __builtin_amdgcn_s_barrier();
    const int slot = s % NS;
    const bool live = s + 3 < ns;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};   // A part of product pr (0 hi, 1 mid, 2 lo), smallest terms first
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    int n = 0;
#pragma unroll
    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][i][PA[pr]]),
                                                               __builtin_bit_cast(bf16x8, fb[set][j][PB[pr]]), acc[i][j], 0, 0, 0);
          if (n == 1) {
            __builtin_amdgcn_sched_barrier(0);
            read_frags((s + 1) % NS, std::integral_constant<int, set ^ 1>{});   // (past the last stage: a dead slot, never used)
            __builtin_amdgcn_sched_barrier(0);
          }
          if (n % 4 == 3 && n / 4 < LPS) {
            dma_item(s + 3, slot, n / 4, live);
            __builtin_amdgcn_sched_barrier(0);
          }
          ++n;
        }
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: three stages in flight, the first one awaited
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int k = 0; k < LPS; ++k) dma_item(s, s, k, s < ns);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(0, std::integral_constant<int, 0>{});
#ifdef OPP_TUNING
  ts1 = __builtin_readcyclecounter();
#endif
  for (int s = 0; s < ns; s += 2) {
    stage(s, std::integral_constant<int, 0>{});
    stage(s + 1, std::integral_constant<int, 1>{});
  }
#ifdef OPP_TUNING
  ts2 = __builtin_readcyclecounter();
#endif

  // ---- epilogues --------------------------------------------------------------------------------------------
  // v = acc * out_mul / out_div, the reference's order of operations (feature scaling, then the temperature division)
  if ((g.out_mul != 1.f) || (g.out_div != 1.f)) {
    const float rd = 1.0f / g.out_div;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = div_invariant(acc[i][j][r] * g.out_mul, g.out_div, rd);
  }
  auto row_of = [&](int i, int r) { return wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half; };   // tile row of acc[i][.][r]
  if (g.row_mask != nullptr) {   // masked image cells: sim += -1e9 (coarse_matching.py:108-114); cells = kernel rows
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float mk = g.row_mask[min(m0 + row_of(i, r), g.M - 1)];      // (rows past M are never used)
        const float add = mk == 0.f ? -1e9f : 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j][r] += add;
      }
  }
  const int nrows = min(BM, g.M - m0), ncols = min(BN, g.N - n0);
  const bool full = nrows == BM && ncols == BN;      // wave-uniform: interior tiles skip every validity select
  float* sc = reinterpret_cast<float*>(smem);         // scratch in the (dead) operand slots
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the zero-sized tail DMAs and the last fragment reads
  __syncthreads();

  // Both epilogues stage the tile once through LDS as T[col][row] (stride 132 floats; a lane's accumulator quad = four
  // consecutive rows of one column = one ds_write_b128): the per-column quantities are then in-register reductions over the
  // lane's own 64 values, the per-row ones a loop over the columns with lane = row (consecutive LDS addresses), and nothing
  // needs a cross-lane butterfly.  v_max via inline asm: fmaxf() costs an extra canonicalising v_max per operand here.
  constexpr int TS = BM + 4;
  float* T = sc;                                   // [BN][TS] = 66 KB of the 72 KB
  float* red = sc + BN * TS;                       // small cross-wave scratch behind it
  auto vmax = [](float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  };
  auto stage_tile = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int col = wn * TN * 32 + j * 32 + l31, row = wm * TM * 32 + i * 32 + 8 * q4 + 4 * half;
          *reinterpret_cast<float4*>(T + col * TS + row) =
              make_float4(acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]);
        }
  };
  const int rrow = tid & (BM - 1), rpart = tid >> 7;          // row pass: thread = (row, half of the columns)
  constexpr int CPP = BN / 2;                                 // columns per part

  if constexpr (MODE == OPP_SS_STATS || MODE == OPP_SS_STATS_STORE) {
    // (max, sum exp(v - max)) of this tile per row (over its columns) and per column (over its rows).  Every partial is a
    // fixed-order function of the row's / column's values only, so duplicated rows or columns get bit-equal statistics
    // wherever they sit (exact ties in the confidence matrix then resolve like the reference's).
  auto body = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    float* red_r = red;                   // [2][BM]
    float* red_c = red + 2 * BM;          // [WM][BN]
    stage_tile();
    // columns, from the accumulators: max
    float cmx[TN], csm[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float m = -INFINITY;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = vmax(m, (FULL || row_of(i, r) < nrows) ? acc[i][j][r] : -INFINITY);
      m = vmax(m, __shfl_xor(m, 32, 64));
      if (half == 0) red_c[wm * BN + wn * TN * 32 + j * 32 + l31] = m;
    }
    __syncthreads();
    // rows, from the staged tile: max over this part's columns
    const int c0 = rpart * CPP, c1 = FULL ? c0 + CPP : min(c0 + CPP, ncols);
    float rv[CPP];
    float rm = -INFINITY;
#pragma unroll
    for (int c = 0; c < CPP; ++c) {
      rv[c] = T[(c0 + c) * TS + rrow];
      rm = vmax(rm, (FULL || c0 + c < c1) ? rv[c] : -INFINITY);
    }
    red_r[rpart * BM + rrow] = rm;
#pragma unroll
    for (int j = 0; j < TN; ++j) cmx[j] = vmax(red_c[wn * TN * 32 + j * 32 + l31], red_c[BN + wn * TN * 32 + j * 32 + l31]);
    __syncthreads();
    rm = vmax(red_r[rrow], red_r[BM + rrow]);
    float rs = 0.f;
#pragma unroll
    for (int c = 0; c < CPP; ++c) rs += (FULL || c0 + c < c1) ? __expf(rv[c] - rm) : 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += (FULL || row_of(i, r) < nrows) ? __expf(acc[i][j][r] - cmx[j]) : 0.f;
      csm[j] = sm + __shfl_xor(sm, 32, 64);
    }
    if constexpr (MODE == OPP_SS_STATS_STORE) {
      // the score tile itself leaves as well (single-sweep matcher: conf is formed in place by conf_reg_kernel), transposed
      // like the confidence tile of OPP_SS_CONF: lane = 16 bytes of a row of out[col][row]
      if (g.vec_store) {
#pragma unroll
        for (int it = 0; it < BN * (BM / 4) / NT; ++it) {
          const int u = tid + it * NT;
          const int col = u / (BM / 4), r4 = (u - col * (BM / 4)) * 4;
          if (FULL || (col < ncols && r4 + 3 < nrows)) {
            *reinterpret_cast<float4*>(g.C + (size_t)(n0 + col) * g.ldc + m0 + r4) = *reinterpret_cast<const float4*>(T + col * TS + r4);
          } else if (col < ncols) {
            for (int e = 0; e < 4; ++e)
              if (r4 + e < nrows) g.C[(size_t)(n0 + col) * g.ldc + m0 + r4 + e] = T[col * TS + r4 + e];
          }
        }
      } else {
        for (int u = tid; u < BN * BM; u += NT) {
          const int col = u / BM, r = u - col * BM;
          if (col < ncols && r < nrows) g.C[(size_t)(n0 + col) * g.ldc + m0 + r] = T[col * TS + r];
        }
      }
    }
    __syncthreads();                       // red_r / red_c maxima consumed
    red_r[rpart * BM + rrow] = rs;
    if (half == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) red_c[wm * BN + wn * TN * 32 + j * 32 + l31] = csm[j];
    }
    if (wm == 0 && half == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        if (wn * TN * 32 + j * 32 + l31 < ncols) g.stat_colmax[(size_t)tile_m * g.N + n0 + wn * TN * 32 + j * 32 + l31] = cmx[j];
    }
    __syncthreads();
    if (tid < BM) {
      if (tid < nrows) {
        const size_t o = (size_t)(m0 + tid) * tiles_n + tile_n;
        g.stat_rowmax[o] = rm;
        g.stat_rowsum[o] = red_r[tid] + red_r[BM + tid];
      }
    } else if (tid - BM < ncols) {
      const int u = tid - BM;
      g.stat_colsum[(size_t)tile_m * g.N + n0 + u] = red_c[u] + red_c[BN + u];
    }
  };
    if (full) body(std::true_type{});
    else body(std::false_type{});
  } else {
    // conf tile from the merged statistics; it is staged as T[col][row] = conf[point][cell] row-major and leaves with fully
    // coalesced 16-byte stores (a wave writes two 512-byte rows per instruction).  Per column (point): best confidence / first
    // row (cell) holding it / how many rows hold it; per row (cell): max over the columns.
  auto body = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    float* s_rm = red;                                  // [BM] row statistic: max
    float* s_rr = red + BM;                             // [BM] row statistic: 1 / sum (v_rcp, as conf_value() of coarse_match.hip)
    float* p_best = red + 2 * BM;                       // [WM][BN]
    int* p_arg = reinterpret_cast<int*>(p_best + WM * BN);
    int* p_ties = p_arg + WM * BN;
    unsigned* red_r = reinterpret_cast<unsigned*>(p_ties + WM * BN);   // [2][BM]
    if (tid < BM) {
      const bool ok = tid < nrows;
      s_rm[tid] = ok ? g.rstat_max[m0 + tid] : 0.f;
      s_rr[tid] = ok ? __frcp_rn(g.rstat_sum[m0 + tid]) : 0.f;
    }
    float cm[TN], crs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + l31;
      const bool ok = col < g.N;
      cm[j] = ok ? g.cstat_max[col] : 0.f;
      crs[j] = ok ? 1.0f / g.cstat_sum[col] : 0.f;
    }
    __syncthreads();
    float best[TN];
    int arg[TN], ties[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      best[j] = -1.f;
      arg[j] = 0x7fffffff;
      ties[j] = 0;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int lr0 = wm * TM * 32 + i * 32 + 8 * q4 + 4 * half;       // 4 consecutive tile rows: r = 4 q4 + e
        const float4 rm4 = *reinterpret_cast<const float4*>(s_rm + lr0);
        const float4 rr4 = *reinterpret_cast<const float4*>(s_rr + lr0);
        const float rm[4] = {rm4.x, rm4.y, rm4.z, rm4.w}, rr[4] = {rr4.x, rr4.y, rr4.z, rr4.w};
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // exp((v - cellmax) + (v - pointmax)) * (1 / cellsum * 1 / pointsum): conf_value() of coarse_match.hip
            const float c = __expf((acc[i][j][4 * q4 + e] - rm[e]) + (acc[i][j][4 * q4 + e] - cm[j])) * (rr[e] * crs[j]);
            acc[i][j][4 * q4 + e] = c;
            if (FULL || lr0 + e < nrows) {     // (invalid columns are never written out)
              if (c > best[j]) {
                best[j] = c;
                arg[j] = lr0 + e;
                ties[j] = 1;
              } else if (c == best[j]) {
                ++ties[j];
              }
            }
          }
        }
      }
    stage_tile();
    // the other wave half holds the interleaved rows (4 half + 0..3 of every 8): combine, lowest row wins a tie
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float ob = __shfl_xor(best[j], 32, 64);
      const int oa = __shfl_xor(arg[j], 32, 64), ot = __shfl_xor(ties[j], 32, 64);
      if (ob > best[j]) {
        best[j] = ob;
        arg[j] = oa;
        ties[j] = ot;
      } else if (ob == best[j]) {
        ties[j] += ot;
        arg[j] = min(arg[j], oa);
      }
      if (half == 0) {
        const int u = wm * BN + wn * TN * 32 + j * 32 + l31;
        p_best[u] = best[j];
        p_arg[u] = arg[j];
        p_ties[u] = ties[j];
      }
    }
    __syncthreads();
    // rows (cells): max over this part's columns; confidences are >= +0, so the unsigned max of the bit patterns is the max
    {
      const int c0 = rpart * CPP, c1 = FULL ? c0 + CPP : min(c0 + CPP, ncols);
      const unsigned* Tu = reinterpret_cast<const unsigned*>(T);
      unsigned m = 0u;
#pragma unroll
      for (int c = 0; c < CPP; ++c) {
        const unsigned v = Tu[(c0 + c) * TS + rrow];
        m = max(m, (FULL || c0 + c < c1) ? v : 0u);
      }
      red_r[rpart * BM + rrow] = m;
    }
    // the tile leaves: lane = 16 bytes of a row of conf
    if (g.vec_store) {
#pragma unroll
      for (int it = 0; it < BN * (BM / 4) / NT; ++it) {
        const int u = tid + it * NT;
        const int col = u / (BM / 4), r4 = (u - col * (BM / 4)) * 4;
        if (FULL || (col < ncols && r4 + 3 < nrows)) {
          *reinterpret_cast<float4*>(g.C + (size_t)(n0 + col) * g.ldc + m0 + r4) = *reinterpret_cast<const float4*>(T + col * TS + r4);
        } else if (col < ncols) {
          for (int e = 0; e < 4; ++e)
            if (r4 + e < nrows) g.C[(size_t)(n0 + col) * g.ldc + m0 + r4 + e] = T[col * TS + r4 + e];
        }
      }
    } else {
      for (int u = tid; u < BN * BM; u += NT) {
        const int col = u / BM, r = u - col * BM;
        if (col < ncols && r < nrows) g.C[(size_t)(n0 + col) * g.ldc + m0 + r] = T[col * TS + r];
      }
    }
    __syncthreads();
    if (tid < BN) {
      if (tid < ncols) {      // wave rows 0 / 1 in ascending row order: ties keep the lowest row
        float b = p_best[tid];
        int a = p_arg[tid], t = p_ties[tid];
        const float ob = p_best[BN + tid];
        if (ob > b) {
          b = ob;
          a = p_arg[BN + tid];
          t = p_ties[BN + tid];
        } else if (ob == b) {
          t += p_ties[BN + tid];
        }
        const size_t o = (size_t)tile_m * g.N + n0 + tid;
        g.part_best[o] = b;
        g.part_arg[o] = m0 + a;
        g.part_ties[o] = t;
      }
    } else if (tid - BN < nrows) {
      const int u = tid - BN;
      g.part_rowmax[(size_t)(m0 + u) * tiles_n + tile_n] = __uint_as_float(max(red_r[u], red_r[BM + u]));
    }
  };
    if (full) body(std::true_type{});
    else body(std::false_type{});
  }
#ifdef OPP_TUNING
  if (g.dbg_ts != nullptr && lane == 0) {
    unsigned long long* o = g.dbg_ts + ((size_t)blockIdx.x * 4 + wave) * 4;
    o[0] = ts0;
    o[1] = ts1;
    o[2] = ts2;
    o[3] = __builtin_readcyclecounter();
  }
#endif
}


// ---- persistent single-sweep kernel (r06): statistics + score matrix, OPP_SS_STATS_STORE ------------------------------------------
// Same tile, same K loop and the same accumulation sequence as gemm_ss_kernel (the score tiles are bit-identical); what changes is what a
// workgroup does BETWEEN K loops.  Measured on the one-tile-per-workgroup kernel (tools/gemm_ss_probe.py, 4096 x 5000 x 256): prologue 7.1 k
// cycles (every workgroup of a generation asks for its first 72 KB at once), K loop 23.7 k, epilogue 9 k, and 1280 tiles on 512 resident
// workgroups = 2.5 generations of which the last runs alone.  Here 2 workgroups per CU stay resident and walk a static tile list
// (XCD x: its contiguous range of the strip order, workgroup idx of the XCD takes tiles idx, idx + 64, idx + 128):
//   * the first k16-stage of the NEXT tile is fetched under the epilogue of the current one: the epilogue stages the tile through LDS in
//     two halves of 64 columns (33 KB instead of 66), which leaves the third operand slot free for that stage;
//   * the per-row statistics (max, sum exp) of the two halves are merged online in the registers of the thread that owns the row
//     (fixed order: a function of the row's values only, as before -- duplicated rows still get bit-equal statistics);
//   * the second resident of a CU (the upper half of an XCD's workgroups: they are dispatched after every CU has its first) starts
//     `g.delay` x 64 cycles late, so that one workgroup's K loop runs under the other's epilogue from the first tile on, and takes the
//     shorter tile list (2 of the CU's 5 tiles at 4096 x 5000).
// LDS: 3 slots of 24 KB + 4 KB of statistics scratch that never aliases the slots = 76 KB, two workgroups per CU.
constexpr int PT_TS = BM + 4;                                   // staging row stride (floats)
constexpr int PT_HC = BN / 2;                                   // columns per staged half
constexpr int PT_MISC = NS * SLOT;                              // byte offset of the statistics scratch
constexpr size_t PT_LDS = (size_t)NS * SLOT + 4096;
static_assert(PT_HC * PT_TS * 4 <= 2 * SLOT, "a staged half tile must leave the third operand slot free");

struct SsTile {
  unsigned a_voff[A_LD], b_voff[B_LD];
  int soffA0, soffB0, m0, n0, tile_m, tile_n;
};

__global__ __launch_bounds__(NT, 2) void gemm_ss_persist_kernel(const OppGemmSS g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  // static tile list: workgroup b runs on XCD b % 8 (observed dispatch order: speed only); the XCD owns a contiguous range of the strip order
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int wgs_x = ((int)gridDim.x + 7 - xcd) >> 3;             // workgroups of this launch on the XCD
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int t_first = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq), t_count = tq + (xcd < tr ? 1 : 0);
  if (idx >= t_count) return;
  // the second resident of a CU starts late (see above); wave-uniform
  if (g.delay > 0 && 2 * idx >= wgs_x)
    for (int i = 0; i < g.delay; i += 1024) __builtin_amdgcn_s_sleep(16);     // s_sleep n = about 64 n cycles

  constexpr int RS = 8;
  auto setup = [&](int local) {
    SsTile t;
    const int tile_lin = t_first + local;
    const int strip = tile_lin / (RS * tiles_n);
    const int within = tile_lin - strip * (RS * tiles_n);
    const int strip_rows = min(RS, tiles_m - strip * RS);
    t.tile_n = within / strip_rows;
    t.tile_m = strip * RS + (within - t.tile_n * strip_rows);
    t.m0 = t.tile_m * BM;
    t.n0 = t.tile_n * BN;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int P = (wave * A_LD + i) * 64 + lane;
      const int row = P / 6, pos = P - row * 6;
      const int q = (pos + 3 * ((row >> 3) & 1)) % 6;
      t.a_voff[i] = t.m0 + row < g.M ? (unsigned)(row * g.lda + q * 16) : kOob;
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int P = (wave * B_LD + i) * 64 + lane;
      const int row = P / 6, pos = P - row * 6;
      const int q = (pos + 3 * ((row >> 3) & 1)) % 6;
      t.b_voff[i] = t.n0 + row < g.N ? (unsigned)(row * g.ldb + q * 16) : kOob;
    }
    t.soffA0 = t.m0 * g.lda;
    t.soffB0 = t.n0 * g.ldb;
    return t;
  };
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  // stage s of a tile lives in slot (s + 2) % 3: the epilogue keeps slot 2 free for stage 0 of the next tile
  auto slot_of = [](int s) { return (s + 2) % NS; };
  auto dma_item = [&](const SsTile& t, int s, int k, bool live) {
    char* base = smem + slot_of(s) * SLOT;
    if (k < A_LD) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, live ? g.a_bytes : 0, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(base + (wave * A_LD + k) * 1024), 16, (int)t.a_voff[k], t.soffA0 + s * ROWB, 0, 0);
    } else {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, live ? g.b_bytes : 0, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(base + A_BYTES + (wave * B_LD + (k - A_LD)) * 1024), 16, (int)t.b_voff[k - A_LD],
                                               t.soffB0 + s * ROWB, 0, 0);
    }
  };
  int foff[3];
  {
    const int b3 = (l31 >> 3) & 1;
#pragma unroll
    for (int p = 0; p < 3; ++p) foff[p] = 16 * ((3 * half + p + 3 * b3) % 6);
  }
  const int a_row = (wm * TM * 32 + l31) * ROWB;
  const int b_row = A_BYTES + (wn * TN * 32 + l31) * ROWB;
  u32x4 fa[2][TM][3], fb[2][TN][3];
  auto read_frags = [&](int s, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const char* base = smem + slot_of(s) * SLOT;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[set][i][p] = *reinterpret_cast<const u32x4*>(base + a_row + i * 32 * ROWB + foff[p]);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[set][j][p] = *reinterpret_cast<const u32x4*>(base + b_row + j * 32 * ROWB + foff[p]);
    }
  };
  f32x16 acc[TM][TN];
  const int ns = g.K / 16;
  SsTile cur = setup(idx);
  auto stage = [&](int s, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS) : "memory");     // stage s + 1 landed; s + 2 may be in flight
    __builtin_amdgcn_s_barrier();
    const bool live = s + 3 < ns;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    int n = 0;
#pragma unroll
    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][i][PA[pr]]),
                                                               __builtin_bit_cast(bf16x8, fb[set][j][PB[pr]]), acc[i][j], 0, 0, 0);
          if (n == 1) {
            __builtin_amdgcn_sched_barrier(0);
            read_frags(s + 1, std::integral_constant<int, set ^ 1>{});
            __builtin_amdgcn_sched_barrier(0);
          }
          if (n % 4 == 3 && n / 4 < LPS) {
            dma_item(cur, s + 3, n / 4, live);
            __builtin_amdgcn_sched_barrier(0);
          }
          ++n;
        }
    __builtin_amdgcn_sched_barrier(0);
  };

  float* T = reinterpret_cast<float*>(smem);                                  // [PT_HC][PT_TS]: slots 0 and (part of) 1
  float* red_cmax = reinterpret_cast<float*>(smem + PT_MISC);                 // [WM][BN]
  float* red_csum = red_cmax + WM * BN;                                       // [WM][BN]
  float* red_rm = red_csum + WM * BN;                                         // [2][BM]
  float* red_rs = red_rm + 2 * BM;                                            // [2][BM]
  auto vmax = [](float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  };
  constexpr int CPS = PT_HC / 2;

  // prologue of the first tile: three stages in flight
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int k = 0; k < LPS; ++k) dma_item(cur, s, k, s < ns);

  for (int local = idx;;) {
#ifdef OPP_TUNING
    const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");            // stage 0 landed (and everything older: the last tile's stores)
    __builtin_amdgcn_s_barrier();
    read_frags(0, std::integral_constant<int, 0>{});
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#ifdef OPP_TUNING
    const unsigned long long ts1 = __builtin_readcyclecounter();
#endif
    for (int s = 0; s < ns; s += 2) {
      stage(s, std::integral_constant<int, 0>{});
      stage(s + 1, std::integral_constant<int, 1>{});
    }
#ifdef OPP_TUNING
    const unsigned long long ts2 = __builtin_readcyclecounter();
#endif
    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
    // (thread coordinates re-derived from an opaque copy of the thread index: hoisted out of the tile loop, the epilogue's address
    // arithmetic would stay live across the K loop -- 120 spilled registers in the first build)
    int te = tid;
    asm volatile("" : "+v"(te));
    const int e_wave = __builtin_amdgcn_readfirstlane(te >> 6);
    const int e_wm = e_wave / WN, e_wn = e_wave % WN, e_half = (te >> 5) & 1, e_l31 = te & 31;
    const int e_rrow = te & (BM - 1), e_rsub = te >> 7;
    if ((g.out_mul != 1.f) || (g.out_div != 1.f)) {
      const float rd = 1.0f / g.out_div;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = div_invariant(acc[i][j][r] * g.out_mul, g.out_div, rd);
    }
    const int m0 = cur.m0, n0 = cur.n0;
    auto row_of = [&](int i, int r) { return e_wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * e_half; };
    if (g.row_mask != nullptr) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float mk = g.row_mask[min(m0 + row_of(i, r), g.M - 1)];
          const float add = mk == 0.f ? -1e9f : 0.f;
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j][r] += add;
        }
    }
    const int nrows = min(BM, g.M - m0), ncols = min(BN, g.N - n0);
    const bool full = nrows == BM && ncols == BN;
    const int next_local = local + wgs_x;
    const bool has_next = next_local < t_count;
    SsTile nxt = cur;
    auto body = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
      // columns, from the accumulators: max over this wave's 64 rows, then over the two wave rows (the arithmetic of gemm_ss_kernel)
      float cmx[TN], csm[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) m = vmax(m, (FULL || row_of(i, r) < nrows) ? acc[i][j][r] : -INFINITY);
        m = vmax(m, __shfl_xor(m, 32, 64));
        if (e_half == 0) red_cmax[e_wm * BN + e_wn * TN * 32 + j * 32 + e_l31] = m;
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");             // the zero-sized tail DMAs and the last fragment reads
      __syncthreads();                                                         // every slot is dead; the column maxima are visible
      if (has_next) {                                                          // stage 0 of the next tile -> slot 2, under this epilogue
        nxt = setup(next_local);
#pragma unroll
        for (int k = 0; k < LPS; ++k) dma_item(nxt, 0, k, true);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) cmx[j] = vmax(red_cmax[e_wn * TN * 32 + j * 32 + e_l31], red_cmax[BN + e_wn * TN * 32 + j * 32 + e_l31]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) sm += (FULL || row_of(i, r) < nrows) ? __expf(acc[i][j][r] - cmx[j]) : 0.f;
        csm[j] = sm + __shfl_xor(sm, 32, 64);
        if (e_half == 0) red_csum[e_wm * BN + e_wn * TN * 32 + j * 32 + e_l31] = csm[j];
      }
      if (e_wm == 0 && e_half == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (e_wn * TN * 32 + j * 32 + e_l31 < ncols) g.stat_colmax[(size_t)cur.tile_m * g.N + n0 + e_wn * TN * 32 + j * 32 + e_l31] = cmx[j];
      }
      // rows: the tile is staged 64 columns at a time; the thread (row, 32 columns) keeps a running (max, sum exp) of its row
      float RM = -INFINITY, RSUM = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (e_wn == h) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                const int col = j * 32 + e_l31, row = e_wm * TM * 32 + i * 32 + 8 * q4 + 4 * e_half;
                *reinterpret_cast<float4*>(T + col * PT_TS + row) =
                    make_float4(acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]);
              }
        }
        __syncthreads();
        {
          const int c0 = e_rsub * CPS;
          float rv[CPS];
          float m = -INFINITY;
#pragma unroll
          for (int c = 0; c < CPS; ++c) {
            rv[c] = T[(c0 + c) * PT_TS + e_rrow];
            m = vmax(m, (FULL || h * PT_HC + c0 + c < ncols) ? rv[c] : -INFINITY);
          }
          float sacc = 0.f;
#pragma unroll
          for (int c = 0; c < CPS; ++c) sacc += (FULL || h * PT_HC + c0 + c < ncols) ? __expf(rv[c] - m) : 0.f;
          // online merge, fixed order (e_half 0 then e_half 1): exp(-inf - finite) = 0 covers the empty side
          const float M2 = vmax(RM, m);
          if (FULL || M2 > -INFINITY) RSUM = RSUM * __expf(RM - M2) + sacc * __expf(m - M2);
          RM = M2;
        }
        // the e_half leaves: lane = 16 bytes of a row of out[col][row] (the launcher takes this kernel for 16-byte aligned outputs only)
#pragma unroll
        for (int it = 0; it < PT_HC * (BM / 4) / NT; ++it) {
          const int u = te + it * NT;
          const int col = u / (BM / 4), r4 = (u - col * (BM / 4)) * 4;
          const int gc = h * PT_HC + col;
          if (FULL || (gc < ncols && r4 + 3 < nrows)) {
            *reinterpret_cast<float4*>(g.C + (size_t)(n0 + gc) * g.ldc + m0 + r4) = *reinterpret_cast<const float4*>(T + col * PT_TS + r4);
          } else if (gc < ncols) {
            for (int e = 0; e < 4; ++e)
              if (r4 + e < nrows) g.C[(size_t)(n0 + gc) * g.ldc + m0 + r4 + e] = T[col * PT_TS + r4 + e];
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                                       // the staged e_half is consumed
      }
      if (has_next) {                                                          // stages 1, 2 of the next tile -> slots 0, 1
#pragma unroll
        for (int s = 1; s < NS; ++s)
#pragma unroll
          for (int k = 0; k < LPS; ++k) dma_item(nxt, s, k, s < ns);
      }
      red_rm[e_rsub * BM + e_rrow] = RM;
      red_rs[e_rsub * BM + e_rrow] = RSUM;
      __syncthreads();
      if (te < BM) {
        if (te < nrows) {
          const float ma = red_rm[te], mb = red_rm[BM + te];
          const float M2 = vmax(ma, mb);
          const size_t o = (size_t)(m0 + te) * tiles_n + cur.tile_n;
          g.stat_rowmax[o] = M2;
          g.stat_rowsum[o] = red_rs[te] * __expf(ma - M2) + red_rs[BM + te] * __expf(mb - M2);
        }
      } else if (te - BM < ncols) {
        const int u = te - BM;
        g.stat_colsum[(size_t)cur.tile_m * g.N + n0 + u] = red_csum[u] + red_csum[BN + u];
      }
    };
    if (full) body(std::true_type{});
    else body(std::false_type{});
#ifdef OPP_TUNING
    if (g.dbg_ts != nullptr && lane == 0) {
      unsigned long long* o = g.dbg_ts + ((size_t)(t_first + local) * 4 + wave) * 4;
      unsigned hw = 0, xcc = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      o[0] = wave == 1 ? (((unsigned long long)xcc << 32) | hw) : ts0;   // wave 1 reports where the workgroup ran instead of its start stamp
      o[1] = ts1;
      o[2] = ts2;
      o[3] = __builtin_readcyclecounter();
    }
#endif
    if (!has_next) break;
    cur = nxt;
    local = next_local;
  }
}

// ---- three residents per CU (r06): statistics + score matrix, OPP_SS_STATS_STORE ---------------------------------------------------
// The timeline of the one-tile kernel (profiles/r06_ss_timeline.txt) says a tile is 12.3 k cycles of MFMAs inside 43 k: 6.3 k of prologue
// latency, 12.4 k of epilogue, and a K loop that runs at half rate whenever the CU's other resident is in its own.  Staggering, static tile
// lists and wave priorities did not change that; what does is a THIRD resident: with three workgroups per CU one of them is in its K loop far
// more often, and the critical path of a CU's five tiles is two tiles long instead of three.  What it costs: LDS for TWO k16-stages instead of
// three (2 x 24 KB + 4 KB of statistics scratch = 52 KB, three workgroups = 156 KB) and <= 168 registers:
//   * ONE operand fragment set, read right behind the stage barrier (the other residents' MFMAs run under that LDS latency);
//   * ONE stage of LDS-DMA in flight: stage s + 1 goes into the slot stage s - 1 was read from, issued one piece per four MFMAs of stage s
//     (every wave read its stage s - 1 fragments before the barrier that opens stage s), and is awaited with vmcnt(0) at the next barrier;
//   * the epilogue of the persistent kernel: the tile staged through LDS in two halves of 64 columns (33 KB), row statistics merged online.
// Same tile, same accumulation sequence: the score tiles are bit-identical to gemm_ss_kernel's; the row statistics are the persistent kernel's
// (a function of the row's values only).
constexpr int R3_NS = 2;
#ifndef R3_DMA_EVERY
#define R3_DMA_EVERY 4         // one DMA piece behind every R3_DMA_EVERY-th MFMA of a stage (6 pieces, 24 MFMAs); 1 / 2 / 3 measured the same (134-138 us)
#endif
constexpr size_t R3_LDS = (size_t)R3_NS * SLOT + 4096;
static_assert(PT_HC * PT_TS * 4 <= R3_NS * SLOT, "a staged half tile must fit the two operand slots");

__global__ __launch_bounds__(NT, 3) void gemm_ss_res3_kernel(const OppGemmSS g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef OPP_TUNING
  const unsigned long long ts0 = __builtin_readcyclecounter();
  unsigned long long ts1 = 0, ts2 = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  // one tile per workgroup, the XCD-aware strip order of gemm_ss_kernel.  (Static tile lists -- 2 + 2 + 1 tiles per CU on 768 resident workgroups
  // -- were measured as well: a tile costs 60 k cycles with three residents, the two-tile critical path is as long as before, matcher 143 vs 143 us;
  // the dynamic grid lets whichever workgroup slot frees first take the next tile: -4 us, profiles/r06_ss_res3_ab.txt.)
  int tile_lin = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    tile_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int RS = 8;
  const int strip = tile_lin / (RS * tiles_n);
  const int within = tile_lin - strip * (RS * tiles_n);
  const int strip_rows = min(RS, tiles_m - strip * RS);
  const int tile_n = within / strip_rows;
  const int tile_m = strip * RS + (within - tile_n * strip_rows);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  unsigned a_voff[A_LD], b_voff[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int P = (wave * A_LD + i) * 64 + lane;
    const int row = P / 6, pos = P - row * 6;
    const int q = (pos + 3 * ((row >> 3) & 1)) % 6;
    a_voff[i] = m0 + row < g.M ? (unsigned)(row * g.lda + q * 16) : kOob;
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int P = (wave * B_LD + i) * 64 + lane;
    const int row = P / 6, pos = P - row * 6;
    const int q = (pos + 3 * ((row >> 3) & 1)) % 6;
    b_voff[i] = n0 + row < g.N ? (unsigned)(row * g.ldb + q * 16) : kOob;
  }
  const int soffA0 = m0 * g.lda, soffB0 = n0 * g.ldb;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  auto dma_item = [&](int s, int k, bool live) {
    char* base = smem + (s & 1) * SLOT;
    if (k < A_LD) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, live ? g.a_bytes : 0, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(base + (wave * A_LD + k) * 1024), 16, (int)a_voff[k], soffA0 + s * ROWB, 0, 0);
    } else {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, live ? g.b_bytes : 0, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(base + A_BYTES + (wave * B_LD + (k - A_LD)) * 1024), 16, (int)b_voff[k - A_LD],
                                               soffB0 + s * ROWB, 0, 0);
    }
  };
  int foff[3];
  {
    const int b3 = (l31 >> 3) & 1;
#pragma unroll
    for (int p = 0; p < 3; ++p) foff[p] = 16 * ((3 * half + p + 3 * b3) % 6);
  }
  const int a_row = (wm * TM * 32 + l31) * ROWB;
  const int b_row = A_BYTES + (wn * TN * 32 + l31) * ROWB;
  u32x4 fa[TM][3], fb[TN][3];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ns = g.K / 16;

  // one stage: my pieces of stage s landed (vmcnt(0): nothing else is in flight) -> barrier (stage s is complete and visible, and every wave has
  // read its fragments of stage s - 1) -> fragments of stage s -> 24 MFMAs with the six pieces of stage s + 1 behind them
  auto stage = [&](int s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* base = smem + (s & 1) * SLOT;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i][p] = *reinterpret_cast<const u32x4*>(base + a_row + i * 32 * ROWB + foff[p]);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j][p] = *reinterpret_cast<const u32x4*>(base + b_row + j * 32 * ROWB + foff[p]);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool live = s + 1 < ns;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    int n = 0;
#pragma unroll
    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][PA[pr]]), __builtin_bit_cast(bf16x8, fb[j][PB[pr]]),
                                                               acc[i][j], 0, 0, 0);
          if (n % R3_DMA_EVERY == R3_DMA_EVERY - 1 && n / R3_DMA_EVERY < LPS) {
            dma_item(s + 1, n / R3_DMA_EVERY, live);
            __builtin_amdgcn_sched_barrier(0);
          }
          ++n;
        }
    __builtin_amdgcn_sched_barrier(0);
  };

#pragma unroll
  for (int k = 0; k < LPS; ++k) dma_item(0, k, true);
#ifdef OPP_TUNING
  ts1 = __builtin_readcyclecounter();
#endif
  for (int s = 0; s < ns; ++s) stage(s);
#ifdef OPP_TUNING
  ts2 = __builtin_readcyclecounter();
#endif

  // ---- epilogue (gemm_ss_persist_kernel's, without the next tile) --------------------------------------------------------------------
  // (thread coordinates re-derived from an opaque copy of the thread index: hoisted out of the tile loop, the epilogue's address arithmetic
  // stays live across the K loop and spills)
  int te = tid;
  asm volatile("" : "+v"(te));
  const int e_wave = __builtin_amdgcn_readfirstlane(te >> 6);
  const int e_wm = e_wave / WN, e_wn = e_wave % WN, e_half = (te >> 5) & 1, e_l31 = te & 31;
  const int e_rrow = te & (BM - 1), e_rsub = te >> 7;
  if ((g.out_mul != 1.f) || (g.out_div != 1.f)) {
    const float rd = 1.0f / g.out_div;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = div_invariant(acc[i][j][r] * g.out_mul, g.out_div, rd);
  }
  auto row_of = [&](int i, int r) { return e_wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * e_half; };
  if (g.row_mask != nullptr) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float mk = g.row_mask[min(m0 + row_of(i, r), g.M - 1)];
        const float add = mk == 0.f ? -1e9f : 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j][r] += add;
      }
  }
  const int nrows = min(BM, g.M - m0), ncols = min(BN, g.N - n0);
  const bool full = nrows == BM && ncols == BN;
  float* T = reinterpret_cast<float*>(smem);                                  // [PT_HC][PT_TS] over the two (dead) operand slots
  float* red_cmax = reinterpret_cast<float*>(smem + R3_NS * SLOT);            // [WM][BN]
  float* red_csum = red_cmax + WM * BN;                                       // [WM][BN]
  float* red_rm = red_csum + WM * BN;                                         // [2][BM]
  float* red_rs = red_rm + 2 * BM;                                            // [2][BM]
  auto vmax = [](float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  };
  constexpr int CPS = PT_HC / 2;
  auto body = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    float cmx[TN], csm[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float m = -INFINITY;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = vmax(m, (FULL || row_of(i, r) < nrows) ? acc[i][j][r] : -INFINITY);
      m = vmax(m, __shfl_xor(m, 32, 64));
      if (e_half == 0) red_cmax[e_wm * BN + e_wn * TN * 32 + j * 32 + e_l31] = m;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");               // the zero-sized tail DMAs and the last fragment reads
    __syncthreads();                                                           // both slots are dead; the column maxima are visible
#pragma unroll
    for (int j = 0; j < TN; ++j) cmx[j] = vmax(red_cmax[e_wn * TN * 32 + j * 32 + e_l31], red_cmax[BN + e_wn * TN * 32 + j * 32 + e_l31]);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += (FULL || row_of(i, r) < nrows) ? __expf(acc[i][j][r] - cmx[j]) : 0.f;
      csm[j] = sm + __shfl_xor(sm, 32, 64);
      if (e_half == 0) red_csum[e_wm * BN + e_wn * TN * 32 + j * 32 + e_l31] = csm[j];
    }
    if (e_wm == 0 && e_half == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        if (e_wn * TN * 32 + j * 32 + e_l31 < ncols) g.stat_colmax[(size_t)tile_m * g.N + n0 + e_wn * TN * 32 + j * 32 + e_l31] = cmx[j];
    }
    float RM = -INFINITY, RSUM = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (e_wn == h) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const int col = j * 32 + e_l31, row = e_wm * TM * 32 + i * 32 + 8 * q4 + 4 * e_half;
              *reinterpret_cast<float4*>(T + col * PT_TS + row) =
                  make_float4(acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]);
            }
      }
      __syncthreads();
      {
        const int c0 = e_rsub * CPS;
        float rv[CPS];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < CPS; ++c) {
          rv[c] = T[(c0 + c) * PT_TS + e_rrow];
          m = vmax(m, (FULL || h * PT_HC + c0 + c < ncols) ? rv[c] : -INFINITY);
        }
        float sacc = 0.f;
#pragma unroll
        for (int c = 0; c < CPS; ++c) sacc += (FULL || h * PT_HC + c0 + c < ncols) ? __expf(rv[c] - m) : 0.f;
        const float M2 = vmax(RM, m);
        if (FULL || M2 > -INFINITY) RSUM = RSUM * __expf(RM - M2) + sacc * __expf(m - M2);
        RM = M2;
      }
#pragma unroll
      for (int it = 0; it < PT_HC * (BM / 4) / NT; ++it) {
        const int u = te + it * NT;
        const int col = u / (BM / 4), r4 = (u - col * (BM / 4)) * 4;
        const int gc = h * PT_HC + col;
        if (FULL || (gc < ncols && r4 + 3 < nrows)) {
          *reinterpret_cast<float4*>(g.C + (size_t)(n0 + gc) * g.ldc + m0 + r4) = *reinterpret_cast<const float4*>(T + col * PT_TS + r4);
        } else if (gc < ncols) {
          for (int e = 0; e < 4; ++e)
            if (r4 + e < nrows) g.C[(size_t)(n0 + gc) * g.ldc + m0 + r4 + e] = T[col * PT_TS + r4 + e];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();                                                         // the staged e_half is consumed
    }
    red_rm[e_rsub * BM + e_rrow] = RM;
    red_rs[e_rsub * BM + e_rrow] = RSUM;
    __syncthreads();
    if (te < BM) {
      if (te < nrows) {
        const float ma = red_rm[te], mb = red_rm[BM + te];
        const float M2 = vmax(ma, mb);
        const size_t o = (size_t)(m0 + te) * tiles_n + tile_n;
        g.stat_rowmax[o] = M2;
        g.stat_rowsum[o] = red_rs[te] * __expf(ma - M2) + red_rs[BM + te] * __expf(mb - M2);
      }
    } else if (te - BM < ncols) {
      const int u = te - BM;
      g.stat_colsum[(size_t)tile_m * g.N + n0 + u] = red_csum[u] + red_csum[BN + u];
    }
  };
  if (full) body(std::true_type{});
  else body(std::false_type{});
#ifdef OPP_TUNING
  if (g.dbg_ts != nullptr && lane == 0) {
    unsigned long long* o = g.dbg_ts + ((size_t)tile_lin * 4 + wave) * 4;
    o[0] = ts0;
    o[1] = ts1;
    o[2] = ts2;
    o[3] = __builtin_readcyclecounter();
  }
#endif
}

#ifdef OPP_TUNING
unsigned long long* g_ss_dbg_ts = nullptr;
int g_ss_dbg_mode = 0;
#endif

constexpr int kPersistDelay = 12288;      // cycles the second resident of a CU starts late (about half a tile); OPP_SS_DELAY overrides

// compute units of the current device (cached per device: a process may drive several GPUs)
int opp_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (!cached[dev]) {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cached[dev] = cus > 0 ? cus : 256;
  }
  return cached[dev];
}

template <int MODE>
int launch_ss(const OppGemmSS& g, hipStream_t stream, int symbol) {
  constexpr size_t lds = (size_t)NS * SLOT;
  auto k = gemm_ss_kernel<MODE>;
  static OppLdsOnce lds_once;            // per device (opp_common.h)
  opp_lds_opt_in(reinterpret_cast<const void*>(k), lds, lds_once);
  const int tiles = opp_cdiv(g.M, BM) * opp_cdiv(g.N, BN);
  OppProfScope prof(symbol, stream, 2.0 * (double)g.M * (double)g.N * (double)g.K);
  hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), lds, stream, g);
  OPP_CHECK_LAUNCH("gemm_ss_kernel");
  return OPP_OK;
}

}  // namespace

void opp_gemm_ss_debug_timestamps(void* buf, int mode) {
#ifdef OPP_TUNING
  g_ss_dbg_ts = static_cast<unsigned long long*>(buf);
  g_ss_dbg_mode = mode;
#else
  (void)buf;
  (void)mode;
#endif
}
int opp_gemm_ss_tile_rows() { return BM; }
int opp_gemm_ss_tile_cols() { return BN; }

int opp_gemm_ss(const OppGemmSS& g_in, hipStream_t stream) {
  OppGemmSS g = g_in;
  OPP_CHECK_ARG(g.A && g.B && g.M > 0 && g.N > 0 && g.K > 0 && g.K % 32 == 0, "gemm_ss: bad operands / K %% 32 (M %d N %d K %d)", g.M, g.N, g.K);
  OPP_CHECK_ARG(g.lda % 16 == 0 && g.ldb % 16 == 0 && g.lda >= g.K * 6 && g.ldb >= g.K * 6, "gemm_ss: operand row strides are bytes, >= 6 K, 16-byte multiples");
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  OPP_CHECK_ARG(al16(g.A) && al16(g.B), "gemm_ss: operands must be 16-byte aligned");
  OPP_CHECK_ARG((size_t)g.M * g.lda < (1ull << 31) && (size_t)g.N * g.ldb < (1ull << 31), "gemm_ss: operand too large for buffer addressing");
  g.a_bytes = (int)((size_t)g.M * g.lda);
  g.b_bytes = (int)((size_t)g.N * g.ldb);
#ifdef OPP_TUNING
  g.dbg_ts = (g_ss_dbg_mode == 0 || g_ss_dbg_mode == g.mode) ? g_ss_dbg_ts : nullptr;
#endif
  if (g.mode == OPP_SS_STATS) {
    OPP_CHECK_ARG(g.stat_rowmax && g.stat_rowsum && g.stat_colmax && g.stat_colsum, "gemm_ss: statistics outputs missing");
    return launch_ss<OPP_SS_STATS>(g, stream, OPP_PROF_SCORE_SWEEP1);
  }
  if (g.mode == OPP_SS_STATS_STORE) {
    OPP_CHECK_ARG(g.stat_rowmax && g.stat_rowsum && g.stat_colmax && g.stat_colsum && g.C && g.ldc >= g.M, "gemm_ss: statistics / score outputs missing");
    OPP_CHECK_ARG((size_t)g.N * (size_t)g.ldc < (1ull << 31), "gemm_ss: output too large for 32-bit indexing");
    g.vec_store = (al16(g.C) && g.ldc % 4 == 0) ? 1 : 0;
    // persistent kernel (two resident workgroups per CU walk static tile lists): OPP_SS_PERSIST=1.  Measured r06 (profiles/r06_ss_persistent_ab.txt,
    // r06_ss_timeline.txt): the prologue disappears (6.3 k -> 0.4 k cycles per tile) but the tile costs the same 42-43 k cycles -- the K loops of the
    // two residents overlap 46-56 % of the time and slow each other down exactly as in the one-tile kernel -- so the matcher is 140 us either way
    // and the default stays the one-tile kernel.  (Static wave priorities for the K loops -- a loop outranks the partner's epilogue, the second
    // resident's loop outranks the first's -- were tried on top, profiles/r06_ss_prio_ab.txt: 142-148 us, no better; removed.)
    static const int persist_env = getenv("OPP_SS_PERSIST") ? atoi(getenv("OPP_SS_PERSIST")) : 0;
    static const int delay_env = getenv("OPP_SS_DELAY") ? atoi(getenv("OPP_SS_DELAY")) : kPersistDelay;
    const int tiles = opp_cdiv(g.M, BM) * opp_cdiv(g.N, BN);
    // default since r06: three resident workgroups per CU (gemm_ss_res3_kernel; OPP_SS_RES3=0 selects the two-resident one-tile kernel).  Measured
    // (profiles/r06_ss_res3_ab.txt): matcher 138.3 -> 135.7 us, matrix pipe busy 0.383 -> 0.401 inside the forward, forward +0.6 % with one
    // forward in flight, unchanged with four
    const char* res3_s = getenv("OPP_SS_RES3");            // (read per call: the tests switch it inside one process)
    const int res3_env = res3_s ? atoi(res3_s) : 1;
    if (res3_env && !persist_env && g.vec_store) {
      auto k = gemm_ss_res3_kernel;
      static OppLdsOnce lds_once;
      opp_lds_opt_in(reinterpret_cast<const void*>(k), R3_LDS, lds_once);
      OppProfScope prof(OPP_PROF_SCORE_SS, stream, 2.0 * (double)g.M * (double)g.N * (double)g.K);
      hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), R3_LDS, stream, g);
      OPP_CHECK_LAUNCH("gemm_ss_res3_kernel");
      return OPP_OK;
    }
    if (persist_env && tiles > 8 && g.vec_store) {
      const int slots = 2 * opp_cu_count();
      g.delay = tiles > slots ? delay_env : 0;         // (one tile per workgroup: nothing to de-phase)
      auto k = gemm_ss_persist_kernel;
      static OppLdsOnce lds_once;
      opp_lds_opt_in(reinterpret_cast<const void*>(k), PT_LDS, lds_once);
      OppProfScope prof(OPP_PROF_SCORE_SS, stream, 2.0 * (double)g.M * (double)g.N * (double)g.K);
      hipLaunchKernelGGL(k, dim3(tiles < slots ? tiles : slots), dim3(NT), PT_LDS, stream, g);
      OPP_CHECK_LAUNCH("gemm_ss_persist_kernel");
      return OPP_OK;
    }
    return launch_ss<OPP_SS_STATS_STORE>(g, stream, OPP_PROF_SCORE_SS);
  }
  if (g.mode == OPP_SS_CONF) {
    OPP_CHECK_ARG(g.C && g.ldc >= g.M && g.rstat_max && g.rstat_sum && g.cstat_max && g.cstat_sum && g.part_best && g.part_arg && g.part_ties &&
                      g.part_rowmax, "gemm_ss: confidence sweep arguments missing");
    OPP_CHECK_ARG((size_t)g.N * (size_t)g.ldc < (1ull << 31), "gemm_ss: output too large for 32-bit indexing");
    g.vec_store = (al16(g.C) && g.ldc % 4 == 0) ? 1 : 0;
    return launch_ss<OPP_SS_CONF>(g, stream, OPP_PROF_SCORE_SWEEP2);
  }
  opp_set_error("gemm_ss: unknown mode %d", g.mode);
  return OPP_ERR_INVALID;
}

#ifdef OPP_TUNING
// tuning builds only: the statistics sweep on caller-given split operands of any K (loop-efficiency probe, tools/gemm_ss_probe.py)
extern "C" int opp_debug_gemm_ss_stats(const void* A, const void* B, int M, int N, int K, float* stats, void* stream) {
  OppGemmSS g;
  g.A = A;
  g.B = B;
  g.lda = K * 6;
  g.ldb = K * 6;
  g.M = M;
  g.N = N;
  g.K = K;
  g.mode = OPP_SS_STATS;
  const size_t tn = opp_cdiv(N, BN), tm = opp_cdiv(M, BM);
  g.stat_rowmax = stats;
  g.stat_rowsum = g.stat_rowmax + (size_t)M * tn;
  g.stat_colmax = g.stat_rowsum + (size_t)M * tn;
  g.stat_colsum = g.stat_colmax + tm * (size_t)N;
  return opp_gemm_ss(g, (hipStream_t)stream);
}
#endif
