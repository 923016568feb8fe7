// Coarse matching: dual-softmax confidence matrix + mutual-nearest-neighbour selection.
//
// Reference: CoarseMatching.forward / get_coarse_match / mask_border
//            src/models/OnePosePlus/utils/coarse_matching.py:10-20, :76-123, :125-242
//
// The similarity matrix S [N][L] (N 3D points x L = hc*wc image cells) is produced by the
// score GEMM (gemm_mfma.hip, temperature scaling fused).  Everything here is HBM/L2-bound
// row / column sweeps of that matrix:
//   conf[i][j] = softmax_over_i(S)[i][j] * softmax_over_j(S)[i][j]        (:115)
//   keep (i,j) if conf > thr, cell not in the first `border_rm` rows/cols (quirk q1),
//   conf == rowmax_i(conf) and conf == colmax_j(conf)                      (:145-163)
//   matches ordered by ascending i, first surviving j per row (quirk q9)   (:166-172)
// All equality tests are done on the conf values this code itself wrote, so the decision is
// self-consistent (SURVEY.md §7 "hard parts").
#include "opp_internal.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_add(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int VEC>
struct Ld;
template <>
struct Ld<4> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <>
struct Ld<1> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
};

// ---- per-row max and sum exp(s - max): one wave per row -----------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ S, int N, int L,
                                                        float* __restrict__ rmax, float* __restrict__ rsum) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* s = S + (size_t)row * L;
  float m = -INFINITY;
  for (int j = lane * VEC; j < L; j += 64 * VEC) {
    float v[VEC];
    Ld<VEC>::load(s + j, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) m = fmaxf(m, v[e]);
  }
  m = wave_max(m);
  float acc = 0.f;
  for (int j = lane * VEC; j < L; j += 64 * VEC) {
    float v[VEC];
    Ld<VEC>::load(s + j, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc += expf(v[e] - m);
  }
  acc = wave_add(acc);
  if (lane == 0) {
    rmax[row] = m;
    rsum[row] = acc;
  }
}

// ---- column sweeps: thread per column, block = 256 columns x one chunk of rows -------------
// MODE 0: partial max ; MODE 1: partial sum exp(s - cmax[col])
template <int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ S, int N, int L, int chunk_rows,
                                                          const float* __restrict__ cmax, float* __restrict__ part) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= L) return;
  const int r0 = blockIdx.y * chunk_rows;
  const int r1 = min(N, r0 + chunk_rows);
  float acc = MODE == 0 ? -INFINITY : 0.f;
  const float cm = MODE == 1 ? cmax[col] : 0.f;
  const float* p = S + (size_t)r0 * L + col;
  for (int r = r0; r < r1; ++r, p += L) {
    const float v = *p;
    if (MODE == 0) acc = fmaxf(acc, v);
    else acc += expf(v - cm);
  }
  part[(size_t)blockIdx.y * L + col] = acc;
}

template <int MODE>
__global__ void col_reduce_kernel(const float* __restrict__ part, int chunks, int L, float* __restrict__ out, float* __restrict__ out_rcp) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= L) return;
  float acc = MODE == 0 ? -INFINITY : 0.f;
  for (int c = 0; c < chunks; ++c) {
    const float v = part[(size_t)c * L + col];
    if (MODE == 0) acc = fmaxf(acc, v);
    else acc += v;
  }
  out[col] = acc;
  if (MODE == 1 && out_rcp != nullptr) out_rcp[col] = __frcp_rn(acc);   // the per-column factor of conf_value(), once per column
}

// ---- merge of the per-tile (max, sum exp) partials produced by the score GEMM epilogue -----------
// rows: part [N][T] ; cols: part [T][L] ; out max / sum with the global max as reference
// rows: 8 lanes per row, lane u takes tiles u, u+8, ...; pairwise combine in a fixed butterfly order
__device__ __forceinline__ void row_merge_body(int bid, const float* __restrict__ pmax, const float* __restrict__ psum, int N, int T,
                                               float* __restrict__ omax, float* __restrict__ osum, float* __restrict__ orcp) {
  const int i = bid * 32 + (threadIdx.x >> 3);
  const int u = threadIdx.x & 7;
  float m = -INFINITY, s = 0.f;
  if (i < N) {
    for (int t = u; t < T; t += 8) m = fmaxf(m, pmax[(size_t)i * T + t]);
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (i < N) {
    for (int t = u; t < T; t += 8) s += psum[(size_t)i * T + t] * expf(pmax[(size_t)i * T + t] - m);
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) s += __shfl_xor(s, o, 64);
  if (i < N && u == 0) {
    omax[i] = m;
    osum[i] = s;
    if (orcp != nullptr) orcp[i] = __frcp_rn(s);
  }
}
__global__ __launch_bounds__(256) void row_merge_kernel(const float* __restrict__ pmax, const float* __restrict__ psum, int N, int T,
                                                        float* __restrict__ omax, float* __restrict__ osum, float* __restrict__ orcp) {
  row_merge_body(blockIdx.x, pmax, psum, N, T, omax, osum, orcp);
}
// columns: block = 64 columns x 4 tile groups (group g takes tiles g, g+4, ...), combined through LDS in group order
__device__ __forceinline__ void col_merge_body(int bid, float (*red)[64], const float* __restrict__ pmax, const float* __restrict__ psum, int L, int T,
                                               float* __restrict__ omax, float* __restrict__ osum, float* __restrict__ orcp) {
  const int c = threadIdx.x & 63, gq = threadIdx.x >> 6;
  const int j = bid * 64 + c;
  float m = -INFINITY;
  if (j < L)
    for (int t = gq; t < T; t += 4) m = fmaxf(m, pmax[(size_t)t * L + j]);
  red[gq][c] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0][c], red[1][c]), fmaxf(red[2][c], red[3][c]));
  __syncthreads();
  float s = 0.f;
  if (j < L)
    for (int t = gq; t < T; t += 4) s += psum[(size_t)t * L + j] * expf(pmax[(size_t)t * L + j] - m);
  red[gq][c] = s;
  __syncthreads();
  if (gq == 0 && j < L) {
    const float tot = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
    omax[j] = m;
    osum[j] = tot;
    if (orcp != nullptr) orcp[j] = __frcp_rn(tot);
  }
}
__global__ __launch_bounds__(256) void col_merge_kernel(const float* __restrict__ pmax, const float* __restrict__ psum, int L, int T,
                                                        float* __restrict__ omax, float* __restrict__ osum, float* __restrict__ orcp) {
  __shared__ float red[4][64];
  col_merge_body(blockIdx.x, red, pmax, psum, L, T, omax, osum, orcp);
}
// both merges of the single-sweep matcher in ONE launch (r05): blocks [0, row_blocks) merge the row partials, the rest the column partials
__global__ __launch_bounds__(256) void rowcol_merge_kernel(int row_blocks, const float* __restrict__ rpmax, const float* __restrict__ rpsum, int Nr, int Tr,
                                                           float* __restrict__ romax, float* __restrict__ rosum, float* __restrict__ rorcp,
                                                           const float* __restrict__ cpmax, const float* __restrict__ cpsum, int Nc, int Tc,
                                                           float* __restrict__ comax, float* __restrict__ cosum, float* __restrict__ corcp) {
  __shared__ float red[4][64];
  if ((int)blockIdx.x < row_blocks) row_merge_body(blockIdx.x, rpmax, rpsum, Nr, Tr, romax, rosum, rorcp);      // block-uniform branch
  else col_merge_body(blockIdx.x - row_blocks, red, cpmax, cpsum, Nc, Tc, comax, cosum, corcp);
}
// column max over `chunks` partial rows [chunks][L] (max is order-independent): 64 columns x 4 chunk groups
__global__ __launch_bounds__(256) void col_max_reduce_kernel(const float* __restrict__ part, int chunks, int L,
                                                             float* __restrict__ out) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, gq = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  float m = -INFINITY;
  if (j < L) {
    float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;   // four loads in flight per thread
    int t = gq;
    for (; t + 12 < chunks; t += 16) {
      m0 = fmaxf(m0, part[(size_t)t * L + j]);
      m1 = fmaxf(m1, part[(size_t)(t + 4) * L + j]);
      m2 = fmaxf(m2, part[(size_t)(t + 8) * L + j]);
      m3 = fmaxf(m3, part[(size_t)(t + 12) * L + j]);
    }
    for (; t < chunks; t += 4) m0 = fmaxf(m0, part[(size_t)t * L + j]);
    m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
  }
  red[gq][c] = m;
  __syncthreads();
  if (gq == 0 && j < L) out[j] = fmaxf(fmaxf(red[0][c], red[1][c]), fmaxf(red[2][c], red[3][c]));
}

// conf = softmax_dim1 * softmax_dim2 (coarse_matching.py:115) for one entry:
//   exp(v - cmax)/csum * exp(v - rmax)/rsum  ==  exp((v - cmax) + (v - rmax)) * (1/csum) * (1/rsum)
// one v_exp_f32 and two multiplies per entry instead of two libm exps and two IEEE divisions (the
// sweep was VALU-bound, not HBM-bound); both exponents are <= 0, relative deviation from the two-softmax form
// ~1e-6, far inside the 1e-4 bar.  rrs = 1 / rsum of the row.
// rcs = v_rcp(csum) of the column, precomputed once per column by the statistics merge (it was one of the two
// transcendentals per ENTRY of this VALU-bound pass)
__device__ __forceinline__ float conf_value(float v, float cm, float rcs, float rm, float rrs) {
  return __expf((v - cm) + (v - rm)) * (rcs * rrs);
}

// ---- conf = colsoftmax * rowsoftmax, in place; per-row max / first argmax / tie count --------
// Block = 4 waves x kConfRows rows each (wave per row at a time).  With col_part != nullptr the block also
// keeps the column maxima of its rows in LDS (confidences are >= +0, so an unsigned max on the bit patterns is
// the float max, and max is order-independent -> deterministic) and writes them as one partial row.
constexpr int kConfRows = 4;
template <int VEC>
__global__ __launch_bounds__(256) void conf_kernel(float* __restrict__ S, int N, int L,
                                                   const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                   const float* __restrict__ cmax, const float* __restrict__ csum /* 1 / column sum */,
                                                   float* __restrict__ row_cmax, int* __restrict__ row_arg,
                                                   int* __restrict__ row_ties, float* __restrict__ col_part) {
  extern __shared__ unsigned colmax_bits[];
  const int lane = threadIdx.x & 63;
  if (col_part != nullptr) {
    for (int j = threadIdx.x; j < L; j += 256) colmax_bits[j] = 0u;
    __syncthreads();
  }
  for (int rr = 0; rr < kConfRows; ++rr) {
    const int row = (blockIdx.x * kConfRows + rr) * 4 + (threadIdx.x >> 6);
    if (row >= N) break;      // wave-uniform
    float* s = S + (size_t)row * L;
    const float rm = rmax[row], rrs = 1.0f / rsum[row];
    float best = -1.f;
    int arg = 0x7fffffff;
    for (int j = lane * VEC; j < L; j += 64 * VEC) {
      float v[VEC], cm[VEC], cs[VEC];
      Ld<VEC>::load(s + j, v);
      Ld<VEC>::load(cmax + j, cm);
      Ld<VEC>::load(csum + j, cs);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float c = conf_value(v[e], cm[e], cs[e], rm, rrs);
        v[e] = c;
        if (c > best) {
          best = c;
          arg = j + e;
        }
      }
      Ld<VEC>::store(s + j, v);
      if (col_part != nullptr) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) atomicMax(&colmax_bits[j + e], __float_as_uint(v[e]));
      }
    }
    // wave arg-max, ties to the lowest column
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oa = __shfl_xor(arg, o, 64);
      if (ob > best || (ob == best && oa < arg)) {
        best = ob;
        arg = oa;
      }
    }
    int ties = 0;
    for (int j = lane * VEC; j < L; j += 64 * VEC) {
      float v[VEC];
      Ld<VEC>::load(s + j, v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) ties += (v[e] == best) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ties += __shfl_xor(ties, o, 64);
    if (lane == 0) {
      row_cmax[row] = best;
      row_arg[row] = arg;
      row_ties[row] = ties;
    }
  }
  if (col_part != nullptr) {
    __syncthreads();
    for (int j = threadIdx.x; j < L; j += 256) col_part[(size_t)blockIdx.x * L + j] = __uint_as_float(colmax_bits[j]);
  }
}

// Register-resident variant for L <= KMAX * 256 (L % 4 == 0): a lane keeps its confidences of the current row
// in registers (the tie count needs no second read of the row) and a running column maximum over the wave's
// rows; the waves then merge their maxima with ONE LDS max per column each.
// r05: balanced grid.  The first version ran cdiv(N, 16) blocks of 4 waves x 4 rows (313 blocks at N = 5000 on 256 CUs: 57 CUs
// swept 32 rows, the rest 16, and the launch lasted as long as the 32).  Now the grid is at most kConfGrid blocks of 8 waves
// (register use admits one per CU), block b owns rows b, b + grid, b + 2 grid, ... (19 or 20 of them at N = 5000) and wave w of
// it takes every eighth of those -- every CU streams the same number of rows and has 8 rows of loads in flight.  Row results
// do not depend on who computes them and max is order-independent: bit-identical outputs.
constexpr int kConfGrid = 256;
constexpr int kConfWaves = 8;
template <int KMAX>
__global__ __launch_bounds__(kConfWaves * 64) void conf_reg_kernel(float* __restrict__ S, int N, int L,
                                                       const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                       const float* __restrict__ cmax, const float* __restrict__ csum,
                                                       float* __restrict__ row_cmax, int* __restrict__ row_arg,
                                                       int* __restrict__ row_ties, float* __restrict__ col_part) {
  extern __shared__ unsigned colmax_bits[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int j = threadIdx.x; j < L; j += kConfWaves * 64) colmax_bits[j] = 0u;
  __syncthreads();
  float cmx[KMAX][4];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) cmx[k][e] = 0.f;
  for (int row = (int)blockIdx.x + (int)gridDim.x * wave; row < N; row += (int)gridDim.x * kConfWaves) {      // wave-uniform
    float* s = S + (size_t)row * L;
    const float rm = rmax[row], rrs = 1.0f / rsum[row];
    float best = -1.f;
    int arg = 0x7fffffff;
    float cv[KMAX][4];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int j = lane * 4 + k * 256;
#pragma unroll
      for (int e = 0; e < 4; ++e) cv[k][e] = -2.f;     // never equals a confidence
      if (j < L) {
        float v[4], cm[4], cs[4];
        Ld<4>::load(s + j, v);
        Ld<4>::load(cmax + j, cm);
        Ld<4>::load(csum + j, cs);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float c = conf_value(v[e], cm[e], cs[e], rm, rrs);
          cv[k][e] = c;
          cmx[k][e] = fmaxf(cmx[k][e], c);
          if (c > best) {
            best = c;
            arg = j + e;
          }
        }
        Ld<4>::store(s + j, cv[k]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {     // wave arg-max, ties to the lowest column
      const float ob = __shfl_xor(best, o, 64);
      const int oa = __shfl_xor(arg, o, 64);
      if (ob > best || (ob == best && oa < arg)) {
        best = ob;
        arg = oa;
      }
    }
    int ties = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) ties += (cv[k][e] == best) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ties += __shfl_xor(ties, o, 64);
    if (lane == 0) {
      row_cmax[row] = best;
      row_arg[row] = arg;
      row_ties[row] = ties;
    }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int j = lane * 4 + k * 256;
    if (j < L) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicMax(&colmax_bits[j + e], __float_as_uint(cmx[k][e]));
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < L; j += kConfWaves * 64) col_part[(size_t)blockIdx.x * L + j] = __uint_as_float(colmax_bits[j]);
}

// ---- selection + ordered compaction: single block ------------------------------------------
// (r05, measured and removed: a one-pass variant -- thread t owns cdiv(N, 1024) consecutive rows, all per-row loads issued together, one
// block scan -- took 14.9 us against this kernel's 13.0 at N = 5000 with 3080 survivors: the scattered output stores dominate, not the
// barriers; and a variant that merged the column maxima in the same launch behind an arrival ticket took 39.6 us against 6.4 + 13:
// the agent-scope fences cost more than the launch they save.)
__global__ __launch_bounds__(1024) void select_kernel(const float* __restrict__ conf, int N, int L, int wc,
                                                      const float* __restrict__ row_cmax, const int* __restrict__ row_arg,
                                                      const int* __restrict__ row_ties, const float* __restrict__ col_cmax,
                                                      float thr, int border, const float* __restrict__ kpts,
                                                      float base_scale, const float* __restrict__ qscale,
                                                      long long* __restrict__ i_ids, long long* __restrict__ j_ids,
                                                      float* __restrict__ mconf, float* __restrict__ mkpts_c,
                                                      float* __restrict__ mkpts_3d, int* __restrict__ count) {
  __shared__ int wave_tot[16];
  __shared__ int running;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) running = 0;
  // scale * query_image_scale[b][[1, 0]]  (coarse_matching.py:223-228); (h_scale, w_scale) order
  const float sx = qscale ? base_scale * qscale[1] : base_scale;
  const float sy = qscale ? base_scale * qscale[0] : base_scale;
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + tid;
    int sel = -1;
    float c = 0.f;
    if (i < N) {
      c = row_cmax[i];
      if (c > thr) {
        if (row_ties[i] == 1) {
          const int j = row_arg[i];
          const int jy = j / wc, jx = j - jy * wc;
          if (jy >= border && jx >= border && c == col_cmax[j]) sel = j;
        } else {  // exact ties inside the row (rare): first column that survives every test
          const float* r = conf + (size_t)i * L;
          for (int j = 0; j < L; ++j) {
            if (r[j] == c && c == col_cmax[j]) {
              const int jy = j / wc, jx = j - jy * wc;
              if (jy >= border && jx >= border) {
                sel = j;
                break;
              }
            }
          }
        }
      }
    }
    const int flag = sel >= 0 ? 1 : 0;
    // block exclusive scan of flags
    const unsigned long long bal = __ballot(flag);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int off = running;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (flag) {
      const int m = off + before;
      i_ids[m] = i;
      j_ids[m] = sel;
      mconf[m] = c;
      const int jy = sel / wc, jx = sel - jy * wc;
      mkpts_c[2 * m + 0] = (float)jx * sx;
      mkpts_c[2 * m + 1] = (float)jy * sy;
      mkpts_3d[3 * m + 0] = kpts[3 * i + 0];
      mkpts_3d[3 * m + 1] = kpts[3 * i + 1];
      mkpts_3d[3 * m + 2] = kpts[3 * i + 2];
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wave_tot[w];
      running += tot;
    }
    __syncthreads();
  }
  if (tid == 0) *count = running;
}

// merges of the per-tile partials of the confidence sweep (gemm_ss.hip, OPP_SS_CONF; cells = GEMM rows, points = columns)
// per point: best = max over the cell tiles, arg = the first cell holding it (tiles are in ascending cell order and each
// partial is its tile's first such cell), ties = how many cells hold it.  Partials [T][N], thread per point.
__global__ __launch_bounds__(256) void point_best_merge_kernel(const float* __restrict__ pbest, const int* __restrict__ parg,
                                                               const int* __restrict__ pties, int N, int T, float* __restrict__ obest,
                                                               int* __restrict__ oarg, int* __restrict__ oties) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float best = -1.f;
  int arg = 0x7fffffff, ties = 0;
  for (int t = 0; t < T; ++t) {
    const float b = pbest[(size_t)t * N + i];
    if (b > best) {
      best = b;
      arg = parg[(size_t)t * N + i];
      ties = pties[(size_t)t * N + i];
    } else if (b == best) {
      ties += pties[(size_t)t * N + i];
    }
  }
  obest[i] = best;
  oarg[i] = arg;
  oties[i] = ties;
}
// per cell: max over the point tiles.  Partials [L][T] (T contiguous), thread per cell; max is order-independent
__global__ __launch_bounds__(256) void cell_max_merge_kernel(const float* __restrict__ part, int L, int T, float* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= L) return;
  float m = 0.f;
  for (int t = 0; t < T; ++t) m = fmaxf(m, part[(size_t)j * T + t]);
  out[j] = m;
}

}  // namespace

// partial statistics written by the score GEMM epilogue (128x128 / 256x128 tiles) or by the two sweeps of gemm_ss
// (sweep 1: 2 x ([N][tn] + [tm][L]); sweep 2: 3 x [N][tn] + [tm][L], laid over the same space after the merge)
size_t opp_coarse_match_stats_floats(int N, int L) {
  const size_t tn = opp_cdiv(L, 128), tm = opp_cdiv(N, 128);
  // single sweep: 2 x ([N][tn] + [tm][L]); two sweeps: 2 x ([L][tm] + [tn][N]), then 3 x [tn][N] + [L][tm] over the same space
  return 3 * (size_t)N * tn + 2 * tm * (size_t)L + 16;
}

size_t opp_coarse_match_scratch_floats(int N, int L) {
  const int chunks = opp_cdiv(N, 128) > opp_cdiv(N, 4 * kConfRows) ? opp_cdiv(N, 128) : opp_cdiv(N, 4 * kConfRows);
  const size_t Np = (size_t)opp_cdiv(N, 4) * 4, Lp = (size_t)opp_cdiv(L, 4) * 4;
  // rmax, rsum, row_cmax [Np] ; row_arg, row_ties [Np] (int) ; cmax, csum, col_cmax, 1 / csum [Lp] ; partials [chunks][L]
  return 5 * Np + 4 * Lp + (size_t)chunks * L + 64;
}

// S (in: similarity, out: confidence matrix) [N][L].  Outputs have capacity N.
int opp_dual_softmax_select(float* S, int N, int L, int wc, float thr, int border, const float* kpts, float base_scale,
                            const float* qscale, const float* stats, int stats_bm, float* scratch, long long* i_ids, long long* j_ids, float* mconf, float* mkpts_c,
                            float* mkpts_3d, int* count, hipStream_t stream) {
  OPP_CHECK_ARG(N > 0 && L > 0 && wc > 0 && L % wc == 0, "coarse match: bad sizes N=%d L=%d wc=%d", N, L, wc);
  OPP_CHECK_ARG(!stats || stats_bm == 128 || stats_bm == 256 || stats_bm == -1, "coarse match: statistics come from 128- or 256-row GEMM tiles");
  const int chunks = opp_cdiv(N, 128);
  const size_t Np = (size_t)opp_cdiv(N, 4) * 4, Lp = (size_t)opp_cdiv(L, 4) * 4;
  float* rmax = scratch;
  float* rsum = rmax + Np;
  float* row_cmax = rsum + Np;
  int* row_arg = reinterpret_cast<int*>(row_cmax + Np);
  int* row_ties = row_arg + Np;
  float* cmax = reinterpret_cast<float*>(row_ties + Np);
  float* csum = cmax + Lp;
  float* col_cmax = csum + Lp;
  float* crcp = col_cmax + Lp;          // v_rcp(csum): the per-column factor of conf_value()
  float* part = crcp + Lp;
  const bool vec4 = (L % 4 == 0) && ((reinterpret_cast<uintptr_t>(S) & 15) == 0) && ((reinterpret_cast<uintptr_t>(cmax) & 15) == 0);
  dim3 rgrid(opp_cdiv(N, 4)), cgrid(opp_cdiv(L, 256), chunks), lgrid(opp_cdiv(L, 256));

  if (stats_bm == -1) {
    // rmax / rsum / cmax / csum were already merged into `scratch` by the caller (opp_dual_softmax_ss_single)
  } else if (stats) {   // (max, sum exp) partials already produced by the score GEMM epilogue
    const int tn = opp_cdiv(L, 128), tm = opp_cdiv(N, stats_bm);   // score GEMM tiles: stats_bm rows x 128 columns
    const float* rpm = stats;
    const float* rps = rpm + (size_t)N * tn;
    const float* cpm = rps + (size_t)N * tn;
    const float* cps = cpm + (size_t)tm * L;
    hipLaunchKernelGGL(row_merge_kernel, dim3(opp_cdiv(N, 32)), dim3(256), 0, stream, rpm, rps, N, tn, rmax, rsum, (float*)nullptr);
    hipLaunchKernelGGL(col_merge_kernel, dim3(opp_cdiv(L, 64)), dim3(256), 0, stream, cpm, cps, L, tm, cmax, csum, crcp);
  } else {
    if (vec4) hipLaunchKernelGGL(row_stats_kernel<4>, rgrid, dim3(256), 0, stream, S, N, L, rmax, rsum);
    else hipLaunchKernelGGL(row_stats_kernel<1>, rgrid, dim3(256), 0, stream, S, N, L, rmax, rsum);
    hipLaunchKernelGGL(col_partial_kernel<0>, cgrid, dim3(256), 0, stream, S, N, L, 128, (const float*)nullptr, part);
    hipLaunchKernelGGL(col_reduce_kernel<0>, lgrid, dim3(256), 0, stream, part, chunks, L, cmax, (float*)nullptr);
    hipLaunchKernelGGL(col_partial_kernel<1>, cgrid, dim3(256), 0, stream, S, N, L, 128, cmax, part);
    hipLaunchKernelGGL(col_reduce_kernel<1>, lgrid, dim3(256), 0, stream, part, chunks, L, csum, crcp);
  }
  // conf in place + per-row arg-max; the column maxima of the confidences come out of the same pass as one
  // partial row per block when the [L] LDS array fits the default dynamic-LDS limit, else from a second sweep
  const int cblocks = opp_cdiv(N, 4 * kConfRows);
  const bool fuse_cmax = (size_t)L * 4 <= 64 * 1024;
  const size_t conf_lds = fuse_cmax ? (size_t)L * 4 : 0;
  float* cpart = fuse_cmax ? part : nullptr;
  auto select = [&]() {
    hipLaunchKernelGGL(select_kernel, dim3(1), dim3(1024), 0, stream, S, N, L, wc, row_cmax, row_arg, row_ties, col_cmax, thr,
                       border, kpts, base_scale, qscale, i_ids, j_ids, mconf, mkpts_c, mkpts_3d, count);
  };
  if (vec4 && fuse_cmax && L <= 16 * 256)
  {
    // balanced grid of 8-wave blocks (<= cblocks, so the partial rows fit the scratch sized for the 4-wave kernels)
    const int rblocks = cblocks < kConfGrid ? cblocks : kConfGrid;
    {
      OppProfScope prof(OPP_PROF_CONF, stream, (double)N * (double)L * 8.0);   // score matrix read + confidence matrix written
      hipLaunchKernelGGL(conf_reg_kernel<16>, dim3(rblocks), dim3(kConfWaves * 64), conf_lds, stream, S, N, L, rmax, rsum, cmax, crcp, row_cmax, row_arg, row_ties,
                         cpart);
    }
    hipLaunchKernelGGL(col_max_reduce_kernel, dim3(opp_cdiv(L, 64)), dim3(256), 0, stream, part, rblocks, L, col_cmax);
    select();
    OPP_CHECK_LAUNCH("coarse match kernels");
    return OPP_OK;
  }
  if (vec4) hipLaunchKernelGGL(conf_kernel<4>, dim3(cblocks), dim3(256), conf_lds, stream, S, N, L, rmax, rsum, cmax, crcp, row_cmax, row_arg, row_ties, cpart);
  else hipLaunchKernelGGL(conf_kernel<1>, dim3(cblocks), dim3(256), conf_lds, stream, S, N, L, rmax, rsum, cmax, crcp, row_cmax, row_arg, row_ties, cpart);
  if (fuse_cmax) {
    hipLaunchKernelGGL(col_max_reduce_kernel, dim3(opp_cdiv(L, 64)), dim3(256), 0, stream, part, cblocks, L, col_cmax);
  } else {
    hipLaunchKernelGGL(col_partial_kernel<0>, cgrid, dim3(256), 0, stream, S, N, L, 128, (const float*)nullptr, part);
    hipLaunchKernelGGL(col_reduce_kernel<0>, lgrid, dim3(256), 0, stream, part, chunks, L, col_cmax, (float*)nullptr);
  }
  select();
  OPP_CHECK_LAUNCH("coarse match kernels");
  return OPP_OK;
}

// ---- two sweeps of the split-operand score GEMM (gemm_ss.hip): statistics, then confidences written once ------------
// The GEMM runs with the image cells as its rows and the 3D points as its columns (conf[point][cell] then leaves the
// accumulators in 16-byte pieces), so "row" statistics below belong to cells and "column" statistics to points.
int opp_dual_softmax_two_sweep(const void* f3s, const void* f2s, int C, int N, int L, int wc, float out_mul, float out_div,
                               const float* col_mask, float thr, int border, const float* kpts, float base_scale, const float* qscale,
                               float* conf, float* stats, float* scratch, long long* i_ids, long long* j_ids, float* mconf,
                               float* mkpts_c, float* mkpts_3d, int* count, hipStream_t stream) {
  OPP_CHECK_ARG(N > 0 && L > 0 && wc > 0 && L % wc == 0 && C % 32 == 0, "coarse match: bad sizes N=%d L=%d wc=%d C=%d", N, L, wc, C);
  const int tl = opp_cdiv(L, opp_gemm_ss_tile_rows()), tp = opp_cdiv(N, opp_gemm_ss_tile_cols());   // cell tiles, point tiles
  const size_t Np = (size_t)opp_cdiv(N, 4) * 4, Lp = (size_t)opp_cdiv(L, 4) * 4;
  float* rmax = scratch;                 // per point: max / sum exp of its score row
  float* rsum = rmax + Np;
  float* row_cmax = rsum + Np;           // per point: best confidence, its first cell, tie count
  int* row_arg = reinterpret_cast<int*>(row_cmax + Np);
  int* row_ties = row_arg + Np;
  float* cmax = reinterpret_cast<float*>(row_ties + Np);   // per cell
  float* csum = cmax + Lp;
  float* col_cmax = csum + Lp;
  OppGemmSS g;
  g.A = f2s;
  g.B = f3s;
  g.lda = C * 6;
  g.ldb = C * 6;
  g.M = L;
  g.N = N;
  g.K = C;
  g.out_mul = out_mul;
  g.out_div = out_div;
  g.row_mask = col_mask;                 // masked image cells (coarse_matching.py:108-114)
  // sweep 1: (max, sum exp) partials; cells [L][tp], points [tl][N]
  g.mode = OPP_SS_STATS;
  g.stat_rowmax = stats;
  g.stat_rowsum = g.stat_rowmax + (size_t)L * tp;
  g.stat_colmax = g.stat_rowsum + (size_t)L * tp;
  g.stat_colsum = g.stat_colmax + (size_t)tl * N;
  OPP_TRY(opp_gemm_ss(g, stream));
  hipLaunchKernelGGL(row_merge_kernel, dim3(opp_cdiv(L, 32)), dim3(256), 0, stream, g.stat_rowmax, g.stat_rowsum, L, tp, cmax, csum, (float*)nullptr);
  hipLaunchKernelGGL(col_merge_kernel, dim3(opp_cdiv(N, 64)), dim3(256), 0, stream, g.stat_colmax, g.stat_colsum, N, tl, rmax, rsum, (float*)nullptr);
  // sweep 2: confidences + per-tile partials; the partial space is reused (the merges above are done with it in stream order)
  g.mode = OPP_SS_CONF;
  g.rstat_max = cmax;
  g.rstat_sum = csum;
  g.cstat_max = rmax;
  g.cstat_sum = rsum;
  g.C = conf;
  g.ldc = L;
  g.part_best = stats;
  g.part_arg = reinterpret_cast<int*>(stats + (size_t)tl * N);
  g.part_ties = g.part_arg + (size_t)tl * N;
  g.part_rowmax = reinterpret_cast<float*>(g.part_ties + (size_t)tl * N);
  OPP_TRY(opp_gemm_ss(g, stream));
  hipLaunchKernelGGL(point_best_merge_kernel, dim3(opp_cdiv(N, 256)), dim3(256), 0, stream, g.part_best, g.part_arg, g.part_ties, N, tl, row_cmax,
                     row_arg, row_ties);
  hipLaunchKernelGGL(cell_max_merge_kernel, dim3(opp_cdiv(L, 256)), dim3(256), 0, stream, g.part_rowmax, L, tp, col_cmax);
  hipLaunchKernelGGL(select_kernel, dim3(1), dim3(1024), 0, stream, conf, N, L, wc, row_cmax, row_arg, row_ties, col_cmax, thr, border, kpts,
                     base_scale, qscale, i_ids, j_ids, mconf, mkpts_c, mkpts_3d, count);
  OPP_CHECK_LAUNCH("coarse match (two sweeps)");
  return OPP_OK;
}

// ---- single sweep on the split-operand GEMM: statistics + score matrix from gemm_ss (cells = GEMM rows), then the in-place
// confidence pass and the selection of the materialised path ------------------------------------------------------------
int opp_dual_softmax_ss_single(const void* f3s, const void* f2s, int C, int N, int L, int wc, float out_mul, float out_div,
                               const float* col_mask, float thr, int border, const float* kpts, float base_scale, const float* qscale,
                               float* conf, float* stats, float* scratch, long long* i_ids, long long* j_ids, float* mconf,
                               float* mkpts_c, float* mkpts_3d, int* count, hipStream_t stream) {
  OPP_CHECK_ARG(N > 0 && L > 0 && wc > 0 && L % wc == 0 && C % 32 == 0, "coarse match: bad sizes N=%d L=%d wc=%d C=%d", N, L, wc, C);
  const int tl = opp_cdiv(L, opp_gemm_ss_tile_rows()), tp = opp_cdiv(N, opp_gemm_ss_tile_cols());
  const size_t Np = (size_t)opp_cdiv(N, 4) * 4, Lp = (size_t)opp_cdiv(L, 4) * 4;
  float* rmax = scratch;                 // same layout as opp_dual_softmax_select
  float* rsum = rmax + Np;
  float* cmax = rsum + Np + Np + Np + Np;
  float* csum = cmax + Lp;
  float* crcp = csum + Lp + Lp;          // behind col_cmax
  OppGemmSS g;
  g.A = f2s;
  g.B = f3s;
  g.lda = C * 6;
  g.ldb = C * 6;
  g.M = L;
  g.N = N;
  g.K = C;
  g.out_mul = out_mul;
  g.out_div = out_div;
  g.row_mask = col_mask;
  g.mode = OPP_SS_STATS_STORE;
  g.C = conf;
  g.ldc = L;
  g.stat_rowmax = stats;
  g.stat_rowsum = g.stat_rowmax + (size_t)L * tp;
  g.stat_colmax = g.stat_rowsum + (size_t)L * tp;
  g.stat_colsum = g.stat_colmax + (size_t)tl * N;
  OPP_TRY(opp_gemm_ss(g, stream));
  hipLaunchKernelGGL(rowcol_merge_kernel, dim3(opp_cdiv(L, 32) + opp_cdiv(N, 64)), dim3(256), 0, stream, opp_cdiv(L, 32), g.stat_rowmax, g.stat_rowsum, L, tp,
                     cmax, csum, crcp, g.stat_colmax, g.stat_colsum, N, tl, rmax, rsum, (float*)nullptr);
  return opp_dual_softmax_select(conf, N, L, wc, thr, border, kpts, base_scale, qscale, stats, -1, scratch, i_ids, j_ids, mconf, mkpts_c, mkpts_3d,
                                 count, stream);
}
