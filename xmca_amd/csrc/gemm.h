// MFMA GEMM for gfx950: C[MxN] = alpha * rs[m] * cs[n] * op(A) * op(B) + beta * C
//
// * f64 operands -> v_mfma_f64_16x16x4_f64, f32 operands -> v_mfma_f32_16x16x4_f32
//   (exact f32, optionally flushed into f64 side accumulators every FLUSH_TILES k-tiles
//   so that long contractions keep f64-class accumulation error: "WIDE").
// * 128x128 block tile, BK = 16, 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles.
// * operands are staged global -> registers -> LDS (k-major, padded) with a register
//   prefetch of the next k-tile while the current one feeds the matrix pipe.
// * All four operand orientations of row-major storage are supported so that no
//   transposed copy of a space x time field is ever materialised:
//       A_KFAST: A(m,k) = A[m*lda + k]   else  A(m,k) = A[k*lda + m]
//       B_NFAST: B(k,n) = B[k*ldb + n]   else  B(k,n) = B[n*ldb + k]
// * upper_only: compute only block tiles with bn >= bm (Gram / Hermitian products);
//   mirror = +1/-1 writes the (anti)symmetric counterpart of off-diagonal tiles.
// * split-K through blockIdx.z into an f64 workspace + reduce kernel (deterministic).
//
// Reference call sites this replaces: numpy `@` / gesdd inner products of
// xmca/array.py:552-566 and :580-584 (see DESIGN.md for the formulation).
#pragma once
#include "common.h"

namespace xmca {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <typename T>
struct Mfma;
template <>
struct Mfma<double> {
  using acc_t = d4_t;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma<float> {
  using acc_t = f4_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 16, GEMM_LDS_LD = 130;
constexpr int GEMM_FLUSH_TILES = 32;  // WIDE: f32 partial sums cover at most 32*16 = 512 products

template <typename TI, typename TO>
struct GemmParams {
  const TI* A;
  const TI* B;
  TO* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  double alpha, beta;
  const double* row_scale;  // nullable, length M
  const double* col_scale;  // nullable, length N
  int upper_only;
  int mirror;
  int k_chunk;              // contraction length handled by one blockIdx.z slice
  int64_t split_stride;     // elements between consecutive split-K slices of C
};

template <typename TI, typename TO, bool A_KFAST, bool B_NFAST, bool WIDE>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams<TI, TO> p) {
  using M_ = Mfma<TI>;
  using acc_t = typename M_::acc_t;
  __shared__ TI As[2][GEMM_BK][GEMM_LDS_LD];
  __shared__ TI Bs[2][GEMM_BK][GEMM_LDS_LD];

  const int tiles_n = (p.N + GEMM_BN - 1) / GEMM_BN;
  const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
  if (p.upper_only && bn < bm) return;
  const int bm0 = bm * GEMM_BM, bn0 = bn * GEMM_BN;
  const int kbeg = blockIdx.z * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  TO* __restrict__ C = p.C + (int64_t)blockIdx.z * p.split_stride;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int l15 = lane & 15, l4 = lane >> 4;

  acc_t acc[4][4];
  d4_t wide[WIDE ? 4 : 1][WIDE ? 4 : 1];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j] = acc_t{0, 0, 0, 0};
      if constexpr (WIDE) wide[i][j] = d4_t{0, 0, 0, 0};
    }

  TI ra[8], rb[8];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int m, kk;
      if constexpr (A_KFAST) { m = (tid >> 4) + 16 * i; kk = tid & 15; }
      else                   { kk = 2 * i + (tid >> 7); m = tid & 127; }
      const int gm = bm0 + m, gk = k0 + kk;
      TI v = TI(0);
      if (gm < p.M && gk < kend)
        v = A_KFAST ? p.A[(int64_t)gm * p.lda + gk] : p.A[(int64_t)gk * p.lda + gm];
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int n, kk;
      if constexpr (B_NFAST) { kk = 2 * i + (tid >> 7); n = tid & 127; }
      else                   { n = (tid >> 4) + 16 * i; kk = tid & 15; }
      const int gn = bn0 + n, gk = k0 + kk;
      TI v = TI(0);
      if (gn < p.N && gk < kend)
        v = B_NFAST ? p.B[(int64_t)gk * p.ldb + gn] : p.B[(int64_t)gn * p.ldb + gk];
      rb[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int m, kk;
      if constexpr (A_KFAST) { m = (tid >> 4) + 16 * i; kk = tid & 15; }
      else                   { kk = 2 * i + (tid >> 7); m = tid & 127; }
      As[buf][kk][m] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int n, kk;
      if constexpr (B_NFAST) { kk = 2 * i + (tid >> 7); n = tid & 127; }
      else                   { n = (tid >> 4) + 16 * i; kk = tid & 15; }
      Bs[buf][kk][n] = rb[i];
    }
  };

  const int nkt = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;
  if (nkt > 0) {
    load_tile(kbeg);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) load_tile(kbeg + (kt + 1) * GEMM_BK);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int kr = k4 * 4 + l4;
      TI a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[cur][kr][wm + i * 16 + l15];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[cur][kr][wn + j * 16 + l15];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = M_::mma(a[i], b[j], acc[i][j]);
    }
    if constexpr (WIDE) {
      if ((kt % GEMM_FLUSH_TILES) == GEMM_FLUSH_TILES - 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) wide[i][j][r] += (double)acc[i][j][r];
            acc[i][j] = acc_t{0, 0, 0, 0};
          }
      }
    }
    if (kt + 1 < nkt) store_tile(cur ^ 1);
    __syncthreads();
  }

  const bool offdiag = (bm != bn);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = bm0 + wm + i * 16 + M_::row(lane, r);
        const int col = bn0 + wn + j * 16 + l15;
        if (row < p.M && col < p.N) {
          double v = (double)acc[i][j][r];
          if constexpr (WIDE) v += wide[i][j][r];
          v *= p.alpha;
          if (p.row_scale) v *= p.row_scale[row];
          if (p.col_scale) v *= p.col_scale[col];
          const int64_t o = (int64_t)row * p.ldc + col;
          if (p.beta != 0.0) v += p.beta * (double)C[o];
          C[o] = (TO)v;
          if (p.mirror != 0 && offdiag) C[(int64_t)col * p.ldc + row] = (TO)(p.mirror > 0 ? v : -v);
        }
      }
}

// C = alpha * rs*cs * sum_z W[z] + beta*C, honouring upper_only / mirror at the GEMM's block-tile granularity
template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const double* __restrict__ W, int splits, int64_t split_stride,
                                                            TO* __restrict__ C, int M, int N, int64_t ldc, double alpha,
                                                            double beta, const double* __restrict__ row_scale,
                                                            const double* __restrict__ col_scale, int upper_only, int mirror) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx % N);
  const int tm = m / GEMM_BM, tn = n / GEMM_BN;
  if (upper_only && tn < tm) return;
  double v = 0.0;
  for (int z = 0; z < splits; ++z) v += W[(int64_t)z * split_stride + idx];
  v *= alpha;
  if (row_scale) v *= row_scale[m];
  if (col_scale) v *= col_scale[n];
  const int64_t o = (int64_t)m * ldc + n;
  if (beta != 0.0) v += beta * (double)C[o];
  C[o] = (TO)v;
  if (mirror != 0 && tn > tm) C[(int64_t)n * ldc + m] = (TO)(mirror > 0 ? v : -v);
}

struct GemmOpts {
  bool a_kfast = true;   // A(m,k) = A[m*lda + k]
  bool b_nfast = true;   // B(k,n) = B[k*ldb + n]
  double alpha = 1.0, beta = 0.0;
  const double* row_scale = nullptr;
  const double* col_scale = nullptr;
  bool upper_only = false;
  int mirror = 0;
  int force_splits = 0;  // 0 = heuristic
};

// scratch for split-K partial sums, owned by the caller (one per stream)
struct GemmWorkspace {
  DevBuf<double> partial;
};

template <typename TI, typename TO, bool WIDE>
static void launch_gemm_variant(hipStream_t st, const GemmParams<TI, TO>& p, dim3 grid, bool a_kfast, bool b_nfast) {
  dim3 block(256);
  if (a_kfast && b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, true, true, WIDE>), grid, block, 0, st, p);
  else if (a_kfast && !b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, true, false, WIDE>), grid, block, 0, st, p);
  else if (!a_kfast && b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, false, true, WIDE>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((gemm_kernel<TI, TO, false, false, WIDE>), grid, block, 0, st, p);
  XMCA_HIP(hipGetLastError());
}

// TI in {float,double}; TO in {float,double}.  f32 operands always use WIDE accumulation.
template <typename TI, typename TO>
void gemm(hipStream_t st, GemmWorkspace& ws, const TI* A, int64_t lda, const TI* B, int64_t ldb, TO* C, int64_t ldc, int M,
          int N, int K, const GemmOpts& o) {
  if (M <= 0 || N <= 0) return;
  XMCA_CHECK(!o.upper_only || M == N, XMCA_ERR_INVALID, "gemm: upper_only needs a square result");
  const int tm = ceil_div(M, GEMM_BM), tn = ceil_div(N, GEMM_BN);
  const int64_t tiles = o.upper_only ? (int64_t)tm * (tm + 1) / 2 : (int64_t)tm * tn;
  const int nkt = ceil_div(K, GEMM_BK);
  int splits = o.force_splits;
  if (splits <= 0) {
    splits = 1;
    if (tiles < 512 && nkt >= 64) {
      splits = (int)((768 + tiles - 1) / tiles);
      const int max_by_k = nkt / 32 > 0 ? nkt / 32 : 1;   // at least 32 k-tiles (512 products) per slice
      if (splits > max_by_k) splits = max_by_k;
      if (splits > 64) splits = 64;
    }
  }
  constexpr bool WIDE = std::is_same<TI, float>::value;
  if (splits <= 1 || K == 0) {
    GemmParams<TI, TO> p{A, B, C, M, N, K, lda, ldb, ldc, o.alpha, o.beta, o.row_scale, o.col_scale,
                         o.upper_only ? 1 : 0, o.mirror, K > 0 ? K : 1, 0};
    launch_gemm_variant<TI, TO, WIDE>(st, p, dim3(tm * tn, 1, 1), o.a_kfast, o.b_nfast);
    return;
  }
  int k_chunk = ceil_div(nkt, splits) * GEMM_BK;
  splits = ceil_div(K, k_chunk);
  const int64_t stride = (int64_t)M * N;
  double* W = ws.partial.ensure((size_t)stride * splits);
  GemmParams<TI, double> p{A, B, W, M, N, K, lda, ldb, (int64_t)N, 1.0, 0.0, nullptr, nullptr,
                           o.upper_only ? 1 : 0, 0, k_chunk, stride};
  launch_gemm_variant<TI, double, WIDE>(st, p, dim3(tm * tn, 1, splits), o.a_kfast, o.b_nfast);
  const int64_t total = stride;
  hipLaunchKernelGGL((splitk_reduce_kernel<TO>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, splits, stride, C,
                     M, N, ldc, o.alpha, o.beta, o.row_scale, o.col_scale, o.upper_only ? 1 : 0, o.mirror);
  XMCA_HIP(hipGetLastError());
}

}  // namespace xmca
