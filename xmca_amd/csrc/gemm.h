// MFMA GEMM for gfx950: C[MxN] = alpha * rs[m] * cs[n] * op(A) * op(B) + beta * C
//
// * f64 operands -> v_mfma_f64_16x16x4_f64, f32 operands -> v_mfma_f32_16x16x4_f32
//   (exact f32, flushed into f64 side accumulators every FLUSH_TILES k-tiles so that long contractions keep
//   f64-class accumulation error: "WIDE").
// * 128x128 block tile, BK = 32, 512 threads = 8 waves (2 x 4), each wave 64x32 = 4x2 MFMA tiles
//   (~130 VGPRs -> two workgroups per CU).
// * operands are staged global -> registers -> LDS (k-major, padded); the loads of the next k-tile are in flight
//   while the current one feeds the matrix pipe.  Global reads are 16-byte vectors along the contiguous axis:
//   a k-contiguous operand is read in 256-byte (f64) / 128-byte (f32) row segments.
// * All four operand orientations of row-major storage are supported, so no transposed copy of a space x time
//   field is ever materialised:
//       A_KFAST: A(m,k) = A[m*lda + k]   else  A(m,k) = A[k*lda + m]
//       B_NFAST: B(k,n) = B[k*ldb + n]   else  B(k,n) = B[n*ldb + k]
// * upper_only: only block tiles with bn >= bm are launched (Gram / Hermitian products);
//   mirror = +1/-1 writes the (anti)symmetric counterpart of off-diagonal tiles.
// * the (split, tile) list is dealt to the 8 XCDs in contiguous ranges (workgroup b runs on XCD b % 8), so the
//   workgroups sharing one L2 work on neighbouring tiles of the same k-slice.
// * split-K through an f64 workspace + reduce kernel (deterministic).
//
// Reference call sites this replaces: numpy `@` / gesdd inner products of
// xmca/array.py:552-566 and :580-584 (see DESIGN.md for the formulation).
#pragma once
#include <algorithm>
#include <memory>
#include <type_traits>
#include <vector>

#include "common.h"

namespace xmca {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

template <typename T>
struct Mfma;
template <>
struct Mfma<double> {
  using acc_t = d4_t;
  using vec_t = d2_t;
  static constexpr int VW = 2;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma<float> {
  using acc_t = f4_t;
  using vec_t = f4_t;
  static constexpr int VW = 4;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 32, GEMM_THREADS = 512;
// LDS row pitch (elements).  A k-contiguous operand is stored TRANSPOSED into the k-major LDS tile with scalar
// writes: pitch 129 puts the 16 (f64) / 32 (f32) lanes of a write group on distinct banks.  An m/n-contiguous operand
// is stored with 16-byte vector writes: pitch 132 keeps every row 16-byte aligned for f32 and f64.
template <bool FAST_K>
struct GemmPitch { static constexpr int value = FAST_K ? 129 : 132; };
constexpr int GEMM_FLUSH_TILES = 16;  // WIDE: f32 partial sums cover at most 16 * 32 = 512 products

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
  int k_chunk;              // contraction length handled by one split
  int64_t split_stride;     // elements between consecutive split-K slices of C
  int n_tiles;              // block tiles per split (upper triangle only when upper_only)
  int n_wg;                 // n_tiles * splits
  const int* tile_map;      // n_tiles packed (bm << 16 | bn) in super-block order
  int vec_a, vec_b;         // 16-byte vector loads allowed (base and leading dimension aligned)
};

// One (128 x 32) operand tile: 8 elements per thread as NV vectors of VW.
//   FAST_K: the global array is contiguous along k (row index = m/n), else contiguous along m/n (row index = k).
template <typename TI, bool FAST_K>
struct GemmTileIO {
  static constexpr int VW = Mfma<TI>::VW, NV = 8 / VW;
  static constexpr int PITCH = GemmPitch<FAST_K>::value;
  using vec_t = typename Mfma<TI>::vec_t;
  // position of vector i of this thread inside the tile: r along the 128-wide axis, k along the contraction axis
  static __device__ __forceinline__ void pos(int i, int& r, int& k) {
    const int v = i * GEMM_THREADS + (int)threadIdx.x;
    if constexpr (FAST_K) { r = v / (GEMM_BK / VW); k = (v % (GEMM_BK / VW)) * VW; }
    else                  { k = v / (GEMM_BM / VW); r = (v % (GEMM_BM / VW)) * VW; }
  }
  // interior tile: unpredicated 16-byte loads from a per-thread base pointer (tile origin already applied)
  static __device__ __forceinline__ void load_fast(const TI* __restrict__ base, int64_t ld, TI (&reg)[8]) {
    int r0, k0;
    pos(0, r0, k0);
    const TI* p0 = FAST_K ? base + (int64_t)r0 * ld + k0 : base + (int64_t)k0 * ld + r0;
    // consecutive vectors of a thread are a fixed number of rows apart
    constexpr int STEP = FAST_K ? GEMM_THREADS / (GEMM_BK / VW) : GEMM_THREADS / (GEMM_BM / VW);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const vec_t x = *reinterpret_cast<const vec_t*>(p0 + (int64_t)i * STEP * ld);
#pragma unroll
      for (int j = 0; j < VW; ++j) reg[i * VW + j] = x[j];
    }
  }
  // edge tile / unaligned operand: element-wise, zero filled
  static __device__ __forceinline__ void load_slow(const TI* __restrict__ P, int64_t ld, int row0, int rows, int k0, int kend,
                                                   TI (&reg)[8]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int r, k;
      pos(i, r, k);
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        const int er = row0 + (FAST_K ? r : r + j), ek = k0 + (FAST_K ? k + j : k);
        TI x = TI(0);
        if (er < rows && ek < kend) x = FAST_K ? P[(int64_t)er * ld + ek] : P[(int64_t)ek * ld + er];
        reg[i * VW + j] = x;
      }
    }
  }
  static __device__ __forceinline__ void store(TI (*S)[PITCH], const TI (&reg)[8]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int r, k;
      pos(i, r, k);
      if constexpr (FAST_K) {
#pragma unroll
        for (int j = 0; j < VW; ++j) S[k + j][r] = reg[i * VW + j];
      } else {
        vec_t x;
#pragma unroll
        for (int j = 0; j < VW; ++j) x[j] = reg[i * VW + j];
        *reinterpret_cast<vec_t*>(&S[k][r]) = x;
      }
    }
  }
};

// MINW = minimum waves per SIMD the register allocation must allow: 4 -> two 512-thread workgroups per CU
// (<= 128 VGPRs), 2 -> one workgroup per CU.
template <typename TI, typename TO, bool A_KFAST, bool B_NFAST, bool WIDE, int MINW>
__global__ __launch_bounds__(GEMM_THREADS, MINW) void gemm_kernel(GemmParams<TI, TO> p) {
  using M_ = Mfma<TI>;
  using acc_t = typename M_::acc_t;
  using IOA = GemmTileIO<TI, A_KFAST>;
  using IOB = GemmTileIO<TI, !B_NFAST>;
  __shared__ __attribute__((aligned(16))) TI As[GEMM_BK][IOA::PITCH];
  __shared__ __attribute__((aligned(16))) TI Bs[GEMM_BK][IOB::PITCH];

  // XCD-aware, bijective remap of the launch index (cdna_hip_programming.md T1)
  int L;
  {
    const int b = blockIdx.x, q = p.n_wg / 8, r = p.n_wg % 8, xcd = b % 8, idx = b / 8;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = L / p.n_tiles;
  const int packed = p.tile_map[L % p.n_tiles];
  const int bm = packed >> 16, bn = packed & 0xffff;
  const int bm0 = bm * GEMM_BM, bn0 = bn * GEMM_BN;
  const int kbeg = split * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  TO* __restrict__ C = p.C + (int64_t)split * p.split_stride;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;
  const int l15 = lane & 15, l4 = lane >> 4;

  acc_t acc[4][2];
  d4_t wide[WIDE ? 4 : 1][WIDE ? 2 : 1];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc[i][j] = acc_t{0, 0, 0, 0};
      if constexpr (WIDE) wide[i][j] = d4_t{0, 0, 0, 0};
    }

  // workgroup-uniform: whole tile inside the matrices and both operands 16-byte aligned -> unpredicated vector loads
  const bool interior = (bm0 + GEMM_BM <= p.M) && (bn0 + GEMM_BN <= p.N) && p.vec_a && p.vec_b;
  const TI* __restrict__ baseA = A_KFAST ? p.A + (int64_t)bm0 * p.lda : p.A + bm0;   // + k offset per tile
  const TI* __restrict__ baseB = B_NFAST ? p.B + bn0 : p.B + (int64_t)bn0 * p.ldb;
  const int64_t kstepA = A_KFAST ? 1 : p.lda, kstepB = B_NFAST ? p.ldb : 1;
  TI ra[8], rb[8];
  auto load = [&](int k0) {
    if (interior && k0 + GEMM_BK <= kend) {
      IOA::load_fast(baseA + (int64_t)k0 * kstepA, p.lda, ra);
      IOB::load_fast(baseB + (int64_t)k0 * kstepB, p.ldb, rb);
    } else {
      IOA::load_slow(p.A, p.lda, bm0, p.M, k0, kend, ra);
      IOB::load_slow(p.B, p.ldb, bn0, p.N, k0, kend, rb);
    }
  };
  const int nkt = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;
  if (nkt > 0) load(kbeg);
  for (int kt = 0; kt < nkt; ++kt) {
    IOA::store(As, ra);
    IOB::store(Bs, rb);
    __syncthreads();
    if (kt + 1 < nkt) load(kbeg + (kt + 1) * GEMM_BK);
#pragma unroll
    for (int k4 = 0; k4 < GEMM_BK / 4; ++k4) {
      const int kr = k4 * 4 + l4;
      TI a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kr][wm + i * 16 + l15];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kr][wn + j * 16 + l15];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = M_::mma(a[i], b[j], acc[i][j]);
    }
    if constexpr (WIDE) {
      if ((kt % GEMM_FLUSH_TILES) == GEMM_FLUSH_TILES - 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) wide[i][j][r] += (double)acc[i][j][r];
            acc[i][j] = acc_t{0, 0, 0, 0};
          }
      }
    }
    __syncthreads();   // every wave is done with this k-tile before it is overwritten
  }

  const bool offdiag = (bm != bn);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
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
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;   // optional: recorded around the MFMA kernel launch alone
};

// stream-K schedule state of the NT kernel (gemm_nt.h)
struct GemmNtWorkspace {
  DevBuf<double> partial;       // raw sums of the partial segments, [grid][2][BM * BN]
  DevBuf<int> split_tiles;      // tiles whose units span more than one workgroup range
  int n_split = 0;
  int64_t key_tiles = -1;
  int key_nkt = -1, key_upw = -1;
  int n_cus = 0;
};

// scratch for split-K partial sums and the tile-order tables, owned by the caller (one per stream)
struct GemmWorkspace {
  DevBuf<double> partial;
  GemmNtWorkspace nt;
  struct Map { int tm, tn, upper, ratio; DevBuf<int> dev; int n; };
  std::vector<std::unique_ptr<Map>> maps;

  // Tiles are enumerated in 8 x 8 super-blocks: the ~64 workgroups that share one XCD's L2 at a time then touch only
  // 8 + 8 operand panels, so each panel slice is fetched from HBM / Infinity Cache once per XCD and reused from L2.
  // `ratio` = tile rows / tile columns (rectangular tiles of gemm_big.h): with `upper`, tile (i, j) is kept when it reaches
  // the diagonal or lies above it, j >= ratio * i.
  const Map& tile_map(hipStream_t st, int tm, int tn, bool upper, int ratio = 1) {
    for (auto& m : maps)
      if (m->tm == tm && m->tn == tn && m->upper == (int)upper && m->ratio == ratio) return *m;
    constexpr int G = 8;
    std::vector<int> order;
    for (int si = 0; si < tm; si += G)
      for (int sj = upper ? (si * ratio) / G * G : 0; sj < tn; sj += G)
        for (int i = si; i < std::min(si + G, tm); ++i)
          for (int j = sj; j < std::min(sj + G, tn); ++j)
            if (!upper || j >= ratio * i) order.push_back((i << 16) | j);
    auto m = std::make_unique<Map>();
    m->tm = tm; m->tn = tn; m->upper = upper; m->ratio = ratio; m->n = (int)order.size();
    XMCA_HIP(hipMemcpyAsync(m->dev.ensure(order.size()), order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice, st));
    XMCA_HIP(hipStreamSynchronize(st));
    maps.push_back(std::move(m));
    return *maps.back();
  }
};

template <typename TI, typename TO, bool WIDE>
static void launch_gemm_variant(hipStream_t st, const GemmParams<TI, TO>& p, bool a_kfast, bool b_nfast) {
  dim3 grid(p.n_wg), block(GEMM_THREADS);
  static const int minw = [] { const char* e = std::getenv("XMCA_GEMM_MINW"); return (e && e[0] == '2') ? 2 : 4; }();
  if (minw == 4) {
    if (a_kfast && b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, true, true, WIDE, 4>), grid, block, 0, st, p);
    else if (a_kfast && !b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, true, false, WIDE, 4>), grid, block, 0, st, p);
    else if (!a_kfast && b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, false, true, WIDE, 4>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_kernel<TI, TO, false, false, WIDE, 4>), grid, block, 0, st, p);
  } else {
    if (a_kfast && b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, true, true, WIDE, 2>), grid, block, 0, st, p);
    else if (a_kfast && !b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, true, false, WIDE, 2>), grid, block, 0, st, p);
    else if (!a_kfast && b_nfast) hipLaunchKernelGGL((gemm_kernel<TI, TO, false, true, WIDE, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_kernel<TI, TO, false, false, WIDE, 2>), grid, block, 0, st, p);
  }
  XMCA_HIP(hipGetLastError());
}

}  // namespace xmca
#include "gemm_nt.h"
namespace xmca {

// TI in {float,double}; TO in {float,double}.  f32 operands always use WIDE accumulation.
template <typename TI, typename TO>
void gemm(hipStream_t st, GemmWorkspace& ws, const TI* A, int64_t lda, const TI* B, int64_t ldb, TO* C, int64_t ldc, int M,
          int N, int K, const GemmOpts& o) {
  if (M <= 0 || N <= 0) return;
  XMCA_CHECK(!o.upper_only || M == N, XMCA_ERR_INVALID, "gemm: upper_only needs a square result");
  // both operands contiguous along the contraction axis (covariance / Gram products): stream-K kernel of gemm_nt.h
  if (o.a_kfast && !o.b_nfast && gemm_nt<TI, TO>(st, ws, ws.nt, A, lda, B, ldb, C, ldc, M, N, K, o)) return;
  const int tm = ceil_div(M, GEMM_BM), tn = ceil_div(N, GEMM_BN);
  const int64_t tiles = o.upper_only ? (int64_t)tm * (tm + 1) / 2 : (int64_t)tm * tn;
  const int nkt = ceil_div(K, GEMM_BK);
  int splits = o.force_splits;
  if (splits <= 0) {
    // Cost model in units of one k-tile of one workgroup (two workgroups share a CU -> 512 slots), fitted to a sweep of
    // the split count on the C2 Gram shape (scripts/gram_splits.py: 3 -> 2.13 ms, 5 -> 1.85, 7 -> 1.74, 9 -> 1.72,
    // 14 -> 1.79):  (tiles * s / 512 + 1/2) rounds - the last, partly filled round costs about half a round because its
    // workgroups meet less contention - times (k-tiles per slice + ~3 tiles of prologue/epilogue), plus the partial-sum
    // traffic of s slices.
    splits = 1;
    if (nkt >= 32) {
      const int max_s = std::min(std::max(nkt / 8, 1), 128);
      const int min_s = std::min(ceil_div(nkt, 512), max_s);
      double best = 1e300;
      for (int s_ = std::max(min_s, 1); s_ <= max_s; ++s_) {
        const double rounds = std::max((double)tiles * s_ / 512.0, 1.0) + 0.5;
        const double cost = rounds * ((double)nkt / s_ + 3.0) + (s_ > 1 ? s_ * (double)tiles / 200.0 : 0.0);
        if (cost < best) { best = cost; splits = s_; }
      }
    }
  }
  const GemmWorkspace::Map& map = ws.tile_map(st, tm, tn, o.upper_only);
  XMCA_CHECK(map.n == tiles && tm < 65536 && tn < 65536, XMCA_ERR_INVALID, "gemm: tile map mismatch");
  constexpr bool WIDE = std::is_same<TI, float>::value;
  constexpr int VW = Mfma<TI>::VW;
  const int vec_a = (lda % VW == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0);
  const int vec_b = (ldb % VW == 0) && (reinterpret_cast<uintptr_t>(B) % 16 == 0);
  if (splits <= 1 || K == 0) {
    GemmParams<TI, TO> p{A, B, C, M, N, K, lda, ldb, ldc, o.alpha, o.beta, o.row_scale, o.col_scale,
                         o.upper_only ? 1 : 0, o.mirror, K > 0 ? K : 1, 0, (int)tiles, (int)tiles, map.dev.get(), vec_a, vec_b};
    if (o.ev_begin) XMCA_HIP(hipEventRecord(o.ev_begin, st));
    launch_gemm_variant<TI, TO, WIDE>(st, p, o.a_kfast, o.b_nfast);
    if (o.ev_end) XMCA_HIP(hipEventRecord(o.ev_end, st));
    return;
  }
  int k_chunk = ceil_div(nkt, splits) * GEMM_BK;
  splits = ceil_div(K, k_chunk);
  const int64_t stride = (int64_t)M * N;
  double* W = ws.partial.ensure((size_t)stride * splits);
  GemmParams<TI, double> p{A, B, W, M, N, K, lda, ldb, (int64_t)N, 1.0, 0.0, nullptr, nullptr,
                           o.upper_only ? 1 : 0, 0, k_chunk, stride, (int)tiles, (int)(tiles * splits), map.dev.get(), vec_a, vec_b};
  if (o.ev_begin) XMCA_HIP(hipEventRecord(o.ev_begin, st));
  launch_gemm_variant<TI, double, WIDE>(st, p, o.a_kfast, o.b_nfast);
  if (o.ev_end) XMCA_HIP(hipEventRecord(o.ev_end, st));
  hipLaunchKernelGGL((splitk_reduce_kernel<TO>), dim3((unsigned)((stride + 255) / 256)), dim3(256), 0, st, W, splits, stride, C,
                     M, N, ldc, o.alpha, o.beta, o.row_scale, o.col_scale, o.upper_only ? 1 : 0, o.mirror);
  XMCA_HIP(hipGetLastError());
}

}  // namespace xmca
