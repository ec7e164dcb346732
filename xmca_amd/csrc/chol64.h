// 64 x 64 building blocks of the blocked Cholesky factorisation (cholesky.h) and of the panel factorisation of the band
// reduction (band.h), written for latency: a panel of 64 columns is a serial chain of 64 pivots whatever is done, so the
// chain is made of the cheapest steps the chip has.
//
//   chol64_panel_kernel  one panel of the factorisation in ONE launch.  R11 = chol(A11) (upper, R11^H R11 = A11) of the 64 x 64
//                        diagonal block, four waves, the block in LDS, right-looking over 16 x 16 sub-blocks:
//                          * the diagonal sub-block is factored by ONE wave with a column per lane and the 16 rows in
//                            registers; the same row operations run on an identity in lanes 16..31, so the factor and
//                            L^-1 = R^-H come out of the SAME 16 pivot steps (a pivot step = one v_readlane pair, a
//                            reciprocal square root, 15 - j multiplier broadcasts by v_readlane: no LDS, no barrier);
//                          * row panel R[k][j] = L_kk^-1 A[k][j] and trailing update A[i][j] -= R[k][i]^H R[k][j] are
//                            v_mfma_f64_16x16x4_f64 products on LDS operands, dealt to the four waves;
//                        then Y = R11^-H A12 (the `trsm` of the panel loop) by block forward substitution with the four
//                        16 x 16 inverses L_kk^-1 - a 64 x 64 inverse is never formed - one wave per 16 columns, the
//                        right-hand sides in registers.  Every workgroup factors the diagonal block itself.
//   chol64_rowupdate_kernel   the left-looking update of a block row by everything above it (split along the contraction,
//                        partial tiles added in a fixed order).
//
// Round 4 had one thread per 4 x 4 sub-block and a barrier per pivot (24 / 41 us real / complex), a forward substitution
// with one thread per column (40 / 62 us, 154 us inside three surrogate lanes) and a right-looking rank-64 update of the whole
// trailing matrix per panel (41 us, bound by its traffic): DESIGN.md 4, VERDICT r04 weak #3.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"
#include "gemm.h"
#include "kernels.h"

namespace xmca {

constexpr int C64 = 64;        // panel width
constexpr int C64_P = 66;      // LDS pitch of the 64 x 64 images (doubles)
constexpr int C64_DP = 17;     // ... of the 16 x 16 inverses

__device__ __forceinline__ double c64_readlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// acc += op(A) B for one 16 x 16 x 16 block product, operands through index functions: a(i, k), b(k, j)
template <class FA, class FB>
__device__ __forceinline__ d4_t c64_mma(d4_t acc, FA a, FB b, int lane) {
  const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a(lo, 4 * s + hi), b(4 * s + hi, lo), acc, 0, 0, 0);
  return acc;
}

// Joint factorisation and inversion of a 16 x 16 Hermitian positive definite block held in an LDS image (upper triangle valid)
// by the calling wave.  On return the block holds R (upper, strictly lower part zero) and Dr / Di hold L^-1 = R^-H (lower
// triangular, row-major, pitch C64_DP).  Returns true when a pivot was not positive.
template <bool CPLX>
__device__ __forceinline__ bool c64_diag16(double* __restrict__ Mr, double* __restrict__ Mi, int o, double* __restrict__ Dr,
                                           double* __restrict__ Di, int lane) {
  const int c = lane & 31, cc = c & 15;
  const bool ident = c >= 16;
  double ar[16], ai[CPLX ? 16 : 1];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    double vr = 0.0, vi = 0.0;
    if (ident) {
      vr = (r == cc) ? 1.0 : 0.0;
    } else if (r <= cc) {
      vr = Mr[(o + r) * C64_P + o + cc];
      if constexpr (CPLX) vi = (r == cc) ? 0.0 : Mi[(o + r) * C64_P + o + cc];
    }
    ar[r] = vr;
    if constexpr (CPLX) ai[r] = vi;
  }
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double d = c64_readlane(ar[j], j);
    if (!(d > 0.0)) { bad = true; d = 1.0; }
    const double rinv = jac_rsqrt(d);
    ar[j] *= rinv;
    if constexpr (CPLX) ai[j] *= rinv;
#pragma unroll
    for (int r = j + 1; r < 16; ++r) {
      // row r -= conj(R[j][r]) * row j
      const double mr = c64_readlane(ar[j], r);
      ar[r] = fma(-mr, ar[j], ar[r]);
      if constexpr (CPLX) {
        const double mi = c64_readlane(ai[j], r);
        ar[r] = fma(-mi, ai[j], ar[r]);
        ai[r] = fma(-mr, ai[j], ai[r]);
        ai[r] = fma(mi, ar[j], ai[r]);
      }
    }
  }
  if (lane < 32) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (!ident) {
        Mr[(o + r) * C64_P + o + cc] = (r <= cc) ? ar[r] : 0.0;
        if constexpr (CPLX) Mi[(o + r) * C64_P + o + cc] = (r < cc) ? ai[r] : 0.0;
      } else {
        Dr[r * C64_DP + cc] = (r >= cc) ? ar[r] : 0.0;
        if constexpr (CPLX) Di[r * C64_DP + cc] = (r > cc) ? ai[r] : 0.0;
      }
    }
  }
  return bad;
}

// LDS of the kernels (dynamic: the complex images exceed the 64 KB of static LDS): one 64 x 64 image per plane + the four inverses
template <bool CPLX>
struct C64Lds {
  double *Mr, *Mi, *Dr, *Di;
  __device__ __forceinline__ explicit C64Lds(double* base) {
    Mr = base;
    Mi = CPLX ? base + C64 * C64_P : base;
    Dr = base + (CPLX ? 2 : 1) * C64 * C64_P;
    Di = CPLX ? Dr + 4 * 16 * C64_DP : Dr;
  }
  static constexpr size_t bytes() { return sizeof(double) * (size_t)(CPLX ? 2 : 1) * (C64 * C64_P + 4 * 16 * C64_DP); }
};

// acc += A * B with the B operand in registers: a 16 x 16 block in the C/D layout of v_mfma_f64_16x16x4_f64 (row = hi + 4 q,
// column = lo) IS the B operand of the next product (k = 4 s + hi, column = lo): b[s] of this lane, no LDS round trip
template <class FA>
__device__ __forceinline__ d4_t c64_mma_regb(d4_t acc, FA a, d4_t b, int lane) {
  const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a(lo, 4 * s + hi), b[s], acc, 0, 0, 0);
  return acc;
}

// Diagonal block (k0, k0) of the planes -> LDS image: upper triangle as stored, strictly lower part zero, rows / columns beyond
// nb (a short last block) = identity.  All sixteen loads of a thread are in flight together (clamped addresses, selects).
template <bool CPLX>
__device__ __forceinline__ void c64_load_diag(const C64Lds<CPLX>& S, const double* __restrict__ Gr, const double* __restrict__ Gi, int64_t ld,
                                              int k0, int nb, int tid) {
  double vr[16], vi[CPLX ? 16 : 1];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 256 * u, r = e >> 6, c = e & 63;
    const bool in = r < nb && c < nb && r <= c;
    const int64_t off = (int64_t)(k0 + (in ? r : 0)) * ld + k0 + (in ? c : 0);
    vr[u] = Gr[off];
    if constexpr (CPLX) vi[u] = Gi[off];
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 256 * u, r = e >> 6, c = e & 63;
    const bool in = r < nb && c < nb && r <= c;
    const bool pad = (r >= nb || c >= nb) && r == c;
    S.Mr[r * C64_P + c] = in ? vr[u] : (pad ? 1.0 : 0.0);
    if constexpr (CPLX) S.Mi[r * C64_P + c] = (in && r != c) ? vi[u] : 0.0;
  }
}

// R = chol(A) of the 64 x 64 LDS image, in place (upper; strictly lower part zero), and the four inverses L_kk^-1 in S.Dr / S.Di.
// Called by all four waves of the workgroup (barriers inside; the image must be complete and a barrier passed).  Returns true
// (in wave 0) when a pivot was not positive.
template <bool CPLX>
__device__ __forceinline__ bool c64_factor(const C64Lds<CPLX>& S, int wave, int lane) {
  bool bad = false;
  const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const int ok = 16 * kb;
    double* Dr = S.Dr + kb * 16 * C64_DP;
    double* Di = S.Di + (CPLX ? kb * 16 * C64_DP : 0);
    if (wave == 0) bad = c64_diag16<CPLX>(S.Mr, S.Mi, ok, Dr, Di, lane) || bad;
    __syncthreads();
    if (kb == 3) break;
    // row panel: R[kb][jb] = L_kk^-1 A[kb][jb], one block per wave
    if (wave < 3 - kb) {
      const int oj = 16 * (kb + 1 + wave);
      d4_t cr = {0.0, 0.0, 0.0, 0.0}, ci = {0.0, 0.0, 0.0, 0.0};
      auto lr = [&](int i, int k) { return Dr[i * C64_DP + k]; };
      auto br = [&](int k, int j) { return S.Mr[(ok + k) * C64_P + oj + j]; };
      cr = c64_mma(cr, lr, br, lane);
      if constexpr (CPLX) {
        auto li = [&](int i, int k) { return Di[i * C64_DP + k]; };
        auto nli = [&](int i, int k) { return -Di[i * C64_DP + k]; };
        auto bi = [&](int k, int j) { return S.Mi[(ok + k) * C64_P + oj + j]; };
        cr = c64_mma(cr, nli, bi, lane);
        ci = c64_mma(ci, lr, bi, lane);
        ci = c64_mma(ci, li, br, lane);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        S.Mr[(ok + hi + 4 * q) * C64_P + oj + lo] = cr[q];
        if constexpr (CPLX) S.Mi[(ok + hi + 4 * q) * C64_P + oj + lo] = ci[q];
      }
    }
    __syncthreads();
    // trailing update: A[ib][jb] -= R[kb][ib]^H R[kb][jb], kb < ib <= jb
    {
      const int nrem = 3 - kb, nblk = nrem * (nrem + 1) / 2;
      for (int t = wave; t < nblk; t += 4) {
        int ib = 0, u = t;
        while (u >= nrem - ib) { u -= nrem - ib; ++ib; }
        const int jb = ib + u;
        const int oi = 16 * (kb + 1 + ib), oj = 16 * (kb + 1 + jb);
        d4_t cr, ci;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          cr[q] = S.Mr[(oi + hi + 4 * q) * C64_P + oj + lo];
          if constexpr (CPLX) ci[q] = S.Mi[(oi + hi + 4 * q) * C64_P + oj + lo];
        }
        auto nar = [&](int i, int k) { return -S.Mr[(ok + k) * C64_P + oi + i]; };       // -conj(R)^T, real part
        auto br = [&](int k, int j) { return S.Mr[(ok + k) * C64_P + oj + j]; };
        cr = c64_mma(cr, nar, br, lane);
        if constexpr (CPLX) {
          auto ai_ = [&](int i, int k) { return S.Mi[(ok + k) * C64_P + oi + i]; };
          auto nai = [&](int i, int k) { return -S.Mi[(ok + k) * C64_P + oi + i]; };
          auto bi = [&](int k, int j) { return S.Mi[(ok + k) * C64_P + oj + j]; };
          // -(Rr - i Ri)^T (Br + i Bi):  re = -Rr^T Br - Ri^T Bi,  im = -Rr^T Bi + Ri^T Br
          cr = c64_mma(cr, nai, bi, lane);
          ci = c64_mma(ci, nar, bi, lane);
          ci = c64_mma(ci, ai_, br, lane);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          S.Mr[(oi + hi + 4 * q) * C64_P + oj + lo] = cr[q];
          if constexpr (CPLX) S.Mi[(oi + hi + 4 * q) * C64_P + oj + lo] = ci[q];
        }
      }
    }
    __syncthreads();
  }
  return bad;
}

// One panel of the Cholesky factorisation in ONE launch: R11 = chol(A11) (the diagonal block (k0, k0), nb <= 64 rows, upper
// triangle read; written by workgroup 0 to the side buffer Dr / Di) and the row panel Y = R11^-H A12 in place
// (A12 = rows k0 .. k0 + nb - 1, columns k0 + 64 + [0, rest)).  grid = max(1, ceil(rest / 64)):
//   * EVERY workgroup factors the diagonal block itself in its LDS (the same instructions on the same numbers: the same bits) -
//     a launch boundary and a round trip of R11 through memory cost more than the 64 pivots do, and nobody waits for anybody;
//   * its 16 (32) right-hand-side values per lane are requested BEFORE the factorisation and arrive under it;
//   * wave w then solves its 16 columns by block forward substitution, Y[i] = L_ii^-1 (A[i] - sum_{k<i} R[k][i]^H Y[k]), on
//     the matrix pipe with the right-hand sides in registers (c64_mma_regb).
// *fail = 1 when a pivot was not positive.
template <bool CPLX>
__global__ __launch_bounds__(256) void chol64_panel_kernel(double* __restrict__ Gr, double* __restrict__ Gi, int64_t ld, int k0, int nb, int rest,
                                                           double* __restrict__ Dr, double* __restrict__ Di, int* __restrict__ fail) {
  extern __shared__ double c64_smem[];
  C64Lds<CPLX> S(c64_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 15, hi = lane >> 4;
  const int c0 = blockIdx.x * 64;
  const int64_t base = (int64_t)k0 * ld + k0 + C64 + c0;
  const int col = c0 + 16 * wave + lo;        // this lane's column of A12
  const bool live = col < rest;
  c64_load_diag<CPLX>(S, Gr, Gi, ld, k0, nb, tid);
  d4_t ar[4], ai[CPLX ? 4 : 1];
#pragma unroll
  for (int ib = 0; ib < 4; ++ib)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 16 * ib + hi + 4 * q;
      const bool ok = live && r < nb;
      const int64_t off = ok ? base + (int64_t)r * ld + 16 * wave + lo : (int64_t)k0 * ld + k0;
      const double vr = Gr[off];
      ar[ib][q] = ok ? vr : 0.0;
      if constexpr (CPLX) {
        const double vi = Gi[off];
        ai[ib][q] = ok ? vi : 0.0;
      }
    }
  __syncthreads();
  const bool bad = c64_factor<CPLX>(S, wave, lane);
  if (blockIdx.x == 0) {
    // R11 goes to a side buffer (Dr / Di: 64 x 64 row-major), NOT over A11: the other workgroups of this launch read A11 whenever
    // they get their compute unit - with other streams' kernels on the device that can be after this one is done (found in
    // round 5 as a factorisation that failed now and then inside three surrogate lanes).  cholesky.h copies the diagonal blocks
    // into the factor at the end.
    if (bad && lane == 0) *fail = 1;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = tid + 256 * u, r = e >> 6, c = e & 63;
      Dr[e] = (r <= c) ? S.Mr[r * C64_P + c] : 0.0;
      if constexpr (CPLX) Di[e] = (r < c) ? S.Mi[r * C64_P + c] : 0.0;
    }
  }
  if (rest <= 0) return;
  d4_t yr[4], yi[CPLX ? 4 : 1];
#pragma unroll
  for (int ib = 0; ib < 4; ++ib) {
    const int oi = 16 * ib;
    d4_t cr = ar[ib], ci = {0.0, 0.0, 0.0, 0.0};
    if constexpr (CPLX) ci = ai[ib];
#pragma unroll
    for (int kb = 0; kb < ib; ++kb) {
      const int ok = 16 * kb;
      auto nar = [&](int i, int k) { return -S.Mr[(ok + k) * C64_P + oi + i]; };
      cr = c64_mma_regb(cr, nar, yr[kb], lane);
      if constexpr (CPLX) {
        auto ai_ = [&](int i, int k) { return S.Mi[(ok + k) * C64_P + oi + i]; };
        auto nai = [&](int i, int k) { return -S.Mi[(ok + k) * C64_P + oi + i]; };
        // -(Rr - i Ri)^T (Yr + i Yi)
        cr = c64_mma_regb(cr, nai, yi[kb], lane);
        ci = c64_mma_regb(ci, nar, yi[kb], lane);
        ci = c64_mma_regb(ci, ai_, yr[kb], lane);
      }
    }
    const double* Dr = S.Dr + ib * 16 * C64_DP;
    const double* Di = S.Di + (CPLX ? ib * 16 * C64_DP : 0);
    d4_t zr = {0.0, 0.0, 0.0, 0.0}, zi = {0.0, 0.0, 0.0, 0.0};
    auto lr = [&](int i, int k) { return Dr[i * C64_DP + k]; };
    zr = c64_mma_regb(zr, lr, cr, lane);
    if constexpr (CPLX) {
      auto li = [&](int i, int k) { return Di[i * C64_DP + k]; };
      auto nli = [&](int i, int k) { return -Di[i * C64_DP + k]; };
      zr = c64_mma_regb(zr, nli, ci, lane);
      zi = c64_mma_regb(zi, lr, ci, lane);
      zi = c64_mma_regb(zi, li, cr, lane);
      yi[ib] = zi;
    }
    yr[ib] = zr;
    if (live) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = oi + hi + 4 * q;
        if (r < nb) {
          Gr[base + (int64_t)r * ld + 16 * wave + lo] = zr[q];
          if constexpr (CPLX) Gi[base + (int64_t)r * ld + 16 * wave + lo] = zi[q];
        }
      }
    }
  }
}

// Left-looking update of one block row before it is factored (cholesky.h):
//     A[k0 : k0+64, k0 : n] -= R[0 : k0, k0 : k0+64]^H R[0 : k0, k0 : n]
// - 64 rows are written, the factor so far is read once.  Workgroup (t, s): 64 x 64 tile t of the block row, slice s of the
// contraction; wave w owns 16 columns of the tile and all four 16-row blocks.  Operands go from global memory (L2) straight into
// the MFMA operand registers: for a fixed k the 16 values of an operand are contiguous (a row of R), so a load instruction
// fetches four 128-byte row segments.  Several slices: the partial tiles go to `slabs` with write-through stores, a ticket per
// tile, and the LAST arriver adds them in slice order (same bits whatever the timing) and updates A.
template <bool CPLX>
__global__ __launch_bounds__(256) void chol64_rowupdate_kernel(double* __restrict__ Gr, double* __restrict__ Gi, int64_t ld, int k0, int nb,
                                                               int n, int kchunk, int nsplit, double* __restrict__ slabs,
                                                               unsigned int* __restrict__ tickets) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 15, hi = lane >> 4;
  const int ntile = gridDim.x / nsplit;
  const int t = blockIdx.x % ntile, sl = blockIdx.x / ntile;
  const int col = k0 + 64 * t + 16 * wave + lo;          // column of A / R this lane works on
  const int colc = col < n ? col : n - 1;                // (clamped for the loads; never stored)
  const int kbeg = sl * kchunk, kend = min(k0, kbeg + kchunk);
  d4_t cr[4], ci[CPLX ? 4 : 1];
#pragma unroll
  for (int ib = 0; ib < 4; ++ib) {
    cr[ib] = d4_t{0.0, 0.0, 0.0, 0.0};
    if constexpr (CPLX) ci[ib] = d4_t{0.0, 0.0, 0.0, 0.0};
  }
  // operands of 16 contraction rows: one B value and four A values per MFMA k-step.  Two sets: the loads of the next 16 rows
  // are in flight while the products of the current ones run (a k-step is 4 x 64 cycles of MFMA, a load from L2 / HBM 1-2 us).
  struct Ops { double br[4], bi[CPLX ? 4 : 1], ar[4][4], ai[CPLX ? 4 : 1][CPLX ? 4 : 1]; };
  auto fetch = [&](Ops& o, int k) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k + 4 * u + hi;
      const bool live = kk < kend;
      const int64_t row = (int64_t)(live ? kk : kbeg) * ld;
      o.br[u] = live ? Gr[row + colc] : 0.0;
      if constexpr (CPLX) o.bi[u] = live ? Gi[row + colc] : 0.0;
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) {
        o.ar[u][ib] = live ? Gr[row + k0 + 16 * ib + lo] : 0.0;
        if constexpr (CPLX) o.ai[u][ib] = live ? Gi[row + k0 + 16 * ib + lo] : 0.0;
      }
    }
  };
  auto multiply = [&](const Ops& o) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) {
        cr[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.ar[u][ib], o.br[u], cr[ib], 0, 0, 0);
        if constexpr (CPLX) {
          // conj(a) b = (ar br + ai bi) + i (ar bi - ai br)
          cr[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.ai[u][ib], o.bi[u], cr[ib], 0, 0, 0);
          ci[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.ar[u][ib], o.bi[u], ci[ib], 0, 0, 0);
          ci[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(-o.ai[u][ib], o.br[u], ci[ib], 0, 0, 0);
        }
      }
  };
  Ops o0, o1;
  if (kbeg < kend) fetch(o0, kbeg);
  for (int k = kbeg; k < kend; k += 32) {
    if (k + 16 < kend) fetch(o1, k + 16);
    multiply(o0);
    if (k + 16 < kend) {
      if (k + 32 < kend) fetch(o0, k + 32);
      multiply(o1);
    }
  }
  constexpr int PL = CPLX ? 2 : 1;
  if (nsplit > 1) {
    double* mine = slabs + ((size_t)sl * ntile + t) * (PL * 4096);
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __hip_atomic_store(mine + (ib * 4 + q) * 256 + tid, cr[ib][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (CPLX) __hip_atomic_store(mine + 4096 + (ib * 4 + q) * 256 + tid, ci[ib][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    // Hand-off form: 8-byte agent-scope (sc1, write-through) atomics on BOTH sides - slab stores here, slab loads by the last arriver
    // below - drained before the workgroup barrier that precedes the ticket (MI355X_MICROARCH.md, "valid forms besides R1 / R2":
    // {8-B agent atomics both sides}); the acquire of the last arriver drops lines of the previous panel's slabs from this CU's L1.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ unsigned int last_sh;
    __syncthreads();
    if (tid == 0) {
      const unsigned int got = __hip_atomic_fetch_add(tickets + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_sh = (got == (unsigned int)(nsplit - 1)) ? 1u : 0u;
      if (last_sh) {
        // acquire: this XCD's L2 may still hold the slab lines of the PREVIOUS panel's launch (same addresses) - without it the
        // sums below were wrong now and then once three surrogate lanes interleaved their launches (a non-positive pivot later:
        // the solver fell back to its other route, results right, 5 of 9 surrogates slow)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(tickets + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
      }
    }
    __syncthreads();
    if (!last_sh) return;
    // slices in ascending order, four at a time: the 64 (128) loads of a batch are independent and in flight together
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
      cr[ib] = d4_t{0.0, 0.0, 0.0, 0.0};
      if constexpr (CPLX) ci[ib] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
    for (int s0 = 0; s0 < nsplit; s0 += 4) {
      double vr[4][16], vi[CPLX ? 4 : 1][CPLX ? 16 : 1];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int s2 = s0 + d < nsplit ? s0 + d : nsplit - 1;
        const double* sb = slabs + ((size_t)s2 * ntile + t) * (PL * 4096);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          vr[d][e] = __hip_atomic_load(sb + e * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if constexpr (CPLX) vi[d][e] = __hip_atomic_load(sb + 4096 + e * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        if (s0 + d >= nsplit) break;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          cr[e >> 2][e & 3] += vr[d][e];
          if constexpr (CPLX) ci[e >> 2][e & 3] += vi[d][e];
        }
      }
    }
  }
  if (col < n) {
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * ib + hi + 4 * q;
        if (r < nb) {
          Gr[(int64_t)(k0 + r) * ld + col] -= cr[ib][q];
          if constexpr (CPLX) Gi[(int64_t)(k0 + r) * ld + col] -= ci[ib][q];
        }
      }
  }
}

}  // namespace xmca
