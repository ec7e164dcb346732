// Kernels of the blocked Cholesky factorisation G = R^H R (R upper triangular) used for values-only two-field solves
// (rule_n) and for the Cholesky LR step of the eigensolver; cholesky_upper (below) holds the panel loop and its GEMMs.
// cgemm, the complex-planes GEMM wrapper the rest of the library uses, also lives here.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "common.h"
#include "gemm.h"
#include "kernels.h"

namespace xmca {

constexpr int CHOL_NB = 64;

// Diagonal block k0..k0+nb of the (already updated) upper triangle of G, one workgroup of 256 threads:
//   R11 = chol(A11) (upper, R11^H R11 = A11) written back over A11 (strictly lower part zeroed),
//   Lt  = R11^T (row i = column i of R11, what the forward substitution of chol_trsm_kernel walks along) and
//   dinv[i] = 1 / R11[i][i] to scratch.
// The block lives in registers: thread (ty, tx) of a 16 x 16 grid owns the 4 x 4 sub-block (4 ty.., 4 tx..); each of
// the 64 elimination steps broadcasts the pivot row through LDS (double-buffered: one barrier per step).
// A non-positive or NaN pivot sets *fail (the caller falls back to the eigen-decomposition route).
template <bool CPLX>
__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ Gr, double* __restrict__ Gi, int64_t ld, int k0, int nb,
                                                        double* __restrict__ Lt_r, double* __restrict__ Lt_i,
                                                        double* __restrict__ dinv, int* __restrict__ fail) {
  __shared__ double row_r[2][CHOL_NB], row_i[CPLX ? 2 : 1][CPLX ? CHOL_NB : 1];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  double ar[4][4], ai[CPLX ? 4 : 1][CPLX ? 4 : 1];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * ty + i, cc = 4 * tx + j;
      double vr = (r == cc) ? 1.0 : 0.0, vi = 0.0;            // padding rows/cols: identity
      if (r < nb && cc < nb) {
        vr = 0.0;
        if (r <= cc) {
          vr = Gr[(int64_t)(k0 + r) * ld + k0 + cc];
          if constexpr (CPLX) vi = (r == cc) ? 0.0 : Gi[(int64_t)(k0 + r) * ld + k0 + cc];
        }
      }
      ar[i][j] = vr;
      if constexpr (CPLX) ai[i][j] = vi;
    }
  bool bad = false;
#pragma unroll
  for (int j = 0; j < CHOL_NB; ++j) {
    const int buf = j & 1;
    if (ty == j / 4) {                                          // owners of row j publish it (unscaled)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        row_r[buf][4 * tx + q] = ar[j % 4][q];
        if constexpr (CPLX) row_i[buf][4 * tx + q] = ai[j % 4][q];
      }
    }
    __syncthreads();
    double d = row_r[buf][j];
    if (!(d > 0.0)) { bad = true; d = 1.0; }
    const double inv = jac_rsqrt(d), inv2 = inv * inv;
    if (ty == j / 4) {                                          // R[j][c] = A[j][c] / sqrt(d)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ar[j % 4][q] *= inv;
        if constexpr (CPLX) ai[j % 4][q] *= inv;
      }
    }
    // A[r][c] -= conj(A[j][r]) A[j][c] / d   for r, c > j (entries at or left of the pivot are never read again)
    double pr[4], pi[CPLX ? 4 : 1], qr[4], qi[CPLX ? 4 : 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pr[q] = row_r[buf][4 * ty + q] * inv2;
      qr[q] = row_r[buf][4 * tx + q];
      if constexpr (CPLX) {
        pi[q] = row_i[buf][4 * ty + q] * inv2;
        qi[q] = row_i[buf][4 * tx + q];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * ty + i <= j) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ar[i][q] -= pr[i] * qr[q];
        if constexpr (CPLX) {
          ar[i][q] -= pi[i] * qi[q];
          ai[i][q] -= pr[i] * qi[q] - pi[i] * qr[q];
        }
      }
    }
  }
  if (bad && tid == 0) *fail = 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * ty + i, cc = 4 * tx + j;
      const double vr = (r <= cc) ? ar[i][j] : 0.0;
      Lt_r[cc * CHOL_NB + r] = vr;
      if (r == cc) dinv[r] = jac_rcp(vr);
      if (r < nb && cc < nb) Gr[(int64_t)(k0 + r) * ld + k0 + cc] = vr;
      if constexpr (CPLX) {
        const double vi = (r < cc) ? ai[i][j] : 0.0;
        Lt_i[cc * CHOL_NB + r] = vi;
        if (r < nb && cc < nb) Gi[(int64_t)(k0 + r) * ld + k0 + cc] = vi;
      }
    }
}

// Row panel R12 = R11^{-H} A12 in place: one thread per column of A12 (64 per workgroup, so that the panel spreads over
// as many CUs as it has wave-sized column groups), forward substitution along the rows of Lt.  Lt (broadcast reads) and
// the columns (lane = bank, conflict-free) both live in LDS; eight rows of Lt x column entries are in flight at a time.
template <bool CPLX>
__global__ __launch_bounds__(64) void chol_trsm_kernel(double* __restrict__ Gr, double* __restrict__ Gi, int64_t ld, int k0, int rest,
                                                       const double* __restrict__ Lt_r, const double* __restrict__ Lt_i,
                                                       const double* __restrict__ dinv, double* __restrict__ S1 = nullptr,
                                                       double* __restrict__ S2 = nullptr, int64_t lds = 0) {
  // S1 / S2 (complex, optional): the panel once more as the stacked real operands [Re R; Im R] and [Im R; -Re R] (128 x rest,
  // pitch lds) - with them the Hermitian rank-64 update is two real products of depth 128 instead of four of depth 64
  __shared__ double Lr[CHOL_NB][CHOL_NB], Li[CPLX ? CHOL_NB : 1][CPLX ? CHOL_NB : 1];
  __shared__ double Yr[CHOL_NB][64], Yi[CPLX ? CHOL_NB : 1][CPLX ? 64 : 1];
  __shared__ double dv_s[CHOL_NB];
  const int lane = threadIdx.x;
  const int c = blockIdx.x * 64 + lane;
  const bool live = c < rest;
  const int64_t base = (int64_t)k0 * ld + k0 + CHOL_NB + (live ? c : 0);
#pragma unroll 16
  for (int i = 0; i < CHOL_NB; ++i) {
    Yr[i][lane] = Gr[base + (int64_t)i * ld];
    if constexpr (CPLX) Yi[i][lane] = Gi[base + (int64_t)i * ld];
  }
#pragma unroll 16
  for (int e = lane; e < CHOL_NB * CHOL_NB; e += 64) {
    (&Lr[0][0])[e] = Lt_r[e];
    if constexpr (CPLX) (&Li[0][0])[e] = Lt_i[e];
  }
  dv_s[lane] = dinv[lane];
  __syncthreads();
  for (int i = 0; i < CHOL_NB; ++i) {
    double s0 = Yr[i][lane], s1 = 0.0, t0 = 0.0, t1 = 0.0;    // independent chains; real part s0 + s1, imaginary t0 + t1
    if constexpr (CPLX) t0 = Yi[i][lane];
    int k = 0;
    for (; k + 8 <= i; k += 8) {                                // y[i] -= conj(R[k][i]) y[k]
      double lr[8], yr[8], li[CPLX ? 8 : 1], yi[CPLX ? 8 : 1];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        lr[q] = Lr[i][k + q];
        yr[q] = Yr[k + q][lane];
        if constexpr (CPLX) {
          li[q] = Li[i][k + q];
          yi[q] = Yi[k + q][lane];
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if constexpr (CPLX) {
          s0 -= lr[q] * yr[q];
          s1 -= li[q] * yi[q];
          t0 -= lr[q] * yi[q];
          t1 += li[q] * yr[q];
        } else {
          if (q & 1) s1 -= lr[q] * yr[q];
          else s0 -= lr[q] * yr[q];
        }
      }
    }
    for (; k < i; ++k) {
      const double lr = Lr[i][k], yr = Yr[k][lane];
      s0 -= lr * yr;
      if constexpr (CPLX) {
        const double li = Li[i][k], yi = Yi[k][lane];
        s1 -= li * yi;
        t0 -= lr * yi;
        t1 += li * yr;
      }
    }
    const double dv = dv_s[i];
    Yr[i][lane] = (s0 + s1) * dv;
    if constexpr (CPLX) Yi[i][lane] = (t0 + t1) * dv;
  }
  if (live) {
#pragma unroll 16
    for (int i = 0; i < CHOL_NB; ++i) {
      Gr[base + (int64_t)i * ld] = Yr[i][lane];
      if constexpr (CPLX) Gi[base + (int64_t)i * ld] = Yi[i][lane];
    }
    if constexpr (CPLX) {
      if (S1) {
#pragma unroll 16
        for (int i = 0; i < CHOL_NB; ++i) {
          const double yr = Yr[i][lane], yi = Yi[i][lane];
          S1[(int64_t)i * lds + c] = yr;
          S1[(int64_t)(CHOL_NB + i) * lds + c] = yi;
          S2[(int64_t)i * lds + c] = yi;
          S2[(int64_t)(CHOL_NB + i) * lds + c] = -yr;
        }
      }
    }
  }
}

// zero the strictly lower triangle (the factor is then a dense operand for the GEMMs) and add `delta` to nothing
__global__ void chol_zero_lower_kernel(double* __restrict__ Gr, double* __restrict__ Gi, int64_t ld, int n) {
  const int64_t total = (int64_t)n * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i % n);
    if (r > c) {
      Gr[(int64_t)r * ld + c] = 0.0;
      if (Gi) Gi[(int64_t)r * ld + c] = 0.0;
    }
  }
}

// G[i][i] += delta
__global__ void chol_shift_diag_kernel(double* __restrict__ Gr, int64_t ld, int n, double delta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) Gr[(int64_t)i * ld + i] += delta;
}

// largest diagonal entry (scale of the regularisation)
__global__ void chol_max_diag_kernel(const double* __restrict__ Gr, int64_t ld, int n, unsigned long long* __restrict__ out) {
  double mx = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) mx = fmax(mx, Gr[(int64_t)i * ld + i]);
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0 && mx > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(mx));
}

// smallest and largest diagonal entry of a matrix with a positive diagonal (bit patterns order like the values)
__global__ void chol_minmax_diag_kernel(const double* __restrict__ Gr, int64_t ld, int n, unsigned long long* __restrict__ out) {
  double mx = 0.0, mn = HUGE_VAL;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double v = Gr[(int64_t)i * ld + i];
    mx = fmax(mx, v);
    mn = fmin(mn, v);
  }
  for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o)); mn = fmin(mn, __shfl_xor(mn, o)); }
  if ((threadIdx.x & 63) == 0) {
    if (mx > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(mx));
    if (mn >= 0.0) atomicMin(out + 1, (unsigned long long)__double_as_longlong(mn));
  }
}

// C = alpha * rs * cs * opA(A) * opB(B) on complex planes through 1-4 real MFMA GEMMs.
//   conj flags negate the imaginary plane of the operand; `herm` computes only the upper block triangle and
//   mirrors (C must then be Hermitian by construction).
template <typename TI>
void cgemm(hipStream_t st, GemmWorkspace& ws, const TI* Ar, const TI* Ai, int64_t lda, bool a_kfast, bool conjA, const TI* Br,
           const TI* Bi, int64_t ldb, bool b_nfast, bool conjB, double* Cr, double* Ci, int64_t ldc, int M, int N, int K,
           double alpha, const double* row_scale, const double* col_scale, bool herm, double beta0 = 0.0,
           const GemmTileList* tiles = nullptr) {
  // beta0: C = ... + beta0 * C (both planes);  tiles: block-sparse product (gemm.h GemmTileList)
  GemmOpts o;
  o.tiles = tiles;
  o.a_kfast = a_kfast;
  o.b_nfast = b_nfast;
  o.row_scale = row_scale;
  o.col_scale = col_scale;
  o.upper_only = herm;
  const double sa = conjA ? -1.0 : 1.0, sb = conjB ? -1.0 : 1.0;
  // real part: Ar Br - sa sb Ai Bi
  o.alpha = alpha; o.beta = beta0; o.mirror = herm ? 1 : 0;
  gemm<TI, double>(st, ws, Ar, lda, Br, ldb, Cr, ldc, M, N, K, o);
  if (Ai && Bi) {
    o.alpha = -sa * sb * alpha; o.beta = 1.0;
    gemm<TI, double>(st, ws, Ai, lda, Bi, ldb, Cr, ldc, M, N, K, o);
  }
  if (!Ci) return;
  // imaginary part: sb Ar Bi + sa Ai Br
  o.mirror = herm ? -1 : 0;
  bool first = true;
  if (Bi) {
    o.alpha = sb * alpha; o.beta = beta0;
    gemm<TI, double>(st, ws, Ar, lda, Bi, ldb, Ci, ldc, M, N, K, o);
    first = false;
  }
  if (Ai) {
    o.alpha = sa * alpha; o.beta = first ? beta0 : 1.0;
    gemm<TI, double>(st, ws, Ai, lda, Br, ldb, Ci, ldc, M, N, K, o);
    first = false;
  }
  if (first && beta0 == 0.0) XMCA_HIP(hipMemsetAsync(Ci, 0, sizeof(double) * (size_t)M * ldc, st));
}

// Blocked Cholesky G + delta I = R^H R (R upper triangular) in place on the planes of an n x n Hermitian matrix whose
// upper triangle is valid; delta = rel_shift * max diag (semi-definite Gram matrices of centered / analytic fields).
// On return the strictly lower triangle is zero, so R is a dense GEMM operand.  Returns false when a pivot was not
// positive (the caller then takes the eigen-decomposition route).  Panels of CHOL_NB columns: diagonal block in one
// workgroup, row panel R12 = R11^{-H} A12 by forward substitution (a thread per column), trailing update
// A22 -= R12^H R12 as an MFMA GEMM.
inline bool cholesky_upper(hipStream_t st, GemmWorkspace& ws, double* Gr, double* Gi, int n, int64_t ld, double rel_shift) {
  const bool cplx = Gi != nullptr;
  DevBuf<double> lt_r, lt_i, dinv;
  DevBuf<unsigned long long> mx;
  DevBuf<int> fail;
  lt_r.ensure((size_t)CHOL_NB * CHOL_NB);
  if (cplx) lt_i.ensure((size_t)CHOL_NB * CHOL_NB);
  DevBuf<double> stack;
  const int64_t lds = ((int64_t)n + 15) & ~(int64_t)15;
  double* s1 = cplx ? stack.ensure((size_t)4 * CHOL_NB * lds) : nullptr;
  double* s2 = cplx ? s1 + (size_t)2 * CHOL_NB * lds : nullptr;
  dinv.ensure(CHOL_NB);
  XMCA_HIP(hipMemsetAsync(mx.ensure(1), 0, sizeof(unsigned long long), st));
  XMCA_HIP(hipMemsetAsync(fail.ensure(1), 0, sizeof(int), st));
  hipLaunchKernelGGL(chol_max_diag_kernel, dim3(std::min(ceil_div(n, 256), 64)), dim3(256), 0, st, Gr, ld, n, mx.get());
  unsigned long long bits = 0;
  XMCA_HIP(hipMemcpyAsync(&bits, mx.get(), sizeof(bits), hipMemcpyDeviceToHost, st));
  XMCA_HIP(hipStreamSynchronize(st));
  double maxdiag = 0.0;
  std::memcpy(&maxdiag, &bits, sizeof(double));
  if (!(maxdiag > 0.0) || !std::isfinite(maxdiag)) return false;
  hipLaunchKernelGGL(chol_shift_diag_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, Gr, ld, n, rel_shift * maxdiag);
  for (int k0 = 0; k0 < n; k0 += CHOL_NB) {
    const int nb = std::min(CHOL_NB, n - k0), rest = n - k0 - nb;
    if (cplx) {
      hipLaunchKernelGGL((chol_diag_kernel<true>), dim3(1), dim3(256), 0, st, Gr, Gi, ld, k0, nb, lt_r.get(), lt_i.get(), dinv.get(), fail.get());
      if (rest > 0)
        hipLaunchKernelGGL((chol_trsm_kernel<true>), dim3(ceil_div(rest, 64)), dim3(64), 0, st, Gr, Gi, ld, k0, rest, lt_r.get(), lt_i.get(),
                           dinv.get(), s1, s2, lds);
    } else {
      hipLaunchKernelGGL((chol_diag_kernel<false>), dim3(1), dim3(256), 0, st, Gr, (double*)nullptr, ld, k0, nb, lt_r.get(),
                         (double*)nullptr, dinv.get(), fail.get());
      if (rest > 0)
        hipLaunchKernelGGL((chol_trsm_kernel<false>), dim3(ceil_div(rest, 64)), dim3(64), 0, st, Gr, (double*)nullptr, ld, k0, rest,
                           lt_r.get(), (const double*)nullptr, dinv.get());
    }
    XMCA_HIP(hipGetLastError());
    if (rest <= 0) break;
    const int64_t o12 = (int64_t)k0 * ld + k0 + nb, o22 = (int64_t)(k0 + nb) * ld + k0 + nb;
    // A22 -= R12^H R12   (upper block triangle, mirrored)
    if (cplx) {
      // Re: S1^T S1,  Im: S1^T S2  with S1 = [Re R12; Im R12], S2 = [Im R12; -Re R12] (written by the substitution kernel):
      // two launches of depth 128 instead of four of depth 64 - the updates of a 2500-row factorisation are launch-bound
      GemmOpts o;
      o.a_kfast = false; o.b_nfast = true; o.alpha = -1.0; o.beta = 1.0; o.upper_only = true;
      o.mirror = 1;
      gemm<double, double>(st, ws, s1, lds, s1, lds, Gr + o22, ld, rest, rest, 2 * CHOL_NB, o);
      o.mirror = -1;
      gemm<double, double>(st, ws, s1, lds, s2, lds, Gi + o22, ld, rest, rest, 2 * CHOL_NB, o);
    } else {
      cgemm<double>(st, ws, Gr + o12, nullptr, ld, false, true, Gr + o12, nullptr, ld, true, false, Gr + o22, nullptr, ld, rest, rest, nb, -1.0,
                    nullptr, nullptr, true, 1.0);
    }
  }
  hipLaunchKernelGGL(chol_zero_lower_kernel, ew_grid((int64_t)n * n), dim3(EW_BLOCK), 0, st, Gr, Gi, ld, n);
  XMCA_HIP(hipGetLastError());
  int failed = 0;
  XMCA_HIP(hipMemcpyAsync(&failed, fail.get(), sizeof(int), hipMemcpyDeviceToHost, st));
  XMCA_HIP(hipStreamSynchronize(st));
  return failed == 0;
}

}  // namespace xmca
