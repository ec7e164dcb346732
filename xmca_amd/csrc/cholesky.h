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
#include "chol64.h"

namespace xmca {

constexpr int CHOL_NB = C64;      // panel width: the 64 x 64 kernels of chol64.h

// finish the factor: zero the strictly lower triangle (R is then a dense operand for the GEMMs) and put the factored diagonal
// blocks (Dr / Di: 64 x 64 per panel, from chol64_panel_kernel) in place
__global__ void chol_zero_lower_kernel(double* __restrict__ Gr, double* __restrict__ Gi, int64_t ld, int n, const double* __restrict__ Dr,
                                       const double* __restrict__ Di) {
  const int64_t total = (int64_t)n * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i % n);
    if (r > c) {
      Gr[(int64_t)r * ld + c] = 0.0;
      if (Gi) Gi[(int64_t)r * ld + c] = 0.0;
    } else if ((r >> 6) == (c >> 6)) {
      const int64_t o = (int64_t)(r >> 6) * 4096 + (r & 63) * 64 + (c & 63);
      Gr[(int64_t)r * ld + c] = Dr[o];
      if (Gi) Gi[(int64_t)r * ld + c] = Di[o];
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

// one panel: diagonal block + row panel in one launch (chol64.h; the LDS image is dynamic: above 64 KB for complex problems)
template <bool CPLX>
static void chol64_launch_panel(hipStream_t st, double* Gr, double* Gi, int64_t ld, int k0, int nb, int rest, double* Dr, double* Di, int* fail) {
  // (before EVERY launch: the attribute is per device - a second GPU's first complex panel would otherwise ask for ~85 KB of dynamic
  //  LDS above the 64 KB default and fail to launch - and two lane threads could race a once-only initialisation; advisor, round 5)
  XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol64_panel_kernel<CPLX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C64Lds<CPLX>::bytes()));
  hipLaunchKernelGGL((chol64_panel_kernel<CPLX>), dim3(std::max(1, ceil_div(rest, 64))), dim3(256), C64Lds<CPLX>::bytes(), st, Gr, Gi, ld, k0, nb,
                     rest, Dr, Di, fail);
}

// Blocked Cholesky G + delta I = R^H R (R upper triangular) in place on the planes of an n x n Hermitian matrix whose
// upper triangle is valid; delta = rel_shift * max diag (semi-definite Gram matrices of centered / analytic fields).
// On return the strictly lower triangle is zero, so R is a dense GEMM operand.  Returns false when a pivot was not
// positive (the caller then takes the eigen-decomposition route).  Panels of CHOL_NB columns: diagonal block in one
// workgroup and row panel R12 = R11^{-H} A12 by block forward substitution on the matrix pipe in one launch
// (chol64_panel_kernel), and before that the update of the panel's block row by everything left of it as ONE MFMA GEMM.
inline bool cholesky_upper(hipStream_t st, GemmWorkspace& ws, double* Gr, double* Gi, int n, int64_t ld, double rel_shift) {
  const bool cplx = Gi != nullptr;
  DevBuf<unsigned long long> mx;
  DevBuf<int> fail;
  DevBuf<double> slabs;
  DevBuf<unsigned int> tickets;
  const int n_tick = ceil_div(n, 64) + 1;
  unsigned int* tick = tickets.ensure((size_t)n_tick);
  XMCA_HIP(hipMemsetAsync(tick, 0, sizeof(unsigned int) * n_tick, st));
  const int n_cus = ws.cus();
  DevBuf<double> diag;                      // the factored diagonal blocks, 64 x 64 each (chol64_panel_kernel), copied into R at the end
  diag.ensure((size_t)n_tick * 4096 * (cplx ? 2 : 1));
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
  // LEFT-looking over panels of 64 columns (round 5).  The right-looking form of rounds 2-4 rewrote the whole trailing matrix
  // for every panel - a rank-64 update is bound by that traffic, not by its flops (41 us per update at n = 2920, 57 % of the
  // factorisation).  Here a block row is brought up to date just before it is factored,
  //     A[k0 : k0+64, k0 : n] -= R[0 : k0, k0 : k0+64]^H R[0 : k0, k0 : n],
  // one 64 x (n - k0) x k0 product (chol64_rowupdate_kernel) that reads the factor so far and writes 64 rows.
  for (int k0 = 0; k0 < n; k0 += CHOL_NB) {
    const int nb = std::min(CHOL_NB, n - k0), rest = n - k0 - nb;
    if (k0 > 0) {
      // slices of the contraction so that tiles x slices fill the chip about twice; chunks in multiples of 16 rows
      const int ntile = ceil_div(n - k0, 64);
      // (swept 64 ... 768 workgroups on MI355X: flat optimum at 128-256; more slices shorten the products and lengthen the sum)
      int nsplit = std::max(1, std::min(ceil_div(3 * n_cus / 4, ntile), ceil_div(k0, 64)));
      const int kchunk = ceil_div(ceil_div(k0, nsplit), 16) * 16;
      nsplit = ceil_div(k0, kchunk);
      double* sl = nsplit > 1 ? slabs.ensure((size_t)nsplit * ntile * (cplx ? 2 : 1) * 4096) : nullptr;
      if (cplx)
        hipLaunchKernelGGL((chol64_rowupdate_kernel<true>), dim3(ntile * nsplit), dim3(256), 0, st, Gr, Gi, ld, k0, nb, n, kchunk, nsplit, sl, tick);
      else
        hipLaunchKernelGGL((chol64_rowupdate_kernel<false>), dim3(ntile * nsplit), dim3(256), 0, st, Gr, (double*)nullptr, ld, k0, nb, n, kchunk,
                           nsplit, sl, tick);
    }
    double* dr = diag.get() + (size_t)(k0 / CHOL_NB) * 4096, *di = cplx ? dr + (size_t)n_tick * 4096 : nullptr;
    if (cplx) chol64_launch_panel<true>(st, Gr, Gi, ld, k0, nb, rest, dr, di, fail.get());
    else chol64_launch_panel<false>(st, Gr, nullptr, ld, k0, nb, rest, dr, nullptr, fail.get());
    XMCA_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(chol_zero_lower_kernel, ew_grid((int64_t)n * n), dim3(EW_BLOCK), 0, st, Gr, Gi, ld, n, (const double*)diag.get(),
                     cplx ? (const double*)(diag.get() + (size_t)n_tick * 4096) : (const double*)nullptr);
  XMCA_HIP(hipGetLastError());
  int failed = 0;
  XMCA_HIP(hipMemcpyAsync(&failed, fail.get(), sizeof(int), hipMemcpyDeviceToHost, st));
  XMCA_HIP(hipStreamSynchronize(st));
  return failed == 0;
}

}  // namespace xmca
