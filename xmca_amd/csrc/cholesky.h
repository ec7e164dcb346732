// Kernels of the blocked Cholesky factorisation G = R^H R (R upper triangular) used for values-only two-field solves
// (rule_n): the panel loop and its GEMMs are in solver.h (cholesky_upper).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace xmca {

constexpr int CHOL_NB = 64;

// Diagonal block k0..k0+nb of the (already updated) upper triangle of G, one workgroup of 256 threads:
//   R11 = chol(A11) (upper, R11^H R11 = A11) written back over A11's upper triangle (strictly lower part zeroed),
//   Rinv = R11^{-1} (upper) to a CHOL_NB x CHOL_NB scratch block (row-major, ld = CHOL_NB).
// A non-positive or NaN pivot sets *fail (the caller falls back to the eigen-decomposition route).
template <bool CPLX>
__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ Gr, double* __restrict__ Gi, int64_t ld, int k0, int nb,
                                                        double* __restrict__ Rinv_r, double* __restrict__ Rinv_i,
                                                        int* __restrict__ fail) {
  __shared__ double Ar[CHOL_NB][CHOL_NB + 1], Ai[CPLX ? CHOL_NB : 1][CPLX ? CHOL_NB + 1 : 1];
  __shared__ double Xr[CHOL_NB][CHOL_NB + 1], Xi[CPLX ? CHOL_NB : 1][CPLX ? CHOL_NB + 1 : 1];
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  for (int e = tid; e < CHOL_NB * CHOL_NB; e += 256) {
    const int r = e / CHOL_NB, c = e % CHOL_NB;
    double vr = (r == c) ? 1.0 : 0.0, vi = 0.0;             // padding rows/cols: identity
    if (r < nb && c < nb && r <= c) {
      vr = Gr[(int64_t)(k0 + r) * ld + k0 + c];
      if constexpr (CPLX) vi = Gi[(int64_t)(k0 + r) * ld + k0 + c];
    } else if (r < nb && c < nb) {
      vr = 0.0;
    }
    Ar[r][c] = vr;
    if constexpr (CPLX) Ai[r][c] = (r == c) ? 0.0 : vi;
  }
  __syncthreads();
  // right-looking, upper: row j is scaled by 1/sqrt(pivot), then A[r][c] -= conj(R[j][r]) R[j][c] for j < r <= c
  for (int j = 0; j < CHOL_NB; ++j) {
    const double d = Ar[j][j];
    if (!(d > 0.0)) { if (tid == 0) bad = 1; break; }      // uniform: every thread reads the same value
    const double inv = 1.0 / sqrt(d);
    __syncthreads();
    if (tid >= j && tid < CHOL_NB) {
      Ar[j][tid] *= inv;
      if constexpr (CPLX) Ai[j][tid] *= inv;
    }
    __syncthreads();
    const int rem = CHOL_NB - 1 - j;                        // rows j+1 .. NB-1
    for (int e = tid; e < rem * rem; e += 256) {
      const int r = j + 1 + e / rem, c = j + 1 + e % rem;
      if (r > c) continue;
      const double ar = Ar[j][r], br = Ar[j][c];
      double pr = ar * br;
      if constexpr (CPLX) {
        const double ai = Ai[j][r], bi = Ai[j][c];
        pr += ai * bi;                                      // conj(a) b
        Ai[r][c] -= ar * bi - ai * br;
      }
      Ar[r][c] -= pr;
    }
    __syncthreads();
  }
  __syncthreads();
  if (bad) {
    if (tid == 0) *fail = 1;
    return;
  }
  // X = R^{-1}: thread c solves R x = e_c by back substitution (upper triangular: x[i] = 0 for i > c)
  if (tid < CHOL_NB) {
    const int c = tid;
    for (int i = CHOL_NB - 1; i >= 0; --i) {
      double sr = (i == c) ? 1.0 : 0.0, si = 0.0;
      for (int k = i + 1; k <= c; ++k) {
        const double rr = Ar[i][k], xr = Xr[k][c];
        sr -= rr * xr;
        if constexpr (CPLX) {
          const double ri = Ai[i][k], xi = Xi[k][c];
          sr += ri * xi;
          si -= rr * xi + ri * xr;
        }
      }
      const double dinv = 1.0 / Ar[i][i];                   // the diagonal of R is real
      Xr[i][c] = (i <= c) ? sr * dinv : 0.0;
      if constexpr (CPLX) Xi[i][c] = (i <= c) ? si * dinv : 0.0;
    }
  }
  __syncthreads();
  for (int e = tid; e < CHOL_NB * CHOL_NB; e += 256) {
    const int r = e / CHOL_NB, c = e % CHOL_NB;
    Rinv_r[e] = Xr[r][c];
    if constexpr (CPLX) Rinv_i[e] = Xi[r][c];
    if (r < nb && c < nb) {
      Gr[(int64_t)(k0 + r) * ld + k0 + c] = (r <= c) ? Ar[r][c] : 0.0;
      if constexpr (CPLX) Gi[(int64_t)(k0 + r) * ld + k0 + c] = (r < c) ? Ai[r][c] : 0.0;
    }
  }
}

// G[k0 + r][c0 + c] = src[r][c]   (rows x cols block; src ld = lds) - places the solved row panel R12
__global__ void chol_place_kernel(const double* __restrict__ src, int64_t lds, double* __restrict__ G, int64_t ld, int k0, int c0,
                                  int rows, int cols) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    G[(int64_t)(k0 + r) * ld + c0 + c] = src[(int64_t)r * lds + c];
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

}  // namespace xmca
