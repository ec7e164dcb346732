// Eigenvectors for the tridiagonal route (tridiag.h):  A = Q T Q^H,  T y_k = lambda_k y_k,  u_k = Q y_k.
//
//   1. trd_twisted_kernel - one lane pair per eigenvalue: the two stationary factorisations of T - lambda_k I (forward L D+ L^T,
//      backward U D- U^T), the twist index r = argmin |gamma_r| and the vector z(r) = 1, z(i) = -(e_i / D+_i) z(i+1) upwards,
//      z(i+1) = -(e_i / D-_{i+1}) z(i) downwards (Parlett & Dhillon; LAPACK dlar1v without the RRR shifts).  With an
//      eigenvalue accurate to eps ||T|| the residual is eps ||T||; vectors of close eigenvalues are orthogonal only to
//      eps ||T|| / gap - the clean-up below restores that;
//   2. back-transformation with blocks of 128 reflectors in compact WY form, Z -= V (T (V^H Z)), the two tall products MFMA
//      GEMMs.  T is never formed: its inverse is explicit, T^{-1} = striu(V^H V) + diag(1 / tau) (from T^{-1} + T^{-H} =
//      V^H V), so T W is a back substitution per column (trd_wy_solve_kernel), and V^H V comes out of the same product as
//      W = V^H Z (the reflectors ride along as 128 extra columns of Z); the 128 x 128 system is solved in two halves of 64
//      with one small product in between;
//   3. S = Z^H Z; max |S - I| decides: <= 0.3 -> one or two Newton-Schulz steps Z <- Z (3/2 I - 1/2 S) (the last one
//      written as (3/2 I - 1/2 S)(rows reversed) Z^H, i.e. straight into the caller's layout: row i = conj(u_i), eigenvalues
//      descending); otherwise (clusters the twisted vectors cannot resolve: repeated eigenvalues, exact null spaces of
//      dimension > 1) the caller falls back to the Jacobi solver.
// CPU model: scripts/experiments/tridiag_model.py.
#pragma once
#include "cholesky.h"   // cgemm
#include "tridiag.h"

namespace xmca {

// Yt[i * ldy + k] = component i of the eigenvector of eigenvalue k (ascending);  W: work plane of the same shape.
// 128 threads per 64 eigenvalues: the forward (D+) and the backward (D-) factorisation are independent recurrences and run
// in the two waves side by side; so do the two halves of the vector (upwards / downwards from the twist index) and of the
// final scaling.  Every loop requests 32 elements ahead of the recurrence (they are memory-latency bound otherwise).
__global__ __launch_bounds__(128) void trd_twisted_kernel(const double* __restrict__ d, const double* __restrict__ e, int n,
                                                         const double* __restrict__ lam, double* __restrict__ W, double* __restrict__ Yt,
                                                         int64_t ldy) {
  __shared__ double nrm_sh[2][64];
  const int lane = threadIdx.x & 63, half = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const bool live = k < n;
  const int kk = live ? k : n - 1;                           // (idle lanes shadow the last eigenvalue and store nothing)
  const double l = lam[kk];
  if (n == 1) { if (live && half == 0) Yt[k] = 1.0; return; }
  constexpr int B = 32;
  double emax = 0.0;
  for (int i = 0; i < n - 1; ++i) emax = fmax(emax, fabs(e[i]));
  const double piv = 2.3e-308 * fmax(1.0, emax * emax) * 1e20 + 1e-300;
  if (half == 0) {
    // forward: D+ into Yt
    double dp = d[0] - l;
    for (int i = 0; i < n - 1; ++i) {
      if (!(fabs(dp) >= piv)) dp = dp < 0.0 ? -piv : piv;
      if (live) Yt[(int64_t)i * ldy + k] = dp;
      const double ei = e[i];
      dp = (d[i + 1] - l) - (ei * trd_rcp(dp)) * ei;
    }
    if (!(fabs(dp) >= piv)) dp = dp < 0.0 ? -piv : piv;
    if (live) Yt[(int64_t)(n - 1) * ldy + k] = dp;
  } else {
    // backward: D- into W
    double dm = d[n - 1] - l;
    if (!(fabs(dm) >= piv)) dm = dm < 0.0 ? -piv : piv;
    if (live) W[(int64_t)(n - 1) * ldy + k] = dm;
    for (int i = n - 2; i >= 0; --i) {
      const double ei = e[i];
      dm = (d[i] - l) - (ei * trd_rcp(dm)) * ei;
      if (!(fabs(dm) >= piv)) dm = dm < 0.0 ? -piv : piv;
      if (live) W[(int64_t)i * ldy + k] = dm;
    }
  }
  __threadfence_block();
  __syncthreads();
  // twist index: gamma_i = D+_i + D-_i - (d_i - lambda), smallest |gamma| (both waves scan: no recurrence, pure streaming)
  int r = 0;
  {
    double gbest = 1.7e308;
    for (int i0 = 0; i0 < n; i0 += B) {
      double yp[B], wm[B];
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int i = i0 + u < n ? i0 + u : n - 1;
        yp[u] = Yt[(int64_t)i * ldy + kk];
        wm[u] = W[(int64_t)i * ldy + kk];
      }
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int i = i0 + u;
        if (i < n) {
          const double g = fabs(yp[u] + wm[u] - (d[i] - l));
          if (g < gbest) { gbest = g; r = i; }
        }
      }
    }
  }
  __syncthreads();                                           // (both waves have read D+ before the upper half overwrites it)
  double nrm = 0.0;
  if (half == 0) {
    // upwards from the twist: z(i) = -(e_i / D+_i) z(i+1)
    double z = 1.0;
    nrm = 1.0;
    if (live) Yt[(int64_t)r * ldy + k] = 1.0;
    for (int i0 = r - 1; i0 >= 0; i0 -= B) {
      double yp[B];
#pragma unroll
      for (int u = 0; u < B; ++u) yp[u] = i0 - u >= 0 ? Yt[(int64_t)(i0 - u) * ldy + kk] : 1.0;
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int i = i0 - u;
        if (i >= 0) {
          z = -(e[i] * trd_rcp(yp[u])) * z;
          if (live) Yt[(int64_t)i * ldy + k] = z;
          nrm += z * z;
        }
      }
    }
  } else {
    // downwards: z(i+1) = -(e_i / D-_{i+1}) z(i)
    double z = 1.0;
    for (int i0 = r; i0 < n - 1; i0 += B) {
      double wm[B];
#pragma unroll
      for (int u = 0; u < B; ++u) wm[u] = i0 + u < n - 1 ? W[(int64_t)(i0 + u + 1) * ldy + kk] : 1.0;
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int i = i0 + u;
        if (i < n - 1) {
          z = -(e[i] * trd_rcp(wm[u])) * z;
          if (live) Yt[(int64_t)(i + 1) * ldy + k] = z;
          nrm += z * z;
        }
      }
    }
  }
  nrm_sh[half][lane] = nrm;
  __threadfence_block();
  __syncthreads();
  const double s = 1.0 / sqrt(nrm_sh[0][lane] + nrm_sh[1][lane]);
  const int h0 = half == 0 ? 0 : n / 2, h1 = half == 0 ? n / 2 : n;
  for (int i0 = h0; i0 < h1; i0 += B) {
    double y[B];
#pragma unroll
    for (int u = 0; u < B; ++u) y[u] = i0 + u < h1 ? Yt[(int64_t)(i0 + u) * ldy + kk] : 0.0;
#pragma unroll
    for (int u = 0; u < B; ++u)
      if (live && i0 + u < h1) Yt[(int64_t)(i0 + u) * ldy + k] = y[u] * s;
  }
}

// Zext[i][r] = Vs[r][i] (i < mb, r < 64; zero for r >= nb): 64 of the block's reflectors as extra columns of Z, so that ONE
// product V^H [Z | V] gives both W = V^H Z and the Gram matrix V^H V
__global__ void trd_vcopy_kernel(const double* __restrict__ Vr, const double* __restrict__ Vi, int64_t ldv, int nb, int mb,
                                 double* __restrict__ Er, double* __restrict__ Ei, int64_t lde) {
  __shared__ double tr[64][65], ti[64][65];
  const int i0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;     // 256 threads: 4 rows at a time
  for (int r = ty; r < 64; r += 4) {
    const int i = i0 + tx;
    const bool in = r < nb && i < mb;
    tr[r][tx] = in ? Vr[(int64_t)r * ldv + i] : 0.0;
    if (Vi) ti[r][tx] = in ? Vi[(int64_t)r * ldv + i] : 0.0;
  }
  __syncthreads();
  for (int ii = ty; ii < 64; ii += 4) {
    const int i = i0 + ii;
    if (i < mb) {
      Er[(int64_t)i * lde + tx] = tr[tx][ii];
      if (Vi) Ei[(int64_t)i * lde + tx] = ti[tx][ii];
    }
  }
}

// Row R of the back substitution x_R = (w_R - sum_{l > R} M[R][l] x_l) tau_R and, recursively, the rows below it in the order
// R, R-1, ..., 0.  The multipliers M[R][l] (LDS, the same for every thread) are requested 16 at a time BEFORE the FMAs that use
// them: left to itself the compiler waits for every LDS operand in turn (one s_waitcnt per two FMAs: 36 us per block of 64
// real reflectors).  Everything is indexed at compile time: x stays in registers.
template <bool CPLX, int R>
struct trd_wy_rows {
  template <int L0>
  static __device__ __forceinline__ void chunk(const double (&xr)[64], const double (&xi)[CPLX ? 64 : 1], const double (*mr)[64],
                                               const double (*mi)[64], double& sr, double& si) {
    if constexpr (L0 < 64) {
      if constexpr (L0 + 15 > R) {
        double mcr[16], mci[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          mcr[u] = mr[R][L0 + u];
          mci[u] = CPLX ? mi[R][L0 + u] : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (L0 + u > R) {
            if constexpr (CPLX) {
              sr -= mcr[u] * xr[L0 + u] - mci[u] * xi[L0 + u];
              si -= mcr[u] * xi[L0 + u] + mci[u] * xr[L0 + u];
            } else {
              sr -= mcr[u] * xr[L0 + u];
            }
          }
        }
      }
      chunk<L0 + 16>(xr, xi, mr, mi, sr, si);
    }
  }
  static __device__ __forceinline__ void run(double (&xr)[64], double (&xi)[CPLX ? 64 : 1], const double (*mr)[64], const double (*mi)[64]) {
    double sr = xr[R], si = CPLX ? xi[R] : 0.0;
    chunk<0>(xr, xi, mr, mi, sr, si);
    // x_R = s / M_RR = s * tau_R (the diagonal of the LDS copy holds tau)
    if constexpr (CPLX) {
      xr[R] = sr * mr[R][R] - si * mi[R][R];
      xi[R] = sr * mi[R][R] + si * mr[R][R];
    } else {
      xr[R] = sr * mr[R][R];
    }
    if constexpr (R > 0) trd_wy_rows<CPLX, R - 1>::run(xr, xi, mr, mi);
  }
};

// X = T W for one block of reflectors without forming T: T^{-1} = M = striu(V^H V) + diag(1 / tau) is explicit, so every
// column of W is back-substituted through M (one thread per column, M in LDS, the column in registers).
// Wx: 64 x ldw planes; columns [0, n) hold W and are overwritten by X, columns [scol, scol + 64) hold V^H V.
template <bool CPLX>
__global__ __launch_bounds__(128) void trd_wy_solve_kernel(double* __restrict__ Wr, double* __restrict__ Wi, int64_t ldw, int n, int nb,
                                                           const double* __restrict__ taur, const double* __restrict__ taui, int scol) {
  // scol: first column of this block's V^H V inside the rows of Wx (n for a block of <= 64 reflectors; the halves of a block
  // of 128 pass their own diagonal block of the 128 x 128 Gram matrix)
  __shared__ __attribute__((aligned(16))) double mr[64][64], mi[CPLX ? 64 : 1][64];
  // (every load of a phase requested before the first use: unconditional loads from clamped addresses, selected afterwards)
  {
    constexpr int PER = 64 * 64 / 128;
    double ga[PER], gb[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = threadIdx.x + 128 * q, r = e >> 6, c = e & 63;
      const int rr = r < nb ? r : 0, cc = c < nb ? c : 0;
      ga[q] = Wr[(int64_t)rr * ldw + scol + cc];
      gb[q] = CPLX ? Wi[(int64_t)rr * ldw + scol + cc] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = threadIdx.x + 128 * q, r = e >> 6, c = e & 63;
      double a = 0.0, b = 0.0;
      if (r < nb && c < nb) {
        if (c > r) {
          a = ga[q];
          b = gb[q];
        } else if (c == r) {
          // the diagonal holds 1 / M_rr = tau_r (tau = 0: the stored reflector is the zero vector, any finite value does)
          a = taur[r];
          if (CPLX) b = taui[r];
        }
      }
      mr[r][c] = a;
      if (CPLX) mi[r][c] = b;
    }
  }
  __syncthreads();
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double xr[64], xi[CPLX ? 64 : 1];
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int rr = r < nb ? r : 0;
    const double vr = Wr[(int64_t)rr * ldw + k], vi = CPLX ? Wi[(int64_t)rr * ldw + k] : 0.0;
    xr[r] = r < nb ? vr : 0.0;
    if (CPLX) xi[r] = r < nb ? vi : 0.0;
  }
  trd_wy_rows<CPLX, 63>::run(xr, xi, mr, mi);
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    if (r < nb) {
      Wr[(int64_t)r * ldw + k] = xr[r];
      if (CPLX) Wi[(int64_t)r * ldw + k] = xi[r];
    }
  }
}

// dst[c][r] = src[r][c] for an n x n plane (64 x 64 tiles through LDS)
__global__ __launch_bounds__(256) void trd_transpose_kernel(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst, int64_t ldd,
                                                            int n) {
  __shared__ double t[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    t[rr][tx] = (r < n && c < n) ? src[(int64_t)r * lds_ + c] : 0.0;
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, r = r0 + tx;
    if (c < n && r < n) dst[(int64_t)c * ldd + r] = t[tx][cc];
  }
}

// out[0] = max |S - I| (as the bit pattern of a non-negative double, atomicMax), NaN -> +inf
__global__ void trd_orth_kernel(const double* __restrict__ Sr, const double* __restrict__ Si, int n, int64_t ld, unsigned long long* out) {
  double m = 0.0;
  for (int r = blockIdx.x; r < n; r += gridDim.x)
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
      const double a = Sr[(int64_t)r * ld + c] - (r == c ? 1.0 : 0.0), b = Si ? Si[(int64_t)r * ld + c] : 0.0;
      double v = fmax(fabs(a), fabs(b));
      if (!(v == v)) v = INFINITY;
      m = fmax(m, v);
    }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// C = 3/2 I - 1/2 S, rows optionally reversed (row i of C = row n-1-i of that matrix)
__global__ void trd_ns_matrix_kernel(const double* __restrict__ Sr, const double* __restrict__ Si, int n, int64_t ld, int reverse,
                                     double* __restrict__ Cr, double* __restrict__ Ci) {
  const int r = blockIdx.x;
  const int src = reverse ? n - 1 - r : r;
  for (int c = threadIdx.x; c < n; c += blockDim.x) {
    Cr[(int64_t)r * ld + c] = (src == c ? 1.5 : 0.0) - 0.5 * Sr[(int64_t)src * ld + c];
    if (Ci) Ci[(int64_t)r * ld + c] = -0.5 * Si[(int64_t)src * ld + c];
  }
}

struct TrdVecWorkspace {
  DevBuf<double> Y[2];        // Yt / Z (n x ld planes)
  DevBuf<double> Wk[2];       // work planes (twisted factorisation; Newton-Schulz)
  DevBuf<double> S[2];        // Gram matrix of the vectors
  DevBuf<double> small;       // S_b, T_b (64 x 64 planes), W_b, X_b (64 x ld planes)
  DevBuf<double> lam_asc;
  DevBuf<unsigned long long> orth;
  double last_orth = 0.0;     // max |Z^H Z - I| before the clean-up of the last call
  int ns_steps = 0;
};

// Eigenvectors of the matrix reduced by trd_reduce(..., keep_reflectors = true): Zr / Zi (n x ldz planes) get row i =
// conj(u_i) for the eigenvalues in DESCENDING order (the layout of hermitian_evd).  Returns false when the vectors of the
// tridiagonal are too far from orthonormal to be repaired (clusters) - the caller then uses the Jacobi solver.
inline bool trd_eigenvectors(hipStream_t st, TrdWorkspace& ws, TrdVecWorkspace& vw, GemmWorkspace& gws, const TrdParams& P, bool cplx,
                             double* Zr, double* Zi, int64_t ldz) {
  const int n = P.n;
  const int64_t ldv = P.ld;                                   // reflectors
  // reflectors per block of the back-transformation: 128 (two skinny GEMMs per block at twice the depth, half the passes over
  // Z; the 128 x 128 system T^{-1} X = W is solved in two halves of 64 with one small product in between) or 64
  constexpr int NBK = 128;       // (blocks of 64 - round 3's first form - took 8.1 instead of 6.5 ms at n = 2920)
  const int64_t ld = ((int64_t)n + NBK + 15) & ~(int64_t)15;   // Z with NBK extra columns (the block's reflectors, see trd_vcopy_kernel)
  const size_t plane = (size_t)n * ld;
  double* Yr = vw.Y[0].ensure(plane);
  double* Yi = cplx ? vw.Y[1].ensure(plane) : nullptr;
  double* Wr = vw.Wk[0].ensure(plane);
  double* Wi = cplx ? vw.Wk[1].ensure(plane) : nullptr;
  hipLaunchKernelGGL(trd_twisted_kernel, dim3(ceil_div(n, 64)), dim3(128), 0, st, P.d, P.e, n, vw.lam_asc.get(), Wr, Yr, ld);
  XMCA_HIP(hipGetLastError());
  if (cplx) XMCA_HIP(hipMemsetAsync(Yi, 0, sizeof(double) * plane, st));
  // ---- Z = H_0 H_1 ... H_{n-2} Yt, blocks of 64 reflectors from the last to the first:  Z -= V (T (V^H Z)) ----
  double* sm = vw.small.ensure(2 * (size_t)NBK * ld);
  double* Wbr = sm; double* Wbi = Wbr + (size_t)NBK * ld;
  const int nref = n - 1;
  auto solve = [&](int row0, int rows, int scol, int jtau) {   // X = T W for the rows [row0, row0 + rows) of the block, in place
    double* wr = Wbr + (int64_t)row0 * ld;
    double* wi = cplx ? Wbi + (int64_t)row0 * ld : nullptr;
    if (cplx) hipLaunchKernelGGL(trd_wy_solve_kernel<true>, dim3(ceil_div(n, 128)), dim3(128), 0, st, wr, wi, ld, n, rows, P.tau[0] + jtau, P.tau[1] + jtau, scol);
    else hipLaunchKernelGGL(trd_wy_solve_kernel<false>, dim3(ceil_div(n, 128)), dim3(128), 0, st, wr, nullptr, ld, n, rows, P.tau[0] + jtau, nullptr, scol);
  };
  for (int j0 = ((nref - 1) / NBK) * NBK; j0 >= 0; j0 -= NBK) {
    const int nb = std::min(NBK, nref - j0);
    const int i0 = j0 + 1, mb = n - i0;                      // support of the block: rows i0 .. n-1
    const double* Vbr = P.Vr + (int64_t)j0 * ldv + i0;       // Vs[r][i - i0], r < nb  (row-major, k fast)
    const double* Vbi = cplx ? P.Vi + (int64_t)j0 * ldv + i0 : nullptr;
    double* Zr0 = Yr + (int64_t)i0 * ld;
    double* Zi0 = cplx ? Yi + (int64_t)i0 * ld : nullptr;
    for (int h = 0; h < NBK; h += 64)                         // (zero columns for the reflectors a short last block does not have)
      hipLaunchKernelGGL(trd_vcopy_kernel, dim3(ceil_div(mb, 64)), dim3(256), 0, st, Vbr + (int64_t)h * ldv, cplx ? Vbi + (int64_t)h * ldv : nullptr,
                         ldv, std::max(0, std::min(64, nb - h)), mb, Zr0 + n + h, cplx ? Zi0 + n + h : nullptr, ld);
    // [W | S] = V^H [Z | V]  (nb x (n + NBK)):  W[r][k] = sum_i conj(Vs[r][i]) Z[i][k]
    cgemm<double>(st, gws, Vbr, Vbi, ldv, true, true, Zr0, Zi0, ld, true, false, Wbr, cplx ? Wbi : nullptr, ld, nb, n + NBK, mb, 1.0, nullptr,
                  nullptr, false);
    // X = T W, in place:  T^{-1} = M = striu(S) + diag(1 / tau) is upper triangular - the rows from 64 on first, then they are
    // taken out of the first 64 rows (W_1 -= M_12 X_2), then those
    if (nb > 64) {
      const int h2 = nb - 64;
      solve(64, h2, n + 64, j0 + 64);
      cgemm<double>(st, gws, Wbr + n + 64, cplx ? Wbi + n + 64 : nullptr, ld, true, false, Wbr + (int64_t)64 * ld, cplx ? Wbi + (int64_t)64 * ld : nullptr, ld,
                    true, false, Wbr, cplx ? Wbi : nullptr, ld, 64, n, h2, -1.0, nullptr, nullptr, false, 1.0);
      solve(0, 64, n, j0);
    } else {
      solve(0, nb, n, j0);
    }
    // Z[i0:, :] -= V X      (A(m = i, k = r) = Vs[r][i]: the k-slow orientation)
    cgemm<double>(st, gws, Vbr, Vbi, ldv, false, false, Wbr, cplx ? Wbi : nullptr, ld, true, false, Zr0, Zi0, ld, mb, n, nb, -1.0, nullptr,
                  nullptr, false, 1.0);
  }
  XMCA_HIP(hipGetLastError());
  // ---- orthonormality of the result, Newton-Schulz clean-up, transposition into the caller's layout ----
  double* Sr = vw.S[0].ensure(plane);
  double* Si = cplx ? vw.S[1].ensure(plane) : nullptr;
  vw.orth.ensure(2);
  vw.ns_steps = 0;
  for (int round = 0; round < 3; ++round) {
    // S = Z^H Z  (Hermitian: upper block triangle, mirrored)
    cgemm<double>(st, gws, Yr, Yi, ld, false, true, Yr, Yi, ld, true, false, Sr, Si, ld, n, n, n, 1.0, nullptr, nullptr, true);
    XMCA_HIP(hipMemsetAsync(vw.orth.get(), 0, sizeof(unsigned long long) * 2, st));
    hipLaunchKernelGGL(trd_orth_kernel, dim3(std::min(n, 1024)), dim3(256), 0, st, Sr, Si, n, ld, vw.orth.get());
    unsigned long long bits = 0;
    XMCA_HIP(hipMemcpyAsync(&bits, vw.orth.get(), sizeof(bits), hipMemcpyDeviceToHost, st));
    XMCA_HIP(hipStreamSynchronize(st));
    double off;
    std::memcpy(&off, &bits, sizeof(off));
    if (round == 0) vw.last_orth = off;
    if (!(off <= 0.3)) return false;                          // clusters (or NaN): not repairable by Newton-Schulz
    const bool last = off <= 1e-6;                            // one more step leaves off^2 ~ 1e-12 or less
    if (last) {
      // out = (3/2 I - 1/2 S)[rows reversed] Z^H :  out[kk][i] = conj(u_{n-1-kk}[i])
      hipLaunchKernelGGL(trd_ns_matrix_kernel, dim3(n), dim3(256), 0, st, Sr, Si, n, ld, 1, Wr, Wi);
      // (through the transposed vectors: the general GEMM kernel runs the k-fast x n-fast orientation at twice the rate of
      //  the k-fast x k-fast one; the transposition is a 0.05 ms pass)
      const dim3 tg((unsigned)ceil_div(n, 64), (unsigned)ceil_div(n, 64));
      hipLaunchKernelGGL(trd_transpose_kernel, tg, dim3(256), 0, st, Yr, ld, Sr, ld, n);
      if (cplx) hipLaunchKernelGGL(trd_transpose_kernel, tg, dim3(256), 0, st, Yi, ld, Si, ld, n);
      cgemm<double>(st, gws, Wr, Wi, ld, true, false, Sr, Si, ld, true, true, Zr, Zi, ldz, n, n, n, 1.0, nullptr, nullptr, false);
      ++vw.ns_steps;
      XMCA_HIP(hipGetLastError());
      return true;
    }
    // Z <- Z (3/2 I - 1/2 S)
    hipLaunchKernelGGL(trd_ns_matrix_kernel, dim3(n), dim3(256), 0, st, Sr, Si, n, ld, 0, Wr, Wi);
    cgemm<double>(st, gws, Yr, Yi, ld, true, false, Wr, Wi, ld, true, false, Sr, Si, ld, n, n, n, 1.0, nullptr, nullptr, false);
    XMCA_HIP(hipMemcpyAsync(Yr, Sr, sizeof(double) * plane, hipMemcpyDeviceToDevice, st));
    if (cplx) XMCA_HIP(hipMemcpyAsync(Yi, Si, sizeof(double) * plane, hipMemcpyDeviceToDevice, st));
    ++vw.ns_steps;
  }
  return false;
}

}  // namespace xmca
