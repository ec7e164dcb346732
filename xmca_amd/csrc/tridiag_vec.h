// Eigenvectors for the tridiagonal route (tridiag.h):  A = Q T Q^H,  T y_k = lambda_k y_k,  u_k = Q y_k.
//
//   1. trd_twisted_kernel - one lane per eigenvalue: the two stationary factorisations of T - lambda_k I (forward L D+ L^T,
//      backward U D- U^T), the twist index r = argmin |gamma_r| and the vector z(r) = 1, z(i) = -(e_i / D+_i) z(i+1) upwards,
//      z(i+1) = -(e_i / D-_{i+1}) z(i) downwards (Parlett & Dhillon; LAPACK dlar1v without the RRR shifts).  With an
//      eigenvalue accurate to eps ||T|| the residual is eps ||T||; vectors of close eigenvalues are orthogonal only to
//      eps ||T|| / gap - the clean-up below restores that;
//   2. back-transformation with blocks of 64 reflectors in compact WY form, I - V T V^H, all products MFMA GEMMs.  T comes
//      from its inverse, which is explicit: T^{-1} = striu(V^H V) + diag(1 / tau) (from T^{-1} + T^{-H} = V^H V);
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
__global__ __launch_bounds__(64) void trd_twisted_kernel(const double* __restrict__ d, const double* __restrict__ e, int n,
                                                         const double* __restrict__ lam, double* __restrict__ W, double* __restrict__ Yt,
                                                         int64_t ldy) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  const double l = lam[k];
  if (n == 1) { Yt[k] = 1.0; return; }
  double emax = 0.0;
  for (int i = 0; i < n - 1; ++i) emax = fmax(emax, fabs(e[i]));
  const double piv = 2.3e-308 * fmax(1.0, emax * emax) * 1e20 + 1e-300;
  // forward: D+ into Yt
  double dp = d[0] - l;
  for (int i = 0; i < n - 1; ++i) {
    if (!(fabs(dp) >= piv)) dp = dp < 0.0 ? -piv : piv;
    Yt[(int64_t)i * ldy + k] = dp;
    const double ei = e[i];
    dp = (d[i + 1] - l) - (ei / dp) * ei;
  }
  if (!(fabs(dp) >= piv)) dp = dp < 0.0 ? -piv : piv;
  Yt[(int64_t)(n - 1) * ldy + k] = dp;
  // backward: D- into W, gamma_i = D+_i + D-_i - (d_i - lambda), twist at the smallest |gamma|
  double dm = d[n - 1] - l;
  int r = n - 1;
  double gbest = fabs(dp + dm - (d[n - 1] - l));
  if (!(fabs(dm) >= piv)) dm = dm < 0.0 ? -piv : piv;
  W[(int64_t)(n - 1) * ldy + k] = dm;
  for (int i = n - 2; i >= 0; --i) {
    const double ei = e[i];
    dm = (d[i] - l) - (ei / dm) * ei;
    const double g = fabs(Yt[(int64_t)i * ldy + k] + dm - (d[i] - l));
    if (g < gbest) { gbest = g; r = i; }
    if (!(fabs(dm) >= piv)) dm = dm < 0.0 ? -piv : piv;
    W[(int64_t)i * ldy + k] = dm;
  }
  // the vector
  double z = 1.0, nrm = 1.0;
  for (int i = r - 1; i >= 0; --i) {
    z = -(e[i] / Yt[(int64_t)i * ldy + k]) * z;
    Yt[(int64_t)i * ldy + k] = z;
    nrm += z * z;
  }
  z = 1.0;
  Yt[(int64_t)r * ldy + k] = 1.0;
  for (int i = r; i < n - 1; ++i) {
    z = -(e[i] / W[(int64_t)(i + 1) * ldy + k]) * z;
    Yt[(int64_t)(i + 1) * ldy + k] = z;
    nrm += z * z;
  }
  const double s = 1.0 / sqrt(nrm);
  for (int i = 0; i < n; ++i) Yt[(int64_t)i * ldy + k] *= s;
}

// T = (striu(S) + diag(1 / tau))^{-1} for one block of nb <= 64 reflectors; tau = 0 (identity reflector, zero vector
// stored): diagonal entry 1.  One thread per row of T.  S, T: nb x nb planes with leading dimension 64.
template <bool CPLX>
__global__ __launch_bounds__(64) void trd_tfactor_kernel(const double* __restrict__ Sr, const double* __restrict__ Si,
                                                         const double* __restrict__ taur, const double* __restrict__ taui, int nb,
                                                         double* __restrict__ Tr, double* __restrict__ Ti) {
  extern __shared__ __attribute__((aligned(16))) double tf_lds[];
  constexpr int LD = 65;
  double* mr = tf_lds;                  // M = T^{-1}
  double* qr = mr + 64 * LD;            // T
  double* mi = qr + 64 * LD;
  double* qi = mi + (CPLX ? 64 * LD : 0);
  const int r = threadIdx.x;
  for (int c = 0; c < 64; ++c) {
    double a = 0.0, b = 0.0;
    if (r < nb && c < nb) {
      if (c > r) {
        a = Sr[r * 64 + c];
        if (CPLX) b = Si[r * 64 + c];
      } else if (c == r) {
        const double tr = taur[r], ti = CPLX ? taui[r] : 0.0;
        const double den = tr * tr + ti * ti;
        if (den > 0.0) { a = tr / den; b = -ti / den; } else { a = 1.0; }
      }
    } else if (r == c) {
      a = 1.0;
    }
    mr[r * LD + c] = a;
    qr[r * LD + c] = 0.0;
    if (CPLX) { mi[r * LD + c] = b; qi[r * LD + c] = 0.0; }
  }
  __syncthreads();
  // row r of T: T[r][r] = 1 / M[r][r],  T[r][c] = -(sum_{l=r}^{c-1} T[r][l] M[l][c]) / M[c][c]
  for (int c = r; c < 64; ++c) {
    double sr = 0.0, si = 0.0;
    if (c == r) {
      sr = -1.0;
    } else {
      for (int l = r; l < c; ++l) {
        const double tr = qr[r * LD + l], m_r = mr[l * LD + c];
        if (CPLX) {
          const double ti = qi[r * LD + l], m_i = mi[l * LD + c];
          sr += tr * m_r - ti * m_i;
          si += tr * m_i + ti * m_r;
        } else {
          sr += tr * m_r;
        }
      }
    }
    const double dr = mr[c * LD + c], di = CPLX ? mi[c * LD + c] : 0.0, den = dr * dr + di * di;
    qr[r * LD + c] = -(sr * dr + si * di) / den;
    if (CPLX) qi[r * LD + c] = -(si * dr - sr * di) / den;
  }
  for (int c = 0; c < 64; ++c) {
    const bool in = r < nb && c < nb;
    Tr[r * 64 + c] = in ? qr[r * LD + c] : 0.0;
    if (CPLX) Ti[r * 64 + c] = in ? qi[r * LD + c] : 0.0;
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
  const int64_t ld = P.ld;
  const size_t plane = (size_t)n * ld;
  double* Yr = vw.Y[0].ensure(plane);
  double* Yi = cplx ? vw.Y[1].ensure(plane) : nullptr;
  double* Wr = vw.Wk[0].ensure(plane);
  double* Wi = cplx ? vw.Wk[1].ensure(plane) : nullptr;
  hipLaunchKernelGGL(trd_twisted_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, st, P.d, P.e, n, vw.lam_asc.get(), Wr, Yr, ld);
  XMCA_HIP(hipGetLastError());
  if (cplx) XMCA_HIP(hipMemsetAsync(Yi, 0, sizeof(double) * plane, st));
  // ---- Z = H_0 H_1 ... H_{n-2} Yt, blocks of 64 reflectors from the last to the first ----
  const size_t small_doubles = 4 * 64 * 64 + 4 * (size_t)64 * ld;
  double* sm = vw.small.ensure(small_doubles);
  double* Sbr = sm; double* Sbi = Sbr + 64 * 64; double* Tbr = Sbi + 64 * 64; double* Tbi = Tbr + 64 * 64;
  double* Wbr = Tbi + 64 * 64; double* Wbi = Wbr + 64 * ld; double* Xbr = Wbi + 64 * ld; double* Xbi = Xbr + 64 * ld;
  const int nref = n - 1;
  const size_t tf_lds = sizeof(double) * 64 * 65 * (cplx ? 4 : 2);
  if (cplx) XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trd_tfactor_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  else XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trd_tfactor_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  for (int j0 = ((nref - 1) / 64) * 64; j0 >= 0; j0 -= 64) {
    const int nb = std::min(64, nref - j0);
    const int i0 = j0 + 1, mb = n - i0;                      // support of the block: rows i0 .. n-1
    const double* Vbr = P.Vr + (int64_t)j0 * ld + i0;        // Vs[r][i - i0], r < nb  (row-major, k fast)
    const double* Vbi = cplx ? P.Vi + (int64_t)j0 * ld + i0 : nullptr;
    // S_b = V^H V  (nb x nb):  S[r][c] = sum_i conj(Vs[r][i]) Vs[c][i]
    cgemm<double>(st, gws, Vbr, Vbi, ld, true, true, Vbr, Vbi, ld, false, false, Sbr, cplx ? Sbi : nullptr, 64, nb, nb, mb, 1.0, nullptr, nullptr, false);
    if (cplx) hipLaunchKernelGGL(trd_tfactor_kernel<true>, dim3(1), dim3(64), tf_lds, st, Sbr, Sbi, P.tau[0] + j0, P.tau[1] + j0, nb, Tbr, Tbi);
    else hipLaunchKernelGGL(trd_tfactor_kernel<false>, dim3(1), dim3(64), tf_lds, st, Sbr, nullptr, P.tau[0] + j0, nullptr, nb, Tbr, nullptr);
    // W_b = V^H Z[i0:, :]  (nb x n)
    cgemm<double>(st, gws, Vbr, Vbi, ld, true, true, Yr + (int64_t)i0 * ld, cplx ? Yi + (int64_t)i0 * ld : nullptr, ld, true, false, Wbr,
                  cplx ? Wbi : nullptr, ld, nb, n, mb, 1.0, nullptr, nullptr, false);
    // X_b = T W_b
    cgemm<double>(st, gws, Tbr, cplx ? Tbi : nullptr, 64, true, false, Wbr, cplx ? Wbi : nullptr, ld, true, false, Xbr, cplx ? Xbi : nullptr, ld,
                  nb, n, nb, 1.0, nullptr, nullptr, false);
    // Z[i0:, :] -= V X_b      (A(m = i, k = r) = Vs[r][i]: the k-slow orientation)
    cgemm<double>(st, gws, Vbr, Vbi, ld, false, false, Xbr, cplx ? Xbi : nullptr, ld, true, false, Yr + (int64_t)i0 * ld,
                  cplx ? Yi + (int64_t)i0 * ld : nullptr, ld, mb, n, nb, -1.0, nullptr, nullptr, false, 1.0);
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
      cgemm<double>(st, gws, Wr, Wi, ld, true, false, Yr, Yi, ld, false, true, Zr, Zi, ldz, n, n, n, 1.0, nullptr, nullptr, false);
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
