// Hermitian eigenvalues (and, with a clean-up, eigenvectors) by reduction to tridiagonal form - the second eigensolver
// of the library, next to the block Jacobi of jacobi.h.  Replaces LAPACK's *gesdd / *heevd on the T x T stage of
// xmca/array.py:479 and :570 at ~(4/3) n^3 flop instead of the ~100 n^3 of twelve Jacobi sweeps.
//
//   1. Householder tridiagonalisation, unblocked and FUSED: ONE launch per column (trd_step_kernel).  Launch j
//        prologue (every workgroup, redundantly - no gather, no grid barrier): w_{j-1} = p_{j-1} + alpha v_{j-1} from the
//          previous launch's p (own rows of every workgroup) and its partial sums of p^H v; column j of the CURRENT matrix
//          from row j of the stored one (which still lacks the rank-2 update of step j-1) and those two vectors; the
//          reflector v_j, tau_j, beta_j of that column (zlarfg), d_j, e_j;
//        pass over the trailing rows (fixed row -> workgroup ownership, two waves per row, 16-byte accesses):
//          a_ik <- a_ik - v'_i conj(w'_k) - w'_i conj(v'_k)   (the update of step j-1, applied now)
//          p_i = tau_j sum_k a_ik v_k                          (the product the next step needs)
//      so the trailing matrix is read and written once per column and the only synchronisation is the launch boundary
//      (1.5-1.9 us on MI355X against 4-5 us for a grid barrier inside a persistent kernel).  Everything is summed in a
//      fixed order: bit-reproducible, independent of lanes / ranks.
//   2. all eigenvalues of the tridiagonal by Sturm-sequence multisection (trd_bisect_kernel): 16 lanes per eigenvalue
//      evaluate the count at 16 interior points of the current bracket, 14 passes from the Gershgorin interval.
//   3. (eigenvectors, trd_vectors_*) twisted factorisations of T - lambda_k I, one lane per eigenvalue, back-transformation
//      by blocked reflectors (compact WY, GEMMs), Newton-Schulz re-orthonormalisation; clusters the twisted vectors
//      cannot resolve are detected from the Gram matrix of the result and handed to the Jacobi solver.
//
// CPU model of every step: scripts/experiments/tridiag_model.py.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"

namespace xmca {

constexpr int TRD_THREADS = 512;   // 8 waves: rows are processed four at a time, two waves (column halves) per row
constexpr int TRD_ROWS = 4;
constexpr int TRD_MAX_WGS = 256;
constexpr int TRD_MAX_ITERS = 32;  // rows per row slot of a workgroup and launch: n <= 4 * 256 * 32

struct TrdParams {
  double* Ar;          // working copy of the matrix, full Hermitian storage, row-major, ld even, padding columns zero
  double* Ai;          // imaginary plane (complex) or nullptr
  int64_t ld;
  int n;
  double* vb[2][2];    // [parity][re / im]: v_j by global row index
  double* pb[2][2];    // p_j = tau_j A v_j
  double* gp[2][2];    // partial sums of p_j^H v_j per workgroup
  double* tau[2];      // tau_j (re / im)
  double* d;           // diagonal of the tridiagonal matrix
  double* e;           // sub-diagonal (real)
  double* Vr;          // reflectors, row j = v_j (n x ld), or nullptr
  double* Vi;
};

__device__ __forceinline__ double trd_wave_sum(double x) {
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

// sum over the workgroup in a fixed order; every thread gets the same bits.  `red` holds one double per wave.
__device__ __forceinline__ double trd_block_sum(double x, double* red) {
  x = trd_wave_sum(x);
  __syncthreads();                       // (red may still be read from the previous use)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < TRD_THREADS / 64; ++w) s += red[w];
  return s;
}

template <bool CPLX>
__global__ __launch_bounds__(TRD_THREADS) void trd_step_kernel(TrdParams P, int j, int wgs_prev) {
  extern __shared__ __attribute__((aligned(16))) double trd_lds[];
  __shared__ double red[TRD_THREADS / 64];
  __shared__ double part[TRD_MAX_ITERS][TRD_THREADS / 64][2];
  __shared__ double dj_sh;
  const int n = P.n;
  const int tid = threadIdx.x;
  const int b0 = j & ~1;                           // LDS slot of global index k is k - b0: even k <-> even slot
  const int L = ((n - b0 + 2) + 1) & ~1;           // slots per vector (covers k = b0 .. n+1)
  double* svr = trd_lds;                           // v_{j-1}
  double* swr = svr + L;                           // w_{j-1}
  double* sxr = swr + L;                           // column j of the current matrix -> v_j
  double* svi = sxr + L;
  double* swi = svi + (CPLX ? L : 0);
  double* sxi = swi + (CPLX ? L : 0);
  const int prev = (j + 1) & 1, cur = j & 1;
  const int m = n - j - 1;                         // length of the reflector

  // ---- alpha of the previous step: alpha = -1/2 tau (p^H v) ----
  double ar_ = 0.0, ai_ = 0.0;
  if (j > 0) {
    double gr = 0.0, gi = 0.0;
    for (int g = tid; g < wgs_prev; g += TRD_THREADS) {
      gr += P.gp[prev][0][g];
      if (CPLX) gi += P.gp[prev][1][g];
    }
    gr = trd_block_sum(gr, red);
    if (CPLX) gi = trd_block_sum(gi, red);
    const double tr = P.tau[0][j - 1], ti = CPLX ? P.tau[1][j - 1] : 0.0;
    ar_ = -0.5 * (tr * gr - ti * gi);
    ai_ = -0.5 * (tr * gi + ti * gr);
  }
  // ---- v_{j-1}, w_{j-1} = p_{j-1} + alpha v_{j-1} (global indices j .. n-1), zero elsewhere ----
  for (int s = tid; s < L; s += TRD_THREADS) {
    const int k = b0 + s;
    double vr = 0.0, vi = 0.0, wr = 0.0, wi = 0.0;
    if (j > 0 && k >= j && k < n) {
      vr = P.vb[prev][0][k];
      const double pr = P.pb[prev][0][k];
      if (CPLX) {
        vi = P.vb[prev][1][k];
        const double pi = P.pb[prev][1][k];
        wr = pr + ar_ * vr - ai_ * vi;
        wi = pi + ar_ * vi + ai_ * vr;
      } else {
        wr = pr + ar_ * vr;
      }
    }
    svr[s] = vr;
    swr[s] = wr;
    if (CPLX) { svi[s] = vi; swi[s] = wi; }
  }
  __syncthreads();
  // ---- column j of the current matrix: x_k = conj(a_jk) - v_k conj(w_j) - w_k conj(v_j), k >= j ----
  const int sj = j - b0;
  const double wjr = swr[sj], vjr = svr[sj];
  const double wji = CPLX ? swi[sj] : 0.0, vji = CPLX ? svi[sj] : 0.0;
  double xn2 = 0.0;
  for (int s = tid; s < L; s += TRD_THREADS) {
    const int k = b0 + s;
    double xr = 0.0, xi = 0.0;
    if (k >= j && k < n) {
      const double a_r = P.Ar[(int64_t)j * P.ld + k];
      if (CPLX) {
        const double a_i = P.Ai[(int64_t)j * P.ld + k];
        xr = a_r - (svr[s] * wjr + svi[s] * wji) - (swr[s] * vjr + swi[s] * vji);
        xi = -a_i - (svi[s] * wjr - svr[s] * wji) - (swi[s] * vjr - swr[s] * vji);
      } else {
        xr = a_r - svr[s] * wjr - swr[s] * vjr;
      }
      if (k == j) {
        dj_sh = xr;
        xr = 0.0;
        xi = 0.0;
      } else if (k > j + 1) {
        xn2 += xr * xr + xi * xi;
      }
    }
    sxr[s] = xr;
    if (CPLX) sxi[s] = xi;
  }
  xn2 = trd_block_sum(xn2, red);                   // (its barriers also publish sx and dj_sh)
  if (m == 0) {                                    // last launch: only the last diagonal entry
    if (blockIdx.x == 0 && tid == 0) P.d[j] = dj_sh;
    return;
  }
  // ---- reflector (zlarfg): H = I - tau v v^H, v_{j+1} = 1, H^H x = beta e_1 ----
  const double a0r = sxr[sj + 1], a0i = CPLX ? sxi[sj + 1] : 0.0;
  double beta, tr, ti = 0.0, scr = 0.0, sci = 0.0;
  if (xn2 == 0.0 && a0i == 0.0) {
    beta = a0r;
    tr = 0.0;
  } else {
    beta = -copysign(sqrt(a0r * a0r + a0i * a0i + xn2), a0r);
    tr = (beta - a0r) / beta;
    ti = -a0i / beta;
    const double dr = a0r - beta, di = a0i, dn = 1.0 / (dr * dr + di * di);
    scr = dr * dn;
    sci = -di * dn;
  }
  __syncthreads();                                 // everybody has read a0 before it is overwritten
  for (int s = tid; s < L; s += TRD_THREADS) {
    const int k = b0 + s;
    if (k == j + 1) {
      sxr[s] = 1.0;
      if (CPLX) sxi[s] = 0.0;
    } else if (k > j + 1 && k < n) {
      const double xr = sxr[s];
      if (CPLX) {
        const double xi = sxi[s];
        sxr[s] = xr * scr - xi * sci;
        sxi[s] = xr * sci + xi * scr;
      } else {
        sxr[s] = xr * scr;
      }
    }
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int k = j + 1 + tid; k < n; k += TRD_THREADS) {
      const int s = k - b0;
      P.vb[cur][0][k] = sxr[s];
      if (CPLX) P.vb[cur][1][k] = sxi[s];
      if (P.Vr) {
        P.Vr[(int64_t)j * P.ld + k] = sxr[s];
        if (CPLX) P.Vi[(int64_t)j * P.ld + k] = sxi[s];
      }
    }
    if (tid == 0) {
      P.d[j] = dj_sh;
      P.e[j] = beta;
      P.tau[0][j] = tr;
      if (CPLX) P.tau[1][j] = ti;
    }
  }
  // ---- pass over the trailing rows: apply the update of step j-1, multiply by v_j ----
  const int wave = tid >> 6, lane = tid & 63;
  const int rs = wave & (TRD_ROWS - 1), half = wave >> 2;
  const int stride = TRD_ROWS * gridDim.x;
  const int rslot = TRD_ROWS * blockIdx.x + rs;
  int i = rslot;
  if (i < j + 1) i += ((j + 1 - i + stride - 1) / stride) * stride;
  const int ks = (j + 1) & ~1;                     // first column of the pass (even; column j itself when j is even: harmless)
  const int ke = (n + 1) & ~1;                     // one past the last column pair (padding column is zero)
  const int kmid = ks + ((((ke - ks) >> 1) + 127) & ~127);
  const int k_lo = half ? kmid : ks, k_hi = half ? ke : (kmid < ke ? kmid : ke);
  int it = 0;
  for (; i < n; i += stride, ++it) {
    const int si = i - b0;
    const double vpr = svr[si], wpr = swr[si];
    const double vpi = CPLX ? svi[si] : 0.0, wpi = CPLX ? swi[si] : 0.0;
    double* rowr = P.Ar + (int64_t)i * P.ld;
    double* rowi = CPLX ? P.Ai + (int64_t)i * P.ld : nullptr;
    double accr = 0.0, acci = 0.0;
#pragma unroll 2
    for (int k = k_lo + 2 * lane; k < k_hi; k += 128) {
      const int s = k - b0;
      double2 a = *reinterpret_cast<const double2*>(rowr + k);
      const double2 wk = *reinterpret_cast<const double2*>(swr + s);
      const double2 vk = *reinterpret_cast<const double2*>(svr + s);
      const double2 xk = *reinterpret_cast<const double2*>(sxr + s);
      if (CPLX) {
        double2 b = *reinterpret_cast<const double2*>(rowi + k);
        const double2 wki = *reinterpret_cast<const double2*>(swi + s);
        const double2 vki = *reinterpret_cast<const double2*>(svi + s);
        const double2 xki = *reinterpret_cast<const double2*>(sxi + s);
        // a -= v'_i conj(w'_k) + w'_i conj(v'_k)
        a.x -= (vpr * wk.x + vpi * wki.x) + (wpr * vk.x + wpi * vki.x);
        b.x -= (vpi * wk.x - vpr * wki.x) + (wpi * vk.x - wpr * vki.x);
        a.y -= (vpr * wk.y + vpi * wki.y) + (wpr * vk.y + wpi * vki.y);
        b.y -= (vpi * wk.y - vpr * wki.y) + (wpi * vk.y - wpr * vki.y);
        *reinterpret_cast<double2*>(rowr + k) = a;
        *reinterpret_cast<double2*>(rowi + k) = b;
        accr += a.x * xk.x - b.x * xki.x;
        acci += a.x * xki.x + b.x * xk.x;
        accr += a.y * xk.y - b.y * xki.y;
        acci += a.y * xki.y + b.y * xk.y;
      } else {
        a.x -= vpr * wk.x + wpr * vk.x;
        a.y -= vpr * wk.y + wpr * vk.y;
        *reinterpret_cast<double2*>(rowr + k) = a;
        accr += a.x * xk.x;
        accr += a.y * xk.y;
      }
    }
    accr = trd_wave_sum(accr);
    if (CPLX) acci = trd_wave_sum(acci);
    if (lane == 0) {
      part[it][wave][0] = accr;
      part[it][wave][1] = acci;
    }
  }
  __syncthreads();
  // ---- p_i = tau (first half + second half), partial sum of conj(p_i) v_i ----
  double gr = 0.0, gi = 0.0;
  if (tid < TRD_ROWS * TRD_MAX_ITERS) {
    const int it2 = tid >> 2, rs2 = tid & (TRD_ROWS - 1);
    int i2 = TRD_ROWS * blockIdx.x + rs2;
    if (i2 < j + 1) i2 += ((j + 1 - i2 + stride - 1) / stride) * stride;
    i2 += it2 * stride;
    if (i2 < n) {
      const double yr = part[it2][rs2][0] + part[it2][rs2 + TRD_ROWS][0];
      const double yi = CPLX ? part[it2][rs2][1] + part[it2][rs2 + TRD_ROWS][1] : 0.0;
      const double pr = tr * yr - ti * yi, pi = tr * yi + ti * yr;
      P.pb[cur][0][i2] = pr;
      if (CPLX) P.pb[cur][1][i2] = pi;
      const int s = i2 - b0;
      const double vr = sxr[s], vi = CPLX ? sxi[s] : 0.0;
      gr = pr * vr + pi * vi;          // conj(p) v
      gi = pr * vi - pi * vr;
    }
  }
  gr = trd_block_sum(gr, red);
  if (CPLX) gi = trd_block_sum(gi, red);
  if (tid == 0) {
    P.gp[cur][0][blockIdx.x] = gr;
    if (CPLX) P.gp[cur][1][blockIdx.x] = gi;
  }
}

// working copy: W = f * A (f a power of two from max |a_ii|: exact), rows padded to ld with zeros
__global__ void trd_maxdiag_kernel(const double* __restrict__ Ar, int n, int64_t lda, double* scal) {
  __shared__ double red[256];
  double m = 0.0;
  bool bad = false;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double a = Ar[(int64_t)i * lda + i];
    if (!(fabs(a) <= 1.7e308)) bad = true;
    m = fmax(m, fabs(a));
  }
  red[threadIdx.x] = bad ? NAN : m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double g = 0.0;
    bool nan = false;
    for (int t = 0; t < (int)blockDim.x; ++t) {
      if (red[t] != red[t]) nan = true;
      else g = fmax(g, red[t]);
    }
    double f = 1.0;
    if (g > 0.0 && !nan) {
      int ex;
      frexp(g, &ex);
      f = ldexp(1.0, -ex);               // f * g in [0.5, 1)
    }
    scal[0] = f;
    scal[1] = nan ? 1.0 : 0.0;
  }
}

__global__ void trd_copy_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai, int n, int64_t lda, double* Wr, double* Wi,
                                int64_t ld, const double* __restrict__ scal) {
  const double f = scal[0];
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < ld; c += blockDim.x) {
    const bool in = c < n;
    Wr[(int64_t)r * ld + c] = in ? f * Ar[(int64_t)r * lda + c] : 0.0;
    if (Wi) Wi[(int64_t)r * ld + c] = in ? f * Ai[(int64_t)r * lda + c] : 0.0;
  }
}

// ---- all eigenvalues of the symmetric tridiagonal (d, e) by Sturm multisection ------------------------------------
// 16 lanes per eigenvalue: count(x) = number of eigenvalues below x at 16 interior points of the bracket per pass (LDL^T
// recurrence with the pivmin safeguard of LAPACK dstebz; the reciprocal is a hardware seed + two Newton steps, i.e. the
// count of a matrix a few ulps away).  14 passes shrink the Gershgorin interval by 17^14 = 1.7e17.
// lam_desc[n-1-k] = eigenvalue k (ascending) / scale factor.  flag[0] != 0: non-finite input.
constexpr int TRD_BIS_THREADS = 256;
__global__ __launch_bounds__(TRD_BIS_THREADS) void trd_bisect_kernel(const double* __restrict__ d, const double* __restrict__ e, int n,
                                                                   const double* __restrict__ scal, double* lam_desc, int* flag) {
  extern __shared__ __attribute__((aligned(16))) double bis_lds[];
  __shared__ double red[2][TRD_BIS_THREADS / 64];
  double* sd = bis_lds;
  double* se2 = bis_lds + n;
  const int tid = threadIdx.x;
  double gl = 1.7e308, gu = -1.7e308, emax = 0.0;
  bool bad = false;
  for (int i = tid; i < n; i += TRD_BIS_THREADS) {
    const double di = d[i];
    const double el = i > 0 ? e[i - 1] : 0.0, er = i < n - 1 ? e[i] : 0.0;
    if (!(fabs(di) <= 1.7e308) || !(fabs(er) <= 1.7e308)) bad = true;
    sd[i] = di;
    se2[i] = er * er;                          // se2[i] couples i and i+1
    const double r = fabs(el) + fabs(er);
    gl = fmin(gl, di - r);
    gu = fmax(gu, di + r);
    emax = fmax(emax, er * er);
  }
  for (int o = 32; o > 0; o >>= 1) {
    gl = fmin(gl, __shfl_xor(gl, o));
    gu = fmax(gu, __shfl_xor(gu, o));
    emax = fmax(emax, __shfl_xor(emax, o));
  }
  if (__any(bad) && flag) { if ((tid & 63) == 0) atomicOr(flag, 1); }
  if ((tid & 63) == 0) { red[0][tid >> 6] = gl; red[1][tid >> 6] = gu; }
  __syncthreads();
  for (int w = 0; w < TRD_BIS_THREADS / 64; ++w) { gl = fmin(gl, red[0][w]); gu = fmax(gu, red[1][w]); }
  __syncthreads();
  if ((tid & 63) == 0) red[0][tid >> 6] = emax;
  __syncthreads();
  for (int w = 0; w < TRD_BIS_THREADS / 64; ++w) emax = fmax(emax, red[0][w]);
  const double bnorm = fmax(fabs(gl), fabs(gu));
  gl -= 2.2e-16 * bnorm * n + 1e-300;
  gu += 2.2e-16 * bnorm * n + 1e-300;
  const double pivmin = 2.3e-308 * fmax(1.0, emax);
  const int grp = tid >> 4, l = tid & 15;
  const int k = blockIdx.x * (TRD_BIS_THREADS / 16) + grp;       // eigenvalue index, ascending
  const bool live = k < n;
  double lo = gl, hi = gu;
  const int gshift = ((tid & 63) >> 4) * 16;
  for (int pass = 0; pass < 14; ++pass) {
    const double x = lo + (hi - lo) * ((double)(l + 1) * (1.0 / 17.0));
    int cnt = 0;
    double q = sd[0] - x;
    if (fabs(q) < pivmin) q = -pivmin;
    cnt += q < 0.0;
#pragma unroll 4
    for (int i = 1; i < n; ++i) {
      double r = __builtin_amdgcn_rcp(q);
      r = fma(fma(-q, r, 1.0), r, r);
      r = fma(fma(-q, r, 1.0), r, r);
      q = fma(-se2[i - 1], r, sd[i] - x);
      if (fabs(q) < pivmin) q = -pivmin;
      cnt += q < 0.0;
    }
    const unsigned long long mask = __ballot(cnt <= k);
    const int s = __popcll((mask >> gshift) & 0xFFFFull);       // points at or below eigenvalue k (counts are monotone in x)
    const double x_lo = __shfl(x, s > 0 ? s - 1 : 0, 16);
    const double x_hi = __shfl(x, s < 16 ? s : 15, 16);
    if (s > 0) lo = x_lo;
    if (s < 16) hi = x_hi;
  }
  if (live && l == 0) lam_desc[n - 1 - k] = 0.5 * (lo + hi) / scal[0];
}

struct TrdWorkspace {
  DevBuf<double> W[2];        // working copy (re / im)
  DevBuf<double> V[2];        // reflectors (vectors only)
  DevBuf<double> vec;         // vb, pb, gp, tau, d, e
  DevBuf<double> scal;
  DevBuf<int> flag;
  double ms = 0.0;            // (profiling: accumulated device time of the reduction, when measured)
};

inline bool trd_enabled() {
  static const bool on = [] { const char* e = std::getenv("XMCA_TRIDIAG"); return !(e && e[0] == '0'); }();
  return on;
}

// LDS of the step kernel for an n x n problem
inline size_t trd_step_lds(int n, bool cplx) { return (size_t)(((n + 3) + 1) & ~1) * 3 * (cplx ? 2 : 1) * sizeof(double); }

inline bool trd_fits(int n, bool cplx) {
  return trd_step_lds(n, cplx) + 8192 <= (size_t)160 * 1024 && n <= TRD_ROWS * TRD_MAX_WGS * TRD_MAX_ITERS && (size_t)n * 16 <= (size_t)150 * 1024;
}

struct TrdLayout {
  int n;
  int64_t ld;
  size_t nv;
  double* base;
  TrdParams params(double* Wr, double* Wi, double* Vr, double* Vi) const {
    TrdParams P{};
    P.Ar = Wr; P.Ai = Wi; P.ld = ld; P.n = n;
    double* p = base;
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) { P.vb[a][c] = p; p += nv; }
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) { P.pb[a][c] = p; p += nv; }
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) { P.gp[a][c] = p; p += TRD_MAX_WGS; }
    for (int c = 0; c < 2; ++c) { P.tau[c] = p; p += nv; }
    P.d = p; p += nv;
    P.e = p; p += nv;
    P.Vr = Vr; P.Vi = Vi;
    return P;
  }
  static size_t doubles(size_t nv) { return nv * 12 + (size_t)TRD_MAX_WGS * 8; }
};

// Reduces the Hermitian matrix (Ar, Ai) to tridiagonal form on `st`.  Afterwards P.d / P.e hold the tridiagonal of
// f * A (f = ws.scal[0]), P.tau and (keep_reflectors) ws.V the reflectors.  Returns the parameter block.
inline TrdParams trd_reduce(hipStream_t st, TrdWorkspace& ws, const double* Ar, const double* Ai, int n, int64_t lda, bool keep_reflectors) {
  const bool cplx = Ai != nullptr;
  const int64_t ld = ((int64_t)n + 2 + 15) & ~(int64_t)15;
  const size_t nv = (size_t)((n + 8 + 15) & ~15);
  ws.W[0].ensure((size_t)n * ld);
  if (cplx) ws.W[1].ensure((size_t)n * ld);
  if (keep_reflectors) {
    ws.V[0].ensure((size_t)n * ld);
    XMCA_HIP(hipMemsetAsync(ws.V[0].get(), 0, sizeof(double) * (size_t)n * ld, st));
    if (cplx) {
      ws.V[1].ensure((size_t)n * ld);
      XMCA_HIP(hipMemsetAsync(ws.V[1].get(), 0, sizeof(double) * (size_t)n * ld, st));
    }
  }
  ws.vec.ensure(TrdLayout::doubles(nv));
  ws.scal.ensure(4);
  ws.flag.ensure(4);
  XMCA_HIP(hipMemsetAsync(ws.vec.get(), 0, sizeof(double) * TrdLayout::doubles(nv), st));
  XMCA_HIP(hipMemsetAsync(ws.flag.get(), 0, sizeof(int) * 4, st));
  TrdLayout lay{n, ld, nv, ws.vec.get()};
  TrdParams P = lay.params(ws.W[0].get(), cplx ? ws.W[1].get() : nullptr, keep_reflectors ? ws.V[0].get() : nullptr,
                           (keep_reflectors && cplx) ? ws.V[1].get() : nullptr);
  hipLaunchKernelGGL(trd_maxdiag_kernel, dim3(1), dim3(256), 0, st, Ar, n, lda, ws.scal.get());
  hipLaunchKernelGGL(trd_copy_kernel, dim3(n), dim3(256), 0, st, Ar, Ai, n, lda, P.Ar, P.Ai, ld, ws.scal.get());
  const size_t lds = trd_step_lds(n, cplx);
  if (cplx)
    XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trd_step_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  else
    XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trd_step_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  int wgs_prev = 1;
  for (int j = 0; j < n; ++j) {
    const int m = n - j - 1;
    const int wgs = std::max(1, std::min(TRD_MAX_WGS, (m + TRD_ROWS - 1) / TRD_ROWS));
    if (cplx) hipLaunchKernelGGL(trd_step_kernel<true>, dim3(wgs), dim3(TRD_THREADS), lds, st, P, j, wgs_prev);
    else hipLaunchKernelGGL(trd_step_kernel<false>, dim3(wgs), dim3(TRD_THREADS), lds, st, P, j, wgs_prev);
    wgs_prev = wgs;
  }
  XMCA_HIP(hipGetLastError());
  return P;
}

// all eigenvalues, descending, into lam_dev (device, n doubles; may be nullptr) and lam_host.  Synchronises `st`.
inline void trd_eigenvalues(hipStream_t st, TrdWorkspace& ws, const TrdParams& P, std::vector<double>& lam_host, double* lam_dev,
                            DevBuf<double>& lam_tmp) {
  const int n = P.n;
  double* out = lam_dev ? lam_dev : lam_tmp.ensure((size_t)n);
  XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trd_bisect_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  hipLaunchKernelGGL(trd_bisect_kernel, dim3(ceil_div(n, TRD_BIS_THREADS / 16)), dim3(TRD_BIS_THREADS), sizeof(double) * 2 * (size_t)n, st, P.d,
                     P.e, n, ws.scal.get(), out, ws.flag.get());
  XMCA_HIP(hipGetLastError());
  lam_host.resize((size_t)n);
  int flag = 0;
  double scal[2] = {1.0, 0.0};
  XMCA_HIP(hipMemcpyAsync(lam_host.data(), out, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  XMCA_HIP(hipMemcpyAsync(&flag, ws.flag.get(), sizeof(int), hipMemcpyDeviceToHost, st));
  XMCA_HIP(hipMemcpyAsync(scal, ws.scal.get(), sizeof(double) * 2, hipMemcpyDeviceToHost, st));
  XMCA_HIP(hipStreamSynchronize(st));
  XMCA_CHECK(flag == 0 && scal[1] == 0.0, XMCA_ERR_NUMERIC, "SVD failed. NaN entries may be the problem.");
}

}  // namespace xmca
