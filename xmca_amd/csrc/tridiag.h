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
  unsigned long long* prof;   // (XMCA_TRD_PROF) 8 s_memtime stamps of workgroup 0 per launch, or nullptr
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

// NS: LDS slots per thread (the vectors have at most NS * TRD_THREADS slots); PF: 128-column chunks of a wave's first row
// requested before the prologue starts.
constexpr int TRD_PF = 4;
// wave sum with DPP row operations (no LDS crossbar): pairs, quads, half rows, rows, then the four row sums in a fixed order
template <int CTRL>
__device__ __forceinline__ double trd_dpp_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double trd_wave_sum_dpp(double x) {
  x += trd_dpp_f64<0xB1>(x);     // quad_perm [1,0,3,2]
  x += trd_dpp_f64<0x4E>(x);     // quad_perm [2,3,0,1]
  x += trd_dpp_f64<0x141>(x);    // row_half_mirror
  x += trd_dpp_f64<0x140>(x);    // row_mirror: every lane of a 16-lane row holds the row sum
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return (r0 + r1) + (r2 + r3);
}
// 1 / b and sqrt(s) from the hardware seeds + Newton steps (the IEEE division / sqrt sequences of the compiler are
// ~200-cycle dependent chains; these sit on the serial path of every column)
__device__ __forceinline__ double trd_rcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double trd_div(double a, double b) {
  const double r = trd_rcp(b);
  const double q = a * r;
  return fma(fma(-q, b, a), r, q);
}
__device__ __forceinline__ double trd_sqrt(double s) {
  if (!(s > 0.0)) return s == 0.0 ? 0.0 : NAN;
  double r = __builtin_amdgcn_rsq(s);
  r = r * fma(-0.5 * s * r, r, 1.5);
  r = r * fma(-0.5 * s * r, r, 1.5);
  double g = s * r;
  g = fma(0.5 * r, fma(-g, g, s), g);
  return g;
}

template <int NWAVES>
__device__ __forceinline__ double trd_block_sum_n(double x, double* red) {
  x = trd_wave_sum_dpp(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < NWAVES; ++w) s += red[w];
  return s;
}

// a value every lane holds alike -> scalar registers (the resident kernels have no vector register to spare for column scalars)
__device__ __forceinline__ double trd_uniform(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

// ... with ONE barrier: the caller hands in a buffer that nobody can still be reading - the resident kernel alternates two by
// the column parity, and every column has further barriers between two uses of the same one.  Two sums ride on one barrier.
template <int NWAVES>
__device__ __forceinline__ void trd_block_sum2_1b(double& a, double& b, double (*buf)[2]) {
  a = trd_wave_sum_dpp(a);
  b = trd_wave_sum_dpp(b);
  if ((threadIdx.x & 63) == 0) { buf[threadIdx.x >> 6][0] = a; buf[threadIdx.x >> 6][1] = b; }
  __syncthreads();
  double s = 0.0, t = 0.0;
#pragma unroll
  for (int w = 0; w < NWAVES; ++w) { s += buf[w][0]; t += buf[w][1]; }
  a = s;
  b = t;
}

template <bool CPLX, int NS>
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
  const bool prof = P.prof && blockIdx.x == 0 && tid == 0;
  if (prof) P.prof[8 * j + 0] = __builtin_amdgcn_s_memtime();

  // ---- every global load of the prologue is requested up front (one memory latency instead of four) ----
  double gr = 0.0, gi = 0.0, tpr = 0.0, tpi = 0.0;
  if (j > 0) {
    if (tid < wgs_prev) {
      gr = P.gp[prev][0][tid];
      if (CPLX) gi = P.gp[prev][1][tid];
    }
    tpr = P.tau[0][j - 1];
    if (CPLX) tpi = P.tau[1][j - 1];
  }
  double vre[NS], pre[NS], are[NS], vim[NS], pim[NS], aim[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const int s = tid + t * TRD_THREADS, k = b0 + s;
    const bool in = s < L && k >= j && k < n;
    vre[t] = pre[t] = are[t] = vim[t] = pim[t] = aim[t] = 0.0;
    if (in) {
      are[t] = P.Ar[(int64_t)j * P.ld + k];
      if (CPLX) aim[t] = P.Ai[(int64_t)j * P.ld + k];
      if (j > 0) {
        vre[t] = P.vb[prev][0][k];
        pre[t] = P.pb[prev][0][k];
        if (CPLX) {
          vim[t] = P.vb[prev][1][k];
          pim[t] = P.pb[prev][1][k];
        }
      }
    }
  }
  // ... and the first chunks of this wave's first row of the pass
  const int wave = tid >> 6, lane = tid & 63;
  const int rs = wave & (TRD_ROWS - 1), half = wave >> 2;
  const int stride = TRD_ROWS * gridDim.x;
  int i = TRD_ROWS * blockIdx.x + rs;
  if (i < j + 1) i += ((j + 1 - i + stride - 1) / stride) * stride;
  const int ks = (j + 1) & ~1;                     // first column of the pass (even; column j itself when j is even: harmless)
  const int ke = (n + 1) & ~1;                     // one past the last column pair (padding column is zero)
  const int kmid = ks + ((((ke - ks) >> 1) + 127) & ~127);
  const int k_lo = half ? kmid : ks, k_hi = half ? ke : (kmid < ke ? kmid : ke);
  double2 fa[TRD_PF], fb[TRD_PF];
#pragma unroll
  for (int c = 0; c < TRD_PF; ++c) {
    const int k = k_lo + 2 * lane + 128 * c;
    fa[c] = make_double2(0.0, 0.0);
    fb[c] = make_double2(0.0, 0.0);
    if (i < n && k < k_hi) {
      fa[c] = *reinterpret_cast<const double2*>(P.Ar + (int64_t)i * P.ld + k);
      if (CPLX) fb[c] = *reinterpret_cast<const double2*>(P.Ai + (int64_t)i * P.ld + k);
    }
  }

  if (prof) P.prof[8 * j + 1] = __builtin_amdgcn_s_memtime();
  // ---- alpha of the previous step: alpha = -1/2 tau (p^H v) ----
  double ar_ = 0.0, ai_ = 0.0;
  if (j > 0) {
    gr = trd_block_sum(gr, red);
    if (CPLX) gi = trd_block_sum(gi, red);
    ar_ = -0.5 * (tpr * gr - tpi * gi);
    ai_ = -0.5 * (tpr * gi + tpi * gr);
  }
  if (prof) P.prof[8 * j + 2] = __builtin_amdgcn_s_memtime();
  // ---- v_{j-1}, w_{j-1} = p_{j-1} + alpha v_{j-1} (global indices j .. n-1), zero elsewhere ----
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const int s = tid + t * TRD_THREADS;
    if (s < L) {
      svr[s] = vre[t];
      if (CPLX) {
        svi[s] = vim[t];
        pre[t] = pre[t] + ar_ * vre[t] - ai_ * vim[t];      // (p <- w)
        pim[t] = pim[t] + ar_ * vim[t] + ai_ * vre[t];
        swi[s] = pim[t];
      } else {
        pre[t] = pre[t] + ar_ * vre[t];
      }
      swr[s] = pre[t];
    }
  }
  __syncthreads();
  // ---- column j of the current matrix: x_k = conj(a_jk) - v_k conj(w_j) - w_k conj(v_j), k >= j ----
  const int sj = j - b0;
  const double wjr = swr[sj], vjr = svr[sj];
  const double wji = CPLX ? swi[sj] : 0.0, vji = CPLX ? svi[sj] : 0.0;
  double xn2 = 0.0;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const int s = tid + t * TRD_THREADS, k = b0 + s;
    if (s < L) {
      double xr = 0.0, xi = 0.0;
      if (k >= j && k < n) {
        if (CPLX) {
          xr = are[t] - (vre[t] * wjr + vim[t] * wji) - (pre[t] * vjr + pim[t] * vji);
          xi = -aim[t] - (vim[t] * wjr - vre[t] * wji) - (pim[t] * vjr - pre[t] * vji);
        } else {
          xr = are[t] - vre[t] * wjr - pre[t] * vjr;
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
      are[t] = xr;                                 // kept for the scaling below
      aim[t] = xi;
    }
  }
  xn2 = trd_block_sum(xn2, red);                   // (its barriers also publish sx and dj_sh)
  if (prof) P.prof[8 * j + 3] = __builtin_amdgcn_s_memtime();
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
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const int s = tid + t * TRD_THREADS, k = b0 + s;
    if (s < L && k > j && k < n) {
      double vr, vi = 0.0;
      if (k == j + 1) {
        vr = 1.0;
      } else if (CPLX) {
        vr = are[t] * scr - aim[t] * sci;
        vi = are[t] * sci + aim[t] * scr;
      } else {
        vr = are[t] * scr;
      }
      sxr[s] = vr;
      if (CPLX) sxi[s] = vi;
      if (blockIdx.x == 0) {
        P.vb[cur][0][k] = vr;
        if (CPLX) P.vb[cur][1][k] = vi;
        if (P.Vr) {                                  // (tau = 0: H_j = I - the stored reflector is the zero vector)
          const bool live = tr != 0.0 || ti != 0.0;
          P.Vr[(int64_t)j * P.ld + k] = live ? vr : 0.0;
          if (CPLX) P.Vi[(int64_t)j * P.ld + k] = live ? vi : 0.0;
        }
      }
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    P.d[j] = dj_sh;
    P.e[j] = beta;
    P.tau[0][j] = tr;
    if (CPLX) P.tau[1][j] = ti;
  }
  __syncthreads();
  if (prof) P.prof[8 * j + 4] = __builtin_amdgcn_s_memtime();
  // ---- pass over the trailing rows: apply the update of step j-1, multiply by v_j ----
  int it = 0;
  for (; i < n; i += stride, ++it) {
    const int si = i - b0;
    const double vpr = svr[si], wpr = swr[si];
    const double vpi = CPLX ? svi[si] : 0.0, wpi = CPLX ? swi[si] : 0.0;
    double* rowr = P.Ar + (int64_t)i * P.ld;
    double* rowi = CPLX ? P.Ai + (int64_t)i * P.ld : nullptr;
    double accr = 0.0, acci = 0.0;
    auto body = [&](int k, double2 a, double2 b) {
      const int s = k - b0;
      const double2 wk = *reinterpret_cast<const double2*>(swr + s);
      const double2 vk = *reinterpret_cast<const double2*>(svr + s);
      const double2 xk = *reinterpret_cast<const double2*>(sxr + s);
      if (CPLX) {
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
    };
    int k = k_lo + 2 * lane;
    if (it == 0) {
#pragma unroll
      for (int c = 0; c < TRD_PF; ++c) {
        if (k < k_hi) body(k, fa[c], fb[c]);
        k += 128;
      }
    }
#pragma unroll 2
    for (; k < k_hi; k += 128) {
      const double2 a = *reinterpret_cast<const double2*>(rowr + k);
      double2 b = make_double2(0.0, 0.0);
      if (CPLX) b = *reinterpret_cast<const double2*>(rowi + k);
      body(k, a, b);
    }
    accr = trd_wave_sum(accr);
    if (CPLX) acci = trd_wave_sum(acci);
    if (lane == 0) {
      part[it][wave][0] = accr;
      part[it][wave][1] = acci;
    }
  }
  __syncthreads();
  if (prof) P.prof[8 * j + 5] = __builtin_amdgcn_s_memtime();
  // ---- p_i = tau (first half + second half), partial sum of conj(p_i) v_i ----
  double gor = 0.0, goi = 0.0;
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
      gor = pr * vr + pi * vi;          // conj(p) v
      goi = pr * vi - pi * vr;
    }
  }
  gor = trd_block_sum(gor, red);
  if (CPLX) goi = trd_block_sum(goi, red);
  if (tid == 0) {
    P.gp[cur][0][blockIdx.x] = gor;
    if (CPLX) P.gp[cur][1][blockIdx.x] = goi;
  }
  if (prof) P.prof[8 * j + 6] = __builtin_amdgcn_s_memtime();
}

// -------------------------------------------------------------------------------------------------------------------
// The same reduction as ONE persistent launch with the trailing matrix RESIDENT IN REGISTERS (trd_resident_kernel).
// The launch-per-column form above streams the trailing matrix through HBM once per column (read + write: 133 GB for
// n = 2920, 6 TB/s -> 22 of its 39 ms).  Here every wave keeps RR whole rows of the matrix in its VGPRs (lane l holds the
// column pairs 128 c + 2 l of a row: NC double2 per row and plane; 256 workgroups x 4 waves (one per SIMD: 512 registers
// per lane) x RR rows = the last 1024 RR
// rows - the ones that live longest; earlier rows are streamed from global memory by their owner wave until they die), the
// three vectors of a step live in LDS as before, and a column costs one all-to-all exchange instead of a pass over HBM:
//   pass: every wave updates its rows, p_i = tau_j (row . v_j) -> published (write-through stores) together with the
//         workgroup's partial p^H v and, by its owner, row j+1 of the stored matrix; drain; ONE epoch flag per workgroup;
//   exchange: one wave polls the flags of all workgroups (bounded), then every thread reads p, the partial sums and row
//         j+1 with agent-scope (sc1) loads - cdna_hip_programming.md G16 form R1 with sc1 loads in place of the acquire;
//   prologue: as in the step kernel, redundantly in every workgroup.
// Buffers alternate with the column parity (a workgroup is at most one column ahead of the slowest).  The grid must be
// resident: one workgroup per CU (the LDS request guarantees it is alone), launched under a process-wide mutex so that two
// lanes never interleave two persistent grids; a workgroup that is missing makes every spin run out -> give_up -> the host
// repeats the reduction with the launch-per-column kernels.
struct TrdSync {
  double* pub[2][2];      // [parity][re / im]  p_j by global row index
  double* gpart[2][2];    // partial p^H v per workgroup
  double* rowbuf[2][2];   // row j of the stored matrix (parity of j), by global column index
  unsigned int* flags;    // epoch per workgroup
  int* give_up;
  int poll_delay;         // 64-cycle units to sleep before the first poll (polling early only disturbs the publishers)
  int tag_delay;          // tagged exchange: 64-cycle units to sleep before the loads of a column are requested
  int contiguous;         // rows of a wave: first_res + RR g + t (its RR rows adjacent) instead of first_res + g + NW t (strided)
  unsigned int spin_limit;  // bounded spins of a wait: 2^20 (~1.5 s) for the first launch of a process on a cold device, 2^17 (~0.2 s) afterwards
  // Chained launches (round 6, tagged form only): a launch reduces the columns [j_begin, j_end) of the problem it is handed and the
  // next one - a smaller instantiation, re-packed: fewer slots per vector, fuller rows, fewer publishers - continues on the
  // trailing block.  Handed over: the resident rows (written back to the working copy by the last pass), u_{j-1} and the four
  // scalars of the previous column (hand_u / hand_s), p_{j-1} and row j in the exchange buffers (they keep their tags: the origin
  // of a later launch is a multiple of 4 columns further on).  cont != 0: the launch continues a previous one.
  int j_begin, j_end, cont;
  double* hand_u[2];      // u_{j_end - 1} by column index (re / im)
  double* hand_s;         // tau (re, im), scale (re, im) of column j_end - 1
  // Tagged form, round 6: ONE record per slot and column parity holds what column j gathers for slot k - row j's entry and
  // p_{j-1}[k], {row, p} (real, 16 bytes) or {row re, row im, p re, p im} (complex, 32 bytes) - so a consumer asks with one (two)
  // 16-byte load(s) per slot instead of two (four) 8-byte ones: the gather of a column is bound by the number of requests the
  // 256 workgroups put to the L2s, not by their bytes.  Every double still carries its own tag: a torn record is just a stale half.
  double* xch;            // parity q at byte offset q * xch_pb (complex: the {re, im} records of the row first, those of p from xch_pb / 2 on)
  int xch_pb;
};


__device__ __forceinline__ double trd_ld_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void trd_st_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Tagged exchange (TAG instantiations): the lowest mantissa bit of every published double carries the epoch tag of the
// column that consumes it, so a value is its own flag - no drain, no flag store, no poll, no partial sums: a consumer asks
// again for exactly the values whose tag is still the old one.  Buffers alternate with the column parity, so the value a slot
// held before is the one of two columns ago: tag(j) = ((j + 1) >> 1) & 1 differs between the two, and the first use of each
// buffer expects 1 over the zero-filled memory.  The tag bit is cleared on arrival: every workgroup sees the same p and
// row j, rounded down by at most one ulp - below the rounding of the products they come from.
#ifndef TRD_TAG_MASK
#define TRD_TAG_MASK 1u
#endif
__device__ __forceinline__ unsigned int trd_tag_of(int j) { return ((unsigned int)(j + 1) >> 1) & TRD_TAG_MASK; }
__device__ __forceinline__ double trd_tagged(double v, unsigned int tag) {
  return __hiloint2double(__double2hiint(v), (int)(((unsigned int)__double2loint(v) & ~TRD_TAG_MASK) | tag));
}
__device__ __forceinline__ bool trd_tag_is(double v, unsigned int tag) { return ((unsigned int)__double2loint(v) & TRD_TAG_MASK) == tag; }
__device__ __forceinline__ double trd_untagged(double v) {
  return __hiloint2double(__double2hiint(v), (int)((unsigned int)__double2loint(v) & ~TRD_TAG_MASK));
}

// 16-byte agent-scope (sc1) accesses of the exchange records: raw buffer instructions (the only 16-byte form with a cache policy
// operand); `soff` is wave-uniform.  Host pass: declarations only.
typedef unsigned int trd_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void trd_ld16_sc1(const double* base, int bytes, int voff, int soff, double& a, double& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, bytes, 0x00020000);
  const trd_u4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, /*sc1*/ 16);
  a = __hiloint2double((int)q.y, (int)q.x);
  b = __hiloint2double((int)q.w, (int)q.z);
#else
  a = b = 0.0;
#endif
}
// Publication of a complex record {a, b}: one 16-byte agent-scope store.  The whole byte offset goes into the VECTOR offset and the
// scalar offset field stays the constant 0 ON PURPOSE: `buffer_store_dwordx4` reads its data registers late, and a VALU write of
// those registers right behind it (here: the tagging of the next pair into the same temporaries) needs wait states.  hipcc
// (ROCm 7.2) inserts the `s_nop` only for the form WITHOUT a scalar-offset register; with the offset in an SGPR it emitted none,
// and one reduction in ~400 inside four surrogate lanes - never alone on the GPU, where the store leaves the issue queue at once -
// then came out different in the 9th digit of d / e (the low dword of the NEXT value in the record).  Found with
// scripts/det_probe.py / XMCA_TRACE=trdsum (round 6); 8-byte stores: 0 in 2000 reductions, this form: see profiles/r06_trd_determinism.txt.
__device__ __forceinline__ void trd_st16_sc1(double* base, int bytes, int voff, int soff, double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
  trd_u4 q;
  q.x = (unsigned int)__double2loint(a); q.y = (unsigned int)__double2hiint(a);
  q.z = (unsigned int)__double2loint(b); q.w = (unsigned int)__double2hiint(b);
  __builtin_amdgcn_raw_buffer_store_b128(q, rs, voff + soff, 0, /*sc1*/ 16);
#endif
}

constexpr int TRD_RES_THREADS = 256;   // one wave per SIMD: 512 registers per lane for the resident rows
template <bool CPLX, int NC, int RR, bool TAG>
__global__ __launch_bounds__(TRD_RES_THREADS) void trd_resident_kernel(TrdParams P, TrdSync S, int first_res) {
  extern __shared__ __attribute__((aligned(16))) double trd_lds[];
  __shared__ double red2[2][2][TRD_RES_THREADS / 64][2];   // [column parity][which sum]: block sums with one barrier each
  __shared__ double gam_sh[TRD_RES_THREADS / 64][2];
  __shared__ double dj_sh;
  __shared__ double pj_sh[2][2], a0_sh[2][2];                // [column parity]: p_{j-1}[j] and column j's first sub-diagonal entry
  __shared__ double rowpart[2][TRD_RES_THREADS / 64][2];     // streamed rows: the waves' shares of row . v
  __shared__ int give_up_sh;
  constexpr int LV = NC * 128;                     // slots per vector = padded matrix order
  constexpr int NS = LV / TRD_RES_THREADS;         // slots per thread
  constexpr int CG = 4;                            // chunks per group of the pass (one liveness branch per group)
  static_assert(LV % TRD_RES_THREADS == 0 && NC % CG == 0, "NC must be a multiple of 4");
  const int n = P.n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = (int)gridDim.x, NW = nwg * (TRD_RES_THREADS / 64), g = (int)blockIdx.x * (TRD_RES_THREADS / 64) + wave;
  // The workgroup that writes d, e, tau and the reflectors is the owner of the LAST row: it publishes something in every column, so
  // everybody waits for it and nobody can run ahead of it.  (Until round 4 this was workgroup 0, whose rows are the first to
  // die: from then on it only listened - and a listener is not flow-controlled in the tagged exchange, see the column loop.)
  // Row t of wave g: first_res + RG g + RT t.  Strided (RG = 1, RT = NW: every wave holds a row of every stripe - the pass
  // shrinks as the stripes die, the workgroups leave during the last stripe only) or contiguous (RG = RR, RT = 1: the RR rows of
  // a wave are adjacent, a workgroup owns 4 RR consecutive rows and leaves - frees its CU for the kernels of another surrogate
  // lane - when they are dead, from early on; the pass keeps its full length).  The arithmetic of a row does not depend on
  // who owns it: both forms give the same bits.
  const int RG = S.contiguous ? RR : 1, RT = S.contiguous ? 1 : NW;
  int rowi[RR];                                    // the wave's rows (scalars: computed once, not per use inside the column loop)
#pragma unroll
  for (int t = 0; t < RR; ++t) rowi[t] = __builtin_amdgcn_readfirstlane(first_res + g * RG + RT * t);
  const int g_last = S.contiguous ? (n - 1 - first_res) / RR : (n - 1 - first_res) % NW;
  const int writer = g_last / (TRD_RES_THREADS / 64);
  const bool is_writer = (int)blockIdx.x == writer;
  // last row of this workgroup: the largest resident row below n, else its last streamed row, else none
  int wg_last = -1;
  {
    const int b4 = (int)blockIdx.x * (TRD_RES_THREADS / 64);
    for (int t = RR - 1; t >= 0; --t)
      for (int w = TRD_RES_THREADS / 64 - 1; w >= 0; --w) {
        const int i = first_res + (b4 + w) * RG + RT * t;
        if (i < n) wg_last = max(wg_last, i);
      }
    if (wg_last < 0 && first_res - 1 >= (int)blockIdx.x) wg_last = ((first_res - 1 - (int)blockIdx.x) / nwg) * nwg + (int)blockIdx.x;
  }
  double* bV[2] = {trd_lds, trd_lds + 3 * LV};                       // v_{j-1}   (re, im)
  double* bW[2] = {trd_lds + LV, trd_lds + 4 * LV};                  // w_{j-1}
  double* bX[2] = {trd_lds + 2 * LV, trd_lds + 5 * LV};              // column j -> v_j
  // every pointer of the exchange in registers once (indexing the kernel arguments with the parity costs a scalar load
  // and a wait per access)
  double* const pub_r[2] = {S.pub[0][0], S.pub[1][0]};
  double* const pub_i[2] = {S.pub[0][1], S.pub[1][1]};
  double* const gp_r[2] = {S.gpart[0][0], S.gpart[1][0]};
  double* const gp_i[2] = {S.gpart[0][1], S.gpart[1][1]};
  double* const rb_r[2] = {S.rowbuf[0][0], S.rowbuf[1][0]};
  double* const rb_i[2] = {S.rowbuf[0][1], S.rowbuf[1][1]};
  unsigned int* const flags = S.flags;
  // ---- resident rows -> registers (rows beyond the matrix: zeros, they stay zero) ----
  double2 ar[RR][NC], ai[RR][NC];
#pragma unroll
  for (int t = 0; t < RR; ++t) {
    const int i = rowi[t];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      ar[t][c] = make_double2(0.0, 0.0);
      ai[t][c] = make_double2(0.0, 0.0);
      if (i < n) {
        ar[t][c] = *reinterpret_cast<const double2*>(P.Ar + (int64_t)i * P.ld + 128 * c + 2 * lane);
        if (CPLX) ai[t][c] = *reinterpret_cast<const double2*>(P.Ai + (int64_t)i * P.ld + 128 * c + 2 * lane);
      }
    }
  }
  for (int s = tid; s < LV; s += TRD_RES_THREADS) {
    bV[0][s] = 0.0; bW[0][s] = 0.0; bX[0][s] = 0.0;
    if (CPLX) { bV[1][s] = 0.0; bW[1][s] = 0.0; bX[1][s] = 0.0; }
  }
  if (tid == 0) give_up_sh = 0;
  double tpr = 0.0, tpi = 0.0;                     // tau of the previous column
  double spr = 0.0, spi = 0.0;                     // ... and the scale of its reflector: v_{j-1} = u_{j-1} (spr + i spi) beyond its leading 1
  __syncthreads();

  if (TAG && S.cont) {                             // continue a previous launch: u_{j-1} (zero below j) and its scalars
    for (int s = tid; s < LV; s += TRD_RES_THREADS) {
      const bool in = s >= S.j_begin && s < n;
      bV[0][s] = in ? S.hand_u[0][s] : 0.0;
      if (CPLX) bV[1][s] = in ? S.hand_u[1][s] : 0.0;
    }
    tpr = S.hand_s[0]; tpi = CPLX ? S.hand_s[1] : 0.0;
    spr = S.hand_s[2]; spi = CPLX ? S.hand_s[3] : 0.0;
    __syncthreads();
  }
  const bool prof = P.prof && is_writer && tid == 0;
  const bool cont = TAG && S.cont != 0;
  const int j_end = S.j_end;
  // (an earlier launch of the chain ran out of its spins: nothing to do, the host starts again)
  for (int j = (cont && S.give_up[0]) ? j_end : S.j_begin; j < j_end; ++j) {
    const bool has_prev = j > 0 || cont;
    // Tagged exchange: a workgroup whose rows are all dead has nothing left to publish - so nobody waits for it - and nothing
    // to update: it leaves.  Staying on as a listener was a hazard: it is not flow-controlled, and once it fell two columns
    // behind (a second surrogate lane's kernels on its CU are enough) the live workgroups had overwritten the slots it was
    // still waiting for with the tag of the column after next - its spins ran out, the whole reduction was repeated launch by
    // launch (0.2 s; with two lanes at n = 451 complex in nearly every call, at n = 2000 in one of thirty).
    if (TAG && !is_writer && wg_last <= j) break;
    const int prev = (j + 1) & 1, cur = j & 1;
    const int m = n - j - 1;
    if (prof) P.prof[8 * j + 0] = __builtin_amdgcn_s_memtime();
    // ---- prologue: p_{j-1}, its partial sums, row j - every load requested at once, no predicates (the buffers are
    // zero-filled before the launch and hold row 0 in rowbuf[0]: column 0 needs no special case) ----
    const double* const prr = pub_r[prev];
    const double* const pri = pub_i[prev];
    // (tagged exchange: row 0 is put into the records of parity 0 by trd_prefill_kernel, tag bits cleared - column 2 expects 1)
    const double* const rwr = rb_r[cur];
    const double* const rwi = rb_i[cur];
    double gr = 0.0, gi = 0.0;
    if (!TAG && tid < nwg) {
      gr = trd_ld_sc1(gp_r[prev] + tid);
      if (CPLX) gi = trd_ld_sc1(gp_i[prev] + tid);
    }
    // ---- the reflector of the PREVIOUS column gets its scale here (round 5): column j-1 left u_{j-1} = x - beta e_j in what is
    // now bV and ran its pass on that (p = tau s A u is linear in the scale s), so the scaling sits in the wait of the exchange -
    // the values of column j are still on their way - instead of between two barriers in front of the pass.
    // v_{j-1} = (0, ..., 0, 1 [slot j], s u [slots above j]); the writer stores it as reflector row j-1.  The thread's own slots stay
    // in registers (vv): the sums below and the column loop take them from there.  All reads first, then all writes: through
    // pointers the compiler cannot tell apart every LDS store waits for the load before it and every load for the store - twelve
    // round trips of ~130 cycles instead of one (what the formation loops of rounds 3-4 spent most of their time on).
    // No barrier of its own: the one inside the block sum below comes before anybody reads bV.
    // (the largest complex instantiation - NC = 20: the m = 2501 problems of C3 / C4 - has no registers left for vv next to its
    //  streamed rows' loads: there the sums and the column loop read bV again, four slots at a time)
    constexpr bool KEEPV = !(CPLX && NC > 16);
    double vv_r[NS], vv_i[CPLX ? NS : 1];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      const int k = tid + t * TRD_RES_THREADS;
      const bool in = has_prev && k > j && k < n;
      vv_r[t] = in ? bV[0][k] : 0.0;
      if (CPLX) vv_i[t] = in ? bV[1][k] : 0.0;
    }
    if (has_prev) {
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        const int k = tid + t * TRD_RES_THREADS;
        if (CPLX) {
          const double xr = vv_r[t], xi = vv_i[t];
          vv_r[t] = xr * spr - xi * spi;
          vv_i[t] = xr * spi + xi * spr;
        } else {
          vv_r[t] *= spr;
        }
        if (k == j) vv_r[t] = 1.0;
      }
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        const int k = tid + t * TRD_RES_THREADS;
        if ((t + 1) * TRD_RES_THREADS <= j - 1) continue;      // dead stride: zeros already
        if (k >= j && k < n) {
          bV[0][k] = vv_r[t];
          if (CPLX) bV[1][k] = vv_i[t];
          if (is_writer && P.Vr) {               // (tau = 0: H = I - the stored reflector is the zero vector)
            const bool live = tpr != 0.0 || tpi != 0.0;
            P.Vr[(int64_t)(j - 1) * P.ld + k] = live ? vv_r[t] : 0.0;
            if (CPLX) P.Vi[(int64_t)(j - 1) * P.ld + k] = live ? vv_i[t] : 0.0;
          }
        }
      }
    }
    if (TAG && has_prev) {
      // (a request that arrives before the values costs a whole round trip: better to ask a little later)
      for (int q = 0; q < S.tag_delay; ++q) __builtin_amdgcn_s_sleep(1);
    }
    double lr_[NS], li_[NS], lp_[NS], lq_[NS];
    // (tagged: the records of this column's parity; slot k = tid + 256 t sits 16 * 256 t bytes - a scalar offset - behind the thread's first)
    const int xrd = cur * S.xch_pb, xwr = prev * S.xch_pb, xbytes = 2 * S.xch_pb, xhalf = S.xch_pb >> 1;
    auto gather_slot = [&](int t) {
      if constexpr (TAG) {
        const int soff = xrd + t * (16 * TRD_RES_THREADS);
        if constexpr (CPLX) {                      // (complex: the {re, im} records of the row, then - half a parity further on - those of p)
          trd_ld16_sc1(S.xch, xbytes, tid * 16, soff, lr_[t], li_[t]);
          trd_ld16_sc1(S.xch, xbytes, tid * 16, soff + xhalf, lp_[t], lq_[t]);
        } else {
          trd_ld16_sc1(S.xch, xbytes, tid * 16, soff, lr_[t], lp_[t]);
          li_[t] = 0.0;
          lq_[t] = 0.0;
        }
      } else {
        const int k = tid + t * TRD_RES_THREADS;
        lr_[t] = trd_ld_sc1(rwr + k);
        lp_[t] = trd_ld_sc1(prr + k);
        li_[t] = 0.0;
        lq_[t] = 0.0;
        if (CPLX) {
          li_[t] = trd_ld_sc1(rwi + k);
          lq_[t] = trd_ld_sc1(pri + k);
        }
      }
    };
#pragma unroll
    for (int t = 0; t < NS; ++t) gather_slot(t);
    if constexpr (TAG) {
      // wait for the values of column j (tag), each thread for its own slots; p^H v from every workgroup's own copy of v_{j-1}
      const unsigned int tg = trd_tag_of(j);
      if (has_prev) {
        unsigned int spins = 0;
        for (;;) {
          bool all = true;
#pragma unroll
          for (int t = 0; t < NS; ++t) {
            const int k = tid + t * TRD_RES_THREADS;
            if (k >= j && k < n) {
              bool ok = trd_tag_is(lr_[t], tg) && trd_tag_is(lp_[t], tg);
              if (CPLX) ok = ok && trd_tag_is(li_[t], tg) && trd_tag_is(lq_[t], tg);
              if (!ok) {
                all = false;
                gather_slot(t);
              }
            }
          }
          if (__all(all)) break;
          if (++spins > S.spin_limit) {
            if (tid == 0 && atomicCAS(S.give_up + 1, 0, 1) == 0) { S.give_up[2] = j; S.give_up[3] = (int)blockIdx.x; }   // (XMCA_TRACE=giveup)
            give_up_sh = 1; break;
          }       // a workgroup is missing: report, never hang.  All persistent launches of THIS process pass one gate (common.h), so only a foreign process can cause it; the bound is ~1.5 s for the first launch of a process (0.2 s was less than a first launch on a cold device can take) and ~0.2 s afterwards (TrdSync::spin_limit; advisor, round 4)
          __builtin_amdgcn_s_sleep(1);
        }
      }
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        const int k = tid + t * TRD_RES_THREADS;
        lr_[t] = trd_untagged(lr_[t]);
        lp_[t] = trd_untagged(lp_[t]);
        if (CPLX) {
          li_[t] = trd_untagged(li_[t]);
          lq_[t] = trd_untagged(lq_[t]);
        }
        if (k >= j && k < n) {
          const double vr = KEEPV ? vv_r[t] : bV[0][k], vi = CPLX ? (KEEPV ? vv_i[t] : bV[1][k]) : 0.0;
          gr += lp_[t] * vr + lq_[t] * vi;              // conj(p) v
          gi += lp_[t] * vi - lq_[t] * vr;
        }
      }
    }
    // p_{j-1}[j] is needed by every thread in front of the fused loop below (w_{j-1}[j] enters column j): its owner parks it
    // next to the partial sums - one barrier publishes both
#pragma unroll
    for (int t = 0; t < NS; ++t)
      if (tid + t * TRD_RES_THREADS == j) {
        pj_sh[cur][0] = lp_[t];
        pj_sh[cur][1] = CPLX ? lq_[t] : 0.0;
      }
    trd_block_sum2_1b<TRD_RES_THREADS / 64>(gr, gi, red2[cur][0]);
    if (TAG && give_up_sh) {
      if (tid == 0) atomicExch(S.give_up, 1);
      break;
    }
    const double ar_ = -0.5 * (tpr * gr - tpi * gi);
    const double ai_ = -0.5 * (tpr * gi + tpi * gr);
    // w_{j-1}[j] = p_{j-1}[j] + alpha v_{j-1}[j]  (v_{j-1}[j] = 1 from column 1 on; 0 in column 0, where p = 0 too)
    const double vjr = has_prev ? 1.0 : 0.0, vji = 0.0;
    const double wjr = pj_sh[cur][0] + ar_ * vjr - ai_ * vji;
    const double wji = CPLX ? pj_sh[cur][1] + ar_ * vji + ai_ * vjr : 0.0;
    if (prof) P.prof[8 * j + 1] = __builtin_amdgcn_s_memtime();
    // ONE loop (round 5; two loops and a barrier before): w_{j-1} = p_{j-1} + alpha v_{j-1} and column j of the current matrix,
    // x = conj(row j) - v_{j-1} conj(w_{j-1}[j]) - w_{j-1} conj(v_{j-1}[j]).  Slots outside [j, n) are written as zeros: dead columns
    // stay zero in all three vectors, so the pass may touch them.  Slot j + 1 keeps x[j+1] (the reflector's leading entry
    // x[j+1] - beta is patched in by the pass: beta needs the norm that is being summed here).
    // Straight-line code: selects instead of branches (a branch per slot splits the loop into basic blocks, and the dependent
    // chains of the slots then run one after the other instead of side by side); dead strides are skipped four at a time.
    double xn2 = 0.0, djv = 0.0, a0v_r = 0.0, a0v_i = 0.0;
#pragma unroll
    for (int t0 = 0; t0 < NS; t0 += 4) {
      if ((t0 + 4 < NS ? t0 + 4 : NS) * TRD_RES_THREADS + 1 <= j) continue;   // (uniform) every slot of these strides below j - 1: zeros already
      double gv_r[4], gv_i[4];
      if constexpr (!KEEPV) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = tid + (t0 + u < NS ? t0 + u : NS - 1) * TRD_RES_THREADS;
          gv_r[u] = bV[0][k];
          gv_i[u] = CPLX ? bV[1][k] : 0.0;
        }
      }
#pragma unroll
      for (int t = t0; t < (t0 + 4 < NS ? t0 + 4 : NS); ++t) {
        const int k = tid + t * TRD_RES_THREADS;
        const bool in = k >= j && k < n;
        const double vr = KEEPV ? vv_r[t] : gv_r[t - t0];
        double wr, wi = 0.0, xr, xi = 0.0;
        if (CPLX) {
          const double vi = KEEPV ? vv_i[t] : gv_i[t - t0];
          wr = lp_[t] + ar_ * vr - ai_ * vi;
          wi = lq_[t] + ar_ * vi + ai_ * vr;
          xr = lr_[t] - (vr * wjr + vi * wji) - (wr * vjr + wi * vji);
          xi = -li_[t] - (vi * wjr - vr * wji) - (wi * vjr - wr * vji);
        } else {
          wr = lp_[t] + ar_ * vr;
          xr = lr_[t] - vr * wjr - wr * vjr;
        }
        wr = in ? wr : 0.0;
        wi = in ? wi : 0.0;
        djv = k == j ? xr : djv;
        a0v_r = k == j + 1 ? xr : a0v_r;
        a0v_i = k == j + 1 ? xi : a0v_i;
        const bool keep = in && k != j;
        xr = keep ? xr : 0.0;
        xi = keep ? xi : 0.0;
        const bool tail = in && k > j + 1;
        xn2 += tail ? xr * xr + xi * xi : 0.0;
        bW[0][k] = wr;
        bX[0][k] = xr;
        if (CPLX) {
          bW[1][k] = wi;
          bX[1][k] = xi;
        }
      }
    }
    if (tid == (j & (TRD_RES_THREADS - 1))) dj_sh = djv;
    if (tid == ((j + 1) & (TRD_RES_THREADS - 1))) {
      a0_sh[cur][0] = a0v_r;
      a0_sh[cur][1] = a0v_i;
    }
    {
      double dummy = 0.0;
      trd_block_sum2_1b<TRD_RES_THREADS / 64>(xn2, dummy, red2[cur][1]);
    }
    if (prof) P.prof[8 * j + 2] = __builtin_amdgcn_s_memtime();
    if (m == 0) {
      if (is_writer && tid == 0) P.d[j] = dj_sh;
      break;
    }
    const double a0r = a0_sh[cur][0], a0i = CPLX ? a0_sh[cur][1] : 0.0;
    double beta, tr, ti = 0.0, scr = 0.0, sci = 0.0;
    if (xn2 == 0.0 && a0i == 0.0) {
      beta = a0r;
      tr = 0.0;
    } else {
      beta = -copysign(trd_sqrt(a0r * a0r + a0i * a0i + xn2), a0r);
      const double rb = trd_rcp(beta);
      tr = fma(fma(-(beta - a0r) * rb, beta, beta - a0r), rb, (beta - a0r) * rb);
      ti = -a0i * rb;
      const double dr = a0r - beta, di = a0i, dn = trd_rcp(dr * dr + di * di);
      scr = dr * dn;
      sci = -di * dn;
    }
    // u_j = x - beta e_{j+1};  v_j = s u_j with s = 1 / u_j[j+1] = (scr, sci), so the pass multiplies by u_j and scales the sums
    // by tau s.  (tau = 0: s = 0, every p is zero whatever u holds.)
    const double mr_ = trd_uniform(tr * scr - ti * sci), mi_ = trd_uniform(tr * sci + ti * scr);
    // (what the next column needs of this one's scalars, in scalar registers from here on: tau and the scale are dead in the pass)
    const double tr_next = trd_uniform(tr), ti_next = trd_uniform(ti), sr_next = trd_uniform(scr), si_next = trd_uniform(sci);
    if (is_writer && tid == 0) {
      P.d[j] = dj_sh;
      P.e[j] = beta;
      P.tau[0][j] = tr;
      if (CPLX) P.tau[1][j] = ti;
    }
    if (prof) P.prof[8 * j + 3] = __builtin_amdgcn_s_memtime();
    // ---- pass: resident rows in registers, early rows streamed from global memory ----
    // No liveness tests per row: a dead row (i <= j) keeps being updated - nobody reads its p_i (consumers start at j+1),
    // and v_j[i] = 0 keeps it out of p^H v; rows beyond the matrix are zero and stay zero (their scalars are zero).  Dead
    // columns are zero in all three vectors; whole groups of CG dead chunks are skipped with one scalar branch.
    const int g0 = ((j + 1) >> 7) / CG;             // groups below hold dead columns only
    const int c0 = (j + 1) >> 7;
    double gwr = 0.0, gwi = 0.0;
    // bX holds x with x[j+1] as stored; the pass multiplies by u_j = x - beta e_{j+1}.  EVERY wave writes the leading entry itself:
    // the LDS operations of a wave complete in order, so its own reads below see it without a barrier (four stores of one value).
    if (lane == 0) bX[0][j + 1] = a0r - beta;
    // v_j[i] = s u_j[i] (1 at i = j + 1) for the partial sums p^H v of the flags form
    auto v_of = [&](int i, double& vr, double& vi) {
      if (i == j + 1) { vr = 1.0; vi = 0.0; return; }
      const double xr = bX[0][i], xi = CPLX ? bX[1][i] : 0.0;
      vr = xr * sr_next - xi * si_next;
      vi = xr * si_next + xi * sr_next;
    };
    double* const pbr = pub_r[cur];
    double* const pbi = pub_i[cur];
    double* const nrr = rb_r[prev];
    double* const nri = rb_i[prev];
    const unsigned int tgn = trd_tag_of(j + 1);      // what is published now is consumed by column j+1
    double* const xnext = TAG ? reinterpret_cast<double*>(reinterpret_cast<char*>(S.xch) + xwr) : nullptr;   // the records column j+1 reads
    auto publish_p = [&](int i, double pr, double pi) {
      if constexpr (!TAG) {
        trd_st_sc1(pbr + i, pr);
        if (CPLX) trd_st_sc1(pbi + i, pi);
      } else if constexpr (CPLX) {
        trd_st16_sc1(S.xch, xbytes, 16 * i, xwr + xhalf, trd_tagged(pr, tgn), trd_tagged(pi, tgn));
      } else {
        trd_st_sc1(xnext + 2 * i + 1, trd_tagged(pr, tgn));
      }
    };
    // early rows (i < first_res: complex problems of more than 2048 rows), owner WORKGROUP i mod nwg: the same from global
    // memory, the live chunks of the row dealt to the four waves (one wave alone needs five dependent trips to memory for
    // a row of 20 chunks: 9 us, which every other workgroup then waits for).  First, so that the stores are on their way
    // while the resident rows are processed.  The partial sums of the waves are added in wave order.
    {
      constexpr int MC = (NC + 3) / 4;               // chunks per wave and row, at most
      int i = (int)blockIdx.x;
      if (i < j + 1) i += ((j + 1 - i + nwg - 1) / nwg) * nwg;
      int par = 0;
      for (; i < first_res; i += nwg, par ^= 1) {
        const double svr_ = bV[0][i], swr_ = bW[0][i];
        const double svi_ = CPLX ? bV[1][i] : 0.0, swi_ = CPLX ? bW[1][i] : 0.0;
        double* rowr = P.Ar + (int64_t)i * P.ld;
        double* rowi = CPLX ? P.Ai + (int64_t)i * P.ld : nullptr;
        double sr = 0.0, si = 0.0;
        const bool pubrow = i == j + 1;
        // (at most four chunks of a wave in flight - 32 registers next to the resident rows; a fifth takes a second trip)
#pragma unroll 1
        for (int u0 = 0; u0 < MC; u0 += 4) {
        double2 la[4], lb[4];
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
          const int u = u0 + uu;
          const int c = u < MC ? c0 + wave + 4 * u : NC;
          la[uu] = make_double2(0.0, 0.0);
          lb[uu] = make_double2(0.0, 0.0);
          if (c < NC) {                              // (wave-uniform)
            la[uu] = *reinterpret_cast<const double2*>(rowr + 128 * c + 2 * lane);
            if (CPLX) lb[uu] = *reinterpret_cast<const double2*>(rowi + 128 * c + 2 * lane);
          }
        }
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
          const int u = u0 + uu;
          const int c = u < MC ? c0 + wave + 4 * u : NC;
          if (c < NC) {
            const int k = 128 * c + 2 * lane;
            double2 a = la[uu];
            const double2 wk = *reinterpret_cast<const double2*>(bW[0] + k);
            const double2 vk = *reinterpret_cast<const double2*>(bV[0] + k);
            const double2 xk = *reinterpret_cast<const double2*>(bX[0] + k);
            if (CPLX) {
              double2 b = lb[uu];
              const double2 wki = *reinterpret_cast<const double2*>(bW[1] + k);
              const double2 vki = *reinterpret_cast<const double2*>(bV[1] + k);
              const double2 xki = *reinterpret_cast<const double2*>(bX[1] + k);
              a.x -= (svr_ * wk.x + svi_ * wki.x) + (swr_ * vk.x + swi_ * vki.x);
              b.x -= (svi_ * wk.x - svr_ * wki.x) + (swi_ * vk.x - swr_ * vki.x);
              a.y -= (svr_ * wk.y + svi_ * wki.y) + (swr_ * vk.y + swi_ * vki.y);
              b.y -= (svi_ * wk.y - svr_ * wki.y) + (swi_ * vk.y - swr_ * vki.y);
              *reinterpret_cast<double2*>(rowr + k) = a;
              *reinterpret_cast<double2*>(rowi + k) = b;
              sr += a.x * xk.x - b.x * xki.x;
              si += a.x * xki.x + b.x * xk.x;
              sr += a.y * xk.y - b.y * xki.y;
              si += a.y * xki.y + b.y * xk.y;
              if (pubrow) {
                if constexpr (TAG) {
                  trd_st16_sc1(S.xch, xbytes, 16 * k, xwr, trd_tagged(a.x, tgn), trd_tagged(b.x, tgn));
                  trd_st16_sc1(S.xch, xbytes, 16 * k + 16, xwr, trd_tagged(a.y, tgn), trd_tagged(b.y, tgn));
                } else {
                  trd_st_sc1(nri + k, b.x);
                  trd_st_sc1(nri + k + 1, b.y);
                }
              }
            } else {
              a.x -= svr_ * wk.x + swr_ * vk.x;
              a.y -= svr_ * wk.y + swr_ * vk.y;
              *reinterpret_cast<double2*>(rowr + k) = a;
              sr += a.x * xk.x;
              sr += a.y * xk.y;
            }
            if (pubrow) {
              if constexpr (!TAG) {
                trd_st_sc1(nrr + k, a.x);
                trd_st_sc1(nrr + k + 1, a.y);
              } else if constexpr (!CPLX) {
                trd_st_sc1(xnext + 2 * k, trd_tagged(a.x, tgn));
                trd_st_sc1(xnext + 2 * k + 2, trd_tagged(a.y, tgn));
              }
            }
          }
        }
        }
        sr = trd_wave_sum_dpp(sr);
        if (CPLX) si = trd_wave_sum_dpp(si);
        if (lane == 0) {
          rowpart[par][wave][0] = sr;
          rowpart[par][wave][1] = si;
        }
        __syncthreads();
        if (wave == 0) {
          const double yr = ((rowpart[par][0][0] + rowpart[par][1][0]) + rowpart[par][2][0]) + rowpart[par][3][0];
          const double yi = CPLX ? ((rowpart[par][0][1] + rowpart[par][1][1]) + rowpart[par][2][1]) + rowpart[par][3][1] : 0.0;
          const double pr = mr_ * yr - mi_ * yi, pi = mr_ * yi + mi_ * yr;
          if (lane == 0) publish_p(i, pr, pi);
          if constexpr (!TAG) {
            double vr, vi;
            v_of(i, vr, vi);
            gwr += pr * vr + pi * vi;
            gwi += pr * vi - pi * vr;
          }
        }
      }
    }
    if (rowi[RR - 1] > j) {        // (uniform) the wave still owns a live resident row
      double vpr[RR], vpi[RR], wpr[RR], wpi[RR], accr[RR], acci[RR];
#pragma unroll
      for (int t = 0; t < RR; ++t) {
        const int i = rowi[t];
        const int ii = i < n ? i : 0;                // (bV[0] = bW[0] = 0 from column 1 on, and the registers of such a row are 0)
        vpr[t] = i < n ? bV[0][ii] : 0.0; wpr[t] = i < n ? bW[0][ii] : 0.0;
        vpi[t] = (CPLX && i < n) ? bV[1][ii] : 0.0; wpi[t] = (CPLX && i < n) ? bW[1][ii] : 0.0;
        accr[t] = 0.0; acci[t] = 0.0;
      }
#pragma unroll
      for (int cg = 0; cg < NC / CG; ++cg) {
        if (cg >= g0) {
#pragma unroll
          for (int cc = 0; cc < CG; ++cc) {
            const int c = cg * CG + cc;
            const int k = 128 * c + 2 * lane;
            const double2 wk = *reinterpret_cast<const double2*>(bW[0] + k);
            const double2 vk = *reinterpret_cast<const double2*>(bV[0] + k);
            const double2 xk = *reinterpret_cast<const double2*>(bX[0] + k);
            double2 wki = make_double2(0.0, 0.0), vki = wki, xki = wki;
            if (CPLX) {
              wki = *reinterpret_cast<const double2*>(bW[1] + k);
              vki = *reinterpret_cast<const double2*>(bV[1] + k);
              xki = *reinterpret_cast<const double2*>(bX[1] + k);
            }
#pragma unroll
            for (int t = 0; t < RR; ++t) {
              double2 a = ar[t][c];
              if (CPLX) {
                double2 b = ai[t][c];
                // a -= v'_i conj(w'_k) + w'_i conj(v'_k)
                a.x = fma(-vpr[t], wk.x, a.x);  a.x = fma(-vpi[t], wki.x, a.x);  a.x = fma(-wpr[t], vk.x, a.x);  a.x = fma(-wpi[t], vki.x, a.x);
                b.x = fma(-vpi[t], wk.x, b.x);  b.x = fma(vpr[t], wki.x, b.x);   b.x = fma(-wpi[t], vk.x, b.x);  b.x = fma(wpr[t], vki.x, b.x);
                a.y = fma(-vpr[t], wk.y, a.y);  a.y = fma(-vpi[t], wki.y, a.y);  a.y = fma(-wpr[t], vk.y, a.y);  a.y = fma(-wpi[t], vki.y, a.y);
                b.y = fma(-vpi[t], wk.y, b.y);  b.y = fma(vpr[t], wki.y, b.y);   b.y = fma(-wpi[t], vk.y, b.y);  b.y = fma(wpr[t], vki.y, b.y);
                ai[t][c] = b;
                accr[t] = fma(a.x, xk.x, accr[t]);  accr[t] = fma(-b.x, xki.x, accr[t]);
                acci[t] = fma(a.x, xki.x, acci[t]); acci[t] = fma(b.x, xk.x, acci[t]);
                accr[t] = fma(a.y, xk.y, accr[t]);  accr[t] = fma(-b.y, xki.y, accr[t]);
                acci[t] = fma(a.y, xki.y, acci[t]); acci[t] = fma(b.y, xk.y, acci[t]);
              } else {
                a.x = fma(-vpr[t], wk.x, a.x);  a.x = fma(-wpr[t], vk.x, a.x);
                a.y = fma(-vpr[t], wk.y, a.y);  a.y = fma(-wpr[t], vk.y, a.y);
                accr[t] = fma(a.x, xk.x, accr[t]);
                accr[t] = fma(a.y, xk.y, accr[t]);
              }
              ar[t][c] = a;
            }
          }
          // (keeps the LDS operands of later groups out of the registers of the resident rows)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the next column's row first (it is on every workgroup's critical path): published as stored, update j is applied by
      // the consumers
#pragma unroll
      for (int t = 0; t < RR; ++t) {
        if (rowi[t] == j + 1) {             // wave-uniform
          // (the lane offset goes through an empty asm: the NC store offsets 128 c + 2 lane are loop invariants, and hipcc
          // hoists them out of the COLUMN loop into registers of their own - 18 to 32 of them next to the resident rows,
          // which is what pushed the tagged complex NC = 20 build into scratch in round 3)
          int lane2 = 2 * lane;
          asm volatile("" : "+v"(lane2));
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            if (c >= c0) {
              const int k = 128 * c + lane2;
              if constexpr (!TAG) {
                trd_st_sc1(nrr + k, ar[t][c].x);
                trd_st_sc1(nrr + k + 1, ar[t][c].y);
                if (CPLX) {
                  trd_st_sc1(nri + k, ai[t][c].x);
                  trd_st_sc1(nri + k + 1, ai[t][c].y);
                }
              } else if constexpr (CPLX) {
                trd_st16_sc1(S.xch, xbytes, 16 * k, xwr, trd_tagged(ar[t][c].x, tgn), trd_tagged(ai[t][c].x, tgn));
                trd_st16_sc1(S.xch, xbytes, 16 * k + 16, xwr, trd_tagged(ar[t][c].y, tgn), trd_tagged(ai[t][c].y, tgn));
              } else {
                trd_st_sc1(xnext + 2 * k, trd_tagged(ar[t][c].x, tgn));
                trd_st_sc1(xnext + 2 * k + 2, trd_tagged(ar[t][c].y, tgn));
              }
            }
          }
        }
      }
      // In the LAST column of a launch that hands over to another one (chain, TrdSync) the live rows go back to the working
      // copy: the next launch loads its resident rows from there.
      if (TAG && j + 1 == j_end && j_end < n) {
#pragma unroll
        for (int t = 0; t < RR; ++t) {
          if (rowi[t] > j && rowi[t] < n) {             // wave-uniform
            int lane2 = 2 * lane;
            asm volatile("" : "+v"(lane2));
            double* const dr_ = P.Ar + (int64_t)rowi[t] * P.ld;
            double* const di_ = CPLX ? P.Ai + (int64_t)rowi[t] * P.ld : nullptr;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              if (c >= c0) {
                const int k = 128 * c + lane2;
                *reinterpret_cast<double2*>(dr_ + k) = ar[t][c];
                if (CPLX) *reinterpret_cast<double2*>(di_ + k) = ai[t][c];
              }
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < RR; ++t) {
        const int i = rowi[t];
        if (i < n && i > j) {                              // wave-uniform
          const double yr = trd_wave_sum_dpp(accr[t]);
          const double yi = CPLX ? trd_wave_sum_dpp(acci[t]) : 0.0;
          const double pr = mr_ * yr - mi_ * yi, pi = mr_ * yi + mi_ * yr;
          if (lane == 0) publish_p(i, pr, pi);
          if constexpr (!TAG) {
            double vr, vi;
            v_of(i, vr, vi);
            gwr += pr * vr + pi * vi;
            gwi += pr * vi - pi * vr;
          }
        }
      }
    }
    if (prof) P.prof[8 * j + 4] = __builtin_amdgcn_s_memtime();
    if constexpr (!TAG) {
      if (lane == 0) {
        gam_sh[wave][0] = gwr;
        gam_sh[wave][1] = gwi;
      }
      __syncthreads();
      if (tid == 0) {
        double sr = 0.0, si = 0.0;
  #pragma unroll
        for (int w = 0; w < TRD_RES_THREADS / 64; ++w) { sr += gam_sh[w][0]; si += gam_sh[w][1]; }
        trd_st_sc1(gp_r[cur] + blockIdx.x, sr);
        if (CPLX) trd_st_sc1(gp_i[cur] + blockIdx.x, si);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains before the flag goes out
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + blockIdx.x, (unsigned int)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prof) P.prof[8 * j + 5] = __builtin_amdgcn_s_memtime();
      // ---- exchange: wait until every workgroup has published column j ----
      if (wave == 0) {
        // the first poll waits ~1000 cycles: polling while the other workgroups still publish only slows them down (measured:
        // n = 2920 26.9 -> 25.2 ms; counting arrivals in 8 sharded counters instead of 256 flags: 32 ms)
        if (S.poll_delay >= 32) __builtin_amdgcn_s_sleep(32);
        else if (S.poll_delay >= 16) __builtin_amdgcn_s_sleep(16);
        else if (S.poll_delay >= 8) __builtin_amdgcn_s_sleep(8);
        else if (S.poll_delay >= 4) __builtin_amdgcn_s_sleep(4);
        unsigned int spins = 0;
        for (;;) {
          bool ok = true;
          for (int w = lane; w < nwg; w += 64)
            ok &= __hip_atomic_load(flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned int)(j + 1);
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > S.spin_limit) { if (lane == 0) give_up_sh = 1; break; }     // a workgroup is missing: report, never hang (bound: TrdSync::spin_limit)
        }
      }
    }
    __syncthreads();
    if (prof) P.prof[8 * j + 6] = __builtin_amdgcn_s_memtime();
    if (give_up_sh) {
      if (tid == 0) atomicExch(S.give_up, 1);
      break;
    }
    // ---- hand-over to the next launch of the chain (last column of this one, columns remain): u_j and its scalars; the resident
    // rows went back to the working copy in the pass ----
    if (TAG && is_writer && j + 1 == j_end && j_end < n) {
      int s0 = tid;
      asm volatile("" : "+v"(s0));                 // (the store addresses are loop invariants: not to be hoisted into registers of the column loop)
      for (int s = s0; s < LV; s += TRD_RES_THREADS) {
        S.hand_u[0][s] = bX[0][s];
        if (CPLX) S.hand_u[1][s] = bX[1][s];
      }
      if (tid == 0) {
        S.hand_s[0] = tr_next; S.hand_s[1] = ti_next; S.hand_s[2] = sr_next; S.hand_s[3] = si_next;
      }
    }
    // v_j becomes v_{j-1}
    { double* t0 = bV[0]; bV[0] = bX[0]; bX[0] = t0; }
    if (CPLX) { double* t1 = bV[1]; bV[1] = bX[1]; bX[1] = t1; }
    tpr = tr_next;
    tpi = ti_next;
    spr = sr_next;
    spi = si_next;
  }
}

// Tagged exchange: the records of parity 0 start as {row 0 of the working copy, p = 0}, every tag bit cleared (their first rewrite,
// for column 2, carries tag 1; parity 1 starts as zeros and expects 1 in column 1).
__global__ void trd_prefill_kernel(const double* __restrict__ Wr, const double* __restrict__ Wi, int n, double* __restrict__ xch) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  if (Wi) {
    xch[2 * k] = trd_untagged(Wr[k]);
    xch[2 * k + 1] = trd_untagged(Wi[k]);
  } else {
    xch[2 * k] = trd_untagged(Wr[k]);
  }
}

// working copy: W = f * A, f a power of two (exact) from max |a_ij| over the WHOLE matrix (max(|re|, |im|) for complex
// entries): the entries of the scaled matrix are then below 1 in magnitude and its norm below n whatever the input - the
// growth bound behind the eight-row rescaling of trd_bisect_kernel and the pivot floors of trd_twisted_kernel hold for
// indefinite matrices with a small diagonal too (advisor, round 3: the diagonal alone was used; for the positive
// semi-definite Gram matrices of the path |a_ij| <= max a_ii, so their scale factor - and every bit downstream - is unchanged).
// acc[0]: the maximum as a bit pattern (non-negative doubles order like their patterns), acc[1] != 0: a non-finite entry.
__global__ __launch_bounds__(256) void trd_maxabs_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai, int n, int64_t lda,
                                                         unsigned long long* __restrict__ acc) {
  double m = 0.0;
  bool bad = false;
  for (int r = blockIdx.x; r < n; r += gridDim.x)
    for (int c = threadIdx.x; c < n; c += 256) {
      const double a = fabs(Ar[(int64_t)r * lda + c]), b = Ai ? fabs(Ai[(int64_t)r * lda + c]) : 0.0;
      if (!(a <= 1.7e308) || !(b <= 1.7e308)) bad = true;
      m = fmax(m, fmax(a, b));
    }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if (__any(bad)) { if ((threadIdx.x & 63) == 0) atomicOr(acc + 1, 1ull); }
  else if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(acc, (unsigned long long)__double_as_longlong(m));
}
__global__ void trd_scale_kernel(const unsigned long long* __restrict__ acc, double* __restrict__ scal) {
  const double g = __longlong_as_double((long long)acc[0]);
  const bool nan = acc[1] != 0ull;
  double f = 1.0;
  if (g > 0.0 && !nan) {
    int ex;
    frexp(g, &ex);
    f = ldexp(1.0, -ex);               // f * g in [0.5, 1)
  }
  scal[0] = f;
  scal[1] = nan ? 1.0 : 0.0;
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
// 16 lanes per eigenvalue: count(x) = number of eigenvalues below x at 16 interior points of the bracket per pass (sign
// changes of the leading principal minors by their three-term recurrence: one dependent FMA per row, no division, no
// compare in the chain).  14 passes shrink the Gershgorin interval by 17^14 = 1.7e17.
// lam_desc[n-1-k] = eigenvalue k (ascending) / scale factor.  flag[0] != 0: non-finite input.
constexpr int TRD_BIS_THREADS = 256;
__global__ __launch_bounds__(TRD_BIS_THREADS) void trd_bisect_kernel(const double* __restrict__ d, const double* __restrict__ e, int n,
                                                                   const double* __restrict__ scal, double* lam_desc, double* lam_asc_scaled, int* flag) {
  extern __shared__ __attribute__((aligned(16))) double bis_lds[];
  __shared__ double red[2][TRD_BIS_THREADS / 64];
  double* sd = bis_lds;
  double* se2 = bis_lds + n;
  const int tid = threadIdx.x;
  double gl = 1.7e308, gu = -1.7e308;
  bool bad = false;
  for (int i = tid; i < n; i += TRD_BIS_THREADS) {
    const double di = d[i];
    const double el = i > 0 ? e[i - 1] : 0.0, er = i < n - 1 ? e[i] : 0.0;
    if (!(fabs(di) <= 1.7e308) || !(fabs(er) <= 1.7e308)) bad = true;
    sd[i] = di;
    // se2[i] couples i and i+1.  Floor 2^-200 (the matrix is scaled to norm <= 1: a perturbation of 8e-31 of an entry) so
    // that an exactly split matrix (e = 0) cannot leave two consecutive zero minors behind - see the count below
    se2[i] = fmax(er * er, 0x1p-200);
    const double r = fabs(el) + fabs(er);
    gl = fmin(gl, di - r);
    gu = fmax(gu, di + r);
  }
  for (int o = 32; o > 0; o >>= 1) {
    gl = fmin(gl, __shfl_xor(gl, o));
    gu = fmax(gu, __shfl_xor(gu, o));
  }
  if (__any(bad) && flag) { if ((tid & 63) == 0) atomicOr(flag, 1); }
  if ((tid & 63) == 0) { red[0][tid >> 6] = gl; red[1][tid >> 6] = gu; }
  __syncthreads();
  for (int w = 0; w < TRD_BIS_THREADS / 64; ++w) { gl = fmin(gl, red[0][w]); gu = fmax(gu, red[1][w]); }
  const double bnorm = fmax(fabs(gl), fabs(gu));
  gl -= 2.2e-16 * bnorm * n + 1e-300;
  gu += 2.2e-16 * bnorm * n + 1e-300;
  const int grp = tid >> 4, l = tid & 15;
  const int k = blockIdx.x * (TRD_BIS_THREADS / 16) + grp;       // eigenvalue index, ascending
  const bool live = k < n;
  double lo = gl, hi = gu;
  const int gshift = ((tid & 63) >> 4) * 16;
  for (int pass = 0; pass < 14; ++pass) {
    const double x = lo + (hi - lo) * ((double)(l + 1) * (1.0 / 17.0));
    // Sturm count by the three-term recurrence of the leading principal minors, p_{i+1} = (d_i - x) p_i - e_{i-1}^2 p_{i-1}:
    // count = sign changes along p_0 = 1, p_1, ..., p_n.  One dependent FMA per row instead of a division; the pair
    // (p_i, p_{i-1}) is rescaled by a power of two every eight rows (growth per row is bounded by ~5 for the scaled matrix, so
    // neither overflows within eight rows, and the larger of the two is kept near 1).
    // An exact zero needs no fix-up in the chain (round 4: the compare + selects were a third of the row's instructions and
    // sat on its dependent path): e^2 has a floor (below), so a zero p_i is followed by p_{i+1} = -e_i^2 p_{i-1} != 0 of the
    // sign opposite to p_{i-1} - exactly one sign change over the two steps whichever sign the zero is given, the same count
    // as dstebz's "a zero takes the sign opposite to its predecessor".  Signs are compared on the high words.
    double pm = 1.0, pc = sd[0] - x;
    unsigned int sc = (unsigned int)__double2hiint(pc);
    int cnt = (int)(sc >> 31);
    int i = 1;
    // (the operands of the NEXT eight rows are requested before the current eight are used: the LDS latency stays off the chain)
    double dv[8], ev[8];
    if (i + 8 <= n) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { dv[u] = sd[i + u]; ev[u] = se2[i + u - 1]; }
    }
    for (; i + 8 <= n; i += 8) {
      double dn[8], en[8];
      const int inext = i + 16 <= n ? i + 8 : i;           // (the last block reads its own rows again)
#pragma unroll
      for (int u = 0; u < 8; ++u) { dn[u] = sd[inext + u]; en[u] = se2[inext + u - 1]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double pn = fma(dv[u] - x, pc, -(ev[u] * pm));
        const unsigned int sn = (unsigned int)__double2hiint(pn);
        cnt += (int)((sn ^ sc) >> 31);
        sc = sn;
        pm = pc;
        pc = pn;
      }
      const double mx = fmax(fabs(pc), fabs(pm));
      if (mx == 0.0) pc = (sc >> 31) ? -1.0 : 1.0;            // (both underflowed: start again with the sign kept - not reachable with the floor on e^2)
      const int ex = __builtin_amdgcn_frexp_exp(mx);
      pc = ldexp(pc, -ex);
      pm = ldexp(pm, -ex);
#pragma unroll
      for (int u = 0; u < 8; ++u) { dv[u] = dn[u]; ev[u] = en[u]; }
    }
    for (; i < n; ++i) {
      const double pn = fma(sd[i] - x, pc, -(se2[i - 1] * pm));
      const unsigned int sn = (unsigned int)__double2hiint(pn);
      cnt += (int)((sn ^ sc) >> 31);
      sc = sn;
      pm = pc;
      pc = pn;
    }
    const unsigned long long mask = __ballot(cnt <= k);
    const int s = __popcll((mask >> gshift) & 0xFFFFull);       // points at or below eigenvalue k (counts are monotone in x)
    const double x_lo = __shfl(x, s > 0 ? s - 1 : 0, 16);
    const double x_hi = __shfl(x, s < 16 ? s : 15, 16);
    if (s > 0) lo = x_lo;
    if (s < 16) hi = x_hi;
  }
  if (live && l == 0) {
    lam_desc[n - 1 - k] = 0.5 * (lo + hi) / scal[0];
    if (lam_asc_scaled) lam_asc_scaled[k] = 0.5 * (lo + hi);
  }
}

struct TrdWorkspace {
  DevBuf<double> W[2];        // working copy (re / im)
  DevBuf<double> V[2];        // reflectors (vectors only)
  DevBuf<double> vec;         // vb, pb, gp, tau, d, e
  DevBuf<double> scal;
  DevBuf<int> flag;
  DevBuf<unsigned long long> prof;   // (XMCA_TRD_PROF=file) s_memtime stamps
  DevBuf<double> sync;               // exchange buffers of the resident kernel
  DevBuf<unsigned int> flags;
  int resident_used = 0;             // 1: the last reduction ran as the persistent resident kernel, 2: it gave up and was repeated
  int stages_used = 0;               // launches of the persistent form in the last reduction (1, or the links of the chain)
  std::string last_desc;             // the kernels of the last reduction by name (xmca_get_reduction_info; bench.py `roofline.kernel`)
  // hipEvents around the reduction kernel(s) of every call (the dominant kernel of a solve: bench.py `roofline`)
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool ev_pending = false;
  double reduce_ms = 0.0;
  long long reduce_calls = 0;
  long long resident_calls = 0;       // ... of which by the persistent resident kernel
  void ev_begin(hipStream_t st) {
    if (!ev[0]) { XMCA_HIP(hipEventCreate(&ev[0])); XMCA_HIP(hipEventCreate(&ev[1])); }
    XMCA_HIP(hipEventRecord(ev[0], st));
  }
  void ev_end(hipStream_t st) { XMCA_HIP(hipEventRecord(ev[1], st)); ev_pending = true; }
  void ev_collect() {      // after the stream has been synchronised
    if (!ev_pending) return;
    float t = 0.f;
    if (hipEventSynchronize(ev[1]) == hipSuccess && hipEventElapsedTime(&t, ev[0], ev[1]) == hipSuccess) { reduce_ms += t; ++reduce_calls; }
    ev_pending = false;
  }
  ~TrdWorkspace() { if (ev[0]) { (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]); } }
  double ms = 0.0;            // (profiling: accumulated device time of the reduction, when measured)
};

// (the switches are read at every call - a getenv per solve - so that one process can run both routes: tests)
inline bool trd_enabled() {
  const char* e = std::getenv("XMCA_TRIDIAG");
  return !(e && e[0] == '0');
}

// LDS of the step kernel for an n x n problem
inline size_t trd_step_lds(int n, bool cplx) { return (size_t)(((n + 3) + 1) & ~1) * 3 * (cplx ? 2 : 1) * sizeof(double); }

inline bool trd_fits(int n, bool cplx) {
  return trd_step_lds(n, cplx) + 8192 <= (size_t)160 * 1024 && n + 4 <= 16 * TRD_THREADS && (size_t)n * 16 <= (size_t)150 * 1024;
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
    P.prof = nullptr;
    return P;
  }
  static size_t doubles(size_t nv) { return nv * 12 + (size_t)TRD_MAX_WGS * 8; }
};


// resident form: chunks of 128 columns per row (NC) and rows per wave (RR) by problem kind; 0 = does not fit
inline int trd_resident_nc(int n, bool cplx) {
  const char* em = std::getenv("XMCA_TRD_RESIDENT_MIN_N");
  const int min_n = em ? std::atoi(em) : 384;
  const char* eo = std::getenv("XMCA_TRD_RESIDENT");
  const bool on = !(eo && eo[0] == '0');
  if (!on || n < min_n) return 0;
  if (n <= 8 * 128) return 8;
  if (n <= 16 * 128) return 16;
  if (cplx) return n <= 20 * 128 ? 20 : 0;
  return n <= 24 * 128 ? 24 : 0;
}

// Chain of persistent launches (round 6).  The per-column cost of the resident kernel grows with the padded order of its
// instantiation (slots per vector gathered and formed by every workgroup, rows per wave in the pass, publishers in the exchange):
// a freshly packed n = 1000 launch runs 3.2 us per column where the last 1000 columns of an n = 2920 launch cost 5.5 us each.  So the
// reduction is handed from instantiation to instantiation as the trailing block shrinks: stage s reduces the columns
// [j_begin, j_end) of the block that starts at column col0 (a multiple of 256: the slot -> thread and chunk -> lane maps of a
// column do not change, so every sum keeps its order - the bits do not depend on where the chain is cut).
// XMCA_TRD_CHAIN=0: one launch.  XMCA_TRD_BREAKS="c1,c2": cut in front of these columns instead (experiments; multiples of 256).
// bounded spins of the persistent kernel's waits: the long bound for the first persistent launch on EACH device of the process
// (a cold first launch can take longer than 0.2 s), the short one afterwards (advisor, round 5: one process-wide counter gave the
// first launch on a second device the short bound)
inline unsigned int trd_spin_limit() {
  static std::mutex mu;
  static std::map<int, int>* warm = new std::map<int, int>;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  return (*warm)[dev]++ == 0 ? (1u << 20) : (1u << 17);
}

struct TrdStage {
  int nc, rr;            // instantiation
  int col0;              // origin of the stage's block
  int j_begin, j_end;    // columns (global indices) the stage reduces
};
// instantiations a LATER stage of a chain may take (tagged form): NC = 4, 8, 12, 16, 20 (real) chunks of 128 columns; the first stage
// is the one a single launch would take (trd_resident_nc)
inline int trd_stage_nc(int n, bool cplx) {
  for (int nc = 4; nc <= (cplx ? 20 : 24); nc += 4)
    if (n <= nc * 128) return nc;
  return 0;
}
// rows per wave: 256 workgroups x 4 waves x RR rows hold the block (complex: 2 at most - the largest problems stream their first rows)
inline int trd_stage_rr(int nc, bool cplx) { return cplx ? (nc <= 8 ? 1 : 2) : (nc + 7) / 8; }
// ... inside several surrogate lanes (rule_n / bootstrapping) the LATE links of a chain are PACKED: more rows per wave, fewer
// workgroups - a reduction holds its CUs with one 512-register wave per SIMD, nothing of another lane fits beside it, so the CUs a
// link does not take are what the other lane's kernels run on (round 6).  The arithmetic of a row does not depend on its owner:
// same bits.  XMCA_TRD_PACK=0 / 1 forces either form everywhere (experiments).
// Measured (C4, two lanes, ms per surrogate / the reduction alone): unpacked 42.2 / 18.9, rows x 1.5-2 (below) 40.9 / 20.2, rows x 4 in
// the last two links 41.5 / 22.7; real problems (C2-shaped EOF surrogates) lose 2 % with packing and keep the plain chain.
inline int trd_stage_rr_packed(int nc, bool cplx) {
  if (!cplx) return trd_stage_rr(nc, cplx);
  if (nc == 12) return 3;
  if (nc == 8 || nc == 4) return 2;
  return trd_stage_rr(nc, cplx);
}
inline bool trd_pack_lanes() {
  const char* e = std::getenv("XMCA_TRD_PACK");
  return e ? e[0] != '0' : in_surrogate_lanes();
}
inline std::vector<TrdStage> trd_plan(int n, bool cplx, int nc) {
  std::vector<TrdStage> st;
  if (nc <= 0) return st;
  std::vector<int> cuts;
  const char* eb = std::getenv("XMCA_TRD_BREAKS");
  const char* ec = std::getenv("XMCA_TRD_CHAIN");
  const bool chain = !(ec && ec[0] == '0');
  const bool tagged = [] { const char* e = std::getenv("XMCA_TRD_TAGGED"); return e ? e[0] != '0' : true; }();
  if (chain && tagged) {
    if (eb && eb[0]) {
      for (const char* q = eb; *q;) {
        const int c = std::atoi(q);
        if (c > 0 && c % 256 == 0 && c < n - 256 && (cuts.empty() || c > cuts.back())) cuts.push_back(c);
        while (*q && *q != ',') ++q;
        if (*q == ',') ++q;
      }
    } else {
      // as soon as the trailing block fits the next smaller instantiation
      for (int cnc = nc - 4; cnc >= 4; cnc -= 4) {
        const int cap = cnc * 128;
        if (n <= cap) continue;
        const int c = ((n - cap + 255) / 256) * 256;
        if (c < n - 256 && (cuts.empty() || c > cuts.back())) cuts.push_back(c);
      }
    }
  }
  int begin = 0;
  for (size_t q = 0; q <= cuts.size(); ++q) {
    const int end = q < cuts.size() ? cuts[q] : n;
    TrdStage sg{};
    sg.col0 = begin;
    sg.nc = (q == 0 && !(chain && tagged)) ? nc : trd_stage_nc(n - begin, cplx);   // (the tagged chain also STARTS in the smallest instantiation that fits)
    sg.rr = (q > 0 && chain && tagged && trd_pack_lanes()) ? trd_stage_rr_packed(sg.nc, cplx) : trd_stage_rr(sg.nc, cplx);
    sg.j_begin = begin;
    sg.j_end = end;
    st.push_back(sg);
    begin = end;
  }
  return st;
}

// Reduces the Hermitian matrix (Ar, Ai) to tridiagonal form on `st`.  Afterwards P.d / P.e hold the tridiagonal of
// f * A (f = ws.scal[0]), P.tau and (keep_reflectors) ws.V the reflectors.  Returns the parameter block.
inline TrdParams trd_reduce(hipStream_t st, TrdWorkspace& ws, const double* Ar, const double* Ai, int n, int64_t lda, bool keep_reflectors) {
  const bool cplx = Ai != nullptr;
  const int nc = trd_resident_nc(n, cplx);
  // the launches of the persistent form (one, or a chain of re-packed ones): every stage reads whole 128-column chunks of its rows
  const std::vector<TrdStage> stages = trd_plan(n, cplx, nc);
  int64_t ld = std::max<int64_t>(((int64_t)n + 2 + 15) & ~(int64_t)15, (int64_t)nc * 128);
  for (const TrdStage& sg : stages) ld = std::max<int64_t>(ld, (int64_t)sg.col0 + (int64_t)sg.nc * 128);
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
  {
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(ws.scal.get() + 2);
    XMCA_HIP(hipMemsetAsync(acc, 0, 2 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(trd_maxabs_kernel, dim3(std::min(n, 1024)), dim3(256), 0, st, Ar, Ai, n, lda, acc);
    hipLaunchKernelGGL(trd_scale_kernel, dim3(1), dim3(1), 0, st, acc, ws.scal.get());
  }
  hipLaunchKernelGGL(trd_copy_kernel, dim3(n), dim3(256), 0, st, Ar, Ai, n, lda, P.Ar, P.Ai, ld, ws.scal.get());
  ws.resident_used = 0;

  // ---- persistent form: the trailing matrix in registers, one exchange per column ----
  if (nc > 0) {
    PersistGate& gate = persist_gate();       // (per device: CU count and the claims of every persistent kernel)
    const int n_cus = gate.n_cus;
    const size_t lv = (size_t)(ld > (int64_t)nc * 128 ? ld : (int64_t)nc * 128);   // slots of an exchange vector: every stage's padded range
    const size_t nvs = std::max(nv, lv);                   // (the prologue loads every slot of a vector, dead or not)
    const size_t rec = cplx ? 4 : 2;                       // doubles per exchange record (tagged form)
    const size_t sync_doubles = 4 * nvs + 4 * (size_t)TRD_MAX_WGS + 4 * lv + 2 * lv + 8 + 2 * rec * lv;
    ws.sync.ensure(sync_doubles);
    ws.flags.ensure(TRD_MAX_WGS + 32);
    XMCA_HIP(hipMemsetAsync(ws.sync.get(), 0, sizeof(double) * sync_doubles, st));
    XMCA_HIP(hipMemsetAsync(ws.flags.get(), 0, sizeof(unsigned int) * (TRD_MAX_WGS + 32), st));
    TrdSync S{};
    double* q = ws.sync.get();
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) { S.pub[a][c] = q; q += nvs; }
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) { S.gpart[a][c] = q; q += TRD_MAX_WGS; }
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) { S.rowbuf[a][c] = q; q += lv; }
    for (int c = 0; c < 2; ++c) { S.hand_u[c] = q; q += lv; }
    S.hand_s = q; q += 8;
    S.xch = q;
    S.xch_pb = (int)(rec * lv * sizeof(double));
    S.flags = ws.flags.get();
    S.poll_delay = 16;      // (flags form; swept 8...48 in round 3: flat)
    S.tag_delay = 0;        // round 5: the scaling of the previous reflector fills what used to be the delay (swept again 0...32: 0 for every shape; rounds 3-4: 24 x 64 cycles of sleep)
    // column 0 reads its row like every other column: from rowbuf (parity 0)
    // (the tagged form reads row 0 from the working copy: the exchange buffers must start as zeros)
    // exchange by tagged values or by epoch flags.  Measured with the request delay tuned (XMCA_TRD_TAG_DELAY), tagged / flags:
    // real n = 1000 3.7 / 5.0 ms, 2048 11.0 / 13.2, 2920 21.6 / 24.3; complex n = 1000 5.7 / 6.9, 2048 18.4 / 20.4, 2501 26.9 / 28.2
    // (round 4: the tagged NC = 20 build no longer spills - see the publication of row j + 1 in the kernel) - so: tagged for
    // everything (XMCA_TRD_TAGGED=0 keeps the flags, which the tests still run)
    const bool tagged = [] { const char* e = std::getenv("XMCA_TRD_TAGGED"); return e ? e[0] != '0' : true; }();
    // Contiguous row ownership only with the tagged exchange: there the arithmetic of a row does not depend on its owner (same
    // bits for any lane count), and workgroups that leave early feed the other lanes.  In the flags form p^H v is a sum of
    // per-workgroup partials - regrouping rows would change its order and the bits - and nobody leaves early (advisor, round 4).
    S.contiguous = (tagged && in_surrogate_lanes()) ? 1 : 0;
    S.spin_limit = trd_spin_limit();
    if (!tagged) {
      XMCA_HIP(hipMemcpyAsync(S.rowbuf[0][0], P.Ar, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
      if (cplx) XMCA_HIP(hipMemcpyAsync(S.rowbuf[0][1], P.Ai, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
    }
    if (tagged) hipLaunchKernelGGL(trd_prefill_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, P.Ar, P.Ai, n, S.xch);
    S.give_up = reinterpret_cast<int*>(ws.flags.get() + TRD_MAX_WGS);
    using ResFn = void (*)(TrdParams, TrdSync, int);
    auto pick = [&](int snc, int srr) -> ResFn {
      if (tagged) {
        if (cplx) {
          switch (snc) {
            case 4: return srr == 2 ? trd_resident_kernel<true, 4, 2, true> : trd_resident_kernel<true, 4, 1, true>;
            case 8: return srr == 2 ? trd_resident_kernel<true, 8, 2, true> : trd_resident_kernel<true, 8, 1, true>;
            case 12: return srr == 3 ? trd_resident_kernel<true, 12, 3, true> : trd_resident_kernel<true, 12, 2, true>;
            case 16: return trd_resident_kernel<true, 16, 2, true>;
            default: return trd_resident_kernel<true, 20, 2, true>;
          }
        }
        switch (snc) {
          case 4: return trd_resident_kernel<false, 4, 1, true>;
          case 8: return trd_resident_kernel<false, 8, 1, true>;
          case 12: return trd_resident_kernel<false, 12, 2, true>;
          case 16: return trd_resident_kernel<false, 16, 2, true>;
          case 20: return trd_resident_kernel<false, 20, 3, true>;
          default: return trd_resident_kernel<false, 24, 3, true>;
        }
      }
      if (cplx) return snc == 8 ? trd_resident_kernel<true, 8, 1, false> : snc == 16 ? trd_resident_kernel<true, 16, 2, false> : trd_resident_kernel<true, 20, 2, false>;
      return snc == 8 ? trd_resident_kernel<false, 8, 1, false> : snc == 16 ? trd_resident_kernel<false, 16, 2, false> : trd_resident_kernel<false, 24, 3, false>;
    };
    static const char* prof_file_r = std::getenv("XMCA_TRD_PROF");
    if (prof_file_r) {
      P.prof = ws.prof.ensure((size_t)8 * n);
      XMCA_HIP(hipMemsetAsync(P.prof, 0, sizeof(unsigned long long) * 8 * (size_t)n, st));
    }
    int gave_up = 0;
    {
      // every CU of the device, exclusively among the persistent kernels (common.h PersistGate): held until the stream
      // has finished the reduction
      PersistGate::Claim claim(gate, n_cus);
      ws.ev_begin(st);
      for (size_t q = 0; q < stages.size(); ++q) {
        const TrdStage& sg = stages[q];
        const int c0 = sg.col0, ns = n - c0;
        const int64_t off = (int64_t)c0 * ld + c0;
        // the stage's view: the trailing block from (c0, c0) on, every vector from slot c0 on
        TrdParams Pq = P;
        Pq.n = ns;
        Pq.Ar = P.Ar + off;
        if (P.Ai) Pq.Ai = P.Ai + off;
        Pq.d = P.d + c0; Pq.e = P.e + c0;
        Pq.tau[0] = P.tau[0] + c0; Pq.tau[1] = P.tau[1] + c0;
        if (P.Vr) Pq.Vr = P.Vr + off;
        if (P.Vi) Pq.Vi = P.Vi + off;
        if (P.prof) Pq.prof = P.prof + (size_t)8 * c0;
        TrdSync Sq = S;
        for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) { Sq.pub[a][c] = S.pub[a][c] + c0; Sq.rowbuf[a][c] = S.rowbuf[a][c] + c0; }
        Sq.hand_u[0] = S.hand_u[0] + c0; Sq.hand_u[1] = S.hand_u[1] + c0;
        Sq.xch = S.xch + 2 * (size_t)c0;     // (16 bytes per slot in each block of records)
        Sq.j_begin = sg.j_begin - c0;
        Sq.j_end = sg.j_end - c0;
        Sq.cont = q > 0 ? 1 : 0;
        const int wgs = std::max(1, std::min(std::min(n_cus, TRD_MAX_WGS), ceil_div(ns, (TRD_RES_THREADS / 64) * sg.rr)));
        const int first_res = std::max(0, ns - wgs * (TRD_RES_THREADS / 64) * sg.rr);
        const size_t lds = (size_t)sg.nc * 128 * 3 * (cplx ? 2 : 1) * sizeof(double);
        ResFn fn = pick(sg.nc, sg.rr);
        XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        hipLaunchKernelGGL(fn, dim3(wgs), dim3(TRD_RES_THREADS), lds, st, Pq, Sq, first_res);
      }
      ws.ev_end(st);
      XMCA_HIP(hipGetLastError());
      int dbg[4] = {0};
      XMCA_HIP(hipMemcpyAsync(dbg, S.give_up, sizeof(dbg), hipMemcpyDeviceToHost, st));
      XMCA_HIP(hipStreamSynchronize(st));
      gave_up = dbg[0];
      if (gave_up && xmca_trace("giveup"))
        std::fprintf(stderr, "xmca: trd_resident_kernel: workgroup %d ran out of its spins at column %d of its launch (n = %d, %d launches)\n", dbg[3], dbg[2], n, (int)stages.size());
    }
    ws.stages_used = (int)stages.size();
    {
      std::string d = stages.size() > 1 ? "chain of " + std::to_string(stages.size()) + " persistent launches: " : "";
      for (size_t q = 0; q < stages.size(); ++q) {
        const TrdStage& sg = stages[q];
        d += (q ? " -> " : "") + std::string("trd_resident_kernel<") + (cplx ? "complex" : "real") + ",NC=" + std::to_string(sg.nc) + ",RR=" + std::to_string(sg.rr) +
             (tagged ? ",tagged>" : ",flags>") + " columns [" + std::to_string(sg.j_begin) + "," + std::to_string(sg.j_end) + ")";
      }
      ws.last_desc = d;
    }
    if (prof_file_r) {
      std::vector<unsigned long long> hp((size_t)8 * n);
      XMCA_HIP(hipMemcpy(hp.data(), P.prof, sizeof(unsigned long long) * hp.size(), hipMemcpyDeviceToHost));
      if (FILE* f = std::fopen(prof_file_r, "w")) {
        for (int j = 0; j < n; ++j) {
          for (int q = 0; q < 7; ++q) std::fprintf(f, "%llu ", hp[(size_t)8 * j + q] - (q ? hp[(size_t)8 * j] : (j ? hp[(size_t)8 * (j - 1)] : hp[0])));
          std::fprintf(f, "\n");
        }
        std::fclose(f);
      }
      P.prof = nullptr;
    }
    if (!gave_up) {
      ws.resident_used = 1;
      ++ws.resident_calls;
      ws.ev_collect();
      return P;
    }
    ws.ev_pending = false;
    // a workgroup never became resident (another process holds CUs?): start again with one launch per column
    ws.resident_used = 2;
    ++persist_giveups();
    if (xmca_trace("giveup")) std::fprintf(stderr, "xmca: trd_resident_kernel (n = %d) gave up - repeated with one launch per column\n", n);
    hipLaunchKernelGGL(trd_copy_kernel, dim3(n), dim3(256), 0, st, Ar, Ai, n, lda, P.Ar, P.Ai, ld, ws.scal.get());
    XMCA_HIP(hipMemsetAsync(ws.vec.get(), 0, sizeof(double) * TrdLayout::doubles(nv), st));
    if (keep_reflectors) {
      XMCA_HIP(hipMemsetAsync(ws.V[0].get(), 0, sizeof(double) * (size_t)n * ld, st));
      if (cplx) XMCA_HIP(hipMemsetAsync(ws.V[1].get(), 0, sizeof(double) * (size_t)n * ld, st));
    }
  }

  ws.last_desc = std::string("trd_step_kernel<") + (cplx ? "complex" : "real") + "> x " + std::to_string(n) + " launches (one per column)";
  const size_t lds = trd_step_lds(n, cplx);
  const int slots = ((n + 3) + 1) & ~1;
  const int ns = slots <= 2 * TRD_THREADS ? 2 : slots <= 4 * TRD_THREADS ? 4 : slots <= 8 * TRD_THREADS ? 8 : 16;
  using StepFn = void (*)(TrdParams, int, int);
  StepFn fn = nullptr;
  XMCA_CHECK(!cplx || ns <= 8, XMCA_ERR_UNSUPPORTED, "trd_reduce: complex problem too large for the LDS of a workgroup");   // (trd_fits)
  if (cplx) fn = ns == 2 ? trd_step_kernel<true, 2> : ns == 4 ? trd_step_kernel<true, 4> : trd_step_kernel<true, 8>;
  else fn = ns == 2 ? trd_step_kernel<false, 2> : ns == 4 ? trd_step_kernel<false, 4> : ns == 8 ? trd_step_kernel<false, 8> : trd_step_kernel<false, 16>;
  XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  static const char* prof_file = std::getenv("XMCA_TRD_PROF");     // experiments: phase stamps of workgroup 0, one line per launch
  if (prof_file) {
    P.prof = ws.prof.ensure((size_t)8 * n);
    XMCA_HIP(hipMemsetAsync(P.prof, 0, sizeof(unsigned long long) * 8 * (size_t)n, st));
  }
  int wgs_prev = 1;
  ws.ev_begin(st);
  for (int j = 0; j < n; ++j) {
    const int m = n - j - 1;
    const int wgs = std::max(1, std::min(TRD_MAX_WGS, (m + TRD_ROWS - 1) / TRD_ROWS));
    hipLaunchKernelGGL(fn, dim3(wgs), dim3(TRD_THREADS), lds, st, P, j, wgs_prev);
    wgs_prev = wgs;
  }
  ws.ev_end(st);
  XMCA_HIP(hipGetLastError());
  if (prof_file) {
    std::vector<unsigned long long> hp((size_t)8 * n);
    XMCA_HIP(hipMemcpyAsync(hp.data(), P.prof, sizeof(unsigned long long) * hp.size(), hipMemcpyDeviceToHost, st));
    XMCA_HIP(hipStreamSynchronize(st));
    if (FILE* f = std::fopen(prof_file, "w")) {
      for (int j = 0; j < n; ++j) {
        for (int q = 0; q < 7; ++q) std::fprintf(f, "%llu ", hp[(size_t)8 * j + q] - (q ? hp[(size_t)8 * j] : 0ull));
        std::fprintf(f, "\n");
      }
      std::fclose(f);
    }
  }
  return P;
}

// all eigenvalues, descending, into lam_dev (device, n doubles; may be nullptr) and lam_host.  Synchronises `st`.
inline void trd_eigenvalues(hipStream_t st, TrdWorkspace& ws, const TrdParams& P, std::vector<double>& lam_host, double* lam_dev,
                            DevBuf<double>& lam_tmp, double* lam_asc_scaled = nullptr) {
  const int n = P.n;
  double* out = lam_dev ? lam_dev : lam_tmp.ensure((size_t)n);
  XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trd_bisect_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  hipLaunchKernelGGL(trd_bisect_kernel, dim3(ceil_div(n, TRD_BIS_THREADS / 16)), dim3(TRD_BIS_THREADS), sizeof(double) * 2 * (size_t)n, st, P.d,
                     P.e, n, ws.scal.get(), out, lam_asc_scaled, ws.flag.get());
  XMCA_HIP(hipGetLastError());
  lam_host.resize((size_t)n);
  int flag = 0;
  double scal[2] = {1.0, 0.0};
  XMCA_HIP(hipMemcpyAsync(lam_host.data(), out, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  XMCA_HIP(hipMemcpyAsync(&flag, ws.flag.get(), sizeof(int), hipMemcpyDeviceToHost, st));
  XMCA_HIP(hipMemcpyAsync(scal, ws.scal.get(), sizeof(double) * 2, hipMemcpyDeviceToHost, st));
  XMCA_HIP(hipStreamSynchronize(st));
  ws.ev_collect();
  XMCA_CHECK(flag == 0 && scal[1] == 0.0, XMCA_ERR_NUMERIC, "SVD failed. NaN entries may be the problem.");
}

}  // namespace xmca
