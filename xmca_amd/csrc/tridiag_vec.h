// Eigenvectors for the tridiagonal route (tridiag.h):  A = Q T Q^H,  T y_k = lambda_k y_k,  u_k = Q y_k.
//
//   1. trd_twisted_kernel - one lane pair per eigenvalue: the two stationary factorisations of T - lambda_k I (forward L D+ L^T,
//      backward U D- U^T), the twist index r = argmin |gamma_r| and the vector z(r) = 1, z(i) = -(e_i / D+_i) z(i+1) upwards,
//      z(i+1) = -(e_i / D-_{i+1}) z(i) downwards (Parlett & Dhillon; LAPACK dlar1v without the RRR shifts).  With an
//      eigenvalue accurate to eps ||T|| the residual is eps ||T||; vectors of close eigenvalues are orthogonal only to
//      eps ||T|| / gap - the clean-up below restores that;
//   2. back-transformation with super-blocks of 512 reflectors in compact WY form, Z -= V (T (V^H Z)): two MFMA GEMMs per
//      super-block, X = (T V^H) Z and Z -= V X.  T depends on the reflectors alone and is built for ALL super-blocks ahead
//      of the loop in a handful of block-sparse launches (gemm.h GemmTileList): with N = I + diag(tau) striu(V^H V),
//      T = N^{-1} diag(tau) (from T^{-1} + T^{-H} = V^H V; valid for tau = 0 as well); the 64 x 64 diagonal blocks of T by
//      back substitution (trd_wy_tinv_kernel, one thread per column), then T12 = -T11 S12 T22 level by level (64 -> 128 ->
//      256 -> 512), two launches per level, and T V^H in one more.  (Rounds 2-3 solved T^{-1} X = W per block of 128 with a
//      thread per column of Z, between the two GEMMs: 23 x 230 us at n = 2920 against 6 x ~200 us + 0.3 ms now.)
//   3. S = Z^H Z; max |S - I| decides: <= 0.3 -> one or two Newton-Schulz steps Z <- Z (3/2 I - 1/2 S) (the last one
//      written as (3/2 I - 1/2 S)(rows reversed) Z^H, i.e. straight into the caller's layout: row i = conj(u_i), eigenvalues
//      descending); otherwise (clusters the twisted vectors cannot resolve: repeated eigenvalues, exact null spaces of
//      dimension > 1) the caller falls back to the Jacobi solver.
// CPU model: scripts/experiments/tridiag_model.py.
#pragma once
#include "cholesky.h"   // cgemm
#include "tridiag.h"

namespace xmca {

// value of lane `u` (uniform) for every lane: the tridiagonal's entries reach the recurrences below as one coalesced load per
// 64 rows and a v_readlane pair per row - an s_load per row (what `e[i]` with a uniform i compiles to) stalls every step of a
// dependent chain on the scalar cache (s_waitcnt lgkmcnt(0) inside the recurrence: 1.6 ms at n = 2920 instead of 0.9)
__device__ __forceinline__ double trd_lane_value(double v, int u) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), u), hi = __builtin_amdgcn_readlane(__double2hiint(v), u);
  return __hiloint2double(hi, lo);
}

// Yt[i * ldy + k] = component i of the eigenvector of eigenvalue k (ascending);  W: work plane of the same shape.
// 128 threads per 64 eigenvalues: the forward (D+) and the backward (D-) factorisation are independent recurrences and run
// in the two waves side by side; so do the two halves of the vector (upwards / downwards from the twist index) and of the
// final scaling.  Every loop requests its operands one chunk (64 rows of d / e, 32 rows of the planes) ahead of the recurrence.
__global__ __launch_bounds__(128) void trd_twisted_kernel(const double* __restrict__ d, const double* __restrict__ e, int n,
                                                         const double* __restrict__ lam, double* __restrict__ W, double* __restrict__ Yt,
                                                         int64_t ldy) {
  __shared__ double nrm_sh[2][64];
  __shared__ int idx_sh[2][64];
  const int lane = threadIdx.x & 63, half = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const bool live = k < n;
  const int kk = live ? k : n - 1;                           // (idle lanes shadow the last eigenvalue and store nothing)
  const double l = lam[kk];
  if (n == 1) { if (live && half == 0) Yt[k] = 1.0; return; }
  constexpr int B = 32;
  double emax = 0.0;
  for (int i = lane; i < n - 1; i += 64) emax = fmax(emax, fabs(e[i]));
  for (int o = 32; o > 0; o >>= 1) emax = fmax(emax, __shfl_xor(emax, o));
  const double piv = 2.3e-308 * fmax(1.0, emax * emax) * 1e20 + 1e-300;
  if (half == 0) {
    // forward: D+ into Yt.  Row i0 + u of the chunk: lane u holds e[i0 + u] and d[i0 + u + 1]
    double dp = d[0] - l;
    auto chunk = [&](int i0) { const int i = i0 + lane < n - 1 ? i0 + lane : n - 2; return i; };
    double ev = e[chunk(0)], dv = d[chunk(0) + 1];
    for (int i0 = 0; i0 < n - 1; i0 += 64) {
      const int nx = i0 + 64 < n - 1 ? i0 + 64 : i0;
      const double ev_n = e[chunk(nx)], dv_n = d[chunk(nx) + 1];
      const int cnt = n - 1 - i0 < 64 ? n - 1 - i0 : 64;
      auto step = [&](int u) {
        if (!(fabs(dp) >= piv)) dp = dp < 0.0 ? -piv : piv;
        if (live) Yt[(int64_t)(i0 + u) * ldy + k] = dp;
        const double ei = trd_lane_value(ev, u);
        dp = (trd_lane_value(dv, u) - l) - (ei * trd_rcp(dp)) * ei;
      };
      if (cnt == 64) {
#pragma unroll
        for (int u = 0; u < 64; ++u) step(u);
      } else {
        for (int u = 0; u < cnt; ++u) step(u);
      }
      ev = ev_n; dv = dv_n;
    }
    if (!(fabs(dp) >= piv)) dp = dp < 0.0 ? -piv : piv;
    if (live) Yt[(int64_t)(n - 1) * ldy + k] = dp;
  } else {
    // backward: D- into W.  Row i0 - u of the chunk (i0 downwards from n - 2): lane u holds e[i0 - u] and d[i0 - u]
    double dm = d[n - 1] - l;
    if (!(fabs(dm) >= piv)) dm = dm < 0.0 ? -piv : piv;
    if (live) W[(int64_t)(n - 1) * ldy + k] = dm;
    auto chunk = [&](int i0) { const int i = i0 - lane >= 0 ? i0 - lane : 0; return i; };
    double ev = e[chunk(n - 2)], dv = d[chunk(n - 2)];
    for (int i0 = n - 2; i0 >= 0; i0 -= 64) {
      const int nx = i0 - 64 >= 0 ? i0 - 64 : i0;
      const double ev_n = e[chunk(nx)], dv_n = d[chunk(nx)];
      const int cnt = i0 + 1 < 64 ? i0 + 1 : 64;
      auto step = [&](int u) {
        const double ei = trd_lane_value(ev, u);
        dm = (trd_lane_value(dv, u) - l) - (ei * trd_rcp(dm)) * ei;
        if (!(fabs(dm) >= piv)) dm = dm < 0.0 ? -piv : piv;
        if (live) W[(int64_t)(i0 - u) * ldy + k] = dm;
      };
      if (cnt == 64) {
#pragma unroll
        for (int u = 0; u < 64; ++u) step(u);
      } else {
        for (int u = 0; u < cnt; ++u) step(u);
      }
      ev = ev_n; dv = dv_n;
    }
  }
  __threadfence_block();
  __syncthreads();
  // twist index: gamma_i = D+_i + D-_i - (d_i - lambda), smallest |gamma| (no recurrence, pure streaming: each wave scans one
  // half of the rows, the lower half wins ties - the first occurrence of the minimum, as a single scan would find it; rows
  // past a half's end repeat its last row, which cannot displace that)
  int r = 0;
  {
    const int h0s = half == 0 ? 0 : n / 2, h1s = half == 0 ? n / 2 : n;     // (n >= 2: neither half is empty)
    double gbest = 1.7e308;
    r = h0s;
    for (int i0 = h0s; i0 < h1s; i0 += B) {
      double yp[B], wm[B];
      const double dv = d[i0 + lane < h1s ? i0 + lane : h1s - 1];
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int i = i0 + u < h1s ? i0 + u : h1s - 1;
        yp[u] = Yt[(int64_t)i * ldy + kk];
        wm[u] = W[(int64_t)i * ldy + kk];
      }
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int i = i0 + u < h1s ? i0 + u : h1s - 1;
        const double g = fabs(yp[u] + wm[u] - (trd_lane_value(dv, u) - l));
        if (g < gbest) { gbest = g; r = i; }
      }
    }
    nrm_sh[half][lane] = gbest;
    idx_sh[half][lane] = r;
    __syncthreads();
    if (nrm_sh[0][lane] <= nrm_sh[1][lane]) r = idx_sh[0][lane];
    else r = idx_sh[1][lane];
  }
  __syncthreads();                                           // (both waves have read D+ before the upper half overwrites it)
  double nrm = 0.0;
  // The twist index differs from lane to lane, the ROW index of the two vector recurrences must not: with per-lane rows every
  // load and store of a wave touched 64 different cache lines (round 3-4: 500 of the kernel's 970 us, 16 x the bytes).  Both
  // loops therefore walk the rows of the whole wave - from the largest twist index down, from the smallest one up - and a lane
  // joins when the walk reaches its own twist: coalesced rows, e[i] by lane broadcast, the same arithmetic per lane as before.
  constexpr int B3 = 64;
  if (half == 0) {
    // upwards from the twist: z(i) = -(e_i / D+_i) z(i+1)
    double z = 1.0;
    nrm = 1.0;
    if (live) Yt[(int64_t)r * ldy + k] = 1.0;
    int rmax = r;
    for (int o = 32; o > 0; o >>= 1) rmax = max(rmax, __shfl_xor(rmax, o));
    for (int i0 = rmax - 1; i0 >= 0; i0 -= B3) {
      double yp[B3];
      const double ev = e[i0 - lane >= 0 ? i0 - lane : 0];
#pragma unroll
      for (int u = 0; u < B3; ++u) yp[u] = Yt[(int64_t)(i0 - u >= 0 ? i0 - u : 0) * ldy + kk];
#pragma unroll
      for (int u = 0; u < B3; ++u) {
        const int i = i0 - u;
        if (i >= 0) {                                        // (uniform)
          const double zn = -(trd_lane_value(ev, u) * trd_rcp(yp[u])) * z;
          if (i < r) {
            z = zn;
            if (live) Yt[(int64_t)i * ldy + k] = z;
            nrm += z * z;
          }
        }
      }
    }
  } else {
    // downwards: z(i+1) = -(e_i / D-_{i+1}) z(i)
    double z = 1.0;
    int rmin = r;
    for (int o = 32; o > 0; o >>= 1) rmin = min(rmin, __shfl_xor(rmin, o));
    for (int i0 = rmin; i0 < n - 1; i0 += B3) {
      double wm[B3];
      const double ev = e[i0 + lane < n - 1 ? i0 + lane : n - 2];
#pragma unroll
      for (int u = 0; u < B3; ++u) wm[u] = W[(int64_t)((i0 + u < n - 1 ? i0 + u : n - 2) + 1) * ldy + kk];
#pragma unroll
      for (int u = 0; u < B3; ++u) {
        const int i = i0 + u;
        if (i < n - 1) {                                     // (uniform)
          const double zn = -(trd_lane_value(ev, u) * trd_rcp(wm[u])) * z;
          if (i >= r) {
            z = zn;
            if (live) Yt[(int64_t)(i + 1) * ldy + k] = z;
            nrm += z * z;
          }
        }
      }
    }
  }
  nrm_sh[half][lane] = nrm;
  __threadfence_block();
  __syncthreads();
  const double s = 1.0 / sqrt(nrm_sh[0][lane] + nrm_sh[1][lane]);
  const int h0 = half == 0 ? 0 : n / 2, h1 = half == 0 ? n / 2 : n;
  for (int i0 = h0; i0 < h1; i0 += B3) {
    double y[B3];
#pragma unroll
    for (int u = 0; u < B3; ++u) y[u] = Yt[(int64_t)(i0 + u < h1 ? i0 + u : h1 - 1) * ldy + kk];
#pragma unroll
    for (int u = 0; u < B3; ++u)
      if (live && i0 + u < h1) Yt[(int64_t)(i0 + u) * ldy + k] = y[u] * s;
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

// The 64 x 64 diagonal blocks of T = N^{-1} diag(tau), N = I + diag(tau) striu(S), S = V^H V: column c of a block is the back
// substitution x_R = (delta_Rc - sum_{l > R} S_Rl x_l) tau_R, R = 63 .. 0 (one thread per column, the block of S in LDS,
// the column in registers; x_R = 0 below the diagonal comes out of the recurrence).  S and T in the band layout of
// trd_eigenvectors: element (r, q) at [r * ldb + q] with absolute indices.  tau beyond nref counts as 0 (T row = 0).
template <bool CPLX>
__global__ __launch_bounds__(64) void trd_wy_tinv_kernel(const double* __restrict__ Sr, const double* __restrict__ Si, double* __restrict__ Tr,
                                                         double* __restrict__ Ti, int64_t ldb, int nref, const double* __restrict__ taur,
                                                         const double* __restrict__ taui) {
  __shared__ __attribute__((aligned(16))) double mr[64][64], mi[CPLX ? 64 : 1][64];
  const int o = blockIdx.x * 64, c = threadIdx.x;
  {
    double ga[64], gb[CPLX ? 64 : 1];
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      ga[r] = Sr[(int64_t)(o + r) * ldb + o + c];
      if (CPLX) gb[r] = Si[(int64_t)(o + r) * ldb + o + c];
    }
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      double a = 0.0, b = 0.0;
      if (c > r) {
        a = ga[r];
        if (CPLX) b = gb[r];
      } else if (c == r && o + r < nref) {
        a = taur[o + r];
        if (CPLX) b = taui[o + r];
      }
      mr[r][c] = a;
      if (CPLX) mi[r][c] = b;
    }
  }
  __syncthreads();
  double xr[64], xi[CPLX ? 64 : 1];
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    xr[r] = r == c ? 1.0 : 0.0;
    if (CPLX) xi[r] = 0.0;
  }
  trd_wy_rows<CPLX, 63>::run(xr, xi, mr, mi);
#pragma unroll
  for (int r = 0; r < 64; ++r) {           // (row r of the block: 64 consecutive doubles across the lanes)
    Tr[(int64_t)(o + r) * ldb + o + c] = xr[r];
    if (CPLX) Ti[(int64_t)(o + r) * ldb + o + c] = xi[r];
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

// Tile lists of the compact-WY precomputation for one matrix order (kept in the workspace: the order rarely changes)
constexpr int TRD_SBK = 512;                 // reflectors per super-block of the back-transformation (4 block tiles)
struct TrdBackPlan {
  int n = 0;
  GemmTileList gram;                         // S = conj(V) V^T inside the diagonal super-blocks, upper block triangle
  GemmTileList lev_p[3], lev_t[3];           // levels h = 64, 128, 256:  P = S12 T22,  T12 = -T11 P
  GemmTileList tv;                           // T conj(V)
  void build(hipStream_t st, int n_) {
    n = n_;
    constexpr int Q = TRD_SBK / 128;
    const int nt = ceil_div(n, 128), nsb = ceil_div(n, TRD_SBK);
    std::vector<int> t, k;
    auto add = [&](int bm, int bn, int k0, int k1) { t.push_back((bm << 16) | bn); k.push_back(k0); k.push_back(k1); };
    auto done = [&](GemmTileList& l) { l.upload(st, t, k); t.clear(); k.clear(); };
    // reflector r is zero up to column r: the contraction starts at the block row's first column
    for (int s = 0; s < nsb; ++s)
      for (int bm = Q * s; bm < std::min(Q * s + Q, nt); ++bm)
        for (int bn = bm; bn < std::min(Q * s + Q, nt); ++bn) add(bm, bn, 128 * bm, n);
    done(gram);
    // h = 64: both halves of a pair inside one block tile
    for (int b = 0; b < nt; ++b) add(b, b, 128 * b + 64, 128 * b + 128);
    done(lev_p[0]);
    for (int b = 0; b < nt; ++b) add(b, b, 128 * b, 128 * b + 64);
    done(lev_t[0]);
    // h = 128: block tiles (2 z, 2 z + 1) of a super-block
    for (int lv = 0; lv < 2; ++lv) {
      for (int s = 0; s < nsb; ++s)
        for (int z = 0; z < Q / 2; ++z) {
          const int a = Q * s + 2 * z;
          if (a + 1 < nt) add(a, a + 1, 128 * (a + (lv == 0 ? 1 : 0)), 128 * (a + (lv == 0 ? 2 : 1)));
        }
      done(lv == 0 ? lev_p[1] : lev_t[1]);
    }
    // h = 256: block tiles (a, b), a in the first half of the super-block, b in the second
    for (int lv = 0; lv < 2; ++lv) {
      for (int s = 0; s < nsb; ++s)
        for (int a = Q * s; a < Q * s + Q / 2; ++a)
          for (int b = Q * s + Q / 2; b < std::min(Q * s + Q, nt); ++b)
            add(a, b, 128 * (Q * s + (lv == 0 ? Q / 2 : 0)), 128 * (Q * s + (lv == 0 ? Q : Q / 2)));
      done(lv == 0 ? lev_p[2] : lev_t[2]);
    }
    // T conj(V): row block bm of T is zero left of its diagonal tile and right of its super-block; V is zero left of the
    // super-block's first column in these rows
    for (int bm = 0; bm < nt; ++bm) {
      const int s = bm / Q;
      for (int bn = Q * s; bn < nt; ++bn) add(bm, bn, 128 * bm, std::min(128 * (Q * s + Q), n));
    }
    done(tv);
  }
};

struct TrdVecWorkspace {
  DevBuf<double> Y[2];        // Yt / Z (n x ld planes)
  DevBuf<double> Wk[2];       // work planes (twisted factorisation; Newton-Schulz)
  DevBuf<double> S[2];        // Gram matrix of the vectors
  DevBuf<double> small;       // X of one super-block (512 x ld planes)
  DevBuf<double> band[2];     // S, T, P of the compact-WY precomputation (band layout), re / im
  DevBuf<double> TV[2];       // T V^H (n x ldv planes)
  TrdBackPlan plan;
  // the compact-WY factors depend on the reflectors alone: they are built on a second stream while the eigenvalues and the
  // vectors of the tridiagonal matrix - two kernels that leave most of the chip idle - run on the caller's
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  GemmWorkspace side_gws;
  bool prepared = false, joined = true;
  int calls = 0;
  // Called once per eigen-decomposition with vectors, BEFORE the reduction is queued.  The second stream is made on the second
  // call - and with every CU of the device claimed at the gate of the persistent kernels: creating a stream (a hardware queue)
  // while a persistent grid of another surrogate lane is in flight stalled that grid long enough for its bounded spins to run
  // out (a give-up and a repeated reduction in the first rotated rule_n call after the warm-up, four lanes).
  void count_call() {
    if (side || in_surrogate_lanes() || ++calls < 2) return;      // (lanes keep to one stream each: common.h)
    PersistGate& gate = persist_gate();
    PersistGate::Claim alone(gate, gate.n_cus);
    side_init();
  }
  void side_init() {
    if (side) return;
    XMCA_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    XMCA_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    XMCA_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  }
  ~TrdVecWorkspace() {
    if (side) {
      (void)hipStreamSynchronize(side);
      (void)hipEventDestroy(ev_fork);
      (void)hipEventDestroy(ev_join);
      (void)hipStreamDestroy(side);
    }
  }
  TrdVecWorkspace() = default;
  TrdVecWorkspace(const TrdVecWorkspace&) = delete;
  TrdVecWorkspace& operator=(const TrdVecWorkspace&) = delete;
  DevBuf<double> lam_asc;
  DevBuf<unsigned long long> orth;
  double last_orth = 0.0;     // max |Z^H Z - I| before the clean-up of the last call
  int ns_steps = 0;
};

// T of every super-block and T V^H from the reflectors of trd_reduce(..., keep_reflectors = true), queued on the workspace's
// second stream behind everything `st` holds so far; trd_eigenvectors joins it in front of the back-transformation.
inline void trd_wy_prepare(hipStream_t st, TrdVecWorkspace& vw, GemmWorkspace& gws_main, const TrdParams& P, bool cplx) {
  const int n = P.n;
  const int64_t ldv = P.ld;                                   // reflectors: row j = v_j, zero outside its support j+1 .. n-1
  constexpr int SBK = TRD_SBK;
  const int nref = n - 1;
  if (vw.prepared && vw.side) XMCA_HIP(hipStreamSynchronize(vw.side));    // (an earlier call left without joining: an exception in between)
  vw.prepared = false;
  if (nref <= 0) return;
  if (vw.plan.n != n) vw.plan.build(st, n);
  // Creating a stream costs ~10 ms (measured: hipStreamCreateWithFlags inside the first solve of a process) - thirty solves'
  // worth of what the overlap saves.  The first call of a workspace therefore stays on `st`; from the second call on the
  // workspace belongs to somebody who solves repeatedly, and the second stream exists (count_call).
  const bool use_side = vw.side != nullptr && !in_surrogate_lanes();   // (made by TrdVecWorkspace::count_call, in front of the reduction)
  // Everything the second stream touches is allocated OUTSIDE the handle's pool: a pool hands a released block to the next
  // taker on the strength of stream order, which holds for one stream only (a split-K workspace that grows between two GEMMs
  // of this function would otherwise go back to the pool - and to a kernel on `st` - while the first GEMM still writes it).
  // hipFree synchronises the device; the sizes settle after the first call of a given order.
  PoolScope scope(use_side ? nullptr : current_pool());
  if (use_side) {
    XMCA_HIP(hipEventRecord(vw.ev_fork, st));
    XMCA_HIP(hipStreamWaitEvent(vw.side, vw.ev_fork, 0));
  }
  {
    hipStream_t st_main = st;
    hipStream_t st = use_side ? vw.side : st_main;            // (everything below runs on the second stream when there is one)
    GemmWorkspace& gws = use_side ? vw.side_gws : gws_main;
    // ---- T of every super-block and T V^H, from the reflectors alone ----
    // Band layout: element (r, q) of S / T / P at [r * SBK + q] with ABSOLUTE indices - only entries of one diagonal
    // super-block are ever touched, so super-block s sits at s * (SBK * SBK + SBK) and the three matrices are ordinary GEMM
    // operands with leading dimension SBK; rows and columns past n stay zero.
    const TrdBackPlan& pl = vw.plan;
    const int nsb = ceil_div(n, SBK), npad = nsb * SBK;
    const size_t bsz = (size_t)nsb * ((size_t)SBK * SBK + SBK) + SBK;
    double* bnd[2] = {vw.band[0].ensure(3 * bsz), cplx ? vw.band[1].ensure(3 * bsz) : nullptr};
    for (int c = 0; c < (cplx ? 2 : 1); ++c) XMCA_HIP(hipMemsetAsync(bnd[c], 0, sizeof(double) * 3 * bsz, st));
    double *Sr_ = bnd[0], *Tr_ = bnd[0] + bsz, *Pr_ = bnd[0] + 2 * bsz;
    double *Si_ = cplx ? bnd[1] : nullptr, *Ti_ = cplx ? bnd[1] + bsz : nullptr, *Pi_ = cplx ? bnd[1] + 2 * bsz : nullptr;
    // S[r][q] = sum_i conj(V[r][i]) V[q][i]
    cgemm<double>(st, gws, P.Vr, P.Vi, ldv, true, true, P.Vr, P.Vi, ldv, false, false, Sr_, Si_, SBK, n, n, n, 1.0, nullptr, nullptr, false,
                  0.0, &pl.gram);
    if (cplx) hipLaunchKernelGGL(trd_wy_tinv_kernel<true>, dim3(ceil_div(n, 64)), dim3(64), 0, st, Sr_, Si_, Tr_, Ti_, (int64_t)SBK, nref, P.tau[0], P.tau[1]);
    else hipLaunchKernelGGL(trd_wy_tinv_kernel<false>, dim3(ceil_div(n, 64)), dim3(64), 0, st, Sr_, nullptr, Tr_, nullptr, (int64_t)SBK, nref, P.tau[0], nullptr);
    for (int lv = 0; lv < 3; ++lv) {
      // P = S12 T22 (contraction over the second half of the pair), T12 = -T11 P (over the first half).  Rows of a block
      // tile outside the wanted half see exact zeros of T (upper triangular, block diagonal so far) on one side of every
      // product, so whole tiles can be computed and stored.
      cgemm<double>(st, gws, Sr_, Si_, SBK, true, false, Tr_, Ti_, SBK, true, false, Pr_, Pi_, SBK, npad, npad, npad, 1.0, nullptr, nullptr,
                    false, 0.0, &pl.lev_p[lv]);
      cgemm<double>(st, gws, Tr_, Ti_, SBK, true, false, Pr_, Pi_, SBK, true, false, Tr_, Ti_, SBK, npad, npad, npad, -1.0, nullptr, nullptr,
                    false, 1.0, &pl.lev_t[lv]);
    }
    // TV[r][i] = sum_q T[r][q] conj(V[q][i])
    double* TVr = vw.TV[0].ensure((size_t)n * ldv);
    double* TVi = cplx ? vw.TV[1].ensure((size_t)n * ldv) : nullptr;
    cgemm<double>(st, gws, Tr_, Ti_, SBK, true, false, P.Vr, P.Vi, ldv, true, true, TVr, TVi, ldv, n, n, n, 1.0, nullptr, nullptr, false, 0.0,
                  &pl.tv);
  }
  if (use_side) XMCA_HIP(hipEventRecord(vw.ev_join, vw.side));
  vw.joined = !use_side;
  vw.prepared = true;
}

// Eigenvectors of the matrix reduced by trd_reduce(..., keep_reflectors = true): Zr / Zi (n x ldz planes) get row i =
// conj(u_i) for the eigenvalues in DESCENDING order (the layout of hermitian_evd).  Returns false when the vectors of the
// tridiagonal are too far from orthonormal to be repaired (clusters) - the caller then uses the Jacobi solver.
inline bool trd_eigenvectors(hipStream_t st, TrdWorkspace& ws, TrdVecWorkspace& vw, GemmWorkspace& gws, const TrdParams& P, bool cplx,
                             double* Zr, double* Zi, int64_t ldz) {
  const int n = P.n;
  const int64_t ldv = P.ld;                                   // reflectors: row j = v_j, zero outside its support j+1 .. n-1
  constexpr int SBK = TRD_SBK;
  const int64_t ld = ((int64_t)n + 15) & ~(int64_t)15;
  const size_t plane = (size_t)n * ld;
  double* Yr = vw.Y[0].ensure(plane);
  double* Yi = cplx ? vw.Y[1].ensure(plane) : nullptr;
  double* Wr = vw.Wk[0].ensure(plane);
  double* Wi = cplx ? vw.Wk[1].ensure(plane) : nullptr;
  hipLaunchKernelGGL(trd_twisted_kernel, dim3(ceil_div(n, 64)), dim3(128), 0, st, P.d, P.e, n, vw.lam_asc.get(), Wr, Yr, ld);
  XMCA_HIP(hipGetLastError());
  if (cplx) XMCA_HIP(hipMemsetAsync(Yi, 0, sizeof(double) * plane, st));
  const int nref = n - 1;
  if (nref > 0) {
    XMCA_CHECK(vw.prepared, XMCA_ERR_STATE, "trd_eigenvectors: trd_wy_prepare has not run");
    vw.prepared = false;
    if (!vw.joined) XMCA_HIP(hipStreamWaitEvent(st, vw.ev_join, 0));      // T V^H of every super-block is ready (second stream)
    vw.joined = true;
    const double* TVr = vw.TV[0].get();
    const double* TVi = cplx ? vw.TV[1].get() : nullptr;
    // ---- Z = H_0 H_1 ... H_{n-2} Yt, super-blocks from the last to the first:  X = (T V^H) Z,  Z -= V X ----
    double* sm = vw.small.ensure(2 * (size_t)SBK * ld);
    double* Xr = sm; double* Xi = cplx ? Xr + (size_t)SBK * ld : nullptr;
    for (int j0 = ((nref - 1) / SBK) * SBK; j0 >= 0; j0 -= SBK) {
      const int nb = std::min(SBK, nref - j0);
      const int i0 = j0 + 1, mb = n - i0;                    // support of the super-block: rows i0 .. n-1
      const int64_t vo = (int64_t)j0 * ldv + i0;             // Vs[r][i - i0], r < nb  (row-major, contraction index fast)
      double* Zr0 = Yr + (int64_t)i0 * ld;
      double* Zi0 = cplx ? Yi + (int64_t)i0 * ld : nullptr;
      cgemm<double>(st, gws, TVr + vo, cplx ? TVi + vo : nullptr, ldv, true, false, Zr0, Zi0, ld, true, false, Xr, Xi, ld, nb, n, mb, 1.0, nullptr,
                    nullptr, false);
      // Z[i0:, :] -= V X      (A(m = i, k = r) = Vs[r][i]: the k-slow orientation)
      cgemm<double>(st, gws, P.Vr + vo, cplx ? P.Vi + vo : nullptr, ldv, false, false, Xr, Xi, ld, true, false, Zr0, Zi0, ld, mb, n, nb, -1.0,
                    nullptr, nullptr, false, 1.0);
    }
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
    // The common case - max |S - I| <= 1e-6, one more step leaves its square - is queued BEFORE the host learns the number:
    // out = (3/2 I - 1/2 S)[rows reversed] Z^H :  out[kk][i] = conj(u_{n-1-kk}[i]), with B(k, n = i) = conj(Z[i][k]), the
    // vectors as they lie (the round-4 GEMM runs every orientation at the same rate; rounds 2-3 went through a transposed
    // copy).  One host round trip per call instead of two; a spectrum that needs a second step (or the Jacobi sweeps) pays
    // for a product whose result is overwritten.
    hipLaunchKernelGGL(trd_ns_matrix_kernel, dim3(n), dim3(256), 0, st, Sr, Si, n, ld, 1, Wr, Wi);
    cgemm<double>(st, gws, Wr, Wi, ld, true, false, Yr, Yi, ld, false, true, Zr, Zi, ldz, n, n, n, 1.0, nullptr, nullptr, false);
    XMCA_HIP(hipGetLastError());
    unsigned long long bits = 0;
    XMCA_HIP(hipMemcpyAsync(&bits, vw.orth.get(), sizeof(bits), hipMemcpyDeviceToHost, st));
    XMCA_HIP(hipStreamSynchronize(st));
    double off;
    std::memcpy(&off, &bits, sizeof(off));
    if (round == 0) vw.last_orth = off;
    if (!(off <= 0.3)) return false;                          // clusters (or NaN): not repairable by Newton-Schulz
    if (off <= 1e-6) {                                        // the step queued above was the last one
      ++vw.ns_steps;
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
