// Varimax / Promax on the device (f64 / complex-f64 throughout, see DESIGN.md "Varimax").
//
// Loadings live as planes in mode-major order:  A[j][n]  (p modes x N grid points, ld = N), so a wave reads
// 64 consecutive grid points of one mode per instruction.
//
// One Varimax iteration (xmca/tools/rotation.py:52-64) = two launches, no host round trip:
//   varimax_accum_kernel : a SINGLE pass over the loadings:  z = a_n R,  w = |z|^2 z - z c/n,  G += a_n^H w
//                          (column sums c_k = sum_n |z_nk|^2 are obtained without a pass over N as
//                          diag(R^H (A^H A) R)); per-workgroup partial G's.
//   varimax_step_kernel  : one workgroup: reduce partials, one-sided Jacobi SVD of the p x p matrix
//                          (16-lane groups, one column pair each), R = U V^H, d = sum(s), the reference's
//                          stopping rule |d - d_old| / d < tol, iteration counter, next c.
// After convergence (or NaN) both kernels turn into no-ops, so the host can enqueue iterations in batches
// and look at the state block only once per batch; the stop iteration is decided on the device.
#pragma once
#include "common.h"
#include "kernels.h"

namespace xmca {

constexpr int ROT_PMAX = 64;     // max number of rotated modes
constexpr int ROT_PB = 64;       // grid points per batch inside a workgroup
constexpr int ROT_LDP = ROT_PB + 1;

// state block (doubles): [0]=iter [1]=converged [2]=d [3]=d_old [4]=nan_flag [5]=svd_sweeps(last)
constexpr int ROT_STATE_N = 8;

// L[j][n] = V_side[j][n - off] * sqrt(sigma_j) for the concatenated sides; A = L / h, h = row norms over j
template <bool CPLX>
__global__ void rot_build_loadings_kernel(const double* __restrict__ Vlr, const double* __restrict__ Vli, int64_t ldl, int64_t Nl,
                                          const double* __restrict__ Vrr, const double* __restrict__ Vri, int64_t ldr, int64_t Nr,
                                          const double* __restrict__ sigma, int p, double* __restrict__ Ar, double* __restrict__ Ai,
                                          double* __restrict__ h) {
  const int64_t N = Nl + Nr;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    const bool left = n < Nl;
    const double* sr = left ? Vlr : Vrr;
    const double* si = left ? Vli : Vri;
    const int64_t ld = left ? ldl : ldr, c = left ? n : n - Nl;
    double acc = 0.0;
    for (int j = 0; j < p; ++j) {
      const double f = sqrt(sigma[j]);
      const double xr = sr[j * ld + c] * f;
      Ar[j * N + n] = xr;
      acc += xr * xr;
      if constexpr (CPLX) {
        const double xi = si[j * ld + c] * f;
        Ai[j * N + n] = xi;
        acc += xi * xi;
      }
    }
    const double hn = sqrt(acc);
    h[n] = hn;
    const double inv = 1.0 / hn;      // zero rows -> inf*0 = NaN, as in the reference (rotation.py:46-48)
    for (int j = 0; j < p; ++j) {
      Ar[j * N + n] *= inv;
      if constexpr (CPLX) Ai[j * N + n] *= inv;
    }
  }
}

// host-provided N x p row-major loadings (interleaved complex or real) -> planes A[j][n] / h, and h
template <bool CPLX>
__global__ void rot_import_loadings_kernel(const double* __restrict__ L, int64_t N, int p, double* __restrict__ Ar,
                                           double* __restrict__ Ai, double* __restrict__ h) {
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int j = 0; j < p; ++j) {
      if constexpr (CPLX) {
        const double xr = L[2 * (n * p + j)], xi = L[2 * (n * p + j) + 1];
        acc += xr * xr + xi * xi;
      } else {
        const double xr = L[n * p + j];
        acc += xr * xr;
      }
    }
    const double hn = sqrt(acc), inv = 1.0 / hn;
    h[n] = hn;
    for (int j = 0; j < p; ++j) {
      if constexpr (CPLX) {
        Ar[j * N + n] = L[2 * (n * p + j)] * inv;
        Ai[j * N + n] = L[2 * (n * p + j) + 1] * inv;
      } else {
        Ar[j * N + n] = L[n * p + j] * inv;
      }
    }
  }
}

// Generic single-pass accumulation over grid points (one p x p result per launch).
//   MODE 0 (Varimax step): Z = A R ;  W = |Z|^2 Z - Z c / N ;             out = A^H W
//   MODE 1 (Gram)        : out = A^H A
//   MODE 2 (Promax fit)  : B = h (A R); h2 = |B_n|; X = B / h2; Xn = X / colmax; Y = Xn |Xn|^(power-1)
//                          SEL 0: X^H X   SEL 1: X^H Y   SEL 2: sum_{n<Nleft} h2^2 x^H x   SEL 3: same over n>=Nleft
//   MODE 3 (column max)  : X as in MODE 2; colmax_k = max_n |X_nk|  (atomicMax on the bit pattern)
// partial results: part[wg * p * p + j * p + k]  (planes)
template <bool CPLX, int MODE, int SEL>
__device__ __forceinline__ void rot_accum_body(double* __restrict__ sm, const double* __restrict__ Ar, const double* __restrict__ Ai,
                                               const double* __restrict__ h, int64_t N, int64_t Nleft, int p,
                                               const double* __restrict__ Rr, const double* __restrict__ Ri,
                                               const double* __restrict__ cvec, const double* __restrict__ colmax, double power,
                                               double* __restrict__ part_r, double* __restrict__ part_i,
                                               unsigned long long* __restrict__ colmax_bits) {
  // layout: Xs[p][LDP], Ys[p][LDP], Rs[p][p]  (x2 planes when complex), wgt[PB]
  const int pl = p * ROT_LDP;
  double* Xr = sm;
  double* Yr = Xr + pl;
  double* Rsr = Yr + pl;
  double* wgt = Rsr + p * p;
  double* Xi = wgt + ROT_PB;
  double* Yi = Xi + pl;
  double* Rsi = Yi + pl;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (MODE != 1) {
    for (int e = tid; e < p * p; e += 256) {
      Rsr[e] = Rr[e];
      if constexpr (CPLX) Rsi[e] = Ri[e];
    }
  }
  constexpr int MAXE = (ROT_PMAX * ROT_PMAX) / 256;   // (j,k) entries per thread
  double accr[MAXE], acci[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) { accr[e] = 0.0; acci[e] = 0.0; }

  const int64_t nbatch = (N + ROT_PB - 1) / ROT_PB;
  for (int64_t bt = blockIdx.x; bt < nbatch; bt += gridDim.x) {
    const int64_t n0 = bt * ROT_PB;
    __syncthreads();
    // stage the A tile into Ys (coalesced over grid points)
    for (int e = tid; e < p * ROT_PB; e += 256) {
      const int j = e / ROT_PB, pt = e % ROT_PB;
      const int64_t n = n0 + pt;
      double vr = 0.0, vi = 0.0;
      if (n < N) {
        vr = Ar[(int64_t)j * N + n];
        if constexpr (CPLX) vi = Ai[(int64_t)j * N + n];
      }
      Yr[j * ROT_LDP + pt] = vr;
      if constexpr (CPLX) Yi[j * ROT_LDP + pt] = vi;
    }
    __syncthreads();
    if (MODE == 1) {
      for (int e = tid; e < p * ROT_PB; e += 256) {
        const int j = e / ROT_PB, pt = e % ROT_PB;
        Xr[j * ROT_LDP + pt] = Yr[j * ROT_LDP + pt];
        if constexpr (CPLX) Xi[j * ROT_LDP + pt] = Yi[j * ROT_LDP + pt];
      }
      __syncthreads();
    } else {
      // z_k(pt) = sum_j a_j(pt) R[j][k]: wave handles k = wave, wave+4, ...; lanes = grid points (R reads broadcast)
      for (int k = wave; k < p; k += 4) {
        double zr = 0.0, zi = 0.0;
        for (int j = 0; j < p; ++j) {
          const double ar = Yr[j * ROT_LDP + lane], rr = Rsr[j * p + k];
          zr += ar * rr;
          if constexpr (CPLX) {
            const double ai = Yi[j * ROT_LDP + lane], ri = Rsi[j * p + k];
            zr -= ai * ri;
            zi += ar * ri + ai * rr;
          }
        }
        if (MODE == 0) {
          // W = (|z|^2 - c_k / N) z     (rotation.py:56-57)
          const double f = zr * zr + zi * zi - cvec[k] / (double)N;
          zr *= f;
          zi *= f;
        }
        Xr[k * ROT_LDP + lane] = zr;
        if constexpr (CPLX) Xi[k * ROT_LDP + lane] = zi;
      }
      __syncthreads();
      if (MODE >= 2) {
        // B = h z ; h2 = |B_n| ; X = B / h2      (rotation.py:74-77, :115-117)
        if (tid < ROT_PB) {
          const int64_t n = n0 + tid;
          const double hn = n < N ? h[n] : 0.0;
          double acc = 0.0;
          for (int k = 0; k < p; ++k) {
            const double br = hn * Xr[k * ROT_LDP + tid];
            acc += br * br;
            if constexpr (CPLX) {
              const double bi = hn * Xi[k * ROT_LDP + tid];
              acc += bi * bi;
            }
          }
          const double h2 = sqrt(acc);
          const double sc = (n < N) ? hn * (1.0 / h2) : 0.0;   // NaN for a zero row, like the reference
          for (int k = 0; k < p; ++k) {
            Xr[k * ROT_LDP + tid] *= sc;
            if constexpr (CPLX) Xi[k * ROT_LDP + tid] *= sc;
          }
          double w = 1.0;
          if (SEL == 2) w = (n < Nleft) ? h2 * h2 : 0.0;
          if (SEL == 3) w = (n < N && n >= Nleft) ? h2 * h2 : 0.0;
          wgt[tid] = w;
        }
        __syncthreads();
        if (MODE == 3) {
          for (int k = wave; k < p; k += 4) {
            const double xr = Xr[k * ROT_LDP + lane];
            double a2 = xr * xr;
            if constexpr (CPLX) { const double xi = Xi[k * ROT_LDP + lane]; a2 += xi * xi; }
            double a = (n0 + lane < N) ? sqrt(a2) : 0.0;
            for (int o = 32; o > 0; o >>= 1) a = fmax(a, __shfl_xor(a, o));
            if (lane == 0 && a > 0.0) atomicMax(&colmax_bits[k], (unsigned long long)__double_as_longlong(a));
          }
          continue;
        }
        if (SEL == 1) {
          // Y = Xn |Xn|^(power-1), Xn = X / colmax_k      (rotation.py:121-124)
          for (int k = wave; k < p; k += 4) {
            const double inv = 1.0 / colmax[k];
            const double xr = Xr[k * ROT_LDP + lane] * inv;
            double xi = 0.0;
            if constexpr (CPLX) xi = Xi[k * ROT_LDP + lane] * inv;
            const double f = pow(sqrt(xr * xr + xi * xi), power - 1.0);
            Yr[k * ROT_LDP + lane] = xr * f;
            if constexpr (CPLX) Yi[k * ROT_LDP + lane] = xi * f;
          }
          __syncthreads();
        }
      }
    }
    // out[j][k] += sum_pt w(pt) conj(L_j(pt)) * Q_k(pt)
    //   MODE 0: L = A (Ys), Q = W (Xs) ; MODE 1: L = Q = A ; MODE 2: L = X, Q = X (SEL 0,2,3) or Y (SEL 1)
    const double* Lr = (MODE == 0) ? Yr : Xr;
    const double* Li = (MODE == 0) ? Yi : Xi;
    const double* Qr = (MODE == 2 && SEL == 1) ? Yr : Xr;
    const double* Qi = (MODE == 2 && SEL == 1) ? Yi : Xi;
#pragma unroll
    for (int sl = 0; sl < MAXE; ++sl) {
      const int e = tid + 256 * sl;
      if (e < p * p) {
        const int j = e / p, k = e % p;
        double sr = 0.0, si = 0.0;
        for (int pt = 0; pt < ROT_PB; ++pt) {
          const double lr = Lr[j * ROT_LDP + pt], qr = Qr[k * ROT_LDP + pt];
          double pr = lr * qr, pi = 0.0;
          if constexpr (CPLX) {
            const double li = Li[j * ROT_LDP + pt], qi = Qi[k * ROT_LDP + pt];
            pr += li * qi;               // conj(l) q
            pi = lr * qi - li * qr;
          }
          if (MODE == 2 && SEL >= 2) { pr *= wgt[pt]; pi *= wgt[pt]; }
          sr += pr;
          si += pi;
        }
        accr[sl] += sr;
        acci[sl] += si;
      }
    }
  }
  if (MODE == 3) return;
#pragma unroll
  for (int sl = 0; sl < MAXE; ++sl) {
    const int e = tid + 256 * sl;
    if (e < p * p) {
      const int64_t idx = (int64_t)blockIdx.x * p * p + e;
      part_r[idx] = accr[sl];
      if constexpr (CPLX) part_i[idx] = acci[sl];
    }
  }
}

template <bool CPLX, int MODE, int SEL>
__global__ __launch_bounds__(256) void rot_accum_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai,
                                                        const double* __restrict__ h, int64_t N, int64_t Nleft, int p,
                                                        const double* __restrict__ Rr, const double* __restrict__ Ri,
                                                        const double* __restrict__ cvec, const double* __restrict__ colmax,
                                                        double power, const double* __restrict__ state,
                                                        double* __restrict__ part_r, double* __restrict__ part_i,
                                                        unsigned long long* __restrict__ colmax_bits) {
  if (MODE == 0 && (state[1] != 0.0 || state[4] != 0.0)) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  rot_accum_body<CPLX, MODE, SEL>(reinterpret_cast<double*>(smem_raw), Ar, Ai, h, N, Nleft, p, Rr, Ri, cvec, colmax, power, part_r,
                                  part_i, colmax_bits);
}

static inline size_t rot_accum_smem(int p, bool cplx) {
  const size_t plane = 2 * (size_t)p * ROT_LDP + (size_t)p * p;
  return sizeof(double) * ((cplx ? 2 : 1) * plane + ROT_PB);
}
static inline int rot_max_modes(bool cplx) { return cplx ? 48 : ROT_PMAX; }   // LDS budget of rot_accum_kernel

// out[e] = sum_wg part[wg*pp + e]
__global__ void rot_reduce_partials_kernel(const double* __restrict__ part_r, const double* __restrict__ part_i, int nwg, int pp,
                                           double* __restrict__ out_r, double* __restrict__ out_i) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= pp) return;
  double sr = 0.0, si = 0.0;
  for (int w = 0; w < nwg; ++w) {
    sr += part_r[(int64_t)w * pp + e];
    if (part_i) si += part_i[(int64_t)w * pp + e];
  }
  out_r[e] = sr;
  if (out_i) out_i[e] = si;
}

// c_k = real( sum_{j,l} conj(R[j][k]) A0[j][l] R[l][k] )
template <bool CPLX>
__device__ void rot_colsums(const double* Rr, const double* Ri, const double* A0r, const double* A0i, int p, double* c, int tid,
                            int nthreads) {
  for (int k = tid; k < p; k += nthreads) {
    double acc = 0.0;
    for (int j = 0; j < p; ++j) {
      // t = sum_l A0[j][l] R[l][k]
      double tr = 0.0, ti = 0.0;
      for (int l = 0; l < p; ++l) {
        const double ar = A0r[j * p + l], rr = Rr[l * p + k];
        tr += ar * rr;
        if (CPLX) {
          const double ai = A0i[j * p + l], ri = Ri[l * p + k];
          tr -= ai * ri;
          ti += ar * ri + ai * rr;
        }
      }
      // real(conj(R[j][k]) * t)
      acc += Rr[j * p + k] * tr + (CPLX ? Ri[j * p + k] * ti : 0.0);
    }
    c[k] = acc;
  }
}

// One workgroup (256 threads): G = sum partials ; SVD by one-sided Jacobi ; R = U V^H ; d ; stop rule ; next c.
template <bool CPLX>
__global__ __launch_bounds__(256) void varimax_step_kernel(const double* __restrict__ part_r, const double* __restrict__ part_i,
                                                           int nwg, int p, const double* __restrict__ A0r,
                                                           const double* __restrict__ A0i, double* __restrict__ Rr,
                                                           double* __restrict__ Ri, double* __restrict__ Wr,
                                                           double* __restrict__ Wi, double* __restrict__ cvec,
                                                           double* __restrict__ state, double tol, int init_only) {
  constexpr int LD = ROT_PMAX + 1;
  __shared__ double Gr[ROT_PMAX][LD], Gi[CPLX ? ROT_PMAX : 1][CPLX ? LD : 1];
  __shared__ double Vr[ROT_PMAX][LD], Vi[CPLX ? ROT_PMAX : 1][CPLX ? LD : 1];
  __shared__ double sig[ROT_PMAX];
  __shared__ double red2r[256], red2i[CPLX ? 256 : 1];
  __shared__ int flag;
  const int tid = threadIdx.x;
  if (init_only) {
    // R = I, c from A0, state reset
    for (int e = tid; e < p * p; e += 256) {
      Rr[e] = Wr[e] = (e / p == e % p) ? 1.0 : 0.0;
      if (CPLX) Ri[e] = Wi[e] = 0.0;
    }
    __syncthreads();
    __threadfence_block();
    rot_colsums<CPLX>(Rr, Ri, A0r, A0i, p, cvec, tid, 256);
    if (tid < ROT_STATE_N) state[tid] = 0.0;
    return;
  }
  if (state[1] != 0.0 || state[4] != 0.0) return;

  // G = sum of the per-workgroup partials: 256 / p^2 thread slices per entry, combined in a fixed order
  const int pp = p * p;
  const int nsl = pp <= 128 ? 256 / pp : 1;
  if (nsl > 1) {
    const int sl = tid / pp, e = tid % pp;
    if (sl < nsl) {
      double sr = 0.0, si = 0.0;
      for (int w = sl; w < nwg; w += nsl) {
        sr += part_r[(int64_t)w * pp + e];
        if (CPLX) si += part_i[(int64_t)w * pp + e];
      }
      red2r[tid] = sr;
      if (CPLX) red2i[tid] = si;
    }
    __syncthreads();
  }
  for (int e = tid; e < p * p; e += 256) {
    double sr = 0.0, si = 0.0;
    if (nsl > 1) {
      for (int sl = 0; sl < nsl; ++sl) {
        sr += red2r[sl * pp + e];
        if (CPLX) si += red2i[sl * pp + e];
      }
    } else {
      for (int w = 0; w < nwg; ++w) {
        sr += part_r[(int64_t)w * pp + e];
        if (CPLX) si += part_i[(int64_t)w * pp + e];
      }
    }
    const int j = e / p, k = e % p;
    // warm start: V <- right singular vectors of the previous iteration (G changes slowly, so G V_prev is already
    // nearly column-orthogonal and the Jacobi sweeps below converge in one or two passes); G is staged in V's
    // place for the product and swapped in afterwards.
    Vr[j][k] = sr;
    if constexpr (CPLX) Vi[j][k] = si;
  }
  __syncthreads();
  for (int e = tid; e < p * p; e += 256) {
    const int j = e / p, k = e % p;
    double ar = 0.0, ai = 0.0;
    for (int m = 0; m < p; ++m) {
      const double gr = Vr[j][m], wr = Wr[m * p + k];
      ar += gr * wr;
      if constexpr (CPLX) {
        const double gi = Vi[j][m], wi = Wi[m * p + k];
        ar -= gi * wi;
        ai += gr * wi + gi * wr;
      }
    }
    Gr[j][k] = ar;
    if constexpr (CPLX) Gi[j][k] = ai;
  }
  __syncthreads();
  for (int e = tid; e < p * p; e += 256) {
    const int j = e / p, k = e % p;
    Vr[j][k] = Wr[e];
    if constexpr (CPLX) Vi[j][k] = Wi[e];
  }
  __syncthreads();

  // one-sided Jacobi: orthogonalise the columns of G (G V = U S).  16-lane groups own one column pair each.
  const int pe = (p + 1) & ~1;             // even number of players (last one is a dummy when p is odd)
  const int npairs = pe / 2;
  const int grp = tid >> 4, gl = tid & 15;
  int sweeps = 0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    if (tid == 0) flag = 0;
    __syncthreads();
    for (int step = 0; step < pe - 1; ++step) {
      for (int pr = grp; pr < npairs; pr += 16) {
        int a, b;
        if (pr == 0) { a = pe - 1; b = step; }
        else { a = (step + pr) % (pe - 1); b = (step - pr + (pe - 1)) % (pe - 1); }
        const int ca = min(a, b), cb = max(a, b);
        if (cb >= p) continue;   // dummy player
        double al = 0.0, be = 0.0, gr = 0.0, gi = 0.0;
        for (int i = gl; i < p; i += 16) {
          const double xr = Gr[i][ca], yr = Gr[i][cb];
          al += xr * xr; be += yr * yr; gr += xr * yr;
          if constexpr (CPLX) {
            const double xi = Gi[i][ca], yi = Gi[i][cb];
            al += xi * xi; be += yi * yi;
            gr += xi * yi;                 // conj(x) y = (xr - i xi)(yr + i yi)
            gi += xr * yi - xi * yr;
          }
        }
        for (int o = 8; o > 0; o >>= 1) {
          al += __shfl_xor(al, o); be += __shfl_xor(be, o); gr += __shfl_xor(gr, o); gi += __shfl_xor(gi, o);
        }
        const double g2 = gr * gr + gi * gi;
        if (g2 > 0.0 && g2 > 1e-29 * al * be) {
          // same rotation as in jacobi.h: t = sign(d) 2|g| / (|d| + sqrt(d^2 + 4|g|^2)), one sqrt + one reciprocal
          const double dd = be - al;
          const double w = (dd >= 0.0 ? 2.0 : -2.0) / (fabs(dd) + sqrt(dd * dd + 4.0 * g2));
          const double c = 1.0 / sqrt(1.0 + w * w * g2);
          const double sr = w * c * gr, si = w * c * gi;
          if (gl == 0) flag = 1;
          for (int i = gl; i < p; i += 16) {
            // new_a = c x - conj(sg) y ; new_b = sg x + c y
            {
              const double xr = Gr[i][ca], yr = Gr[i][cb];
              if constexpr (!CPLX) {
                Gr[i][ca] = c * xr - sr * yr; Gr[i][cb] = sr * xr + c * yr;
              } else {
                const double xi = Gi[i][ca], yi = Gi[i][cb];
                Gr[i][ca] = c * xr - (sr * yr + si * yi); Gi[i][ca] = c * xi - (sr * yi - si * yr);
                Gr[i][cb] = (sr * xr - si * xi) + c * yr; Gi[i][cb] = (sr * xi + si * xr) + c * yi;
              }
            }
            {
              const double xr = Vr[i][ca], yr = Vr[i][cb];
              if constexpr (!CPLX) {
                Vr[i][ca] = c * xr - sr * yr; Vr[i][cb] = sr * xr + c * yr;
              } else {
                const double xi = Vi[i][ca], yi = Vi[i][cb];
                Vr[i][ca] = c * xr - (sr * yr + si * yi); Vi[i][ca] = c * xi - (sr * yi - si * yr);
                Vr[i][cb] = (sr * xr - si * xi) + c * yr; Vi[i][cb] = (sr * xi + si * xr) + c * yi;
              }
            }
          }
        }
      }
      __syncthreads();
    }
    ++sweeps;
    const int f = flag;
    __syncthreads();
    if (!f) break;
  }

  // singular values = column norms; U = G / s
  for (int k = tid; k < p; k += 256) {
    double acc = 0.0;
    for (int i = 0; i < p; ++i) {
      acc += Gr[i][k] * Gr[i][k];
      if constexpr (CPLX) acc += Gi[i][k] * Gi[i][k];
    }
    sig[k] = sqrt(acc);
  }
  __syncthreads();
  // R[j][k] = sum_m U[j][m] conj(V[k][m]) = sum_m G[j][m]/s_m * conj(V[k][m])
  double rr_loc[ROT_PMAX * ROT_PMAX / 256], ri_loc[ROT_PMAX * ROT_PMAX / 256];
  {
    int slot = 0;
    for (int e = tid; e < p * p; e += 256, ++slot) {
      const int j = e / p, k = e % p;
      double rr = 0.0, ri = 0.0;
      for (int m = 0; m < p; ++m) {
        const double inv = 1.0 / sig[m];
        const double ur = Gr[j][m] * inv, vr = Vr[k][m];
        rr += ur * vr;
        if constexpr (CPLX) {
          const double ui = Gi[j][m] * inv, vi = Vi[k][m];
          rr += ui * vi;              // (ur + i ui)(vr - i vi)
          ri += ui * vr - ur * vi;
        }
      }
      Rr[e] = rr;
      Wr[e] = Vr[j][k];
      if (CPLX) { Ri[e] = ri; Wi[e] = Vi[j][k]; }
#pragma unroll
      for (int sl = 0; sl < ROT_PMAX * ROT_PMAX / 256; ++sl)
        if (sl == slot) { rr_loc[sl] = rr; ri_loc[sl] = ri; }
    }
  }
  __syncthreads();
  // next column sums c_k = Re sum_j conj(R[j][k]) (A0 R)[j][k]: stage R in G's place and A0 in V's place
  {
    int slot = 0;
    for (int e = tid; e < p * p; e += 256, ++slot) {
      const int j = e / p, k = e % p;
#pragma unroll
      for (int sl = 0; sl < ROT_PMAX * ROT_PMAX / 256; ++sl)
        if (sl == slot) { Gr[j][k] = rr_loc[sl]; if constexpr (CPLX) Gi[j][k] = ri_loc[sl]; }
      Vr[j][k] = A0r[e];
      if constexpr (CPLX) Vi[j][k] = A0i[e];
    }
  }
  __syncthreads();
  {
    int slot = 0;
    for (int e = tid; e < p * p; e += 256, ++slot) {
      const int j = e / p, k = e % p;
      double tr = 0.0, ti = 0.0;
      for (int l = 0; l < p; ++l) {
        const double ar = Vr[j][l], r_ = Gr[l][k];
        tr += ar * r_;
        if constexpr (CPLX) {
          const double ai = Vi[j][l], ri_ = Gi[l][k];
          tr -= ai * ri_;
          ti += ar * ri_ + ai * r_;
        }
      }
      double prod = Gr[j][k] * tr;
      if constexpr (CPLX) prod += Gi[j][k] * ti;
#pragma unroll
      for (int sl = 0; sl < ROT_PMAX * ROT_PMAX / 256; ++sl)
        if (sl == slot) rr_loc[sl] = prod;
    }
  }
  __syncthreads();
  {
    int slot = 0;
    for (int e = tid; e < p * p; e += 256, ++slot) {
#pragma unroll
      for (int sl = 0; sl < ROT_PMAX * ROT_PMAX / 256; ++sl)
        if (sl == slot) Vr[e / p][e % p] = rr_loc[sl];
    }
  }
  __syncthreads();
  for (int k = tid; k < p; k += 256) {
    double acc = 0.0;
    for (int j = 0; j < p; ++j) acc += Vr[j][k];
    cvec[k] = acc;
  }
  if (tid == 0) {
    double d = 0.0;
    for (int k = 0; k < p; ++k) d += sig[k];
    const double d_old = state[2];
    state[3] = d_old;
    state[2] = d;
    state[0] += 1.0;
    state[5] = (double)sweeps;
    if (!(d == d)) state[4] = 1.0;                          // NaN (e.g. a zero row in the loadings)
    else if (fabs(d - d_old) / d < tol) state[1] = 1.0;     // rotation.py:62
  }
}

// ---------------------------------------------------------------------------------------------------------------
// One whole Varimax iteration in ONE launch: every workgroup accumulates its partial G (rot_accum_body<0>), the
// workgroup that arrives last (agent-scope release / ticket / acquire, cdna_hip_programming.md G16) reduces the
// partials in a fixed order and finishes the step.  R = U V^H is the unitary polar factor of G, obtained by the
// Newton-Schulz iteration X <- X (1.5 I - 0.5 X^H X) from X0 = G / ||G||_F (only p x p products; quadratic
// convergence; 11-20 iterations for cond(G) <= 1e4), and d = sum(s) = Re tr(R^H G) - first-order insensitive to
// errors in R because R^H dR is skew-Hermitian.  Same R and d as rotation.py:59-61 to rounding, no SVD.
// ---------------------------------------------------------------------------------------------------------------
static inline size_t rot_polar_smem(int p, bool cplx) { return sizeof(double) * ((cplx ? 2 : 1) * 4 * (size_t)p * p + 1024); }

template <bool CPLX>
__device__ void varimax_polar_step(double* __restrict__ sm, const double* __restrict__ part_r, const double* __restrict__ part_i,
                                   int nwg, int p, const double* __restrict__ A0r, const double* __restrict__ A0i,
                                   double* __restrict__ Rr, double* __restrict__ Ri, double* __restrict__ cvec,
                                   double* __restrict__ state, double tol) {
  const int tid = threadIdx.x, pp = p * p;
  double* Gr = sm;            // G, later A0 R
  double* Xr = Gr + pp;
  double* Tr = Xr + pp;
  double* Yr = Tr + pp;
  double* scr = Yr + pp;      // 1024 doubles of scratch
  double* Gi = scr + 1024;
  double* Xi = Gi + pp;
  double* Ti = Xi + pp;
  double* Yi = Ti + pp;
  auto block_reduce = [&](double v, bool take_max) -> double {
    for (int o = 32; o > 0; o >>= 1) {
      const double w = __shfl_xor(v, o);
      v = take_max ? fmax(v, w) : v + w;
    }
    __syncthreads();
    if ((tid & 63) == 0) scr[1000 + (tid >> 6)] = v;
    __syncthreads();
    const double a = scr[1000], b = scr[1001], c = scr[1002], d = scr[1003];
    return take_max ? fmax(fmax(a, b), fmax(c, d)) : (a + b) + (c + d);
  };

  // G = sum of partials (fixed order): pp <= 128 uses 256 / pp thread slices per entry
  const int nsl = pp <= 128 ? 256 / pp : 1;
  if (nsl > 1) {
    const int sl = tid / pp, e = tid % pp;
    if (sl < nsl) {
      // 8 loads in flight per thread (the partials sit in L2; a dependent chain would cost ~1 us per load)
      double ar[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ai[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int w = sl; w < nwg; w += 8 * nsl) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ww = w + u * nsl;
          if (ww < nwg) {
            ar[u] += part_r[(int64_t)ww * pp + e];
            if constexpr (CPLX) ai[u] += part_i[(int64_t)ww * pp + e];
          }
        }
      }
      scr[tid] = ((ar[0] + ar[1]) + (ar[2] + ar[3])) + ((ar[4] + ar[5]) + (ar[6] + ar[7]));
      if constexpr (CPLX) scr[256 + tid] = ((ai[0] + ai[1]) + (ai[2] + ai[3])) + ((ai[4] + ai[5]) + (ai[6] + ai[7]));
    }
    __syncthreads();
  }
  double fro2 = 0.0;
  for (int e = tid; e < pp; e += 256) {
    double sr = 0.0, si = 0.0;
    if (nsl > 1) {
      for (int sl = 0; sl < nsl; ++sl) {
        sr += scr[sl * pp + e];
        if constexpr (CPLX) si += scr[256 + sl * pp + e];
      }
    } else {
      double ar[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ai[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int w = 0; w < nwg; w += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (w + u < nwg) {
            ar[u] += part_r[(int64_t)(w + u) * pp + e];
            if constexpr (CPLX) ai[u] += part_i[(int64_t)(w + u) * pp + e];
          }
        }
      }
      sr = ((ar[0] + ar[1]) + (ar[2] + ar[3])) + ((ar[4] + ar[5]) + (ar[6] + ar[7]));
      si = ((ai[0] + ai[1]) + (ai[2] + ai[3])) + ((ai[4] + ai[5]) + (ai[6] + ai[7]));
    }
    Gr[e] = sr;
    fro2 += sr * sr;
    if constexpr (CPLX) { Gi[e] = si; fro2 += si * si; }
  }
  fro2 = block_reduce(fro2, false);
  const double inv = 1.0 / sqrt(fro2);
  for (int e = tid; e < pp; e += 256) {
    Xr[e] = Gr[e] * inv;
    if constexpr (CPLX) Xi[e] = Gi[e] * inv;
  }
  __syncthreads();

  // Newton-Schulz with two barriers per iteration: X and Y alternate roles (no copy), the convergence measure of
  // iteration k is reduced per wave and read by everybody after the barrier that also publishes T.
  int it = 0;
  bool ok = false;
  double* Cr = Xr;   // current iterate
  double* Ci = Xi;
  double* Nr = Yr;   // next iterate
  double* Ni = Yi;
  for (; it < 100; ++it) {
    double err = 0.0;
    for (int e = tid; e < pp; e += 256) {
      const int j = e / p, k = e % p;
      double tr = 0.0, ti = 0.0;
#pragma unroll 5
      for (int m = 0; m < p; ++m) {
        const double ar = Cr[m * p + j], br = Cr[m * p + k];
        tr += ar * br;
        if constexpr (CPLX) {
          const double ai = Ci[m * p + j], bi = Ci[m * p + k];
          tr += ai * bi;                 // conj(a) b
          ti += ar * bi - ai * br;
        }
      }
      Tr[e] = tr;
      if constexpr (CPLX) Ti[e] = ti;
      err = fmax(err, fmax(fabs(tr - (j == k ? 1.0 : 0.0)), fabs(ti)));
      if (!(tr == tr)) err = HUGE_VAL;
    }
    for (int o = 32; o > 0; o >>= 1) err = fmax(err, __shfl_xor(err, o));
    if ((tid & 63) == 0) scr[1000 + 4 * (it & 1) + (tid >> 6)] = err;
    __syncthreads();
    {
      const double* e4 = scr + 1000 + 4 * (it & 1);
      err = fmax(fmax(e4[0], e4[1]), fmax(e4[2], e4[3]));
    }
    if (!(err < HUGE_VAL)) break;        // NaN / inf
    if (err < 1e-14) { ok = true; break; }
    for (int e = tid; e < pp; e += 256) {
      const int j = e / p, k = e % p;
      double yr = 0.0, yi = 0.0;
#pragma unroll 5
      for (int m = 0; m < p; ++m) {
        const double xr = Cr[j * p + m], t_r = Tr[m * p + k];
        yr += xr * t_r;
        if constexpr (CPLX) {
          const double xi = Ci[j * p + m], t_i = Ti[m * p + k];
          yr -= xi * t_i;
          yi += xr * t_i + xi * t_r;
        }
      }
      Nr[e] = 1.5 * Cr[e] - 0.5 * yr;
      if constexpr (CPLX) Ni[e] = 1.5 * Ci[e] - 0.5 * yi;
    }
    __syncthreads();
    { double* t = Cr; Cr = Nr; Nr = t; }
    { double* t = Ci; Ci = Ni; Ni = t; }
  }
  if (Cr != Xr) {   // the tail below expects the converged factor in X
    for (int e = tid; e < pp; e += 256) {
      Xr[e] = Cr[e];
      if constexpr (CPLX) Xi[e] = Ci[e];
    }
  }
  __syncthreads();

  // d = Re tr(R^H G);  R -> global;  T <- A0 (staging);  c_k = Re sum_j conj(R[j][k]) (A0 R)[j][k]
  double dsum = 0.0;
  for (int e = tid; e < pp; e += 256) {
    dsum += Xr[e] * Gr[e];
    if constexpr (CPLX) dsum += Xi[e] * Gi[e];
    Rr[e] = Xr[e];
    Tr[e] = A0r[e];
    if constexpr (CPLX) { Ri[e] = Xi[e]; Ti[e] = A0i[e]; }
  }
  dsum = block_reduce(dsum, false);
  for (int e = tid; e < pp; e += 256) {
    const int j = e / p, k = e % p;
    double tr = 0.0, ti = 0.0;
#pragma unroll 5
    for (int l = 0; l < p; ++l) {
      const double ar = Tr[j * p + l], rr = Xr[l * p + k];
      tr += ar * rr;
      if constexpr (CPLX) {
        const double ai = Ti[j * p + l], ri = Xi[l * p + k];
        tr -= ai * ri;
        ti += ar * ri + ai * rr;
      }
    }
    double prod = Xr[e] * tr;
    if constexpr (CPLX) prod += Xi[e] * ti;
    Yr[e] = prod;
  }
  __syncthreads();
  for (int k = tid; k < p; k += 256) {
    double acc = 0.0;
    for (int j = 0; j < p; ++j) acc += Yr[j * p + k];
    cvec[k] = acc;
  }
  if (tid == 0) {
    const double d_old = state[2];
    state[3] = d_old;
    state[2] = dsum;
    state[0] += 1.0;
    state[5] = (double)it;
    if (!ok || !(dsum == dsum)) state[4] = 1.0;                  // NaN / singular G
    else if (fabs(dsum - d_old) / dsum < tol) state[1] = 1.0;    // rotation.py:62
  }
}

template <bool CPLX>
__global__ __launch_bounds__(256) void varimax_iter_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai,
                                                           const double* __restrict__ h, int64_t N, int p,
                                                           const double* __restrict__ A0r, const double* __restrict__ A0i,
                                                           double* Rr, double* Ri, double* cvec, double* state, double* part_r,
                                                           double* part_i, unsigned int* counter, double tol) {
  if (state[1] != 0.0 || state[4] != 0.0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  rot_accum_body<CPLX, 0, 0>(sm, Ar, Ai, h, N, N, p, Rr, Ri, cvec, nullptr, 1.0, part_r, part_i, nullptr);
  // publish the partial, take a ticket (release before, acquire after: placement independent)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == gridDim.x - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    is_last = last;
  }
  __syncthreads();
  if (!is_last) return;
  varimax_polar_step<CPLX>(sm, part_r, part_i, (int)gridDim.x, p, A0r, A0i, Rr, Ri, cvec, state, tol);
}

// B[n][k] = scale_n * sum_j a_j(n) M[j][k]   ->  N x p row-major (interleaved complex) for the host
template <bool CPLX>
__global__ void rot_apply_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai, const double* __restrict__ h,
                                 int64_t N, int p, const double* __restrict__ Mr, const double* __restrict__ Mi,
                                 double* __restrict__ out) {
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    const double hn = h[n];
    for (int k = 0; k < p; ++k) {
      double zr = 0.0, zi = 0.0;
      for (int j = 0; j < p; ++j) {
        const double ar = Ar[(int64_t)j * N + n], mr = Mr[j * p + k];
        zr += ar * mr;
        if constexpr (CPLX) {
          const double ai = Ai[(int64_t)j * N + n], mi = Mi[j * p + k];
          zr -= ai * mi;
          zi += ar * mi + ai * mr;
        }
      }
      if constexpr (CPLX) {
        out[2 * (n * p + k)] = hn * zr;
        out[2 * (n * p + k) + 1] = hn * zi;
      } else {
        out[n * p + k] = hn * zr;
      }
    }
  }
}

}  // namespace xmca
