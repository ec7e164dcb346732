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
#include "gemm.h"
#include "kernels.h"

namespace xmca {

constexpr int ROT_PMAX = 64;     // max number of rotated modes
constexpr int ROT_PB = 64;       // grid points per batch inside a workgroup
constexpr int ROT_LDP = ROT_PB + 1;

// state block (doubles): [0]=iter [1]=converged [2]=d [3]=d_old [4]=nan_flag [5]=svd_sweeps(last)
constexpr int ROT_STATE_N = 8;

// the same for the float32-resident vectors of a real one-field float32 model: L = float(V) * float(sqrt(float(sigma))), a
// float32 product, exactly the values the reference's host code rotates (array.py:818-822 on float32 `_V` and
// `_singular_values`); everything after that is float64 like the uploaded-loadings path
__global__ void rot_build_loadings_f32_kernel(const float* __restrict__ V, int64_t ld, int64_t N, const double* __restrict__ sigma, int p,
                                              double* __restrict__ Ar, double* __restrict__ h) {
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int j = 0; j < p; ++j) {
      // (float32 square root, correctly rounded: the float64 root of a float32 number rounds to it - 53 > 2 x 24 + 2 bits;
      // HIP's __fsqrt_rn is the 1-ulp native instruction)
      const float f = (float)sqrt((double)(float)sigma[j]);
      const float xf = V[j * ld + n] * f;
      const double x = (double)xf;
      Ar[j * N + n] = x;
      acc += x * x;
    }
    const double hn = sqrt(acc);
    h[n] = hn;
    const double inv = 1.0 / hn;      // zero rows -> inf*0 = NaN, as in the reference (rotation.py:46-48)
    for (int j = 0; j < p; ++j) Ar[j * N + n] *= inv;
  }
}

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

// ---- element-wise / row kernels of the GEMM-based rotation path (any number of modes; Rotator::run_generic) ----------
// out[k] = sum_n |Zt[k][n]|^2  (or max_n |Zt[k][n]| with `take_max`); one workgroup per mode row
__global__ __launch_bounds__(256) void rot_row_reduce_kernel(const double* __restrict__ Zr, const double* __restrict__ Zi, int64_t N,
                                                             int take_max, double* __restrict__ out) {
  __shared__ double red[4];
  const int64_t k = blockIdx.x;
  double acc = 0.0;
  for (int64_t n = threadIdx.x; n < N; n += 256) {
    const double a = Zr[k * N + n], b = Zi ? Zi[k * N + n] : 0.0;
    const double v = a * a + b * b;
    acc = take_max ? fmax(acc, v) : acc + v;
  }
  if (take_max) {
    for (int o = 32; o > 0; o >>= 1) acc = fmax(acc, __shfl_xor(acc, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    acc = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (threadIdx.x == 0) out[k] = sqrt(acc);
  } else {
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[k] = red[0] + red[1] + red[2] + red[3];
  }
}
// Varimax: Z <- (|Z|^2 - gamma c_k / N) Z   (rotation.py:56-57), in place on the mode-major planes
__global__ void rot_w_kernel(double* __restrict__ Zr, double* __restrict__ Zi, int64_t N, int p, const double* __restrict__ c,
                             double gamma) {
  const int64_t tot = (int64_t)p * N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e / N);
    const double a = Zr[e], b = Zi ? Zi[e] : 0.0;
    const double f = a * a + b * b - gamma * (c[k] / (double)N);
    Zr[e] = f * a;
    if (Zi) Zi[e] = f * b;
  }
}
// Promax target (rotation.py:121-124): P = Xn |Xn|^(power - 1) with Xn = X / colmax, into (Pr, Pi)
__global__ void rot_target_kernel(const double* __restrict__ Xr, const double* __restrict__ Xi, int64_t N, int p,
                                  const double* __restrict__ colmax, double power, double* __restrict__ Pr, double* __restrict__ Pi) {
  const int64_t tot = (int64_t)p * N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e / N);
    const double a = Xr[e] / colmax[k], b = Xi ? Xi[e] / colmax[k] : 0.0;
    const double f = pow(sqrt(a * a + b * b), power - 1.0);
    Pr[e] = f * a;
    if (Pi) Pi[e] = f * b;
  }
}
// rows of X (mode-major p x N planes) <- X[:, n] * w[n] / ||X[:, n]|| (the rotated loadings re-normalised per grid point
// and weighted by the original row norms: h2-weighted X of the block-norm Grams); `renorm_only`: w is ignored
__global__ void rot_point_scale_kernel(double* __restrict__ Xr, double* __restrict__ Xi, int64_t N, int p, const double* __restrict__ w,
                                       int renorm_only) {
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int k = 0; k < p; ++k) {
      const double a = Xr[(int64_t)k * N + n], b = Xi ? Xi[(int64_t)k * N + n] : 0.0;
      acc += a * a + b * b;
    }
    const double nrm = sqrt(acc);
    // renorm_only: X / |X_n| ;  otherwise h2 X_normalised = (h |(AR)_n|) (AR)_n / |(AR)_n| = h (A R)_n
    const double f = renorm_only ? 1.0 / nrm : w[n];
    for (int k = 0; k < p; ++k) {
      Xr[(int64_t)k * N + n] *= f;
      if (Xi) Xi[(int64_t)k * N + n] *= f;
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
                                               unsigned long long* __restrict__ colmax_bits,
                                               double* __restrict__ local_r = nullptr, double* __restrict__ local_i = nullptr,
                                               double* __restrict__ res_r = nullptr, double* __restrict__ res_i = nullptr,
                                               const bool res_ready = false) {
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
    // persistent caller: the A tiles of this workgroup stay in LDS (res_*) for the whole loop and are staged once
    double* Ybr = Yr;
    double* Ybi = Yi;
    if (res_r) {
      const int64_t bl = (bt - blockIdx.x) / gridDim.x;
      Ybr = res_r + bl * pl;
      Ybi = res_i + bl * pl;
    }
    __syncthreads();
    if (!res_r || !res_ready) {
      // stage the A tile (coalesced over grid points)
      for (int e = tid; e < p * ROT_PB; e += 256) {
        const int j = e / ROT_PB, pt = e % ROT_PB;
        const int64_t n = n0 + pt;
        double vr = 0.0, vi = 0.0;
        if (n < N) {
          vr = Ar[(int64_t)j * N + n];
          if constexpr (CPLX) vi = Ai[(int64_t)j * N + n];
        }
        Ybr[j * ROT_LDP + pt] = vr;
        if constexpr (CPLX) Ybi[j * ROT_LDP + pt] = vi;
      }
      __syncthreads();
    }
    if (MODE == 1) {
      for (int e = tid; e < p * ROT_PB; e += 256) {
        const int j = e / ROT_PB, pt = e % ROT_PB;
        Xr[j * ROT_LDP + pt] = Ybr[j * ROT_LDP + pt];
        if constexpr (CPLX) Xi[j * ROT_LDP + pt] = Ybi[j * ROT_LDP + pt];
      }
      __syncthreads();
    } else {
      // z_k(pt) = sum_j a_j(pt) R[j][k]: wave handles k = wave, wave+4, ...; lanes = grid points (R reads broadcast)
      for (int k = wave; k < p; k += 4) {
        double zr = 0.0, zi = 0.0;
        for (int j = 0; j < p; ++j) {
          const double ar = Ybr[j * ROT_LDP + lane], rr = Rsr[j * p + k];
          zr += ar * rr;
          if constexpr (CPLX) {
            const double ai = Ybi[j * ROT_LDP + lane], ri = Rsi[j * p + k];
            zr -= ai * ri;
            zi += ar * ri + ai * rr;
          }
        }
        if (MODE == 0) {
          // W = (|z|^2 - gamma c_k / N) z     (rotation.py:56-57; MODE 0 carries gamma in the `power` argument)
          const double f = zr * zr + zi * zi - power * (cvec[k] / (double)N);
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
          // running maximum per lane and mode slot (mode k = wave + 4 s); ONE atomic per wave and mode behind the tile loop (round 6:
          // one per tile and mode - 162 000 atomics on ten words at C5 - took 1.5 ms)
#pragma unroll
          for (int s4 = 0; s4 < ROT_PMAX / 4; ++s4) {
            const int k = wave + 4 * s4;
            if (k < p) {
              const double xr = Xr[k * ROT_LDP + lane];
              double a2 = xr * xr;
              if constexpr (CPLX) { const double xi = Xi[k * ROT_LDP + lane]; a2 += xi * xi; }
              const double a = (n0 + lane < N) ? sqrt(a2) : 0.0;
              accr[s4] = fmax(accr[s4], a);                 // (MODE 3 has no sums: the accumulators carry the maxima)
            }
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
    const double* Lr = (MODE == 0) ? Ybr : Xr;
    const double* Li = (MODE == 0) ? Ybi : Xi;
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
  if (MODE == 3) {
#pragma unroll
    for (int s4 = 0; s4 < ROT_PMAX / 4; ++s4) {
      const int k = wave + 4 * s4;
      if (k < p) {
        double a = accr[s4];
        for (int o = 32; o > 0; o >>= 1) a = fmax(a, __shfl_xor(a, o));
        if (lane == 0 && a > 0.0) atomicMax(&colmax_bits[k], (unsigned long long)__double_as_longlong(a));
      }
    }
    return;
  }
#pragma unroll
  for (int sl = 0; sl < MAXE; ++sl) {
    const int e = tid + 256 * sl;
    if (e < p * p) {
      if (local_r) {   // persistent caller: the partial stays in this workgroup's LDS
        local_r[e] = accr[sl];
        if constexpr (CPLX) local_i[e] = acci[sl];
      } else {
        const int64_t idx = (int64_t)blockIdx.x * p * p + e;
        part_r[idx] = accr[sl];
        if constexpr (CPLX) part_i[idx] = acci[sl];
      }
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
// One workgroup per entry (grid = pp): thread t adds the partials t, t + 256, ... in that order, thread 0 then adds the 256 slice
// sums in order - one fixed summation order (up to 256 partials: the plain sequence 0, 1, 2, ... of rounds 1-5, bit for bit; the
// 2048-workgroup grids of long fields: 8 dependent loads per thread instead of 2048 - 482 us per call at C5 before).
__global__ __launch_bounds__(256) void rot_reduce_partials_kernel(const double* __restrict__ part_r, const double* __restrict__ part_i, int nwg, int pp,
                                                                 double* __restrict__ out_r, double* __restrict__ out_i) {
  __shared__ double sr_sh[256], si_sh[256];
  const int e = blockIdx.x, t = threadIdx.x;
  double sr = 0.0, si = 0.0;
  for (int w = t; w < nwg; w += 256) {
    sr += part_r[(int64_t)w * pp + e];
    if (part_i) si += part_i[(int64_t)w * pp + e];
  }
  sr_sh[t] = sr;
  si_sh[t] = si;
  __syncthreads();
  if (t == 0) {
    double ar = 0.0, ai = 0.0;
    const int m = nwg < 256 ? nwg : 256;
    for (int q = 0; q < m; ++q) { ar += sr_sh[q]; ai += si_sh[q]; }
    out_r[e] = ar;
    if (out_i) out_i[e] = ai;
  }
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
#ifdef XMCA_ROT_PROF
__device__ long long rot_prof[16];
#define ROT_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) rot_prof[k] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define ROT_STAMP(k) do { } while (0)
#endif

// Scaled Newton-Schulz step  X <- a X - b X (X^H X)  for singular values in [ell, 1]:  a = 3 rho / 2, b = rho^3 / 2 with
// rho = sqrt(3 / (1 + ell + ell^2)) maps [ell, 1] onto [ell', 1], ell' = rho ell (3 - rho^2 ell^2) / 2, the largest ell' a cubic
// of this form can reach (Chen & Chow 2014): small singular values grow by up to 2.6 per step instead of 1.5.  Safe for
// any guess of ell - the polynomial stays within (0, 1] on (0, 1] - a guess too large only delays the values below it.
// ell is advanced in place; rho = 1 (the plain iteration) once ell has reached 1.
__device__ __forceinline__ void ns_scaled_coefficients(double& ell, double& a, double& b) {
  if (ell < 1.0 - 1e-9) {
    // (hardware reciprocal square root, ~1e-8: any rho <= sqrt(3) is admissible, and every workgroup computes the same one)
    const double rs = __builtin_amdgcn_rsq(1.0 + ell + ell * ell);
    const double rho = fmin(1.7320508075688772 * rs, 1.7320508075688772), rho2 = rho * rho;
    a = 1.5 * rho;
    b = 0.5 * rho * rho2;
    ell = fmin(1.0, 0.5 * rho * ell * (3.0 - rho2 * ell * ell));
  } else {
    a = 1.5;
    b = 0.5;
    ell = 1.0;
  }
}

// Newton-Schulz for 16 < p <= 32 keeps X, T and the next X zero-padded to 32 rows at pitch ROT_NSLD (no clamped or masked
// operand, no bounds in the epilogues); the three buffers of a plane follow G and R and are time-shared with the tail's
// two p x p work matrices.
constexpr int ROT_NSLD = 34;
constexpr int ROT_NSP = 32 * ROT_NSLD;
__host__ __device__ static inline size_t rot_polar_plane(int p) {
  const size_t pp = (size_t)p * p;
  return (p > 16 && p <= 32) ? 2 * pp + 3 * (size_t)ROT_NSP : 4 * pp;
}
static inline size_t rot_polar_smem(int p, bool cplx) { return sizeof(double) * ((cplx ? 2 : 1) * rot_polar_plane(p) + 1024); }

// Newton-Schulz for p <= 16 by ONE wave with everything in registers.  X (padded to 16 x 16) lives in the C/D layout
// of v_mfma_f64_16x16x4_f64 (lane l, register r <-> entry [l/16 + 4r][l%16]), and so does its plain transpose Xt.
// With that layout  S(P, Q) = sum_r mfma(P[r], Q[r]) = P^T Q  for any two such register sets (the k index of MFMA
// number r is taken to be the row l/16 + 4r - a permutation of the summation order), so
//     T = X^H X :  Tr = S(Xr,Xr) + S(Xi,Xi),   Ti = S(Xr,Xi) - S(Xi,Xr)
//     Y = X T   :  Yr = S(Xtr,Tr) - S(Xti,Ti), Yi = S(Xtr,Ti) + S(Xti,Tr)
//     Y^T       :  Ytr = S(Tr,Xtr) - S(Ti,Xti), Yti = S(Ti,Xtr) + S(Tr,Xti)        (Tr symmetric, Ti antisymmetric)
// need no transposition, no LDS and no barrier: 12 (real) / 48 (complex) MFMAs per iteration.
// Returns the iteration count, or -1 (NaN / no convergence).  Call with one full wave; X goes to Xr/Xi (p x p).
template <bool CPLX>
__device__ __forceinline__ int varimax_ns_wave16(const double* __restrict__ Gr, const double* __restrict__ Gi, const int p,
                                                 const double inv, double* __restrict__ Xr, double* __restrict__ Xi, double ell) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  double xr[4], xtr[4], xi[4], xti[4];
  bool in[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = l4 + 4 * r, col = l15;
    in[r] = row < p && col < p;
    xr[r] = in[r] ? Gr[row * p + col] * inv : 0.0;
    xtr[r] = in[r] ? Gr[col * p + row] * inv : 0.0;
    xi[r] = (CPLX && in[r]) ? Gi[row * p + col] * inv : 0.0;
    xti[r] = (CPLX && in[r]) ? Gi[col * p + row] * inv : 0.0;
  }
  auto S = [](const double (&P)[4], const double (&Q)[4], d4_t acc, const double sign) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = Mfma<double>::mma(sign * P[r], Q[r], acc);
    return acc;
  };
  const d4_t zero = {0, 0, 0, 0};
  int it = 0;
  for (; it < 100; ++it) {
    d4_t tr = S(xr, xr, zero, 1.0), ti = zero;
    if constexpr (CPLX) {
      tr = S(xi, xi, tr, 1.0);
      ti = S(xr, xi, zero, 1.0);
      ti = S(xi, xr, ti, -1.0);
    }
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double err = fmax(fabs(tr[r] - ((l4 + 4 * r == l15) ? 1.0 : 0.0)), fabs(ti[r]));
      if (in[r] && !(err < 1e-14)) bad = true;
    }
    if (!__any(bad)) break;
    double trr[4] = {tr[0], tr[1], tr[2], tr[3]}, tii[4] = {ti[0], ti[1], ti[2], ti[3]};
    d4_t yr = S(xtr, trr, zero, 1.0), yi = zero, ytr = S(trr, xtr, zero, 1.0), yti = zero;
    if constexpr (CPLX) {
      yr = S(xti, tii, yr, -1.0);
      yi = S(xtr, tii, zero, 1.0);
      yi = S(xti, trr, yi, 1.0);
      ytr = S(tii, xti, ytr, -1.0);
      yti = S(tii, xtr, zero, 1.0);
      yti = S(trr, xti, yti, 1.0);
    }
    double ca, cb;
    ns_scaled_coefficients(ell, ca, cb);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xr[r] = ca * xr[r] - cb * yr[r];
      xtr[r] = ca * xtr[r] - cb * ytr[r];
      if constexpr (CPLX) {
        xi[r] = ca * xi[r] - cb * yi[r];
        xti[r] = ca * xti[r] - cb * yti[r];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (in[r]) {
      Xr[(l4 + 4 * r) * p + l15] = xr[r];
      if constexpr (CPLX) Xi[(l4 + 4 * r) * p + l15] = xi[r];
    }
  }
  return it < 100 ? it : -1;
}

// ---- the whole polar step for p <= 16 in ONE wave (round 4) ----
// DPP move of a double (both halves) and sums / minima over the 16 lanes of a DPP row and over the wave.
template <int CTRL>
__device__ __forceinline__ double rot_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: every lane of a row ends up with the row's total
__device__ __forceinline__ double rot_row_sum(double v) {
  v += rot_dpp<0xB1>(v);
  v += rot_dpp<0x4E>(v);
  v += rot_dpp<0x141>(v);
  v += rot_dpp<0x140>(v);
  return v;
}
__device__ __forceinline__ double rot_row_min(double v) {
  v = fmin(v, rot_dpp<0xB1>(v));
  v = fmin(v, rot_dpp<0x4E>(v));
  v = fmin(v, rot_dpp<0x141>(v));
  v = fmin(v, rot_dpp<0x140>(v));
  return v;
}
__device__ __forceinline__ double rot_lane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double rot_wave_sum(double v) {      // (rows added in a fixed order; uniform)
  v = rot_row_sum(v);
  return ((rot_lane(v, 0) + rot_lane(v, 16)) + rot_lane(v, 32)) + rot_lane(v, 48);
}

// G (p x p, LDS) -> R = polar factor, c, d and the stopping rule: what varimax_polar_step does behind the sum of the partials,
// by one wave without a barrier or an LDS round trip.  Everything lives in the C/D layout of varimax_ns_wave16 (lane l,
// register r <-> entry [l / 16 + 4 r][l % 16]): ||G||_F, d = Re tr(R^H G), H_kk and the column sums c_k = Re sum_j conj(R_jk)
// (A0 R)_jk are in-lane sums over r, two cross-row exchanges and DPP row reductions; A0 R = S(A0^T, R) is four (real) MFMAs on
// the registers the iteration leaves behind.  The block-wide form took 5k cycles for this tail and two barriers for the norm.
// NR = ceil(p / 4) (round 5): the contraction index of product number r is the row l / 16 + 4 r, and the rows from p on are
// zero padding - the products r >= NR multiply zeros by zeros.  Leaving them out changes no bit and takes a quarter (p <= 12)
// or half (p <= 8) of the dependent MFMAs out of every Newton-Schulz iteration.
template <bool CPLX, int NR>
__device__ __forceinline__ void varimax_polar_wave16_full(const double* __restrict__ Gr, const double* __restrict__ Gi, const int p,
                                                          const double* __restrict__ A0r, const double* __restrict__ A0i,
                                                          double* __restrict__ Rr, double* __restrict__ Ri, double* __restrict__ cvec,
                                                          double* __restrict__ state, const double tol) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  double g_r[4], g_i[4], xr[4], xtr[4], xi[4], xti[4];
  bool in[4];
  double f2 = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = l4 + 4 * r, col = l15;
    in[r] = row < p && col < p;
    g_r[r] = in[r] ? Gr[row * p + col] : 0.0;
    xtr[r] = in[r] ? Gr[col * p + row] : 0.0;
    g_i[r] = (CPLX && in[r]) ? Gi[row * p + col] : 0.0;
    xti[r] = (CPLX && in[r]) ? Gi[col * p + row] : 0.0;
    f2 += g_r[r] * g_r[r] + g_i[r] * g_i[r];
  }
  const double inv = 1.0 / sqrt(rot_wave_sum(f2));
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    xr[r] = g_r[r] * inv;
    xtr[r] *= inv;
    xi[r] = g_i[r] * inv;
    xti[r] *= inv;
  }
  auto S = [](const double (&P)[4], const double (&Q)[4], d4_t acc, const double sign) {
#pragma unroll
    for (int r = 0; r < NR; ++r) acc = Mfma<double>::mma(sign * P[r], Q[r], acc);
    return acc;
  };
  const d4_t zero = {0, 0, 0, 0};
  double ell = (state[6] > 1e-6 && state[6] < 1.0) ? state[6] : 1e-3;
  int it = 0;
  for (; it < 100; ++it) {
    d4_t tr = S(xr, xr, zero, 1.0), ti = zero;
    if constexpr (CPLX) {
      tr = S(xi, xi, tr, 1.0);
      ti = S(xr, xi, zero, 1.0);
      ti = S(xi, xr, ti, -1.0);
    }
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double err = fmax(fabs(tr[r] - ((l4 + 4 * r == l15) ? 1.0 : 0.0)), fabs(ti[r]));
      if (in[r] && !(err < 1e-14)) bad = true;
    }
    if (!__any(bad)) break;
    double trr[4] = {tr[0], tr[1], tr[2], tr[3]}, tii[4] = {ti[0], ti[1], ti[2], ti[3]};
    d4_t yr = S(xtr, trr, zero, 1.0), yi = zero, ytr = S(trr, xtr, zero, 1.0), yti = zero;
    if constexpr (CPLX) {
      yr = S(xti, tii, yr, -1.0);
      yi = S(xtr, tii, zero, 1.0);
      yi = S(xti, trr, yi, 1.0);
      ytr = S(tii, xti, ytr, -1.0);
      yti = S(tii, xtr, zero, 1.0);
      yti = S(trr, xti, yti, 1.0);
    }
    double ca, cb;
    ns_scaled_coefficients(ell, ca, cb);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xr[r] = ca * xr[r] - cb * yr[r];
      xtr[r] = ca * xtr[r] - cb * ytr[r];
      if constexpr (CPLX) {
        xi[r] = ca * xi[r] - cb * yi[r];
        xti[r] = ca * xti[r] - cb * yti[r];
      }
    }
  }
  const bool ok = it < 100;
  // A0 R = S(A0^T, R), A0^T = conj(A0) (Hermitian): real part symmetric, imaginary part antisymmetric
  double a_r[4], a_i[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a_r[r] = in[r] ? A0r[(l4 + 4 * r) * p + l15] : 0.0;
    a_i[r] = (CPLX && in[r]) ? A0i[(l4 + 4 * r) * p + l15] : 0.0;
  }
  d4_t wr = S(a_r, xr, zero, 1.0), wi = zero;
  if constexpr (CPLX) {
    wr = S(a_i, xi, wr, 1.0);
    wi = S(a_r, xi, zero, 1.0);
    wi = S(a_i, xr, wi, -1.0);
  }
  double hk = 0.0, ck = 0.0;     // this lane's share of column l15: rows l4, l4 + 4, l4 + 8, l4 + 12
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    hk += xr[r] * g_r[r];
    ck += xr[r] * wr[r];
    if constexpr (CPLX) {
      hk += xi[r] * g_i[r];
      ck += xi[r] * wi[r];
    }
  }
  hk += __shfl_xor(hk, 16);
  ck += __shfl_xor(ck, 16);
  hk += __shfl_xor(hk, 32);
  ck += __shfl_xor(ck, 32);
  // every lane holds H_kk and c_k of its column k = l15 now (the four rows agree bit for bit: the additions commute)
  const double dsum = rot_lane(rot_row_sum(hk), 0);
  const double hmin = rot_lane(rot_row_min(l15 < p ? hk : 1.7e308), 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (in[r]) {
      Rr[(l4 + 4 * r) * p + l15] = xr[r];
      if constexpr (CPLX) Ri[(l4 + 4 * r) * p + l15] = xi[r];
    }
  }
  if (lane < p) cvec[lane] = ck;
  if (lane == 0) {
    state[6] = 0.5 * hmin * inv;                  // next iteration's guess (ignored unless within (1e-6, 1))
    const double d_old = state[2];
    state[3] = d_old;
    state[2] = dsum;
    state[0] += 1.0;
    state[5] = (double)(ok ? it : 100);
    if (!ok || !(dsum == dsum)) state[4] = 1.0;                  // NaN / singular G
    else if (fabs(dsum - d_old) / dsum < tol) state[1] = 1.0;    // rotation.py:62
  }
}

// One 16 x 16 tile of T = X^H X (TSTEP) or of Y = X T (!TSTEP) with the KS k-steps unrolled: the accumulators then stay
// in the MFMA's registers from the first to the last step (as a run-time loop the compiler moves all of them to VGPRs
// and back every step and waits for each MFMA - 2.3x the time of the whole iteration).  (ar, ai) is the A operand at
// a0 + s * sa, (br, bi) the B operand at b0 + s * sb; the last step's summation index is clamped and masked.
template <bool CPLX, bool TSTEP, int KS>
__device__ __forceinline__ void ns_tile_unrolled(const double* __restrict__ Ar, const double* __restrict__ Ai,
                                                 const double* __restrict__ Br, const double* __restrict__ Bi, const int a0,
                                                 const int sa, const int b0, const int sb, const int last_a, const int last_b,
                                                 const double last_mask, d4_t& acc_r, d4_t& acc_i) {
  double ar[KS], ai[KS], br[KS], bi[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int ia = s == KS - 1 ? last_a : a0 + s * sa, ib = s == KS - 1 ? last_b : b0 + s * sb;
    ar[s] = Ar[ia];
    br[s] = Br[ib];
    if constexpr (CPLX) { ai[s] = Ai[ia]; bi[s] = Bi[ib]; }
  }
  __builtin_amdgcn_sched_barrier(0);   // every load is in flight before the first MFMA (otherwise: load, wait, MFMA, load, ...)
  ar[KS - 1] *= last_mask;
  if constexpr (CPLX) ai[KS - 1] *= last_mask;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    acc_r = Mfma<double>::mma(ar[s], br[s], acc_r);
    if constexpr (CPLX) {                                                  // (the two chains alternate: no MFMA waits for its predecessor)
      acc_i = Mfma<double>::mma(ar[s], bi[s], acc_i);
      acc_r = Mfma<double>::mma(TSTEP ? ai[s] : -ai[s], bi[s], acc_r);   // T: conj(a) b;  Y: a b
      acc_i = Mfma<double>::mma(TSTEP ? -ai[s] : ai[s], br[s], acc_i);
    }
  }
}

template <bool CPLX, bool TSTEP>
__device__ __forceinline__ void ns_tile(const int ksteps, const double* __restrict__ Ar, const double* __restrict__ Ai,
                                        const double* __restrict__ Br, const double* __restrict__ Bi, const int a0, const int sa,
                                        const int b0, const int sb, const int last_a, const int last_b, const double last_mask,
                                        d4_t& acc_r, d4_t& acc_i) {
#define XMCA_NS_CASE(KS) case KS: ns_tile_unrolled<CPLX, TSTEP, KS>(Ar, Ai, Br, Bi, a0, sa, b0, sb, last_a, last_b, last_mask, acc_r, acc_i); break;
  switch (ksteps) {   // p <= 64 (1 .. 4: the tail's A0 R of few modes)
    XMCA_NS_CASE(1) XMCA_NS_CASE(2) XMCA_NS_CASE(3) XMCA_NS_CASE(4) XMCA_NS_CASE(5) XMCA_NS_CASE(6) XMCA_NS_CASE(7) XMCA_NS_CASE(8) XMCA_NS_CASE(9) XMCA_NS_CASE(10) XMCA_NS_CASE(11)
    XMCA_NS_CASE(12) XMCA_NS_CASE(13) XMCA_NS_CASE(14) XMCA_NS_CASE(15) XMCA_NS_CASE(16)
    default: break;
  }
#undef XMCA_NS_CASE
}

// Newton-Schulz for 16 < p <= 32: one 16 x 16 tile of T = X^H X and of Y = X T per wave, KS = ceil(p / 4) summation steps,
// every LDS offset computed once before the loop.  P0 / P1 / PT: the padded buffers (real planes; imaginary at + plane).
// Leaves the polar factor in (Xr, Xi) at pitch p.  Returns the iteration count, -1 on NaN / no convergence.
template <bool CPLX, int KS>
__device__ __forceinline__ int varimax_ns_pt2(const double* __restrict__ Gr, const double* __restrict__ Gi, const int p, const double inv,
                                              double* __restrict__ Xr, double* __restrict__ Xi, double* __restrict__ padr,
                                              double* __restrict__ padi, int* __restrict__ nsflag, double ell) {
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4, wv = tid >> 6;
  const int ti = wv >> 1, tj = wv & 1;
  double* Cr = padr;                 // current X
  double* Ci = padi;
  double* Nr = padr + ROT_NSP;       // next X
  double* Ni = padi + ROT_NSP;
  double* Tr = padr + 2 * ROT_NSP;
  double* Ti = padi + 2 * ROT_NSP;
  for (int idx = tid; idx < ROT_NSP; idx += 256) {
    const int row = idx / ROT_NSLD, col = idx - row * ROT_NSLD;
    const bool in = row < p && col < p;
    const int g = in ? row * p + col : 0;
    const double vr = Gr[g], vi = CPLX ? Gi[g] : 0.0;
    Cr[idx] = in ? vr * inv : 0.0;
    if constexpr (CPLX) Ci[idx] = in ? vi * inv : 0.0;
  }
  const int oa_t = l4 * ROT_NSLD + 16 * ti + l15;        // T step: a = X[k][16 ti + i], b = X[k][16 tj + j], k = l4 + 4 s
  const int ob = l4 * ROT_NSLD + 16 * tj + l15;          //         (and the b operand T[k][16 tj + j] of the Y step)
  const int oa_y = (16 * ti + l15) * ROT_NSLD + l4;      // Y step: a = X[16 ti + i][k]
  const int oo = (16 * ti + l4) * ROT_NSLD + 16 * tj + l15;   // this lane's four entries of the tile: rows + 4 r
  double tgt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + l4 + 4 * r, col = 16 * tj + l15;
    tgt[r] = (row == col && row < p) ? 1.0 : 0.0;
  }
  __syncthreads();
  int it = 0;
  bool ok = false;
  for (; it < 100; ++it) {
    if (it == 1) ROT_STAMP(8);
    {
      double ar[KS], ai[KS], br[KS], bi[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        ar[s] = Cr[oa_t + 4 * s * ROT_NSLD];
        br[s] = Cr[ob + 4 * s * ROT_NSLD];
        if constexpr (CPLX) {
          ai[s] = Ci[oa_t + 4 * s * ROT_NSLD];
          bi[s] = Ci[ob + 4 * s * ROT_NSLD];
        }
      }
      d4_t tr = {0, 0, 0, 0}, tim = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        tr = Mfma<double>::mma(ar[s], br[s], tr);
        if constexpr (CPLX) {                         // conj(a) b
          tim = Mfma<double>::mma(ar[s], bi[s], tim);
          tr = Mfma<double>::mma(ai[s], bi[s], tr);
          tim = Mfma<double>::mma(-ai[s], br[s], tim);
        }
      }
      if (it == 1) ROT_STAMP(14);
      bool open = false, nan = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Tr[oo + 4 * r * ROT_NSLD] = tr[r];
        if constexpr (CPLX) Ti[oo + 4 * r * ROT_NSLD] = tim[r];
        const double err = fmax(fabs(tr[r] - tgt[r]), fabs(tim[r]));
        open |= !(err < 1e-14);
        nan |= !(err == err) || err == HUGE_VAL;
      }
      if (open) nsflag[it & 1] = 1;
      if (nan) nsflag[2] = 1;
    }
    if (tid == 0) nsflag[(it + 1) & 1] = 0;   // next iteration's flag (its last reader passed the previous barrier)
    if (it == 1) ROT_STAMP(9);
    __syncthreads();
    if (it == 1) ROT_STAMP(10);
    if (nsflag[2]) break;                       // NaN / inf
    if (!nsflag[it & 1]) { ok = true; break; }
    if (it == 1) ROT_STAMP(11);
    double coef_a, coef_b;
    ns_scaled_coefficients(ell, coef_a, coef_b);
    {
      double ar[KS], ai[KS], br[KS], bi[KS], xr[4], xi[4];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        ar[s] = Cr[oa_y + 4 * s];
        br[s] = Tr[ob + 4 * s * ROT_NSLD];
        if constexpr (CPLX) {
          ai[s] = Ci[oa_y + 4 * s];
          bi[s] = Ti[ob + 4 * s * ROT_NSLD];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xr[r] = Cr[oo + 4 * r * ROT_NSLD];
        xi[r] = CPLX ? Ci[oo + 4 * r * ROT_NSLD] : 0.0;
      }
      d4_t yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        yr = Mfma<double>::mma(ar[s], br[s], yr);
        if constexpr (CPLX) {                         // a b
          yi = Mfma<double>::mma(ar[s], bi[s], yi);
          yr = Mfma<double>::mma(-ai[s], bi[s], yr);
          yi = Mfma<double>::mma(ai[s], br[s], yi);
        }
      }
      if (it == 1) ROT_STAMP(15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Nr[oo + 4 * r * ROT_NSLD] = coef_a * xr[r] - coef_b * yr[r];
        if constexpr (CPLX) Ni[oo + 4 * r * ROT_NSLD] = coef_a * xi[r] - coef_b * yi[r];
      }
    }
    if (it == 1) ROT_STAMP(12);
    __syncthreads();
    if (it == 1) ROT_STAMP(13);
    { double* t = Cr; Cr = Nr; Nr = t; }
    { double* t = Ci; Ci = Ni; Ni = t; }
  }
  for (int e = tid; e < p * p; e += 256) {      // the tail works at pitch p
    const int row = e / p, col = e - row * p;
    Xr[e] = Cr[row * ROT_NSLD + col];
    if constexpr (CPLX) Xi[e] = Ci[row * ROT_NSLD + col];
  }
  __syncthreads();
  return ok ? it : -1;
}

template <bool CPLX>
__device__ __forceinline__ void varimax_polar_step(double* __restrict__ sm, const double* __restrict__ part_r, const double* __restrict__ part_i,
                                   int nwg, int p, const double* __restrict__ A0r, const double* __restrict__ A0i,
                                   double* __restrict__ Rr, double* __restrict__ Ri, double* __restrict__ cvec,
                                   double* __restrict__ state, double tol) {
  const int tid = threadIdx.x, pp = p * p;
  double* Gr = sm;            // G, later A0 R
  double* Xr = Gr + pp;
  double* Tr = Xr + pp;       // (16 < p <= 32: T and Y share their space with the padded buffers of varimax_ns_pt2)
  double* Yr = Tr + pp;
  double* scr = sm + rot_polar_plane(p);      // 1024 doubles of scratch
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

  // G = sum of partials, always in the same order: thread slice sl of entry e adds the workgroups sl, sl + nsl, ...
  // (16 loads in flight - the partials come from L2 after the acquire), the slices are then added in order.
  const int nsl = pp <= 128 ? 256 / pp : 1;
  double fro2 = 0.0;
  const int pairs = pp >> 1;
  if (nwg > 1 && (pp & 1) == 0 && pairs <= 64) {
    // few modes, even p^2: two entries per load, four slices of the workgroups, 20 loads in flight - one or two trips to L2
    // instead of three or more (C2: 79 partials of 100 entries).  Slice sl adds the workgroups sl, sl + 4, ... in order, the
    // slices are then added in order: one fixed summation order per entry.
    const int sl = tid / pairs, e2 = tid - sl * pairs;
    if (sl < 4) {
      double2 sr = make_double2(0.0, 0.0), si = make_double2(0.0, 0.0);
      for (int w0 = sl; w0 < nwg; w0 += 20 * 4) {
        double2 vr[20], vi[20];
#pragma unroll
        for (int u = 0; u < 20; ++u) {
          const int ww = w0 + 4 * u;
          vr[u] = make_double2(0.0, 0.0);
          vi[u] = make_double2(0.0, 0.0);
          if (ww < nwg) {
            vr[u] = *reinterpret_cast<const double2*>(part_r + (int64_t)ww * pp + 2 * e2);
            if constexpr (CPLX) vi[u] = *reinterpret_cast<const double2*>(part_i + (int64_t)ww * pp + 2 * e2);
          }
        }
#pragma unroll
        for (int u = 0; u < 20; ++u) {
          sr.x += vr[u].x; sr.y += vr[u].y;
          if constexpr (CPLX) { si.x += vi[u].x; si.y += vi[u].y; }
        }
      }
      scr[sl * pp + 2 * e2] = sr.x;
      scr[sl * pp + 2 * e2 + 1] = sr.y;
      if constexpr (CPLX) {
        scr[512 + sl * pp + 2 * e2] = si.x;
        scr[512 + sl * pp + 2 * e2 + 1] = si.y;
      }
    }
    __syncthreads();
    for (int e = tid; e < pp; e += 256) {
      const double sr = ((scr[e] + scr[pp + e]) + scr[2 * pp + e]) + scr[3 * pp + e];
      Gr[e] = sr;
      fro2 += sr * sr;
      if constexpr (CPLX) {
        const double si = ((scr[512 + e] + scr[512 + pp + e]) + scr[512 + 2 * pp + e]) + scr[512 + 3 * pp + e];
        Gi[e] = si;
        fro2 += si * si;
      }
    }
  } else if (nsl > 1) {
    const int sl = tid / pp, e = tid % pp;
    if (sl < nsl) {
      double sr = 0.0, si = 0.0;
      for (int w0 = sl; w0 < nwg; w0 += 16 * nsl) {
        double vr[16], vi[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int ww = w0 + u * nsl;
          vr[u] = ww < nwg ? part_r[(int64_t)ww * pp + e] : 0.0;
          if constexpr (CPLX) vi[u] = ww < nwg ? part_i[(int64_t)ww * pp + e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          sr += vr[u];
          if constexpr (CPLX) si += vi[u];
        }
      }
      scr[tid] = sr;
      if constexpr (CPLX) scr[256 + tid] = si;
    }
    __syncthreads();
    for (int e2 = tid; e2 < pp; e2 += 256) {
      double sr = 0.0, si = 0.0;
      for (int s2 = 0; s2 < nsl; ++s2) {
        sr += scr[s2 * pp + e2];
        if constexpr (CPLX) si += scr[256 + s2 * pp + e2];
      }
      Gr[e2] = sr;
      fro2 += sr * sr;
      if constexpr (CPLX) { Gi[e2] = si; fro2 += si * si; }
    }
  } else {
    for (int e = tid; e < pp; e += 256) {
      double sr = 0.0, si = 0.0;
      for (int w0 = 0; w0 < nwg; w0 += 16) {
        double vr[16], vi[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          vr[u] = w0 + u < nwg ? part_r[(int64_t)(w0 + u) * pp + e] : 0.0;
          if constexpr (CPLX) vi[u] = w0 + u < nwg ? part_i[(int64_t)(w0 + u) * pp + e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          sr += vr[u];
          if constexpr (CPLX) si += vi[u];
        }
      }
      Gr[e] = sr;
      fro2 += sr * sr;
      if constexpr (CPLX) { Gi[e] = si; fro2 += si * si; }
    }
  }
  ROT_STAMP(3);
  static_assert(ROT_PMAX >= 16, "");
  if (p <= 16) {
    // one wave, everything in registers (it starts again from G in LDS; the block-wide norm is not needed)
    __syncthreads();
    if (tid < 64) {
      if (p <= 4) varimax_polar_wave16_full<CPLX, 1>(Gr, Gi, p, A0r, A0i, Rr, Ri, cvec, state, tol);
      else if (p <= 8) varimax_polar_wave16_full<CPLX, 2>(Gr, Gi, p, A0r, A0i, Rr, Ri, cvec, state, tol);
      else if (p <= 12) varimax_polar_wave16_full<CPLX, 3>(Gr, Gi, p, A0r, A0i, Rr, Ri, cvec, state, tol);
      else varimax_polar_wave16_full<CPLX, 4>(Gr, Gi, p, A0r, A0i, Rr, Ri, cvec, state, tol);
    }
    ROT_STAMP(4);
    ROT_STAMP(5);
    return;                        // (the callers synchronise before anybody reads R, c or the state)
  }
  fro2 = block_reduce(fro2, false);
  const double inv = 1.0 / sqrt(fro2);
  for (int e = tid; e < pp; e += 256) {
    Xr[e] = Gr[e] * inv;
    if constexpr (CPLX) Xi[e] = Gi[e] * inv;
  }
  __syncthreads();

  // Newton-Schulz with two barriers per iteration: X and Y alternate roles (no copy); "some entry of X^H X is still
  // 1e-14 away from I" is a flag in LDS that the barrier publishing T makes visible (no reduction tree), and the
  // (j, k) of a thread's first entry is computed once (a 32-bit division per matmul is as long as the matmul).
  int it = 0;
  bool ok = false;
  // lower bound guess for the singular values of X0 = G / ||G||_F: half the smallest diagonal entry of H = R^H G of the
  // previous Varimax iteration (state[6], written by the tail below; H changes slowly), 1e-3 at the start
  const double ell0 = (state[6] > 1e-6 && state[6] < 1.0) ? state[6] : 1e-3;
  double ell = ell0;
  double* Cr = Xr;   // current iterate
  double* Ci = Xi;
  double* Nr = Yr;   // next iterate
  double* Ni = Yi;
  int* nsflag = reinterpret_cast<int*>(scr + 1008);    // [0..1]: "not converged" by parity, [2]: NaN seen, [3]: wave16 result
  if (tid < 3) nsflag[tid] = 0;
  __syncthreads();
  static_assert(ROT_PMAX >= 16, "");
  if (p <= 16) {
    // register-resident MFMA form; it restarts from G (the scaled copy in X is not needed)
    if (tid < 64) {
      const int n_it = varimax_ns_wave16<CPLX>(Gr, Gi, p, inv, Xr, Xi, ell0);
      if (tid == 0) nsflag[3] = n_it;
    }
    __syncthreads();
    it = nsflag[3];
    ok = it >= 0;
    if (!ok) it = 100;
  }
  if (p > 16 && p <= 32) {
    __syncthreads();            // (X0 at pitch p, written above, is not used by this form: it starts from G)
    int n_it;
    switch ((p + 3) >> 2) {
      case 5: n_it = varimax_ns_pt2<CPLX, 5>(Gr, Gi, p, inv, Xr, Xi, Tr, Ti, nsflag, ell0); break;
      case 6: n_it = varimax_ns_pt2<CPLX, 6>(Gr, Gi, p, inv, Xr, Xi, Tr, Ti, nsflag, ell0); break;
      case 7: n_it = varimax_ns_pt2<CPLX, 7>(Gr, Gi, p, inv, Xr, Xi, Tr, Ti, nsflag, ell0); break;
      default: n_it = varimax_ns_pt2<CPLX, 8>(Gr, Gi, p, inv, Xr, Xi, Tr, Ti, nsflag, ell0); break;
    }
    ok = n_it >= 0;
    it = ok ? n_it : 100;
  }
  // p > 32: the two products of an iteration as 16 x 16 MFMA tiles (v_mfma_f64_16x16x4_f64), the PT x PT output tiles
  // dealt round-robin to the four waves; operands straight from LDS (X and T at pitch p).  Rows / columns beyond p are
  // read from a clamped address - an operand row (column) only reaches the same row (column) of the product, which is
  // not stored - and the summation index beyond p (last step only) is masked with a factor 0: no branch around any load.
  //   T = X^H X :  a = X[k][16 ti + i], b = X[k][16 tj + j]:   Tr = ar br + ai bi,  Ti = ar bi - ai br
  //   Y = X T   :  a = X[16 ti + i][k], b = T[k][16 tj + j]:   Yr = ar br - ai bi,  Yi = ar bi + ai br
  const int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4, wv = tid >> 6;
  const int PT = (p + 15) >> 4, ksteps = (p + 3) >> 2;
  const int klast = min(4 * (ksteps - 1) + l4, p - 1);                    // summation index of the last step, clamped ...
  const double kmask = 4 * (ksteps - 1) + l4 < p ? 1.0 : 0.0;             // ... and masked
  for (; p > 32 && it < 100; ++it) {
    if (it == 1) ROT_STAMP(8);
    for (int t = wv; t < PT * PT; t += 4) {
      const int ia = 16 * (t / PT) + l15, jb = 16 * (t % PT) + l15, ca = min(ia, p - 1), cb = min(jb, p - 1);
      d4_t tr = {0, 0, 0, 0}, ti = {0, 0, 0, 0};
      ns_tile<CPLX, true>(ksteps, Cr, Ci, Cr, Ci, l4 * p + ca, 4 * p, l4 * p + cb, 4 * p, klast * p + ca, klast * p + cb, kmask, tr, ti);
      if (it == 1) ROT_STAMP(14);
      bool open = false, nan = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * (t / PT) + l4 + 4 * r;
        if (row < p && jb < p) {
          Tr[row * p + jb] = tr[r];
          if constexpr (CPLX) Ti[row * p + jb] = ti[r];
          const double err = fmax(fabs(tr[r] - (row == jb ? 1.0 : 0.0)), fabs(ti[r]));
          open |= !(err < 1e-14);
          nan |= !(tr[r] == tr[r]) || !(ti[r] == ti[r]) || err == HUGE_VAL;
        }
      }
      if (open) nsflag[it & 1] = 1;
      if (nan) nsflag[2] = 1;
    }
    if (tid == 0) nsflag[(it + 1) & 1] = 0;   // next iteration's flag (its last reader passed the previous barrier)
    if (it == 1) ROT_STAMP(9);
    __syncthreads();
    if (it == 1) ROT_STAMP(10);
    if (nsflag[2]) break;                       // NaN / inf
    if (!nsflag[it & 1]) { ok = true; break; }
    if (it == 1) ROT_STAMP(11);
    double coef_a, coef_b;
    ns_scaled_coefficients(ell, coef_a, coef_b);
    for (int t = wv; t < PT * PT; t += 4) {
      const int ia = 16 * (t / PT) + l15, jb = 16 * (t % PT) + l15, ca = min(ia, p - 1), cb = min(jb, p - 1);
      d4_t yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0};
      ns_tile<CPLX, false>(ksteps, Cr, Ci, Tr, Ti, ca * p + l4, 4, l4 * p + cb, 4 * p, ca * p + klast, klast * p + cb, kmask, yr, yi);
      if (it == 1) ROT_STAMP(15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * (t / PT) + l4 + 4 * r;
        if (row < p && jb < p) {
          Nr[row * p + jb] = coef_a * Cr[row * p + jb] - coef_b * yr[r];
          if constexpr (CPLX) Ni[row * p + jb] = coef_a * Ci[row * p + jb] - coef_b * yi[r];
        }
      }
    }
    if (it == 1) ROT_STAMP(12);
    __syncthreads();
    if (it == 1) ROT_STAMP(13);
    { double* t = Cr; Cr = Nr; Nr = t; }
    { double* t = Ci; Ci = Ni; Ni = t; }
  }
  ROT_STAMP(4);
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
  if (tid < 4 * p) {                              // H_kk = Re sum_j conj(R[j][k]) G[j][k], four threads per column
    const int k = tid >> 2;
    double hk = 0.0;
    for (int j = tid & 3; j < p; j += 4) {
      hk += Xr[j * p + k] * Gr[j * p + k];
      if constexpr (CPLX) hk += Xi[j * p + k] * Gi[j * p + k];
    }
    hk += __shfl_xor(hk, 1);
    hk += __shfl_xor(hk, 2);
    if ((tid & 3) == 0) scr[512 + k] = hk;
  }
  dsum = block_reduce(dsum, false);
  if (tid == 0) {
    double hmin = scr[512];
    for (int k = 1; k < p; ++k) hmin = fmin(hmin, scr[512 + k]);
    state[6] = 0.5 * hmin * inv;                  // next iteration's guess (ignored unless within (1e-6, 1))
  }
  {
    // prod = Re conj(R) . (A0 R), the product as MFMA tiles like the Newton-Schulz Y step (operands: A0 staged in T, R in X)
    __syncthreads();
    for (int t = wv; t < PT * PT; t += 4) {
      const int ia = 16 * (t / PT) + l15, jb = 16 * (t % PT) + l15, ca = min(ia, p - 1), cb = min(jb, p - 1);
      d4_t yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0};
      ns_tile<CPLX, false>(ksteps, Tr, Ti, Xr, Xi, ca * p + l4, 4, l4 * p + cb, 4 * p, ca * p + klast, klast * p + cb, kmask, yr, yi);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * (t / PT) + l4 + 4 * r;
        if (row < p && jb < p) {
          double prod = Xr[row * p + jb] * yr[r];
          if constexpr (CPLX) prod += Xi[row * p + jb] * yi[r];
          Yr[row * p + jb] = prod;
        }
      }
    }
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
    ROT_STAMP(5);
    if (!ok || !(dsum == dsum)) state[4] = 1.0;                  // NaN / singular G
    else if (fabs(dsum - d_old) / dsum < tol) state[1] = 1.0;    // rotation.py:62
  }
}

template <bool CPLX>
__global__ __launch_bounds__(256) void varimax_iter_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai,
                                                           const double* __restrict__ h, int64_t N, int p,
                                                           const double* __restrict__ A0r, const double* __restrict__ A0i,
                                                           double* Rr, double* Ri, double* cvec, double* state, double* part_r,
                                                           double* part_i, unsigned int* counter, double tol, double gamma) {
  if (state[1] != 0.0 || state[4] != 0.0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
#ifdef XMCA_ROT_PROF
  const long long t_start = (long long)__builtin_readcyclecounter();
#endif
  rot_accum_body<CPLX, 0, 0>(sm, Ar, Ai, h, N, N, p, Rr, Ri, cvec, nullptr, gamma, part_r, part_i, nullptr);
#ifdef XMCA_ROT_PROF
  const long long t_acc = (long long)__builtin_readcyclecounter();
#endif
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
#ifdef XMCA_ROT_PROF
  if (threadIdx.x == 0 && blockIdx.x == 0) { rot_prof[0] = t_start; rot_prof[1] = t_acc; }
#endif
  ROT_STAMP(2);
  varimax_polar_step<CPLX>(sm, part_r, part_i, (int)gridDim.x, p, A0r, A0i, Rr, Ri, cvec, state, tol);
}

// The accumulation below for p <= 16 (one mode tile) with every loop unrolled: KS = ceil(p / 4) summation steps in the Z
// product.  Wave w owns the points 16 w .. 16 w + 15 of a tile for BOTH stages: in the C/D layout of the MFMA register r of
// lane (l4, l15) holds Z[point l4 + 4 r][mode l15], which is exactly the B operand of summation step r of G = A^H W - so W
// never leaves the registers, and a tile costs neither an LDS round trip nor a barrier (the general form: W to LDS, barrier,
// W back, run-time loops: 8.9k cycles per iteration at C2 for 14 MFMAs per wave).  The four partial G of the waves are
// added in wave order at the end.
template <bool CPLX, int KS>
__device__ __forceinline__ void varimax_accum_mfma_pt1(double* __restrict__ sm, const double* __restrict__ Ar,
                                                       const double* __restrict__ Ai, const int64_t N, const int p,
                                                       const double* __restrict__ Rr, const double* __restrict__ Ri,
                                                       const double* __restrict__ cvec, double* __restrict__ out_r,
                                                       double* __restrict__ out_i, double* __restrict__ res_r,
                                                       double* __restrict__ res_i, const bool res_ready, const double gamma) {
  const int pl = p * ROT_LDP, pp = p * p;
  double* Yr = sm + pl;                   // (the layout of varimax_accum_mfma: Xs, tile, R copy, weights per plane)
  double* Yi = sm + 3 * pl + pp + ROT_PB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  // B operands of Z = A R:  R[4 s + l4][l15], zero outside p x p
  double rbr[KS], rbi[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int j = 4 * s + l4;
    const bool in = j < p && l15 < p;
    const int idx = in ? j * p + l15 : 0;
    const double vr = Rr[idx], vi = CPLX ? Ri[idx] : 0.0;
    rbr[s] = in ? vr : 0.0;
    rbi[s] = in ? vi : 0.0;
  }
  const double cn = l15 < p ? gamma * (cvec[l15] / (double)N) : 0.0;
  const bool gjin = l15 < p;
  const int gja = (gjin ? l15 : 0) * ROT_LDP + wave * 16 + l4;     // A operand of G: conj(A[j = l15][16 w + 4 s + l4])
  d4_t gr = {0, 0, 0, 0}, gi = {0, 0, 0, 0};

  const int64_t nbatch = (N + ROT_PB - 1) / ROT_PB;
  // Tiles that do not stay in LDS (long grids: 63 tiles per workgroup and iteration at C5) are PREFETCHED: the loads of the next
  // PF tiles of this workgroup are in flight - in registers, at most 4 values per thread and tile for p <= 16 - while a tile is
  // multiplied (round 6; one tile at a time read the planes at 0.4 TB/s: a trip to memory per tile, 207 us per iteration at C5).
  // Each thread fetches PAIRS of neighbouring points (16-byte loads: half the requests of 8-byte ones - the accumulation of a long
  // grid is bound by the number of requests in flight, 2.1 TB/s with 8-byte loads): pair e = tid + 256 u of a tile = mode e / 32,
  // points 2 (e % 32), + 1.
  constexpr int PF = CPLX ? 1 : 8, NV = 2;      // (complex: the kernel is at its 512 registers - one tile ahead)
  double2 pfr[PF][NV], pfi[CPLX ? PF : 1][CPLX ? NV : 1];
  // (issue only: a select on the loaded value here would make the compiler wait for the load at once - out-of-range pairs read
  //  a valid pair and are masked when the tile is written to LDS.  0: both points inside, 1: only the first (the pair then read is
  //  (N - 2, N - 1): the point is its SECOND element), 2: none)
  auto pf_state = [&](const int64_t bt, const int u) {
    const int e = tid + 256 * u, j = e / (ROT_PB / 2), pt = 2 * (e % (ROT_PB / 2));
    const int64_t n = bt * ROT_PB + pt;
    if (!(bt < nbatch && j < p) || n >= N) return 2;
    return n + 1 < N ? 0 : 1;
  };
  auto pf_issue = [&](const int64_t bt, double2 (&vr)[NV], double2* vi) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = tid + 256 * u, j = e / (ROT_PB / 2), pt = 2 * (e % (ROT_PB / 2));
      const int st8 = pf_state(bt, u);
      const int64_t a = st8 == 0 ? (int64_t)j * N + bt * ROT_PB + pt : (st8 == 1 && N >= 2 ? (int64_t)j * N + N - 2 : 0);
      vr[u] = *reinterpret_cast<const double2*>(Ar + a);
      if constexpr (CPLX) vi[u] = *reinterpret_cast<const double2*>(Ai + a);
    }
  };
  auto pf_first = [&](const double2 v, const int st8) { return st8 == 0 ? v.x : (st8 == 1 ? v.y : 0.0); };
  auto pf_second = [&](const double2 v, const int st8) { return st8 == 0 ? v.y : 0.0; };
  // one 64-point tile staged at (Tr, Ti): Z = A R on this wave's 16 points, W, G += A^H W (everything in registers)
  auto tile_ops = [&](const double* __restrict__ Tr, const double* __restrict__ Ti, d4_t& g_r, d4_t& g_i) {
    double ar[KS], ai[KS], qa[4], qb[4];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int j = 4 * s + l4;
      const bool jin = j < p;
      const int a = (jin ? j : 0) * ROT_LDP + wave * 16 + l15;
      const double vr = Tr[a], vi = CPLX ? Ti[a] : 0.0;
      ar[s] = jin ? vr : 0.0;
      ai[s] = jin ? vi : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double va = Tr[gja + 4 * s], vb = CPLX ? Ti[gja + 4 * s] : 0.0;
      qa[s] = gjin ? va : 0.0;
      qb[s] = gjin ? vb : 0.0;
    }
    d4_t zr = {0, 0, 0, 0}, zi = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      zr = Mfma<double>::mma(ar[s], rbr[s], zr);
      if constexpr (CPLX) {
        zi = Mfma<double>::mma(ar[s], rbi[s], zi);
        zr = Mfma<double>::mma(-ai[s], rbi[s], zr);
        zi = Mfma<double>::mma(ai[s], rbr[s], zi);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // W = (|z|^2 - c_k / N) z     (rotation.py:56-57);  G[j][k] += sum_pt conj(A[pt][j]) W[pt][k], step r = points l4 + 4 r
      const double f = zr[r] * zr[r] + zi[r] * zi[r] - cn;
      const double wr = f * zr[r], wi = f * zi[r];
      g_r = Mfma<double>::mma(qa[r], wr, g_r);
      if constexpr (CPLX) {
        g_i = Mfma<double>::mma(qa[r], wi, g_i);
        g_r = Mfma<double>::mma(qb[r], wi, g_r);      // conj(a) w
        g_i = Mfma<double>::mma(-qb[r], wr, g_i);
      }
    }
  };
  const bool streaming = !res_r;
  if (streaming) {
    // Two tiles per step, each with its own accumulator: the seven (real) dependent MFMAs of a tile - ~100 cycles each for the one
    // wave of a SIMD - overlap with the other tile's (round 6: 1.2 us per tile -> see DESIGN.md); the second staging buffer is the
    // area the general form keeps W in.  A slot beyond the last tile holds zeros: it adds nothing.
    // (complex: one tile per step - the kernel is at its 512 registers)
    constexpr int STEP = CPLX ? 1 : 2;
    static_assert(PF % STEP == 0, "tiles are processed in pairs");
    double* T2r = sm;
    double* T2i = sm + 2 * pl + pp + ROT_PB;
    d4_t gr2 = {0, 0, 0, 0}, gi2 = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < PF; ++d) pf_issue((int64_t)blockIdx.x + (int64_t)d * gridDim.x, pfr[d], CPLX ? pfi[d] : nullptr);
    for (int64_t bt0 = blockIdx.x; bt0 < nbatch; bt0 += (int64_t)PF * gridDim.x) {
#pragma unroll
      for (int d = 0; d < PF; d += STEP) {
        const int64_t bt = bt0 + (int64_t)d * gridDim.x, btb = bt + gridDim.x;
        constexpr int D2 = STEP - 1;        // (index offset of the pair's second slot)
        if (bt >= nbatch) break;
        __syncthreads();                    // the staging buffers are no longer read
#pragma unroll
        for (int u = 0; u < NV; ++u) {
          const int e = tid + 256 * u, j = e / (ROT_PB / 2), pt = 2 * (e % (ROT_PB / 2));
          if (j < p) {
            const int sa = pf_state(bt, u);
            Yr[j * ROT_LDP + pt] = pf_first(pfr[d][u], sa);
            Yr[j * ROT_LDP + pt + 1] = pf_second(pfr[d][u], sa);
            if constexpr (CPLX) {
              Yi[j * ROT_LDP + pt] = pf_first(pfi[d][u], sa);
              Yi[j * ROT_LDP + pt + 1] = pf_second(pfi[d][u], sa);
            }
            if constexpr (STEP == 2) {
              const int sb = pf_state(btb, u);
              T2r[j * ROT_LDP + pt] = pf_first(pfr[d + D2][u], sb);
              T2r[j * ROT_LDP + pt + 1] = pf_second(pfr[d + D2][u], sb);
              if constexpr (CPLX) {
                T2i[j * ROT_LDP + pt] = pf_first(pfi[d + D2][u], sb);
                T2i[j * ROT_LDP + pt + 1] = pf_second(pfi[d + D2][u], sb);
              }
            }
          }
        }
        pf_issue(bt + (int64_t)PF * gridDim.x, pfr[d], CPLX ? pfi[d] : nullptr);            // (these slots' next tiles, PF tiles ahead)
        if constexpr (STEP == 2) pf_issue(btb + (int64_t)PF * gridDim.x, pfr[d + D2], CPLX ? pfi[d + D2] : nullptr);
        __syncthreads();
        tile_ops(Yr, Yi, gr, gi);
        if constexpr (STEP == 2) tile_ops(T2r, T2i, gr2, gi2);
      }
    }
    if constexpr (STEP == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { gr[r] += gr2[r]; gi[r] += gi2[r]; }
    }
  } else {
    for (int64_t bt = blockIdx.x; bt < nbatch; bt += gridDim.x) {
      const int64_t n0 = bt * ROT_PB;
      const int64_t bl = (bt - blockIdx.x) / gridDim.x;
      double* Tr = res_r + bl * pl;
      double* Ti = res_i + bl * pl;
      if (!res_ready) {
        __syncthreads();
        for (int e = tid; e < p * ROT_PB; e += 256) {
          const int j = e / ROT_PB, pt = e % ROT_PB;
          const int64_t n = n0 + pt;
          double vr = 0.0, vi = 0.0;
          if (n < N) {
            vr = Ar[(int64_t)j * N + n];
            if constexpr (CPLX) vi = Ai[(int64_t)j * N + n];
          }
          Tr[j * ROT_LDP + pt] = vr;
          if constexpr (CPLX) Ti[j * ROT_LDP + pt] = vi;
        }
        __syncthreads();
      }
      tile_ops(Tr, Ti, gr, gi);
    }
  }
  __syncthreads();      // the tiles are no longer read: the scratch below may overlap them
  double* scr_r = sm;                 // [wave][p][p]
  double* scr_i = sm + 4 * pp;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = l4 + 4 * r, k = l15;
    if (j < p && k < p) {
      scr_r[wave * pp + j * p + k] = gr[r];
      if constexpr (CPLX) scr_i[wave * pp + j * p + k] = gi[r];
    }
  }
  __syncthreads();
  for (int e = tid; e < pp; e += 256) {
    out_r[e] = ((scr_r[e] + scr_r[pp + e]) + scr_r[2 * pp + e]) + scr_r[3 * pp + e];
    if constexpr (CPLX) out_i[e] = ((scr_i[e] + scr_i[pp + e]) + scr_i[2 * pp + e]) + scr_i[3 * pp + e];
  }
}

// The accumulation below for 16 < p <= 32 (two mode tiles per dimension) with every loop unrolled: KS = ceil(p / 4) summation
// steps in the Z products.  R's MFMA operands are loaded once per iteration instead of once per tile and product, all LDS
// operands of a stage are requested before its first MFMA, and the four accumulator chains of a stage alternate (the
// run-time loops of the general form wait for every LDS operand and every dependent MFMA in turn: 180 cycles per MFMA
// against 64 of issue - C3's 20 complex modes: 93k cycles per iteration in this stage).  One G tile per wave.
template <bool CPLX, int KS>
__device__ __forceinline__ void varimax_accum_mfma_pt2(double* __restrict__ sm, const double* __restrict__ Ar,
                                                       const double* __restrict__ Ai, const int64_t N, const int p,
                                                       const double* __restrict__ Rr, const double* __restrict__ Ri,
                                                       const double* __restrict__ cvec, double* __restrict__ out_r,
                                                       double* __restrict__ out_i, double* __restrict__ res_r,
                                                       double* __restrict__ res_i, const bool res_ready, const double gamma) {
  const int pl = p * ROT_LDP, pp = p * p;
  double* Xr = sm;
  double* Yr = Xr + pl;
  double* Xi = Yr + pl + pp + ROT_PB;     // (the layout of varimax_accum_mfma: Xs, tile, R copy, weights per plane)
  double* Yi = Xi + pl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  // B operands of Z = A R:  R[4 s + l4][16 kt + l15], zero outside p x p
  double rbr[2][KS], rbi[2][KS];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int j = 4 * s + l4, k = 16 * kt + l15;
      const bool in = j < p && k < p;
      const int idx = in ? j * p + k : 0;
      const double vr = Rr[idx], vi = CPLX ? Ri[idx] : 0.0;
      rbr[kt][s] = in ? vr : 0.0;
      rbi[kt][s] = in ? vi : 0.0;
    }
  }
  double cn[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int k = 16 * kt + l15;
    cn[kt] = k < p ? gamma * (cvec[k] / (double)N) : 0.0;
  }
  // this wave's tile of G
  const int gjt = wave >> 1, gkt = wave & 1;
  const int gj = 16 * gjt + l15, gk = 16 * gkt + l15;
  const bool gjin = gj < p, gkin = gk < p;
  const int gja = (gjin ? gj : 0) * ROT_LDP + l4, gka = (gkin ? gk : 0) * ROT_LDP + l4;
  d4_t gr = {0, 0, 0, 0}, gi = {0, 0, 0, 0};

  const int64_t nbatch = (N + ROT_PB - 1) / ROT_PB;
  for (int64_t bt = blockIdx.x; bt < nbatch; bt += gridDim.x) {
    const int64_t n0 = bt * ROT_PB;
    double* Tr = Yr;
    double* Ti = Yi;
    if (res_r) {
      const int64_t bl = (bt - blockIdx.x) / gridDim.x;
      Tr = res_r + bl * pl;
      Ti = res_i + bl * pl;
    }
    __syncthreads();                      // the previous tile's W is no longer read
    if (!res_r || !res_ready) {
      for (int e = tid; e < p * ROT_PB; e += 256) {
        const int j = e / ROT_PB, pt = e % ROT_PB;
        const int64_t n = n0 + pt;
        double vr = 0.0, vi = 0.0;
        if (n < N) {
          vr = Ar[(int64_t)j * N + n];
          if constexpr (CPLX) vi = Ai[(int64_t)j * N + n];
        }
        Tr[j * ROT_LDP + pt] = vr;
        if constexpr (CPLX) Ti[j * ROT_LDP + pt] = vi;
      }
      __syncthreads();
    }
    // ---- Z rows of this wave's 16 points, both mode tiles at once ----
    double ar[KS], ai[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int j = 4 * s + l4;
      const bool jin = j < p;
      const int a = (jin ? j : 0) * ROT_LDP + wave * 16 + l15;
      const double vr = Tr[a], vi = CPLX ? Ti[a] : 0.0;
      ar[s] = jin ? vr : 0.0;
      ai[s] = jin ? vi : 0.0;
    }
    d4_t zr[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, zi[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      zr[0] = Mfma<double>::mma(ar[s], rbr[0][s], zr[0]);
      zr[1] = Mfma<double>::mma(ar[s], rbr[1][s], zr[1]);
      if constexpr (CPLX) {
        zi[0] = Mfma<double>::mma(ar[s], rbi[0][s], zi[0]);
        zi[1] = Mfma<double>::mma(ar[s], rbi[1][s], zi[1]);
        zr[0] = Mfma<double>::mma(-ai[s], rbi[0][s], zr[0]);
        zr[1] = Mfma<double>::mma(-ai[s], rbi[1][s], zr[1]);
        zi[0] = Mfma<double>::mma(ai[s], rbr[0][s], zi[0]);
        zi[1] = Mfma<double>::mma(ai[s], rbr[1][s], zi[1]);
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int k = 16 * kt + l15;
      if (k < p) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // W = (|z|^2 - c_k / N) z     (rotation.py:56-57)
          const double f = zr[kt][r] * zr[kt][r] + zi[kt][r] * zi[kt][r] - cn[kt];
          const int pt = wave * 16 + l4 + 4 * r;
          Xr[k * ROT_LDP + pt] = f * zr[kt][r];
          if constexpr (CPLX) Xi[k * ROT_LDP + pt] = f * zi[kt][r];
        }
      }
    }
    __syncthreads();
    // ---- G[j][k] += sum_pt conj(A[pt][j]) W[pt][k]: 16 steps of four points, eight steps' operands in flight ----
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      double qa[8], qb[8], wa[8], wb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int o = 4 * (8 * half + u);
        const double va = Tr[gja + o], vw = Xr[gka + o];
        const double vb = CPLX ? Ti[gja + o] : 0.0, vx = CPLX ? Xi[gka + o] : 0.0;
        qa[u] = gjin ? va : 0.0;
        qb[u] = gjin ? vb : 0.0;
        wa[u] = gkin ? vw : 0.0;
        wb[u] = gkin ? vx : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        gr = Mfma<double>::mma(qa[u], wa[u], gr);
        if constexpr (CPLX) {
          gi = Mfma<double>::mma(qa[u], wb[u], gi);
          gr = Mfma<double>::mma(qb[u], wb[u], gr);      // conj(a) w
          gi = Mfma<double>::mma(-qb[u], wa[u], gi);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = 16 * gjt + l4 + 4 * r, k = 16 * gkt + l15;
    if (j < p && k < p) {
      out_r[j * p + k] = gr[r];
      if constexpr (CPLX) out_i[j * p + k] = gi[r];
    }
  }
}

// MFMA form of the Varimax accumulation (MODE 0 of rot_accum_body) for the persistent kernel:
//   Z = A R (64 grid points x p per tile),  W = (|Z|^2 - c/N) Z,  G += A^H W
// as v_mfma_f64_16x16x4_f64 products on 16-wide mode tiles (PT = ceil(p/16) per dimension).  Wave w computes the Z rows
// of points 16w..16w+15; the G tiles are split over the waves (PT >= 2) or, when there is a single tile, the 64
// points are (4 partial tiles, added in wave order).  The accumulators stay in registers over all tiles of the
// workgroup.  LDS: Xs = W[k][pt], tiles = A[j][pt] (both pitch ROT_LDP: conflict-free operand reads), Rs = R[j][k].
template <bool CPLX>
__device__ __forceinline__ void varimax_accum_mfma(double* __restrict__ sm, const double* __restrict__ Ar,
                                                   const double* __restrict__ Ai, const int64_t N, const int p,
                                                   const double* __restrict__ Rr, const double* __restrict__ Ri,
                                                   const double* __restrict__ cvec, double* __restrict__ out_r,
                                                   double* __restrict__ out_i, double* __restrict__ res_r,
                                                   double* __restrict__ res_i, const bool res_ready, const double gamma) {
  const int pl = p * ROT_LDP, pp = p * p;
  double* Xr = sm;
  double* Yr = Xr + pl;
  double* Rsr = Yr + pl;
  double* wgt = Rsr + pp;
  double* Xi = wgt + ROT_PB;
  double* Yi = Xi + pl;
  double* Rsi = Yi + pl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int PT = (p + 15) / 16;
  if (PT == 1) {
    switch ((p + 3) >> 2) {
      case 1: varimax_accum_mfma_pt1<CPLX, 1>(sm, Ar, Ai, N, p, Rr, Ri, cvec, out_r, out_i, res_r, res_i, res_ready, gamma); return;
      case 2: varimax_accum_mfma_pt1<CPLX, 2>(sm, Ar, Ai, N, p, Rr, Ri, cvec, out_r, out_i, res_r, res_i, res_ready, gamma); return;
      case 3: varimax_accum_mfma_pt1<CPLX, 3>(sm, Ar, Ai, N, p, Rr, Ri, cvec, out_r, out_i, res_r, res_i, res_ready, gamma); return;
      default: varimax_accum_mfma_pt1<CPLX, 4>(sm, Ar, Ai, N, p, Rr, Ri, cvec, out_r, out_i, res_r, res_i, res_ready, gamma); return;
    }
  }
  if (PT == 2) {
    switch ((p + 3) >> 2) {
      case 5: varimax_accum_mfma_pt2<CPLX, 5>(sm, Ar, Ai, N, p, Rr, Ri, cvec, out_r, out_i, res_r, res_i, res_ready, gamma); return;
      case 6: varimax_accum_mfma_pt2<CPLX, 6>(sm, Ar, Ai, N, p, Rr, Ri, cvec, out_r, out_i, res_r, res_i, res_ready, gamma); return;
      case 7: varimax_accum_mfma_pt2<CPLX, 7>(sm, Ar, Ai, N, p, Rr, Ri, cvec, out_r, out_i, res_r, res_i, res_ready, gamma); return;
      default: varimax_accum_mfma_pt2<CPLX, 8>(sm, Ar, Ai, N, p, Rr, Ri, cvec, out_r, out_i, res_r, res_i, res_ready, gamma); return;
    }
  }
  constexpr int MAXT = 4;                 // G tiles per wave: ceil(PT^2 / 4) <= 4
  for (int e = tid; e < pp; e += 256) {
    Rsr[e] = Rr[e];
    if constexpr (CPLX) Rsi[e] = Ri[e];
  }
  double cn[MAXT];                        // gamma c_k / N for this lane's column of mode tile kt
#pragma unroll
  for (int kt = 0; kt < MAXT; ++kt) {
    const int k = kt * 16 + l15;
    cn[kt] = (kt < PT && k < p) ? gamma * (cvec[k] / (double)N) : 0.0;
  }
  d4_t gr[MAXT], gi[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) { gr[t] = d4_t{0, 0, 0, 0}; gi[t] = d4_t{0, 0, 0, 0}; }
  const bool ksplit = PT == 1;

  const int64_t nbatch = (N + ROT_PB - 1) / ROT_PB;
  for (int64_t bt = blockIdx.x; bt < nbatch; bt += gridDim.x) {
    const int64_t n0 = bt * ROT_PB;
    double* Tr = Yr;
    double* Ti = Yi;
    if (res_r) {
      const int64_t bl = (bt - blockIdx.x) / gridDim.x;
      Tr = res_r + bl * pl;
      Ti = res_i + bl * pl;
    }
    __syncthreads();
    if (!res_r || !res_ready) {
      for (int e = tid; e < p * ROT_PB; e += 256) {
        const int j = e / ROT_PB, pt = e % ROT_PB;
        const int64_t n = n0 + pt;
        double vr = 0.0, vi = 0.0;
        if (n < N) {
          vr = Ar[(int64_t)j * N + n];
          if constexpr (CPLX) vi = Ai[(int64_t)j * N + n];
        }
        Tr[j * ROT_LDP + pt] = vr;
        if constexpr (CPLX) Ti[j * ROT_LDP + pt] = vi;
      }
      __syncthreads();
    }
    // Z rows of this wave's 16 points, one mode tile at a time; W goes to Xs
    for (int kt = 0; kt < PT; ++kt) {
      d4_t zr = {0, 0, 0, 0}, zi = {0, 0, 0, 0};
      const int k = kt * 16 + l15;
      for (int j0 = 0; j0 < p; j0 += 4) {
        const int j = j0 + l4;
        const bool jin = j < p, kin = jin && k < p;
        const int ja = jin ? j : 0, ka = k < p ? k : 0;
        double ar = Tr[ja * ROT_LDP + wave * 16 + l15], br = Rsr[ja * p + ka];
        ar = jin ? ar : 0.0;
        br = kin ? br : 0.0;
        zr = Mfma<double>::mma(ar, br, zr);
        if constexpr (CPLX) {
          double ai = Ti[ja * ROT_LDP + wave * 16 + l15], bi = Rsi[ja * p + ka];
          ai = jin ? ai : 0.0;
          bi = kin ? bi : 0.0;
          zr = Mfma<double>::mma(-ai, bi, zr);
          zi = Mfma<double>::mma(ar, bi, zi);
          zi = Mfma<double>::mma(ai, br, zi);
        }
      }
      if (k < p) {
        double cnk = cn[0];
#pragma unroll
        for (int q = 1; q < MAXT; ++q) cnk = (kt == q) ? cn[q] : cnk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // W = (|z|^2 - c_k / N) z     (rotation.py:56-57)
          const double f = zr[r] * zr[r] + zi[r] * zi[r] - cnk;
          const int pt = wave * 16 + l4 + 4 * r;
          Xr[k * ROT_LDP + pt] = f * zr[r];
          if constexpr (CPLX) Xi[k * ROT_LDP + pt] = f * zi[r];
        }
      }
    }
    __syncthreads();
    // G[j][k] += sum_pt conj(A[pt][j]) W[pt][k]
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int tile = ksplit ? 0 : wave + 4 * t;
      if ((ksplit && t > 0) || tile >= PT * PT) continue;
      const int jt = tile / PT, kt = tile % PT;
      const int j = jt * 16 + l15, k = kt * 16 + l15;
      const bool jin = j < p, kin = k < p;
      const int ja = jin ? j : 0, ka = kin ? k : 0;
      const int pbeg = ksplit ? wave * 16 : 0, pend = ksplit ? wave * 16 + 16 : ROT_PB;
      d4_t ar4 = gr[t], ai4 = gi[t];
      for (int pt0 = pbeg; pt0 < pend; pt0 += 4) {
        double ar = Tr[ja * ROT_LDP + pt0 + l4], wr = Xr[ka * ROT_LDP + pt0 + l4];
        ar = jin ? ar : 0.0;
        wr = kin ? wr : 0.0;
        ar4 = Mfma<double>::mma(ar, wr, ar4);
        if constexpr (CPLX) {
          double ai = Ti[ja * ROT_LDP + pt0 + l4], wi = Xi[ka * ROT_LDP + pt0 + l4];
          ai = jin ? ai : 0.0;
          wi = kin ? wi : 0.0;
          ar4 = Mfma<double>::mma(ai, wi, ar4);      // conj(a) w
          ai4 = Mfma<double>::mma(ar, wi, ai4);
          ai4 = Mfma<double>::mma(-ai, wr, ai4);
        }
      }
      gr[t] = ar4;
      gi[t] = ai4;
    }
  }
  __syncthreads();      // Xs / tiles are no longer read: the scratch below may overlap them
  if (ksplit) {
    double* scr_r = sm;                 // [wave][p][p]
    double* scr_i = sm + 4 * pp;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = l4 + 4 * r, k = l15;
      if (j < p && k < p) {
        scr_r[wave * pp + j * p + k] = gr[0][r];
        if constexpr (CPLX) scr_i[wave * pp + j * p + k] = gi[0][r];
      }
    }
    __syncthreads();
    for (int e = tid; e < pp; e += 256) {
      out_r[e] = ((scr_r[e] + scr_r[pp + e]) + scr_r[2 * pp + e]) + scr_r[3 * pp + e];
      if constexpr (CPLX) out_i[e] = ((scr_i[e] + scr_i[pp + e]) + scr_i[2 * pp + e]) + scr_i[3 * pp + e];
    }
  } else {
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int tile = wave + 4 * t;
      if (tile >= PT * PT) continue;
      const int jt = tile / PT, kt = tile % PT;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = jt * 16 + l4 + 4 * r, k = kt * 16 + l15;
        if (j < p && k < p) {
          out_r[j * p + k] = gr[t][r];
          if constexpr (CPLX) out_i[j * p + k] = gi[t][r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The whole Varimax loop in ONE launch.  Every workgroup keeps R, c and the iteration state in its own LDS; per
// iteration it (1) accumulates its partial G over its share of the grid points, (2) publishes it (write-through
// stores + one epoch flag per workgroup; cdna_hip_programming.md G16 form R1), (3) waits for the flags of all
// workgroups, takes one agent-scope acquire and (4) reduces the partials in a fixed order and runs the polar step
// itself.  All workgroups execute the same arithmetic on the same numbers, so they agree on R and on the stopping
// iteration without a second exchange.  Payload buffers alternate with the epoch parity: a workgroup can be at most
// one epoch ahead of the slowest one, because epoch e+1 cannot be completed by anybody before everybody published it.
// The grid must be resident (<= one workgroup per CU is requested by the host); every spin is bounded.
// ---------------------------------------------------------------------------------------------------------------
static inline size_t rot_persistent_smem(int p, bool cplx) {
  return std::max(rot_accum_smem(p, cplx), rot_polar_smem(p, cplx)) + sizeof(double) * (6 * (size_t)p * p + p + ROT_STATE_N + 8);
}
// LDS for keeping `nb` A tiles (64 grid points each) resident
static inline size_t rot_resident_smem(int p, bool cplx, int nb) { return sizeof(double) * (cplx ? 2 : 1) * (size_t)nb * p * ROT_LDP; }

template <bool CPLX>
__global__ __launch_bounds__(256) void varimax_persistent_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai,
                                                                 const double* __restrict__ h, int64_t N, int p,
                                                                 const double* __restrict__ A0r, const double* __restrict__ A0i,
                                                                 double* Rr, double* Ri, double* cvec, double* state, double* part_r,
                                                                 double* part_i, unsigned int* flags, double tol, int max_iter,
                                                                 size_t work_doubles, int resident_tiles, double gamma, int poll_delay, double* sums) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);          // accumulate / polar scratch (time-shared)
  const int pp = p * p, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
  double* Rl_r = sm + work_doubles;
  double* Rl_i = Rl_r + pp;
  double* acc_r = Rl_i + pp;
  double* acc_i = acc_r + pp;
  double* a0_r = acc_i + pp;
  double* a0_i = a0_r + pp;
  double* cl = a0_i + pp;
  double* stl = cl + p;
  double* res_r = resident_tiles > 0 ? stl + ROT_STATE_N + 8 : nullptr;      // [tile][p][ROT_LDP], then the imaginary planes
  double* res_i = (CPLX && resident_tiles > 0) ? res_r + (size_t)resident_tiles * p * ROT_LDP : res_r;
  __shared__ int give_up;
  for (int e = tid; e < pp; e += 256) {
    Rl_r[e] = Rr[e];
    Rl_i[e] = CPLX ? Ri[e] : 0.0;
    a0_r[e] = A0r[e];
    a0_i[e] = CPLX ? A0i[e] : 0.0;
  }
  for (int e = tid; e < p; e += 256) cl[e] = cvec[e];
  if (tid < ROT_STATE_N) stl[tid] = state[tid];
  if (tid == 0) give_up = 0;
  __syncthreads();

  for (int it = 0; it < max_iter; ++it) {
    if (stl[1] != 0.0 || stl[4] != 0.0) break;                // converged / NaN: identical in every workgroup
    const unsigned int epoch = (unsigned int)it + 1u;
    double* pub_r = part_r + (size_t)(it & 1) * nwg * pp;
    double* pub_i = CPLX ? part_i + (size_t)(it & 1) * nwg * pp : nullptr;
    ROT_STAMP(0);
    varimax_accum_mfma<CPLX>(sm, Ar, Ai, N, p, Rl_r, Rl_i, cl, acc_r, acc_i, res_r, res_i, it > 0, gamma);
    __syncthreads();
    ROT_STAMP(1);
    for (int e = tid; e < pp; e += 256) {
      __hip_atomic_store(pub_r + (size_t)bid * pp + e, acc_r[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (CPLX) __hip_atomic_store(pub_i + (size_t)bid * pp + e, acc_i[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains before the flag goes out
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + bid, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ROT_STAMP(6);
    // Two-stage reduction (sums != nullptr): entry e of G is added up - over the workgroups in a fixed order - by workgroup
    // e / chunk alone, which publishes the sums under a second flag; everybody then reads one p x p matrix.  The all-to-all
    // form (every workgroup reads every partial: nwg x p^2 doubles each, 34k cycles for C3's 110 x 800) remains for grids
    // too small to split the entries.
    constexpr int PLANES = CPLX ? 2 : 1;
    const int chunk = (pp + nwg - 1) / nwg;                    // entries per reducing workgroup
    const int nred = (pp + chunk - 1) / chunk;                 // workgroups that reduce
    const bool two_stage = sums != nullptr && chunk * PLANES <= 64 && nwg >= 16;
    auto wait_flags = [&](const unsigned int* f, int count) {
      if (wave == 0) {
        // (the first poll waits: polling while the others still publish slows them down - measured on the tridiagonal
        //  reduction's exchange, csrc/tridiag.h)
        if (poll_delay >= 16) __builtin_amdgcn_s_sleep(16);
        else if (poll_delay >= 8) __builtin_amdgcn_s_sleep(8);
        else if (poll_delay >= 4) __builtin_amdgcn_s_sleep(4);
        unsigned int spins = 0;
        for (;;) {
          bool ok = true;
          for (int w = lane; w < count; w += 64) ok &= __hip_atomic_load(f + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 22)) { if (lane == 0) give_up = 1; break; }   // a workgroup is missing: report, never hang
        }
        if (lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    };
    const double* red_r = pub_r;
    const double* red_i = pub_i;
    int red_n = nwg;
    if (two_stage) {
      double* sum_r = sums + (size_t)(it & 1) * 2 * pp;
      double* sum_i = sum_r + pp;
      if (bid < nred) {
        wait_flags(flags, nwg);
        ROT_STAMP(7);
        if (!give_up) {
          // column = (entry of my chunk, plane); slice sl adds the workgroups sl, sl + nsl, ... (8 loads in flight), the slices
          // are then added in order: one fixed summation order per entry
          const int e0 = bid * chunk, ne = min(chunk, pp - e0), cols = ne * PLANES, nsl = 256 / cols;
          const int col = tid % cols, sl = tid / cols;
          if (sl < nsl) {
            const double* src = (CPLX && col >= ne) ? pub_i + e0 + (col - ne) : pub_r + e0 + col;
            double acc = 0.0;
            for (int w0 = sl; w0 < nwg; w0 += 8 * nsl) {
              double v[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int ww = w0 + u * nsl;
                v[u] = ww < nwg ? __hip_atomic_load(src + (size_t)ww * pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
              }
#pragma unroll
              for (int u = 0; u < 8; ++u) acc += v[u];
            }
            sm[sl * cols + col] = acc;
          }
          __syncthreads();
          // the slice sums of a column, in slice order.  Many slices (one or two columns per workgroup: grids of 200 and more
          // workgroups with few modes - the long grids of round 6): first 16 runs of consecutive slices by 16 threads per column, then
          // the 16 run sums - 256 dependent LDS loads by ONE thread were 16k of the 143k cycles of a C5 iteration.  Up to 64 slices
          // (C3: 6 columns per workgroup, 42 slices): the plain sequence of rounds 3-5, bit for bit.
          if (nsl > 64) {
            const int run = (nsl + 15) / 16;
            double part = 0.0;
            const int c2 = tid % cols, r2 = tid / cols;
            if (r2 < 16) {
              for (int q = r2 * run; q < min((r2 + 1) * run, nsl); ++q) part += sm[q * cols + c2];
            }
            __syncthreads();
            if (r2 < 16) sm[r2 * cols + c2] = part;
            __syncthreads();
            if (tid < cols) {
              double tot = 0.0;
              for (int q = 0; q < 16; ++q) tot += sm[q * cols + tid];
              double* dst = (CPLX && tid >= ne) ? sum_i + e0 + (tid - ne) : sum_r + e0 + tid;
              __hip_atomic_store(dst, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          } else if (tid < cols) {
            double tot = 0.0;
            for (int q = 0; q < nsl; ++q) tot += sm[q * cols + tid];
            double* dst = (CPLX && tid >= ne) ? sum_i + e0 + (tid - ne) : sum_r + e0 + tid;
            __hip_atomic_store(dst, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0) __hip_atomic_store(flags + nwg + bid, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (!give_up) wait_flags(flags + nwg, nred);
      red_r = sum_r;
      red_i = sum_i;
      red_n = 1;
    } else {
      wait_flags(flags, nwg);
      ROT_STAMP(7);
    }
    if (give_up) { if (tid == 0) stl[4] = 2.0; __syncthreads(); break; }
    ROT_STAMP(2);
    varimax_polar_step<CPLX>(sm, red_r, red_i, red_n, p, a0_r, a0_i, Rl_r, Rl_i, cl, stl, tol);
    __syncthreads();
  }
  if (bid == 0) {
    for (int e = tid; e < pp; e += 256) {
      Rr[e] = Rl_r[e];
      if constexpr (CPLX) Ri[e] = Rl_i[e];
    }
    for (int e = tid; e < p; e += 256) cvec[e] = cl[e];
    if (tid < ROT_STATE_N) state[tid] = stl[tid];
  }
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
