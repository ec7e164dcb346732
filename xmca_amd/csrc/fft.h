// Batched complex DFTs of length n <= 5120 = 2^a 3^b 5^c 7^d, one workgroup per transform, the whole transform in LDS
// (Stockham autosort, in place through registers: no bit reversal, one buffer of n complex f64 = 16 n bytes).  gfx950 only.
//
// Used for the Fourier reduction of the analytic-signal path, Gy = D Phi^H (X X^T) Phi D (solver.h analytic_gram): as two
// GEMMs against the explicit Fourier vectors that is 3.75e11 flop at T = 5000 (8 ms per field); as a 2-D DFT of the T x T
// Gram matrix it is 2 x 5000 transforms of 5000 points - memory-bound on 200 MB.
#pragma once
#include "common.h"

namespace xmca {

constexpr int FFT_THREADS = 512;
constexpr int FFT_MAX_PER_THREAD = 10;           // complex values per thread and stage: 5120 / 512

struct FftPlan {
  int n = 0;
  int n_stages = 0;
  int radix[24] = {0};
};

// false when n has a prime factor above 7 or does not fit the LDS of a workgroup
static inline bool fft_plan(int n, FftPlan& p) {
  p.n = n;
  p.n_stages = 0;
  if (n < 2 || n > FFT_THREADS * FFT_MAX_PER_THREAD) return false;           // (FFT_MAX_PER_THREAD values per thread; 16 n bytes of LDS)
  int rest = n;
  for (int r : {4, 2, 3, 5, 7}) {
    while (rest % r == 0 && p.n_stages < 24) {
      p.radix[p.n_stages++] = r;
      rest /= r;
    }
  }
  return rest == 1;
}

template <int R>
struct FftRoots;   // cos / sin of 2 pi k / R
template <> struct FftRoots<2> { static constexpr double c[2] = {1.0, -1.0}; static constexpr double s[2] = {0.0, 0.0}; };
template <> struct FftRoots<3> {
  static constexpr double c[3] = {1.0, -0.5, -0.5};
  static constexpr double s[3] = {0.0, 0.86602540378443864676, -0.86602540378443864676};
};
template <> struct FftRoots<4> { static constexpr double c[4] = {1.0, 0.0, -1.0, 0.0}; static constexpr double s[4] = {0.0, 1.0, 0.0, -1.0}; };
template <> struct FftRoots<5> {
  static constexpr double c[5] = {1.0, 0.30901699437494742410, -0.80901699437494742410, -0.80901699437494742410, 0.30901699437494742410};
  static constexpr double s[5] = {0.0, 0.95105651629515357212, 0.58778525229247312917, -0.58778525229247312917, -0.95105651629515357212};
};
template <> struct FftRoots<7> {
  static constexpr double c[7] = {1.0, 0.62348980185873353053, -0.22252093395631440429, -0.90096886790241912624,
                                  -0.90096886790241912624, -0.22252093395631440429, 0.62348980185873353053};
  static constexpr double s[7] = {0.0, 0.78183148246802980871, 0.97492791218182360702, 0.43388373911755812048,
                                  -0.43388373911755812048, -0.97492791218182360702, -0.78183148246802980871};
};

// One Stockham stage of radix R on n points IN PLACE: Ns = product of the radices of the stages before it.  Every thread
// computes the outputs of all its butterflies into registers (at most FFT_MAX_PER_THREAD complex values: n <= 5120 points on
// 512 threads), a barrier, then they are written to the autosort positions.  One buffer of n complex values instead of two:
// a transform of 5000 points takes 80 KB of LDS, not all 160 - two transforms per CU, or one beside the GEMM workgroups of
// another surrogate lane (round 4; with two buffers an FFT workgroup needed a CU to itself and waited for one - up to 17 ms
// per launch inside rule_n - while blocking everybody else's workgroups from it).
template <int R>
__device__ __forceinline__ void fft_stage(double* __restrict__ ar, double* __restrict__ ai, const int n, const int Ns, const double sign) {
  const int nr = n / R;
  constexpr int ITS = (FFT_MAX_PER_THREAD + R - 1) / R;
  double yr[ITS][R], yi[ITS][R];
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int j = (int)threadIdx.x + it * FFT_THREADS;
    if (j < nr) {
      const int k = j % Ns;
      const double ang = 2.0 * (double)k / (double)(Ns * R);     // in units of pi
      double vr[R], vi[R];
      // twiddles w^r, w = exp(sign i pi ang): ONE sincospi per butterfly and R - 2 complex products; none in the first stage,
      // where w = 1
      double c1 = 1.0, s1 = 0.0;
      if (Ns > 1) {
        sincospi(ang, &s1, &c1);
        s1 *= sign;
      }
      double wr = c1, wi = s1;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double xr = ar[j + r * nr], xi = ai[j + r * nr];
        if (r == 0) { vr[0] = xr; vi[0] = xi; continue; }
        vr[r] = xr * wr - xi * wi;
        vi[r] = xr * wi + xi * wr;
        if (r + 1 < R) {
          const double t = wr * c1 - wi * s1;
          wi = wr * s1 + wi * c1;
          wr = t;
        }
      }
#pragma unroll
      for (int q = 0; q < R; ++q) {
        double zr = vr[0], zi = vi[0];
#pragma unroll
        for (int r = 1; r < R; ++r) {
          const double c = FftRoots<R>::c[(q * r) % R], sn = sign * FftRoots<R>::s[(q * r) % R];
          zr += vr[r] * c - vi[r] * sn;
          zi += vr[r] * sn + vi[r] * c;
        }
        yr[it][q] = zr;
        yi[it][q] = zi;
      }
    }
  }
  __syncthreads();                                   // every input of the stage has been read
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int j = (int)threadIdx.x + it * FFT_THREADS;
    if (j < nr) {
      const int k = j % Ns;
      const int j0 = (j - k) * R + k;
#pragma unroll
      for (int q = 0; q < R; ++q) {
        ar[j0 + q * Ns] = yr[it][q];
        ai[j0 + q * Ns] = yi[it][q];
      }
    }
  }
}

// out[b][k] = sa[k] sb[b] scale * sum_{t < n_in} x[b][t] exp(sign 2 pi i k t / n),  k < n_keep, for the transforms b = blockIdx.x,
// with x = in (in_i may be null: real input) or sin[t] * conj(in) when conj_in is set (sin may be null).
// Element (b, t) of the input is at b * in_bs + t * in_es, of the output at b * out_bs + k * out_es.
__global__ __launch_bounds__(FFT_THREADS) void fft_batch_kernel(const double* __restrict__ in_r, const double* __restrict__ in_i, int64_t in_bs,
                                                        int64_t in_es, int n_in, int conj_in, const double* __restrict__ sin_,
                                                        FftPlan plan, double sign, double* __restrict__ out_r,
                                                        double* __restrict__ out_i, int64_t out_bs, int64_t out_es, int n_keep,
                                                        const double* __restrict__ sa, const double* __restrict__ sb, double scale) {
  extern __shared__ __attribute__((aligned(16))) char fft_smem[];
  const int n = plan.n;
  double* ar = reinterpret_cast<double*>(fft_smem);
  double* ai = ar + n;
  const int64_t b = blockIdx.x;
  // (eight elements per thread requested before the first one is stored: with one wave per SIMD - a transform of 5000 points
  //  takes the whole LDS of a CU - a load per loop iteration left the memory pipe idle for most of the 14 us this took)
  for (int t0 = threadIdx.x; t0 < n; t0 += 8 * (int)blockDim.x) {
    double xr[8], xi[8], fs[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + u * (int)blockDim.x;
      const bool in = t < n_in;
      const int tc = in ? t : 0;
      fs[u] = (in && sin_) ? sin_[tc] : 1.0;
      xr[u] = in_r[b * in_bs + tc * in_es];
      xi[u] = in_i ? in_i[b * in_bs + tc * in_es] : 0.0;
      if (!in) { xr[u] = 0.0; xi[u] = 0.0; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + u * (int)blockDim.x;
      if (t < n) {
        ar[t] = fs[u] * xr[u];
        ai[t] = (conj_in ? -fs[u] : fs[u]) * xi[u];
      }
    }
  }
  __syncthreads();
  int Ns = 1;
  for (int st = 0; st < plan.n_stages; ++st) {
    const int R = plan.radix[st];
    switch (R) {
      case 2: fft_stage<2>(ar, ai, n, Ns, sign); break;
      case 3: fft_stage<3>(ar, ai, n, Ns, sign); break;
      case 4: fft_stage<4>(ar, ai, n, Ns, sign); break;
      case 5: fft_stage<5>(ar, ai, n, Ns, sign); break;
      default: fft_stage<7>(ar, ai, n, Ns, sign); break;
    }
    Ns *= R;
    __syncthreads();
  }
  const double fb = scale * (sb ? sb[b] : 1.0);
  for (int k0 = threadIdx.x; k0 < n_keep; k0 += 4 * (int)blockDim.x) {
    double fk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * (int)blockDim.x;
      fk[u] = fb * ((sa && k < n_keep) ? sa[k] : 1.0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * (int)blockDim.x;
      if (k < n_keep) {
        out_r[b * out_bs + k * out_es] = ar[k] * fk[u];
        out_i[b * out_bs + k * out_es] = ai[k] * fk[u];
      }
    }
  }
}

inline void fft_batch(hipStream_t st, const FftPlan& plan, int batch, const double* in_r, const double* in_i, int64_t in_bs, int64_t in_es,
                      double sign, double* out_r, double* out_i, int64_t out_bs, int64_t out_es, int n_keep, const double* sa,
                      const double* sb, double scale, int n_in = -1, bool conj_in = false, const double* sin_ = nullptr) {
  if (n_in < 0) n_in = plan.n;
  const size_t smem = (size_t)plan.n * 16;
  // the attribute is per device and several lane threads launch side by side: always ask for the same (maximal) limit
  // before the launch instead of caching what some other device / thread last set
  XMCA_CHECK(smem <= (size_t)160 * 1024, XMCA_ERR_UNSUPPORTED, "fft_batch: transform does not fit the LDS of a workgroup");
  XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fft_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(fft_batch_kernel, dim3(batch), dim3(FFT_THREADS), smem, st, in_r, in_i, in_bs, in_es, n_in, conj_in ? 1 : 0, sin_, plan, sign,
                     out_r, out_i, out_bs, out_es, n_keep, sa, sb, scale);
  XMCA_HIP(hipGetLastError());
}

}  // namespace xmca
