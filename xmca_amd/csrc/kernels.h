// Streaming (HBM-bound) helper kernels: conversions, plane packing, row/column scaling, row
// normalisation, column centering and the Philox surrogate generator.
#pragma once
#include "common.h"

namespace xmca {

// 1/sqrt(x) and 1/x for normal positive x, full double precision: hardware seed (5e-8 relative on gfx950,
// scripts/probes/rsq_accuracy.cpp) + two Newton steps (1.4e-16; a third changes nothing)
__device__ __forceinline__ double jac_rsqrt(const double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double h = 0.5 * x * y;
    y = fma(y, fma(-h, y, 0.5), y);
  }
  return y;
}
__device__ __forceinline__ double jac_rcp(const double x) {
  double y = __builtin_amdgcn_rcp(x);
#pragma unroll
  for (int it = 0; it < 2; ++it) y = fma(y, fma(-x, y, 1.0), y);
  return y;
}

// single precision: the hardware seeds are accurate to 1 ulp
__device__ __forceinline__ float jac_rsqrt(const float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float jac_rcp(const float x) { return __builtin_amdgcn_rcpf(x); }

constexpr int EW_BLOCK = 256;
static inline dim3 ew_grid(int64_t n, int per_thread = 1) {
  int64_t b = (n + (int64_t)EW_BLOCK * per_thread - 1) / ((int64_t)EW_BLOCK * per_thread);
  if (b > 8192) b = 8192;   // grid-stride beyond that
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

template <typename TI, typename TO>
__global__ void convert_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (TO)in[i];
}

// interleaved complex (re,im,re,im,...) -> two planes
template <typename TI, typename TO>
__global__ void split_complex_kernel(const TI* __restrict__ in, TO* __restrict__ re, TO* __restrict__ im, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    re[i] = (TO)in[2 * i];
    im[i] = (TO)in[2 * i + 1];
  }
}

// rows x cols block of planes (ld) -> dense output: real (im == nullptr) or interleaved complex; optional conj
template <typename TO>
__global__ void pack_rows_kernel(const double* __restrict__ re, const double* __restrict__ im, int64_t ld, int rows, int cols,
                                 TO* __restrict__ out, int conj) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    const double vr = re[r * ld + c];
    if (im) {
      const double vi = im[r * ld + c];
      out[2 * i] = (TO)vr;
      out[2 * i + 1] = (TO)(conj ? -vi : vi);
    } else {
      out[i] = (TO)vr;
    }
  }
}

// EOFs in their final layout (xmca_get_eofs): out[n][c] = sum_m V[m][n] W[m][c] for the mode-major vectors V (planes of TV, rows
// ld apart; Vi == nullptr: real) and a small mixing matrix W (m x q planes in float64; Wi == nullptr: real).  One thread per grid
// point n: the reads of a mode are coalesced over n, the q values of a point are written side by side.  The output is complex
// (interleaved) when V or W is.
template <typename TV, typename TO>
__global__ void eof_mix_kernel(const TV* __restrict__ Vr, const TV* __restrict__ Vi, int64_t ld, int64_t N, int m, int q,
                               const double* __restrict__ Wr, const double* __restrict__ Wi, TO* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const bool oc = Vi || Wi;
  for (int c0 = 0; c0 < q; c0 += 8) {
    double ar[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ai[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int mm = 0; mm < m; ++mm) {
      const double vr = (double)Vr[(int64_t)mm * ld + n], vi = Vi ? (double)Vi[(int64_t)mm * ld + n] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u < q ? c0 + u : q - 1;
        const double wr = Wr[mm * q + c], wi = Wi ? Wi[mm * q + c] : 0.0;
        ar[u] += vr * wr - vi * wi;
        ai[u] += vr * wi + vi * wr;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + u;
      if (c < q) {
        if (oc) { out[2 * (n * q + c)] = (TO)ar[u]; out[2 * (n * q + c) + 1] = (TO)ai[u]; }
        else out[n * q + c] = (TO)ar[u];
      }
    }
  }
}
// ... and without a mixing matrix: out[n][c] = V[c][n], a 32 x 32 tile at a time through LDS (both sides coalesced)
template <typename TV, typename TO>
__global__ __launch_bounds__(256) void eof_transpose_kernel(const TV* __restrict__ Vr, const TV* __restrict__ Vi, int64_t ld, int64_t N, int q,
                                                           TO* __restrict__ out) {
  __shared__ double tr[32][33], ti[32][33];
  const int64_t n0 = (int64_t)blockIdx.x * 32;
  const int c0 = (int)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const int64_t n = n0 + tx;
    const bool in = c < q && n < N;
    tr[r][tx] = in ? (double)Vr[(int64_t)c * ld + n] : 0.0;
    ti[r][tx] = (in && Vi) ? (double)Vi[(int64_t)c * ld + n] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t n = n0 + r;
    const int c = c0 + tx;
    if (n < N && c < q) {
      if (Vi) { out[2 * (n * q + c)] = (TO)tr[tx][r]; out[2 * (n * q + c) + 1] = (TO)ti[tx][r]; }
      else out[n * q + c] = (TO)tr[tx][r];
    }
  }
}

// X[r][c] *= s[r] (by_row) or s[c]; both planes
__global__ void scale_kernel(double* __restrict__ re, double* __restrict__ im, int64_t ld, int rows, int cols,
                             const double* __restrict__ s, int by_row, int invert) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    double f = s[by_row ? r : c];
    if (invert) f = 1.0 / f;
    re[r * ld + c] *= f;
    if (im) im[r * ld + c] *= f;
  }
}

// G[i][i] += v
__global__ void add_diag_kernel(double* __restrict__ G, int64_t ld, int n, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) G[(int64_t)i * ld + i] += v;
}

// largest |C_ij| / sqrt(C_ii C_jj), i != j, of an n x n Hermitian Gram matrix with a positive diagonal (rows with a zero
// diagonal - null modes - are skipped): how far a set of vectors is from orthogonal.  One row per workgroup (grid-stride);
// row_worst[i] (optional) = the same over j < i only.
__global__ void coherence_kernel(const double* __restrict__ Cr, const double* __restrict__ Ci, int n, int n_check,
                                 unsigned long long* __restrict__ out, double* __restrict__ row_worst) {
  __shared__ double red[4];
  double worst = 0.0;
  for (int i = blockIdx.x; i < n_check; i += gridDim.x) {      // the leading n_check rows / columns of the n x n matrix
    const double dii = Cr[(int64_t)i * n + i];
    double mine = 0.0;                                         // row i against the rows before it (the stronger modes)
    if (dii > 0.0) {
      for (int j = threadIdx.x; j < n_check; j += blockDim.x) {
        const double djj = Cr[(int64_t)j * n + j];
        if (j == i || !(djj > 0.0)) continue;
        const double re = Cr[(int64_t)i * n + j], im = Ci ? Ci[(int64_t)i * n + j] : 0.0;
        double c = sqrt((re * re + im * im) / (dii * djj));
        if (!(c == c)) c = HUGE_VAL;
        worst = fmax(worst, c);
        if (j < i) mine = fmax(mine, c);
      }
    }
    if (row_worst) {
      for (int o = 32; o > 0; o >>= 1) mine = fmax(mine, __shfl_xor(mine, o));
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mine;
      __syncthreads();
      if (threadIdx.x == 0) row_worst[i] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    }
  }
  for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o));
  if ((threadIdx.x & 63) == 0 && worst > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(worst));
}

// dinv[r] = 1 / Re sum_k A[r][k] conj(B[r][k])   (rows of two r x n plane pairs; 0 when the sum is not positive)
__global__ void row_dot_inverse_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai, const double* __restrict__ Br,
                                       const double* __restrict__ Bi, int n, double* __restrict__ dinv) {
  __shared__ double red[4];
  const int64_t row = (int64_t)blockIdx.x * n;
  double acc = 0.0;
  for (int k = threadIdx.x; k < n; k += blockDim.x) {
    acc += Ar[row + k] * Br[row + k];
    if (Ai) acc += Ai[row + k] * Bi[row + k];
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double d = (red[0] + red[1]) + (red[2] + red[3]);
    dinv[blockIdx.x] = d > 0.0 ? 1.0 / d : 0.0;
  }
}

// out[r] = sum_k A[r][k]   (n x n plane, one row per workgroup)
__global__ void row_sum_kernel(const double* __restrict__ A, int n, double* __restrict__ out) {
  __shared__ double red[4];
  const int64_t row = (int64_t)blockIdx.x * n;
  double acc = 0.0;
  for (int k = threadIdx.x; k < n; k += blockDim.x) acc += A[row + k];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// exactly Hermitian from nearly Hermitian: (G + G^H) / 2 on the planes of an n x n matrix (imaginary diagonal -> 0)
__global__ void hermitize_kernel(double* __restrict__ Gr, double* __restrict__ Gi, int n) {
  const int64_t total = (int64_t)n * n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / n), j = (int)(idx % n);
    if (j < i) continue;
    const int64_t a = (int64_t)i * n + j, b = (int64_t)j * n + i;
    const double re = 0.5 * (Gr[a] + Gr[b]);
    Gr[a] = re;
    Gr[b] = re;
    if (Gi) {
      const double im = i == j ? 0.0 : 0.5 * (Gi[a] - Gi[b]);
      Gi[a] = im;
      Gi[b] = -im;
    }
  }
}

// x[i] += v
__global__ void add_const_kernel(double* __restrict__ x, int64_t n, double v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] += v;
}

// Ht[t][s] = col[(t - s) mod T]  (circulant operator from its first column)
template <typename TO>
__global__ void circulant_kernel(const double* __restrict__ col, int T, TO* __restrict__ out) {
  const int64_t n = (int64_t)T * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i / T), s = (int)(i % T);
    int d = t - s;
    if (d < 0) d += T;
    out[i] = (TO)col[d];
  }
}

// Orthonormal Fourier vectors of the frequencies kept by the analytic signal: Phi[t][f] = exp(2 pi i f t / T) / sqrt(T),
// f = 0 .. m-1 (m = T/2 + 1 for even T, (T+1)/2 for odd T), and the Hilbert weights h_f of scipy.signal.hilbert
// (1 for DC and Nyquist, 2 otherwise), so that hilbert(x) = Phi diag(h) Phi^H x.
__global__ void fourier_basis_kernel(int T, int m, double* __restrict__ Pr, double* __restrict__ Pi, double* __restrict__ hvec) {
  const int64_t n = (int64_t)T * m;
  const double inv = 1.0 / sqrt((double)T);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i / m), f = (int)(i % m);
    const int64_t ft = ((int64_t)f * t) % T;
    double sn, cs;
    sincospi(2.0 * (double)ft / (double)T, &sn, &cs);
    Pr[i] = cs * inv;
    Pi[i] = sn * inv;
    if (i < m) hvec[i] = (i == 0 || (T % 2 == 0 && i == T / 2)) ? 1.0 : 2.0;
  }
}

__global__ void sqrt_clamp_kernel(const double* __restrict__ lam, double* __restrict__ s, int n, double scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) s[i] = sqrt(fmax(lam[i] * scale, 0.0));
}

__device__ __forceinline__ double block_sum_256(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// one workgroup per row: row <- conj?(row) / ||row||   (planes, f64).  norms_out (nullable) gets ||row||.
template <typename T>
__global__ __launch_bounds__(256) void normalize_rows_kernel(T* __restrict__ re, T* __restrict__ im, int64_t ld, int cols,
                                                             int conj, double* __restrict__ norms_out) {
  __shared__ double red[4];
  const int64_t r = blockIdx.x;
  double acc = 0.0;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const double a = (double)re[r * ld + c];
    acc += a * a;
    if (im) {
      const double b = (double)im[r * ld + c];
      acc += b * b;
    }
  }
  const double nrm = sqrt(block_sum_256(acc, red));
  const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
  for (int c = threadIdx.x; c < cols; c += 256) {
    re[r * ld + c] = (T)((double)re[r * ld + c] * inv);
    if (im) im[r * ld + c] = (T)((double)im[r * ld + c] * (conj ? -inv : inv));
  }
  if (norms_out && threadIdx.x == 0) norms_out[r] = nrm;
}

// The same for a few LONG rows (the null mode of a centered float32 field with 10^6 grid points: one workgroup walking a
// whole row takes 2.5 ms): a row is cut into chunks of ROWN_CHUNK columns, partial sums of squares per (row, chunk), then
// every chunk scales itself by the sum of its row's partials taken in chunk order (deterministic, no atomics).
constexpr int ROWN_CHUNK = 8192;
template <typename T>
__global__ __launch_bounds__(256) void row_sumsq_chunk_kernel(const T* __restrict__ re, const T* __restrict__ im, int64_t ld, int cols,
                                                              double* __restrict__ part) {
  __shared__ double red[4];
  const int64_t r = blockIdx.y;
  const int c0 = blockIdx.x * ROWN_CHUNK, c1 = min(cols, c0 + ROWN_CHUNK);
  double acc = 0.0;
  for (int c = c0 + threadIdx.x; c < c1; c += 256) {
    const double a = (double)re[r * ld + c];
    acc += a * a;
    if (im) {
      const double b = (double)im[r * ld + c];
      acc += b * b;
    }
  }
  const double s = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[r * gridDim.x + blockIdx.x] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void row_scale_chunk_kernel(T* __restrict__ re, T* __restrict__ im, int64_t ld, int cols, int conj,
                                                              const double* __restrict__ part, double* __restrict__ norms_out) {
  const int64_t r = blockIdx.y;
  double tot = 0.0;
  for (int q = 0; q < (int)gridDim.x; ++q) tot += part[r * gridDim.x + q];
  const double nrm = sqrt(tot);
  const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
  const int c0 = blockIdx.x * ROWN_CHUNK, c1 = min(cols, c0 + ROWN_CHUNK);
  for (int c = c0 + threadIdx.x; c < c1; c += 256) {
    re[r * ld + c] = (T)((double)re[r * ld + c] * inv);
    if (im) im[r * ld + c] = (T)((double)im[r * ld + c] * (conj ? -inv : inv));
  }
  if (norms_out && blockIdx.x == 0 && threadIdx.x == 0) norms_out[r] = nrm;
}
// rows <- conj?(rows) / ||row||: one workgroup per row, or - few long rows - one per chunk of a row
template <typename T>
static inline void normalize_rows(hipStream_t st, T* re, T* im, int64_t ld, int rows, int cols, int conj, double* norms_out) {
  if (rows <= 0) return;
  const int chunks = ceil_div(cols, ROWN_CHUNK);
  if (rows >= 256 || chunks < 4) {
    hipLaunchKernelGGL((normalize_rows_kernel<T>), dim3(rows), dim3(256), 0, st, re, im, ld, cols, conj, norms_out);
  } else {
    DevBuf<double> part;       // (returned to the stream's pool: the kernels queued here are ahead of its next user)
    part.ensure((size_t)rows * chunks);
    hipLaunchKernelGGL((row_sumsq_chunk_kernel<T>), dim3(chunks, rows), dim3(256), 0, st, (const T*)re, (const T*)im, ld, cols, part.get());
    hipLaunchKernelGGL((row_scale_chunk_kernel<T>), dim3(chunks, rows), dim3(256), 0, st, re, im, ld, cols, conj, (const double*)part.get(), norms_out);
  }
  XMCA_HIP(hipGetLastError());
}

// Column passes of the constructor stage.  A thread per column alone leaves the chip empty (10^4 columns = 40 workgroups):
// the rows are split into gridDim.y chunks, every (chunk, column) writes its partial result to part[chunk][c], and a
// finishing kernel adds the chunks in a fixed order (no floating-point atomics: results do not depend on scheduling).
constexpr int COL_CHUNKS = 32;

// part_nan[chunk][c] = NaN entries, part_sum[chunk][c] = sum of column c over the rows of the chunk (part_sum may be null)
template <typename T>
__global__ void column_partial_sums_kernel(const T* __restrict__ x, int rows, int64_t cols, int* __restrict__ part_nan,
                                           double* __restrict__ part_sum) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const int per = (rows + (int)gridDim.y - 1) / (int)gridDim.y;
  const int r0 = (int)blockIdx.y * per, r1 = min(rows, r0 + per);
  double s = 0.0;
  int nans = 0;
  for (int r = r0; r < r1; ++r) {
    const double v = (double)x[(int64_t)r * cols + c];
    if (v != v) ++nans;
    s += v;
  }
  part_nan[(int64_t)blockIdx.y * cols + c] = nans;
  if (part_sum) part_sum[(int64_t)blockIdx.y * cols + c] = s;
}

// nan_count[c] = sum over chunks; mean[c] = (sum over chunks) / rows   (mean may be null)
__global__ void column_finish_sums_kernel(const int* __restrict__ part_nan, const double* __restrict__ part_sum, int chunks, int rows,
                                          int64_t cols, int* __restrict__ nan_count, double* __restrict__ mean) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  int n = 0;
  double s = 0.0;
  for (int k = 0; k < chunks; ++k) {
    n += part_nan[(int64_t)k * cols + c];
    if (part_sum) s += part_sum[(int64_t)k * cols + c];
  }
  nan_count[c] = n;
  if (mean) mean[c] = s / rows;
}

// x[r][c] -= mean[c]
template <typename T>
__global__ void subtract_column_means_kernel(T* __restrict__ x, int rows, int64_t cols, const double* __restrict__ mean) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = (T)((double)x[i] - mean[i % cols]);
}

// x[r][c] -= mean[c] for the columns without NaN (a column holding one is left as it is); part_sq[chunk][c] = sum of squared deviations over the rows of the chunk
template <typename T>
__global__ void center_columns_chunk_kernel(T* __restrict__ x, int rows, int64_t cols, const double* __restrict__ mean,
                                            const int* __restrict__ nan_count, double* __restrict__ part_sq) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const int per = (rows + (int)gridDim.y - 1) / (int)gridDim.y;
  const int r0 = (int)blockIdx.y * per, r1 = min(rows, r0 + per);
  double q = 0.0;
  if (nan_count[c] == 0) {
    const double m = mean[c];
    for (int r = r0; r < r1; ++r) {
      const double d = (double)x[(int64_t)r * cols + c] - m;
      q += d * d;
      x[(int64_t)r * cols + c] = (T)d;
    }
  }
  part_sq[(int64_t)blockIdx.y * cols + c] = q;
}

// stdev[c] = sqrt(sum over chunks / rows)   (columns with NaN: the mean, as before)
__global__ void column_finish_std_kernel(const double* __restrict__ part_sq, int chunks, int rows, int64_t cols,
                                         const double* __restrict__ mean, const int* __restrict__ nan_count,
                                         double* __restrict__ stdev) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  double q = 0.0;
  for (int k = 0; k < chunks; ++k) q += part_sq[(int64_t)k * cols + c];
  stdev[c] = nan_count[c] ? mean[c] : sqrt(q / rows);
}

// out[r][j] = in[r][idx[j]]   (column selection: rows x cols_in -> rows x cols_out)
template <typename T>
__global__ void gather_columns_kernel(const T* __restrict__ in, int64_t cols_in, T* __restrict__ out, int64_t cols_out,
                                      const int64_t* __restrict__ idx, int rows) {
  const int64_t total = (int64_t)rows * cols_out;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols_out, j = i % cols_out;
    out[i] = in[r * cols_in + idx[j]];
  }
}

// x[r][c] = x[r][c] * w[c]  or  x[r][c] / w[c]   (w already in the element type: the host's own operation, bit for bit)
template <typename T>
__global__ void scale_columns_kernel(T* __restrict__ x, int rows, int64_t cols, const T* __restrict__ w, int divide) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const T f = w[i % cols];
    x[i] = divide ? x[i] / f : x[i] * f;
  }
}

// out[t][:] = in[idx[t]][:]   (row resampling of a rows x cols matrix)
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ in, T* __restrict__ out, const int64_t* __restrict__ idx, int rows,
                                   int64_t cols) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / cols, c = i % cols;
    out[i] = in[idx[t] * cols + c];
  }
}

// column sums and sums of squares of a rows x cols matrix (one thread per column, coalesced over columns)
template <typename T>
__global__ void column_moments_kernel(const T* __restrict__ x, int rows, int64_t cols, double* __restrict__ sum,
                                      double* __restrict__ sumsq) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  double s = 0.0, q = 0.0;
  for (int r = 0; r < rows; ++r) {
    const double v = (double)x[(int64_t)r * cols + c];
    s += v;
    q += v * v;
  }
  sum[c] = s;
  sumsq[c] = q;
}

// Pearson correlation from the raw cross products: r[n][j] = (C[n][j] - sx[n] sy[j] / T) / sqrt((qx[n] - sx[n]^2 / T) (qy[j] - sy[j]^2 / T))
__global__ void pearson_finish_kernel(double* __restrict__ C, int64_t N, int m, int T, const double* __restrict__ sx,
                                      const double* __restrict__ qx, const double* __restrict__ sy, const double* __restrict__ qy) {
  const int64_t total = N * m;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / m;
    const int j = (int)(i % m);
    const double cov = C[i] - sx[n] * sy[j] / T;
    const double vx = qx[n] - sx[n] * sx[n] / T, vy = qy[j] - sy[j] * sy[j] / T;
    C[i] = cov / sqrt(vx * vy);          // constant columns give NaN, like numpy.corrcoef
  }
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al. 2011) -> standard normals (Box-Muller).
// counter = (element pair index lo, hi, run, side), key = seed: the stream of a surrogate depends only on
// (seed, run, side), never on which GPU generates it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <typename T>
__global__ void philox_normal_kernel(T* __restrict__ out, int64_t n, uint64_t seed, uint32_t run, uint32_t side) {
  const int64_t pairs = (n + 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), run, side, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const uint64_t a = ((uint64_t)r[0] << 32) | r[1], b = ((uint64_t)r[2] << 32) | r[3];
    const double u1 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);   // (0,1)
    const double u2 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    out[2 * i] = (T)(rad * cs);
    if (2 * i + 1 < n) out[2 * i + 1] = (T)(rad * sn);
  }
}

}  // namespace xmca
