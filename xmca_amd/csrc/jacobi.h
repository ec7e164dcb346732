// Hermitian eigensolver for the small (T x T or N x N) stage of solve():
// two-sided BLOCK Jacobi in f64 with round-robin pair slots (NT x NT tiles: 64 real, 32 complex).
//
//   per round, ONE launch (jacobi_fused_round_kernel; problems of one or two pair slots use the plain
//   jacobi_tile_evd_kernel + jacobi_update_kernel pair):
//     * tile solve - the first S workgroups assemble the diagonal tiles of the NEXT round from this round's inputs
//       (jacobi_assemble_next_diag) and sweep them once with a parallel-order cyclic Jacobi held in LDS
//       (jacobi_tile_evd_body; two-level form jacobi_cross_sweep_twolevel for the 64 x 64 real tiles).  Rotations
//       start from the identity and always take the inner angle, so the accumulated J_P stays close to the identity
//       -> quadratic outer convergence;
//     * update - all workgroups (the first S too, once done) process G'[P,Q] = J_P^H G[P,Q] J_Q (upper triangle, each
//       quarter written once in its upper orientation) and Z'[P,c] = J_P^H Z[P,c] on the f64 matrix pipe
//       (v_mfma_f64_16x16x4_f64) as statically assigned, software-pipelined work items (jacobi_persistent_update),
//       written straight to the slots of the NEXT round (ping-pong buffers), so the tournament permutation costs no
//       extra pass.
//   after 2S-1 rounds every pair of half-blocks has met once (= one sweep); jacobi_offmax_kernel decides when to stop.
//
// Z accumulates Q^H: at the end row i of Z is the conjugated eigenvector i.
// Replaces the LAPACK *gesdd calls of xmca/array.py:479 and :570.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "gemm.h"
#include "cholesky.h"

namespace xmca {

__host__ __device__ inline int jacobi_dest_block(int p, int h, int S) {
  // half h (0 = top, 1 = bottom) of slot p moves to this half-block index for the next round
  if (S == 1) return h;
  if (h == 0) {
    if (p == 0) return 0;
    if (p == S - 1) return 2 * (S - 1) + 1;
    return 2 * (p + 1);
  }
  if (p == 0) return 2;
  return 2 * (p - 1) + 1;
}

// scal[3] += sum of |a_ij|^2 (the Frobenius norm bounds every eigenvalue: the padding has to sit below all of them)
__global__ void jacobi_frobenius_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai, int n, int64_t lda, double* scal) {
  double s = 0.0;
  for (int r = blockIdx.x; r < n; r += gridDim.x)
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
      const double a = Ar[(int64_t)r * lda + c], b = Ai ? Ai[(int64_t)r * lda + c] : 0.0;
      s += a * a + b * b;
    }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(scal + 3, s);
}

// scal[0] = scale of the matrix (max |diag|), scal[1] = absolute rotation floor, scal[2] = diagonal value of the padding
// (below every eigenvalue; from scal[3] = squared Frobenius norm)
__global__ void jacobi_init_scale_kernel(const double* __restrict__ Ar, int n, int64_t lda, double tol, double* scal) {
  __shared__ double red[256];
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmax(m, fabs(Ar[(int64_t)i * lda + i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double g = red[0];
    if (!(g > 0.0)) g = 1.0;
    scal[0] = g;
    scal[1] = 1e-13 * g;   // rotations below this absolute size are rounding noise of the null space
    scal[2] = -2.0 * fmax(sqrt(scal[3]), g);
  }
}

// G0 = [A 0; 0 pad*I], Z0 = I   (npad x npad, planes; pad = scal[2])
__global__ void jacobi_init_kernel(const double* __restrict__ Ar, const double* __restrict__ Ai, int n, int64_t lda,
                                   double* __restrict__ Gr, double* __restrict__ Gi, double* __restrict__ Zr,
                                   double* __restrict__ Zi, int npad, const double* __restrict__ scal) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)npad * npad) return;
  const int r = (int)(idx / npad), c = (int)(idx % npad);
  double gr = 0.0, gi = 0.0;
  if (r < n && c < n) {
    gr = Ar[(int64_t)r * lda + c];
    if (Ai) gi = Ai[(int64_t)r * lda + c];
  } else if (r == c) {
    gr = scal[2];
  }
  Gr[idx] = gr;
  if (Gi) Gi[idx] = gi;
  if (Zr) {      // eigenvectors wanted
    Zr[idx] = (r == c) ? 1.0 : 0.0;
    if (Zi) Zi[idx] = 0.0;
  }
}

#ifdef XMCA_JAC_PROF
// phase stamps of the persistent update (profiling builds only): [workgroup][iteration][stamp]
constexpr int JAC_PROF_IT = 12, JAC_PROF_ST = 10, JAC_PROF_WG = 512;
__device__ long long jac_prof[JAC_PROF_WG * JAC_PROF_IT * JAC_PROF_ST];
#define JAC_STAMP(k)                                                                                                  \
  do {                                                                                                                \
    if (tid == 0 && iter < JAC_PROF_IT && blockIdx.x < JAC_PROF_WG)                                                   \
      jac_prof[((int)blockIdx.x * JAC_PROF_IT + iter) * JAC_PROF_ST + (k)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define JAC_STAMP(k) do { } while (0)
#endif
// threads per workgroup of the tile kernels: 512 for the 64 x 64 real tiles (two waves per SIMD hide the LDS / f64 latency
// of the rotation steps, four per SIMD with two workgroups per CU feed the matrix pipe), 256 for the 32 x 32 tiles
#ifndef XMCA_JAC_THREADS64
#define XMCA_JAC_THREADS64 512
#endif
template <int NT>
constexpr int jac_threads() { return NT >= 64 ? XMCA_JAC_THREADS64 : 256; }

// LDS images of the two kernel bodies (a fused launch runs both kinds of workgroups, so they share one union)
template <int NT, bool CPLX>
struct JacTileSmem {
  double Mr[NT][NT + 1];
  double Vr[NT][NT + 1];
  double Mi[CPLX ? NT : 1][CPLX ? NT + 1 : 1];
  double Vi[CPLX ? NT : 1][CPLX ? NT + 1 : 1];
  double sub[(NT == 64 && !CPLX) ? 4 : 1][16][17];   // two-level cross sweep: accumulated rotations of the four sub-tiles
  double red[8];
  int flag;
};
template <int NT, bool CPLX>
struct JacUpdSmem {
  double Ar[NT][NT + 1], Br[NT][NT + 1];
  double Ai[CPLX ? NT : 1][CPLX ? NT + 1 : 1], Bi[CPLX ? NT : 1][CPLX ? NT + 1 : 1];
};

// ---------------------------------------------------------------------------------------------------------------
// Two-level form of the cross-block sweep of a 64 x 64 real tile (512 threads).  The rotation steps of the flat sweep
// are a serial chain of 32 steps, each a full pass over M and V in LDS with two workgroup barriers.  Here the 32 x 32
// cross pairs are visited as 4 x 4 pairs of 8-index sub-blocks, four disjoint pairs at a time (sub-round r pairs
// sub-block i of the first half with sub-block (i + r) % 4 of the second).  A sub-round:
//   1. waves 0..3 each sweep ONE 16 x 16 sub-tile (8 steps of 8 rotations, one lane per 2 x 2 block, wave-synchronous:
//      no workgroup barrier) in place in M, accumulate its rotations in sub[w] (16 x 16) and put the original entries
//      back;
//   2. all waves apply the four 16 x 16 orthogonal factors to the rows, then to the columns of M and to the columns of
//      V as v_mfma_f64_16x16x4_f64 products on gathered rows / columns (24 MFMAs per wave).
// Every cross pair is rotated exactly once per visit, as in the flat sweep (same convergence: 12 sweeps either way in
// the numpy model of the ordering); the chain per tile visit drops from 32 x ~3.6k to 4 x ~7k cycles.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void jacobi_cross_sweep_twolevel(JacTileSmem<64, false>& sm, const double tol, const double abs_floor) {
  constexpr int H = 32, SB = 8;
  auto& Mr = sm.Mr;
  auto& Vr = sm.Vr;
  auto& sub = sm.sub;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const double floor2 = abs_floor * abs_floor, tol2 = tol * tol;
  for (int r = 0; r < 4; ++r) {
    // tile index of entry a (0..15) of sub-tile w
    auto tix = [r](const int w, const int a) { return a < SB ? SB * w + a : H + SB * ((w + r) & 3) + (a - SB); };
#ifdef XMCA_JAC_PROF
    if (tid == 0) jac_prof[((int)blockIdx.x * JAC_PROF_IT + r + 1) * JAC_PROF_ST + 2] = (long long)__builtin_readcyclecounter();
#endif
    if (wave < 4) {
      const int w = wave, k1 = lane >> 3, k2 = lane & 7;
      double bak[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q;
        sub[w][e >> 4][e & 15] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
        bak[q] = Mr[tix(w, e >> 4)][tix(w, e & 15)];
      }
      __builtin_amdgcn_wave_barrier();
      for (int t = 0; t < SB; ++t) {
        const int a_p2 = k2, a_q2 = SB + ((k2 + t) & 7), a_p1 = k1, a_q1 = SB + ((k1 + t) & 7);
        const int p2 = tix(w, a_p2), q2 = tix(w, a_q2), p1 = tix(w, a_p1), q1 = tix(w, a_q1);
        // rotation of pair k2 (every lane; lane l < 8 holds the rotation of pair l)
        double c2, s2;
        {
          const double app = Mr[p2][p2], aqq = Mr[q2][q2], g = Mr[p2][q2];
          const double g2 = g * g;
          const bool rot = g2 > floor2 && g2 > tol2 * fabs(app * aqq);
          const double d = aqq - app;
          const double x = rot ? d * d + 4.0 * g2 : 1.0;
          const double inv = jac_rcp(fabs(d) + x * jac_rsqrt(x));
          const double wt = (d >= 0.0 ? 2.0 : -2.0) * inv;       // t / |g|
          const double c = jac_rsqrt(1.0 + (rot ? wt * wt * g2 : 0.0));
          c2 = rot ? c : 1.0;
          s2 = rot ? wt * c * g : 0.0;
        }
        const double c1 = __shfl(c2, k1), s1 = __shfl(s2, k1);
        const double b00 = Mr[p1][p2], b01 = Mr[p1][q2], b10 = Mr[q1][p2], b11 = Mr[q1][q2];
        const int jr = lane >> 3;                                   // rows jr and jr + 8 of the accumulated factor
        const double j0p = sub[w][jr][a_p2], j0q = sub[w][jr][a_q2], j1p = sub[w][jr + 8][a_p2], j1q = sub[w][jr + 8][a_q2];
        // rows: x0 = c1 b0 - s1 b1 ; x1 = s1 b0 + c1 b1 ; cols: y_i0 = c2 x_i0 - s2 x_i1 ; y_i1 = s2 x_i0 + c2 x_i1
        const double x00 = c1 * b00 - s1 * b10, x01 = c1 * b01 - s1 * b11;
        const double x10 = s1 * b00 + c1 * b10, x11 = s1 * b01 + c1 * b11;
        double y00 = c2 * x00 - s2 * x01, y01 = s2 * x00 + c2 * x01;
        double y10 = c2 * x10 - s2 * x11, y11 = s2 * x10 + c2 * x11;
        if (k1 == k2) { y01 = 0.0; y10 = 0.0; }
        __builtin_amdgcn_wave_barrier();                            // (a wave issues one instruction for all lanes: reads above, writes below)
        Mr[p1][p2] = y00; Mr[p1][q2] = y01; Mr[q1][p2] = y10; Mr[q1][q2] = y11;
        sub[w][jr][a_p2] = c2 * j0p - s2 * j0q;
        sub[w][jr][a_q2] = s2 * j0p + c2 * j0q;
        sub[w][jr + 8][a_p2] = c2 * j1p - s2 * j1q;
        sub[w][jr + 8][a_q2] = s2 * j1p + c2 * j1q;
        __builtin_amdgcn_wave_barrier();
      }
      // the working copy goes back to what it was: the products below transform whole rows and columns
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q;
        Mr[tix(w, e >> 4)][tix(w, e & 15)] = bak[q];
      }
    }
#ifdef XMCA_JAC_PROF
    if (tid == 0) jac_prof[((int)blockIdx.x * JAC_PROF_IT + r + 1) * JAC_PROF_ST + 3] = (long long)__builtin_readcyclecounter();
#endif
    __syncthreads();
#ifdef XMCA_JAC_PROF
    if (tid == 0) jac_prof[((int)blockIdx.x * JAC_PROF_IT + r + 1) * JAC_PROF_ST + 4] = (long long)__builtin_readcyclecounter();
#endif
    // rows: M[idx_w, ct] <- Js_w^T M[idx_w, ct]   (set w = wave / 2, column tiles 2 (wave % 2) + {0, 1})
    {
      const int w = wave >> 1;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ct = 2 * (wave & 1) + u;
        d4_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int k = 4 * kk + l4;
          acc = Mfma<double>::mma(sub[w][k][l15], Mr[tix(w, k)][ct * 16 + l15], acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) Mr[tix(w, l4 + 4 * q)][ct * 16 + l15] = acc[q];
      }
    }
    __syncthreads();
    // columns: X[rt, idx_w] <- X[rt, idx_w] Js_w for X = M and V   (set w = wave / 2, row tiles 2 (wave % 2) + {0, 1})
    {
      const int w = wave >> 1;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int rt = 2 * (wave & 1) + u;
        d4_t am = {0, 0, 0, 0}, av = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int k = 4 * kk + l4;
          const double js = sub[w][k][l15];
          am = Mfma<double>::mma(Mr[rt * 16 + l15][tix(w, k)], js, am);
          av = Mfma<double>::mma(Vr[rt * 16 + l15][tix(w, k)], js, av);
        }
        const int col = tix(w, l15);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          Mr[rt * 16 + l4 + 4 * q][col] = am[q];
          Vr[rt * 16 + l4 + 4 * q][col] = av[q];
        }
      }
    }
    __syncthreads();
#ifdef XMCA_JAC_PROF
    if (tid == 0) jac_prof[((int)blockIdx.x * JAC_PROF_IT + r + 1) * JAC_PROF_ST + 5] = (long long)__builtin_readcyclecounter();
#endif
  }
}

template <int NT, bool CPLX, bool PRELOADED = false>
__device__ __forceinline__ void jacobi_tile_evd_body(JacTileSmem<NT, CPLX>& sm, const int P, const double* __restrict__ Gr,
                                                     const double* __restrict__ Gi, int ld, double* __restrict__ Jr,
                                                     double* __restrict__ Ji, double* __restrict__ Dr, double* __restrict__ Di,
                                                     double tol, const double* __restrict__ scal,
                                                     unsigned long long* __restrict__ sweep_off, int max_sweeps,
                                                     const bool cross_only, const bool twolevel = false) {
  constexpr int THR = jac_threads<NT>();
  constexpr int NW = THR / 64;
  constexpr int H = NT / 2;
  constexpr int LD = NT + 1;
  auto& Mr = sm.Mr;
  auto& Mi = sm.Mi;
  auto& Vr = sm.Vr;
  auto& Vi = sm.Vi;
  auto& red = sm.red;
  int& flag = sm.flag;

  const int tid = threadIdx.x;
  const double gscale = scal[0], abs_floor = scal[1];
  if constexpr (!PRELOADED) {
    const int64_t base = (int64_t)P * NT * ld + (int64_t)P * NT;
    for (int e = tid; e < NT * NT; e += THR) {
      const int i = e / NT, j = e % NT;
      Mr[i][j] = Gr[base + (int64_t)i * ld + j];
      Vr[i][j] = (i == j) ? 1.0 : 0.0;
      if constexpr (CPLX) {
        Mi[i][j] = Gi[base + (int64_t)i * ld + j];
        Vi[i][j] = 0.0;
      }
    }
    __syncthreads();
  }

  // off-diagonal measure of this tile before it is touched (trace / single-tile problems; the outer loop stops on
  // jacobi_offmax_kernel).  Not on the serial path of the fused rounds: PRELOADED callers skip it.
  if constexpr (!PRELOADED) {
    double mx = 0.0;
    for (int e = tid; e < NT * NT; e += THR) {
      const int i = e / NT, j = e % NT;
      if (i <= j) {
        double g2 = Mr[i][j] * Mr[i][j];
        if constexpr (CPLX) g2 += Mi[i][j] * Mi[i][j];
        if (!(g2 == g2)) mx = HUGE_VAL;                       // NaN in the matrix: reported to the host as +inf
        else if (i < j && g2 > abs_floor * abs_floor) mx = fmax(mx, g2);
      }
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < NW; ++w) mx = fmax(mx, red[w]);   // (mx still holds this wave's = red[0])
      if (mx < HUGE_VAL) mx = sqrt(mx) / gscale;
      if (mx > 0.0) atomicMax(sweep_off, (unsigned long long)__double_as_longlong(mx));
    }
  }

  // Parallel-order cyclic Jacobi.  Thread t owns, for the whole kernel, the column pair k2 = t % H (for the
  // 2x2 blocks (k1, k2) it transforms and for the rows of V it rotates), so the rotation of k2 is read once per
  // step and only the k1 rotations are re-read per block: ~3x fewer LDS operations than a generic item loop.
  constexpr int KSTRIDE = THR / H;            // 8 (NT = 64) or 16 (NT = 32)
  constexpr int NBLK = (H * H) / THR;         // 2x2 blocks per thread: 4 or 1
  constexpr int NROW = NT / KSTRIDE;          // rows of V per thread: 8 or 2
  const int k2 = tid % H, kb = tid / H, lane = tid & 63;
  // full mode: round-robin tournament over all NT indices (NT-1 steps).  cross mode: only the pairs (p, q) with p in
  // the first and q in the second half-block (NT/2 steps of cyclic shifts): the pairs inside a half-block have been
  // rotated when that half-block was last swept in full mode and need it only once per outer sweep.
  __builtin_amdgcn_s_setprio(3);   // latency-bound: when sharing a CU with MFMA-bound update workgroups, issue first
  // The sweep is compiled twice: in cross mode (all rounds but the first of an outer sweep) p = k is fixed, so every
  // row / column base of a thread is loop invariant and the step loses a third of its (integer) instructions.
  auto run_sweeps = [&](auto cross_tag) {
  constexpr bool CROSS = decltype(cross_tag)::value;
  auto pair_of = [](int k, int step, int& p, int& q) {
    if constexpr (CROSS) {
      p = k;
      q = H + ((k + step) & (H - 1));
    } else {
      int a, b;
      if (k == 0) { a = NT - 1; b = step; }
      else { a = step + k; if (a >= NT - 1) a -= NT - 1; b = step - k; if (b < 0) b += NT - 1; }
      p = min(a, b); q = max(a, b);
    }
  };
  const int n_steps = CROSS ? H : NT - 1;
  // One barrier per step.  Every wave computes all H rotations itself (lane l holds the rotation of pair l % H, which
  // is also this thread's column pair k2), so there is no "one wave computes, everybody waits" phase; the rotations of
  // the row pairs k1 come from the lanes that hold them.  The V update of a step does not feed the next rotation
  // angles, so it runs one step late, next to the (latency-bound, scalar) angle computation of the following step.
  const double floor2 = abs_floor * abs_floor, tol2 = tol * tol;
  double cv = 1.0, svr = 0.0, svi = 0.0;   // rotation of the pending V update (identity: nothing pending)
  int pv = 0, qv = 1;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    if (tid == 0) flag = 0;
    __syncthreads();
    for (int step = 0; step < n_steps; ++step) {
#ifdef XMCA_JAC_PROF
      if (PRELOADED && tid == 0 && step >= 1 && step < JAC_PROF_IT) jac_prof[((int)blockIdx.x * JAC_PROF_IT + step) * JAC_PROF_ST + 2] = (long long)__builtin_readcyclecounter();
#endif
      int p2, q2;
      pair_of(k2, step, p2, q2);
      double c2, s2r, s2i;
      {
        const double app = Mr[p2][p2], aqq = Mr[q2][q2];
        const double gr = Mr[p2][q2];
        double gi = 0.0;
        if constexpr (CPLX) gi = Mi[p2][q2];
        const double g2 = gr * gr + gi * gi;
        const bool rot = g2 > floor2 && g2 > tol2 * fabs(app * aqq);
        // t = sign(d) 2|g| / (|d| + sqrt(d^2 + 4|g|^2)),  c = 1/sqrt(1+t^2),  s e^{i phi} = t c g/|g|
        // (hardware rsq/rcp seeds + Newton steps; c^2 + |s|^2 = 1 holds to rounding)
        const double d = aqq - app;
        const double x = rot ? d * d + 4.0 * g2 : 1.0;
        const double inv = jac_rcp(fabs(d) + x * jac_rsqrt(x));
        const double w = (d >= 0.0 ? 2.0 : -2.0) * inv;       // t / |g|
        const double c = jac_rsqrt(1.0 + (rot ? w * w * g2 : 0.0));
        c2 = rot ? c : 1.0;
        s2r = rot ? w * c * gr : 0.0;
        s2i = rot ? w * c * gi : 0.0;
        if (rot) flag = 1;
      }
      // pending V <- V J of the previous step (columns pv, qv of this thread's rows)
      {
        double vr[NROW][2], vi[CPLX ? NROW : 1][2];
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
          const int i = kb + KSTRIDE * r;
          vr[r][0] = Vr[i][pv]; vr[r][1] = Vr[i][qv];
          if constexpr (CPLX) { vi[r][0] = Vi[i][pv]; vi[r][1] = Vi[i][qv]; }
        }
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
          const int i = kb + KSTRIDE * r;
          const double vpr = vr[r][0], vqr = vr[r][1];
          if constexpr (!CPLX) {
            Vr[i][pv] = cv * vpr - svr * vqr;
            Vr[i][qv] = svr * vpr + cv * vqr;
          } else {
            const double vpi = vi[r][0], vqi = vi[r][1];
            // new_p = c vp - conj(sg) vq ; new_q = sg vp + c vq
            Vr[i][pv] = cv * vpr - (svr * vqr + svi * vqi);
            Vi[i][pv] = cv * vpi - (svr * vqi - svi * vqr);
            Vr[i][qv] = (svr * vpr - svi * vpi) + cv * vqr;
            Vi[i][qv] = (svr * vpi + svi * vpr) + cv * vqi;
          }
        }
      }
      cv = c2; svr = s2r; svi = s2i; pv = p2; qv = q2;
#ifdef XMCA_JAC_PROF
      if (PRELOADED && tid == 0 && step >= 1 && step < JAC_PROF_IT) jac_prof[((int)blockIdx.x * JAC_PROF_IT + step) * JAC_PROF_ST + 3] = (long long)__builtin_readcyclecounter();
#endif
      // M <- J^H M J as (NT/2)^2 independent 2x2 blocks.  Everything is read into registers first and written back
      // at the end: the blocks of one thread never overlap, but the compiler cannot know that, and a
      // read-compute-write loop per block costs one LDS round trip per block on the serial path of the solver.  An
      // identity rotation (c = 1, s = 0) reproduces its operands exactly, so no block is skipped.
      double mr[NBLK][4], mi[CPLX ? NBLK : 1][4];
      int p1[NBLK], q1[NBLK];
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        pair_of(kb + KSTRIDE * b, step, p1[b], q1[b]);
        mr[b][0] = Mr[p1[b]][p2]; mr[b][1] = Mr[p1[b]][q2]; mr[b][2] = Mr[q1[b]][p2]; mr[b][3] = Mr[q1[b]][q2];
        if constexpr (CPLX) { mi[b][0] = Mi[p1[b]][p2]; mi[b][1] = Mi[p1[b]][q2]; mi[b][2] = Mi[q1[b]][p2]; mi[b][3] = Mi[q1[b]][q2]; }
      }
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        const int k1 = kb + KSTRIDE * b;
        const int src = k1 + (lane & ~(H - 1) & 31);   // a lane of this wave that holds rotation k1 (lane % H == k1)
        const double cc = __shfl(c2, src), sr = __shfl(s2r, src), si = CPLX ? __shfl(s2i, src) : 0.0;
        const double b00r = mr[b][0], b01r = mr[b][1], b10r = mr[b][2], b11r = mr[b][3];
        if constexpr (!CPLX) {
          // rows: x0 = c1 b0 - s1 b1 ; x1 = s1 b0 + c1 b1
          const double x00 = cc * b00r - sr * b10r, x01 = cc * b01r - sr * b11r;
          const double x10 = sr * b00r + cc * b10r, x11 = sr * b01r + cc * b11r;
          // cols: y_i0 = c2 x_i0 - s2 x_i1 ; y_i1 = s2 x_i0 + c2 x_i1
          double y00 = c2 * x00 - s2r * x01, y01 = s2r * x00 + c2 * x01;
          double y10 = c2 * x10 - s2r * x11, y11 = s2r * x10 + c2 * x11;
          if (k1 == k2) { y01 = 0.0; y10 = 0.0; }
          mr[b][0] = y00; mr[b][1] = y01; mr[b][2] = y10; mr[b][3] = y11;
        } else {
          const double b00i = mi[b][0], b01i = mi[b][1], b10i = mi[b][2], b11i = mi[b][3];
          // x0j = c1 b0j - sg1 b1j ; x1j = conj(sg1) b0j + c1 b1j        (sg = sr + i si)
          const double x00r = cc * b00r - (sr * b10r - si * b10i), x00i = cc * b00i - (sr * b10i + si * b10r);
          const double x01r = cc * b01r - (sr * b11r - si * b11i), x01i = cc * b01i - (sr * b11i + si * b11r);
          const double x10r = (sr * b00r + si * b00i) + cc * b10r, x10i = (sr * b00i - si * b00r) + cc * b10i;
          const double x11r = (sr * b01r + si * b01i) + cc * b11r, x11i = (sr * b01i - si * b01r) + cc * b11i;
          // yi0 = c2 xi0 - conj(sg2) xi1 ; yi1 = sg2 xi0 + c2 xi1
          double y00r = c2 * x00r - (s2r * x01r + s2i * x01i), y00i = c2 * x00i - (s2r * x01i - s2i * x01r);
          double y01r = (s2r * x00r - s2i * x00i) + c2 * x01r, y01i = (s2r * x00i + s2i * x00r) + c2 * x01i;
          double y10r = c2 * x10r - (s2r * x11r + s2i * x11i), y10i = c2 * x10i - (s2r * x11i - s2i * x11r);
          double y11r = (s2r * x10r - s2i * x10i) + c2 * x11r, y11i = (s2r * x10i + s2i * x10r) + c2 * x11i;
          if (k1 == k2) { y01r = y01i = y10r = y10i = 0.0; y00i = 0.0; y11i = 0.0; }
          mr[b][0] = y00r; mr[b][1] = y01r; mr[b][2] = y10r; mr[b][3] = y11r;
          mi[b][0] = y00i; mi[b][1] = y01i; mi[b][2] = y10i; mi[b][3] = y11i;
        }
      }
      // all M reads of this step (also the angle inputs of the other waves) must be done before anything is rewritten
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        Mr[p1[b]][p2] = mr[b][0]; Mr[p1[b]][q2] = mr[b][1]; Mr[q1[b]][p2] = mr[b][2]; Mr[q1[b]][q2] = mr[b][3];
        if constexpr (CPLX) { Mi[p1[b]][p2] = mi[b][0]; Mi[p1[b]][q2] = mi[b][1]; Mi[q1[b]][p2] = mi[b][2]; Mi[q1[b]][q2] = mi[b][3]; }
      }
#ifdef XMCA_JAC_PROF
      if (PRELOADED && tid == 0 && step >= 1 && step < JAC_PROF_IT) jac_prof[((int)blockIdx.x * JAC_PROF_IT + step) * JAC_PROF_ST + 4] = (long long)__builtin_readcyclecounter();
#endif
      __syncthreads();
#ifdef XMCA_JAC_PROF
      if (PRELOADED && tid == 0 && step >= 1 && step < JAC_PROF_IT) jac_prof[((int)blockIdx.x * JAC_PROF_IT + step) * JAC_PROF_ST + 5] = (long long)__builtin_readcyclecounter();
#endif
    }
    const int f = flag;
    __syncthreads();
    if (!f) break;
  }
  // the V update of the very last step
  {
#pragma unroll
    for (int r = 0; r < NROW; ++r) {
      const int i = kb + KSTRIDE * r;
      const double vpr = Vr[i][pv], vqr = Vr[i][qv];
      if constexpr (!CPLX) {
        Vr[i][pv] = cv * vpr - svr * vqr;
        Vr[i][qv] = svr * vpr + cv * vqr;
      } else {
        const double vpi = Vi[i][pv], vqi = Vi[i][qv];
        Vr[i][pv] = cv * vpr - (svr * vqr + svi * vqi);
        Vi[i][pv] = cv * vpi - (svr * vqi - svi * vqr);
        Vr[i][qv] = (svr * vpr - svi * vpi) + cv * vqr;
        Vi[i][qv] = (svr * vpi + svi * vpr) + cv * vqi;
      }
    }
    __syncthreads();
  }

  };
  if constexpr (NT == 64 && !CPLX && jac_threads<NT>() == 512) {
    static_assert(sizeof(sm.sub) >= sizeof(double) * 4 * 16 * 17, "");
    if (cross_only && max_sweeps == 1 && twolevel) {
      jacobi_cross_sweep_twolevel(sm, tol, abs_floor);
    } else if (cross_only) {
      run_sweeps(std::true_type{});
    } else {
      run_sweeps(std::false_type{});
    }
  } else {
    if (cross_only) run_sweeps(std::true_type{});
    else run_sweeps(std::false_type{});
  }

#ifdef XMCA_JAC_PROF
  if (PRELOADED && tid == 0) jac_prof[((int)blockIdx.x * JAC_PROF_IT) * JAC_PROF_ST + 6] = (long long)__builtin_readcyclecounter();
#endif
  const int64_t jb = (int64_t)P * NT * NT;
  for (int e = tid; e < NT * NT; e += THR) {
    const int i = e / NT, j = e % NT;
    Jr[jb + e] = Vr[i][j];
    Dr[jb + e] = Mr[i][j];          // J^H M J: diagonal only when the tile was swept to convergence
    if constexpr (CPLX) { Ji[jb + e] = Vi[i][j]; Di[jb + e] = (i == j) ? 0.0 : Mi[i][j]; }
  }
}

#ifndef XMCA_JAC_ZW
#define XMCA_JAC_ZW 4
#endif
constexpr int JAC_ZW = XMCA_JAC_ZW;   // eigenvector tiles (NT x NT) handled per workgroup, sharing one J_P
#ifdef XMCA_JAC_NT_STORE
#define JAC_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define JAC_STORE(ptr, val) (*(ptr) = (val))
#endif

// One round of the two-sided update, one workgroup per tile (the plain form, used for problems of one or two pair
// slots; the fused rounds below use jacobi_persistent_update):  G'[P,Q] = J_P^H G[P,Q] J_Q (upper tiles + mirrored
// write), Z'[P,c] = J_P^H Z[P,c], both written to the slots of the next round.
// Two LDS buffers (J and tile) so that two workgroups fit a CU; results are staged through LDS and leave as
// full 256-byte row segments (also the mirrored, transposed copy).
template <int NT, bool CPLX>
__device__ __forceinline__ void jacobi_update_body(JacUpdSmem<NT, CPLX>& sm, const int block_id, const double* __restrict__ Gr_in,
                                                   const double* __restrict__ Gi_in, double* __restrict__ Gr_out,
                                                   double* __restrict__ Gi_out, const double* __restrict__ Zr_in,
                                                   const double* __restrict__ Zi_in, double* __restrict__ Zr_out,
                                                   double* __restrict__ Zi_out, const double* __restrict__ Jr,
                                                   const double* __restrict__ Ji, const double* __restrict__ Dr,
                                                   const double* __restrict__ Di, int S, int ld, const bool upper_store) {
  constexpr int LD = NT + 1;
  constexpr int HB = NT / 2;
  constexpr int THR = jac_threads<NT>();
  constexpr int TPD = NT / 16;          // MFMA tiles per dimension
  constexpr int NACC = TPD * TPD / (THR / 64);   // output tiles per wave
  constexpr int EPT = NT * NT / THR;    // tile elements per thread
  auto& Ar = sm.Ar;
  auto& Br = sm.Br;
  auto& Ai = sm.Ai;
  auto& Bi = sm.Bi;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int n_off = S * (S - 1) / 2;
  const int zchunks = (S + JAC_ZW - 1) / JAC_ZW;
  int kind, P, Q;   // kind 0: diagonal tile, 1: off-diagonal G tile, 2: Z chunk
  {
    int id = block_id;
    if (id < S) { kind = 0; P = Q = id; }
    else {
      id -= S;
      if (id < n_off) {
        kind = 1;
        P = 0;
        int rem = id;
        while (rem >= S - 1 - P) { rem -= S - 1 - P; ++P; }
        Q = P + 1 + rem;
      } else {
        id -= n_off;
        kind = 2;
        P = id / zchunks;
        Q = (id % zchunks) * JAC_ZW;
      }
    }
  }

  if (kind == 0) {
    // the diagonal tile was transformed by the tile solver itself (J_P^H G[P,P] J_P); move it to its destination blocks
    // upper_store: nothing ever reads a half-block below the block diagonal again (tiles are taken from the upper
    // triangle, diagonal tiles from D), so its off-diagonal quarter is written once, in whichever orientation is upper
    for (int e = tid; e < NT * NT; e += THR) {
      const int r = e / NT, c = e % NT;
      const int br = jacobi_dest_block(P, r / HB, S), bc = jacobi_dest_block(P, c / HB, S);
      if (upper_store && br > bc) continue;
      const int dr = br * HB + r % HB, dc = bc * HB + c % HB;
      const int64_t o = (int64_t)dr * ld + dc;
      Gr_out[o] = Dr[(int64_t)P * NT * NT + e];
      if constexpr (CPLX) Gi_out[o] = Di[(int64_t)P * NT * NT + e];
    }
    return;
  }

  const bool is_g = (kind == 1);
  const int64_t jpb = (int64_t)P * NT * NT, jqb = (int64_t)Q * NT * NT;
  double jqr[EPT], jqi[CPLX ? EPT : 1];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = tid + THR * i;
    Ar[e / NT][e % NT] = Jr[jpb + e];
    if constexpr (CPLX) Ai[e / NT][e % NT] = Ji[jpb + e];
    if (is_g) {
      jqr[i] = Jr[jqb + e];
      if constexpr (CPLX) jqi[i] = Ji[jqb + e];
    }
  }
  const double* __restrict__ Sr = is_g ? Gr_in : Zr_in;
  const double* __restrict__ Si = is_g ? Gi_in : Zi_in;
  const int nsub = is_g ? 1 : min(JAC_ZW, S - Q);

  for (int sub = 0; sub < nsub; ++sub) {
    const int Qc = Q + sub;
    const int64_t tbase = (int64_t)P * NT * ld + (int64_t)Qc * NT;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int e = tid + THR * i, r = e / NT, c = e % NT;
      Br[r][c] = Sr[tbase + (int64_t)r * ld + c];
      if constexpr (CPLX) Bi[r][c] = Si[tbase + (int64_t)r * ld + c];
    }
    __syncthreads();

    // X = J_P^H T
    d4_t xr[NACC], xi[CPLX ? NACC : 1];
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
      d4_t ar = {0, 0, 0, 0}, ai = {0, 0, 0, 0};
#pragma unroll 4
      for (int k0 = 0; k0 < NT; k0 += 4) {
        const int k = k0 + l4;
        const double jr = Ar[k][ti * 16 + l15];
        const double tr = Br[k][tj * 16 + l15];
        ar = Mfma<double>::mma(jr, tr, ar);
        if constexpr (CPLX) {
          const double ji = Ai[k][ti * 16 + l15];
          const double tim = Bi[k][tj * 16 + l15];
          ar = Mfma<double>::mma(ji, tim, ar);     // + JPi^T Ti
          ai = Mfma<double>::mma(jr, tim, ai);     // + JPr^T Ti
          ai = Mfma<double>::mma(-ji, tr, ai);     // - JPi^T Tr
        }
      }
      xr[a] = ar;
      if constexpr (CPLX) xi[a] = ai;
    }
    __syncthreads();   // every wave is done reading the tile (and J_P when this is a G tile)
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + Mfma<double>::row(lane, r), col = tj * 16 + l15;
        Br[row][col] = xr[a][r];
        if constexpr (CPLX) Bi[row][col] = xi[a][r];
      }
    }
    if (is_g) {
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int e = tid + THR * i;
        Ar[e / NT][e % NT] = jqr[i];
        if constexpr (CPLX) Ai[e / NT][e % NT] = jqi[i];
      }
    }
    __syncthreads();

    if (!is_g) {
      // Z'[dest(P,h) rows, chunk Qc] = X   (row segments of NT doubles)
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int e = tid + THR * i, r = e / NT, c = e % NT;
        const int dr = jacobi_dest_block(P, r / HB, S) * HB + r % HB;
        const int64_t o = (int64_t)dr * ld + (int64_t)Qc * NT + c;
        JAC_STORE(&Zr_out[o], Br[r][c]);
        if constexpr (CPLX) JAC_STORE(&Zi_out[o], Bi[r][c]);
      }
      __syncthreads();   // the next sub-tile overwrites B
      continue;
    }

    // Y = X J_Q
    d4_t yr[NACC], yi[CPLX ? NACC : 1];
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
      d4_t ar = {0, 0, 0, 0}, ai = {0, 0, 0, 0};
#pragma unroll 4
      for (int k0 = 0; k0 < NT; k0 += 4) {
        const int k = k0 + l4;
        const double xre = Br[ti * 16 + l15][k];
        const double qr = Ar[k][tj * 16 + l15];
        ar = Mfma<double>::mma(xre, qr, ar);
        if constexpr (CPLX) {
          const double xim = Bi[ti * 16 + l15][k];
          const double qi = Ai[k][tj * 16 + l15];
          ar = Mfma<double>::mma(-xim, qi, ar);
          ai = Mfma<double>::mma(xre, qi, ai);
          ai = Mfma<double>::mma(xim, qr, ai);
        }
      }
      yr[a] = ar;
      if constexpr (CPLX) yi[a] = ai;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + Mfma<double>::row(lane, r), col = tj * 16 + l15;
        Br[row][col] = yr[a][r];
        if constexpr (CPLX) Bi[row][col] = yi[a][r];
      }
    }
    __syncthreads();
    // scatter to the next round's slots: the tile itself (rows) and its Hermitian mirror (columns read from LDS)
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int e = tid + THR * i, r = e / NT, c = e % NT;
      const int br = jacobi_dest_block(P, r / HB, S), bc = jacobi_dest_block(Q, c / HB, S);
      if (!upper_store || br < bc) {
        const int dr = br * HB + r % HB, dc = bc * HB + c % HB;
        JAC_STORE(&Gr_out[(int64_t)dr * ld + dc], Br[r][c]);
        if constexpr (CPLX) JAC_STORE(&Gi_out[(int64_t)dr * ld + dc], Bi[r][c]);
      }
      // mirrored element: this thread now plays (row' = c-index, col' = r-index) with r fastest
      const int r2 = e % NT, c2 = e / NT;
      const int br2 = jacobi_dest_block(P, r2 / HB, S), bc2 = jacobi_dest_block(Q, c2 / HB, S);
      if (!upper_store || br2 > bc2) {
        const int dr2 = br2 * HB + r2 % HB, dc2 = bc2 * HB + c2 % HB;
        JAC_STORE(&Gr_out[(int64_t)dc2 * ld + dr2], Br[r2][c2]);
        if constexpr (CPLX) JAC_STORE(&Gi_out[(int64_t)dc2 * ld + dr2], -Bi[r2][c2]);
      }
    }
  }
}

// The two half-blocks that form pair slot Pn in the NEXT round: (A, hA) becomes its top, (B, hB) its bottom half
// (inverse of jacobi_dest_block; S >= 3).
__host__ __device__ inline void jacobi_next_diag_halves(int Pn, int S, int& A, int& hA, int& B, int& hB) {
  if (Pn == 0) { A = 0; hA = 0; B = 1; hB = 1; }
  else if (Pn == S - 1) { A = S - 2; hA = 0; B = S - 1; hB = 0; }
  else if (Pn == 1) { A = 0; hA = 1; B = 2; hB = 1; }
  else { A = Pn - 1; hA = 0; B = Pn + 1; hB = 1; }
}

// Builds, in LDS, the diagonal tile that pair slot Pn will hold in the next round - without waiting for the update of
// the current round to be written: its diagonal quarters are quarters of the transformed diagonal tiles D_A, D_B of
// this round and its off-diagonal quarter is the (hA, hB) quarter of J_A^H G[A,B] J_B, recomputed here (1/3 of a tile
// update).  This removes the separate "head" launch from every round.
template <int NT, bool CPLX>
__device__ __forceinline__ void jacobi_assemble_next_diag(JacTileSmem<NT, CPLX>& st, JacUpdSmem<NT, CPLX>& su, const int Pn, const int S,
                                                          const double* __restrict__ Gr_in, const double* __restrict__ Gi_in,
                                                          const int ld, const double* __restrict__ Jr, const double* __restrict__ Ji,
                                                          const double* __restrict__ Dr, const double* __restrict__ Di) {
  constexpr int THR = jac_threads<NT>();
  constexpr int NW = THR / 64;
  constexpr int HB = NT / 2;
  constexpr int XT = (HB / 16) * (NT / 16);     // MFMA tiles of X (HB x NT): 8 or 2
  constexpr int XPW = (XT + NW - 1) / NW;       // per wave
  constexpr int YT = (HB / 16) * (HB / 16);     // MFMA tiles of Yq (HB x HB): 4 or 1
  constexpr int EPT = NT * NT / THR;            // elements per thread of a full tile
  constexpr int EPH = EPT / 2;                  // ... of a half tile (NT x HB) / of two quarters
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  int A, hA, B, hB;
  jacobi_next_diag_halves(Pn, S, A, hA, B, hB);
  const int64_t ja = (int64_t)A * NT * NT, jb = (int64_t)B * NT * NT;
  const int64_t tbase = (int64_t)A * NT * ld + (int64_t)B * NT;
  // This chain (assemble -> sweep -> next round's assemble) is the serial path of the whole solver: every global
  // operand is requested before the first one is used, so one memory latency is exposed instead of three.
  double tr[EPT], ti[CPLX ? EPT : 1];           // T = G[A,B]
  double jar[EPH], jai[CPLX ? EPH : 1];         // J_A[:, hA half]   (NT x HB)
  double jbr[EPH], jbi[CPLX ? EPH : 1];         // J_B[:, hB half]
  double dar[EPH / 2], dai[CPLX ? EPH / 2 : 1]; // D_A[hA, hA] quarter (HB x HB)
  double dbr[EPH / 2], dbi[CPLX ? EPH / 2 : 1]; // D_B[hB, hB] quarter
#pragma unroll
  for (int i = 0; i < EPH; ++i) {
    const int e = tid + THR * i, r = e / HB, c = e % HB;
    jar[i] = Jr[ja + (int64_t)r * NT + hA * HB + c];
    if constexpr (CPLX) jai[i] = Ji[ja + (int64_t)r * NT + hA * HB + c];
  }
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = tid + THR * i, r = e / NT, c = e % NT;
    tr[i] = Gr_in[tbase + (int64_t)r * ld + c];
    if constexpr (CPLX) ti[i] = Gi_in[tbase + (int64_t)r * ld + c];
  }
#pragma unroll
  for (int i = 0; i < EPH; ++i) {
    const int e = tid + THR * i, r = e / HB, c = e % HB;
    jbr[i] = Jr[jb + (int64_t)r * NT + hB * HB + c];
    if constexpr (CPLX) jbi[i] = Ji[jb + (int64_t)r * NT + hB * HB + c];
  }
#pragma unroll
  for (int i = 0; i < EPH / 2; ++i) {
    const int e = tid + THR * i, r = e / HB, c = e % HB;
    const int64_t oa = ja + (int64_t)(hA * HB + r) * NT + hA * HB + c, ob = jb + (int64_t)(hB * HB + r) * NT + hB * HB + c;
    dar[i] = Dr[oa];
    dbr[i] = Dr[ob];
    if constexpr (CPLX) { dai[i] = Di[oa]; dbi[i] = Di[ob]; }
  }
  // A[:, 0..HB) <- J_A half, B <- T
#pragma unroll
  for (int i = 0; i < EPH; ++i) {
    const int e = tid + THR * i, r = e / HB, c = e % HB;
    su.Ar[r][c] = jar[i];
    if constexpr (CPLX) su.Ai[r][c] = jai[i];
  }
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = tid + THR * i, r = e / NT, c = e % NT;
    su.Br[r][c] = tr[i];
    if constexpr (CPLX) su.Bi[r][c] = ti[i];
  }
  __syncthreads();
  // X = J_A[:, hA]^H T      (HB x NT)
  d4_t xr[XPW], xi[CPLX ? XPW : 1];
#pragma unroll
  for (int a = 0; a < XPW; ++a) {
    const int t = wave * XPW + a;
    d4_t ar = {0, 0, 0, 0}, ai = {0, 0, 0, 0};
    if (t < XT) {
      const int tii = t / (NT / 16), tj = t % (NT / 16);
#pragma unroll 4
      for (int k0 = 0; k0 < NT; k0 += 4) {
        const int k = k0 + l4;
        const double jr = su.Ar[k][tii * 16 + l15];
        const double trr = su.Br[k][tj * 16 + l15];
        ar = Mfma<double>::mma(jr, trr, ar);
        if constexpr (CPLX) {
          const double ji = su.Ai[k][tii * 16 + l15];
          const double tim = su.Bi[k][tj * 16 + l15];
          ar = Mfma<double>::mma(ji, tim, ar);
          ai = Mfma<double>::mma(jr, tim, ai);
          ai = Mfma<double>::mma(-ji, trr, ai);
        }
      }
    }
    xr[a] = ar;
    if constexpr (CPLX) xi[a] = ai;
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < XPW; ++a) {
    const int t = wave * XPW + a;
    if (t < XT) {
      const int tii = t / (NT / 16), tj = t % (NT / 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = tii * 16 + Mfma<double>::row(lane, r), col = tj * 16 + l15;
        su.Br[row][col] = xr[a][r];
        if constexpr (CPLX) su.Bi[row][col] = xi[a][r];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < EPH; ++i) {
    const int e = tid + THR * i, r = e / HB, c = e % HB;
    su.Ar[r][c] = jbr[i];
    if constexpr (CPLX) su.Ai[r][c] = jbi[i];
  }
  __syncthreads();
  // Yq = X J_B[:, hB]      (HB x HB), one MFMA tile per wave
  d4_t yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0};
  const int yti = wave / (HB / 16), ytj = wave % (HB / 16);
  if (wave < YT) {
#pragma unroll 4
    for (int k0 = 0; k0 < NT; k0 += 4) {
      const int k = k0 + l4;
      const double xre = su.Br[yti * 16 + l15][k];
      const double qr = su.Ar[k][ytj * 16 + l15];
      yr = Mfma<double>::mma(xre, qr, yr);
      if constexpr (CPLX) {
        const double xim = su.Bi[yti * 16 + l15][k];
        const double qi = su.Ai[k][ytj * 16 + l15];
        yr = Mfma<double>::mma(-xim, qi, yr);
        yi = Mfma<double>::mma(xre, qi, yi);
        yi = Mfma<double>::mma(xim, qr, yi);
      }
    }
  }
  __syncthreads();   // the update-shaped buffers are dead from here: the tile image overwrites them
  if (wave < YT) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = yti * 16 + Mfma<double>::row(lane, r), col = ytj * 16 + l15;
      st.Mr[row][HB + col] = yr[r];
      st.Mr[HB + col][row] = yr[r];
      if constexpr (CPLX) {
        st.Mi[row][HB + col] = yi[r];
        st.Mi[HB + col][row] = -yi[r];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < EPH / 2; ++i) {
    const int e = tid + THR * i, r = e / HB, c = e % HB;
    st.Mr[r][c] = dar[i];
    st.Mr[HB + r][HB + c] = dbr[i];
    if constexpr (CPLX) {
      st.Mi[r][c] = dai[i];
      st.Mi[HB + r][HB + c] = dbi[i];
    }
  }
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = tid + THR * i;
    st.Vr[e / NT][e % NT] = (e / NT == e % NT) ? 1.0 : 0.0;
    if constexpr (CPLX) st.Vi[e / NT][e % NT] = 0.0;
  }
  __syncthreads();
}

template <int NT, bool CPLX>
__global__ __launch_bounds__(jac_threads<NT>(), 2 * jac_threads<NT>() / 256) void jacobi_tile_evd_kernel(const double* Gr, const double* Gi, int ld, double* Jr, double* Ji,
                                                                 double* Dr, double* Di, double tol, const double* scal,
                                                                 unsigned long long* sweep_off, int max_sweeps, int cross_only) {
  __shared__ JacTileSmem<NT, CPLX> sm;
  jacobi_tile_evd_body<NT, CPLX>(sm, blockIdx.x, Gr, Gi, ld, Jr, Ji, Dr, Di, tol, scal, sweep_off, max_sweeps, (cross_only & 1) != 0, (cross_only & 2) != 0);
}

template <int NT, bool CPLX>
__global__ __launch_bounds__(jac_threads<NT>(), 2 * jac_threads<NT>() / 256) void jacobi_update_kernel(const double* Gr_in, const double* Gi_in, double* Gr_out,
                                                               double* Gi_out, const double* Zr_in, const double* Zi_in,
                                                               double* Zr_out, double* Zi_out, const double* Jr, const double* Ji,
                                                               const double* Dr, const double* Di, int S, int ld) {
  __shared__ JacUpdSmem<NT, CPLX> sm;
  jacobi_update_body<NT, CPLX>(sm, blockIdx.x, Gr_in, Gi_in, Gr_out, Gi_out, Zr_in, Zi_in, Zr_out, Zi_out, Jr, Ji, Dr, Di, S,
                                     ld, false);
}

// ---- persistent, software-pipelined form of the update (fused round kernel) --------------------------------
// Work items of a round, taken from an atomic counter (heaviest first):
//   kind 1: G tile (P,Q), P < Q          J_P^H G[P,Q] J_Q      loads J_P, T, J_Q   - two tile products
//   kind 2: eigenvector tiles (P, Q..Q+1) J_P^H Z[P,Q..]       loads J_P, T0, T1   - two tile products
//   kind 0: diagonal tile P              move D_P to its destination quarters
// While an item is in the MFMA / LDS / store phases the three tiles of the NEXT item are already in flight into
// registers, so the HBM/L2 latency that the one-item-per-workgroup form exposes (waves idle 40 % of the time on
// s_waitcnt, MI355X PMC) hides behind arithmetic.
struct JacItem {
  int kind, P, Q, nsub;
};

__device__ __forceinline__ JacItem jacobi_decode_item(int id, const int S, const int n_off, const int zch, const int zw) {
  JacItem it;
  it.kind = -1; it.P = 0; it.Q = 0; it.nsub = 0;
  if (id < 0) return it;
  if (id < n_off) {
    it.kind = 1;
    // row P of the strictly upper triangle starts at off(P) = P (2S - P - 1) / 2: closed form + fix-up (a search loop
    // here costs a few hundred cycles per item on the scalar unit)
    const int b = 2 * S - 1;
    int P = (int)(0.5f * ((float)b - sqrtf((float)(b * b - 8 * id))));
    P = max(0, min(P, S - 2));
    while (P > 0 && id < P * (2 * S - P - 1) / 2) --P;
    while (id >= (P + 1) * (2 * S - P - 2) / 2) ++P;
    it.P = P; it.Q = P + 1 + (id - P * (2 * S - P - 1) / 2);
    return it;
  }
  id -= n_off;
  if (id < S * zch) {
    it.kind = 2;
    it.P = id / zch;
    it.Q = (id % zch) * zw;
    it.nsub = min(zw, S - it.Q);
    return it;
  }
  return it;      // past the end: kind -1
}

typedef unsigned int jac_u32x2 __attribute__((ext_vector_type(2)));
// buffer addressing: one SGPR descriptor per plane, a per-thread byte offset that is the same for every tile of an
// item, and a wave-uniform (SGPR) byte offset per access - no 64-bit per-access address VGPRs, which is what lets three
// prefetched tiles + the accumulators fit the 256 VGPRs of a 2-workgroup-per-CU kernel
__device__ __forceinline__ __amdgpu_buffer_rsrc_t jac_rsrc(const double* p, unsigned int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double jac_ld(__amdgpu_buffer_rsrc_t rs, unsigned int voff, unsigned int soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}
__device__ __forceinline__ void jac_st(double x, __amdgpu_buffer_rsrc_t rs, unsigned int voff, unsigned int soff) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(jac_u32x2, x), rs, voff, soff, 0);
}

template <int NT, bool CPLX>
__device__ __forceinline__ void jacobi_persistent_update(JacUpdSmem<NT, CPLX>& sm, int* slot, unsigned int* __restrict__ counter,
                                                         const double* __restrict__ Gr_in, const double* __restrict__ Gi_in,
                                                         double* __restrict__ Gr_out, double* __restrict__ Gi_out,
                                                         const double* __restrict__ Zr_in, const double* __restrict__ Zi_in,
                                                         double* __restrict__ Zr_out, double* __restrict__ Zi_out,
                                                         const double* __restrict__ Jr, const double* __restrict__ Ji,
                                                         const double* __restrict__ Dr, const double* __restrict__ Di, const int S,
                                                         const int ld, const int zch, const int n_static, const int worker,
                                                         const int n_workers) {
  constexpr int HB = NT / 2;
  constexpr int THR = jac_threads<NT>();
  constexpr int TPD = NT / 16;
  constexpr int NACC = TPD * TPD / (THR / 64);
  constexpr int EPT = NT * NT / THR;   // tile elements per thread = passes over the tile
  constexpr int RP = THR / NT;         // tile rows covered by one pass
  static_assert(HB % RP == 0, "a pass must not straddle the two half-blocks");
  auto& Ar = sm.Ar;
  auto& Br = sm.Br;
  auto& Ai = sm.Ai;
  auto& Bi = sm.Bi;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int row0 = tid / NT, col = tid % NT;
  const int n_off = S * (S - 1) / 2;

  const unsigned int plane_bytes = (unsigned int)((size_t)ld * ld * sizeof(double));
  const unsigned int j_bytes = (unsigned int)((size_t)S * NT * NT * sizeof(double));
  const __amdgpu_buffer_rsrc_t rGr = jac_rsrc(Gr_in, plane_bytes), rGi = jac_rsrc(Gi_in, plane_bytes);
  const __amdgpu_buffer_rsrc_t rZr = jac_rsrc(Zr_in, plane_bytes), rZi = jac_rsrc(Zi_in, plane_bytes);
  const __amdgpu_buffer_rsrc_t oGr = jac_rsrc(Gr_out, plane_bytes), oGi = jac_rsrc(Gi_out, plane_bytes);
  const __amdgpu_buffer_rsrc_t oZr = jac_rsrc(Zr_out, plane_bytes), oZi = jac_rsrc(Zi_out, plane_bytes);
  const __amdgpu_buffer_rsrc_t rJr = jac_rsrc(Jr, j_bytes), rJi = jac_rsrc(Ji, j_bytes);
  const __amdgpu_buffer_rsrc_t rDr = jac_rsrc(Dr, j_bytes), rDi = jac_rsrc(Di, j_bytes);
  const unsigned int voff_T = (unsigned int)(row0 * ld + col) * 8u;   // element (row0, col) of a tile inside a plane
  const unsigned int voff_J = (unsigned int)tid * 8u;                 // the same element of a packed NT x NT tile

  struct Tile {
    double r[EPT];
    double i[CPLX ? EPT : 1];
  };
  Tile tP, tT, tQ;
  // tile whose (0,0) element is at element offset `base`; consecutive passes are `pass` elements apart (both uniform)
  auto fetch = [&](Tile& t, const __amdgpu_buffer_rsrc_t rr, const __amdgpu_buffer_rsrc_t ri, const unsigned int voff,
                   const unsigned int base, const unsigned int pass) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const unsigned int soff = (base + (unsigned int)i * pass) * 8u;
      t.r[i] = jac_ld(rr, voff, soff);
      if constexpr (CPLX) t.i[i] = jac_ld(ri, voff, soff);
    }
  };
  // Operand tiles (J_P, J_Q, T) are read by the MFMA loops as S[k][16-column window] with k = k0 + lane / 16: with the
  // plain pitch NT + 1 the two k rows of a 32-lane group start one bank pair apart and collide 2-way.  Odd rows are
  // therefore stored rotated by 16 columns (rcol): the two windows then lie 16 / 17 bank pairs apart.  The products
  // X, Y written back into B keep the plain layout (they are read row-wise, where the odd pitch is what is wanted).
  auto rcol = [](const int k, const int c) { return (c + 16 * (k & 1)) & (NT - 1); };
  auto to_A = [&](const Tile& t) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int rr = row0 + RP * i;
      Ar[rr][rcol(rr, col)] = t.r[i];
      if constexpr (CPLX) Ai[rr][rcol(rr, col)] = t.i[i];
    }
  };
  auto to_B = [&](const Tile& t) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int rr = row0 + RP * i;
      Br[rr][rcol(rr, col)] = t.r[i];
      if constexpr (CPLX) Bi[rr][rcol(rr, col)] = t.i[i];
    }
  };
  auto issue_PT = [&](const JacItem& it) {
    if (it.kind <= 0) return;
    fetch(tP, rJr, rJi, voff_J, (unsigned int)(it.P * NT * NT), (unsigned int)THR);
    const unsigned int tb = (unsigned int)(it.P * NT * ld + it.Q * NT);
    if (it.kind == 1) fetch(tT, rGr, rGi, voff_T, tb, (unsigned int)(RP * ld));
    else fetch(tT, rZr, rZi, voff_T, tb, (unsigned int)(RP * ld));
  };
  auto issue_Q = [&](const JacItem& it) {
    if (it.kind == 1) fetch(tQ, rJr, rJi, voff_J, (unsigned int)(it.Q * NT * NT), (unsigned int)THR);
  };
  // B <- A^H B   (A = J_P, B = tile), through registers
  auto mul_AhB = [&]() {
    d4_t xr[NACC], xi[CPLX ? NACC : 1];
    {
      // complex: three real products instead of four (the f64 matrix pipe is the bound of the update):
      //   (a - ib)(c + id):  k1 = ac, k2 = bd, k3 = (a + b)(d - c)  ->  Re = k1 + k2,  Im = k3 + k1 - k2
      d4_t k1[NACC], k2[CPLX ? NACC : 1], k3[CPLX ? NACC : 1];
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        k1[a] = d4_t{0, 0, 0, 0};
        if constexpr (CPLX) { k2[a] = d4_t{0, 0, 0, 0}; k3[a] = d4_t{0, 0, 0, 0}; }
      }
      const int rot = 16 * (l4 & 1);      // k = k0 + l4 and k0 is a multiple of 4
#pragma unroll 4
      for (int k0 = 0; k0 < NT; k0 += 4) {
        const int k = k0 + l4;
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
          const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
          const int ca = (ti * 16 + l15 + rot) & (NT - 1), cb = (tj * 16 + l15 + rot) & (NT - 1);
          const double jr = Ar[k][ca];          // (the tiles of a wave share ti: one read after CSE)
          const double tr = Br[k][cb];
          k1[a] = Mfma<double>::mma(jr, tr, k1[a]);
          if constexpr (CPLX) {
            const double ji = Ai[k][ca];
            const double tim = Bi[k][cb];
            k2[a] = Mfma<double>::mma(ji, tim, k2[a]);
            k3[a] = Mfma<double>::mma(jr + ji, tim - tr, k3[a]);
          }
        }
      }
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        if constexpr (CPLX) {
          xr[a] = k1[a] + k2[a];
          xi[a] = k3[a] + k1[a] - k2[a];
        } else {
          xr[a] = k1[a];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + Mfma<double>::row(lane, r), cc = tj * 16 + l15;
        Br[row][cc] = xr[a][r];
        if constexpr (CPLX) Bi[row][cc] = xi[a][r];
      }
    }
  };
  // B <- B A   (B = X, A = J_Q)
  auto mul_BA = [&]() {
    d4_t yr[NACC], yi[CPLX ? NACC : 1];
    {
      //   (a + ib)(c + id):  k1 = ac, k2 = bd, k3 = (a + b)(c + d)  ->  Re = k1 - k2,  Im = k3 - k1 - k2
      d4_t k1[NACC], k2[CPLX ? NACC : 1], k3[CPLX ? NACC : 1];
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        k1[a] = d4_t{0, 0, 0, 0};
        if constexpr (CPLX) { k2[a] = d4_t{0, 0, 0, 0}; k3[a] = d4_t{0, 0, 0, 0}; }
      }
      const int rot = 16 * (l4 & 1);
#pragma unroll 4
      for (int k0 = 0; k0 < NT; k0 += 4) {
        const int k = k0 + l4;
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
          const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
          const int cq = (tj * 16 + l15 + rot) & (NT - 1);
          const double xre = Br[ti * 16 + l15][k];       // X: plain layout, read row-wise
          const double qr = Ar[k][cq];
          k1[a] = Mfma<double>::mma(xre, qr, k1[a]);
          if constexpr (CPLX) {
            const double xim = Bi[ti * 16 + l15][k];
            const double qi = Ai[k][cq];
            k2[a] = Mfma<double>::mma(xim, qi, k2[a]);
            k3[a] = Mfma<double>::mma(xre + xim, qr + qi, k3[a]);
          }
        }
      }
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        if constexpr (CPLX) {
          yr[a] = k1[a] - k2[a];
          yi[a] = k3[a] - k1[a] - k2[a];
        } else {
          yr[a] = k1[a];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + Mfma<double>::row(lane, r), cc = tj * 16 + l15;
        Br[row][cc] = yr[a][r];
        if constexpr (CPLX) Bi[row][cc] = yi[a][r];
      }
    }
  };
  // rows of pass i belong to half (RP*i)/HB of the tile; inside the destination half-block they start at (RP*i)%HB
  auto store_z = [&](const int P, const int Qc) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int drow = jacobi_dest_block(P, (RP * i) / HB, S) * HB + (RP * i) % HB;
      const unsigned int soff = (unsigned int)(drow * ld + Qc * NT) * 8u;
      jac_st(Br[row0 + RP * i][col], oZr, voff_T, soff);
      if constexpr (CPLX) jac_st(Bi[row0 + RP * i][col], oZi, voff_T, soff);
    }
  };

  // ---- diagonal tiles: transformed by the tile solver itself (D_P = J_P^H G[P,P] J_P); the workers move them to their
  // destination quarters.  Nothing reads a half-block below the block diagonal again (tiles are taken from the upper
  // triangle, diagonal tiles from D), so only the upper ones are written.
  if (worker >= 0) {
    for (int P = worker; P < S; P += n_workers) {
      const int bc = jacobi_dest_block(P, col / HB, S);
      const unsigned int voff = (unsigned int)(row0 * ld + bc * HB + col % HB) * 8u;
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int br = jacobi_dest_block(P, (RP * i) / HB, S);
        const unsigned int src = (unsigned int)(P * NT * NT + THR * i) * 8u;
        const unsigned int soff = (unsigned int)((br * HB + (RP * i) % HB) * ld) * 8u;
        if (br <= bc) {
          jac_st(jac_ld(rDr, voff_J, src), oGr, voff, soff);
          if constexpr (CPLX) jac_st(jac_ld(rDi, voff_J, src), oGi, voff, soff);
        }
      }
    }
  }

  // ---- item sequence of this workgroup.  A worker (a workgroup that is not busy with a tile solve) takes, without any
  // communication, the G tiles worker, worker + W, ... and then its static share of the eigenvector tiles (handed out in
  // reverse worker order, which evens out the extra G tile some workers get).  What is left of the eigenvector tiles -
  // and everything the late-joining solver workgroups do - is claimed from the counter.  A claim costs an L2 round trip
  // that the compiler waits for on the spot (vmcnt(0): it would also wait for the prefetch in flight), so only the
  // items that balance the tail pay it.
  const int W = n_workers;
  const int n_g_w = (worker >= 0 && worker < n_off) ? (n_off - worker + W - 1) / W : 0;
  const int n_z_w = worker >= 0 ? n_static : 0;
  const int dyn_base = n_off + n_static * W;
  auto static_id = [&](const int k) { return k < n_g_w ? worker + k * W : n_off + (W - 1 - worker) + (k - n_g_w) * W; };
  const int n_stat = n_g_w + n_z_w;
  // slot[2], slot[3]: the first two item ids; slot[0], slot[1]: alternate per iteration, so that the single barrier at
  // the bottom of an iteration is enough (a slot is rewritten two iterations after it was read)
  unsigned int pending = 0;
  if (tid == 0) {
    slot[2] = n_stat > 0 ? static_id(0) : dyn_base + (int)atomicAdd(counter, 1u);
    slot[3] = n_stat > 1 ? static_id(1) : dyn_base + (int)atomicAdd(counter, 1u);
  }
  __syncthreads();
  JacItem cur = jacobi_decode_item(__builtin_amdgcn_readfirstlane(slot[2]), S, n_off, zch, 1);
  int nxt_id = __builtin_amdgcn_readfirstlane(slot[3]);
  issue_PT(cur);
  issue_Q(cur);
  int iter = 0;

  // ---- G tiles: two products, three operand tiles.  The loop body is straight-line (one kind of item): the prefetch
  // registers keep their place and the compiler's waits stay where the data is needed.
  while (cur.kind == 1) {
    JAC_STAMP(0);
#ifdef XMCA_JAC_PROF
    if (tid == 0 && iter < JAC_PROF_IT && blockIdx.x < JAC_PROF_WG) jac_prof[((int)blockIdx.x * JAC_PROF_IT + iter) * JAC_PROF_ST + 9] = 1;
#endif
    const JacItem nxt = jacobi_decode_item(nxt_id, S, n_off, zch, 1);
    to_A(tP);
    to_B(tT);
    JAC_STAMP(1);
    __syncthreads();
    JAC_STAMP(2);
    issue_PT(nxt);      // the next G tile, or the first eigenvector tile
    mul_AhB();          // B = X = J_P^H T  (contains the barrier between reading and overwriting B)
    JAC_STAMP(3);
    to_A(tQ);           // J_P is dead: every wave passed the barrier inside mul_AhB
    if (tid == 0 && iter + 2 >= n_stat) pending = dyn_base + atomicAdd(counter, 1u);
    __syncthreads();
    JAC_STAMP(4);
    issue_Q(nxt);
    mul_BA();           // B = Y = X J_Q
    __syncthreads();
    JAC_STAMP(5);
    {
      const int P = cur.P, Q = cur.Q;
      // each quarter (hr, hc) goes out once, in the orientation that lies above the block diagonal of the next round
      // (uniform per quarter): as it is, or conjugate-transposed.  Either way a wave writes full row segments.
      constexpr int QR = THR / HB;          // quarter rows per pass
      constexpr int QP = HB / QR;           // passes per quarter
      const int qrow = tid / HB, qcol = tid % HB;
      const unsigned int voff_q = (unsigned int)(qrow * ld + qcol) * 8u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int hr = q >> 1, hc = q & 1;
        const int br = jacobi_dest_block(P, hr, S), bc = jacobi_dest_block(Q, hc, S);
        if (br < bc) {
#pragma unroll
          for (int ps = 0; ps < QP; ++ps) {
            const unsigned int soff = (unsigned int)((br * HB + QR * ps) * ld + bc * HB) * 8u;
            jac_st(Br[hr * HB + qrow + QR * ps][hc * HB + qcol], oGr, voff_q, soff);
            if constexpr (CPLX) jac_st(Bi[hr * HB + qrow + QR * ps][hc * HB + qcol], oGi, voff_q, soff);
          }
        } else {
#pragma unroll
          for (int ps = 0; ps < QP; ++ps) {
            const unsigned int soff = (unsigned int)((bc * HB + QR * ps) * ld + br * HB) * 8u;
            jac_st(Br[hr * HB + qcol][hc * HB + qrow + QR * ps], oGr, voff_q, soff);
            if constexpr (CPLX) jac_st(-Bi[hr * HB + qcol][hc * HB + qrow + QR * ps], oGi, voff_q, soff);
          }
        }
      }
    }
    JAC_STAMP(6);
    if (tid == 0) slot[iter & 1] = (iter + 2 < n_stat) ? static_id(iter + 2) : (int)pending;
    __syncthreads();       // the LDS tiles are free again and the slot is visible
    JAC_STAMP(7);
    cur = nxt;
    nxt_id = __builtin_amdgcn_readfirstlane(slot[iter & 1]);
    ++iter;
  }

  // ---- eigenvector tiles: one product, two operand tiles
  while (cur.kind == 2) {
    JAC_STAMP(0);
#ifdef XMCA_JAC_PROF
    if (tid == 0 && iter < JAC_PROF_IT && blockIdx.x < JAC_PROF_WG) jac_prof[((int)blockIdx.x * JAC_PROF_IT + iter) * JAC_PROF_ST + 9] = 2;
#endif
    const JacItem nxt = jacobi_decode_item(nxt_id, S, n_off, zch, 1);
    to_A(tP);
    to_B(tT);
    JAC_STAMP(1);
    __syncthreads();
    JAC_STAMP(2);
    issue_PT(nxt);
    mul_AhB();
    JAC_STAMP(3);
    if (tid == 0 && iter + 2 >= n_stat) pending = dyn_base + atomicAdd(counter, 1u);
    __syncthreads();
    store_z(cur.P, cur.Q);
    JAC_STAMP(6);
    if (tid == 0) slot[iter & 1] = (iter + 2 < n_stat) ? static_id(iter + 2) : (int)pending;
    __syncthreads();
    JAC_STAMP(7);
    cur = nxt;
    nxt_id = __builtin_amdgcn_readfirstlane(slot[iter & 1]);
    ++iter;
  }
}

// One round = ONE launch of at most two workgroups per CU.  The first S workgroups assemble and sweep the diagonal
// tiles of round r+1 (from G, J, D of round r: jacobi_assemble_next_diag); every workgroup (those S too, once they
// are done) then pulls update items of round r from the work counter until none is left.
// workgroups per CU of the fused round kernel: the 64 x 64 real tiles need 66 KB of LDS (two fit), the 32 x 32 complex
// ones 34 KB (XMCA_JAC_WGS32 of them, registers permitting)
#ifndef XMCA_JAC_WGS32
#define XMCA_JAC_WGS32 4
#endif
template <int NT>
constexpr int jacobi_fused_wgs_per_cu() { return NT >= 64 ? 2 : XMCA_JAC_WGS32; }

template <int NT, bool CPLX>
__global__ __launch_bounds__(jac_threads<NT>(), jacobi_fused_wgs_per_cu<NT>() * jac_threads<NT>() / 256) void jacobi_fused_round_kernel(const double* Gr_in, const double* Gi_in, double* Gr_out,
                                                                    double* Gi_out, const double* Zr_in, const double* Zi_in,
                                                                    double* Zr_out, double* Zi_out, const double* Jr,
                                                                    const double* Ji, const double* Dr, const double* Di,
                                                                    double* Jr_next, double* Ji_next, double* Dr_next,
                                                                    double* Di_next, double tol, const double* scal,
                                                                    unsigned long long* sweep_off, int max_sweeps, int cross_only,
                                                                    int S, int ld, unsigned int* work_counter, int zch, int n_static,
                                                                    int exile_ncu) {
  __shared__ union U {
    JacTileSmem<NT, CPLX> t;
    JacUpdSmem<NT, CPLX> u;
    __device__ U() {}
  } sm;
  __shared__ int slot[4];
#ifdef XMCA_JAC_PROF
  if (threadIdx.x == 0 && blockIdx.x < JAC_PROF_WG) jac_prof[(int)blockIdx.x * JAC_PROF_IT * JAC_PROF_ST + 8] = (long long)__builtin_readcyclecounter();
#endif
  if ((int)blockIdx.x < S) {
    jacobi_assemble_next_diag<NT, CPLX>(sm.t, sm.u, blockIdx.x, S, Gr_in, Gi_in, ld, Jr, Ji, Dr, Di);
#ifdef XMCA_JAC_PROF
    if (threadIdx.x == 0) jac_prof[(int)blockIdx.x * JAC_PROF_IT * JAC_PROF_ST + 0] = (long long)__builtin_readcyclecounter();
#endif
    jacobi_tile_evd_body<NT, CPLX, true>(sm.t, blockIdx.x, nullptr, nullptr, ld, Jr_next, Ji_next, Dr_next, Di_next, tol, scal,
                                         sweep_off, max_sweeps, (cross_only & 1) != 0, (cross_only & 2) != 0);
    __syncthreads();     // the tile image is dead; this workgroup now helps with what is left of the update
#ifdef XMCA_JAC_PROF
    if (threadIdx.x == 0) jac_prof[(int)blockIdx.x * JAC_PROF_IT * JAC_PROF_ST + 1] = (long long)__builtin_readcyclecounter();
#endif
  }
#ifdef XMCA_JAC_PROF_NOUPD
  return;
#endif
  // Worker numbering.  exile_ncu > 0 (eigenvalues-only solves: the update is half as long as the tile-solve chain):
  // blocks b, b + #CU, b + 2 #CU, ... land on the same CU (dispatch order observed on MI355X, scripts/probes/placement.cpp),
  // so the workers with b % #CU < S would share a CU with a tile solve; they leave, and the others are renumbered
  // densely.  The placement only steers who works - any assignment gives the same result.
  int worker = (int)blockIdx.x < S ? -1 : (int)blockIdx.x - S, n_workers = (int)gridDim.x - S;
  if (exile_ncu > 0) {
    auto dense = [&](const int B) { return (B / exile_ncu) * (exile_ncu - S) + max(0, B % exile_ncu - S); };   // workers below block B
    if ((int)blockIdx.x >= S && (int)blockIdx.x % exile_ncu < S) return;
    n_workers = dense((int)gridDim.x);
    if (worker >= 0) worker = dense((int)blockIdx.x);
  }
  jacobi_persistent_update<NT, CPLX>(sm.u, slot, work_counter, Gr_in, Gi_in, Gr_out, Gi_out, Zr_in, Zi_in, Zr_out, Zi_out, Jr, Ji, Dr,
                                     Di, S, ld, zch, n_static, worker, n_workers);
}

__global__ void jacobi_diag_kernel(const double* __restrict__ Gr, int npad, double* __restrict__ d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) d[i] = Gr[(int64_t)i * npad + i];
}

// Largest off-diagonal entry (relative to the matrix scale, entries at the rotation floor ignored; with `diag`, after
// a Cholesky LR step, relative to sqrt(g_ii g_jj) - what the orthogonality of the back-transformed vectors needs) of the state a
// sweep leaves behind.  Only the block-upper triangle of half-blocks is maintained by the fused rounds.
__global__ void jacobi_offmax_kernel(const double* __restrict__ Gr, const double* __restrict__ Gi, int npad, int hb,
                                     const double* __restrict__ scal, const double* __restrict__ diag,
                                     unsigned long long* __restrict__ out) {
  const double gscale = scal[0], floor2 = scal[1] * scal[1];
  double mx = 0.0;
  // one row per block (grid-stride over rows), columns from the row's own half-block on
  for (int r = blockIdx.x; r < npad; r += gridDim.x) {
    const int64_t row = (int64_t)r * npad;
    const double dr = diag ? fabs(diag[r]) : 0.0;
    for (int c = (r / hb) * hb + threadIdx.x; c < npad; c += blockDim.x) {
      if (c == r) continue;
      double g2 = Gr[row + c] * Gr[row + c];
      if (Gi) g2 += Gi[row + c] * Gi[row + c];
      if (!(g2 == g2)) mx = HUGE_VAL;
      else if (diag) { if (g2 > 0.0) mx = fmax(mx, g2 / (dr * fabs(diag[c]))); }   // scaled measure |g_ij|^2 / (g_ii g_jj)
      else if (g2 > floor2) mx = fmax(mx, g2);
    }
  }
  if (mx < HUGE_VAL) mx = diag ? sqrt(mx) : sqrt(mx) / gscale;
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0 && mx > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(mx));
}

// Symmetric permutation between two sweeps: G'[i][j] = G(perm[i], perm[j]), Z'[i][:] = Z[perm[i]][:].  G is read
// through its maintained part (block-upper triangle of half-blocks, diagonal half-blocks in full).
__global__ void jacobi_permute_kernel(const double* __restrict__ Gr, const double* __restrict__ Gi, const double* __restrict__ Zr,
                                      const double* __restrict__ Zi, int npad, int hb, const int* __restrict__ perm,
                                      double* __restrict__ Gor, double* __restrict__ Goi, double* __restrict__ Zor,
                                      double* __restrict__ Zoi) {
  const int i = blockIdx.y;
  const int si = perm[i];
  const int64_t orow = (int64_t)i * npad, srow = (int64_t)si * npad;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < npad; c += gridDim.x * blockDim.x) {
    const int sj = perm[c];
    const bool direct = si / hb <= sj / hb;
    const int64_t src = direct ? srow + sj : (int64_t)sj * npad + si;
    Gor[orow + c] = Gr[src];
    if (Gi) Goi[orow + c] = direct ? Gi[src] : -Gi[src];
    if (Zr) {
      Zor[orow + c] = Zr[srow + c];
      if (Zi) Zoi[orow + c] = Zi[srow + c];
    }
  }
}

// Zs[i][0..n) = Z[perm[i]][0..n)
__global__ void jacobi_gather_kernel(const double* __restrict__ Zr, const double* __restrict__ Zi, int npad,
                                     const int* __restrict__ perm, int n, double* __restrict__ Or, double* __restrict__ Oi,
                                     int64_t ldo) {
  const int i = blockIdx.y;
  const int src = perm[i];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
    Or[(int64_t)i * ldo + c] = Zr[(int64_t)src * npad + c];
    if (Oi) Oi[(int64_t)i * ldo + c] = Zi[(int64_t)src * npad + c];
  }
}

// Zs[i][0..n) = Z[perm[i]][0..n) / ||Z[perm[i]][0..n)||   (one workgroup per row; after a Cholesky LR step the rows carry
// the factor sqrt(lambda_i))
__global__ __launch_bounds__(256) void jacobi_gather_normalize_kernel(const double* __restrict__ Zr, const double* __restrict__ Zi,
                                                                      int npad, const int* __restrict__ perm, int n,
                                                                      double* __restrict__ Or, double* __restrict__ Oi, int64_t ldo) {
  __shared__ double red[4];
  const int i = blockIdx.x;
  const int64_t src = (int64_t)perm[i] * npad;
  double mx = 0.0;
  for (int c = threadIdx.x; c < n; c += 256) mx = fmax(mx, fmax(fabs(Zr[src + c]), Zi ? fabs(Zi[src + c]) : 0.0));
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  __syncthreads();
  const double inv_mx = mx > 0.0 ? 1.0 / mx : 0.0;    // scaled sum of squares: rows of null modes are ~1e-7 sqrt(scale)
  double ss = 0.0;
  for (int c = threadIdx.x; c < n; c += 256) {
    const double a = Zr[src + c] * inv_mx, b = Zi ? Zi[src + c] * inv_mx : 0.0;
    ss += a * a + b * b;
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  ss = red[0] + red[1] + red[2] + red[3];
  const double f = ss > 0.0 ? inv_mx / sqrt(ss) : 0.0;
  for (int c = threadIdx.x; c < n; c += 256) {
    Or[(int64_t)i * ldo + c] = Zr[src + c] * f;
    if (Oi) Oi[(int64_t)i * ldo + c] = Zi[src + c] * f;
  }
}

struct EvdWorkspace {
  DevBuf<double> G[2][2], Z[2][2];  // [ping-pong][plane]
  DevBuf<double> J[2][2], D[2][2];  // [round parity][plane]: rotations J_P and transformed diagonal tiles; the lookahead
                                    // solve of round r+1 writes one parity while round r reads the other
  DevBuf<double> diag, scal;
  DevBuf<unsigned long long> off;   // one accumulator per sweep (ring)
  DevBuf<int> perm;
  DevBuf<unsigned int> work;        // one work counter per round (fused round kernel)
  GemmWorkspace gws;                // Cholesky LR step
  DevBuf<double> lr_R[2], lr_T[2];
};

struct EvdInfo {
  int sweeps = 0;
  int tile = 0;
  int slots = 0;
  double last_off = 0.0;
  int lr_step = 0;        // 1: a Cholesky LR step was inserted (graded spectrum)
  double diag_spread = 0; // q10/q90 of the diagonal when that was decided
};

constexpr int JAC_OFF_RING = 64;

// Hermitian EVD  A = U diag(lam) U^H, lam descending.
//   Ar/Ai : n x n row-major planes (Ai == nullptr for a real symmetric matrix), lda
//   lam_host : n eigenvalues (descending); lam_dev (nullable) gets the same on the device
//   Zr/Zi : n x n, row i = conj(u_i)   (ldz)
template <bool CPLX, int NT>
void hermitian_evd_impl(hipStream_t st, EvdWorkspace& ws, const double* Ar, const double* Ai, int n, int64_t lda,
                        std::vector<double>& lam_host, double* lam_dev, double* Zr, double* Zi, int64_t ldz, double tol,
                        int max_sweeps, EvdInfo* info) {
  if (info) *info = EvdInfo{};
  const int S = std::max(ceil_div(n, NT), 1);
  const int npad = S * NT;
  const size_t nn = (size_t)npad * npad;
  const bool want_z = Zr != nullptr;      // eigenvalues only: the eigenvector tiles (half of the work) are skipped
  for (int b = 0; b < 2; ++b) {
    ws.G[b][0].ensure(nn);
    if (want_z) ws.Z[b][0].ensure(nn);
    ws.J[b][0].ensure((size_t)S * NT * NT);
    ws.D[b][0].ensure((size_t)S * NT * NT);
    if (CPLX) {
      ws.G[b][1].ensure(nn);
      if (want_z) ws.Z[b][1].ensure(nn);
      ws.J[b][1].ensure((size_t)S * NT * NT);
      ws.D[b][1].ensure((size_t)S * NT * NT);
    }
  }
  ws.diag.ensure((size_t)npad);
  ws.scal.ensure(4);
  ws.off.ensure(JAC_OFF_RING);
  ws.perm.ensure((size_t)npad);
  if (max_sweeps > JAC_OFF_RING - 2) max_sweeps = JAC_OFF_RING - 2;   // the last slot holds the post-sweep measure

  XMCA_HIP(hipMemsetAsync(ws.off.get(), 0, sizeof(unsigned long long) * JAC_OFF_RING, st));
  XMCA_HIP(hipMemsetAsync(ws.scal.get(), 0, sizeof(double) * 4, st));
  if (npad > n)
    hipLaunchKernelGGL(jacobi_frobenius_kernel, dim3(std::min(n, 1024)), dim3(256), 0, st, Ar, CPLX ? Ai : nullptr, n, lda, ws.scal.get());
  hipLaunchKernelGGL(jacobi_init_scale_kernel, dim3(1), dim3(256), 0, st, Ar, n, lda, tol, ws.scal.get());
  hipLaunchKernelGGL(jacobi_init_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, Ar, CPLX ? Ai : nullptr, n, lda,
                     ws.G[0][0].get(), CPLX ? ws.G[0][1].get() : nullptr, want_z ? ws.Z[0][0].get() : nullptr,
                     (CPLX && want_z) ? ws.Z[0][1].get() : nullptr, npad, ws.scal.get());
  XMCA_HIP(hipGetLastError());

  const double tile_tol = 2e-15;
  // Inexact inner solves: a diagonal tile is swept only `inner_cap` times per visit.  Measured on MI355X (C2,
  // T = 2920): 1 sweep per visit needs the same 13 outer sweeps as a full tile solve at a third of the time.
  static const int inner_cap = [] { const char* e = std::getenv("XMCA_JACOBI_INNER"); int v = e ? std::atoi(e) : 0; return v > 0 ? v : 1; }();
  static const bool lookahead_on = [] { const char* e = std::getenv("XMCA_JACOBI_LOOKAHEAD"); return !(e && e[0] == '0'); }();
  const bool lookahead = lookahead_on && S >= 3;

  int cur = 0;
  const int rounds = (S == 1) ? 1 : 2 * S - 1;
  const int zchunks = want_z ? (S + JAC_ZW - 1) / JAC_ZW : 0;   // 0: the launches simply do not contain Z tiles
  const int n_off = S * (S - 1) / 2;
  // tiles are swept in full once per outer sweep (its first round), cross-block only otherwise
  static const bool cross_on = [] { const char* e = std::getenv("XMCA_JACOBI_CROSS"); return !(e && e[0] == '0'); }();
  auto is_cross = [&](int round_in_sweep) { return cross_on && S > 1 && inner_cap == 1 && round_in_sweep != 0; };
  // bit 1: two-level form of the cross sweep (64 x 64 real tiles; XMCA_JACOBI_TWOLEVEL=0 selects the flat sweep)
  static const bool twolevel_on = [] { const char* e = std::getenv("XMCA_JACOBI_TWOLEVEL"); return !(e && e[0] == '0'); }();
  const int cross_code = twolevel_on ? 3 : 1;
  auto evd = [&](hipStream_t s, int gbuf, int par, int sweep_slot, int round_in_sweep) {
    hipLaunchKernelGGL((jacobi_tile_evd_kernel<NT, CPLX>), dim3(S), dim3(jac_threads<NT>()), 0, s, ws.G[gbuf][0].get(),
                       CPLX ? ws.G[gbuf][1].get() : nullptr, npad, ws.J[par][0].get(), CPLX ? ws.J[par][1].get() : nullptr,
                       ws.D[par][0].get(), CPLX ? ws.D[par][1].get() : nullptr, tile_tol, ws.scal.get(),
                       ws.off.get() + sweep_slot, S == 1 ? 60 : inner_cap, is_cross(round_in_sweep) ? 1 : 0);
  };
  auto update = [&](hipStream_t s, int par, int grid) {
    hipLaunchKernelGGL((jacobi_update_kernel<NT, CPLX>), dim3(grid), dim3(jac_threads<NT>()), 0, s, ws.G[cur][0].get(),
                       CPLX ? ws.G[cur][1].get() : nullptr, ws.G[cur ^ 1][0].get(), CPLX ? ws.G[cur ^ 1][1].get() : nullptr,
                       ws.Z[cur][0].get(), CPLX ? ws.Z[cur][1].get() : nullptr, ws.Z[cur ^ 1][0].get(),
                       CPLX ? ws.Z[cur ^ 1][1].get() : nullptr, ws.J[par][0].get(), CPLX ? ws.J[par][1].get() : nullptr,
                       ws.D[par][0].get(), CPLX ? ws.D[par][1].get() : nullptr, S, npad);
  };

  // fused rounds: persistent workgroups (two per CU) pull items from one counter per round
  const int zch2 = want_z ? S : 0;
  const int fused_items = n_off + S * zch2;
  static const int resident_wgs = [] {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return jacobi_fused_wgs_per_cu<NT>() * cus;
  }();
  const int fused_grid = std::max(S, std::min(resident_wgs, S + fused_items));
  // XMCA_JACOBI_STATIC: percentage of an even split of the eigenvector tiles that is handed out statically
  static const int static_pct = [] { const char* e = std::getenv("XMCA_JACOBI_STATIC"); const int v = e ? std::atoi(e) : 100; return std::min(std::max(v, 0), 100); }();
  const int n_workers = fused_grid - S;
  // eigenvector tiles handed out statically per worker (the G tiles always are)
  const int n_static = n_workers <= 0 ? 0 : (int)((int64_t)S * zch2 * static_pct / 100 / n_workers);
  // eigenvalues only: the tile solves get their CUs to themselves (XMCA_JACOBI_EXILE=0 disables, =2 forces it always)
  static const int exile_mode = [] { const char* e = std::getenv("XMCA_JACOBI_EXILE"); return e ? std::atoi(e) : 1; }();
  const int ncu = resident_wgs / jacobi_fused_wgs_per_cu<NT>();
  // (measured: eigenvalues of a 2920^2 real matrix 57.5 -> 51.8 ms; with four workgroups per CU - the 32 x 32 complex
  //  tiles - a quarter of the workers would leave and the solve gets slower, 86 -> 95 ms at n = 2501)
  const bool exile = (exile_mode == 2 || (exile_mode == 1 && !want_z && jacobi_fused_wgs_per_cu<NT>() == 2)) &&
                     fused_grid == resident_wgs && S < ncu / 2;
  const int exile_ncu = exile ? ncu : 0;
  if (lookahead) {
    ws.work.ensure((size_t)max_sweeps * rounds);
    XMCA_HIP(hipMemsetAsync(ws.work.get(), 0, sizeof(unsigned int) * (size_t)max_sweeps * rounds, st));
  }

  int sweeps = 0;
  double off = 0.0;
  int64_t round_no = 0;
  bool lr_applied = false;
  double lr_delta = 0.0;
  if (lookahead) evd(st, cur, 0, 0, 0);  // diagonal tiles of the very first round
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    for (int r = 0; r < rounds; ++r, ++round_no) {
      const int par = (int)(round_no & 1);
      if (!lookahead) {
        evd(st, cur, par, sweep, r);
        update(st, par, S + n_off + S * zchunks);
      } else {
        // ONE launch: tile solves of round r+1 (assembled from this round's G, J, D) + the whole update of round r
        const int next_slot = (r == rounds - 1) ? sweep + 1 : sweep;
        hipLaunchKernelGGL((jacobi_fused_round_kernel<NT, CPLX>), dim3(fused_grid), dim3(jac_threads<NT>()), 0, st,
                           ws.G[cur][0].get(), CPLX ? ws.G[cur][1].get() : nullptr, ws.G[cur ^ 1][0].get(),
                           CPLX ? ws.G[cur ^ 1][1].get() : nullptr, ws.Z[cur][0].get(), CPLX ? ws.Z[cur][1].get() : nullptr,
                           ws.Z[cur ^ 1][0].get(), CPLX ? ws.Z[cur ^ 1][1].get() : nullptr, ws.J[par][0].get(),
                           CPLX ? ws.J[par][1].get() : nullptr, ws.D[par][0].get(), CPLX ? ws.D[par][1].get() : nullptr,
                           ws.J[par ^ 1][0].get(), CPLX ? ws.J[par ^ 1][1].get() : nullptr, ws.D[par ^ 1][0].get(),
                           CPLX ? ws.D[par ^ 1][1].get() : nullptr, tile_tol, ws.scal.get(), ws.off.get() + next_slot,
                           inner_cap, is_cross((r + 1) % rounds) ? cross_code : 0, S, npad, ws.work.get() + round_no, zch2, n_static,
                           exile_ncu);
      }
#ifdef XMCA_JAC_PROF
      if (lookahead && round_no == 300) {   // stamps of a typical (cross-block) round in the middle of the solve
        XMCA_HIP(hipStreamSynchronize(st));
        std::vector<long long> h((size_t)JAC_PROF_WG * JAC_PROF_IT * JAC_PROF_ST);
        XMCA_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(jac_prof), h.size() * sizeof(long long)));
        if (FILE* f = std::fopen("gpurun_out/jac_prof.txt", "w")) {
          for (int w = 0; w < JAC_PROF_WG; ++w)
            for (int it = 0; it < JAC_PROF_IT; ++it) {
              std::fprintf(f, "%d %d", w, it);
              for (int k = 0; k < JAC_PROF_ST; ++k) std::fprintf(f, " %lld", h[((size_t)w * JAC_PROF_IT + it) * JAC_PROF_ST + k]);
              std::fprintf(f, "\n");
            }
          std::fclose(f);
        }
      }
#endif
      cur ^= 1;
    }
    XMCA_HIP(hipGetLastError());
    // two measures per sweep: `off` = the largest entry the sweep met when it visited the tiles (also carries the NaN
    // flag), `left` = the largest entry of the matrix it leaves behind (one 25 us pass).  Stopping on `left` saves the
    // sweep that would only confirm convergence.
    unsigned long long bits[2] = {0, 0};
    if (S > 1) {
      XMCA_HIP(hipMemsetAsync(ws.off.get() + JAC_OFF_RING - 1, 0, sizeof(unsigned long long), st));
      if (lr_applied)
        hipLaunchKernelGGL(jacobi_diag_kernel, dim3(ceil_div(npad, 256)), dim3(256), 0, st, ws.G[cur][0].get(), npad, ws.diag.get());
      hipLaunchKernelGGL(jacobi_offmax_kernel, dim3(2048), dim3(256), 0, st, ws.G[cur][0].get(), CPLX ? ws.G[cur][1].get() : nullptr,
                         npad, NT / 2, ws.scal.get(), lr_applied ? ws.diag.get() : nullptr, ws.off.get() + JAC_OFF_RING - 1);
      XMCA_HIP(hipMemcpyAsync(&bits[1], ws.off.get() + JAC_OFF_RING - 1, sizeof(bits[1]), hipMemcpyDeviceToHost, st));
    }
    XMCA_HIP(hipMemcpyAsync(&bits[0], ws.off.get() + sweep, sizeof(bits[0]), hipMemcpyDeviceToHost, st));
    XMCA_HIP(hipStreamSynchronize(st));
    std::memcpy(&off, &bits[0], sizeof(double));
    double left = off;
    if (S > 1) std::memcpy(&left, &bits[1], sizeof(double));
    ++sweeps;
    static const bool trace = std::getenv("XMCA_JACOBI_TRACE") != nullptr;
    if (trace) std::fprintf(stderr, "[xmca jacobi] n=%d NT=%d cplx=%d sweep %d: max off/scale seen = %.3e, left = %.3e\n", n, NT, (int)CPLX, sweeps, off, left);
    if (S == 1) break;
    if (!(left >= tol) || !std::isfinite(left)) { off = left; break; }
    // Cholesky LR step for graded spectra.  With eigenvalues spread evenly over many decades the couplings between
    // large and small eigenvalues have to fall far below the small ones before those start to converge, and the
    // sweeps only converge linearly (measured: 30 sweeps for 12 decades at n = 2920, 12 for a flat bulk).  One step of
    // the Cholesky LR iteration, G + delta I = R^H R -> M = R R^H (= R G R^-1 + delta I), removes exactly these
    // long-range couplings (11 sweeps for the same matrix); it costs about one sweep, so it is taken only when the
    // diagonal after `lr_after` sweeps says the spectrum is graded.  Z <- R Z turns the accumulated rows into
    // sqrt(lambda_i) x eigenvector, which the final gather normalises.
    static const int lr_mode = [] { const char* e = std::getenv("XMCA_JACOBI_LR"); return e ? std::atoi(e) : 1; }();   // 0 off, 1 auto, 2 always
    static const double lr_spread = [] { const char* e = std::getenv("XMCA_JACOBI_LR_SPREAD"); return e ? std::atof(e) : 100.0; }();
    static const int lr_after = [] { const char* e = std::getenv("XMCA_JACOBI_LR_AFTER"); return e ? std::max(std::atoi(e), 1) : 2; }();
    if (lookahead && lr_mode != 0 && sweeps == lr_after && !lr_applied) {
      hipLaunchKernelGGL(jacobi_diag_kernel, dim3(ceil_div(npad, 256)), dim3(256), 0, st, ws.G[cur][0].get(), npad, ws.diag.get());
      std::vector<double> dd(npad);
      XMCA_HIP(hipMemcpyAsync(dd.data(), ws.diag.get(), sizeof(double) * npad, hipMemcpyDeviceToHost, st));
      XMCA_HIP(hipStreamSynchronize(st));
      std::vector<int> pm(npad);
      for (int i = 0; i < npad; ++i) pm[i] = i;
      std::stable_sort(pm.begin(), pm.end(), [&](int a, int b) { return dd[a] > dd[b]; });
      // the padding (-scale, decoupled) must be what sorts last; a matrix with diagonal entries down there is not a
      // Gram matrix and stays on the plain path
      const double dmax = dd[pm[0]], dmin = dd[pm[n - 1]];
      const double q10 = dd[pm[n / 10]], q90 = std::max(dd[pm[(int64_t)n * 9 / 10]], 1e-14 * dmax);
      const double spread = (dmax > 0.0 && q10 > 0.0) ? q10 / q90 : 0.0;
      if (info) info->diag_spread = spread;
      if (dmin > -0.25 * dmax && dmax > 0.0 && std::isfinite(dmax) && (lr_mode == 2 || spread > lr_spread)) {
        XMCA_HIP(hipMemcpyAsync(ws.perm.get(), pm.data(), sizeof(int) * npad, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(jacobi_permute_kernel, dim3(std::min(ceil_div(npad, 256), 16), npad), dim3(256), 0, st, ws.G[cur][0].get(),
                           CPLX ? ws.G[cur][1].get() : nullptr, want_z ? ws.Z[cur][0].get() : nullptr,
                           (CPLX && want_z) ? ws.Z[cur][1].get() : nullptr, npad, NT / 2, ws.perm.get(), ws.G[cur ^ 1][0].get(),
                           CPLX ? ws.G[cur ^ 1][1].get() : nullptr, want_z ? ws.Z[cur ^ 1][0].get() : nullptr,
                           (CPLX && want_z) ? ws.Z[cur ^ 1][1].get() : nullptr);
        XMCA_HIP(hipGetLastError());
        cur ^= 1;
        const size_t row = sizeof(double) * (size_t)n, pitch = sizeof(double) * (size_t)npad;
        for (int pl = 0; pl < (CPLX ? 2 : 1); ++pl) {
          ws.lr_R[pl].ensure((size_t)n * n);
          XMCA_HIP(hipMemcpy2DAsync(ws.lr_R[pl].get(), row, ws.G[cur][pl].get(), pitch, row, n, hipMemcpyDeviceToDevice, st));
        }
        double* Rr = ws.lr_R[0].get();
        double* Ri = CPLX ? ws.lr_R[1].get() : nullptr;
        const double rel_shift = 1e-13;
        if (cholesky_upper(st, ws.gws, Rr, Ri, n, n, rel_shift)) {
          // M = R R^H over the leading block of G (the padding stays decoupled)
          cgemm<double>(st, ws.gws, Rr, Ri, n, true, false, Rr, Ri, n, false, true, ws.G[cur][0].get(),
                        CPLX ? ws.G[cur][1].get() : nullptr, npad, n, n, n, 1.0, nullptr, nullptr, true);
          if (want_z) {
            for (int pl = 0; pl < (CPLX ? 2 : 1); ++pl) ws.lr_T[pl].ensure((size_t)n * n);
            cgemm<double>(st, ws.gws, Rr, Ri, n, true, false, ws.Z[cur][0].get(), CPLX ? ws.Z[cur][1].get() : nullptr, npad, true, false,
                          ws.lr_T[0].get(), CPLX ? ws.lr_T[1].get() : nullptr, n, n, n, n, 1.0, nullptr, nullptr, false);
            for (int pl = 0; pl < (CPLX ? 2 : 1); ++pl)
              XMCA_HIP(hipMemcpy2DAsync(ws.Z[cur][pl].get(), pitch, ws.lr_T[pl].get(), row, row, n, hipMemcpyDeviceToDevice, st));
          }
          lr_applied = true;
          lr_delta = rel_shift * dmax;
          // M is positive definite with a meaningful (graded) diagonal: from here on rotations and the stopping rule are
          // relative to sqrt(m_ii m_jj) alone - the orthogonality of the back-transformed vectors is the scaled
          // off-diagonal part of the final M - and the absolute rotation floor is dropped
          XMCA_HIP(hipMemsetAsync(ws.scal.get() + 1, 0, sizeof(double), st));
        }
        XMCA_HIP(hipStreamSynchronize(st));   // pm goes out of scope
        evd(st, cur, (int)(round_no & 1), sweep + 1, 0);   // the lookahead solve of the next round saw the old matrix
        if (trace) std::fprintf(stderr, "[xmca jacobi] n=%d diagonal spread q10/q90 = %.3e: Cholesky LR step %s\n", n, spread, lr_applied ? "taken" : "failed (not positive definite)");
      } else if (trace) {
        std::fprintf(stderr, "[xmca jacobi] n=%d diagonal spread q10/q90 = %.3e: no LR step\n", n, spread);
      }
    }
  }
  XMCA_CHECK(std::isfinite(off), XMCA_ERR_NUMERIC, "SVD failed. NaN entries may be the problem.");

  // eigenvalues = diagonal; sort descending on the host, drop the padding (= the most negative entries)
  hipLaunchKernelGGL(jacobi_diag_kernel, dim3(ceil_div(npad, 256)), dim3(256), 0, st, ws.G[cur][0].get(), npad, ws.diag.get());
  std::vector<double> d(npad);
  XMCA_HIP(hipMemcpyAsync(d.data(), ws.diag.get(), sizeof(double) * npad, hipMemcpyDeviceToHost, st));
  XMCA_HIP(hipStreamSynchronize(st));
  std::vector<int> perm(npad);
  for (int i = 0; i < npad; ++i) perm[i] = i;
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return d[a] > d[b]; });
  lam_host.resize(n);
  for (int i = 0; i < n; ++i) lam_host[i] = d[perm[i]] - lr_delta;
  XMCA_HIP(hipMemcpyAsync(ws.perm.get(), perm.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
  if (lam_dev) XMCA_HIP(hipMemcpyAsync(lam_dev, lam_host.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
  if (Zr && lr_applied) {
    hipLaunchKernelGGL(jacobi_gather_normalize_kernel, dim3(n), dim3(256), 0, st, ws.Z[cur][0].get(), CPLX ? ws.Z[cur][1].get() : nullptr,
                       npad, ws.perm.get(), n, Zr, CPLX ? Zi : nullptr, ldz);
    XMCA_HIP(hipGetLastError());
  } else if (Zr) {
    hipLaunchKernelGGL(jacobi_gather_kernel, dim3(std::min(ceil_div(n, 256), 64), n), dim3(256), 0, st, ws.Z[cur][0].get(),
                       CPLX ? ws.Z[cur][1].get() : nullptr, npad, ws.perm.get(), n, Zr, CPLX ? Zi : nullptr, ldz);
    XMCA_HIP(hipGetLastError());
  }
  XMCA_HIP(hipStreamSynchronize(st));   // perm / lam_host staging buffers go out of scope
  if (info) { info->sweeps = sweeps; info->tile = NT; info->slots = S; info->last_off = off; info->lr_step = lr_applied ? 1 : 0; }
}

inline void hermitian_evd(hipStream_t st, EvdWorkspace& ws, const double* Ar, const double* Ai, int n, int64_t lda,
                          std::vector<double>& lam_host, double* lam_dev, double* Zr, double* Zi, int64_t ldz,
                          EvdInfo* info = nullptr, int force_tile = 0) {
  // stop after the first sweep whose largest off-diagonal entry (seen when its tile is visited) is below
  // 1e-10 * max|diag|: with the (at least fast-linear, normally quadratic) convergence the state left
  // behind is at the 1e-13 rotation floor.
  const double tol = 1e-10;
  const int max_sweeps = 50;
  if (Ai) {
    // 64 x 64 complex tiles do not fit the LDS of the update kernel
    hermitian_evd_impl<true, 32>(st, ws, Ar, Ai, n, lda, lam_host, lam_dev, Zr, Zi, ldz, tol, max_sweeps, info);
  } else {
    const int nt = force_tile ? force_tile : (n > 32 ? 64 : 32);
    if (nt == 32) hermitian_evd_impl<false, 32>(st, ws, Ar, nullptr, n, lda, lam_host, lam_dev, Zr, nullptr, ldz, tol, max_sweeps, info);
    else hermitian_evd_impl<false, 64>(st, ws, Ar, nullptr, n, lda, lam_host, lam_dev, Zr, nullptr, ldz, tol, max_sweeps, info);
  }
}

}  // namespace xmca
