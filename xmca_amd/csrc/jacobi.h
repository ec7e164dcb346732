// Hermitian eigensolver for the small (T x T or N x N) stage of solve():
// two-sided BLOCK Jacobi in f64 with round-robin pair slots.
//
//   per round:  (1) jacobi_tile_evd_kernel   one workgroup per pair slot P diagonalises the
//                   NT x NT diagonal tile G[P,P] with a parallel-order cyclic Jacobi held in LDS
//                   (rotations start from the identity and always take the inner angle, so the
//                   accumulated J_P stays close to the identity -> quadratic outer convergence);
//               (2) jacobi_update_kernel     every off-diagonal tile  G'[P,Q] = J_P^H G[P,Q] J_Q
//                   (upper triangle + mirrored write) and every eigenvector tile
//                   Z'[P,c] = J_P^H Z[P,c]  on the f64 matrix pipe (v_mfma_f64_16x16x4_f64),
//                   written straight to the slots of the NEXT round (ping-pong buffers), so the
//                   tournament permutation costs no extra pass.
//   after 2S-1 rounds every pair of half-blocks has met once (= one sweep).
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

// scal[0] = scale of the matrix (max |diag|), scal[1] = absolute rotation floor
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
  }
}

// G0 = [A 0; 0 -scale*I], Z0 = I   (npad x npad, planes)
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
    gr = -scal[0];
  }
  Gr[idx] = gr;
  Zr[idx] = (r == c) ? 1.0 : 0.0;
  if (Gi) { Gi[idx] = gi; Zi[idx] = 0.0; }
}

template <int NT, bool CPLX>
__global__ __launch_bounds__(256) void jacobi_tile_evd_kernel(const double* __restrict__ Gr, const double* __restrict__ Gi,
                                                              int ld, double* __restrict__ Jr, double* __restrict__ Ji,
                                                              double* __restrict__ lam, double tol,
                                                              const double* __restrict__ scal,
                                                              unsigned long long* __restrict__ sweep_off, int max_sweeps) {
  constexpr int H = NT / 2;
  constexpr int LD = NT + 1;
  __shared__ double Mr[NT][LD];
  __shared__ double Mi[CPLX ? NT : 1][CPLX ? LD : 1];
  __shared__ double Vr[NT][LD];
  __shared__ double Vi[CPLX ? NT : 1][CPLX ? LD : 1];
  __shared__ double rc[H], rsr[H], rsi[H];
  __shared__ int rp[H], rq[H];
  __shared__ int flag;
  __shared__ double red[4];

  const int P = blockIdx.x, tid = threadIdx.x;
  const double gscale = scal[0], abs_floor = scal[1];
  const int64_t base = (int64_t)P * NT * ld + (int64_t)P * NT;
  for (int e = tid; e < NT * NT; e += 256) {
    const int i = e / NT, j = e % NT;
    Mr[i][j] = Gr[base + (int64_t)i * ld + j];
    Vr[i][j] = (i == j) ? 1.0 : 0.0;
    if constexpr (CPLX) {
      Mi[i][j] = Gi[base + (int64_t)i * ld + j];
      Vi[i][j] = 0.0;
    }
  }
  __syncthreads();

  // off-diagonal measure of this tile before it is touched (drives the outer sweep loop)
  {
    double mx = 0.0;
    for (int e = tid; e < NT * NT; e += 256) {
      const int i = e / NT, j = e % NT;
      if (i <= j) {
        double g2 = Mr[i][j] * Mr[i][j];
        if constexpr (CPLX) g2 += Mi[i][j] * Mi[i][j];
        if (!(g2 == g2)) mx = HUGE_VAL;                       // NaN in the matrix: reported to the host as +inf
        else if (i < j && g2 > abs_floor * abs_floor) mx = fmax(mx, sqrt(g2) / gscale);
      }
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
      mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
      if (mx > 0.0) atomicMax(sweep_off, (unsigned long long)__double_as_longlong(mx));
    }
  }

  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    if (tid == 0) flag = 0;
    __syncthreads();
    for (int step = 0; step < NT - 1; ++step) {
      if (tid < H) {
        int a, b;
        if (tid == 0) { a = NT - 1; b = step; }
        else { a = (step + tid) % (NT - 1); b = (step - tid + (NT - 1)) % (NT - 1); }
        const int p = min(a, b), q = max(a, b);
        const double app = Mr[p][p], aqq = Mr[q][q];
        const double gr = Mr[p][q];
        double gi = 0.0;
        if constexpr (CPLX) gi = Mi[p][q];
        const double g2 = gr * gr + gi * gi;
        double c = 1.0, sr = 0.0, si = 0.0;
        if (g2 > 0.0 && g2 > abs_floor * abs_floor && g2 > tol * tol * fabs(app * aqq)) {
          const double ag = sqrt(g2);
          const double tau = (aqq - app) / (2.0 * ag);
          const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          c = 1.0 / sqrt(1.0 + t * t);
          const double s = t * c;
          sr = s * gr / ag;
          si = s * gi / ag;
          flag = 1;
        }
        rp[tid] = p; rq[tid] = q; rc[tid] = c; rsr[tid] = sr; rsi[tid] = si;
      }
      __syncthreads();
      // M <- J^H M J as (NT/2)^2 independent 2x2 blocks
      for (int e = tid; e < H * H; e += 256) {
        const int k1 = e / H, k2 = e % H;
        const int p1 = rp[k1], q1 = rq[k1], p2 = rp[k2], q2 = rq[k2];
        const double c1 = rc[k1], s1r = rsr[k1], s1i = rsi[k1];
        const double c2 = rc[k2], s2r = rsr[k2], s2i = rsi[k2];
        if (c1 == 1.0 && c2 == 1.0 && s1r == 0.0 && s2r == 0.0 && s1i == 0.0 && s2i == 0.0) continue;
        double b00r = Mr[p1][p2], b01r = Mr[p1][q2], b10r = Mr[q1][p2], b11r = Mr[q1][q2];
        if constexpr (!CPLX) {
          // rows: x0 = c1 b0 - s1 b1 ; x1 = s1 b0 + c1 b1
          const double x00 = c1 * b00r - s1r * b10r, x01 = c1 * b01r - s1r * b11r;
          const double x10 = s1r * b00r + c1 * b10r, x11 = s1r * b01r + c1 * b11r;
          // cols: y_i0 = c2 x_i0 - s2 x_i1 ; y_i1 = s2 x_i0 + c2 x_i1
          double y00 = c2 * x00 - s2r * x01, y01 = s2r * x00 + c2 * x01;
          double y10 = c2 * x10 - s2r * x11, y11 = s2r * x10 + c2 * x11;
          if (k1 == k2) { y01 = 0.0; y10 = 0.0; }
          Mr[p1][p2] = y00; Mr[p1][q2] = y01; Mr[q1][p2] = y10; Mr[q1][q2] = y11;
        } else {
          double b00i = Mi[p1][p2], b01i = Mi[p1][q2], b10i = Mi[q1][p2], b11i = Mi[q1][q2];
          // x0j = c1 b0j - sg1 b1j ; x1j = conj(sg1) b0j + c1 b1j        (sg = sr + i si)
          const double x00r = c1 * b00r - (s1r * b10r - s1i * b10i), x00i = c1 * b00i - (s1r * b10i + s1i * b10r);
          const double x01r = c1 * b01r - (s1r * b11r - s1i * b11i), x01i = c1 * b01i - (s1r * b11i + s1i * b11r);
          const double x10r = (s1r * b00r + s1i * b00i) + c1 * b10r, x10i = (s1r * b00i - s1i * b00r) + c1 * b10i;
          const double x11r = (s1r * b01r + s1i * b01i) + c1 * b11r, x11i = (s1r * b01i - s1i * b01r) + c1 * b11i;
          // yi0 = c2 xi0 - conj(sg2) xi1 ; yi1 = sg2 xi0 + c2 xi1
          double y00r = c2 * x00r - (s2r * x01r + s2i * x01i), y00i = c2 * x00i - (s2r * x01i - s2i * x01r);
          double y01r = (s2r * x00r - s2i * x00i) + c2 * x01r, y01i = (s2r * x00i + s2i * x00r) + c2 * x01i;
          double y10r = c2 * x10r - (s2r * x11r + s2i * x11i), y10i = c2 * x10i - (s2r * x11i - s2i * x11r);
          double y11r = (s2r * x10r - s2i * x10i) + c2 * x11r, y11i = (s2r * x10i + s2i * x10r) + c2 * x11i;
          if (k1 == k2) { y01r = y01i = y10r = y10i = 0.0; y00i = 0.0; y11i = 0.0; }
          Mr[p1][p2] = y00r; Mr[p1][q2] = y01r; Mr[q1][p2] = y10r; Mr[q1][q2] = y11r;
          Mi[p1][p2] = y00i; Mi[p1][q2] = y01i; Mi[q1][p2] = y10i; Mi[q1][q2] = y11i;
        }
      }
      // V <- V J   (columns p,q of every row)
      for (int e = tid; e < NT * H; e += 256) {
        const int i = e / H, k = e % H;
        const double c = rc[k], sr = rsr[k], si = rsi[k];
        if (c == 1.0 && sr == 0.0 && si == 0.0) continue;
        const int p = rp[k], q = rq[k];
        const double vpr = Vr[i][p], vqr = Vr[i][q];
        if constexpr (!CPLX) {
          Vr[i][p] = c * vpr - sr * vqr;
          Vr[i][q] = sr * vpr + c * vqr;
        } else {
          const double vpi = Vi[i][p], vqi = Vi[i][q];
          // new_p = c vp - conj(sg) vq ; new_q = sg vp + c vq
          Vr[i][p] = c * vpr - (sr * vqr + si * vqi);
          Vi[i][p] = c * vpi - (sr * vqi - si * vqr);
          Vr[i][q] = (sr * vpr - si * vpi) + c * vqr;
          Vi[i][q] = (sr * vpi + si * vpr) + c * vqi;
        }
      }
      __syncthreads();
    }
    const int f = flag;
    __syncthreads();
    if (!f) break;
  }

  const int64_t jb = (int64_t)P * NT * NT;
  for (int e = tid; e < NT * NT; e += 256) {
    const int i = e / NT, j = e % NT;
    Jr[jb + e] = Vr[i][j];
    if constexpr (CPLX) Ji[jb + e] = Vi[i][j];
  }
  if (tid < NT) lam[(int64_t)P * NT + tid] = Mr[tid][tid];
}

// One round of the two-sided update.  blockIdx.x enumerates the S(S+1)/2 upper G tiles, then the S*S Z tiles.
template <int NT, bool CPLX>
__global__ __launch_bounds__(256) void jacobi_update_kernel(const double* __restrict__ Gr_in, const double* __restrict__ Gi_in,
                                                            double* __restrict__ Gr_out, double* __restrict__ Gi_out,
                                                            const double* __restrict__ Zr_in, const double* __restrict__ Zi_in,
                                                            double* __restrict__ Zr_out, double* __restrict__ Zi_out,
                                                            const double* __restrict__ Jr, const double* __restrict__ Ji,
                                                            const double* __restrict__ lam, int S, int ld) {
  constexpr int LD = NT + 1;
  constexpr int HB = NT / 2;
  constexpr int TPD = NT / 16;          // MFMA tiles per dimension
  constexpr int NACC = TPD * TPD / 4;   // output tiles per wave
  __shared__ double JPr[NT][LD], JQr[NT][LD], Tr[NT][LD];
  __shared__ double JPi[CPLX ? NT : 1][CPLX ? LD : 1], JQi[CPLX ? NT : 1][CPLX ? LD : 1], Ti[CPLX ? NT : 1][CPLX ? LD : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int nG = S * (S + 1) / 2;
  int id = blockIdx.x;
  bool is_g = id < nG;
  int P, Q;
  if (is_g) {
    P = 0;
    int rem = id;
    while (rem >= S - P) { rem -= S - P; ++P; }
    Q = P + rem;
  } else {
    id -= nG;
    P = id / S;
    Q = id % S;   // column chunk of Z
  }

  if (is_g && P == Q) {
    // the diagonal tile becomes diag(lam_P); write it (and zeros) to its destination blocks
    for (int e = tid; e < NT * NT; e += 256) {
      const int r = e / NT, c = e % NT;
      const int dr = jacobi_dest_block(P, r / HB, S) * HB + r % HB;
      const int dc = jacobi_dest_block(P, c / HB, S) * HB + c % HB;
      const int64_t o = (int64_t)dr * ld + dc;
      Gr_out[o] = (r == c) ? lam[(int64_t)P * NT + r] : 0.0;
      if constexpr (CPLX) Gi_out[o] = 0.0;
    }
    return;
  }

  const double* __restrict__ Sr = is_g ? Gr_in : Zr_in;
  const double* __restrict__ Si = is_g ? Gi_in : Zi_in;
  const int64_t tbase = (int64_t)P * NT * ld + (int64_t)Q * NT;
  const int64_t jpb = (int64_t)P * NT * NT, jqb = (int64_t)Q * NT * NT;
  for (int e = tid; e < NT * NT; e += 256) {
    const int r = e / NT, c = e % NT;
    Tr[r][c] = Sr[tbase + (int64_t)r * ld + c];
    JPr[r][c] = Jr[jpb + e];
    if (is_g) JQr[r][c] = Jr[jqb + e];
    if constexpr (CPLX) {
      Ti[r][c] = Si[tbase + (int64_t)r * ld + c];
      JPi[r][c] = Ji[jpb + e];
      if (is_g) JQi[r][c] = Ji[jqb + e];
    }
  }
  __syncthreads();

  // X = J_P^H T
  d4_t xr[NACC], xi[CPLX ? NACC : 1];
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
    d4_t ar = {0, 0, 0, 0}, ai = {0, 0, 0, 0};
    for (int k0 = 0; k0 < NT; k0 += 4) {
      const int k = k0 + l4;
      const double jr = JPr[k][ti * 16 + l15];
      const double tr = Tr[k][tj * 16 + l15];
      ar = Mfma<double>::mma(jr, tr, ar);
      if constexpr (CPLX) {
        const double ji = JPi[k][ti * 16 + l15];
        const double tim = Ti[k][tj * 16 + l15];
        ar = Mfma<double>::mma(ji, tim, ar);     // + JPi^T Ti
        ai = Mfma<double>::mma(jr, tim, ai);     // + JPr^T Ti
        ai = Mfma<double>::mma(-ji, tr, ai);     // - JPi^T Tr
      }
    }
    xr[a] = ar;
    if constexpr (CPLX) xi[a] = ai;
  }

  if (!is_g) {
    // Z'[dest(P,h) rows, chunk Q] = X
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + Mfma<double>::row(lane, r), col = tj * 16 + l15;
        const int dr = jacobi_dest_block(P, row / HB, S) * HB + row % HB;
        const int64_t o = (int64_t)dr * ld + (int64_t)Q * NT + col;
        Zr_out[o] = xr[a][r];
        if constexpr (CPLX) Zi_out[o] = xi[a][r];
      }
    }
    return;
  }

  __syncthreads();   // every wave is done reading T
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = ti * 16 + Mfma<double>::row(lane, r), col = tj * 16 + l15;
      Tr[row][col] = xr[a][r];
      if constexpr (CPLX) Ti[row][col] = xi[a][r];
    }
  }
  __syncthreads();

  // Y = X J_Q, scattered to the next round's slots (+ Hermitian mirror)
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int t = wave * NACC + a, ti = t / TPD, tj = t % TPD;
    d4_t yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0};
    for (int k0 = 0; k0 < NT; k0 += 4) {
      const int k = k0 + l4;
      const double xre = Tr[ti * 16 + l15][k];
      const double qr = JQr[k][tj * 16 + l15];
      yr = Mfma<double>::mma(xre, qr, yr);
      if constexpr (CPLX) {
        const double xim = Ti[ti * 16 + l15][k];
        const double qi = JQi[k][tj * 16 + l15];
        yr = Mfma<double>::mma(-xim, qi, yr);
        yi = Mfma<double>::mma(xre, qi, yi);
        yi = Mfma<double>::mma(xim, qr, yi);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = ti * 16 + Mfma<double>::row(lane, r), col = tj * 16 + l15;
      const int dr = jacobi_dest_block(P, row / HB, S) * HB + row % HB;
      const int dc = jacobi_dest_block(Q, col / HB, S) * HB + col % HB;
      Gr_out[(int64_t)dr * ld + dc] = yr[r];
      Gr_out[(int64_t)dc * ld + dr] = yr[r];
      if constexpr (CPLX) {
        Gi_out[(int64_t)dr * ld + dc] = yi[r];
        Gi_out[(int64_t)dc * ld + dr] = -yi[r];
      }
    }
  }
}

__global__ void jacobi_diag_kernel(const double* __restrict__ Gr, int npad, double* __restrict__ d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) d[i] = Gr[(int64_t)i * npad + i];
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

struct EvdWorkspace {
  DevBuf<double> G[2][2], Z[2][2];  // [ping-pong][plane]
  DevBuf<double> J[2], lam, diag, scal;
  DevBuf<unsigned long long> off;
  DevBuf<int> perm;
};

struct EvdInfo {
  int sweeps = 0;
  int tile = 0;
  int slots = 0;
  double last_off = 0.0;
};

// Hermitian EVD  A = U diag(lam) U^H, lam descending.
//   Ar/Ai : n x n row-major planes (Ai == nullptr for a real symmetric matrix), lda
//   lam_host : n eigenvalues (descending); lam_dev (nullable) gets the same on the device
//   Zr/Zi : n x n, row i = conj(u_i)   (ldz)
template <bool CPLX>
void hermitian_evd_impl(hipStream_t st, EvdWorkspace& ws, const double* Ar, const double* Ai, int n, int64_t lda,
                        std::vector<double>& lam_host, double* lam_dev, double* Zr, double* Zi, int64_t ldz, int nt,
                        double tol, int max_sweeps, EvdInfo* info) {
  const int NT = nt;
  const int S = std::max(ceil_div(n, NT), 1);
  const int npad = S * NT;
  const size_t nn = (size_t)npad * npad;
  for (int b = 0; b < 2; ++b) {
    ws.G[b][0].ensure(nn);
    ws.Z[b][0].ensure(nn);
    if (CPLX) { ws.G[b][1].ensure(nn); ws.Z[b][1].ensure(nn); }
  }
  ws.J[0].ensure((size_t)S * NT * NT);
  if (CPLX) ws.J[1].ensure((size_t)S * NT * NT);
  ws.lam.ensure((size_t)npad);
  ws.diag.ensure((size_t)npad);
  ws.scal.ensure(2);
  ws.off.ensure(1);
  ws.perm.ensure((size_t)npad);

  hipLaunchKernelGGL(jacobi_init_scale_kernel, dim3(1), dim3(256), 0, st, Ar, n, lda, tol, ws.scal.get());
  hipLaunchKernelGGL(jacobi_init_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, Ar, CPLX ? Ai : nullptr, n, lda,
                     ws.G[0][0].get(), CPLX ? ws.G[0][1].get() : nullptr, ws.Z[0][0].get(), CPLX ? ws.Z[0][1].get() : nullptr,
                     npad, ws.scal.get());
  XMCA_HIP(hipGetLastError());

  int cur = 0;
  const int rounds = (S == 1) ? 1 : 2 * S - 1;
  const int n_tiles = S * (S + 1) / 2 + S * S;
  int sweeps = 0;
  double off = 0.0;
  const double tile_tol = 2e-15;
  auto launch_round = [&](auto nt_tag) {
    constexpr int NTC = decltype(nt_tag)::value;
    hipLaunchKernelGGL((jacobi_tile_evd_kernel<NTC, CPLX>), dim3(S), dim3(256), 0, st, ws.G[cur][0].get(),
                       CPLX ? ws.G[cur][1].get() : nullptr, npad, ws.J[0].get(), CPLX ? ws.J[1].get() : nullptr,
                       ws.lam.get(), tile_tol, ws.scal.get(), ws.off.get(), S == 1 ? 60 : 30);
    hipLaunchKernelGGL((jacobi_update_kernel<NTC, CPLX>), dim3(n_tiles), dim3(256), 0, st, ws.G[cur][0].get(),
                       CPLX ? ws.G[cur][1].get() : nullptr, ws.G[cur ^ 1][0].get(), CPLX ? ws.G[cur ^ 1][1].get() : nullptr,
                       ws.Z[cur][0].get(), CPLX ? ws.Z[cur][1].get() : nullptr, ws.Z[cur ^ 1][0].get(),
                       CPLX ? ws.Z[cur ^ 1][1].get() : nullptr, ws.J[0].get(), CPLX ? ws.J[1].get() : nullptr, ws.lam.get(),
                       S, npad);
  };
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    XMCA_HIP(hipMemsetAsync(ws.off.get(), 0, sizeof(unsigned long long), st));
    for (int r = 0; r < rounds; ++r) {
      if constexpr (CPLX) {
        launch_round(std::integral_constant<int, 32>{});   // 64x64 complex tiles do not fit the LDS of the update kernel
      } else {
        if (NT == 32) launch_round(std::integral_constant<int, 32>{});
        else launch_round(std::integral_constant<int, 64>{});
      }
      cur ^= 1;
    }
    XMCA_HIP(hipGetLastError());
    unsigned long long bits = 0;
    XMCA_HIP(hipMemcpyAsync(&bits, ws.off.get(), sizeof(bits), hipMemcpyDeviceToHost, st));
    XMCA_HIP(hipStreamSynchronize(st));
    std::memcpy(&off, &bits, sizeof(double));
    ++sweeps;
    if (S == 1 || !(off >= tol) || !std::isfinite(off)) break;
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
  for (int i = 0; i < n; ++i) lam_host[i] = d[perm[i]];
  XMCA_HIP(hipMemcpyAsync(ws.perm.get(), perm.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
  if (lam_dev) XMCA_HIP(hipMemcpyAsync(lam_dev, lam_host.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
  if (Zr) {
    hipLaunchKernelGGL(jacobi_gather_kernel, dim3(std::min(ceil_div(n, 256), 64), n), dim3(256), 0, st, ws.Z[cur][0].get(),
                       CPLX ? ws.Z[cur][1].get() : nullptr, npad, ws.perm.get(), n, Zr, CPLX ? Zi : nullptr, ldz);
    XMCA_HIP(hipGetLastError());
  }
  XMCA_HIP(hipStreamSynchronize(st));   // perm / lam_host staging buffers go out of scope
  if (info) { info->sweeps = sweeps; info->tile = NT; info->slots = S; info->last_off = off; }
}

inline void hermitian_evd(hipStream_t st, EvdWorkspace& ws, const double* Ar, const double* Ai, int n, int64_t lda,
                          std::vector<double>& lam_host, double* lam_dev, double* Zr, double* Zi, int64_t ldz,
                          EvdInfo* info = nullptr, int force_tile = 0) {
  // stop after the first sweep whose largest off-diagonal entry (seen when its tile is visited) is below
  // 1e-10 * max|diag|: with the (at least fast-linear, normally quadratic) convergence the state left
  // behind is at the 1e-13 rotation floor.
  const double tol = 1e-10;
  const int max_sweeps = 40;
  if (Ai) {
    hermitian_evd_impl<true>(st, ws, Ar, Ai, n, lda, lam_host, lam_dev, Zr, Zi, ldz, 32, tol, max_sweeps, info);
  } else {
    int nt = force_tile ? force_tile : (n > 32 ? 64 : 32);
    hermitian_evd_impl<false>(st, ws, Ar, nullptr, n, lda, lam_host, lam_dev, Zr, nullptr, ldz, nt, tol, max_sweeps, info);
  }
}

}  // namespace xmca
