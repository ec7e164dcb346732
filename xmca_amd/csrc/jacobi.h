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
#include "tridiag.h"
#include "tridiag_vec.h"

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

// scal[0] = scale of the working matrix (max |diag|), scal[1] = absolute rotation floor (floor_rel * scale), scal[2] =
// diagonal value of the padding (below every eigenvalue; from scal[3] = squared Frobenius norm), scal[4] = factor applied to
// the input when the working copy is made: 1, or 1 / max |diag| for the single-precision sweeps (normalise != 0), whose
// working matrix then has scale 1
__global__ void jacobi_init_scale_kernel(const double* __restrict__ Ar, int n, int64_t lda, double floor_rel, int normalise, double* scal) {
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
    const double f = normalise ? 1.0 / g : 1.0;
    scal[0] = g * f;
    scal[1] = floor_rel * g * f;   // rotations below this absolute size are rounding noise of the null space
    scal[2] = -2.0 * fmax(sqrt(scal[3]) * f, g * f);
    scal[4] = f;
  }
}

struct EvdInfo {
  int sweeps = 0;
  int tile = 0;
  int slots = 0;
  double last_off = 0.0;
  int lr_step = 0;        // 1: a Cholesky LR step was inserted (graded spectrum)
  double diag_spread = 0; // q10/q90 of the diagonal when that was decided
  int tridiag = 0;        // 1: solved by reduction to tridiagonal form (tridiag.h), no Jacobi sweeps
};


// knobs of one eigensolver run (hermitian_evd_impl)
struct EvdParams {
  double tol = 1e-10;          // stop when the largest off-diagonal entry a sweep leaves behind is below tol * scale
  int max_sweeps = 50;
  double stall = 0.0;          // > 0: also stop when a sweep (from the third on) leaves more than stall * (previous) behind
  const double* Z0r = nullptr; // start basis instead of the identity (n x n planes, row i = i-th basis vector, conjugated)
  const double* Z0i = nullptr;
  int64_t ldz0 = 0;
  bool no_lr = false;          // never insert the Cholesky LR step
};

namespace jac64 {
using jreal = double;
using jacc_t = d4_t;
#include "jacobi_impl.inc"
}  // namespace jac64

struct EvdWorkspace {
  jac64::EvdWorkspaceT w64;
  GemmWorkspace gws;             // products of the tridiagonal route's back-transformation and clean-up
  TrdWorkspace trd;              // tridiagonal route (tridiag.h)
  TrdVecWorkspace trdv;          // ... its eigenvectors (tridiag_vec.h)
  DevBuf<double> lam_tmp;
};

// one run in one precision; tile size by problem kind (64 x 64 complex tiles do not fit the LDS of the update kernel)
#define XMCA_EVD_RUN(NS, WS)                                                                                                   \
  do {                                                                                                                         \
    if (Ai) {                                                                                                                  \
      NS::hermitian_evd_impl<true, 32>(st, WS, Ar, Ai, n, lda, lam_host, lam_dev, Zr, Zi, ldz, prm, info);                     \
    } else {                                                                                                                   \
      const int nt = force_tile ? force_tile : (n > 32 ? 64 : 32);                                                             \
      if (nt == 32) NS::hermitian_evd_impl<false, 32>(st, WS, Ar, nullptr, n, lda, lam_host, lam_dev, Zr, nullptr, ldz, prm, info); \
      else NS::hermitian_evd_impl<false, 64>(st, WS, Ar, nullptr, n, lda, lam_host, lam_dev, Zr, nullptr, ldz, prm, info);     \
    }                                                                                                                          \
  } while (0)

inline void hermitian_evd_f64(hipStream_t st, EvdWorkspace& ws, const double* Ar, const double* Ai, int n, int64_t lda,
                              std::vector<double>& lam_host, double* lam_dev, double* Zr, double* Zi, int64_t ldz,
                              const EvdParams& prm, EvdInfo* info, int force_tile) {
  XMCA_EVD_RUN(jac64, ws.w64);
}
#undef XMCA_EVD_RUN

// Hermitian EVD  A = U diag(lam) U^H, lam descending (see jacobi_impl.inc for the arguments).
//
// Two solvers share this entry point: the reduction to tridiagonal form (tridiag.h, tridiag_vec.h: eigenproblems of 192 and
// more without vectors, 768 and more with vectors) and the block Jacobi sweeps of this file (everything else, nearly
// diagonal problems, and spectra with clusters the tridiagonal route hands back).  The mixed-precision variant of the
// sweeps of rounds 1-2 (float sweeps, Newton-Schulz, double sweeps) was measured slower on MI355X and has been removed.
inline void hermitian_evd(hipStream_t st, EvdWorkspace& ws, const double* Ar, const double* Ai, int n, int64_t lda,
                          std::vector<double>& lam_host, double* lam_dev, double* Zr, double* Zi, int64_t ldz,
                          EvdInfo* info = nullptr, int force_tile = 0, bool nearly_diagonal = false) {
  // `nearly_diagonal`: the caller knows that a few Jacobi sweeps finish the problem (the weak block of solver.h, three
  // sweeps) - cheaper than any reduction, whose cost does not depend on the matrix.
  // eigenvalues only (rule_n without rotation, every n_vec = 0 solve): Householder tridiagonalisation + Sturm multisection
  // (tridiag.h) - (4/3) n^3 flop in n launches instead of ~11 sweeps of 4 n^3.  XMCA_TRIDIAG=0 keeps the Jacobi sweeps.
  const int trd_min_n = [] { const char* e = std::getenv("XMCA_TRIDIAG_MIN_N"); return e ? std::atoi(e) : 192; }();
  if (!Zr && !nearly_diagonal && trd_enabled() && n >= trd_min_n && trd_fits(n, Ai != nullptr)) {
    TrdParams P = trd_reduce(st, ws.trd, Ar, Ai, n, lda, false);
    if (xmca_trace("trdsum")) {       // (debug: checksums of the input matrix and of (d, e) - pairs a reduction's output with its input)
      std::vector<double> in((size_t)n * lda), d(n), e(n);
      XMCA_HIP(hipStreamSynchronize(st));
      XMCA_HIP(hipMemcpy(in.data(), Ar, sizeof(double) * in.size(), hipMemcpyDeviceToHost));
      XMCA_HIP(hipMemcpy(d.data(), P.d, sizeof(double) * n, hipMemcpyDeviceToHost));
      XMCA_HIP(hipMemcpy(e.data(), P.e, sizeof(double) * n, hipMemcpyDeviceToHost));
      unsigned long long ci = 1469598103934665603ull, co = ci;
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { unsigned long long b; std::memcpy(&b, &in[(size_t)i * lda + j], 8); ci = (ci ^ b) * 1099511628211ull; }
      for (int i = 0; i < n; ++i) { unsigned long long b; std::memcpy(&b, &d[i], 8); co = (co ^ b) * 1099511628211ull; if (i + 1 < n) { std::memcpy(&b, &e[i], 8); co = (co ^ b) * 1099511628211ull; } }
      std::fprintf(stderr, "trdsum n=%d in=%016llx out=%016llx\n", n, ci, co);
      if (const char* dir = std::getenv("XMCA_TRD_DUMP_DIR")) {      // (d, e) of every distinct (input, output) pair: scripts/de_diff.py
        char name[512];
        std::snprintf(name, sizeof(name), "%s/de_%016llx_%016llx.bin", dir, ci, co);
        if (FILE* f0 = std::fopen(name, "rb")) std::fclose(f0);
        else if (FILE* f1 = std::fopen(name, "wb")) { std::fwrite(d.data(), 8, n, f1); std::fwrite(e.data(), 8, n, f1); std::fclose(f1); }
      }
    }
    trd_eigenvalues(st, ws.trd, P, lam_host, lam_dev, ws.lam_tmp);
    if (info) {
      *info = EvdInfo{};
      info->tridiag = 1;
    }
    return;
  }
  // with eigenvectors: the same reduction, twisted-factorisation vectors of the tridiagonal matrix, back-transformation by
  // blocked reflectors and a Newton-Schulz clean-up (tridiag_vec.h).  Spectra with clusters the clean-up cannot repair
  // (repeated eigenvalues, null spaces of dimension > 1) come back here and take the Jacobi sweeps below.
  const int trd_vec_min_n = [] { const char* e = std::getenv("XMCA_TRIDIAG_VEC_MIN_N"); return e ? std::atoi(e) : 768; }();
  if (Zr && !nearly_diagonal && trd_enabled() && n >= trd_vec_min_n && trd_fits(n, Ai != nullptr)) {
    ws.trdv.count_call();
    TrdParams P = trd_reduce(st, ws.trd, Ar, Ai, n, lda, true);
    trd_wy_prepare(st, ws.trdv, ws.gws, P, Ai != nullptr);  // (from the second call on: on a second stream, under the two kernels below)
    std::vector<double> lam_t;
    trd_eigenvalues(st, ws.trd, P, lam_t, lam_dev, ws.lam_tmp, ws.trdv.lam_asc.ensure((size_t)n));
    // Eigenvalues that coincide to the last bit of the largest one (a null space of dimension > 1: repeated samples, low-rank
    // fields) are not resolved by the twisted vectors - the clean-up would find that out only after the back-transformation
    // and two n^3 products (advisor, round 3); they are on the host already, so the sweeps below start right away.
    bool repeated = false;
    {
      const double lmax = n > 0 ? std::max(std::fabs(lam_t.front()), std::fabs(lam_t.back())) : 0.0;
      for (int i = 0; i + 1 < n && !repeated; ++i) repeated = lam_t[(size_t)i] - lam_t[(size_t)i + 1] <= 2.2e-16 * lmax;
      if (repeated && xmca_trace("solve")) std::fprintf(stderr, "xmca: eigh n = %d: repeated eigenvalues - block Jacobi instead of twisted vectors\n", n);
    }
    if (repeated) {
      // The compact-WY factors queued above are not going to be used: wait for them here (second stream) so that they do not
      // run beside the Jacobi rounds, and leave the workspace as if nothing had been prepared (advisor, round 4).
      if (ws.trdv.prepared && !ws.trdv.joined && ws.trdv.side) XMCA_HIP(hipStreamSynchronize(ws.trdv.side));
      ws.trdv.prepared = false;
      ws.trdv.joined = true;
    }
    if (!repeated && trd_eigenvectors(st, ws.trd, ws.trdv, ws.gws, P, Ai != nullptr, Zr, Zi, ldz)) {
      XMCA_HIP(hipStreamSynchronize(st));
      lam_host = lam_t;
      if (info) {
        *info = EvdInfo{};
        info->tridiag = 1;
      }
      return;
    }
  }
  // stop after the first sweep that leaves no off-diagonal entry above 1e-10 * max|diag| behind: with the (at least
  // fast-linear, normally quadratic) convergence the next sweep would only confirm it
  EvdParams prm;
  // eigenvalues only (rule_n): an eigenvalue is off by sum_j |g_ij|^2 / (lam_i - lam_j), second order in what is left -
  // 1e-8 left behind bounds that by ~n 1e-16 lam_max for a spectrum without exact clusters, and the sweep that would
  // push the vectors' first-order error down is not needed (C4 surrogates: 12 -> 11 sweeps)
  if (!Zr) prm.tol = 1e-8;
  hermitian_evd_f64(st, ws, Ar, Ai, n, lda, lam_host, lam_dev, Zr, Zi, ldz, prm, info, force_tile);
}

}  // namespace xmca
