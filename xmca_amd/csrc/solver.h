// Device pipelines behind the C ABI: solve (covariance / Gram + eigen stage + back-projection),
// rotate (Varimax / Promax) and the Rule-N surrogate loop.  Formulation and reference line numbers:
// DESIGN.md sections 2-4.
#pragma once
#include <atomic>
#include <complex>
#include <map>
#include <memory>

#include "common.h"
#include "gemm.h"
#include "cholesky.h"
#include "fft.h"
#include "jacobi.h"
#include "kernels.h"
#include "rotate.h"

namespace xmca {

// ---------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------
struct StageTimer {
  // hipEvent based per-stage timers (ms), accumulated per name
  std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> open;
  std::map<std::string, double> ms;
  std::vector<std::string> order;
  hipStream_t st = nullptr;
  bool enabled = true;
  void begin(const std::string& name) {
    if (!enabled) return;
    hipEvent_t a, b;
    XMCA_HIP(hipEventCreate(&a));
    XMCA_HIP(hipEventCreate(&b));
    XMCA_HIP(hipEventRecord(a, st));
    open.push_back({name, {a, b}});
  }
  void end() {
    if (!enabled) return;
    XMCA_HIP(hipEventRecord(open.back().second.second, st));
    pending.push_back(open.back());
    open.pop_back();
  }
  std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
  void collect() {
    for (auto& e : pending) {
      XMCA_HIP(hipEventSynchronize(e.second.second));
      float t = 0.f;
      XMCA_HIP(hipEventElapsedTime(&t, e.second.first, e.second.second));
      if (!ms.count(e.first)) order.push_back(e.first);
      ms[e.first] += t;
      (void)hipEventDestroy(e.second.first);
      (void)hipEventDestroy(e.second.second);
    }
    pending.clear();
  }
  void reset() { collect(); ms.clear(); order.clear(); }
};

// complex matrix as two f64 planes on the device (im.p == nullptr for real data)
struct CPlanes {
  DevBuf<double> re, im;
  void ensure(size_t n, bool cplx) {
    re.ensure(n);
    if (cplx) im.ensure(n);
  }
  double* r() const { return re.get(); }
  double* i(bool cplx) const { return cplx ? im.get() : nullptr; }
};

// One input field: T x N row-major, element type TI (f32 or f64), optional imaginary plane.
template <typename TI>
struct FieldData {
  int64_t T = 0, N = 0;
  DevBuf<TI> re, im;
  const TI* ext_re = nullptr;   // adopted external device pointers (not owned)
  const TI* ext_im = nullptr;
  bool has_im = false;
  const TI* r() const { return ext_re ? ext_re : re.get(); }
  const TI* i() const { return has_im ? (ext_im ? ext_im : im.get()) : nullptr; }
};

template <typename TO>
static void launch_convert(hipStream_t st, const double* in, TO* out, int64_t n) {
  if (n <= 0) return;
  hipLaunchKernelGGL((convert_kernel<double, TO>), ew_grid(n), dim3(EW_BLOCK), 0, st, in, out, n);
  XMCA_HIP(hipGetLastError());
}

// f64 planes -> planes of the field's element type (identity for f64: returns the inputs)
template <typename TI>
struct Narrow {
  DevBuf<TI> re, im;
  const TI* r = nullptr;
  const TI* i = nullptr;
  void from(hipStream_t st, const double* pr, const double* pi, int64_t n) {
    if constexpr (std::is_same<TI, double>::value) {
      r = pr; i = pi;
    } else {
      launch_convert<TI>(st, pr, re.ensure((size_t)n), n);
      r = re.get();
      i = nullptr;
      if (pi) { launch_convert<TI>(st, pi, im.ensure((size_t)n), n); i = im.get(); }
    }
  }
};

struct SolveResult {
  int rank = 0;
  int n_vec = 0;                       // back-projected modes
  std::vector<double> sigma;           // singular values of A^H B / (T-1), descending
  CPlanes Vt[2];                       // n_vec x N_k planes (row m = mode m)
  DevBuf<float> Vt32[2];               // ... or, vt_f32[side], ONE float32 plane: a real float32 field reduced on the dual side keeps
  bool vt_f32[2] = {false, false};     // its vectors in the model's dtype, as the reference does (`_V` follows the input, array.py:584)
  int64_t ldv[2] = {0, 0};
  bool cplx = false;
  bool weak_refined = false;          // the route refined its weak modes itself (Solver::refine_weak_block)
  int consistent = 0;                 // leading modes whose sigma passed the route's own consistency check (solve_one_sided)
  EvdInfo evd_info[3];
};

// ---------------------------------------------------------------------------------------------------
// solve
// ---------------------------------------------------------------------------------------------------
template <typename TI>
class Solver {
 public:
  hipStream_t st;
  GemmWorkspace& gws;
  EvdWorkspace& ews;
  StageTimer& tm;
  bool f32_vectors = false;    // xmca_solve on float32 fields: vectors of the one-field dual route stay float32 (SolveResult::Vt32)
  Solver(hipStream_t s, GemmWorkspace& g, EvdWorkspace& e, StageTimer& t) : st(s), gws(g), ews(e), tm(t) {}

  // per-field reduction: eigen-decomposition of the T x T Gram matrix when N > T
  struct Reduced {
    bool reduced = false;
    int r = 0;                 // min(T, N)
    CPlanes Z;                 // r x T : row i = conj(u_i)          (reduced only)
    DevBuf<double> s;          // r     : singular values of the field (reduced only)
    std::vector<double> lam;
  };

  // G = X X^H (T x T, both triangles):  Gr = Xr Xr^T + Xi Xi^T ;  Gi = Xi Xr^T - Xr Xi^T
  void gram(const FieldData<TI>& f, bool cplx, CPlanes& G) {
    const int T = (int)f.T;
    G.ensure((size_t)T * T, cplx);
    tm.begin("gram");
    cgemm<TI>(st, gws, f.r(), f.i(), f.N, true, false, f.r(), f.i(), f.N, false, true, G.r(), G.i(cplx), T, T, T, (int)f.N, 1.0,
              nullptr, nullptr, true);
    tm.end();
  }

  void reduce_field(const FieldData<TI>& f, bool cplx, Reduced& R, CPlanes& G, EvdInfo* info, bool want_vectors = true) {
    const int T = (int)f.T;
    R.reduced = f.N > f.T;
    R.r = (int)std::min(f.T, f.N);
    if (!R.reduced) return;
    gram(f, cplx, G);
    reduce_field_gram(f, cplx, R, G, info, want_vectors);
  }
  // ... from the Gram matrix G of f (T x T)
  void reduce_field_gram(const FieldData<TI>& f, bool cplx, Reduced& R, const CPlanes& G, EvdInfo* info, bool want_vectors = true) {
    const int T = (int)f.T;
    R.reduced = true;
    R.r = T;
    tm.begin("eigh");
    if (want_vectors) R.Z.ensure((size_t)T * T, cplx);
    R.s.ensure((size_t)T);
    DevBuf<double> lam_dev;
    lam_dev.ensure((size_t)T);
    hermitian_evd(st, ews, G.r(), G.i(cplx), T, T, R.lam, lam_dev.get(), want_vectors ? R.Z.r() : nullptr,
                  want_vectors ? R.Z.i(cplx) : nullptr, T, info);
    hipLaunchKernelGGL(sqrt_clamp_kernel, dim3(ceil_div(T, 256)), dim3(256), 0, st, lam_dev.get(), R.s.get(), T, 1.0);
    XMCA_HIP(hipStreamSynchronize(st));
    tm.end();
  }

  // rows-normalised conj(Yh * X~) -> Vt  (Yh: m x T f64 planes).
  // `row_norm` (host, nullable): the norms of the first `n_known` rows when the caller knows them - for the rows of an
  // eigenvector basis of the Gram matrix, ||u_k^H X~|| = sqrt(lambda_k) - so that 1 / norm rides in the GEMM epilogue and the
  // conjugation in the operand flags: no second pass over the m x N result (C5: 6.4 ms and 10 GB of traffic).  Rows
  // beyond n_known (null modes: their norm is rounding noise) are normalised by what they are.
  // `Vt32` (float32 fields, real): the result is written in float32 by the GEMM's epilogue - the reference's `_V` has the
  // input's dtype (array.py:584), and at C5 the float64 result is a 10 GB store against 5 GB.
  void back_project(const FieldData<TI>& f, bool cplx, const double* Yr, const double* Yi, int m, CPlanes& Vt,
                    const double* row_norm = nullptr, int n_known = 0, DevBuf<float>* Vt32 = nullptr) {
    const int T = (int)f.T;
    Narrow<TI> y;
    y.from(st, Yr, Yi, (int64_t)m * T);
    DevBuf<double> inv_dev;
    std::vector<double> inv_host;
    if (!row_norm) n_known = 0;
    if constexpr (std::is_same<TI, float>::value) {
      if (Vt32 && !cplx) {
        Vt.re.release();                   // (the float64 planes of an earlier model are not part of this result)
        Vt.im.release();
        float* V = Vt32->ensure((size_t)m * f.N);
        GemmOpts o;
        o.a_kfast = true; o.b_nfast = true;
        if (n_known > 0) {
          inv_host.assign((size_t)m, 1.0);
          for (int k = 0; k < n_known; ++k) inv_host[(size_t)k] = 1.0 / row_norm[k];
          XMCA_HIP(hipMemcpyAsync(inv_dev.ensure((size_t)m), inv_host.data(), sizeof(double) * m, hipMemcpyHostToDevice, st));
          o.row_scale = inv_dev.get();
        }
        gemm<float, float>(st, gws, y.r, T, f.r(), f.N, V, f.N, m, (int)f.N, T, o);
        if (n_known < m)
          normalize_rows<float>(st, V + (int64_t)n_known * f.N, nullptr, f.N, m - n_known, (int)f.N, 0, nullptr);
        XMCA_HIP(hipGetLastError());
        XMCA_HIP(hipStreamSynchronize(st));
        return;
      }
    }
    Vt.ensure((size_t)m * f.N, cplx);
    if (n_known > 0) {
      inv_host.assign((size_t)m, 1.0);
      for (int k = 0; k < n_known; ++k) inv_host[(size_t)k] = 1.0 / row_norm[k];
      XMCA_HIP(hipMemcpyAsync(inv_dev.ensure((size_t)m), inv_host.data(), sizeof(double) * m, hipMemcpyHostToDevice, st));
      // conj(Yh X~) = conj(Yh) conj(X~)
      cgemm<TI>(st, gws, y.r, y.i, T, true, true, f.r(), f.i(), f.N, true, true, Vt.r(), Vt.i(cplx), f.N, m, (int)f.N, T, 1.0,
                inv_dev.get(), nullptr, false);
      if (n_known < m)
        normalize_rows<double>(st, Vt.r() + (int64_t)n_known * f.N, cplx ? Vt.i(cplx) + (int64_t)n_known * f.N : nullptr, f.N, m - n_known,
                               (int)f.N, 0, nullptr);
    } else {
      cgemm<TI>(st, gws, y.r, y.i, T, true, false, f.r(), f.i(), f.N, true, false, Vt.r(), Vt.i(cplx), f.N, m, (int)f.N, T, 1.0,
                nullptr, nullptr, false);
      normalize_rows<double>(st, Vt.r(), Vt.i(cplx), f.N, m, (int)f.N, 1, nullptr);
    }
    XMCA_HIP(hipGetLastError());
    XMCA_HIP(hipStreamSynchronize(st));   // `y` temporaries are released on return
  }

  // n_vec < 0: all modes
  void solve(const FieldData<TI>* fields, int n_fields, bool cplx, int n_vec_req, SolveResult& out) {
    out.weak_refined = false;            // (the caller may hand in the result object of an earlier solve)
    out.consistent = 0;
    out.vt_f32[0] = out.vt_f32[1] = false;
    for (EvdInfo& e : out.evd_info) e = EvdInfo();
    solve_core(fields, n_fields, cplx, n_vec_req, out);
    if (n_fields == 2) refine_by_deflation(fields, cplx, out);
    drop_stale_vectors(out);
  }

  // A result object keeps float64 planes AND a float32 plane per side; only one of them is written by a solve (vt_f32[side]
  // says which).  The other one may be left from an earlier model on the same handle - 5 GB at C5 - and must not be reachable
  // as if it belonged to this result: give it back (advisor, round 4).
  static void drop_stale_vectors(SolveResult& out) {
    for (int s = 0; s < 2; ++s) {
      if (out.vt_f32[s]) { out.Vt[s].re.release(); out.Vt[s].im.release(); }
      else out.Vt32[s].release();
    }
  }

  // -------------------------------------------------------------------------------------------------------------
  // Weak modes of a two-field model.  sigma^2 comes out of the eigen-decomposition of K K^H (or K^H K), formed with an
  // absolute error of ~1e-13 sigma_1^2: sigma_m and its vectors carry a relative error of ~5e-14 (sigma_1 / sigma_m)^2,
  // 1e-5 down to ~2e-4 sigma_1 for sigma and ~1e-3 sigma_1 for the vectors (their error carries another factor lambda / gap) -
  // the reference's gesdd works on K itself (array.py:569-578) and keeps eps sigma_1 / sigma_m.
  // The modes above 1e-3 of the current top are accurate; they are DEFLATED and the rest is solved again:
  //   U_a, U_b = the accurate singular vectors so far, re-orthonormalised (two Newton-Schulz steps on their Gram matrix);
  //   X~a' = X~a (I - U_a U_a^H),  X~b' = X~b (I - U_b U_b^H)   (two tall GEMMs per field, from the ORIGINAL fields);
  //   C' = X~a'^H X~b' / dof = the remaining part of C: the same solver on (X~a', X~b') sees sigma'_1 = the largest
  //   remaining singular value and resolves another 3.7 decades; its modes are spliced in behind the deflated ones.
  // Up to three levels (sigma down to ~1e-11 sigma_1).  Measured on a spectrum graded over 10 decades (T = 300, N = 700 /
  // 650; scripts/deflation_probe.py): every mode down to 1e-9 sigma_1 at sigma 2e-12, vectors 3e-7, orthonormality 1e-7,
  // where the single solve loses sigma below 1e-5 sigma_1 altogether.  Costs one more solve per level and is only paid
  // when a mode with vectors lies below 1e-3 of the top (XMCA_DEFLATE_BELOW; 0 switches it off); rule_n / bootstrap
  // replicates (leading modes or values only) and the bench configurations (one field, or analytic) never do.  float64 fields
  // on the general path only: float32 fields cannot be deflated to better than 6e-8 sigma_1, and the analytic-signal
  // subspace path (whose fields are implicit) is measured accurate to 1e-6 sigma_1 as it is (profiles/r02_two_field_*).
  // -------------------------------------------------------------------------------------------------------------
  // U = X~ V for the k rows of (Vr, Vi) (k x N planes: row j = mode j): T x k planes
  void project_modes(const FieldData<TI>& f, bool cplx, const double* Vr, const double* Vi, int k, CPlanes& U) {
    const int T = (int)f.T;
    const int64_t N = f.N;
    U.ensure((size_t)T * k, cplx);
    Narrow<TI> v;
    v.from(st, Vr, Vi, (int64_t)k * N);
    // U[t][j] = sum_n X~[t][n] V[n][j]  with  V[n][j] = Vt[j][n]
    cgemm<TI>(st, gws, f.r(), f.i(), N, true, false, v.r, v.i, N, false, false, U.r(), U.i(cplx), k, T, k, (int)N, 1.0, nullptr, nullptr,
              false);
    XMCA_HIP(hipStreamSynchronize(st));   // `v` is released on return
  }

  void refine_by_deflation(const FieldData<TI>* fields, bool cplx, SolveResult& out) {
    if constexpr (!std::is_same<TI, double>::value) {
      return;
    } else {
      static const double thr_plain = [] { const char* e = std::getenv("XMCA_DEFLATE_BELOW"); return e ? std::atof(e) : 1e-3; }();   // 0: off
      const int n_vec = out.n_vec;
      if (thr_plain <= 0.0 || n_vec <= 1 || out.sigma.empty() || !(out.sigma[0] > 0.0)) return;
      const int T = (int)fields[0].T;
      int done = 0;
      bool tail_refined = out.weak_refined;      // the solve that produced the current tail refined its weak block itself:
      int tail_consistent = out.weak_refined ? out.consistent : 0;   // ... and checked this many of its leading modes
      for (int level = 0; level < 3; ++level) {  // its modes are good down to 1e-5 of its top instead of 1e-3
        const double thr = tail_refined ? std::min(thr_plain, 1e-5) : thr_plain;
        const double top = out.sigma[done];
        if (!(top > 0.0)) break;
        int ns = done;
        while (ns < n_vec && (out.sigma[ns] >= thr * top || ns - done < tail_consistent)) ++ns;
        if (ns >= n_vec || ns == done) break;                            // nothing with vectors lies below
        if (!(out.sigma[ns] > 1e-13 * out.sigma[0])) break;              // what is left is null
        tm.begin("deflate");
        CPlanes Us[2], Xd[2];
        FieldData<double> fd[2];
        for (int s = 0; s < 2; ++s) {
          const int64_t N = fields[s].N;
          const size_t nu = (size_t)ns * N;
          CPlanes G, Tmp;
          Us[s].ensure(nu, cplx);
          Tmp.ensure(nu, cplx);
          G.ensure((size_t)ns * ns, cplx);
          XMCA_HIP(hipMemcpyAsync(Us[s].r(), out.Vt[s].r(), sizeof(double) * nu, hipMemcpyDeviceToDevice, st));
          if (cplx) XMCA_HIP(hipMemcpyAsync(Us[s].im.get(), out.Vt[s].im.get(), sizeof(double) * nu, hipMemcpyDeviceToDevice, st));
          for (int it = 0; it < 2; ++it) {       // rows <- (1.5 I - 0.5 U U^H) rows
            cgemm<double>(st, gws, Us[s].r(), Us[s].i(cplx), N, true, false, Us[s].r(), Us[s].i(cplx), N, false, true, G.r(), G.i(cplx), ns, ns,
                          ns, (int)N, -0.5, nullptr, nullptr, true);
            hipLaunchKernelGGL(add_diag_kernel, dim3(ceil_div(ns, 256)), dim3(256), 0, st, G.r(), (int64_t)ns, ns, 1.5);
            cgemm<double>(st, gws, G.r(), G.i(cplx), ns, true, false, Us[s].r(), Us[s].i(cplx), N, true, false, Tmp.r(), Tmp.i(cplx), N, ns,
                          (int)N, ns, 1.0, nullptr, nullptr, false);
            XMCA_HIP(hipMemcpyAsync(Us[s].r(), Tmp.r(), sizeof(double) * nu, hipMemcpyDeviceToDevice, st));
            if (cplx) XMCA_HIP(hipMemcpyAsync(Us[s].im.get(), Tmp.im.get(), sizeof(double) * nu, hipMemcpyDeviceToDevice, st));
          }
          CPlanes W;
          project_modes(fields[s], cplx, Us[s].r(), Us[s].i(cplx), ns, W);
          // X' = X - W U^H  (T x N):  B(k = j, n) = conj(U[n][j]) = conj(Us[j][n])
          const size_t nx = (size_t)T * N;
          Xd[s].ensure(nx, cplx);
          XMCA_HIP(hipMemcpyAsync(Xd[s].r(), fields[s].r(), sizeof(double) * nx, hipMemcpyDeviceToDevice, st));
          if (cplx) XMCA_HIP(hipMemcpyAsync(Xd[s].im.get(), fields[s].i(), sizeof(double) * nx, hipMemcpyDeviceToDevice, st));
          cgemm<double>(st, gws, W.r(), W.i(cplx), ns, true, false, Us[s].r(), Us[s].i(cplx), N, true, true, Xd[s].r(), Xd[s].i(cplx), N, T,
                        (int)N, ns, -1.0, nullptr, nullptr, false, 1.0);
          XMCA_HIP(hipGetLastError());
          XMCA_HIP(hipStreamSynchronize(st));     // temporaries of this side are released
          fd[s].T = T;
          fd[s].N = N;
          fd[s].ext_re = Xd[s].r();
          fd[s].ext_im = cplx ? Xd[s].im.get() : nullptr;
          fd[s].has_im = cplx;
        }
        tm.end();
        SolveResult r2;
        solve_core(fd, 2, cplx, n_vec - ns, r2);
        tail_refined = r2.weak_refined;
        tail_consistent = r2.weak_refined ? r2.consistent : 0;
        const int k = std::min(n_vec - ns, r2.n_vec);
        for (int j = 0; j < k && ns + j < (int)out.sigma.size(); ++j) out.sigma[ns + j] = r2.sigma[j];
        for (int s = 0; s < 2; ++s) {
          const int64_t N = fields[s].N;
          XMCA_HIP(hipMemcpyAsync(out.Vt[s].r() + (int64_t)ns * N, r2.Vt[s].r(), sizeof(double) * (size_t)k * N, hipMemcpyDeviceToDevice, st));
          if (cplx)
            XMCA_HIP(hipMemcpyAsync(out.Vt[s].im.get() + (int64_t)ns * N, r2.Vt[s].im.get(), sizeof(double) * (size_t)k * N,
                                    hipMemcpyDeviceToDevice, st));
        }
        XMCA_HIP(hipStreamSynchronize(st));
        done = ns;
      }
    }
  }

  void solve_core(const FieldData<TI>* fields, int n_fields, bool cplx, int n_vec_req, SolveResult& out) {
    const FieldData<TI>& A = fields[0];
    const int T = (int)A.T;
    const double dof = (double)(T - 1);
    out.cplx = cplx;
    Reduced Ra, Rb;
    CPlanes G;
    if (n_fields == 2 && n_vec_req == 0 && A.N > A.T && fields[1].N > fields[1].T && cholesky_enabled()) {
      CPlanes Ga, Gb;
      gram(A, cplx, Ga);
      gram(fields[1], cplx, Gb);
      if (values_by_cholesky(Ga, Gb, T, cplx, dof, out)) {
        out.rank = T;
        out.n_vec = 0;
        out.ldv[0] = A.N;
        out.ldv[1] = fields[1].N;
        return;
      }
    }
    static const bool one_sided_on = [] { const char* e = std::getenv("XMCA_ONE_SIDED"); return !(e && e[0] == '0'); }();
    if (n_fields == 2 && one_sided_on && A.N > A.T && fields[1].N > fields[1].T) {
      // two wide fields: the first one enters through a factor M, M^H M = G_a - its Cholesky factor when that passes the
      // guards (factor_by_cholesky, solve_one_sided), else its eigen-factor
      gram(A, cplx, G);
      CPlanes Fm;
      if (factor_by_cholesky(G, T, cplx, rows_sum_to_zero(G, T, cplx), Fm) &&
          solve_one_sided(A, fields[1], cplx, Fm, nullptr, n_vec_req, out, G, true))
        return;
      reduce_field_gram(A, cplx, Ra, G, &out.evd_info[0], true);
      solve_one_sided(A, fields[1], cplx, Ra.Z, Ra.s.get(), n_vec_req, out, G, false);
      return;
    }
    reduce_field(A, cplx, Ra, G, &out.evd_info[0], n_fields == 2 || n_vec_req != 0);

    if (n_fields == 1) {
      if (Ra.reduced) {
        out.rank = T;
        out.sigma.resize(T);
        for (int i = 0; i < T; ++i) out.sigma[i] = std::max(Ra.lam[i], 0.0) / dof;
        const int m = n_vec_req < 0 ? T : std::min(n_vec_req, T);
        out.n_vec = m;
        out.ldv[0] = A.N;
        tm.begin("backproject");
        if (m > 0) {
          // ||u_k^H X~|| = sqrt(lambda_k) for every mode clear of the rounding noise of the Gram matrix: lambda_k carries an
          // absolute error of ~1e-7 lambda_0 for float32 fields (f32 products) and ~1e-14 lambda_0 for float64 ones (the
          // tridiagonal route), so 1 / sqrt(lambda_k) is a unit-norm scale to 1e-3 / 1e-8 above these cuts; the modes below
          // them are normalised by their measured norm (advisor, round 3: a dtype-blind 1e-9 left weak float32 modes off unit norm)
          const double cut = (std::is_same<TI, float>::value ? 1e-4 : 1e-6) * Ra.lam[0];
          std::vector<double> nrm((size_t)m, 0.0);
          int known = 0;
          while (known < m && Ra.lam[(size_t)known] > cut) { nrm[(size_t)known] = std::sqrt(Ra.lam[(size_t)known]); ++known; }
          const bool v32 = f32_vectors && std::is_same<TI, float>::value && !cplx;
          out.vt_f32[0] = v32;
          back_project(A, cplx, Ra.Z.r(), Ra.Z.i(cplx), m, out.Vt[0], nrm.data(), known, v32 ? &out.Vt32[0] : nullptr);
        }
        tm.end();
      } else {
        // primal: C = X^H X / dof  (N x N), V = eigenvectors
        const int N = (int)A.N;
        CPlanes C;
        C.ensure((size_t)N * N, cplx);
        tm.begin("gram");
        cgemm<TI>(st, gws, A.r(), A.i(), A.N, false, true, A.r(), A.i(), A.N, true, false, C.r(), C.i(cplx), N, N, N, T,
                  1.0 / dof, nullptr, nullptr, true);
        tm.end();
        tm.begin("eigh");
        std::vector<double> lam;
        CPlanes Z;
        Z.ensure((size_t)N * N, cplx);
        hermitian_evd(st, ews, C.r(), C.i(cplx), N, N, lam, nullptr, Z.r(), Z.i(cplx), N, &out.evd_info[0]);
        tm.end();
        out.rank = N;
        out.sigma.resize(N);
        for (int i = 0; i < N; ++i) out.sigma[i] = std::max(lam[i], 0.0);
        const int m = n_vec_req < 0 ? N : std::min(n_vec_req, N);
        out.n_vec = m;
        out.ldv[0] = N;
        out.Vt[0].ensure((size_t)m * N, cplx);
        // Vt[i][n] = V[n][i] = conj(Z[i][n])
        XMCA_HIP(hipMemcpyAsync(out.Vt[0].r(), Z.r(), sizeof(double) * (size_t)m * N, hipMemcpyDeviceToDevice, st));
        if (cplx) {
          XMCA_HIP(hipMemcpyAsync(out.Vt[0].im.get(), Z.im.get(), sizeof(double) * (size_t)m * N, hipMemcpyDeviceToDevice, st));
          DevBuf<double> minus1;
          // negate the imaginary plane: scale by -1 through the column-scale kernel with a 1-element trick is
          // overkill; a dedicated lambda kernel keeps it simple
          negate(out.Vt[0].im.get(), (int64_t)m * N);
        }
        XMCA_HIP(hipStreamSynchronize(st));
      }
      return;
    }

    // ------------------------------- two fields ------------------------------------------------
    const FieldData<TI>& B = fields[1];
    reduce_field(B, cplx, Rb, G, &out.evd_info[1]);
    const int ra = Ra.r, rb = Rb.r;
    const int rank = std::min(ra, rb);
    out.rank = rank;

    // K = F_a^H F_b / dof   (ra x rb)
    CPlanes K;
    K.ensure((size_t)ra * rb, cplx);
    tm.begin("kernel");
    {
      Narrow<TI> za, zb;
      if (Ra.reduced && Rb.reduced) {
        // K[i][j] = s_a,i s_b,j sum_t Za[i,t] conj(Zb[j,t]) / dof     (f64 x f64)
        cgemm<double>(st, gws, Ra.Z.r(), Ra.Z.i(cplx), T, true, false, Rb.Z.r(), Rb.Z.i(cplx), T, false, true, K.r(), K.i(cplx), rb,
                      ra, rb, T, 1.0 / dof, Ra.s.get(), Rb.s.get(), false);
      } else if (Ra.reduced && !Rb.reduced) {
        // K = S_a Za X~b / dof
        za.from(st, Ra.Z.r(), Ra.Z.i(cplx), (int64_t)ra * T);
        cgemm<TI>(st, gws, za.r, za.i, T, true, false, B.r(), B.i(), B.N, true, false, K.r(), K.i(cplx), rb, ra, rb, T, 1.0 / dof,
                  Ra.s.get(), nullptr, false);
      } else if (!Ra.reduced && Rb.reduced) {
        // K[n][j] = conj( sum_t X~a[t,n] Zb[j,t] ) s_b,j / dof
        zb.from(st, Rb.Z.r(), Rb.Z.i(cplx), (int64_t)rb * T);
        cgemm<TI>(st, gws, A.r(), A.i(), A.N, false, true, zb.r, zb.i, T, false, true, K.r(), K.i(cplx), rb, ra, rb, T, 1.0 / dof,
                  nullptr, Rb.s.get(), false);
      } else {
        // K = X~a^H X~b / dof
        cgemm<TI>(st, gws, A.r(), A.i(), A.N, false, true, B.r(), B.i(), B.N, true, false, K.r(), K.i(cplx), rb, ra, rb, T,
                  1.0 / dof, nullptr, nullptr, false);
      }
      XMCA_HIP(hipStreamSynchronize(st));
    }
    tm.end();

    // SVD of K through the Hermitian EVD of the smaller Gram matrix
    CPlanes H, Ph, Qh;
    Ph.ensure((size_t)rank * ra, cplx);
    Qh.ensure((size_t)rank * rb, cplx);
    std::vector<double> lam;
    tm.begin("kernel_svd");
    if (rb <= ra) {
      H.ensure((size_t)rb * rb, cplx);
      // H = K^H K
      cgemm<double>(st, gws, K.r(), K.i(cplx), rb, false, true, K.r(), K.i(cplx), rb, true, false, H.r(), H.i(cplx), rb, rb, rb, ra,
                    1.0, nullptr, nullptr, true);
      hermitian_evd(st, ews, H.r(), H.i(cplx), rb, rb, lam, nullptr, Qh.r(), Qh.i(cplx), rb, &out.evd_info[2]);
      // Ph = Qh K^H  (rows = conj(p_m)), then normalise rows
      cgemm<double>(st, gws, Qh.r(), Qh.i(cplx), rb, true, false, K.r(), K.i(cplx), rb, false, true, Ph.r(), Ph.i(cplx), ra, rank, ra,
                    rb, 1.0, nullptr, nullptr, false);
      hipLaunchKernelGGL((normalize_rows_kernel<double>), dim3(rank), dim3(256), 0, st, Ph.r(), Ph.i(cplx), (int64_t)ra, ra, 0,
                         (double*)nullptr);
    } else {
      H.ensure((size_t)ra * ra, cplx);
      // H = K K^H
      cgemm<double>(st, gws, K.r(), K.i(cplx), rb, true, false, K.r(), K.i(cplx), rb, false, true, H.r(), H.i(cplx), ra, ra, ra, rb,
                    1.0, nullptr, nullptr, true);
      hermitian_evd(st, ews, H.r(), H.i(cplx), ra, ra, lam, nullptr, Ph.r(), Ph.i(cplx), ra, &out.evd_info[2]);
      // Qh = Ph K
      cgemm<double>(st, gws, Ph.r(), Ph.i(cplx), ra, true, false, K.r(), K.i(cplx), rb, true, false, Qh.r(), Qh.i(cplx), rb, rank, rb,
                    ra, 1.0, nullptr, nullptr, false);
      hipLaunchKernelGGL((normalize_rows_kernel<double>), dim3(rank), dim3(256), 0, st, Qh.r(), Qh.i(cplx), (int64_t)rb, rb, 0,
                         (double*)nullptr);
    }
    XMCA_HIP(hipGetLastError());
    XMCA_HIP(hipStreamSynchronize(st));
    tm.end();
    out.sigma.resize(rank);
    for (int i = 0; i < rank; ++i) out.sigma[i] = std::sqrt(std::max(lam[i], 0.0));

    const int m = n_vec_req < 0 ? rank : std::min(n_vec_req, rank);
    out.n_vec = m;
    out.ldv[0] = A.N;
    out.ldv[1] = B.N;
    if (m == 0) return;
    tm.begin("backproject");
    project_side(A, B, cplx, Ra, Rb, Ph, Qh, ra, rb, m, out.Vt[0]);   // V_left  from (F_b Q)
    project_side(B, A, cplx, Rb, Ra, Qh, Ph, rb, ra, m, out.Vt[1]);   // V_right from (F_a P)
    tm.end();
  }

  // -------------------------------------------------------------------------------------------------------------
  // Singular values only (rule_n without rotation) of two wide fields: no field is diagonalised at all.  With the
  // Cholesky factor G_a + delta I = R^H R (R^H = F_a times a unitary matrix), sigma^2 dof^2 = eig(R G_b R^H): one blocked
  // Cholesky, two products and ONE values-only eigenproblem instead of an eigenproblem with vectors plus one without.
  // delta = 1e-13 max diag (2e-6 for f32 fields) makes the (centered, hence singular) Gram matrix definite; it moves
  // sigma^2 by that much, relative.
  // Returns false (nothing written) when the factorisation meets a non-positive pivot.
  // -------------------------------------------------------------------------------------------------------------
  static bool cholesky_enabled() {
    static const bool on = [] { const char* e = std::getenv("XMCA_CHOLESKY"); return !(e && e[0] == '0'); }();
    return on;
  }
  bool values_by_cholesky(CPlanes& Ga, CPlanes& Gb, int n, bool cplx, double dof, SolveResult& out) {
    tm.begin("cholesky");
    // (f32 fields: their Gram matrix carries f32 accumulation noise of ~1e-7 max diag, also below zero)
    const bool ok = cholesky_upper(st, gws, Ga.r(), Ga.i(cplx), n, n, std::is_same<TI, float>::value ? 2e-6 : 1e-13);
    tm.end();
    if (!ok) return false;
    CPlanes M1, H;
    M1.ensure((size_t)n * n, cplx);
    H.ensure((size_t)n * n, cplx);
    tm.begin("kernel");
    // M1 = R G_b ;  H = M1 R^H / dof^2
    cgemm<double>(st, gws, Ga.r(), Ga.i(cplx), n, true, false, Gb.r(), Gb.i(cplx), n, true, false, M1.r(), M1.i(cplx), n, n, n, n, 1.0,
                  nullptr, nullptr, false);
    cgemm<double>(st, gws, M1.r(), M1.i(cplx), n, true, false, Ga.r(), Ga.i(cplx), n, false, true, H.r(), H.i(cplx), n, n, n, n,
                  1.0 / (dof * dof), nullptr, nullptr, true);
    tm.end();
    std::vector<double> lam;
    tm.begin("kernel_svd");
    hermitian_evd(st, ews, H.r(), H.i(cplx), n, n, lam, nullptr, nullptr, nullptr, n, &out.evd_info[2]);
    XMCA_HIP(hipStreamSynchronize(st));
    tm.end();
    if ((int)out.sigma.size() < n) out.sigma.assign(n, 0.0);
    for (int i = 0; i < n; ++i) out.sigma[i] = std::sqrt(std::max(lam[i], 0.0));
    return true;
  }

  // -------------------------------------------------------------------------------------------------------------
  // Two fields, both wider than T: only ONE of them is decomposed.  With F_a = U_a S_a from G_a and the kernel
  // K = F_a^H F_b / dof of the reference (array.py:553-566), K K^H = F_a^H (F_b F_b^H) F_a / dof^2 = F_a^H G_b F_a / dof^2:
  // the Gram matrix of the second field enters as an operator and is never diagonalised (one T x T eigenproblem with
  // vectors less per solve).  Left small singular vectors p_m and sigma_m^2 = eigenpairs of H = K K^H; in grid space
  //   v_right,m ~ X~b^H (F_a p_m),    v_left,m ~ X~a^H (G_b F_a p_m) = dof C v_right,m     (rows normalised afterwards),
  // which also fixes the shared phase gauge u^H C v = +sigma.
  // -------------------------------------------------------------------------------------------------------------
  // Mf (T x T planes), ms (row scale or null): the factor M = diag(ms) Mf with M^H M = G_a (Ga).  guard: M is a Cholesky
  // factor on probation - the Gram matrix of the left vectors is checked in time-space coordinates (Thl G_a Thl^H, as in
  // solve_analytic) and `false` is returned, with nothing usable in `out`, when they are not orthogonal to 1e-6.
  bool solve_one_sided(const FieldData<TI>& A, const FieldData<TI>& B, bool cplx, const CPlanes& Mf, const double* ms, int n_vec_req,
                       SolveResult& out, const CPlanes& Ga, bool guard) {
    const int T = (int)A.T, ra = T;
    const double dof = (double)(T - 1);
    const int rank = ra;                       // = T = min(T, Nx, Ny)
    out.rank = rank;
    CPlanes Gb, M1, H, Ph;
    gram(B, cplx, Gb);
    tm.begin("kernel");
    M1.ensure((size_t)ra * T, cplx);
    H.ensure((size_t)ra * ra, cplx);
    // M1 = M G_b ;  H = M1 M^H / dof^2
    cgemm<double>(st, gws, Mf.r(), Mf.i(cplx), T, true, false, Gb.r(), Gb.i(cplx), T, true, false, M1.r(), M1.i(cplx), T, ra, T, T,
                  1.0, ms, nullptr, false);
    cgemm<double>(st, gws, M1.r(), M1.i(cplx), T, true, false, Mf.r(), Mf.i(cplx), T, false, true, H.r(), H.i(cplx), ra, ra, ra, T,
                  1.0 / (dof * dof), nullptr, ms, true);
    tm.end();
    const int m = n_vec_req < 0 ? rank : std::min(n_vec_req, rank);
    std::vector<double> lam;
    tm.begin("kernel_svd");
    if (m > 0) Ph.ensure((size_t)ra * ra, cplx);
    hermitian_evd(st, ews, H.r(), H.i(cplx), ra, ra, lam, nullptr, m > 0 ? Ph.r() : nullptr, m > 0 ? Ph.i(cplx) : nullptr, ra,
                  &out.evd_info[2]);
    XMCA_HIP(hipStreamSynchronize(st));
    tm.end();
    out.sigma.resize(rank);
    for (int i = 0; i < rank; ++i) out.sigma[i] = std::sqrt(std::max(lam[i], 0.0));
    out.n_vec = m;
    out.ldv[0] = A.N;
    out.ldv[1] = B.N;
    if (m == 0) return true;
    tm.begin("backproject");
    // Th_a[m][t] = conj((M^H p_m)[t]) = ((Ph diag(ms)) Mf)[m][t] ;  Th_l = Th_a G_b  (G_b Hermitian)
    CPlanes Ws, Tha, Thl;
    Ws.ensure((size_t)m * ra, cplx);
    Tha.ensure((size_t)m * T, cplx);
    Thl.ensure((size_t)m * T, cplx);
    XMCA_HIP(hipMemcpyAsync(Ws.r(), Ph.r(), sizeof(double) * (size_t)m * ra, hipMemcpyDeviceToDevice, st));
    if (cplx) XMCA_HIP(hipMemcpyAsync(Ws.im.get(), Ph.im.get(), sizeof(double) * (size_t)m * ra, hipMemcpyDeviceToDevice, st));
    if (ms)
      hipLaunchKernelGGL(scale_kernel, ew_grid((int64_t)m * ra), dim3(EW_BLOCK), 0, st, Ws.r(), Ws.i(cplx), (int64_t)ra, m, ra, ms, 0, 0);
    cgemm<double>(st, gws, Ws.r(), Ws.i(cplx), ra, true, false, Mf.r(), Mf.i(cplx), T, true, false, Tha.r(), Tha.i(cplx), T, m, T, ra,
                  1.0, nullptr, nullptr, false);
    cgemm<double>(st, gws, Tha.r(), Tha.i(cplx), T, true, false, Gb.r(), Gb.i(cplx), T, true, false, Thl.r(), Thl.i(cplx), T, m, T, T, 1.0,
                  nullptr, nullptr, false);
    XMCA_HIP(hipStreamSynchronize(st));
    tm.end();
    // weak block of H from the rows of Tha / Thl and projection of the weak rows against the strong ones, as in the analytic
    // route (time-space coordinates: metrics G_a, G_b).  In this route it is trusted down to 1e-5 sigma_1 (on the 10-decade
    // probe its absolute error is ~7e-12 sigma_1); what lies below is left to refine_by_deflation.
    refine_weak_block(Tha, Thl, m, T, dof, out, cplx, &Ga, &Gb);
    out.consistent = 0;
    if (m > 1) {
      // C = Thl G_a Thl^H (two products, 10 ms at T = 5000).  Off the diagonal: the Gram matrix of the left vectors (guard of
      // the Cholesky factor).  On the diagonal: with g = G_b q, g^H G_a g = mu g^H q = mu^2 for an eigenpair of G_a G_b, a
      // Rayleigh quotient of the pencil (G_b G_a G_b, G_b) that uses G_a itself and not its factor - second order in the
      // vector's error, so C_ii / mu_i^2 - 1 shows what the factor's absolute error eps lambda_a1 did to sigma_i^2.  The
      // leading modes that agree to 1e-6 need no second solve (refine_by_deflation): for fields whose weak modes are noise
      // against noise that is all of them (real C3: down to 1e-8 sigma_1), for variance spectra graded over 10 decades it
      // stops where the weak block stops being good.
      tm.begin("orthogonality_check");
      int n_check = 0;                                     // null modes carry arbitrary vectors
      while (n_check < m && out.sigma[n_check] > 1e-9 * out.sigma[0]) ++n_check;
      std::vector<double> cdiag, crow;
      const double worst = coherence(Thl, Ga, m, T, n_check, cplx, &cdiag, &crow);
      tm.end();
      int ok = 0;
      double worst_c = 0.0;
      for (; ok < m; ++ok) {
        const double mu = out.sigma[ok] * out.sigma[ok] * dof * dof;
        if (!(out.sigma[ok] > 1e-11 * out.sigma[0])) break;
        const double dev = std::fabs(cdiag[ok] / (mu * mu) - 1.0);
        if (!(dev < 1e-6) || !(crow[ok] < 1e-6)) break;       // ... and its left vector must be orthogonal to the stronger ones
        worst_c = std::max(worst_c, dev);
      }
      out.consistent = ok;
      static const bool trace = xmca_trace("solve");
      if (trace)
        std::fprintf(stderr, "[xmca solve] one-sided (time space): left-vector coherence %.3e over %d modes; sigma consistent (%.1e) for the "
                             "leading %d of %d modes, sigma there %.2e sigma_1\n", worst, n_check, worst_c, ok, m,
                     ok > 0 ? out.sigma[ok - 1] / out.sigma[0] : 1.0);
      // (guard of the Cholesky factor: over the modes that will be kept from this solve - what lies behind them is solved again)
      double worst_kept = 0.0;
      for (int i = 0; i < ok; ++i) worst_kept = std::max(worst_kept, crow[i]);
      if (guard && !((ok == n_check ? worst : worst_kept) < 1e-6)) return false;
      if (guard && ok < std::min(n_check, 2)) return false;
    }
    tm.begin("backproject");
    back_project(B, cplx, Tha.r(), Tha.i(cplx), m, out.Vt[1]);
    back_project(A, cplx, Thl.r(), Thl.i(cplx), m, out.Vt[0]);
    tm.end();
    return true;
  }

  // largest |C_ij| / sqrt(C_ii C_jj) over the leading n_check rows of C = E G E^H (E: nv x n rows, G: n x n metric)
  // diag != null: the diagonal of C (nv values) on the host
  // row_worst != null: per row, the worst coherence with any EARLIER row
  double coherence(const CPlanes& E, const CPlanes& G, int nv, int n, int n_check, bool cplx, std::vector<double>* diag = nullptr,
                   std::vector<double>* row_worst = nullptr) {
    CPlanes T1, C;
    DevBuf<double> coh, rw;
    if (row_worst) XMCA_HIP(hipMemsetAsync(rw.ensure((size_t)nv), 0, sizeof(double) * nv, st));
    T1.ensure((size_t)nv * n, cplx);
    C.ensure((size_t)nv * nv, cplx);
    cgemm<double>(st, gws, E.r(), E.i(cplx), n, true, false, G.r(), G.i(cplx), n, true, false, T1.r(), T1.i(cplx), n, nv, n, n, 1.0, nullptr,
                  nullptr, false);
    cgemm<double>(st, gws, T1.r(), T1.i(cplx), n, true, false, E.r(), E.i(cplx), n, false, true, C.r(), C.i(cplx), nv, nv, nv, n, 1.0,
                  nullptr, nullptr, false);
    XMCA_HIP(hipMemsetAsync(coh.ensure(1), 0, sizeof(double), st));
    hipLaunchKernelGGL(coherence_kernel, dim3(std::min(nv, 1024)), dim3(256), 0, st, C.r(), C.i(cplx), nv, n_check,
                       reinterpret_cast<unsigned long long*>(coh.get()), row_worst ? rw.get() : nullptr);
    double worst = 0.0;
    XMCA_HIP(hipMemcpyAsync(&worst, coh.get(), sizeof(double), hipMemcpyDeviceToHost, st));
    if (row_worst) {
      row_worst->resize((size_t)nv);
      XMCA_HIP(hipMemcpyAsync(row_worst->data(), rw.get(), sizeof(double) * nv, hipMemcpyDeviceToHost, st));
    }
    if (diag) {
      diag->resize((size_t)nv);
      XMCA_HIP(hipMemcpy2DAsync(diag->data(), sizeof(double), C.r(), sizeof(double) * (nv + 1), sizeof(double), (size_t)nv,
                                hipMemcpyDeviceToHost, st));
    }
    XMCA_HIP(hipStreamSynchronize(st));
    return worst;
  }

  // true when every row of the Hermitian G sums to (numerically) zero: G 1 = 0, the Gram matrix of column-centered fields
  bool rows_sum_to_zero(const CPlanes& G, int n, bool cplx) {
    DevBuf<double> y;
    std::vector<double> hy((size_t)n);
    y.ensure((size_t)n);
    double worst = 0.0, maxdiag = 0.0;
    for (int pl = 0; pl < (cplx ? 2 : 1); ++pl) {
      hipLaunchKernelGGL(row_sum_kernel, dim3(n), dim3(256), 0, st, pl ? G.im.get() : G.r(), n, y.get());
      XMCA_HIP(hipMemcpyAsync(hy.data(), y.get(), sizeof(double) * n, hipMemcpyDeviceToHost, st));
      XMCA_HIP(hipStreamSynchronize(st));
      for (int i = 0; i < n; ++i) worst = std::max(worst, std::fabs(hy[i]));
    }
    std::vector<double> d((size_t)n);
    XMCA_HIP(hipMemcpy2DAsync(d.data(), sizeof(double), G.r(), sizeof(double) * (n + 1), sizeof(double), (size_t)n, hipMemcpyDeviceToHost, st));
    XMCA_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < n; ++i) maxdiag = std::max(maxdiag, d[i]);
    return maxdiag > 0.0 && worst < 1e-9 * maxdiag * std::sqrt((double)n);
  }

  // -------------------------------------------------------------------------------------------------------------
  // Analytic-signal models (solve(complexify=True) without extension) whose fields are all wider than T.
  // hilbert(x) = Phi D Phi^H x with Phi the m = T/2+1 retained Fourier vectors, so X~ = Phi D Phi^H X lives in an
  // m-dimensional subspace: G~ = X~ X~^H = Phi Gy Phi^H with Gy = D Phi^H (X X^T) Phi D (m x m).  The device therefore
  //  * never forms the imaginary plane of the field (one REAL Gram GEMM, real-field back-projections),
  //  * diagonalises m x m instead of T x T matrices (8x fewer flops per decomposition),
  //  * reports the remaining rank - m modes as exact zeros (the reference gets ~1e-16 noise with arbitrary vectors).
  // -------------------------------------------------------------------------------------------------------------
  struct Analytic {
    int m = 0;
    CPlanes Phi;            // T x m
    DevBuf<double> h;       // m Hilbert weights
  };
  struct AReduced {
    CPlanes Wh;             // m x m, row i = conj(w_i): eigenvectors of Gy
    DevBuf<double> s;       // m
    std::vector<double> lam;
  };

  void analytic_basis(int T, Analytic& an) {
    an.m = (T % 2 == 0) ? T / 2 + 1 : (T + 1) / 2;
    an.Phi.ensure((size_t)T * an.m, true);
    an.h.ensure((size_t)an.m);
    hipLaunchKernelGGL(fourier_basis_kernel, ew_grid((int64_t)T * an.m), dim3(EW_BLOCK), 0, st, T, an.m, an.Phi.r(), an.Phi.im.get(),
                       an.h.get());
    XMCA_HIP(hipGetLastError());
  }

  // Gy = D Phi^H (X X^T) Phi D   (m x m Hermitian, both triangles)
  void analytic_gram(const FieldData<TI>& f, const Analytic& an, CPlanes& Gy) {
    const int T = (int)f.T, m = an.m;
    DevBuf<double> G;
    CPlanes P1;
    G.ensure((size_t)T * T);
    P1.ensure((size_t)T * m, true);
    Gy.ensure((size_t)m * m, true);
    tm.begin("gram");
    {
      GemmOpts o;   // G = X X^T (real field)
      o.b_nfast = false; o.upper_only = true; o.mirror = 1;
      gemm<TI, double>(st, gws, f.r(), f.N, f.r(), f.N, G.get(), T, T, T, (int)f.N, o);
    }
    tm.end();
    tm.begin("fourier_reduce");
    // Gy[k][l] = h_k h_l / T  sum_s sum_t exp(-2 pi i k s / T) G[s][t] exp(+2 pi i l t / T): a 2-D DFT of G of which the
    // m x m corner is kept (fft.h: T transforms along the rows, m along the columns, each in the LDS of one workgroup) when T
    // factors into 2, 3, 5, 7 and fits; otherwise two products with the explicit Fourier vectors: P1 = G Phi, Gy = D (Phi^H P1) D
    constexpr bool fft_on = true;   // (the DFT-by-GEMM form was a run-time switch until round 4)
    FftPlan plan;
    if (fft_on && fft_plan(T, plan)) {
      fft_batch(st, plan, T, G.get(), nullptr, T, 1, +1.0, P1.r(), P1.im.get(), m, 1, m, nullptr, nullptr, 1.0);
      fft_batch(st, plan, m, P1.r(), P1.im.get(), 1, m, -1.0, Gy.r(), Gy.im.get(), 1, m, m, an.h.get(), an.h.get(), 1.0 / (double)T);
      hipLaunchKernelGGL(hermitize_kernel, ew_grid((int64_t)m * m), dim3(EW_BLOCK), 0, st, Gy.r(), Gy.im.get(), m);
      XMCA_HIP(hipGetLastError());
    } else {
      cgemm<double>(st, gws, G.get(), nullptr, T, true, false, an.Phi.r(), an.Phi.im.get(), m, true, false, P1.r(), P1.im.get(), m, T, m, T,
                    1.0, nullptr, nullptr, false);
      cgemm<double>(st, gws, an.Phi.r(), an.Phi.im.get(), m, false, true, P1.r(), P1.im.get(), m, true, false, Gy.r(), Gy.im.get(), m, m, m,
                    T, 1.0, an.h.get(), an.h.get(), true);
    }
    tm.end();
    XMCA_HIP(hipStreamSynchronize(st));     // G, P1 are released on return
  }

  void reduce_analytic(const FieldData<TI>& f, const Analytic& an, AReduced& R, EvdInfo* info, bool want_vectors) {
    CPlanes Gy;
    analytic_gram(f, an, Gy);
    reduce_analytic_gram(Gy, an, R, info, want_vectors);
  }
  void reduce_analytic_gram(const CPlanes& Gy, const Analytic& an, AReduced& R, EvdInfo* info, bool want_vectors) {
    const int m = an.m;
    tm.begin("eigh");
    if (want_vectors) R.Wh.ensure((size_t)m * m, true);
    R.s.ensure((size_t)m);
    DevBuf<double> lam_dev;
    lam_dev.ensure((size_t)m);
    hermitian_evd(st, ews, Gy.r(), Gy.im.get(), m, m, R.lam, lam_dev.get(), want_vectors ? R.Wh.r() : nullptr,
                  want_vectors ? R.Wh.im.get() : nullptr, m, info);
    hipLaunchKernelGGL(sqrt_clamp_kernel, dim3(ceil_div(m, 256)), dim3(256), 0, st, lam_dev.get(), R.s.get(), m, 1.0);
    XMCA_HIP(hipStreamSynchronize(st));
    tm.end();
  }

  // Vt[i][:] = normalised  sum_t b_i[t] X[t][:]  with  b_i = Phi D conj(E[i][:])  for i < nv; rows nv..rows_total-1 are zero
  void analytic_project(const FieldData<TI>& f, const Analytic& an, const double* Er, const double* Ei, int nv, int rows_total,
                        CPlanes& Vt) {
    const int T = (int)f.T, m = an.m;
    Vt.ensure((size_t)rows_total * f.N, true);
    XMCA_HIP(hipMemsetAsync(Vt.r(), 0, sizeof(double) * (size_t)rows_total * f.N, st));
    XMCA_HIP(hipMemsetAsync(Vt.im.get(), 0, sizeof(double) * (size_t)rows_total * f.N, st));
    if (nv <= 0) return;
    CPlanes Es, Bt;
    Bt.ensure((size_t)nv * T, true);
    // Bt[i][t] = sum_k h_k conj(E[i][k]) exp(2 pi i k t / T) / sqrt(T)      (nv x T): nv zero-padded DFTs of length T, or
    // Bt = conj(E D) Phi^T as a product with the explicit Fourier vectors
    constexpr bool fft_on = true;   // (the DFT-by-GEMM form was a run-time switch until round 4)
    FftPlan plan;
    if (fft_on && fft_plan(T, plan)) {
      fft_batch(st, plan, nv, Er, Ei, m, 1, +1.0, Bt.r(), Bt.im.get(), T, 1, T, nullptr, nullptr, 1.0 / std::sqrt((double)T), m, true,
                an.h.get());
    } else {
      Es.ensure((size_t)nv * m, true);
      XMCA_HIP(hipMemcpyAsync(Es.r(), Er, sizeof(double) * (size_t)nv * m, hipMemcpyDeviceToDevice, st));
      XMCA_HIP(hipMemcpyAsync(Es.im.get(), Ei, sizeof(double) * (size_t)nv * m, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(scale_kernel, ew_grid((int64_t)nv * m), dim3(EW_BLOCK), 0, st, Es.r(), Es.im.get(), (int64_t)m, nv, m, an.h.get(),
                         0, 0);
      cgemm<double>(st, gws, Es.r(), Es.im.get(), m, true, true, an.Phi.r(), an.Phi.im.get(), m, false, false, Bt.r(), Bt.im.get(), T, nv, T,
                    m, 1.0, nullptr, nullptr, false);
    }
    Narrow<TI> bt;
    bt.from(st, Bt.r(), Bt.im.get(), (int64_t)nv * T);
    // Vt = Bt X  (complex x real field)
    cgemm<TI>(st, gws, bt.r, bt.i, T, true, false, f.r(), nullptr, f.N, true, false, Vt.r(), Vt.im.get(), f.N, nv, (int)f.N, T, 1.0,
              nullptr, nullptr, false);
    normalize_rows<double>(st, Vt.r(), Vt.im.get(), f.N, nv, (int)f.N, 0, nullptr);
    XMCA_HIP(hipStreamSynchronize(st));
  }

  // Weak modes of the analytic two-field model (the counterpart of refine_by_deflation in the Fourier subspace).
  // H = M Gy_b M^H / dof^2 is formed with an absolute error of ~50 eps lambda_1: its eigenvalues below ~1e-3 sigma_1 lose
  // digits like (sigma_1 / sigma_m)^2 (measured at BASELINE configs[2], sigma_1 / sigma_2500 = 9e4: 49 of the 2500
  // singular values beyond 1e-5 of the reference's, median 7e-7 over the noise floor; near-degenerate weak vectors mixed
  // at 1e-2) while its eigenVECTORS still split the strong from the weak part cleanly.  The rows of Er (= p_i^H M) and
  // El (= Er Gy_b) below the 1e-3 line give the weak block of H without passing through lambda_1:
  //     H_w[i][j] = p_i^H H p_j = El_i Er_j^H / dof^2        (norm sigma_{ns+1}^2, error eps sigma_1 sigma_w)
  // and its eigen-decomposition H_w = Z^H L Z - a nearly diagonal matrix, two or three Jacobi sweeps - replaces sigma and
  // rotates the weak rows: Er_w <- Z Er_w, El_w <- Z El_w.  Up to three levels, as in refine_by_deflation; same switch.
  // Rows W = E[ns..nv) made orthogonal to the rows S = E[s0..ns) under the metric G (m x m Hermitian):
  //   W_w -= sum_s  (S_s G W_w^H)^* / (S_s G S_s^H)  S_s.
  // E_l rows are the left singular vectors' coefficients (u_i ~ X~a^H E_l,i^H, metric Gy_a), E_r rows the right ones' (metric
  // Gy_b).  A weak row is formed by products that cancel from lambda_1 down to its own size, which leaves it a component
  // along the strong rows of ~eps lambda_1 / lambda_w - times s_a1 / s_a,w once it is back in grid space: measured at
  // full-size C3 |u_18^H u_2499| = 7e-4 (right side 4e-6) with every weak-weak product below 8e-8.  The exact vectors are
  // orthogonal, so the component is error and is projected out (the strong rows are mutually orthogonal to 1e-11).
  void project_out_rows(CPlanes& E, const CPlanes& G, int s0, int ns, int nv, int m, bool cplx) {
    const int k = ns - s0, nw = nv - ns;
    if (k <= 0 || nw <= 0) return;
    CPlanes T1, C;
    DevBuf<double> dinv;
    T1.ensure((size_t)k * m, cplx);
    C.ensure((size_t)k * nw, cplx);
    dinv.ensure((size_t)k);
    double* sr = E.r() + (int64_t)s0 * m;
    double* si = cplx ? E.im.get() + (int64_t)s0 * m : nullptr;
    double* wr = E.r() + (int64_t)ns * m;
    double* wi = cplx ? E.im.get() + (int64_t)ns * m : nullptr;
    cgemm<double>(st, gws, sr, si, m, true, false, G.r(), G.i(cplx), m, true, false, T1.r(), T1.i(cplx), m, k, m, m, 1.0, nullptr, nullptr,
                  false);
    hipLaunchKernelGGL(row_dot_inverse_kernel, dim3(k), dim3(256), 0, st, T1.r(), T1.i(cplx), sr, si, m, dinv.get());
    // C[s][w] = (S_s G W_w^H) / d_s ;  W -= C^H S
    cgemm<double>(st, gws, T1.r(), T1.i(cplx), m, true, false, wr, wi, m, false, true, C.r(), C.i(cplx), nw, k, nw, m, 1.0, dinv.get(),
                  nullptr, false);
    cgemm<double>(st, gws, C.r(), C.i(cplx), nw, false, true, sr, si, m, true, false, wr, wi, m, nw, m, k, -1.0, nullptr, nullptr, false,
                  1.0);
    XMCA_HIP(hipGetLastError());
    XMCA_HIP(hipStreamSynchronize(st));       // temporaries
  }

  // Ga / Gb: metrics of the left / right coefficient rows (Gy_a, Gy_b) for the projection above, or null
  void refine_weak_block(CPlanes& Er, CPlanes& El, int nv, int m, double dof, SolveResult& out, bool cplx = true,
                         const CPlanes* Ga = nullptr, const CPlanes* Gb = nullptr) {
    out.weak_refined = false;
    if constexpr (std::is_same<TI, float>::value) return;        // float32 fields: sigma is resolved to 6e-8 sigma_1 at best
    static const double thr = [] { const char* e = std::getenv("XMCA_DEFLATE_BELOW"); return e ? std::atof(e) : 1e-3; }();   // 0: off
    if (thr <= 0.0 || nv <= 1 || !(out.sigma[0] > 0.0)) return;
    int done = 0;
    for (int level = 0; level < 3; ++level) {
      const double top = out.sigma[done];
      if (!(top > 0.0)) break;
      int ns = done;
      while (ns < nv && out.sigma[ns] >= thr * top) ++ns;
      if (ns >= nv || ns == done) break;
      if (!(out.sigma[ns] > 1e-13 * out.sigma[0])) break;              // what is left is null
      const int nw = nv - ns;
      tm.begin("refine_weak");
      CPlanes Hw, Z, Tmp;
      Hw.ensure((size_t)nw * nw, cplx);
      Z.ensure((size_t)nw * nw, cplx);
      Tmp.ensure((size_t)nw * m, cplx);
      double* er_r = Er.r() + (int64_t)ns * m;
      double* er_i = cplx ? Er.im.get() + (int64_t)ns * m : nullptr;
      double* el_r = El.r() + (int64_t)ns * m;
      double* el_i = cplx ? El.im.get() + (int64_t)ns * m : nullptr;
      cgemm<double>(st, gws, el_r, el_i, m, true, false, er_r, er_i, m, false, true, Hw.r(), Hw.i(cplx), nw, nw, nw, m,
                    1.0 / (dof * dof), nullptr, nullptr, true);
      std::vector<double> lam;
      hermitian_evd(st, ews, Hw.r(), Hw.i(cplx), nw, nw, lam, nullptr, Z.r(), Z.i(cplx), nw, &out.evd_info[1], 0, true);   // nearly diagonal: ~3 sweeps
      for (double* base : {er_r, el_r}) {
        double* im = base == er_r ? er_i : el_i;
        cgemm<double>(st, gws, Z.r(), Z.i(cplx), nw, true, false, base, im, m, true, false, Tmp.r(), Tmp.i(cplx), m, nw, m, nw, 1.0,
                      nullptr, nullptr, false);
        XMCA_HIP(hipMemcpyAsync(base, Tmp.r(), sizeof(double) * (size_t)nw * m, hipMemcpyDeviceToDevice, st));
        if (cplx) XMCA_HIP(hipMemcpyAsync(im, Tmp.im.get(), sizeof(double) * (size_t)nw * m, hipMemcpyDeviceToDevice, st));
      }
      XMCA_HIP(hipStreamSynchronize(st));
      if (Ga && Gb) {
        project_out_rows(Er, *Gb, done, ns, nv, m, cplx);
        project_out_rows(El, *Ga, done, ns, nv, m, cplx);
      }
      tm.end();
      for (int j = 0; j < nw; ++j) out.sigma[ns + j] = std::sqrt(std::max(lam[j], 0.0));
      done = ns;
    }
    out.weak_refined = true;
  }

  static bool analytic_applicable(const FieldData<TI>* fields, int n_fields) {
    for (int k = 0; k < n_fields; ++k)
      if (fields[k].N <= fields[k].T || fields[k].has_im) return false;
    return true;
  }

  // M with M^H M = G (n x n Hermitian, positive semi-definite) as the UNSHIFTED Cholesky factor - the cheap replacement of
  // the eigen-factor S W^H of the field that the one-sided routes decompose.  Any factor serves them: with H = M G_b M^H / dof^2
  // = P L P^H, q = M^H p solves G_a G_b q = sigma^2 dof^2 q, and the singular vectors are X~b^H q and X~a^H G_b q, whatever M is.
  // Both factors are exact for a matrix eps |G| away from G; what the eigen-factor has on top is the eigensolver's handling
  // of graded spectra (LR step), so a factor whose pivots span more than `1e8` (variances) is refused and the caller
  // decomposes the field as before.  No shift (a ridge delta moves sigma_i^2 by delta / lambda_a,i, relative): the one null
  // direction that is always there is handled exactly instead -
  //   analytic subspace: the zero-frequency row/column of a centered field (G_00 ~ 0): factor the block behind it;
  //   time space (null_is_mean): the constant vector of a centered field: G + mean(diag) 1 1^T / n has the same eigenpairs
  //   except for that one, and G_b 1 = 0 keeps it out of H.
  bool factor_by_cholesky(const CPlanes& G, int n, bool cplx, bool null_is_mean, CPlanes& M) {
    static const bool on = [] { const char* e = std::getenv("XMCA_CHOLESKY_FACTOR"); return !(e && e[0] == '0'); }();
    if (!on || !cholesky_enabled() || n < 2) return false;
    tm.begin("cholesky");
    const size_t nn = (size_t)n * n;
    M.ensure(nn, cplx);
    XMCA_HIP(hipMemcpyAsync(M.r(), G.r(), sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
    if (cplx) XMCA_HIP(hipMemcpyAsync(M.im.get(), G.im.get(), sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
    DevBuf<unsigned long long> mm;
    unsigned long long init[2] = {0ull, 0x7ff0000000000000ull};
    double g00 = 0.0;
    XMCA_HIP(hipMemcpyAsync(mm.ensure(2), init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(chol_minmax_diag_kernel, dim3(std::min(ceil_div(n, 256), 64)), dim3(256), 0, st, M.r(), (int64_t)n, n, mm.get());
    unsigned long long bits[2];
    XMCA_HIP(hipMemcpyAsync(bits, mm.get(), sizeof(bits), hipMemcpyDeviceToHost, st));
    XMCA_HIP(hipMemcpyAsync(&g00, M.r(), sizeof(double), hipMemcpyDeviceToHost, st));
    XMCA_HIP(hipStreamSynchronize(st));
    double maxdiag = 0.0;
    std::memcpy(&maxdiag, &bits[0], sizeof(double));
    bool ok = maxdiag > 0.0 && std::isfinite(maxdiag);
    int k0 = 0;
    if (ok) {
      if (null_is_mean) {
        hipLaunchKernelGGL(add_const_kernel, ew_grid((int64_t)nn), dim3(EW_BLOCK), 0, st, M.r(), (int64_t)nn, maxdiag / (double)n);
      } else if (std::fabs(g00) < 1e-12 * maxdiag) {
        k0 = 1;
      }
      const int64_t off = (int64_t)k0 * (n + 1);
      ok = cholesky_upper(st, gws, M.r() + off, cplx ? M.im.get() + off : nullptr, n - k0, n, 0.0);
    }
    if (ok) {
      if (k0) {   // row 0 and column 0 of the factor are zero: the null mode stays a null mode
        XMCA_HIP(hipMemsetAsync(M.r(), 0, sizeof(double) * n, st));
        XMCA_HIP(hipMemset2DAsync(M.r(), sizeof(double) * n, 0, sizeof(double), (size_t)n, st));
        if (cplx) {
          XMCA_HIP(hipMemsetAsync(M.im.get(), 0, sizeof(double) * n, st));
          XMCA_HIP(hipMemset2DAsync(M.im.get(), sizeof(double) * n, 0, sizeof(double), (size_t)n, st));
        }
      }
      XMCA_HIP(hipMemcpyAsync(mm.get(), init, sizeof(init), hipMemcpyHostToDevice, st));
      const int64_t off = (int64_t)k0 * (n + 1);
      hipLaunchKernelGGL(chol_minmax_diag_kernel, dim3(std::min(ceil_div(n - k0, 256), 64)), dim3(256), 0, st, M.r() + off, (int64_t)n, n - k0,
                         mm.get());
      XMCA_HIP(hipMemcpyAsync(bits, mm.get(), sizeof(bits), hipMemcpyDeviceToHost, st));
      XMCA_HIP(hipStreamSynchronize(st));
      double rmax = 0.0, rmin = 0.0;
      std::memcpy(&rmax, &bits[0], sizeof(double));
      std::memcpy(&rmin, &bits[1], sizeof(double));
      ok = rmax > 0.0 && rmin > 1e-4 * rmax;          // pivots r_ii^2 within 1e8 of each other
    }
    tm.end();
    return ok;
  }

  void solve_analytic(const FieldData<TI>* fields, int n_fields, int n_vec_req, SolveResult& out) {
    out.vt_f32[0] = out.vt_f32[1] = false;     // (complex vectors: float64 planes)
    out.Vt32[0].release();                     // (a float32 plane left by an earlier model on this handle)
    out.Vt32[1].release();
    const FieldData<TI>& A = fields[0];
    const int T = (int)A.T;
    const double dof = (double)(T - 1);
    Analytic an;
    analytic_basis(T, an);
    const int m = an.m;
    out.cplx = true;
    out.weak_refined = false;
    for (EvdInfo& e : out.evd_info) e = EvdInfo();
    out.rank = T;                                   // min(T, N) as the reference reports it (array.py:597)
    const int n_vec = n_vec_req < 0 ? T : std::min(n_vec_req, T);
    const int nv = std::min(n_vec, m);              // modes that can be non-null
    out.n_vec = n_vec;
    out.sigma.assign(T, 0.0);
    AReduced Ra, Rb;
    if (n_fields == 2 && n_vec == 0 && cholesky_enabled()) {
      CPlanes Gya, Gyb;
      analytic_gram(A, an, Gya);
      analytic_gram(fields[1], an, Gyb);
      if (values_by_cholesky(Gya, Gyb, m, true, dof, out)) {      // sigma[m..T) stay exact zeros
        out.ldv[0] = A.N;
        out.ldv[1] = fields[1].N;
        return;
      }
    }
    static const bool one_sided_on = [] { const char* e = std::getenv("XMCA_ONE_SIDED"); return !(e && e[0] == '0'); }();
    CPlanes Fm, Gya;                   // factor of Gy_a used by the one-sided route: Cholesky (Fm) or eigen (Ra.s, Ra.Wh)
    bool by_chol = false;
    if (n_fields == 2 && one_sided_on) {
      analytic_gram(A, an, Gya);
      by_chol = factor_by_cholesky(Gya, m, true, false, Fm);
      if (!by_chol) reduce_analytic_gram(Gya, an, Ra, &out.evd_info[0], true);
    } else {
      reduce_analytic(A, an, Ra, &out.evd_info[0], n_fields == 2 || n_vec != 0);
    }
    if (n_fields == 1) {
      for (int i = 0; i < m; ++i) out.sigma[i] = std::max(Ra.lam[i], 0.0) / dof;
      out.ldv[0] = A.N;
      tm.begin("backproject");
      if (n_vec > 0) analytic_project(A, an, Ra.Wh.r(), Ra.Wh.im.get(), nv, n_vec, out.Vt[0]);
      tm.end();
      return;
    }
    const FieldData<TI>& B = fields[1];
    if (one_sided_on) {
      // second field as an operator (see solve_one_sided): H = M Gy_b M^H / dof^2 with M^H M = Gy_a - the Cholesky factor
      // (factor_by_cholesky: no eigen-decomposition of the first field at all) or the eigen-factor S_a Wh_a
      CPlanes Gyb;
      analytic_gram(B, an, Gyb);
      out.ldv[0] = A.N;
      out.ldv[1] = B.N;
      // returns false when the Cholesky factor turned out not to be good enough (left vectors not orthogonal: see below)
      auto with_factor = [&](const CPlanes& Mf, const double* ms, bool guard) -> bool {
        CPlanes M1, H, Ph;
        tm.begin("kernel");
        M1.ensure((size_t)m * m, true);
        H.ensure((size_t)m * m, true);
        cgemm<double>(st, gws, Mf.r(), Mf.im.get(), m, true, false, Gyb.r(), Gyb.im.get(), m, true, false, M1.r(), M1.im.get(), m, m, m,
                      m, 1.0, ms, nullptr, false);
        cgemm<double>(st, gws, M1.r(), M1.im.get(), m, true, false, Mf.r(), Mf.im.get(), m, false, true, H.r(), H.im.get(), m, m, m, m,
                      1.0 / (dof * dof), nullptr, ms, true);
        tm.end();
        std::vector<double> lam;
        tm.begin("kernel_svd");
        if (n_vec > 0) Ph.ensure((size_t)m * m, true);
        hermitian_evd(st, ews, H.r(), H.im.get(), m, m, lam, nullptr, n_vec > 0 ? Ph.r() : nullptr, n_vec > 0 ? Ph.im.get() : nullptr, m,
                      &out.evd_info[2]);
        XMCA_HIP(hipStreamSynchronize(st));
        tm.end();
        for (int i = 0; i < m; ++i) out.sigma[i] = std::sqrt(std::max(lam[i], 0.0));
        if (n_vec == 0) return true;
        tm.begin("backproject");
        // E_right = Ph M (rows = conj of the subspace coefficients of q_m = M^H p_m),  E_left = E_right Gy_b
        CPlanes Ws, Er, El;
        Ws.ensure((size_t)nv * m, true);
        Er.ensure((size_t)nv * m, true);
        El.ensure((size_t)nv * m, true);
        XMCA_HIP(hipMemcpyAsync(Ws.r(), Ph.r(), sizeof(double) * (size_t)nv * m, hipMemcpyDeviceToDevice, st));
        XMCA_HIP(hipMemcpyAsync(Ws.im.get(), Ph.im.get(), sizeof(double) * (size_t)nv * m, hipMemcpyDeviceToDevice, st));
        if (ms)
          hipLaunchKernelGGL(scale_kernel, ew_grid((int64_t)nv * m), dim3(EW_BLOCK), 0, st, Ws.r(), Ws.im.get(), (int64_t)m, nv, m, ms, 0, 0);
        cgemm<double>(st, gws, Ws.r(), Ws.im.get(), m, true, false, Mf.r(), Mf.im.get(), m, true, false, Er.r(), Er.im.get(), m, nv, m,
                      m, 1.0, nullptr, nullptr, false);
        cgemm<double>(st, gws, Er.r(), Er.im.get(), m, true, false, Gyb.r(), Gyb.im.get(), m, true, false, El.r(), El.im.get(), m, nv, m, m,
                      1.0, nullptr, nullptr, false);
        tm.end();
        refine_weak_block(Er, El, nv, m, dof, out, true, &Gya, &Gyb);
        if (guard && nv > 1) {
          // The left vectors are u_i ~ X~a^H g_i with g_i = E_left,i^H: their Gram matrix is E_l Gy_a E_l^H (nv x nv, two small
          // products).  On spectra graded over many decades E_l = E_r Gy_b cancels down to the weak rows and the Cholesky
          // factor - unlike the eigen-factor, whose weak rows are small numbers to begin with - leaves u_weak with a component
          // along the strong modes that project_out_rows has to remove (2e-8 left on the 10-decade probe, 4e-8 at C3 - the leading
          // modes among themselves, the same with either factor).  Above 1e-6 the factor is not trusted: decompose the field.
          tm.begin("orthogonality_check");
          int n_check = 0;                                     // null modes carry arbitrary vectors
          while (n_check < nv && out.sigma[n_check] > 1e-9 * out.sigma[0]) ++n_check;
          const double worst = coherence(El, Gya, nv, m, n_check, true);
          tm.end();
          static const bool trace = xmca_trace("solve");
          if (trace) std::fprintf(stderr, "[xmca solve] Cholesky factor: left-vector coherence %.3e over %d modes\n", worst, n_check);
          if (!(worst < 1e-6)) return false;
        }
        tm.begin("backproject");
        analytic_project(B, an, Er.r(), Er.im.get(), nv, n_vec, out.Vt[1]);
        analytic_project(A, an, El.r(), El.im.get(), nv, n_vec, out.Vt[0]);
        tm.end();
        return true;
      };
      if (by_chol && with_factor(Fm, nullptr, true)) return;
      if (by_chol) reduce_analytic_gram(Gya, an, Ra, &out.evd_info[0], true);
      with_factor(Ra.Wh, Ra.s.get(), false);
      return;
    }
    reduce_analytic(B, an, Rb, &out.evd_info[1], true);
    // K = S_a Wh_a Wh_b^H S_b / dof   (m x m)
    CPlanes K, H, Ph, Qh;
    K.ensure((size_t)m * m, true);
    tm.begin("kernel");
    cgemm<double>(st, gws, Ra.Wh.r(), Ra.Wh.im.get(), m, true, false, Rb.Wh.r(), Rb.Wh.im.get(), m, false, true, K.r(), K.im.get(), m, m, m,
                  m, 1.0 / dof, Ra.s.get(), Rb.s.get(), false);
    tm.end();
    tm.begin("kernel_svd");
    H.ensure((size_t)m * m, true);
    std::vector<double> lam;
    cgemm<double>(st, gws, K.r(), K.im.get(), m, false, true, K.r(), K.im.get(), m, true, false, H.r(), H.im.get(), m, m, m, m, 1.0,
                  nullptr, nullptr, true);
    if (n_vec > 0) {
      Qh.ensure((size_t)m * m, true);
      Ph.ensure((size_t)m * m, true);
      hermitian_evd(st, ews, H.r(), H.im.get(), m, m, lam, nullptr, Qh.r(), Qh.im.get(), m, &out.evd_info[2]);
      cgemm<double>(st, gws, Qh.r(), Qh.im.get(), m, true, false, K.r(), K.im.get(), m, false, true, Ph.r(), Ph.im.get(), m, m, m, m, 1.0,
                    nullptr, nullptr, false);
      hipLaunchKernelGGL((normalize_rows_kernel<double>), dim3(m), dim3(256), 0, st, Ph.r(), Ph.im.get(), (int64_t)m, m, 0,
                         (double*)nullptr);
      XMCA_HIP(hipGetLastError());
    } else {
      hermitian_evd(st, ews, H.r(), H.im.get(), m, m, lam, nullptr, nullptr, nullptr, m, &out.evd_info[2]);
    }
    XMCA_HIP(hipStreamSynchronize(st));
    tm.end();
    for (int i = 0; i < m; ++i) out.sigma[i] = std::sqrt(std::max(lam[i], 0.0));
    out.ldv[0] = A.N;
    out.ldv[1] = B.N;
    if (n_vec == 0) return;
    tm.begin("backproject");
    // E_left = (Qh diag(s_b)) Wh_b ,  E_right = (Ph diag(s_a)) Wh_a   (first nv rows)
    auto side = [&](const FieldData<TI>& self, const CPlanes& Wsmall, const AReduced& other, CPlanes& Vt) {
      CPlanes Ws, E;
      Ws.ensure((size_t)nv * m, true);
      E.ensure((size_t)nv * m, true);
      XMCA_HIP(hipMemcpyAsync(Ws.r(), Wsmall.r(), sizeof(double) * (size_t)nv * m, hipMemcpyDeviceToDevice, st));
      XMCA_HIP(hipMemcpyAsync(Ws.im.get(), Wsmall.im.get(), sizeof(double) * (size_t)nv * m, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(scale_kernel, ew_grid((int64_t)nv * m), dim3(EW_BLOCK), 0, st, Ws.r(), Ws.im.get(), (int64_t)m, nv, m,
                         other.s.get(), 0, 0);
      cgemm<double>(st, gws, Ws.r(), Ws.im.get(), m, true, false, other.Wh.r(), other.Wh.im.get(), m, true, false, E.r(), E.im.get(), m,
                    nv, m, m, 1.0, nullptr, nullptr, false);
      analytic_project(self, an, E.r(), E.im.get(), nv, n_vec, Vt);
    };
    side(A, Qh, Rb, out.Vt[0]);
    side(B, Ph, Ra, out.Vt[1]);
    tm.end();
  }

 private:
  void negate(double* p, int64_t n) {
    DevBuf<double> m1;
    double v = -1.0;
    XMCA_HIP(hipMemcpyAsync(m1.ensure(1), &v, sizeof(double), hipMemcpyHostToDevice, st));
    // rows = n, cols = 1 view with a column scale of length 1
    hipLaunchKernelGGL(scale_kernel, ew_grid(n), dim3(EW_BLOCK), 0, st, p, (double*)nullptr, (int64_t)1, (int)n, 1, m1.get(), 0, 0);
    XMCA_HIP(hipGetLastError());
    XMCA_HIP(hipStreamSynchronize(st));
  }

  // singular vectors of `self` in grid space.
  //   self unreduced : V = own small singular vectors  -> Vt = conj(Oh)
  //   self reduced   : V ~ X~self^H (F_other w_m)  with w the OTHER side's small singular vectors (Wh rows = conj(w_m))
  void project_side(const FieldData<TI>& self, const FieldData<TI>& other, bool cplx, const Reduced& Rs, const Reduced& Ro,
                    const CPlanes& Oh, const CPlanes& Wh, int r_self, int r_other, int m, CPlanes& Vt) {
    const int T = (int)self.T;
    if (!Rs.reduced) {
      Vt.ensure((size_t)m * r_self, cplx);
      XMCA_HIP(hipMemcpyAsync(Vt.r(), Oh.r(), sizeof(double) * (size_t)m * r_self, hipMemcpyDeviceToDevice, st));
      if (cplx) {
        XMCA_HIP(hipMemcpyAsync(Vt.im.get(), Oh.im.get(), sizeof(double) * (size_t)m * r_self, hipMemcpyDeviceToDevice, st));
        negate(Vt.im.get(), (int64_t)m * r_self);
      }
      XMCA_HIP(hipStreamSynchronize(st));
      return;
    }
    // Th[m][t] = sum_j Wh[m][j] conj(F_other[t][j])
    CPlanes Th;
    Th.ensure((size_t)m * T, cplx);
    if (Ro.reduced) {
      // conj(F_o[t][j]) = Z_o[j][t] s_j  ->  Th = (Wh diag(s_o)) Z_o
      CPlanes Ws;
      Ws.ensure((size_t)m * r_other, cplx);
      XMCA_HIP(hipMemcpyAsync(Ws.r(), Wh.r(), sizeof(double) * (size_t)m * r_other, hipMemcpyDeviceToDevice, st));
      if (cplx) XMCA_HIP(hipMemcpyAsync(Ws.im.get(), Wh.im.get(), sizeof(double) * (size_t)m * r_other, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(scale_kernel, ew_grid((int64_t)m * r_other), dim3(EW_BLOCK), 0, st, Ws.r(), Ws.i(cplx), (int64_t)r_other, m,
                         r_other, Ro.s.get(), 0, 0);
      cgemm<double>(st, gws, Ws.r(), Ws.i(cplx), r_other, true, false, Ro.Z.r(), Ro.Z.i(cplx), T, true, false, Th.r(), Th.i(cplx), T,
                    m, T, r_other, 1.0, nullptr, nullptr, false);
      XMCA_HIP(hipStreamSynchronize(st));
    } else {
      // conj(F_o[t][j]) = conj(X~o[t][j])
      Narrow<TI> w;
      w.from(st, Wh.r(), Wh.i(cplx), (int64_t)m * r_other);
      cgemm<TI>(st, gws, w.r, w.i, r_other, true, false, other.r(), other.i(), other.N, false, true, Th.r(), Th.i(cplx), T, m, T,
                r_other, 1.0, nullptr, nullptr, false);
      XMCA_HIP(hipStreamSynchronize(st));
    }
    back_project(self, cplx, Th.r(), Th.i(cplx), m, Vt);
  }
};

// ---------------------------------------------------------------------------------------------------
// host-side p x p complex algebra for the Promax tail (rotation.py:128-147)
// ---------------------------------------------------------------------------------------------------
using cd = std::complex<double>;
struct SmallMat {
  int n = 0;
  std::vector<cd> a;
  SmallMat() = default;
  explicit SmallMat(int n_) : n(n_), a((size_t)n_ * n_) {}
  cd& operator()(int i, int j) { return a[(size_t)i * n + j]; }
  const cd& operator()(int i, int j) const { return a[(size_t)i * n + j]; }
  static SmallMat eye(int n) { SmallMat m(n); for (int i = 0; i < n; ++i) m(i, i) = 1.0; return m; }
  SmallMat operator*(const SmallMat& o) const {
    SmallMat r(n);
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < n; ++k) {
        const cd v = (*this)(i, k);
        for (int j = 0; j < n; ++j) r(i, j) += v * o(k, j);
      }
    return r;
  }
  SmallMat H() const {
    SmallMat r(n);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) r(j, i) = std::conj((*this)(i, j));
    return r;
  }
  // Gauss-Jordan with partial pivoting; returns false when exactly singular
  bool inverse(SmallMat& out) const {
    SmallMat w = *this;
    out = eye(n);
    for (int c = 0; c < n; ++c) {
      int piv = c;
      double best = std::abs(w(c, c));
      for (int r = c + 1; r < n; ++r)
        if (std::abs(w(r, c)) > best) { best = std::abs(w(r, c)); piv = r; }
      if (!(best > 0.0)) return false;
      if (piv != c)
        for (int j = 0; j < n; ++j) { std::swap(w(c, j), w(piv, j)); std::swap(out(c, j), out(piv, j)); }
      const cd inv = 1.0 / w(c, c);
      for (int j = 0; j < n; ++j) { w(c, j) *= inv; out(c, j) *= inv; }
      for (int r = 0; r < n; ++r) {
        if (r == c) continue;
        const cd f = w(r, c);
        if (f == cd(0.0)) continue;
        for (int j = 0; j < n; ++j) { w(r, j) -= f * w(c, j); out(r, j) -= f * out(c, j); }
      }
    }
    return true;
  }
};

struct RotateResult {
  int p = 0;
  int iters = 0;
  bool converged = false;
  bool nan = false;
  bool cplx = false;
  std::vector<cd> R, Phi;              // p x p row-major
  std::vector<double> norm_left, norm_right;
  double last_d = 0.0;
};

// Device state of one rotation problem: normalised loadings A (p x N planes), h, R, ...
struct RotationDevice {
  CPlanes A, R, A0, acc, W;   // W: right singular vectors of the previous Varimax step (warm start)
  DevBuf<double> h, cvec, state, part_r, part_i, colmax;
  DevBuf<unsigned int> counter;     // arrival ticket of the fused Varimax iteration kernel
  DevBuf<unsigned int> pflags;      // persistent Varimax kernel: epoch flag per workgroup
  DevBuf<double> ppart_r, ppart_i;  // ... and its double-buffered partials
  int64_t N = 0, Nleft = 0;
  int p = 0;
  bool cplx = false;
  int nwg = 0;            // grid of the Varimax kernels (persistent: one workgroup per CU at most)
  int nacc = 0;           // grid of the one-pass accumulation kernels (Gram, column maxima, Promax fits): sized by N
};

class Rotator {
 public:
  hipStream_t st;
  StageTimer& tm;
  double gamma = 1.0;   // Varimax family parameter (rotation.py:15,56-57): 1 = Varimax (what MCA.rotate uses), 0 = Quartimax
  GemmWorkspace* gws = nullptr;   // only the GEMM-based path (more modes than the fused kernels hold, run_generic) needs these
  EvdWorkspace* ews = nullptr;
  Rotator(hipStream_t s, StageTimer& t) : st(s), tm(t) {}
  Rotator(hipStream_t s, StageTimer& t, GemmWorkspace& g, EvdWorkspace& e) : st(s), tm(t), gws(&g), ews(&e) {}
  static bool fused_fits(int p, bool cplx) { return p <= rot_max_modes(cplx); }

  // `wide`: many modes (p^2 x planes >= 512 doubles per partial) - the accumulation dominates an iteration and the partials are
  // summed in two stages whose cost does not grow with the grid (varimax_persistent_kernel): up to one workgroup per CU.
  // Few modes: the all-to-all reduction and the exchange dominate - flat between 40 and 128 workgroups, slower beyond.
  static bool wide_grid(int p, bool cplx) { return (size_t)p * p * (cplx ? 2 : 1) >= 512; }
  // Long grids (more than 1024 tiles of 64 points: the 0.25-degree global grid of C5 has 16 200): the passes are streaming
  // kernels - the persistent Varimax loop takes one workgroup on every CU (two-stage sum of the partials) and prefetches its
  // tiles, the one-pass kernels take 8 workgroups per CU (128 workgroups with one tile in flight each read
  // the 83 MB of planes at 0.4 TB/s).  Up to 1024 tiles everything is as it was (same grids, same bits).
  static bool long_grid(int64_t N) { return (N + ROT_PB - 1) / ROT_PB > 1024; }
  static int pick_nwg(int64_t N, bool wide = false, int cap_override = 0) {
    const int64_t nb = (N + ROT_PB - 1) / ROT_PB;
    const int cap = cap_override > 0 ? cap_override : ((wide || long_grid(N)) ? 256 : 128);
    // equal shares: with 157 tiles and a cap of 128 workgroups, 79 workgroups of 2 tiles beat 128 of 1-2
    const int64_t per = (nb + cap - 1) / cap;
    return (int)std::max<int64_t>(1, (nb + per - 1) / per);
  }

  template <bool CPLX, int MODE, int SEL>
  void accum(RotationDevice& d, double power, double* out_r, double* out_i) {
    const size_t smem = rot_accum_smem(d.p, CPLX);
    auto kern = rot_accum_kernel<CPLX, MODE, SEL>;
    XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int grid = MODE == 0 ? d.nwg : d.nacc;       // (MODE 0: the partials go to the step kernel, which adds up d.nwg of them)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, d.A.r(), d.A.i(CPLX), d.h.get(), d.N, d.Nleft, d.p, d.R.r(),
                       d.R.i(CPLX), d.cvec.get(), d.colmax.get(), power, d.state.get(), d.part_r.get(),
                       CPLX ? d.part_i.get() : nullptr, reinterpret_cast<unsigned long long*>(d.colmax.get()));
    XMCA_HIP(hipGetLastError());
    if (MODE != 0 && MODE != 3) {
      hipLaunchKernelGGL(rot_reduce_partials_kernel, dim3(d.p * d.p), dim3(256), 0, st, d.part_r.get(),
                         CPLX ? d.part_i.get() : nullptr, grid, d.p * d.p, out_r, CPLX ? out_i : nullptr);
      XMCA_HIP(hipGetLastError());
    }
  }

  void alloc(RotationDevice& d, int64_t N, int64_t Nleft, int p, bool cplx) {
    XMCA_CHECK(p >= 2, XMCA_ERR_INVALID, "rotate: n_rot must be > 1");
    XMCA_CHECK(fused_fits(p, cplx) || (gws && ews), XMCA_ERR_UNSUPPORTED,
               "rotate: n_rot = " + std::to_string(p) + " needs the GEMM-based path (no workspace given)");
    d.N = N; d.Nleft = Nleft; d.p = p; d.cplx = cplx;
    d.nwg = pick_nwg(N, wide_grid(p, cplx));
    d.nacc = long_grid(N) ? pick_nwg(N, false, 2048) : d.nwg;
    d.A.ensure((size_t)p * N, cplx);
    d.h.ensure((size_t)N);
    d.R.ensure((size_t)p * p, cplx);
    d.A0.ensure((size_t)p * p, cplx);
    d.W.ensure((size_t)p * p, cplx);
    d.acc.ensure((size_t)p * p, cplx);
    d.cvec.ensure((size_t)p);
    d.state.ensure(ROT_STATE_N);
    d.part_r.ensure((size_t)std::max(d.nwg, d.nacc) * p * p);
    if (cplx) d.part_i.ensure((size_t)std::max(d.nwg, d.nacc) * p * p);
    d.colmax.ensure((size_t)p);
    if (!d.counter.get()) { d.counter.ensure(1); XMCA_HIP(hipMemsetAsync(d.counter.get(), 0, sizeof(unsigned int), st)); }
  }

  // runs Varimax + Promax on d.A / d.h (already normalised).  B_out (nullable): N x p rotated loadings for the host.
  template <bool CPLX>
  void run(RotationDevice& d, int power, double tol, int max_iter, RotateResult& res, double* B_out_dev, bool varimax_only) {
    const int p = d.p;
    res.p = p; res.cplx = CPLX;
    if (!fused_fits(p, CPLX)) { run_generic<CPLX>(d, power, tol, max_iter, res, B_out_dev, varimax_only); return; }
    tm.begin("varimax");
    accum<CPLX, 1, 0>(d, 1.0, d.A0.r(), d.A0.i(CPLX));
    hipLaunchKernelGGL((varimax_step_kernel<CPLX>), dim3(1), dim3(256), 0, st, d.part_r.get(), CPLX ? d.part_i.get() : nullptr, d.nwg,
                       p, d.A0.r(), d.A0.i(CPLX), d.R.r(), d.R.i(CPLX), d.W.r(), d.W.i(CPLX), d.cvec.get(), d.state.get(), tol, 1);
    XMCA_HIP(hipGetLastError());
    double state[ROT_STATE_N] = {0};
    int launched = 0;
    // default: one launch per iteration (partial G + last-arriving workgroup finishes the step with a Newton-Schulz
    // polar factor); XMCA_VARIMAX_FUSED=0 selects the two-launch variant with the Jacobi SVD.
    static const bool fused_on = [] { const char* e = std::getenv("XMCA_VARIMAX_FUSED"); return !(e && e[0] == '0'); }();
    const bool fused = fused_on;
    const size_t fused_smem = std::max(rot_accum_smem(p, CPLX), rot_polar_smem(p, CPLX));
    if (fused)
      XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(varimax_iter_kernel<CPLX>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_smem));
    // XMCA_VARIMAX_PERSIST (default on): the whole loop in one launch, workgroups exchange their partial G through
    // epoch-tagged buffers.  Needs a resident grid (<= 1 workgroup per CU) and the LDS for both scratch areas.
    static const bool persist_on = [] { const char* e = std::getenv("XMCA_VARIMAX_PERSIST"); return !(e && e[0] == '0'); }();
    const size_t work_bytes = std::max(rot_accum_smem(p, CPLX), rot_polar_smem(p, CPLX));
    size_t persist_smem = rot_persistent_smem(p, CPLX);
    int tiles_per_wg = (int)(((d.N + ROT_PB - 1) / ROT_PB + d.nwg - 1) / d.nwg);
    bool resident = persist_smem + rot_resident_smem(p, CPLX, tiles_per_wg) <= 160 * 1024;
    if (resident) persist_smem += rot_resident_smem(p, CPLX, tiles_per_wg);
    // the epoch exchange spins on every workgroup of the grid: all of them must be co-resident, one per CU (a
    // partitioned / CU-masked device has fewer CUs than the 128-workgroup cap -> per-iteration launches instead)
    PersistGate& gate = persist_gate();
    const int n_cus = gate.n_cus;
    // ... and so must the persistent grids of the OTHER surrogate lanes of this process (rule_n / bootstrap keep several
    // replicates in flight), Varimax loops and tridiagonal reductions alike: the launch claims its CUs at the device's gate
    // (common.h PersistGate) and waits there for its turn - a reduction holds every CU for its ~20 ms, Varimax grids of
    // several lanes run side by side while they fit
    const bool persist_ok = fused && persist_on && persist_smem <= 160 * 1024 && d.nwg <= 256 && d.nwg <= n_cus && max_iter > 0;
    if (persist_ok) {
      PersistGate::Claim claim(gate, d.nwg);           // given back at the end of this block: the stream has been synchronised by then
      // XMCA_VARIMAX_TEST_GIVEUP=k (tests): the persistent launch stops after k iterations, as if a workgroup had gone
      // missing there, and the per-iteration launches take over - the hand-over must not change R or the stop iteration
      const char* tg = std::getenv("XMCA_VARIMAX_TEST_GIVEUP");
      const int persist_iters = (tg && std::atoi(tg) > 0) ? std::min(std::atoi(tg), max_iter) : max_iter;
      d.pflags.ensure((size_t)2 * d.nwg);                        // partial published / chunk of the sum published
      d.ppart_r.ensure((size_t)2 * d.nwg * p * p + 4 * (size_t)p * p);   // ... + the summed G (two parities, two planes)
      if (CPLX) d.ppart_i.ensure((size_t)2 * d.nwg * p * p);
      XMCA_HIP(hipMemsetAsync(d.pflags.get(), 0, sizeof(unsigned int) * 2 * d.nwg, st));
      // two-stage sum of the partials when every workgroup would otherwise read more than ~32k doubles (XMCA_ROT_TWO_STAGE=0 / 1 forces)
      const bool rot_two_stage = [&] { const char* e = std::getenv("XMCA_ROT_TWO_STAGE"); return e ? e[0] != '0' : ((size_t)d.nwg * p * p * (CPLX ? 2 : 1) > 32768 || (long_grid(d.N) && d.nwg > 128)); }();
      constexpr int rot_poll_delay = 0;     // (a delayed first poll, the lever of the tridiagonal reduction, has no measurable effect here)
      XMCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(varimax_persistent_kernel<CPLX>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)persist_smem));
      hipLaunchKernelGGL((varimax_persistent_kernel<CPLX>), dim3(d.nwg), dim3(256), persist_smem, st, d.A.r(), d.A.i(CPLX), d.h.get(),
                         d.N, p, d.A0.r(), d.A0.i(CPLX), d.R.r(), d.R.i(CPLX), d.cvec.get(), d.state.get(), d.ppart_r.get(),
                         CPLX ? d.ppart_i.get() : nullptr, d.pflags.get(), tol, persist_iters, work_bytes / sizeof(double),
                         resident ? tiles_per_wg : 0, gamma, rot_poll_delay,
                         rot_two_stage ? d.ppart_r.get() + (size_t)2 * d.nwg * p * p : nullptr);
      XMCA_HIP(hipGetLastError());
      XMCA_HIP(hipMemcpyAsync(state, d.state.get(), sizeof(state), hipMemcpyDeviceToHost, st));
      XMCA_HIP(hipStreamSynchronize(st));
      if (state[4] == 2.0 || (persist_iters < max_iter && state[1] == 0.0 && state[4] == 0.0)) {
        // a workgroup never arrived (grid not resident: another process holds CUs).  Workgroup 0 has written the R, c
        // and iteration state of the last completed iteration: clear the flag and let the per-iteration launches below
        // finish the loop from there.
        ++persist_giveups();
        if (xmca_trace("giveup")) std::fprintf(stderr, "xmca: varimax_persistent_kernel (%d workgroups) gave up after %d iterations\n", d.nwg, (int)state[0]);
        state[4] = 0.0;
        XMCA_HIP(hipMemcpyAsync(d.state.get(), state, sizeof(state), hipMemcpyHostToDevice, st));
        XMCA_HIP(hipStreamSynchronize(st));
        launched = (int)state[0];
      } else {
        launched = max_iter;
      }
    }
    while (launched < max_iter) {
      const int batch = std::min(32, max_iter - launched);
      for (int b = 0; b < batch; ++b) {
        if (fused) {
          hipLaunchKernelGGL((varimax_iter_kernel<CPLX>), dim3(d.nwg), dim3(256), fused_smem, st, d.A.r(), d.A.i(CPLX), d.h.get(), d.N,
                             p, d.A0.r(), d.A0.i(CPLX), d.R.r(), d.R.i(CPLX), d.cvec.get(), d.state.get(), d.part_r.get(),
                             CPLX ? d.part_i.get() : nullptr, d.counter.get(), tol, gamma);
        } else {
          accum<CPLX, 0, 0>(d, gamma, nullptr, nullptr);      // (MODE 0 carries gamma in the `power` slot)
          hipLaunchKernelGGL((varimax_step_kernel<CPLX>), dim3(1), dim3(256), 0, st, d.part_r.get(), CPLX ? d.part_i.get() : nullptr,
                             d.nwg, p, d.A0.r(), d.A0.i(CPLX), d.R.r(), d.R.i(CPLX), d.W.r(), d.W.i(CPLX), d.cvec.get(), d.state.get(), tol, 0);
        }
      }
      XMCA_HIP(hipGetLastError());
      launched += batch;
      XMCA_HIP(hipMemcpyAsync(state, d.state.get(), sizeof(state), hipMemcpyDeviceToHost, st));
      XMCA_HIP(hipStreamSynchronize(st));
      static const bool trace = xmca_trace("rot");
      if (trace)
        std::fprintf(stderr, "[xmca varimax] p=%d N=%lld cplx=%d launched=%d iter=%g conv=%g d=%.12g polar_its=%g nan=%g\n", p,
                     (long long)d.N, (int)CPLX, launched, state[0], state[1], state[2], state[5], state[4]);
      if (state[1] != 0.0 || state[4] != 0.0) break;
    }
    tm.end();
#ifdef XMCA_ROT_PROF
    {
      long long hs[16];
      XMCA_HIP(hipMemcpyFromSymbol(hs, HIP_SYMBOL(rot_prof), sizeof(hs)));
      std::fprintf(stderr, "[xmca varimax prof] accum %lld  publish %lld  wait %lld  acquire %lld  reduce %lld  newton-schulz %lld (its %g)  tail %lld  total %lld cycles (nwg %d)\n",
                   hs[1] - hs[0], hs[6] - hs[1], hs[7] - hs[6], hs[2] - hs[7], hs[3] - hs[2], hs[4] - hs[3], state[5], hs[5] - hs[4], hs[5] - hs[0], d.nwg);
      std::fprintf(stderr, "[xmca varimax prof] NS iteration 1: T tiles %lld  barrier %lld  flags %lld  Y tiles %lld  barrier %lld\n", hs[9] - hs[8],
                   hs[10] - hs[9], hs[11] - hs[10], hs[12] - hs[11], hs[13] - hs[12]);
      std::fprintf(stderr, "[xmca varimax prof] NS iteration 1: T products %lld  T epilogue %lld  Y products %lld  Y epilogue %lld\n", hs[14] - hs[8],
                   hs[9] - hs[14], hs[15] - hs[11], hs[12] - hs[15]);
    }
#endif
    res.iters = (int)state[0];
    res.converged = state[1] != 0.0;
    res.nan = state[4] != 0.0;
    res.last_d = state[2];
    if (!res.converged) return;

    std::vector<double> Rr((size_t)p * p), Ri((size_t)p * p, 0.0);
    XMCA_HIP(hipMemcpyAsync(Rr.data(), d.R.r(), sizeof(double) * p * p, hipMemcpyDeviceToHost, st));
    if (CPLX) XMCA_HIP(hipMemcpyAsync(Ri.data(), d.R.im.get(), sizeof(double) * p * p, hipMemcpyDeviceToHost, st));
    XMCA_HIP(hipStreamSynchronize(st));
    SmallMat Rv(p);
    for (int e = 0; e < p * p; ++e) Rv.a[e] = cd(Rr[e], Ri[e]);

    if (varimax_only) {
      res.R = Rv.a;
      res.Phi = SmallMat::eye(p).a;
      if (B_out_dev) apply<CPLX>(d, Rv, B_out_dev);
      return;
    }

    tm.begin("promax");
    // column maxima of X = rownormalised(B),  B = h (A R)
    XMCA_HIP(hipMemsetAsync(d.colmax.get(), 0, sizeof(double) * p, st));
    accum<CPLX, 3, 0>(d, (double)power, nullptr, nullptr);
    auto fetch = [&](SmallMat& M) {
      std::vector<double> r((size_t)p * p), i((size_t)p * p, 0.0);
      XMCA_HIP(hipMemcpyAsync(r.data(), d.acc.r(), sizeof(double) * p * p, hipMemcpyDeviceToHost, st));
      if (CPLX) XMCA_HIP(hipMemcpyAsync(i.data(), d.acc.im.get(), sizeof(double) * p * p, hipMemcpyDeviceToHost, st));
      XMCA_HIP(hipStreamSynchronize(st));
      M = SmallMat(p);
      for (int e = 0; e < p * p; ++e) M.a[e] = cd(r[e], i[e]);
    };
    SmallMat XX, XP, SL, SR;
    accum<CPLX, 2, 0>(d, (double)power, d.acc.r(), d.acc.i(CPLX)); fetch(XX);
    accum<CPLX, 2, 1>(d, (double)power, d.acc.r(), d.acc.i(CPLX)); fetch(XP);
    accum<CPLX, 2, 2>(d, (double)power, d.acc.r(), d.acc.i(CPLX)); fetch(SL);
    accum<CPLX, 2, 3>(d, (double)power, d.acc.r(), d.acc.i(CPLX)); fetch(SR);
    tm.end();

    finish_promax<CPLX>(d, Rv, XX, XP, SL, SR, res, B_out_dev);
  }

  // p x p tail of Promax on the host (rotation.py:128-147) from the four p x p sums of the N-sized passes
  template <bool CPLX>
  void finish_promax(RotationDevice& d, const SmallMat& Rv, const SmallMat& XX, const SmallMat& XP, const SmallMat& SL, const SmallMat& SR,
                     RotateResult& res, double* B_out_dev) {
    const int p = d.p;
    // L = inv(X^H X) X^H P ; sigma = diag(inv(L^H L)) ; L <- L sqrt(sigma) ; R <- R L ; Phi = L^-1 L^-H
    SmallMat XXinv, L, LLinv, Linv;
    XMCA_CHECK(XX.inverse(XXinv), XMCA_ERR_NUMERIC, "promax: X^H X is singular");
    L = XXinv * XP;
    const SmallMat LL = L.H() * L;
    XMCA_CHECK(LL.inverse(LLinv), XMCA_ERR_NUMERIC, "promax: L^H L is singular (the reference falls back to pinv here)");
    for (int j = 0; j < p; ++j) {
      const cd sc = std::sqrt(LLinv(j, j));
      for (int i = 0; i < p; ++i) L(i, j) *= sc;
    }
    const SmallMat Rf = Rv * L;
    XMCA_CHECK(L.inverse(Linv), XMCA_ERR_NUMERIC, "promax: L is singular");
    const SmallMat Phi = Linv * Linv.H();
    res.R = Rf.a;
    res.Phi = Phi.a;
    // column norms of the two row blocks of B = h2 (X L):  ||B_blk[:,k]||^2 = (L^H S_blk L)_kk
    const SmallMat nl = L.H() * SL * L, nr = L.H() * SR * L;
    res.norm_left.resize(p);
    res.norm_right.resize(p);
    for (int k = 0; k < p; ++k) {
      res.norm_left[k] = std::sqrt(std::max(nl(k, k).real(), 0.0));
      res.norm_right[k] = std::sqrt(std::max(nr(k, k).real(), 0.0));
    }
    if (B_out_dev) apply<CPLX>(d, Rf, B_out_dev);
  }

  // -------------------------------------------------------------------------------------------------------------
  // GEMM-based Varimax / Promax for ANY number of modes (the reference has no limit on n_rot; the fused kernels keep
  // R, a 64-point slab of Z and of W in the LDS of one workgroup: <= 64 real / 48 complex modes).  Per iteration
  // (rotation.py:52-64):  Zt = R^T A (p x N GEMM), c_k = sum_n |Zt_kn|^2, W = (|Z|^2 - gamma c / N) Z in place,
  // G = A^H W (p x p GEMM), and the unitary polar factor of G through the Hermitian EVD of G^H G = Q L Q^H:
  // R = G Q L^-1/2 Q^H, polished by one Newton-Schulz step (the squared conditioning of G^H G), d = sum sqrt(l_i).
  // The stopping rule runs on the host (one synchronisation per iteration, which the EVD needs anyway).  ~2 ms per
  // iteration instead of ~20 us: a fallback for unusual n_rot, not a fast path.
  // -------------------------------------------------------------------------------------------------------------
  template <bool CPLX>
  void run_generic(RotationDevice& d, int power, double tol, int max_iter, RotateResult& res, double* B_out_dev, bool varimax_only) {
    const int p = d.p;
    const int64_t N = d.N;
    const size_t pp = (size_t)p * p;
    CPlanes Zt, G, H, Qh, T1, R2;
    Zt.ensure((size_t)p * N, CPLX);
    G.ensure(pp, CPLX); H.ensure(pp, CPLX); Qh.ensure(pp, CPLX); T1.ensure(pp, CPLX); R2.ensure(pp, CPLX);
    DevBuf<double> cvec, isq;
    cvec.ensure((size_t)p);
    isq.ensure((size_t)p);
    std::vector<double> eye(pp, 0.0), lam, inv_s((size_t)p);
    for (int i = 0; i < p; ++i) eye[(size_t)i * p + i] = 1.0;
    XMCA_HIP(hipMemcpyAsync(d.R.r(), eye.data(), sizeof(double) * pp, hipMemcpyHostToDevice, st));
    if (CPLX) XMCA_HIP(hipMemsetAsync(d.R.im.get(), 0, sizeof(double) * pp, st));
    auto rotate_A = [&](CPlanes& out) {     // out = R^T A  (p x N, mode-major like A)
      cgemm<double>(st, *gws, d.R.r(), d.R.i(CPLX), p, false, false, d.A.r(), d.A.i(CPLX), N, true, false, out.r(), out.i(CPLX), N, p, (int)N,
                    p, 1.0, nullptr, nullptr, false);
    };
    // C = X^H Y over the grid points [n0, n1):  C[j][k] = sum_n conj(Xt[j][n]) Yt[k][n]
    auto gram = [&](const CPlanes& X, const CPlanes& Y, int64_t n0, int64_t n1, CPlanes& C) {
      cgemm<double>(st, *gws, X.r() + n0, CPLX ? X.im.get() + n0 : nullptr, N, true, true, Y.r() + n0, CPLX ? Y.im.get() + n0 : nullptr, N, false,
                    false, C.r(), C.i(CPLX), p, p, p, (int)(n1 - n0), 1.0, nullptr, nullptr, false);
    };
    tm.begin("varimax");
    double dsum = 0.0;
    res.iters = 0; res.converged = false; res.nan = false;
    for (int it = 0; it < max_iter; ++it) {
      const double d_old = dsum;
      rotate_A(Zt);
      hipLaunchKernelGGL(rot_row_reduce_kernel, dim3(p), dim3(256), 0, st, Zt.r(), Zt.i(CPLX), N, 0, cvec.get());
      hipLaunchKernelGGL(rot_w_kernel, ew_grid((int64_t)p * N), dim3(EW_BLOCK), 0, st, Zt.r(), Zt.i(CPLX), N, p, cvec.get(), gamma);
      gram(d.A, Zt, 0, N, G);
      // polar factor of G
      cgemm<double>(st, *gws, G.r(), G.i(CPLX), p, false, true, G.r(), G.i(CPLX), p, true, false, H.r(), H.i(CPLX), p, p, p, p, 1.0, nullptr,
                    nullptr, true);
      hermitian_evd(st, *ews, H.r(), H.i(CPLX), p, p, lam, nullptr, Qh.r(), Qh.i(CPLX), p);
      dsum = 0.0;
      bool bad = false;
      for (int i = 0; i < p; ++i) {
        const double sv = std::sqrt(std::max(lam[i], 0.0));
        dsum += sv;
        if (!(sv > 1e-14 * std::sqrt(std::max(lam[0], 0.0))) || !std::isfinite(sv)) bad = true;
        inv_s[i] = 1.0 / sv;
      }
      res.iters = it + 1;
      if (bad || !(dsum == dsum)) { res.nan = true; break; }
      XMCA_HIP(hipMemcpyAsync(isq.get(), inv_s.data(), sizeof(double) * p, hipMemcpyHostToDevice, st));
      // T1 = G Q diag(1/s)  (Q[l][i] = conj(Qh[i][l]));  R = T1 Q^H  (Q^H[i][k] = Qh[i][k])
      cgemm<double>(st, *gws, G.r(), G.i(CPLX), p, true, false, Qh.r(), Qh.i(CPLX), p, false, true, T1.r(), T1.i(CPLX), p, p, p, p, 1.0, nullptr,
                    isq.get(), false);
      cgemm<double>(st, *gws, T1.r(), T1.i(CPLX), p, true, false, Qh.r(), Qh.i(CPLX), p, true, false, R2.r(), R2.i(CPLX), p, p, p, p, 1.0, nullptr,
                    nullptr, false);
      // one Newton-Schulz step: R = R2 (1.5 I - 0.5 R2^H R2)
      cgemm<double>(st, *gws, R2.r(), R2.i(CPLX), p, false, true, R2.r(), R2.i(CPLX), p, true, false, H.r(), H.i(CPLX), p, p, p, p, -0.5, nullptr,
                    nullptr, true);
      hipLaunchKernelGGL(add_diag_kernel, dim3(ceil_div(p, 256)), dim3(256), 0, st, H.r(), (int64_t)p, p, 1.5);
      cgemm<double>(st, *gws, R2.r(), R2.i(CPLX), p, true, false, H.r(), H.i(CPLX), p, true, false, d.R.r(), d.R.i(CPLX), p, p, p, p, 1.0, nullptr,
                    nullptr, false);
      XMCA_HIP(hipGetLastError());
      XMCA_HIP(hipStreamSynchronize(st));       // inv_s is reused by the next iteration
      if (std::fabs(dsum - d_old) / dsum < tol) { res.converged = true; break; }      // rotation.py:62
    }
    tm.end();
    res.last_d = dsum;
    if (!res.converged) return;

    std::vector<double> Rr(pp), Ri(pp, 0.0);
    XMCA_HIP(hipMemcpyAsync(Rr.data(), d.R.r(), sizeof(double) * pp, hipMemcpyDeviceToHost, st));
    if (CPLX) XMCA_HIP(hipMemcpyAsync(Ri.data(), d.R.im.get(), sizeof(double) * pp, hipMemcpyDeviceToHost, st));
    XMCA_HIP(hipStreamSynchronize(st));
    SmallMat Rv(p);
    for (size_t e = 0; e < pp; ++e) Rv.a[e] = cd(Rr[e], Ri[e]);
    if (varimax_only) {
      res.R = Rv.a;
      res.Phi = SmallMat::eye(p).a;
      if (B_out_dev) apply<CPLX>(d, Rv, B_out_dev);
      return;
    }
    tm.begin("promax");
    // X = row-normalised (A R), B = h (A R) per grid point (rotation.py:115-117), target P (:121-124), then the four sums
    CPlanes Bs, Pt, C;
    Bs.ensure((size_t)p * N, CPLX); Pt.ensure((size_t)p * N, CPLX); C.ensure(pp, CPLX);
    rotate_A(Zt);
    XMCA_HIP(hipMemcpyAsync(Bs.r(), Zt.r(), sizeof(double) * (size_t)p * N, hipMemcpyDeviceToDevice, st));
    if (CPLX) XMCA_HIP(hipMemcpyAsync(Bs.im.get(), Zt.im.get(), sizeof(double) * (size_t)p * N, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(rot_point_scale_kernel, ew_grid(N), dim3(EW_BLOCK), 0, st, Bs.r(), Bs.i(CPLX), N, p, d.h.get(), 0);
    hipLaunchKernelGGL(rot_point_scale_kernel, ew_grid(N), dim3(EW_BLOCK), 0, st, Zt.r(), Zt.i(CPLX), N, p, (const double*)nullptr, 1);
    hipLaunchKernelGGL(rot_row_reduce_kernel, dim3(p), dim3(256), 0, st, Zt.r(), Zt.i(CPLX), N, 1, cvec.get());
    hipLaunchKernelGGL(rot_target_kernel, ew_grid((int64_t)p * N), dim3(EW_BLOCK), 0, st, Zt.r(), Zt.i(CPLX), N, p, cvec.get(), (double)power,
                       Pt.r(), Pt.i(CPLX));
    XMCA_HIP(hipGetLastError());
    auto fetch = [&](SmallMat& M) {
      std::vector<double> r(pp), i(pp, 0.0);
      XMCA_HIP(hipMemcpyAsync(r.data(), C.r(), sizeof(double) * pp, hipMemcpyDeviceToHost, st));
      if (CPLX) XMCA_HIP(hipMemcpyAsync(i.data(), C.im.get(), sizeof(double) * pp, hipMemcpyDeviceToHost, st));
      XMCA_HIP(hipStreamSynchronize(st));
      M = SmallMat(p);
      for (size_t e = 0; e < pp; ++e) M.a[e] = cd(r[e], i[e]);
    };
    SmallMat XX, XP, SL, SR;
    gram(Zt, Zt, 0, N, C); fetch(XX);
    gram(Zt, Pt, 0, N, C); fetch(XP);
    SL = SmallMat(p); SR = SmallMat(p);
    if (d.Nleft > 0) { gram(Bs, Bs, 0, d.Nleft, C); fetch(SL); }
    if (d.Nleft < N) { gram(Bs, Bs, d.Nleft, N, C); fetch(SR); }
    tm.end();
    finish_promax<CPLX>(d, Rv, XX, XP, SL, SR, res, B_out_dev);
  }

  // B = h (A M)  -> N x p row-major (interleaved complex) on the device
  template <bool CPLX>
  void apply(RotationDevice& d, const SmallMat& M, double* B_out_dev) {
    const int p = d.p;
    std::vector<double> mr((size_t)p * p), mi((size_t)p * p);
    for (int e = 0; e < p * p; ++e) { mr[e] = M.a[e].real(); mi[e] = M.a[e].imag(); }
    XMCA_HIP(hipMemcpyAsync(d.acc.r(), mr.data(), sizeof(double) * p * p, hipMemcpyHostToDevice, st));
    if (CPLX) XMCA_HIP(hipMemcpyAsync(d.acc.im.get(), mi.data(), sizeof(double) * p * p, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL((rot_apply_kernel<CPLX>), ew_grid(d.N), dim3(EW_BLOCK), 0, st, d.A.r(), d.A.i(CPLX), d.h.get(), d.N, p,
                       d.acc.r(), d.acc.i(CPLX), B_out_dev);
    XMCA_HIP(hipGetLastError());
    XMCA_HIP(hipStreamSynchronize(st));
  }
};

}  // namespace xmca
