// Stand-alone check + timing of csrc/gemm.h (builds in seconds; the library takes minutes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include scripts/probes/gemm_probe.cpp -o scripts/probes/gemm_probe
//   scripts/probes/gemm_probe check        # all orientations / dtypes / edges against a host reference
//   scripts/probes/gemm_probe bench        # the shapes of the path
#include <cmath>
#include <cstring>
#include <random>

#include "../../xmca_amd/csrc/gemm.h"

using namespace xmca;

template <typename T>
__global__ void fill_kernel(T* x, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    x[i] = (T)((double)h / 4294967296.0 * 2.0 - 1.0);
  }
}

template <typename TI>
static int check_case(hipStream_t st, GemmWorkspace& ws, int M, int N, int K, bool akf, bool bnf, bool upper, int splits, int pad,
                      bool misalign, double alpha, double beta, bool scales) {
  const int64_t lda = (akf ? K : M) + pad, ldb = (bnf ? N : K) + pad, ldc = N + 3;
  const size_t na = (size_t)(akf ? M : K) * lda, nb = (size_t)(bnf ? K : N) * ldb, nc = (size_t)M * ldc;
  std::vector<TI> hA(na + 1), hB(nb + 1);
  std::vector<double> hC(nc), hC0(nc), rs(M), cs(N);
  std::mt19937 rng(M * 7 + N * 3 + K);
  std::uniform_real_distribution<double> U(-1, 1);
  for (auto& x : hA) x = (TI)U(rng);
  for (auto& x : hB) x = (TI)U(rng);
  for (auto& x : hC0) x = U(rng);
  for (auto& x : rs) x = 0.5 + U(rng) * 0.25;
  for (auto& x : cs) x = 0.5 + U(rng) * 0.25;
  DevBuf<TI> dA, dB;
  DevBuf<double> dC, drs, dcs;
  const int off = misalign ? 1 : 0;
  XMCA_HIP(hipMemcpy(dA.ensure(na + 1), hA.data(), sizeof(TI) * (na + 1), hipMemcpyHostToDevice));
  XMCA_HIP(hipMemcpy(dB.ensure(nb + 1), hB.data(), sizeof(TI) * (nb + 1), hipMemcpyHostToDevice));
  XMCA_HIP(hipMemcpy(dC.ensure(nc), hC0.data(), sizeof(double) * nc, hipMemcpyHostToDevice));
  XMCA_HIP(hipMemcpy(drs.ensure(M), rs.data(), sizeof(double) * M, hipMemcpyHostToDevice));
  XMCA_HIP(hipMemcpy(dcs.ensure(N), cs.data(), sizeof(double) * N, hipMemcpyHostToDevice));
  GemmOpts o;
  o.a_kfast = akf; o.b_nfast = bnf; o.alpha = alpha; o.beta = beta; o.upper_only = upper; o.mirror = upper ? 1 : 0; o.force_splits = splits;
  if (scales) { o.row_scale = drs.get(); o.col_scale = dcs.get(); }
  const TI* A = dA.get() + off;
  const TI* B = (upper ? dA.get() : dB.get()) + off;
  const TI* hAp = hA.data() + off;
  const TI* hBp = (upper ? hA.data() : hB.data()) + off;
  const int64_t ldb_ = upper ? lda : ldb;
  const bool bnf_ = upper ? !akf : bnf;       // Gram: B = A^T of the same array
  o.b_nfast = bnf_;
  for (int rep = 0; rep < 2; ++rep) {         // twice: the tickets must be back at zero
    XMCA_HIP(hipMemcpy(dC.get(), hC0.data(), sizeof(double) * nc, hipMemcpyHostToDevice));
    gemm<TI, double>(st, ws, A, lda, B, ldb_, dC.get(), ldc, M, N, K, o);
    XMCA_HIP(hipStreamSynchronize(st));
  }
  XMCA_HIP(hipMemcpy(hC.data(), dC.get(), sizeof(double) * nc, hipMemcpyDeviceToHost));
  double err = 0, scale = 0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) {
        const double a = akf ? hAp[(int64_t)m * lda + k] : hAp[(int64_t)k * lda + m];
        const double b = bnf_ ? hBp[(int64_t)k * ldb_ + n] : hBp[(int64_t)n * ldb_ + k];
        s += a * b;
      }
      s *= alpha;
      if (scales) s *= rs[m] * cs[n];
      // upper_only + mirror (used with beta = 0, no scales): entries below the block diagonal are the mirrored tile
      const double ref = s + beta * hC0[(size_t)m * ldc + n];
      const double got = hC[(size_t)m * ldc + n];
      err = std::max(err, std::fabs(got - ref));
      scale = std::max(scale, std::fabs(ref));
    }
  // padding columns of C must be untouched
  int touched = 0;
  for (int m = 0; m < M; ++m)
    for (int n = N; n < ldc; ++n) touched += hC[(size_t)m * ldc + n] != hC0[(size_t)m * ldc + n];
  const double tol = (sizeof(TI) == 4 ? 3e-6 : 1e-13) * std::max(1.0, scale);
  const bool ok = err <= tol && touched == 0;
  printf("%s %s M=%d N=%d K=%d %s%s upper=%d splits=%d pad=%d misalign=%d beta=%g scales=%d  err=%.3e (tol %.1e) touched=%d\n",
         ok ? "ok  " : "FAIL", sizeof(TI) == 4 ? "f32" : "f64", M, N, K, akf ? "K" : "M", bnf_ ? "N" : "K", (int)upper, splits, pad, (int)misalign,
         beta, (int)scales, err, tol, touched);
  return ok ? 0 : 1;
}

template <typename TI>
static void bench_case(hipStream_t st, GemmWorkspace& ws, const char* name, int M, int N, int K, bool akf, bool bnf, bool upper, int splits,
                       int reps, double useful_flops) {
  const int64_t lda = akf ? K : M, ldb = bnf ? N : K;
  const size_t na = (size_t)(akf ? M : K) * lda, nb = (size_t)(bnf ? K : N) * ldb;
  DevBuf<TI> dA, dB;
  DevBuf<double> dC;
  hipLaunchKernelGGL((fill_kernel<TI>), dim3(2048), dim3(256), 0, st, dA.ensure(na), na, 17u);
  if (!upper) hipLaunchKernelGGL((fill_kernel<TI>), dim3(2048), dim3(256), 0, st, dB.ensure(nb), nb, 91u);
  dC.ensure((size_t)M * N);
  GemmOpts o;
  o.a_kfast = akf; o.b_nfast = upper ? !akf : bnf; o.upper_only = upper; o.mirror = upper ? 1 : 0; o.force_splits = splits;
  const TI* B = upper ? dA.get() : dB.get();
  const int64_t ldb_ = upper ? lda : ldb;
  hipEvent_t e0, e1;
  XMCA_HIP(hipEventCreate(&e0));
  XMCA_HIP(hipEventCreate(&e1));
  gemm<TI, double>(st, ws, dA.get(), lda, B, ldb_, dC.get(), N, M, N, K, o);
  XMCA_HIP(hipStreamSynchronize(st));
  float best = 1e30f, sum = 0;
  for (int r = 0; r < reps; ++r) {
    XMCA_HIP(hipEventRecord(e0, st));
    gemm<TI, double>(st, ws, dA.get(), lda, B, ldb_, dC.get(), N, M, N, K, o);
    XMCA_HIP(hipEventRecord(e1, st));
    XMCA_HIP(hipEventSynchronize(e1));
    float ms;
    XMCA_HIP(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
    sum += ms;
  }
  const double peak = sizeof(TI) == 4 ? 157.3e12 : 78.6e12;
  const double avg = sum / reps;
  const int tm = (M + 127) / 128, tn = (N + 127) / 128;
  const double tiles = upper ? tm * (tm + 1) / 2.0 : (double)tm * tn;
  const double launched = tiles * 2.0 * 128 * 128 * (double)K;
  printf("%-28s %s %s%s M=%d N=%d K=%d upper=%d splits=%d: avg %.3f ms (best %.3f)  useful %.1f TF = %.3f of peak   [MFMA work %.3f of peak]\n", name,
         sizeof(TI) == 4 ? "f32" : "f64", akf ? "K" : "M", o.b_nfast ? "N" : "K", M, N, K, (int)upper, splits, avg, best, useful_flops / avg / 1e9,
         useful_flops / (avg * 1e-3) / peak, launched / (avg * 1e-3) / peak);
  fflush(stdout);
}

struct Shape { const char* name; int M, N, K; bool akf, bnf, upper; double fl; };

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "check";
  hipStream_t st;
  XMCA_HIP(hipStreamCreate(&st));
  GemmWorkspace ws;
  int bad = 0;
  try {
    if (!strcmp(mode, "check")) {
      for (int akf = 0; akf < 2; ++akf)
        for (int bnf = 0; bnf < 2; ++bnf) {
          bad += check_case<double>(st, ws, 128, 128, 64, akf, bnf, false, 1, 0, false, 1.0, 0.0, false);
          bad += check_case<float>(st, ws, 128, 128, 64, akf, bnf, false, 1, 0, false, 1.0, 0.0, false);
          bad += check_case<double>(st, ws, 200, 300, 100, akf, bnf, false, 1, 0, false, 0.5, 0.0, true);     // edges, K tail
          bad += check_case<float>(st, ws, 200, 300, 100, akf, bnf, false, 1, 0, false, 0.5, 0.25, false);
          bad += check_case<double>(st, ws, 131, 257, 77, akf, bnf, false, 1, 1, false, 1.0, -0.5, false);    // odd leading dimension -> slow path
          bad += check_case<float>(st, ws, 131, 257, 77, akf, bnf, false, 1, 1, false, 1.0, 0.0, true);
          bad += check_case<double>(st, ws, 130, 260, 96, akf, bnf, false, 1, 0, true, 1.0, 0.0, false);      // misaligned base -> slow path
          bad += check_case<double>(st, ws, 260, 250, 1000, akf, bnf, false, 4, 0, false, 1.0, 0.75, true);   // split-K
          bad += check_case<float>(st, ws, 260, 250, 2000, akf, bnf, false, 5, 0, false, 2.0, 0.0, false);
          bad += check_case<float>(st, ws, 66, 130, 1201, akf, bnf, false, 3, 3, false, 1.0, 0.0, false);     // f32, lda % 4 != 0
          bad += check_case<double>(st, ws, 3, 5, 7, akf, bnf, false, 1, 1, false, 1.0, 0.0, false);
          bad += check_case<double>(st, ws, 131, 257, 64, akf, bnf, false, 1, 0, false, 1.0, 0.0, false);     // ragged rows, K a multiple of BK, no padding
          bad += check_case<float>(st, ws, 130, 259, 128, akf, bnf, false, 2, 0, false, 1.0, 0.0, false);
          bad += check_case<float>(st, ws, 129, 257, 96, akf, bnf, false, 1, 0, true, 1.0, 0.0, false);       // ... and a base that is only element aligned
          bad += check_case<double>(st, ws, 2501, 300, 2501, akf, bnf, false, 0, 0, false, 1.0, 0.0, false);  // odd leading dimension, heuristic splits
        }
      for (int akf = 0; akf < 2; ++akf) {                      // Gram products
        bad += check_case<double>(st, ws, 300, 300, 500, akf, 0, true, 1, 0, false, 1.0, 0.0, false);
        bad += check_case<double>(st, ws, 300, 300, 2000, akf, 0, true, 3, 0, false, 0.25, 0.0, false);
        bad += check_case<float>(st, ws, 390, 390, 3000, akf, 0, true, 4, 0, false, 1.0, 0.0, false);
      }
      printf("%s\n", bad ? "FAILURES" : "all ok");
    } else if (!strcmp(mode, "shape")) {      // shape M N K akf bnf upper splits reps [f32]
      const int M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]), akf = atoi(argv[5]), bnf = atoi(argv[6]), up = atoi(argv[7]);
      const int sp = atoi(argv[8]), reps = atoi(argv[9]);
      const double fl = up ? (double)M * (M + 1) * K : 2.0 * M * N * K;
      if (argc > 10) bench_case<float>(st, ws, "shape", M, N, K, akf, bnf, up, sp, reps, fl);
      else bench_case<double>(st, ws, "shape", M, N, K, akf, bnf, up, sp, reps, fl);
    } else if (!strcmp(mode, "sweep")) {
      const int reps = argc > 2 ? atoi(argv[2]) : 5;
      const Shape shapes[] = {
          {"C2 gram", 2920, 2920, 10000, true, false, true, 2920.0 * 2921 * 10000},
          {"TxTxT 2501", 2501, 2501, 2501, true, true, false, 2.0 * 2501 * 2501 * 2501},
          {"TxTxT 2920", 2920, 2920, 2920, true, true, false, 2.0 * 2920 * 2920 * 2920},
          {"WY VhZ 128", 128, 3048, 2920, false, true, false, 2.0 * 128 * 3048 * 2920},
          {"WY VhZ 256", 256, 3176, 2920, false, true, false, 2.0 * 256 * 3176 * 2920},
          {"WY Z-=VX 128", 2920, 2920, 128, true, true, false, 2.0 * 128 * 2920 * 2920},
          {"WY Z-=VX 256", 2920, 2920, 256, true, true, false, 2.0 * 256 * 2920 * 2920},
          {"C3 gram y (5000x15000)", 5000, 5000, 15000, true, false, true, 5000.0 * 5001 * 15000},
          {"C4 kernel 2501^3 NT", 2501, 2501, 2501, true, false, false, 2.0 * 2501 * 2501 * 2501},
          {"back-proj 2500x35000x2501", 2500, 35000, 2501, true, true, false, 2.0 * 2500 * 35000 * 2501},
      };
      for (const Shape& sh : shapes) {
        const int tm = (sh.M + 127) / 128, tn = (sh.N + 127) / 128;
        const int64_t tiles = sh.upper ? (int64_t)tm * (tm + 1) / 2 : (int64_t)tm * tn;
        printf("--- %s: tiles %lld, k-tiles %d, heuristic splits %d\n", sh.name, (long long)tiles, (sh.K + 15) / 16,
               gemm_choose_splits(tiles, (sh.K + 15) / 16, 256));
        for (int sp : {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 13, 16, 20}) {
          if (sp > 1 && (sh.K + 15) / 16 / sp < 2) continue;
          bench_case<double>(st, ws, sh.name, sh.M, sh.N, sh.K, sh.akf, sh.bnf, sh.upper, sp, reps, sh.fl);
        }
      }
      printf("--- C5 gram f32\n");
      for (int sp : {0, 4, 7, 8, 9, 10, 12, 16, 18, 27})
        bench_case<float>(st, ws, "C5 gram f32", 1200, 1200, 1036800, true, false, true, sp, 3, 1200.0 * 1201 * 1036800);
    } else {
      const int reps = argc > 2 ? atoi(argv[2]) : 10;
      const int sp = argc > 3 ? atoi(argv[3]) : 0;
      // C2 Gram (f64), T = 2920, N = 10 000
      bench_case<double>(st, ws, "C2 gram XX^T", 2920, 2920, 10000, true, false, true, sp, reps, 2920.0 * 2921 * 10000);
      bench_case<double>(st, ws, "C2 back-projection", 2920, 10000, 2920, true, true, false, sp, reps, 2.0 * 2920 * 2920 * 10000);
      bench_case<double>(st, ws, "C3 gram x (5000x20000)", 5000, 5000, 20000, true, false, true, sp, reps, 5000.0 * 5001 * 20000);
      bench_case<double>(st, ws, "dense 4096^3 NN", 4096, 4096, 4096, true, true, false, sp, reps, 2.0 * 4096 * 4096 * 4096);
      bench_case<double>(st, ws, "dense 4096^3 NT", 4096, 4096, 4096, true, false, false, sp, reps, 2.0 * 4096 * 4096 * 4096);
      bench_case<double>(st, ws, "dense 4096^3 TN", 4096, 4096, 4096, false, true, false, sp, reps, 2.0 * 4096 * 4096 * 4096);
      bench_case<double>(st, ws, "WY K=128: V^H Z", 128, 3048, 2920, false, true, false, sp, reps, 2.0 * 128 * 3048 * 2920);
      bench_case<double>(st, ws, "WY K=128: Z -= V X", 2920, 2920, 128, true, true, false, sp, reps, 2.0 * 128 * 2920 * 2920);
      bench_case<double>(st, ws, "T x T x T (2501)", 2501, 2501, 2501, true, true, false, sp, reps, 2.0 * 2501 * 2501 * 2501);
      bench_case<float>(st, ws, "C5 gram f32", 1200, 1200, 1036800, true, false, true, sp, std::max(reps / 3, 2), 1200.0 * 1201 * 1036800);
      bench_case<float>(st, ws, "C5 back-projection f32", 1200, 1036800 / 4, 1200, true, true, false, sp, std::max(reps / 3, 2), 2.0 * 1200 * 1200 * 259200);
      bench_case<float>(st, ws, "C3 gram f32 (5000x20000)", 5000, 5000, 20000, true, false, true, sp, reps, 5000.0 * 5001 * 20000);
      bench_case<float>(st, ws, "dense 4096^3 NN f32", 4096, 4096, 4096, true, true, false, sp, reps, 2.0 * 4096 * 4096 * 4096);
      bench_case<float>(st, ws, "dense 4096^3 NT f32", 4096, 4096, 4096, true, false, false, sp, reps, 2.0 * 4096 * 4096 * 4096);
    }
  } catch (const Error& e) {
    printf("ERROR %d: %s\n", e.code, e.what());
    return 2;
  }
  return bad ? 1 : 0;
}
