// Stand-alone check + timing of csrc/chol64.h / cholesky.h (builds in well under a minute; the library takes two).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include scripts/probes/chol_probe.cpp -o scripts/probes/chol_probe
//   scripts/probes/chol_probe            # check n = 64 ... 700 real and complex against a host factorisation, then time n = 2920 / 2501c
#include <chrono>
#include <cmath>
#include <complex>
#include <cstring>
#include <random>

#include "../../xmca_amd/csrc/cholesky.h"

using namespace xmca;
using cd = std::complex<double>;

static void make_spd(int n, bool cplx, std::vector<double>& Ar, std::vector<double>& Ai, unsigned seed) {
  // A = B B^H / k + I  with B n x k, k = n + 8
  const int k = n + 8;
  std::mt19937 rng(seed);
  std::normal_distribution<double> N(0, 1);
  std::vector<cd> B((size_t)n * k);
  for (auto& x : B) x = cd(N(rng), cplx ? N(rng) : 0.0);
  Ar.assign((size_t)n * n, 0.0);
  Ai.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) {
      cd s = 0;
      for (int q = 0; q < k; ++q) s += B[(size_t)i * k + q] * std::conj(B[(size_t)j * k + q]);
      s /= (double)k;
      if (i == j) s = cd(s.real() + 1.0, 0.0);
      Ar[(size_t)i * n + j] = s.real(); Ai[(size_t)i * n + j] = s.imag();
      Ar[(size_t)j * n + i] = s.real(); Ai[(size_t)j * n + i] = -s.imag();
    }
}

static int check(hipStream_t st, GemmWorkspace& ws, int n, bool cplx) {
  std::vector<double> Ar, Ai;
  make_spd(n, cplx, Ar, Ai, 17 * n + cplx);
  DevBuf<double> dr, di;
  const int64_t ld = n + 3;
  std::vector<double> pr((size_t)n * ld, 0.0), pi((size_t)n * ld, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) { pr[(size_t)i * ld + j] = Ar[(size_t)i * n + j]; pi[(size_t)i * ld + j] = Ai[(size_t)i * n + j]; }
  XMCA_HIP(hipMemcpy(dr.ensure(pr.size()), pr.data(), sizeof(double) * pr.size(), hipMemcpyHostToDevice));
  if (cplx) XMCA_HIP(hipMemcpy(di.ensure(pi.size()), pi.data(), sizeof(double) * pi.size(), hipMemcpyHostToDevice));
  const bool ok = cholesky_upper(st, ws, dr.get(), cplx ? di.get() : nullptr, n, ld, 0.0);
  XMCA_HIP(hipMemcpy(pr.data(), dr.get(), sizeof(double) * pr.size(), hipMemcpyDeviceToHost));
  if (cplx) XMCA_HIP(hipMemcpy(pi.data(), di.get(), sizeof(double) * pi.size(), hipMemcpyDeviceToHost));
  // R^H R against A, and the strictly lower triangle must be zero
  double err = 0, low = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      if (i > j) { low = std::max(low, std::fabs(pr[(size_t)i * ld + j]) + std::fabs(pi[(size_t)i * ld + j])); continue; }
      cd s = 0;
      for (int q = 0; q <= i; ++q) s += std::conj(cd(pr[(size_t)q * ld + i], pi[(size_t)q * ld + i])) * cd(pr[(size_t)q * ld + j], pi[(size_t)q * ld + j]);
      err = std::max(err, std::abs(s - cd(Ar[(size_t)i * n + j], Ai[(size_t)i * n + j])));
    }
  const bool pass = ok && err < 1e-12 * n && low == 0.0;
  std::printf("check n = %4d %s: ok = %d  max|R^H R - A| = %.2e  lower = %.1e  %s\n", n, cplx ? "complex" : "real   ", (int)ok, err, low, pass ? "PASS" : "FAIL");
  return pass ? 0 : 1;
}

__global__ void fill_spd_kernel(double* Ar, double* Ai, int n, int64_t ld) {
  // diagonally dominant Hermitian matrix: cheap, positive definite, dense
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)n * n; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / n), j = (int)(e % n);
    const int a = i < j ? i : j, b = i < j ? j : i;
    uint32_t h = (uint32_t)(a * 7919 + b) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const double v = (double)h / 4294967296.0 - 0.5, w = (double)(h * 3266489917u) / 4294967296.0 - 0.5;
    Ar[(int64_t)i * ld + j] = i == j ? (double)n : v;
    if (Ai) Ai[(int64_t)i * ld + j] = i == j ? 0.0 : (i < j ? w : -w);
  }
}

static void bench(hipStream_t st, GemmWorkspace& ws, int n, bool cplx, int reps) {
  const int64_t ld = (n + 15) & ~15;
  DevBuf<double> dr, di;
  dr.ensure((size_t)n * ld);
  if (cplx) di.ensure((size_t)n * ld);
  double best = 1e30, sum = 0;
  for (int r = 0; r < reps + 1; ++r) {
    hipLaunchKernelGGL(fill_spd_kernel, dim3(1024), dim3(256), 0, st, dr.get(), cplx ? di.get() : nullptr, n, ld);
    XMCA_HIP(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = cholesky_upper(st, ws, dr.get(), cplx ? di.get() : nullptr, n, ld, 1e-13);
    XMCA_HIP(hipStreamSynchronize(st));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!ok) std::printf("  (factorisation failed)\n");
    if (r) { best = std::min(best, ms); sum += ms; }
  }
  std::printf("bench n = %4d %s: %.3f ms best, %.3f ms mean of %d (%d panels: %.1f us per panel)\n", n, cplx ? "complex" : "real   ", best, sum / reps, reps,
              (n + 63) / 64, 1e3 * best / ((n + 63) / 64));
}

int main(int argc, char** argv) {
  hipStream_t st;
  XMCA_HIP(hipStreamCreate(&st));
  GemmWorkspace ws;
  DevPool pool;
  PoolScope scope(&pool);
  int bad = 0;
  try {
    for (int n : {16, 40, 64, 65, 100, 128, 200, 333, 700})
      for (int c = 0; c < 2; ++c) bad += check(st, ws, n, c != 0);
    bench(st, ws, 2920, false, 5);
    bench(st, ws, 2501, true, 5);
    bench(st, ws, 1000, false, 5);
    bench(st, ws, 451, true, 5);
  } catch (const Error& e) {
    std::printf("error: %s\n", e.what());
    return 2;
  }
  std::printf(bad ? "FAILED: %d cases\n" : "all checks passed\n", bad);
  return bad ? 1 : 0;
}
