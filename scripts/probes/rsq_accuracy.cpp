// Relative error of the hardware v_rsq_f64 / v_rcp_f64 seeds and after 1, 2, 3 Newton steps (gfx950).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double y = __builtin_amdgcn_rsq(v);
  out[i * 8 + 0] = y;
  for (int it = 0; it < 3; ++it) { const double h = 0.5 * v * y; y = fma(y, fma(-h, y, 0.5), y); out[i * 8 + 1 + it] = y; }
  double r = __builtin_amdgcn_rcp(v);
  out[i * 8 + 4] = r;
  for (int it = 0; it < 3; ++it) { r = fma(r, fma(-v, r, 1.0), r); out[i * 8 + 5 + it] = r; }
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o((size_t)n * 8);
  for (int i = 0; i < n; ++i) x[i] = std::exp(-40.0 + 80.0 * (double)i / n) * (1.0 + 0.37 * ((i * 2654435761u) % 1000) / 1000.0);
  double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, (size_t)n * 64);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, (size_t)n * 64, hipMemcpyDeviceToHost);
  double e[8] = {0};
  for (int i = 0; i < n; ++i) {
    const long double rs = 1.0L / sqrtl((long double)x[i]), rc = 1.0L / (long double)x[i];
    for (int j = 0; j < 4; ++j) e[j] = std::fmax(e[j], (double)fabsl(((long double)o[(size_t)i * 8 + j] - rs) / rs));
    for (int j = 4; j < 8; ++j) e[j] = std::fmax(e[j], (double)fabsl(((long double)o[(size_t)i * 8 + j] - rc) / rc));
  }
  printf("rsq: seed %.2e, +1 %.2e, +2 %.2e, +3 %.2e\nrcp: seed %.2e, +1 %.2e, +2 %.2e, +3 %.2e\n", e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
  return 0;
}
