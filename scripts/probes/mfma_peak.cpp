// MFMA issue-rate ceilings on this chip under sustained load (clock included): f64 16x16x4, f32 16x16x4, f32 32x32x2.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_peak.cpp -o scripts/probes/mfma_peak
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ double rnd(unsigned i, double a0) {
  unsigned h = i * 2654435761u + 12345u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  unsigned g = h * 747796405u + 2891336453u;
  return ((double)h + (double)g / 4294967296.0) / 4294967296.0 * 2.0 - 1.0 + a0;
}
template <int KIND>
__global__ __launch_bounds__(256, 2) void k(double* out, int iters, double a0) {
  if constexpr (KIND == 0) {
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd(threadIdx.x * 8 + i, a0); b[i] = rnd(threadIdx.x * 8 + 4 + i, a0); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i >> 2], b[i & 3], acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else if constexpr (KIND == 1) {
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = (float)rnd(threadIdx.x * 8 + i, a0); b[i] = (float)rnd(threadIdx.x * 8 + 4 + i, a0); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i >> 2], b[i & 3], acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = (float)rnd(threadIdx.x * 8 + i, a0); b[i] = (float)rnd(threadIdx.x * 8 + 4 + i, a0); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i >> 1], b[i & 1], acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}
template <int KIND>
void run(const char* name, int blocks, int iters, double flop_per_mfma, int mfma_per_iter) {
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-30);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-30);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * mfma_per_iter * flop_per_mfma;
    printf("%-22s blocks=%d  %.3f ms  %.1f TF\n", name, blocks, ms, fl / ms / 1e9);
  }
  hipFree(out);
}
int main() {
  for (int b : {256, 512}) {
    run<0>("f64 16x16x4", b, 20000, 2048, 16);
    run<1>("f32 16x16x4", b, 40000, 2048, 16);
    run<2>("f32 32x32x2", b, 40000, 4096, 4);
  }
}
