// hipMalloc cost by size (first touch included or not): is a solve's first-call overhead per call or per byte?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(nullptr);
  void* warm; hipMalloc(&warm, 1 << 20);
  for (size_t mb : {1, 8, 50, 50, 50, 200, 500, 2000}) {
    void* p = nullptr;
    double t0 = now();
    hipMalloc(&p, mb << 20);
    double t1 = now();
    hipMemsetAsync(p, 0, mb << 20, 0);
    hipStreamSynchronize(0);
    double t2 = now();
    hipMemsetAsync(p, 0, mb << 20, 0);
    hipStreamSynchronize(0);
    double t3 = now();
    printf("%5zu MB: hipMalloc %.3f ms, first memset %.3f ms, second %.3f ms\n", mb, t1 - t0, t2 - t1, t3 - t2);
  }
  // free + malloc again (does the runtime cache?)
  void* q; hipMalloc(&q, 50 << 20); hipFree(q);
  double t0 = now(); hipMalloc(&q, 50 << 20); double t1 = now();
  printf("50 MB after a free of the same size: %.3f ms\n", t1 - t0);
  return 0;
}
