// Reproducer of the VMEM-store data hazard met in round 6 ("a hardware hazard found on the way"):
//   buffer_store_dwordx4 v[a:a+3], voff, s[rsrc], s_off offen sc1      <- byte offset (also) in a SCALAR register
//   v_mov_b32 v[a], ...                                                <- VALU write of a data register right behind it
// hipcc (ROCm 7.2) inserts the wait state (s_nop) between the two only when the scalar-offset field is NOT a register; the form
// WITH an SGPR offset gets none, and under memory-pipe back-pressure the store then ships the NEW contents of v[a].
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/store_hazard_probe.cpp -o scripts/probes/store_hazard_probe
//   scripts/probes/store_hazard_probe [rounds=200]
// Three variants of the same store loop, each next to a streaming kernel on a second stream (the back-pressure):
//   sgpr_nonop : SGPR offset, data register overwritten at once          -> corrupted dwords expected under load
//   sgpr_nop   : the same with `s_nop 1` in between                       -> clean
//   vgpr_only  : whole offset in the VGPR, constant 0 scalar offset, s_nop 1 (what the compiler emits itself) -> clean
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(_e)); return 2; } } while (0)

constexpr int ITERS = 64;          // stores per thread
constexpr unsigned POISON = 0xDEADBEEFu;

template <int VARIANT>
__global__ __launch_bounds__(256) void store_kernel(unsigned int* out, int soff_bytes) {
  // record r of thread t: 16 bytes at ((r * gridDim.x + blockIdx.x) * 256 + threadIdx.x) * 16, dword d = pattern(t, r, d)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
  const unsigned gt = blockIdx.x * 256u + threadIdx.x;
  for (int r = 0; r < ITERS; ++r) {
    const unsigned base = gt * 2654435761u + (unsigned)r * 40503u;
    unsigned voff = ((unsigned)(r * (int)gridDim.x + (int)blockIdx.x) * 256u + threadIdx.x) * 16u;
    unsigned soff = (unsigned)soff_bytes;                      // (a kernel argument: lives in an SGPR)
    if (VARIANT == 2) { voff += soff; soff = 0; }
    const unsigned d0 = base, d1 = base ^ 0x11111111u, d2 = base ^ 0x22222222u, d3 = base ^ 0x33333333u;
    if (VARIANT == 2) {
      asm volatile(
          "v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\t"
          "buffer_store_dwordx4 v[10:13], %4, %5, 0 offen sc1\n\t"
          "s_nop 1\n\t"
          "v_mov_b32 v10, %6\n\tv_mov_b32 v11, %6\n\tv_mov_b32 v12, %6\n\tv_mov_b32 v13, %6\n\t"
          :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(voff), "s"(rs), "v"(POISON) : "v10", "v11", "v12", "v13", "memory");
    } else if (VARIANT == 1) {
      asm volatile(
          "v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\t"
          "buffer_store_dwordx4 v[10:13], %4, %5, %7 offen sc1\n\t"
          "s_nop 1\n\t"
          "v_mov_b32 v10, %6\n\tv_mov_b32 v11, %6\n\tv_mov_b32 v12, %6\n\tv_mov_b32 v13, %6\n\t"
          :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(voff), "s"(rs), "v"(POISON), "s"(soff) : "v10", "v11", "v12", "v13", "memory");
    } else {
      asm volatile(
          "v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\t"
          "buffer_store_dwordx4 v[10:13], %4, %5, %7 offen sc1\n\t"
          "v_mov_b32 v10, %6\n\tv_mov_b32 v11, %6\n\tv_mov_b32 v12, %6\n\tv_mov_b32 v13, %6\n\t"
          :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(voff), "s"(rs), "v"(POISON), "s"(soff) : "v10", "v11", "v12", "v13", "memory");
    }
  }
}

__global__ void stream_kernel(const double4* __restrict__ in, double4* __restrict__ out, size_t n, int reps) {
  for (int q = 0; q < reps; ++q)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      double4 v = in[i];
      v.x += 1.0;
      out[i] = v;
    }
}

__global__ void verify_kernel(const unsigned int* __restrict__ buf, int grid, int soff_words, unsigned long long* bad) {
  const unsigned gt = blockIdx.x * 256u + threadIdx.x;
  unsigned long long n = 0;
  for (int r = 0; r < ITERS; ++r) {
    const unsigned base = gt * 2654435761u + (unsigned)r * 40503u;
    const size_t w = ((size_t)(r * grid + (int)blockIdx.x) * 256 + threadIdx.x) * 4 + soff_words;
    const unsigned want[4] = {base, base ^ 0x11111111u, base ^ 0x22222222u, base ^ 0x33333333u};
    for (int d = 0; d < 4; ++d) n += buf[w + d] != want[d];
  }
  if (n) atomicAdd(bad, n);
}

template <int VARIANT>
static long long run(const char* name, int rounds, bool loaded, hipStream_t s1, hipStream_t s2, unsigned int* dbuf, size_t words, double4* a, double4* b,
                     size_t nstream, unsigned long long* dbad) {
  const int grid = 1024, soff = 64;       // (the records start 64 bytes into the buffer)
  (void)hipMemset(dbad, 0, 8);
  for (int it = 0; it < rounds; ++it) {
    (void)hipMemsetAsync(dbuf, 0, words * 4, s1);
    if (loaded) hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s2, a, b, nstream, 2);
    hipLaunchKernelGGL((store_kernel<VARIANT>), dim3(grid), dim3(256), 0, s1, dbuf, soff);
    hipLaunchKernelGGL(verify_kernel, dim3(grid), dim3(256), 0, s1, dbuf, grid, soff / 4, dbad);
    (void)hipStreamSynchronize(s1);
    (void)hipStreamSynchronize(s2);
  }
  unsigned long long bad = 0;
  (void)hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
  std::printf("%-12s %-8s: %llu wrong dwords of %lld (%d launches of %d x 256 threads x %d stores)\n", name, loaded ? "loaded" : "idle", bad,
              (long long)rounds * grid * 256 * ITERS * 4, rounds, grid, ITERS);
  return (long long)bad;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 100;
  hipStream_t s1, s2;
  CHECK(hipStreamCreate(&s1));
  CHECK(hipStreamCreate(&s2));
  const size_t words = (size_t)ITERS * 1024 * 256 * 4 + 64;
  unsigned int* dbuf = nullptr;
  CHECK(hipMalloc(&dbuf, words * 4));
  const size_t nstream = (size_t)64 << 20;         // 2 GB of double4 each way
  double4 *a = nullptr, *b = nullptr;
  CHECK(hipMalloc(&a, nstream * sizeof(double4)));
  CHECK(hipMalloc(&b, nstream * sizeof(double4)));
  CHECK(hipMemset(a, 0, nstream * sizeof(double4)));
  unsigned long long* dbad = nullptr;
  CHECK(hipMalloc(&dbad, 8));
  for (int loaded = 0; loaded < 2; ++loaded) {
    run<0>("sgpr_nonop", rounds, loaded, s1, s2, dbuf, words, a, b, nstream, dbad);
    run<1>("sgpr_nop", rounds, loaded, s1, s2, dbuf, words, a, b, nstream, dbad);
    run<2>("vgpr_only", rounds, loaded, s1, s2, dbuf, words, a, b, nstream, dbad);
  }
  return 0;
}
