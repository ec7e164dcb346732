// Is the workgroup -> CU placement of identical back-to-back launches repeatable?  (gfx950, 512 workgroups of 256
// threads with 66 KB of LDS = two per CU.)  Prints, per launch, how many workgroups sit on another CU than in launch 0
// and how many of the first 46 workgroups share their CU with another of the first 46.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* key, int spin) {
  __shared__ double pad[8400];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned k = (__builtin_amdgcn_s_getreg(6164) << 8) | __builtin_amdgcn_s_getreg(14852);
  // keep the workgroup alive for a while so that all 512 are resident together; low block ids stay longer
  long long t0 = __builtin_readcyclecounter();
  const long long wait = (blockIdx.x < 46) ? 3LL * spin : spin;
  while (__builtin_readcyclecounter() - t0 < wait) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0) key[blockIdx.x] = k + (unsigned)(pad[5] == 12345.0);
}
int main() {
  unsigned* d; hipMalloc(&d, 512 * 4);
  std::vector<unsigned> ref(512), cur(512);
  for (int it = 0; it < 8; ++it) {
    hipLaunchKernelGGL(probe, dim3(512), dim3(256), 0, 0, d, 100000);
    hipMemcpy(cur.data(), d, 512 * 4, hipMemcpyDeviceToHost);
    if (it == 0) ref = cur;
    int diff = 0, shared = 0;
    for (int i = 0; i < 512; ++i) diff += cur[i] != ref[i];
    for (int i = 0; i < 46; ++i) for (int j = 0; j < 46; ++j) if (i != j && cur[i] == cur[j]) { ++shared; break; }
    std::vector<int> cnt(1 << 16, 0); int distinct = 0;
    for (int i = 0; i < 512; ++i) if (cnt[cur[i] & 0xffff]++ == 0) ++distinct;
    printf("launch %d: %d of 512 workgroups on a different CU than in launch 0; %d of the first 46 share a CU among themselves; %d distinct CUs\n", it, diff, shared, distinct);
  }
  printf("partners of the first 46 workgroups (launch 7):");
  for (int i = 0; i < 46; ++i) { for (int j = 46; j < 512; ++j) if (cur[j] == cur[i]) printf(" %d:%d", i, j); }
  printf("\n");
  // back-to-back without host sync in between
  for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(probe, dim3(512), dim3(256), 0, 0, d, 100000);
  hipMemcpy(cur.data(), d, 512 * 4, hipMemcpyDeviceToHost);
  int diff = 0; for (int i = 0; i < 512; ++i) diff += cur[i] != ref[i];
  printf("after 4 back-to-back launches: %d differ from launch 0\n", diff);
  return 0;
}
