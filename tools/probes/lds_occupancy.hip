// Development probe: how many single-wave workgroups with a given LDS footprint does one gfx950 CU hold at once?
// Each workgroup spins for a fixed wall time; the launch time steps up when the grid exceeds the resident capacity.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_occ tools/probes/lds_occupancy.hip && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) spin(double* out, long long ticks) {
  extern __shared__ double sm[];
  sm[threadIdx.x] = threadIdx.x;
  const long long t0 = wall_clock64();
  double a = sm[(threadIdx.x + 1) & 63];
  while (wall_clock64() - t0 < ticks) a = a * 1.0000001 + 1e-9;
  if (a == 12345.0) out[0] = a;
}
int main() {
  double* out; hipMalloc(&out, 8);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int kbs[] = {16, 24, 30, 32, 34, 36, 38, 40, 48, 52, 64, 80};
  for (int kb : kbs) {
    printf("LDS %2d KB:", kb);
    for (int per_cu = 1; per_cu <= 10; per_cu++) {
      hipLaunchKernelGGL(spin, dim3(256 * per_cu), dim3(64), kb * 1024, 0, out, 20000LL);   // 200 us
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(spin, dim3(256 * per_cu), dim3(64), kb * 1024, 0, out, 20000LL);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf(" %d:%.2f", per_cu, ms);
    }
    printf("\n");
  }
  return 0;
}
