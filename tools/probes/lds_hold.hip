// Does LDS above 64 KB keep its contents?  One workgroup per CU fills NB bytes of LDS with a pattern, spins, and verifies -- repeatedly.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_hold tools/probes/lds_hold.hip && /tmp/lds_hold
// (written while chasing a sporadic extra rejected step in the wide local-energy kernels with more than 64 KB of LDS)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int NW>
__global__ void __launch_bounds__(256) hold(unsigned long long* bad, unsigned long long* first_bad, int rounds, long long spin) {
  __shared__ unsigned long long buf[NW];
  const int tid = threadIdx.x;
  for (int r = 0; r < rounds; r++) {
    const unsigned long long key = 0x9E3779B97F4A7C15ULL * (blockIdx.x + 1) + r;
    for (int e = tid; e < NW; e += 256) buf[e] = key ^ (unsigned long long)e * 0xD1B54A32D192ED03ULL;
    __syncthreads();
    const long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    __syncthreads();
    for (int e = tid; e < NW; e += 256)
      if (buf[e] != (key ^ (unsigned long long)e * 0xD1B54A32D192ED03ULL)) {
        atomicAdd(bad, 1ULL);
        atomicMin(first_bad, (unsigned long long)e * 8);
      }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 200;
  unsigned long long *d, h[2];
  hipMalloc(&d, 16);
  for (int big = 0; big < 2; big++) {
    h[0] = 0; h[1] = ~0ULL;
    hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    for (int l = 0; l < launches; l++) {
      if (big) hipLaunchKernelGGL(hold<17408>, dim3(256), dim3(256), 0, 0, d, d + 1, 8, 200000LL);     // 136 KB
      else hipLaunchKernelGGL(hold<7680>, dim3(256), dim3(256), 0, 0, d, d + 1, 8, 200000LL);           // 60 KB
    }
    hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%s LDS per workgroup: %d launches x 256 workgroups x 8 rounds: corrupted words %llu, lowest byte offset %lld\n",
           big ? "136 KB" : " 60 KB", launches, h[0], h[0] ? (long long)h[1] : -1LL);
  }
  return 0;
}
