// Probe (not part of the product): lane layouts and issue cost of the fp64 MFMA instructions on gfx950, alone and
// interleaved with fp64 VALU work of the same wave.   hipcc --offload-arch=gfx950 -O3 mfma_f64.hip -o mfma_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout4(const double* a, const double* b, double* d) {
  int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
__global__ void layout16(const double* a, const double* b, double* d) {
  int l = threadIdx.x;
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) d[r * 64 + l] = c[r];
}
// timing: K iterations of (NM mfma + NV dfma), independent chains
template <int NM, int NV, int WHICH>
__global__ void timing(double* out, long long* cyc, int iters) {
  int l = threadIdx.x;
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  d4 m16[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  double v[16];
  for (int q = 0; q < 16; q++) v[q] = q * 0.1;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < NM; q++) {
      if (WHICH == 4) m[q % 8] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m[q % 8], 0, 0, 0);
      else m16[q % 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m16[q % 4], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < NV; q++) v[q % 16] = fma(v[q % 16], a, b);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int q = 0; q < 8; q++) s += m[q];
  for (int q = 0; q < 4; q++) s += m16[q][0] + m16[q][1] + m16[q][2] + m16[q][3];
  for (int q = 0; q < 16; q++) s += v[q];
  out[blockIdx.x * blockDim.x + l] = s;
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NM, int NV, int WHICH>
void run(const char* name, int blocks, int threads) {
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, sizeof(long long) * blocks);
  int iters = 2000;
  timing<NM, NV, WHICH><<<blocks, threads>>>(out, cyc, iters);
  timing<NM, NV, WHICH><<<blocks, threads>>>(out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  printf("%-34s blocks %4d x %3d thr: %7.1f cycles/iter  (%d mfma + %d dfma per iter)\n", name, blocks, threads, (double)h[0] / iters, NM, NV);
  hipFree(out); hipFree(cyc);
}

int main() {
  double ha[64], hb[64], hd[256];
  double *a, *b, *d;
  hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 2048);
  // layout of 4x4x4 (4 blocks): probe with one-hot A and B entries
  printf("== v_mfma_f64_4x4x4_4b: for each (la, lb) one-hot pair with nonzero output: which lane receives\n");
  int mapA_i[64], mapA_k[64];
  for (int la = 0; la < 16; la++) {
    for (int lb = 0; lb < 16; lb++) {
      for (int q = 0; q < 64; q++) { ha[q] = 0; hb[q] = 0; }
      ha[la] = 1.0; hb[lb] = 1.0;
      hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
      layout4<<<1, 64>>>(a, b, d);
      hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
      for (int q = 0; q < 64; q++) if (hd[q] != 0.0) printf("  A lane %2d x B lane %2d -> D lane %2d\n", la, lb, q);
    }
  }
  // cross-block check: A in block 0, B in block 1 should give nothing
  for (int q = 0; q < 64; q++) { ha[q] = 0; hb[q] = 0; }
  ha[0] = 1.0; hb[16] = 1.0;
  hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
  layout4<<<1, 64>>>(a, b, d); hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
  int nz = 0; for (int q = 0; q < 64; q++) nz += hd[q] != 0.0;
  printf("  A lane 0 x B lane 16 -> %d nonzero outputs (blocks are independent if 0)\n", nz);
  printf("== v_mfma_f64_16x16x4: A lane la, B lane lb one-hot -> (reg, lane) of nonzero output (first 8 la x all lb with hits)\n");
  for (int la = 0; la < 64; la += 9) {
    for (int lb = 0; lb < 64; lb++) {
      for (int q = 0; q < 64; q++) { ha[q] = 0; hb[q] = 0; }
      ha[la] = 1.0; hb[lb] = 1.0;
      hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
      layout16<<<1, 64>>>(a, b, d);
      hipMemcpy(hd, d, 2048, hipMemcpyDeviceToHost);
      for (int q = 0; q < 256; q++) if (hd[q] != 0.0) printf("  A lane %2d x B lane %2d -> D reg %d lane %2d\n", la, lb, q / 64, q % 64);
    }
  }
  printf("== issue cost, one wave per SIMD (1024 blocks of 64)\n");
  run<8, 0, 4>("mfma 4x4x4 only", 1024, 64);
  run<0, 16, 4>("dfma only", 1024, 64);
  run<8, 16, 4>("mfma 4x4x4 + dfma", 1024, 64);
  run<8, 32, 4>("mfma 4x4x4 + 2x dfma", 1024, 64);
  run<4, 0, 16>("mfma 16x16x4 only", 1024, 64);
  run<4, 16, 16>("mfma 16x16x4 + dfma", 1024, 64);
  run<4, 64, 16>("mfma 16x16x4 + 4x dfma", 1024, 64);
  printf("== two waves per SIMD (1024 blocks of 128)\n");
  run<8, 0, 4>("mfma 4x4x4 only", 1024, 128);
  run<0, 16, 4>("dfma only", 1024, 128);
  run<8, 16, 4>("mfma 4x4x4 + dfma", 1024, 128);
  run<4, 64, 16>("mfma 16x16x4 + 4x dfma", 1024, 128);
  return 0;
}
