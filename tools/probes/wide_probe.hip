// Probe (not part of the product): (1) a kernel with > 64 KB of static LDS launches on gfx950; (2) lane layouts of
// v_mfma_f32_32x32x2f32 / v_mfma_f32_16x16x4f32 / v_mfma_f64_16x16x4f64 with a vector_size operand type; (3) issue cost.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double d4 __attribute__((vector_size(32)));
typedef float f4 __attribute__((vector_size(16)));
typedef float f16v __attribute__((vector_size(64)));

__global__ void __launch_bounds__(256) big_lds(double* out) {
  __shared__ double s[17000];   // 136 KB
  for (int e = threadIdx.x; e < 17000; e += 256) s[e] = e;
  __syncthreads();
  double t = 0;
  for (int e = threadIdx.x; e < 17000; e += 256) t += s[16999 - e];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
__global__ void lay_f64(const double* a, const double* b, double* d) {
  int l = threadIdx.x; d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) d[r * 64 + l] = c[r];
}
__global__ void lay_f32_16(const float* a, const float* b, float* d) {
  int l = threadIdx.x; f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) d[r * 64 + l] = c[r];
}
__global__ void lay_f32_32(const float* a, const float* b, float* d) {
  int l = threadIdx.x; f16v c = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 16; r++) d[r * 64 + l] = c[r];
}
template <int WHICH>
__global__ void timing(float* out, long long* cyc, int iters) {
  int l = threadIdx.x;
  float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
  f4 m4[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  f16v m16[4]; for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) m16[q][r] = 0;
  d4 md[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (WHICH == 0) m4[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m4[q], 0, 0, 0);
      else if (WHICH == 1) m16[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, m16[q], 0, 0, 0);
      else md[q] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a, (double)b, md[q], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int q = 0; q < 4; q++) { for (int r = 0; r < 4; r++) s += m4[q][r] + (float)md[q][r]; for (int r = 0; r < 16; r++) s += m16[q][r]; }
  out[blockIdx.x * blockDim.x + l] = s;
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int WHICH> void run(const char* name) {
  float* out; long long* cyc; int blocks = 1024;
  hipMalloc(&out, 4 * blocks * 64); hipMalloc(&cyc, 8 * blocks);
  timing<WHICH><<<blocks, 64>>>(out, cyc, 2000); timing<WHICH><<<blocks, 64>>>(out, cyc, 2000);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s %7.1f ticks per 4 mfma (one wave per SIMD)\n", name, (double)h / 2000);
}
template <class TT, class K> void layout(const char* name, K kern, int nreg) {
  TT ha[64], hb[64]; std::vector<TT> hd(64 * nreg);
  TT *a, *b, *d; hipMalloc(&a, 64 * sizeof(TT)); hipMalloc(&b, 64 * sizeof(TT)); hipMalloc(&d, 64 * nreg * sizeof(TT));
  printf("== %s: one-hot A lane la x B lane lb -> (reg, lane)\n", name);
  // A lane -> (i, k); B lane -> (k, j): find for A lane la the set of B lanes that hit and where
  for (int la = 0; la < 64; la++) {
    int nh = 0, first_lb = -1, first_reg = -1, first_lane = -1, lb_stride = -1;
    for (int lb = 0; lb < 64; lb++) {
      for (int q = 0; q < 64; q++) { ha[q] = 0; hb[q] = 0; }
      ha[la] = 1; hb[lb] = 1;
      hipMemcpy(a, ha, 64 * sizeof(TT), hipMemcpyHostToDevice); hipMemcpy(b, hb, 64 * sizeof(TT), hipMemcpyHostToDevice);
      kern<<<1, 64>>>(a, b, d);
      hipMemcpy(hd.data(), d, 64 * nreg * sizeof(TT), hipMemcpyDeviceToHost);
      for (int q = 0; q < 64 * nreg; q++) if (hd[q] != 0) { if (nh == 0) { first_lb = lb; first_reg = q / 64; first_lane = q % 64; } if (nh == 1) lb_stride = lb - first_lb; nh++; }
    }
    printf("  A lane %2d: %2d B lanes hit, first lb %2d (stride %d) -> reg %2d lane %2d\n", la, nh, first_lb, lb_stride, first_reg, first_lane);
  }
  // and for A lane 0: the full list of (lb -> reg, lane)
  for (int lb = 0; lb < 64; lb++) {
    for (int q = 0; q < 64; q++) { ha[q] = 0; hb[q] = 0; }
    ha[0] = 1; hb[lb] = 1;
    hipMemcpy(a, ha, 64 * sizeof(TT), hipMemcpyHostToDevice); hipMemcpy(b, hb, 64 * sizeof(TT), hipMemcpyHostToDevice);
    kern<<<1, 64>>>(a, b, d);
    hipMemcpy(hd.data(), d, 64 * nreg * sizeof(TT), hipMemcpyDeviceToHost);
    for (int q = 0; q < 64 * nreg; q++) if (hd[q] != 0) printf("  A lane 0 x B lane %2d -> reg %2d lane %2d\n", lb, q / 64, q % 64);
  }
}
int main() {
  double* out; hipMalloc(&out, 8 * 256 * 4);
  big_lds<<<4, 256>>>(out);
  hipError_t e = hipDeviceSynchronize();
  double h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
  printf("136 KB static LDS kernel: %s, out[0] = %.1f (expect sum)\n", hipGetErrorString(e), h[0]);
  layout<double>("v_mfma_f64_16x16x4f64", lay_f64, 4);
  layout<float>("v_mfma_f32_16x16x4f32", lay_f32_16, 4);
  layout<float>("v_mfma_f32_32x32x2f32", lay_f32_32, 16);
  run<0>("mfma_f32_16x16x4f32"); run<1>("mfma_f32_32x32x2f32"); run<2>("mfma_f64_16x16x4f64");
  return 0;
}
