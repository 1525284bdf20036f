// hip_shim.h -- TEST-ONLY host simulation of the small HIP subset the FermiFlow kernels use.
//
// Lets tests/hostsim/ compile fermiflow_amd/csrc/*.hip with g++ (-DFF_HOSTSIM) so the kernel sources can be
// exercised in the GPU-less build container: every workgroup runs as blockDim.x host threads with a real
// barrier for __syncthreads(); workgroups run one after another; "device" pointers are host pointers.
// Never part of the product: fermiflow_amd loads only the hipcc-built library and needs a GPU.
#pragma once
#include <pthread.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

// ---- the FF_* macros of ff_common.h / ff_ode.h / ff_cnf_adj.hip that carry an #ifndef (host forms)
#define FF_RBLOCK(n) 4                                       // tiny reduction workgroups: thread creation is what costs here
#define FF_OPAQUE(x) asm volatile("" : "+m"(x))              // an opaque value of any type ("+v" is a GPU register class)
#define FF_TAB_GRID 16                                       // (radial table build: 16 x 128 host threads per call)
#define FF_DEPR_EX 2
#define FF_DEPR_TY 2
#ifdef FF_HOSTSIM_TRACE      // FF_TRACE_STEPS=1 prints every accept / reject decision of ff_stepper
#define FF_STEP_TRACE(t, h, err, acc) do { if (getenv("FF_TRACE_STEPS")) fprintf(stderr, "step t=%.6f h=%.3e err=%.3e %s\n", t, h, err, (acc) ? "acc" : "REJ"); } while (0)
#endif

struct ff_sim_dim3 { unsigned x = 1, y = 1, z = 1; };
inline thread_local ff_sim_dim3 threadIdx, blockIdx, blockDim, gridDim;
inline pthread_barrier_t* ff_sim_bar = nullptr;
// workgroups of several INDEPENDENT waves (the two-wave tabulated adjoint): one barrier per wave of 64 "lanes"
inline pthread_barrier_t ff_sim_wave_bar[16];
#define FF_WAVE_SYNC() pthread_barrier_wait(&ff_sim_wave_bar[threadIdx.x >> 6])
#define FF_ASSUME(x) do { } while (0)
#define FF_LANE_SELF() ((int)(threadIdx.x & 63))
#define FF_WAVE_ORDER() FF_WAVE_SYNC()
#define FF_WG1_SYNC() pthread_barrier_wait(ff_sim_bar)      // (single-wave workgroups: the workgroup barrier IS the wave barrier)
#define FF_LOAD_ORDER() std::atomic_thread_fence(std::memory_order_seq_cst)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __constant__ static const
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __syncthreads() pthread_barrier_wait(ff_sim_bar)

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline int hipGetLastError() { return 0; }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_AGENT 1
inline unsigned long long atomicExch(unsigned long long* p, unsigned long long v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, sizeof d); return d; }
inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long cmp, unsigned long long val) {
  __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;      // (the value found: == the expected one exactly when the exchange happened)
}
template <class T> inline T __hip_atomic_load(T* p, int, int) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
template <class T, class V> inline void __hip_atomic_store(T* p, V v, int, int) { __atomic_store_n(p, (T)v, __ATOMIC_SEQ_CST); }
inline const char* hipGetErrorString(int) { return "hostsim"; }
inline int hipMemsetAsync(void* p, int v, size_t n, void*) { memset(p, v, n); return 0; }
#define hipMemcpyDeviceToDevice 3
inline int hipMemcpyAsync(void* dst, const void* src, size_t n, int, void*) { memcpy(dst, src, n); return 0; }
#define hipDeviceAttributeMultiprocessorCount 0
// (launches run to completion inside hipLaunchKernelGGL: streams and events have nothing to order)
typedef void* hipEvent_t;
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
inline int hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static int one; *s = &one; return 0; }
inline int hipEventCreateWithFlags(hipEvent_t* e, unsigned) { static int one; *e = &one; return 0; }
inline int hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline int hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline int hipGetDevice(int* d) { *d = 0; return 0; }
inline int hipSetDevice(int) { return 0; }
inline int hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline int hipEventDestroy(hipEvent_t) { return 0; }
inline int hipStreamDestroy(hipStream_t) { return 0; }
inline int hipStreamSynchronize(hipStream_t) { return 0; }
inline int hipDeviceGetAttribute(int* v, int, int) { *v = 2; return 0; }   // the simulator pretends to have two CUs

template <class T> inline T ff_sim_atomic_add(T* p, T v) {
  std::atomic_ref<T> a(*p);
  T old = a.load();
  while (!a.compare_exchange_weak(old, old + v)) {}
  return old;
}
inline double atomicAdd(double* p, double v) { return ff_sim_atomic_add(p, v); }
inline int atomicAdd(int* p, int v) { return ff_sim_atomic_add(p, v); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return ff_sim_atomic_add(p, v); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return ff_sim_atomic_add(p, v); }
inline int atomicOr(int* p, int v) { std::atomic_ref<int> a(*p); return a.fetch_or(v); }
inline int atomicMax(int* p, int v) {
  std::atomic_ref<int> a(*p);
  int old = a.load();
  while (old < v && !a.compare_exchange_weak(old, v)) {}
  return old;
}
inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
struct double2 { double x, y; };
inline int __double2hiint(double d) { long long b; memcpy(&b, &d, 8); return (int)(b >> 32); }
inline int __double2loint(double d) { long long b; memcpy(&b, &d, 8); return (int)(b & 0xffffffffll); }
inline double __hiloint2double(int hi, int lo) { long long b = ((long long)hi << 32) | (unsigned int)lo; double d; memcpy(&d, &b, 8); return d; }
inline void sincospi(double x, double* s, double* c) { *s = sin(M_PI * x); *c = cos(M_PI * x); }

// ---- gfx950 builtins, emulated.  Cross-lane ones go through a static buffer and the workgroup barrier: EVERY lane of the
// workgroup must call them (the kernels call them in workgroup-uniform control flow).
template <class T> inline T __builtin_amdgcn_readfirstlane(T x) { return x; }      // (callers pass wave-uniform values)
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
inline double __builtin_amdgcn_rsq(double x) { return 1.0 / std::sqrt(x); }
inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
inline float __builtin_amdgcn_logf(float x) { return log2f(x); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_sleep(int) {}
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
inline float __builtin_amdgcn_sinf(float x) { return (float)sin(6.283185307179586 * (double)x); }    // v_sin_f32: argument in revolutions
inline float __builtin_amdgcn_cosf(float x) { return (float)cos(6.283185307179586 * (double)x); }
inline unsigned long long wall_clock64() {      // 100 MHz, as the GPU's constant clock
  return (unsigned long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10);
}
#define __expf expf
static int ff_sim_xlane_buf[1024];
// v_mov_b32 with a DPP quad permutation (ctrl < 0x100: two bits per lane of the quad).  Cross-lane operations are WAVE-level on the
// hardware: the emulation synchronises the 64 threads of the caller's wave only (every lane of that wave must call).
inline int __builtin_amdgcn_mov_dpp(int v, int ctrl, int, int, bool) {
  const unsigned t = threadIdx.x;
  ff_sim_xlane_buf[t] = v;
  FF_WAVE_SYNC();
  const int o = (ctrl >= 0x121 && ctrl <= 0x12F) ? ff_sim_xlane_buf[(t & ~15u) | ((t - (unsigned)(ctrl - 0x120)) & 15)]      // row_ror:n within 16 lanes
                                                 : ff_sim_xlane_buf[(t & ~3u) | ((ctrl >> (2 * (t & 3))) & 3)];
  FF_WAVE_SYNC();
  return o;
}
// ds_bpermute_b32: lane (addr / 4) % 64 of the caller's own wave
inline int __builtin_amdgcn_ds_bpermute(int addr, int v) {
  const unsigned t = threadIdx.x;
  ff_sim_xlane_buf[t] = v;
  FF_WAVE_SYNC();
  const int o = ff_sim_xlane_buf[(t & ~63u) | ((addr >> 2) & 63)];
  FF_WAVE_SYNC();
  return o;
}
// wave vote (the kernels only ask whether any lane of a single-wave workgroup voted yes)
static std::atomic<unsigned long long> ff_sim_vote{0};
// the same vote among the 64 lanes of the caller's own wave only (waves of a workgroup that do not run in lockstep)
static std::atomic<unsigned long long> ff_sim_wave_vote[16];
inline unsigned long long ff_wave_ballot(bool pred) {
  const unsigned w = threadIdx.x >> 6;
  FF_WAVE_SYNC();
  if ((threadIdx.x & 63) == 0) ff_sim_wave_vote[w].store(0);
  FF_WAVE_SYNC();
  if (pred) ff_sim_wave_vote[w].fetch_or(1ULL << (threadIdx.x & 63));      // (a real lane mask, as on the GPU)
  FF_WAVE_SYNC();
  return (unsigned long long)ff_sim_wave_vote[w].load();
}
#define FF_HAVE_WAVE_BALLOT
inline unsigned long long __ballot(bool pred) {
  __syncthreads();
  if (threadIdx.x == 0) ff_sim_vote.store(0);
  __syncthreads();
  if (pred) ff_sim_vote.fetch_or(1ULL << (threadIdx.x & 63));
  __syncthreads();
  return (unsigned long long)ff_sim_vote.load();
}
// v_mfma_f64_4x4x4_4b_f64 (one f64 per lane and operand; block = (lane / 4) % 4, ff_common.h)
static double ff_sim_mfma_a[64], ff_sim_mfma_b[64];
inline double __builtin_amdgcn_mfma_f64_4x4x4f64(double a, double b, double c, int, int, int) {
  const int l = threadIdx.x;
  ff_sim_mfma_a[l] = a; ff_sim_mfma_b[l] = b;
  __syncthreads();
  const int i = l / 16, blk = (l / 4) % 4, j = l % 4;   // this lane holds D_blk[i][j]
  double acc = c;
  for (int k = 0; k < 4; k++) acc = __builtin_fma(ff_sim_mfma_a[16 * k + 4 * blk + i], ff_sim_mfma_b[16 * k + 4 * blk + j], acc);
  __syncthreads();
  return acc;
}

// v_mfma_f64_16x16x4_f64 per wave of a (possibly multi-wave) workgroup; every lane of the workgroup must call it
typedef double ff_sim_d4 __attribute__((vector_size(32)));
static double ff_sim_mfma16_a[1024], ff_sim_mfma16_b[1024];
inline ff_sim_d4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, ff_sim_d4 c, int, int, int) {
  const int t = threadIdx.x, w0 = t & ~63, l = t & 63;
  ff_sim_mfma16_a[t] = a; ff_sim_mfma16_b[t] = b;
  __syncthreads();
  for (int v = 0; v < 4; v++) {
    const int i = 4 * v + l / 16, j = l % 16;
    double acc = c[v];
    for (int k = 0; k < 4; k++) acc = __builtin_fma(ff_sim_mfma16_a[w0 + 16 * k + i], ff_sim_mfma16_b[w0 + 16 * k + j], acc);
    c[v] = acc;
  }
  __syncthreads();
  return c;
}

// v_mfma_f32_16x16x4_f32: register v of lane l holds C/D[4 (l / 16) + v][l % 16] (NOT the fp64 form's 4 v + l / 16)
typedef float ff_sim_f4 __attribute__((vector_size(16)));
static float ff_sim_mfma16f_a[1024], ff_sim_mfma16f_b[1024];
inline ff_sim_f4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, ff_sim_f4 c, int, int, int) {
  const int t = threadIdx.x, w0 = t & ~63, l = t & 63;
  ff_sim_mfma16f_a[t] = a; ff_sim_mfma16f_b[t] = b;
  __syncthreads();
  for (int v = 0; v < 4; v++) {
    const int i = 4 * (l / 16) + v, j = l % 16;
    float acc = c[v];
    for (int k = 0; k < 4; k++) acc = __builtin_fmaf(ff_sim_mfma16f_a[w0 + 16 * k + i], ff_sim_mfma16f_b[w0 + 16 * k + j], acc);
    c[v] = acc;
  }
  __syncthreads();
  return c;
}
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }

template <class K, class... A>
inline void ff_sim_launch(K kernel, unsigned grid, unsigned block, A... args) {
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, nullptr, block);
  ff_sim_bar = &bar;
  const unsigned nwave = (block + 63) / 64;
  for (unsigned w = 0; w < nwave && w < 16; w++) pthread_barrier_init(&ff_sim_wave_bar[w], nullptr, (block - 64 * w) < 64 ? (block - 64 * w) : 64);
  for (unsigned b = 0; b < grid; b++) {
    std::vector<std::thread> th;
    th.reserve(block);
    for (unsigned t = 0; t < block; t++)
      th.emplace_back([=]() {
        threadIdx.x = t; blockIdx.x = b; blockDim.x = block; gridDim.x = grid;
        kernel(args...);
      });
    for (auto& x : th) x.join();
  }
  pthread_barrier_destroy(&bar);
  for (unsigned w = 0; w < nwave && w < 16; w++) pthread_barrier_destroy(&ff_sim_wave_bar[w]);
  ff_sim_bar = nullptr;
}
#define FF_LAUNCH(kernel, grid, block, stream, ...) ff_sim_launch(kernel, (unsigned)(grid), (unsigned)(block), __VA_ARGS__)
// dynamic LDS: one static buffer (workgroups run one after the other)
static double ff_sim_dyn_lds[20480];
#define FF_LAUNCH_LDS(kernel, grid, block, lds_bytes, stream, ...) \
  do { if ((size_t)(lds_bytes) > sizeof(ff_sim_dyn_lds)) abort(); ff_sim_launch(kernel, (unsigned)(grid), (unsigned)(block), __VA_ARGS__); } while (0)
#define FF_DYN_LDS(name) double* const name = ff_sim_dyn_lds
