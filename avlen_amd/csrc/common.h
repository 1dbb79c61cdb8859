// Shared device/host helpers for the avlen_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define AVLEN_OK 0
#define AVLEN_ERR_ARG 1
#define AVLEN_ERR_LAUNCH 2
#define AVLEN_ERR_WS 3

#define AVLEN_WAVE 64

static inline int avlen_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? AVLEN_OK : AVLEN_ERR_LAUNCH;
}

// Tuning decisions that were measured with environment-variable A/B switches are CONSTANTS in the shipped library;
// a lab build (-DAVLEN_LAB_KNOBS) restores the overrides for re-measuring.
#ifdef AVLEN_LAB_KNOBS
#include <stdlib.h>
static inline long avlen_knob(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }
#else
#define avlen_knob(name, dflt) (dflt)
#endif

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): `done` is the call site's own static bit mask.
// Returns AVLEN_OK or AVLEN_ERR_LAUNCH (a process may drive several devices; the attribute is per device).
static inline int avlen_set_dyn_lds(const void* fn, int bytes, unsigned long long* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return AVLEN_ERR_LAUNCH;
  if (*done & (1ull << dev)) return AVLEN_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return AVLEN_ERR_LAUNCH;
  *done |= 1ull << dev;
  return AVLEN_OK;
}
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// bump allocator over a caller-provided workspace (never allocates device memory itself)
struct WsBump {
  char* base; size_t off; size_t cap;
  WsBump(void* p, size_t c) : base((char*)p), off(0), cap(c) {}
  template <typename T> T* take(size_t n) {
    size_t o = align_up(off, 256);
    off = o + n * sizeof(T);
    return (T*)(base + o);          // base may be null in "measure" mode
  }
  bool ok() const { return base == nullptr || off <= cap; }
};

#ifdef __HIPCC__
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x <= 1024; `sh` must hold >= 16 floats; all threads get the result
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; i++) r += sh[i];
  return r;
}
#endif
