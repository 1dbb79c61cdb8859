// Lab: latency of a step-to-step hand-off between resident workgroups (the GRU sequence kernels of csrc/train_gru.hip).
// G workgroups, each step: every workgroup stages the previous step's whole vector (WORDS floats written by all workgroups) into
// LDS, polling for a sentinel, then stores its own slice of the next vector.  Modes:
//   placement 0: G workgroups launched as they come (round-robin over the 8 XCDs)
//   placement 1: 8 * G workgroups launched, only those whose XCC_ID equals that of workgroup 0 ... (simply blockIdx % 8 == 0) work
//   loads  0: agent-scope (sc1) loads      1: sc0 loads (L1 miss, L2 hit allowed)
//   stores 0: agent-scope (sc1) stores     1: plain stores
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/seq_lab.hip -o tools/bin/seq_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr unsigned SENT = 0xffffffffu;

template <int LD, int ST>
__global__ __launch_bounds__(512) void chain_kernel(float* buf, int T, int words, int G, int stride_wg, unsigned* xcc_out, unsigned* err,
                                                    long long* cyc) {
  extern __shared__ float sh[];
  if (blockIdx.x % stride_wg) return;
  const int wg = blockIdx.x / stride_wg, tid = threadIdx.x;
  if (tid == 0) xcc_out[wg] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
  const int per = words / G;                    // floats this workgroup writes per step
  const long long t0 = wall_clock64();
  bool dead = false;                            // after one timed-out wait: no more waiting (the run is reported as failed)
  for (int t = 1; t <= T; t++) {
    const float* src = buf + (size_t)(t - 1) * words;
    for (int i = tid; i < words; i += 512) {
      unsigned v; unsigned spins = 0;
      for (;;) {
        if (LD == 0) v = __hip_atomic_load((gu32*)src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src + i) : "memory");
        if (v != SENT || dead) break;
        if (++spins > (1u << 18)) { *err = 1; dead = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      sh[i] = __uint_as_float(v);
    }
    __syncthreads();
    if (tid < per) {
      float acc = 0.f;
      for (int k = 0; k < 16; k++) acc += sh[(tid * 16 + k) % words];
      float* dst = buf + (size_t)t * words + wg * per + tid;
      const float val = acc * 0.001f + 1.f;
      if (ST == 0) __hip_atomic_store((gu32*)dst, __float_as_uint(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *dst = val;
    }
    __syncthreads();
  }
  if (tid == 0) cyc[wg] = wall_clock64() - t0;
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 64, words = argc > 2 ? atoi(argv[2]) : 4096, T = 150;
  float* buf; CK(hipMalloc((void**)&buf, (size_t)(T + 1) * words * 4));
  unsigned *xcc, *err; long long* cyc;
  CK(hipMalloc((void**)&xcc, 4096)); CK(hipMalloc((void**)&err, 4)); CK(hipMalloc((void**)&cyc, 8 * 1024));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int place = 0; place < 2; place++)
    for (int ld = 0; ld < 2; ld++)
      for (int st = 0; st < 2; st++) {
        const int stride = place ? 8 : 1;
        float best = 1e9f; unsigned herr = 0; std::vector<unsigned> hx(G);
        for (int it = 0; it < 4; it++) {
          CK(hipMemset(buf, 0xff, (size_t)(T + 1) * words * 4));
          CK(hipMemset(buf, 0, (size_t)words * 4));
          CK(hipMemset(err, 0, 4));
          CK(hipDeviceSynchronize());
          hipEventRecord(e0, 0);
          auto k = ld == 0 ? (st == 0 ? chain_kernel<0, 0> : chain_kernel<0, 1>) : (st == 0 ? chain_kernel<1, 0> : chain_kernel<1, 1>);
          hipLaunchKernelGGL(k, dim3(G * stride), dim3(512), words * 4, 0, buf, T, words, G, stride, xcc, err, cyc);
          hipEventRecord(e1, 0); CK(hipDeviceSynchronize());
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
          unsigned e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost)); herr |= e;
        }
        CK(hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost));
        unsigned mask = 0; for (int i = 0; i < G; i++) mask |= 1u << (hx[i] & 15);
        printf("G=%d words=%d placement=%d loads=%s stores=%s: %.2f us per step%s  (XCC ids seen: mask 0x%x)\n", G, words, place,
               ld ? "sc0" : "sc1", st ? "plain" : "sc1", best * 1000.f / T, herr ? "  ** TIMED OUT (stale data) **" : "", mask);
      }
  return 0;
}
