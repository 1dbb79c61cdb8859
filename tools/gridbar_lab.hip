// Grid-barrier microbenchmark: what does a device-wide barrier between dependent phases cost on 8 XCDs (256 CUs), with the
// phase outputs made visible across XCD L2s?  Variants: (a) agent-scope release/acquire fences (bulk L2 write-back / invalidate),
// (b) write-through (nontemporal / sc1) stores + sc1 loads and a relaxed barrier.
//   hipcc --offload-arch=gfx950 -O3 tools/gridbar_lab.hip -o tools/bin/gridbar_lab && tools/bin/gridbar_lab [wgs=256] [iters=200] [kb_per_wg=16]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// every phase: WG w writes `words` words of (iter, w) into its slab, barrier, reads the slab of WG (w + 37) % n and checks
template <int MODE>
__global__ __launch_bounds__(256) void phases(unsigned* data, unsigned* ctr, int words, int iters, unsigned* bad) {
  const int w = blockIdx.x, n = gridDim.x;
  unsigned errs = 0;
  for (int it = 0; it < iters; it++) {
    unsigned* mine = data + ((size_t)(it & 1) * n + w) * words;
    for (int i = threadIdx.x; i < words; i += 256) {
      const unsigned v = (unsigned)it * 65536u + (unsigned)w;
      if (MODE == 1) __builtin_nontemporal_store(v, mine + i);
      else mine[i] = v;
    }
    if (MODE == 0) __threadfence();
    grid_barrier(ctr, (unsigned)(it + 1) * n);
    const int o = (w + 37) % n;
    const unsigned* theirs = data + ((size_t)(it & 1) * n + o) * words;
    for (int i = threadIdx.x; i < words; i += 256) {
      const unsigned v = MODE == 1 ? __builtin_nontemporal_load(theirs + i) : __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      errs += v != (unsigned)it * 65536u + (unsigned)o;
    }
  }
  if (errs) atomicAdd(bad, errs);
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 200, kb = argc > 3 ? atoi(argv[3]) : 16;
  const int words = kb * 256;
  unsigned *data, *ctr, *bad;
  hipMalloc(&data, (size_t)2 * wgs * words * 4); hipMalloc(&ctr, 4); hipMalloc(&bad, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipMemset(ctr, 0, 4); hipMemset(bad, 0, 4);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(phases<0>, dim3(wgs), dim3(256), 0, 0, data, ctr, words, iters, bad);
      else hipLaunchKernelGGL(phases<1>, dim3(wgs), dim3(256), 0, 0, data, ctr, words, iters, bad);
      hipEventRecord(e1);
      if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
      if (rep) printf("mode %d (%s): %d WGs, %d KB per WG per phase: %.2f us per phase, %u stale words\n", mode,
                      mode ? "nontemporal stores/loads" : "plain stores + __threadfence, agent-scope loads", wgs, kb, ms * 1000 / iters, hb);
    }
  }
  return 0;
}
