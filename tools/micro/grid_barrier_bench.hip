// Microbenchmark (gfx950): cost of a grid barrier on one device-scope counter (the barrier of pcg_tail, csrc/miba_pcg.hpp) by
// grid size, with and without the release / acquire fences, and of a 1 MB read after it (L2 state after the acquire).
// Build: hipcc -O3 --offload-arch=gfx950 -o grid_barrier_bench grid_barrier_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int FENCE>
__device__ __forceinline__ void gbar(unsigned* count, unsigned nb) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned old = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (old / nb + 1u) * nb;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (FENCE != 2) __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 200000000ll) break;
    }
  }
  __syncthreads();
  if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// iters barriers; between them every thread reads `reads` doubles (grid-strided over n) and writes one
template <int FENCE>
__global__ __launch_bounds__(1024) void k(unsigned* count, int iters, double* buf, size_t n, int reads, double* out) {
  double acc = 0.0;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
  for (int it = 0; it < iters; ++it) {
    for (int r = 0; r < reads; ++r) acc += buf[(gid + (size_t)r * gsz + it) % n];
    if (reads) buf[gid % n] = acc;
    gbar<FENCE>(count, gridDim.x);
  }
  if (acc == 123.456) out[0] = acc;
}

int main() {
  unsigned* count; double* buf; double* out;
  const size_t n = 4 << 20;   // 32 MB
  hipMalloc(&count, 4); hipMalloc(&buf, n * 8); hipMalloc(&out, 8);
  hipMemset(buf, 0, n * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 200;
  for (int fence = 0; fence < 3; ++fence)
    for (int reads : {0, 1, 8})
      for (int nb : {1, 16, 64, 128, 256}) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
          hipMemset(count, 0, 4);
          hipEventRecord(a);
          if (fence == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(1024), 0, 0, count, iters, buf, n, reads, out);
          else if (fence == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(1024), 0, 0, count, iters, buf, n, reads, out);
          else hipLaunchKernelGGL(k<2>, dim3(nb), dim3(1024), 0, 0, count, iters, buf, n, reads, out);
          hipEventRecord(b); hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b);
          if (ms < best) best = ms;
        }
        printf("fence %d (0 none, 1 fences + sleep, 2 fences, no sleep)  reads/thread %d  blocks %3d : %.2f us per barrier round\n", fence, reads, nb,
               1e3 * best / iters);
      }
  return 0;
}
