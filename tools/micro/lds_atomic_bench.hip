// Microbenchmark (gfx950): cost of ds_add_f64 by number of active lanes and address pattern, beside ds_read_b128 and
// ds_bpermute_b32, per wave-instruction, with 1 or 4 waves per SIMD issuing.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIters = 2000;

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int active, int stride, int same) {
  __shared__ double s[4 * 64 * 9 + 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 64 * 9 + 64; i += 256) s[i] = 0.0;
  __syncthreads();
  double* base = s + (same ? 0 : wv * 64 * 9);
  const bool on = (lane % (64 / active)) == 0;
  const int slot = same == 2 ? 0 : (lane * stride) % 64;
  double v = 1.0 + lane;
  double acc = 0.0;
  const long long t0 = clock64();
  for (int it = 0; it < kIters; ++it) {
    if (MODE == 0) {
      if (on) {
#pragma unroll
        for (int k9 = 0; k9 < 9; ++k9) atomicAdd(base + 9 * slot + k9, v);
      }
    } else if (MODE == 1) {
      const double2* p = reinterpret_cast<const double2*>(s) + (lane * 13) % 128;
#pragma unroll
      for (int k9 = 0; k9 < 9; ++k9) { const double2 q = p[k9]; acc += q.x + q.y; }
    } else if (MODE == 2) {
#pragma unroll
      for (int k9 = 0; k9 < 9; ++k9) {
        const int lo = __builtin_amdgcn_ds_bpermute(((lane + 5) & 63) << 2, __double2loint(v));
        v += lo;
      }
    }
    asm volatile("" ::: "memory");
  }
  __builtin_amdgcn_s_waitcnt(0);
  const long long t1 = clock64();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = acc + v + s[threadIdx.x];
}

template <int MODE>
double run(int blocks_per_cu, int active, int stride, int same) {
  const int nb = 256 * blocks_per_cu;
  double* out; long long* cyc;
  hipMalloc(&out, nb * 256 * sizeof(double));
  hipMalloc(&cyc, nb * sizeof(long long));
  hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, out, cyc, active, stride, same);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, out, cyc, active, stride, same);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(nb);
  hipMemcpy(h.data(), cyc, nb * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0; for (auto c : h) mean += (double)c; mean /= nb;
  hipFree(out); hipFree(cyc);
  // per CU: blocks_per_cu * 4 waves * kIters * 9 instructions in `ms`; LDS cycles per wave-instruction at 2.4 GHz if the LDS pipe is the limit
  const double per_inst_cu_cycles = ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * 4 * kIters * 9);
  printf("mode %d blocks/CU %d active %2d stride %2d same %d: %.3f ms, wave clock64 per inst %.1f, CU-cycles(2.4GHz) per wave-inst %.2f\n", MODE,
         blocks_per_cu, active, stride, same, ms, mean / (kIters * 9.0), per_inst_cu_cycles);
  return ms;
}

int main() {
  for (int bpc : {1, 4}) {
    for (int active : {64, 32, 16, 8, 4}) run<0>(bpc, active, 1, 0);
    run<0>(bpc, 64, 1, 1);    // four waves share one accumulator copy, distinct slots per lane
    run<0>(bpc, 64, 2, 0);    // pairs of lanes on the same address
    run<0>(bpc, 64, 4, 0);    // four lanes per address
    run<0>(bpc, 64, 16, 0);   // sixteen lanes per address
    run<0>(bpc, 64, 1, 2);    // every lane and wave on one address
    run<1>(bpc, 64, 1, 0);
    run<2>(bpc, 64, 1, 0);
  }
  return 0;
}
