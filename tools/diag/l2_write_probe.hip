// Does gfx950's L2 absorb repeated stores to a small, L2-resident global buffer (write-back), or does every store reach the
// fabric?  A persistent-style kernel rewrites an N-byte buffer `passes` times: (a) all 64 lanes store 8 bytes (whole 128-byte
// lines), (b) 44 of 64 lanes store (partial lines, like the correction-pair ring's global part), each from hipMalloc and from
// hipMallocAsync memory.  Run under rocprofv3 --pmc WRITE_SIZE (KiB at the L2 -> fabric interface):
//   write-back: WRITE_SIZE ~ N once;  write-through: ~ N x passes.
//   hipcc -O3 --offload-arch=gfx950 -o l2_write_probe l2_write_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool PARTIAL, bool NT = false> __global__ __launch_bounds__(64) void rewrite(double* buf, int slots, int passes, int spin, const double* big = nullptr, size_t big_n = 0) {
  // one wave per workgroup, its own `slots` x 64 doubles (like a wave's [ring slot][lane] block)
  double* mine = buf + (size_t)blockIdx.x * slots * 64;
  const int lane = threadIdx.x;
  double v = lane;
  for (int p = 0; p < passes; ++p) {
    for (int s = 0; s < slots; ++s) {
      v = v * 1.0000001 + 1.0;
      if (!PARTIAL || ((lane * 11 + p) & 15) < 11) mine[s * 64 + lane] = v;   // PARTIAL: ~44 of 64 lanes, changing per pass
    }
    // read one slot back (as build_b() does), so that the compiler cannot drop the stores and the lines stay in use
    v += mine[((p * 7) % slots) * 64 + lane] * 1e-30;
    if (big) {  // a read stream through the same L2 beside the rewrites: 4 KiB per wave and pass, every line once
      const size_t at = (((size_t)blockIdx.x * passes + p) * 8) * 64 % (big_n - 8 * 64);
      for (int k = 0; k < 8; ++k) v += (NT ? __builtin_nontemporal_load(&big[at + k * 64 + lane]) : big[at + k * 64 + lane]) * 1e-30;
    }
    for (int k = 0; k < spin; ++k) v = __builtin_fma(v, 1.0000001, 1e-9);  // time between two rewrites of a line (dependent chain: ~8 cycles a step)
  }
  if (v == 12345.678) buf[0] = v;
}

int main() {
  const int waves = 2048, slots = 10;
  const size_t bytes = (size_t)waves * slots * 64 * sizeof(double);  // 10 MiB, as the ring's global part
  double *a, *b;
  hipMalloc(&a, bytes);
  hipStream_t st;
  hipStreamCreate(&st);
  hipMallocAsync((void**)&b, bytes, st);
  hipMemsetAsync(a, 0, bytes, st);
  hipMemsetAsync(b, 0, bytes, st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // launches in this order (the PMC rows come out in the same order): full/partial x hipMalloc/hipMallocAsync back to back, then
  // partial stores with more and more time between two rewrites of a line
  const int passes = 200;
  hipLaunchKernelGGL(rewrite<false>, dim3(waves), dim3(64), 0, st, a, slots, passes, 0);
  hipLaunchKernelGGL(rewrite<true>, dim3(waves), dim3(64), 0, st, a, slots, passes, 0);
  hipLaunchKernelGGL(rewrite<false>, dim3(waves), dim3(64), 0, st, b, slots, passes, 0);
  hipLaunchKernelGGL(rewrite<true>, dim3(waves), dim3(64), 0, st, b, slots, passes, 0);
  hipStreamSynchronize(st);
  printf("buffer %.1f MiB, %d passes: write-back would move %.1f MiB per launch, write-through %.0f MiB (full) / ~%.0f MiB (partial)\n",
         bytes / 1048576.0, passes, bytes / 1048576.0, bytes / 1048576.0 * passes, bytes / 1048576.0 * passes * 11 / 16);
  for (int spin : {0, 300, 3000, 30000}) {
    const int p2 = 60;
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(rewrite<true>, dim3(waves), dim3(64), 0, st, b, slots, p2, spin);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("partial stores, %d passes, spin %d: %.3f ms = %.1f us between two rewrites of a line; write-through would be ~%.0f MiB\n", p2, spin, ms,
           ms * 1e3 / p2, bytes / 1048576.0 * p2 * 11 / 16);
  }
  // the same with a read stream through the L2 beside it (0.5 GiB per launch)
  double* big;
  const size_t big_n = (size_t)1 << 27;  // 1 GiB of doubles
  hipMalloc(&big, big_n * sizeof(double));
  hipMemsetAsync(big, 0, big_n * sizeof(double), st);
  for (int spin : {300, 3000}) {
    hipLaunchKernelGGL(rewrite<true>, dim3(waves), dim3(64), 0, st, b, slots, 60, spin, big, big_n);
    hipStreamSynchronize(st);
    printf("partial stores + read stream, 60 passes, spin %d\n", spin);
  }
  for (int spin : {300, 3000}) {  // the stream read with the non-temporal hint
    hipLaunchKernelGGL((rewrite<true, true>), dim3(waves), dim3(64), 0, st, b, slots, 60, spin, big, big_n);
    hipStreamSynchronize(st);
    printf("partial stores + non-temporal read stream, 60 passes, spin %d\n", spin);
  }
  return 0;
}
