// Does packed float32 math (v_pk_fma_f32) raise VALU throughput on gfx950 when no MFMA runs beside it?
// Two kernels with the same number of float32 FMAs per lane: 16 independent chains of v_fma_f32, and the same chains
// paired into 8 chains of v_pk_fma_f32.  Launched at 1, 2 and 4 waves per SIMD (blocks of 256 threads, 1 / 2 / 4 per CU
// by an LDS pad).  Prints ns per launch and the FMA rate.   hipcc -O3 --offload-arch=gfx950 -o pk_fma_probe pk_fma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PAD> __global__ __launch_bounds__(256) void scalar_fma(float* out, int iters, float a, float b) {
  extern __shared__ float pad[];
  float x[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 1e-3f + j;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = __builtin_fmaf(x[j], a, b);
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += x[j];
  if (s == 12345.678f) out[0] = s + pad[0];
}

template <int PAD> __global__ __launch_bounds__(256) void packed_fma(float* out, int iters, float a, float b) {
  extern __shared__ float pad[];
  f2 x[8];
  const f2 av = {a, a}, bv = {b, b};
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = f2{threadIdx.x * 1e-3f + 2 * j, threadIdx.x * 1e-3f + 2 * j + 1};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __builtin_elementwise_fma(x[j], av, bv);
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += x[j].x + x[j].y;
  if (s == 12345.678f) out[0] = s + pad[0];
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int per_cu : {1, 2, 4}) {
    const size_t lds = per_cu == 1 ? 160 * 1024 - 256 : (per_cu == 2 ? 80 * 1024 - 256 : 40 * 1024 - 256);
    hipFuncSetAttribute((const void*)scalar_fma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)packed_fma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int which = 0; which < 2; ++which) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(scalar_fma<0>, dim3(256 * per_cu), dim3(256), lds, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(packed_fma<0>, dim3(256 * per_cu), dim3(256), lds, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const double fmas = 256.0 * per_cu * 256 * 16.0 * iters;
      printf("%s  waves/SIMD %d  %.3f ms  %.1f TFLOP/s (2 flop per FMA)\n", which ? "v_pk_fma_f32" : "v_fma_f32   ", per_cu, best,
             2 * fmas / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
