// Issue cost of the float64 vector instructions the lane solvers are made of, on gfx950: per instruction kind, 16 independent
// chains per lane (no dependency stalls), one wave per SIMD (256-thread blocks, one per CU by an LDS pad) and two waves per SIMD.
// Prints cycles per wave-instruction at the 2.4 GHz the guide quotes.   hipcc -O3 --offload-arch=gfx950 -o f64_issue_probe f64_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

enum { FMA, MUL, ADD, RCP, RSQ, CND, MOV, CVT, LDEXP, FMA_AS_MUL, FMA_AS_ADD, KINDS };
static const char* kNames[KINDS] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64", "v_cndmask_b32", "v_mov_b64",
                                    "v_cvt_f64_f32+back", "v_ldexp_f64", "v_fma_f64 as x*a (+ -0)", "v_fma_f64 as x+b (x * 1)"};

template <int KIND> __global__ __launch_bounds__(256) void probe(double* out, int iters, double a, double b) {
  extern __shared__ double pad[];
  double x[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) x[j] = 1.0 + threadIdx.x * 1e-3 + j;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if constexpr (KIND == FMA) x[j] = __builtin_fma(x[j], a, b);
      else if constexpr (KIND == MUL) x[j] = x[j] * a;
      else if constexpr (KIND == ADD) x[j] = x[j] + b;
      else if constexpr (KIND == RCP) x[j] = __builtin_amdgcn_rcp(x[j]);
      else if constexpr (KIND == RSQ) x[j] = __builtin_amdgcn_rsq(x[j]);
      else if constexpr (KIND == CND) {  // one v_cndmask_b32 on the low half (a float64 select is two of them)
        unsigned lo = (unsigned)__double2loint(x[j]), lo2 = (unsigned)__double2loint(x[(j + 1) & 15]), t;
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(t) : "v"(lo), "v"(lo2));
        x[j] = __hiloint2double(__double2hiint(x[j]), (int)t);
      }
      else if constexpr (KIND == MOV) { double t; asm volatile("v_mov_b64 %0, %1" : "=v"(t) : "v"(x[j])); x[j] = t; }
      else if constexpr (KIND == CVT) x[j] = (double)(float)x[j];
      else if constexpr (KIND == LDEXP) x[j] = __builtin_ldexp(x[j], (int)(i & 1));
      else if constexpr (KIND == FMA_AS_MUL) x[j] = __builtin_fma(x[j], a, -0.0);
      else if constexpr (KIND == FMA_AS_ADD) x[j] = __builtin_fma(x[j], 1.0, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += x[j];
  if (s == 12345.678) out[0] = s + pad[0];
}

template <int KIND> void run(double* out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 40000;
  for (int per_cu : {1, 2}) {
    const size_t lds = per_cu == 1 ? 160 * 1024 - 256 : 80 * 1024 - 256;
    hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe<KIND>, dim3(256 * per_cu), dim3(256), lds, 0, out, iters, 1.00000001, 1e-9);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    // one wave per SIMD issues iters * 16 instructions (CVT: two per step)
    const double per = KIND == CVT ? 2.0 : 1.0;
    const double cycles = best * 1e-3 * 2.4e9 / (iters * 16.0 * per * per_cu);
    printf("%-22s %d wave(s) per SIMD: %7.3f ms, %5.2f cycles per wave-instruction (SIMD time)\n", kNames[KIND], per_cu, best, cycles);
  }
}

int main() {
  double* out;
  hipMalloc(&out, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  run<FMA>(out, e0, e1); run<MUL>(out, e0, e1); run<ADD>(out, e0, e1); run<RCP>(out, e0, e1); run<RSQ>(out, e0, e1);
  run<CND>(out, e0, e1); run<MOV>(out, e0, e1); run<CVT>(out, e0, e1); run<LDEXP>(out, e0, e1);
  run<FMA_AS_MUL>(out, e0, e1); run<FMA_AS_ADD>(out, e0, e1); run<FMA>(out, e0, e1);
  return 0;
}
