// t2fit_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the per-voxel T2 fit and the C ABI of
// include/t2fit.h.  Built with:  hipcc -O3 --offload-arch=gfx950 -shared -fPIC
//
// Mapping: one lane fits one voxel at a time, 256-thread workgroups (4 wave64).  The fit kernels
// are persistent: waves pull chunks of consecutive voxels from a global atomic counter and every
// lane that finishes a voxel takes the next one (fit_persistent_kernel below); the streaming
// kernels (closed-form log-linear fit, residual map, mask union, per-label statistics) use
// grid = ceil(N / tile) >> 256 CUs.
// HBM layout: echoes (nTE, N) float32, TE-major; every sample of a fitted voxel is read once and
// parked in LDS ([nTE][257] floats per workgroup, one column per lane, padded so the voxel-major
// staging transpose is conflict-free).  The solver re-reads its column from LDS on every objective
// evaluation instead of holding nTE samples in VGPRs (nTE is a run-time value).  Each map is
// written once.  Algorithmic HBM traffic per voxel: 4*nTE (samples) + 1 (mask) + 16 (t2, k, sigma,
// res) = 49 B at 8 TE.  The fit itself is ALU work (exp/sqrt/div/fma, fp64 or fp32); no MFMA:
// nothing here is a contraction.
#include <hip/hip_runtime.h>
#if defined(T2_WG_SHAPE_DIAG)
#include <array>
#include <map>
#endif

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "t2fit_config.h"
#include "t2fit_context.h"
#include "t2fit_dispatch.h"

using namespace t2fit;

namespace {

constexpr int kBlock = 256;
constexpr int kLdsStride = kBlock + 1;
bool g_use_persistent = true;  // LM: false selects the one-voxel-per-lane kernel (T2FIT_ONE_SHOT=1)
int g_refill_min = 0;           // > 0 overrides the per-solver refill batch (T2FIT_REFILL_MIN)
#ifndef T2_SAMPLE_LOAD
#define T2_SAMPLE_LOAD(p) (*(p))  // A/B: -DT2_SAMPLE_LOAD(p)=__builtin_nontemporal_load(p), profiles/r02_exp42_nt_loads.txt
#endif
#ifndef T2_WAVE_HINT
#define T2_WAVE_HINT 2
#endif
constexpr int kWaveHint = T2_WAVE_HINT;  // occupancy the register allocator / scheduler is told for the one-wave workgroups
int g_waves_per_cu = 0;         // T2FIT_WAVES_PER_CU: cap of the above (A/B runs)
int g_wave_wg = 1;         // one-wave workgroups for the large-volume L-BFGS-B kernels (T2FIT_WAVE_WG=0: 256-lane workgroups)
int g_persistent_blocks = 2048;  // grid of the persistent kernel (T2FIT_PERSISTENT_BLOCKS overrides)
// Process-wide tuning state.  Everything in this block except g_reserve_cus is written once, by the first launch of the
// process (the T2FIT_* environment switches, read under a function-local static's lock) and only read afterwards.
// g_reserve_cus can be set at any time from any thread (t2fit_set_reserve_cus) while other threads launch: atomic.
// None of it can change a result: the maps do not depend on how many workgroups fit a volume.
std::atomic<bool> g_reserve_set{false};  // t2fit_set_reserve_cus() was called: it wins over the environment
std::atomic<int> g_reserve_cus{0};       // T2FIT_RESERVE_CUS: CUs' worth of workgroups the L-BFGS-B kernel is launched short
bool g_nte_special = true;       // T2FIT_NTE_SPECIAL=0: always the generic-echo-count lane (A/B switch)
int g_park_min = 1;              // T2FIT_PARK_MIN: lanes of a wave that must be waiting for begin() before it runs (A/B switch)

thread_local std::string g_err;
thread_local bool g_timing = false;
// start / stop events of the last kEvRing timed launches (a caller that pipelines launches over several streams reads a
// launch's time a few launches later, when it is long done, instead of stalling on the one just queued)
constexpr int kEvRing = 16;
thread_local hipEvent_t g_ev0[kEvRing] = {}, g_ev1[kEvRing] = {}, g_ev2[kEvRing] = {};  // start, end of the fit kernel, end of the epilogue pass
thread_local long g_ev_count = 0;   // timed launches so far
thread_local int g_ev_slot = 0;     // ring slot of the launch being queued

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define T2_HIP(call)                                                                            \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(T2FIT_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));             \
  } while (0)

struct DevMaps {
  float *t2, *k, *sigma, *res, *r2, *fun, *se;
  int32_t* nit;
  uint8_t* status;
  double* xd;    // optional: float64 parameters, 3 per voxel (voxel seam)
  double* fund;  // optional: float64 objective value
  double* trace = nullptr;         // trace kernels only: [n_vox][trace_cap][4] doubles (k, T2, sigma, f)
  int32_t* trace_len = nullptr;    //                     iterations recorded per voxel
  int trace_cap = 0;
};

// Stage this workgroup's samples into LDS.  TE-major: each lane copies its own column (coalesced
// per plane, no exchange needed).  Voxel-major: the (256, nTE) tile is contiguous in HBM, so it is
// read linearly by the whole workgroup and transposed on the way into LDS.
__device__ __forceinline__ void stage_echoes(float* lds, const float* __restrict__ echoes, int layout,
                                             int n_te, int64_t n_vox, int64_t base, bool load_own) {
  const int lane = threadIdx.x;
  if (layout == T2FIT_LAYOUT_TE_MAJOR) {
    if (load_own) {
      const float* src = echoes + base + lane;
      for (int i = 0; i < n_te; ++i) lds[i * kLdsStride + lane] = src[(int64_t)i * n_vox];
    }
  } else {
    const int64_t rows = n_vox - base < kBlock ? n_vox - base : kBlock;
    const int total = (int)rows * n_te;
    const float* src = echoes + base * n_te;
    for (int j = lane; j < total; j += kBlock) {
      const int v = j / n_te;
      const int i = j - v * n_te;
      lds[i * kLdsStride + v] = src[j];
    }
    __syncthreads();
  }
}

// kExtras = false: the caller asked for the reference's four maps only; the optional outputs are not even tested
// for (their pointers would otherwise live in scalar registers across the whole persistent loop)
// kEpilogueFollows: the persistent fits are followed by residuals_kernel, which writes res / r2 / se of every voxel
// (zeros outside the mask): storing those zeros here as well would write the same lines twice
template <bool kExtras = true, bool kEpilogueFollows = false>
__device__ __forceinline__ void store_masked(const DevMaps& m, int64_t v) {
  // zeros outside the mask (run_t2mapping.py:415-418)
  m.t2[v] = 0.0f; m.k[v] = 0.0f; m.sigma[v] = 0.0f;
  if constexpr (!kEpilogueFollows) m.res[v] = 0.0f;
  if constexpr (!kExtras) return;
  if constexpr (!kEpilogueFollows) {
    if (m.r2) m.r2[v] = 0.0f;
    if (m.se) m.se[v] = 0.0f;
  }
  if (m.fun) m.fun[v] = 0.0f;
  if (m.nit) m.nit[v] = 0;
  if (m.status) m.status[v] = T2FIT_ST_MASKED;
  if (m.xd) { m.xd[3 * v] = 0.0; m.xd[3 * v + 1] = 0.0; m.xd[3 * v + 2] = 0.0; }
  if (m.fund) m.fund[v] = 0.0;
}

__device__ __forceinline__ void store_result(const DevMaps& m, int64_t v, const LaneOutputs& o, const LaneResult& r) {
  m.t2[v] = o.t2; m.k[v] = o.k; m.sigma[v] = o.sigma; m.res[v] = o.res;
  if (m.r2) m.r2[v] = o.r2;
  if (m.se) m.se[v] = o.se;
  if (m.fun) m.fun[v] = o.fun;
  if (m.nit) m.nit[v] = o.nit;
  if (m.status) m.status[v] = o.status;
  if (m.xd) { m.xd[3 * v] = r.x[0]; m.xd[3 * v + 1] = r.x[1]; m.xd[3 * v + 2] = r.x[2]; }
  if (m.fund) m.fund[v] = r.fun;
}

// parameters and per-voxel extras only; res / r2 come from the epilogue pass
template <bool kExtras = true>
__device__ __forceinline__ void store_fit(const DevMaps& m, int64_t v, const LaneResult& r) {
  m.k[v] = (float)r.x[0]; m.t2[v] = (float)r.x[1]; m.sigma[v] = (float)r.x[2];
  if constexpr (!kExtras) return;
  if (m.fun) m.fun[v] = (float)r.fun;
  if (m.nit) m.nit[v] = r.nit;
  if (m.status) m.status[v] = r.status;
  if (m.xd) { m.xd[3 * v] = r.x[0]; m.xd[3 * v + 1] = r.x[1]; m.xd[3 * v + 2] = r.x[2]; }
  if (m.fund) m.fund[v] = r.fun;
}

// One lane per voxel, one voxel per lane: used for the LM solver and as the fallback of the
// persistent kernel below.
template <int SOLVER, int PREC, int MODEL>
__global__ __launch_bounds__(kBlock) void fit_volume_kernel(const LaneParams P, const float* __restrict__ echoes,
                                                            int layout, const uint8_t* __restrict__ mask,
                                                            int64_t n_vox, DevMaps m) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * kBlock;
  const int64_t v = base + lane;
  const bool in_range = v < n_vox;
  const bool active = in_range && (mask == nullptr || mask[v] != 0);
  stage_echoes(lds, echoes, layout, P.n_te, n_vox, base, active);
  if (!in_range) return;
  if (!active) {
    store_masked(m, v);
    return;
  }
  bool finite;
  float y0_raw;
  const ObjCtx c = prepare_samples(P, lds + lane, kLdsStride, finite, y0_raw);
  LaneResult r;
  fit_lane_t<SOLVER, PREC, MODEL>(P, c, finite, y0_raw, r);
  LaneOutputs o;
  lane_epilogue(c, r, o, m.r2 != nullptr, m.se != nullptr);
  store_result(m, v, o, r);
}

// Closed-form log-linear fit (T2FIT_SOLVER_LOGLIN), TE-major stacks: one streaming pass, four
// consecutive voxels per lane so that every echo plane is read with 16-byte loads and every map is
// written with 16-byte stores (1 KiB per wave instruction).  Fit, bounds, float32 casts, residual map
// and the optional extras all happen in this pass: algorithmic HBM bytes = 4*nTE + 1 + 16 per voxel,
// and nothing but HBM bounds it.  A lane whose four voxels are all outside the mask issues no echo
// loads.  Samples are parked in LDS as [nTE][4][256] (lane-contiguous: conflict-free) so that the lane
// math shared with the other solvers can address them by a run-time echo index.
constexpr int kLoglinVec = 4;
constexpr int kLoglinTile = kBlock * kLoglinVec;

// kExtras = false is the four-map form (no r2 / se / fun / nit / status / float64 outputs): its code holds none
// of those evaluations, which keeps the four unrolled voxel bodies inside the instruction cache.
template <bool kExtras>
__global__ __launch_bounds__(kBlock) void loglin_volume_kernel(const LaneParams P, const float* __restrict__ echoes,
                                                               const uint8_t* __restrict__ mask, int64_t n_vox,
                                                               DevMaps m) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  const int64_t v0 = (int64_t)blockIdx.x * kLoglinTile + (int64_t)lane * kLoglinVec;
  if (v0 >= n_vox) return;  // n_vox is a multiple of 4 on this path: a lane is all in or all out
  const int n_te = P.n_te;
  bool act[kLoglinVec] = {true, true, true, true};
  if (mask) {
    const uchar4 mk = *reinterpret_cast<const uchar4*>(mask + v0);
    act[0] = mk.x != 0; act[1] = mk.y != 0; act[2] = mk.z != 0; act[3] = mk.w != 0;
  }
  const bool any = act[0] || act[1] || act[2] || act[3];
  // finite check, first sample and row maximum are taken while the samples are still in registers
  bool fin[kLoglinVec] = {true, true, true, true};
  float ymax[kLoglinVec] = {0.0f, 0.0f, 0.0f, 0.0f}, y0[kLoglinVec] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (any) {
    const float* src = echoes + v0;
    for (int i0 = 0; i0 < n_te; i0 += 8) {  // up to eight 16-byte loads in flight per lane
      float4 tmp[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (i0 + j < n_te) tmp[j] = *reinterpret_cast<const float4*>(src + (int64_t)(i0 + j) * n_vox);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (i0 + j < n_te) {
          float* dst = lds + (i0 + j) * kLoglinTile + lane;
          const float sv[kLoglinVec] = {tmp[j].x, tmp[j].y, tmp[j].z, tmp[j].w};
#pragma unroll
          for (int q = 0; q < kLoglinVec; ++q) {
            dst[q * kBlock] = sv[q];
            fin[q] = fin[q] && t2_finite(sv[q]);
            if (i0 + j == 0) { y0[q] = sv[q]; ymax[q] = sv[q]; }
            ymax[q] = sv[q] > ymax[q] ? sv[q] : ymax[q];
          }
        }
    }
  }
  float o_t2[kLoglinVec], o_k[kLoglinVec], o_res[kLoglinVec];
  const bool want_fun = kExtras && (m.fun != nullptr || m.fund != nullptr);
#pragma unroll
  for (int q = 0; q < kLoglinVec; ++q) {
    o_t2[q] = 0.0f; o_k[q] = 0.0f; o_res[q] = 0.0f;
    const int64_t v = v0 + q;
    if (!act[q]) {
      if constexpr (kExtras) {
        if (m.r2) m.r2[v] = 0.0f;
        if (m.se) m.se[v] = 0.0f;
        if (m.fun) m.fun[v] = 0.0f;
        if (m.nit) m.nit[v] = 0;
        if (m.status) m.status[v] = T2FIT_ST_MASKED;
        if (m.xd) { m.xd[3 * v] = 0.0; m.xd[3 * v + 1] = 0.0; m.xd[3 * v + 2] = 0.0; }
        if (m.fund) m.fund[v] = 0.0;
      }
      continue;
    }
    float* col = lds + q * kBlock + lane;
    bool finite = fin[q];
    if (P.norm) {  // run_t2mapping.py:237-238: float32 / float32, as prepare_samples() does it
      for (int i = 0; i < n_te; ++i) {
        const float sv = col[i * kLoglinTile] / ymax[q];
        col[i * kLoglinTile] = sv;
        finite = finite && t2_finite(sv);
      }
    }
    ObjCtx c;
    c.P = &P;
    c.y = EchoView{col, kLoglinTile};
    double lb[3], ub[3];
    const bool feasible = lane_bounds(P, y0[q], lb, ub);
    LaneResult r;
    r.nit = 0; r.nfev = 0; r.fun = NAN;
    if (!feasible) {
      r.x[0] = NAN; r.x[1] = NAN; r.x[2] = 0.0;
      r.status = T2FIT_ST_INFEASIBLE;
    } else if (!finite) {
      r.x[0] = t2_clip(P.x0[0], lb[0], ub[0]); r.x[1] = t2_clip(P.x0[1], lb[1], ub[1]); r.x[2] = 0.0;
      r.status = T2FIT_ST_NONFINITE;
    } else {
      loglin_solve(c, lb, ub, want_fun, r);
    }
    LaneOutputs o;
    lane_epilogue(c, r, o, kExtras && m.r2 != nullptr, kExtras && m.se != nullptr);
    o_t2[q] = o.t2; o_k[q] = o.k; o_res[q] = o.res;
    if constexpr (kExtras) {
      if (m.r2) m.r2[v] = o.r2;
      if (m.se) m.se[v] = o.se;
      if (m.fun) m.fun[v] = o.fun;
      if (m.nit) m.nit[v] = o.nit;
      if (m.status) m.status[v] = o.status;
      if (m.xd) { m.xd[3 * v] = r.x[0]; m.xd[3 * v + 1] = r.x[1]; m.xd[3 * v + 2] = r.x[2]; }
      if (m.fund) m.fund[v] = r.fun;
    }
  }
  *reinterpret_cast<float4*>(m.t2 + v0) = make_float4(o_t2[0], o_t2[1], o_t2[2], o_t2[3]);
  *reinterpret_cast<float4*>(m.k + v0) = make_float4(o_k[0], o_k[1], o_k[2], o_k[3]);
  *reinterpret_cast<float4*>(m.sigma + v0) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  *reinterpret_cast<float4*>(m.res + v0) = make_float4(o_res[0], o_res[1], o_res[2], o_res[3]);
}

// Persistent form of the reference-trajectory fit.  The number of objective evaluations per voxel
// varies 4..200 (line searches), so with one voxel per lane a wave waits for its slowest voxel
// (measured lane efficiency 0.53).  Here a wave keeps all 64 lanes busy instead: waves pull chunks
// of kChunk (256, or 64 for small volumes) consecutive voxels from a global atomic counter, zero-fill the masked-out ones, queue
// the active ones in LDS, and every lane that finishes a voxel pops the next one.  The loop body is
// one objective+gradient evaluation (uniform, expensive) followed by the lane's solver advance
// (divergent, cheap); the wave leaves when a __ballot shows no lane has work and the queue is dry.
constexpr int kChunkLarge = 256;  // voxels a wave takes from the global queue at a time
constexpr int kChunkSmall = 64;   // small volumes (phantoms): more, smaller chunks so that every wave gets work
constexpr int64_t kSmallVolumeDefault = 1 << 20;
int64_t kSmallVolume = kSmallVolumeDefault;  // T2FIT_SMALL_VOLUME overrides (A/B runs): at or below, the generic small-chunk kernels
// T2FIT_TAKE: 64-voxel chunks per counter increment in the one-wave-workgroup kernels.  Measured (256^3 x 8 TE and its
// 1/2, 1/4, 1/8 shares, profiles/r02_exp58_mid_size.txt): 2 is best throughout -- 4 lengthens the drain of a launch
// (2.55 against 2.32 ms on a 1/8 share), 1 costs a little on whole volumes (13.73 against 13.55 ms)
int g_take = 0;
constexpr int kQueueCap = 64 + kChunkLarge;
constexpr int kDiagBlocks = 11, kDiagWords = 3 * kDiagBlocks;  // -DT2_PHASE_STAMPS: (cycles, lanes, entries) per block
#if defined(T2_WG_SHAPE_DIAG)
constexpr int kPlaceWords = 4096;  // diagnostic build: HW_ID / XCC_ID of every wave of the persistent kernel
#else
constexpr int kPlaceWords = 0;
#endif
constexpr int kCounterWords = 16 + kDiagWords + kPlaceWords;    // chunk counter + diagnostic totals

// what the persistent kernel needs to know about a resumable lane solver
template <int MODEL, int NTE = 0, bool GSPLIT = false> struct LbfgsbLane {
  using Solver = Lbfgsb<MODEL, NTE, GSPLIT>;
  // the one-wave-workgroup kernels keep one number of each pair in global memory (three parameters: eight waves per CU)
  using WaveWg = LbfgsbLane<MODEL, NTE, MODEL != T2FIT_MODEL_GAUSSIAN>;
  static constexpr bool kGlobalPart = Solver::kSplit;
  static constexpr int NP = Solver::N;
  static constexpr int kNte = NTE;  // > 0: the echo count is a compile-time constant (the refill loops flatten too)
  static constexpr int kHistDoubles = Solver::M * Solver::PAIR_L;  // correction pairs, per lane, in LDS
  static constexpr int kWavesPerSimd = 1;
  // one-wave workgroups, two waves on a SIMD (256 registers): every model.  (The Rician-likelihood lane needed 370
  // registers while its evaluation was unrolled over echoes and i0e coefficients; as loops -- t2fit_lbfgsb.h eval(),
  // t2fit_lane.h t2_log_i0e4 -- it fits.)
  static constexpr bool kWaveWgOk = true;
#ifndef T2_WAVE_HINT_2PAR
#define T2_WAVE_HINT_2PAR 3
#endif
  // (two parameters: 240 B of pairs per lane let ten waves share a CU, three on two of its SIMDs: 168 registers.  In
  // round 2 the lane spilled 29-45 registers at that limit and 7.28 ms became 7.75 -- profiles/r02_exp50_2par_three_waves.txt;
  // since the main loop has one way into the round the lane needs 155-168 and nothing spills: 6.37 -> 5.96 ms at
  // 256 x 256 x 180 x 6 TE, profiles/r03_exp23_2par_ten_waves.txt.  -DT2_WAVE_HINT_2PAR=2: eight waves.)
  static constexpr int kWaveWgHint = MODEL == T2FIT_MODEL_GAUSSIAN ? T2_WAVE_HINT_2PAR : kWaveHint;
  // lanes that must be idle before a wave refills.  Measured on MI355X at eight waves per CU (profiles/
  // r03_exp6_refill_take.txt), 256^3 x 8 TE: 1 -> 12.48 ms, 4 -> 12.13, 8 -> 12.03, 12 -> 12.20, 16 -> 12.42; the Rician
  // likelihood, whose evaluation is four times as long (an idle lane costs more): 4 -> 19.77, 8 -> 20.01, 16 -> 20.76
  static constexpr int kRefillMin = MODEL == T2FIT_MODEL_RICIAN ? 4 : 8;
  static constexpr bool kSplit = true;   // advance() = digest() + begin(): the kernel may batch begin() (T2FIT_PARK_MIN)
  __device__ static void init(Solver& s, const ObjCtx&, const double* x0, const double* lb, const double* ub,
                              double* hist, int hstride, double* ghist) { s.init(x0, lb, ub, hist, hstride, ghist, 64); }
  __device__ static void result(const Solver& s, const ObjCtx&, LaneResult& r) { s.result(r); }
  template <int J> __device__ static void take_sample(Solver& s, float y) {
    if constexpr (NTE > 0 && J < NTE) s.ys[J] = y;
  }
};
template <typename T, int NPAR, int NTE = 0> struct LmLaneAdaptor {
  using Solver = LmLane<T, NPAR, NTE>;
  using WaveWg = LmLaneAdaptor<T, NPAR, NTE>;
  static constexpr bool kGlobalPart = false;
  static constexpr int NP = NPAR;
  static constexpr int kNte = NTE;
  static constexpr int kHistDoubles = 0;
  static constexpr bool kWaveWgOk = false;
  static constexpr int kWaveWgHint = 1;
  // float32: four waves per SIMD (128 registers); float64: what the 229 registers of the lane allow (two, LDS permitting)
  static constexpr int kWavesPerSimd = sizeof(T) == 4 ? 4 : 1;
  static constexpr int kRefillMin = 24;  // measured (f32, 3 parameters, MI355X): 8 -> 1.62 ms, 16 -> 1.42 ms, 24 -> 1.35 ms, 32 -> 1.35 ms
  static constexpr bool kSplit = false;
  __device__ static void init(Solver& s, const ObjCtx& c, const double* x0, const double* lb, const double* ub,
                              double*, int, double*) { s.init(c, x0, lb, ub); }
  __device__ static void result(const Solver& s, const ObjCtx& c, LaneResult& r) { s.result(c, r); }
  template <int J> __device__ static void take_sample(Solver&, float) {}
};

template <class A, int kChunk, bool kTrace, bool kExtras, int kWg = kBlock, bool kRegs = false>
__device__ __forceinline__ void persistent_fit(const LaneParams& P, const float* __restrict__ echoes, int layout,
                                               const uint8_t* __restrict__ mask, int64_t n_vox, const DevMaps& m,
                                               unsigned long long* next_chunk, int refill_min, int park_min, int take,
                                               double* ghist_all) {
  extern __shared__ float lds[];
  constexpr int NP = A::NP;
  // kRegs: samples and voxel queue in registers, LDS for the correction pairs only -- the form launched as one-wave
  // workgroups (kWg == 64; see launch_persistent: six or eight waves per CU instead of four)
  constexpr bool kWaveWg = kRegs;
  static_assert(kRegs ? (kChunk == 64 && A::kNte > 0 && A::kNte <= 8) : kWg == kBlock,
                "workgroup of 256 lanes, or waves taking 64-voxel chunks with the samples in registers");
  constexpr int kStride = kWaveWg ? 64 : kLdsStride;  // sample columns: the +1 pad is for the 256-wide transposes only
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* col = lds + threadIdx.x;
  // LDS, kWg == 256: [n_te][257] float sample columns | [kHistDoubles][256] double solver history | 4 queues
  //      kRegs:      [kHistDoubles][kWg] double solver history, nothing else (samples and queue are in registers)
  double* hist = reinterpret_cast<double*>(lds + (kWaveWg ? 0 : ((P.n_te * kStride + 1) & ~1))) + threadIdx.x;
  uint32_t* queue = reinterpret_cast<uint32_t*>(hist - threadIdx.x + A::kHistDoubles * kWg) + wave * kQueueCap;
  // A::kGlobalPart: this wave's M x 64 doubles of the pairs' global part ([ring slot][lane], 5 KiB, L2-resident)
  double* ghist = nullptr;
  if constexpr (A::kGlobalPart)
    ghist = ghist_all + ((size_t)blockIdx.x * (kWg / 64) + wave) * (A::Solver::M * 64) + lane;
  uint32_t qv = 0;  // kWaveWg: lane r holds the r-th waiting voxel of the chunk last taken
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const EchoView y{col, kStride};
  int q_head = 0, q_count = 0;
  bool chunks_left = true;
  bool busy = false, done = false;
  int64_t v = 0;
  typename A::Solver s;
  ObjCtx c;
  c.P = &P;
  c.y = y;
  // table start point and bounds: uniform, read once
  double box_x0[3], box_lb[3], box_ub[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { box_x0[j] = P.x0[j]; box_lb[j] = P.lb[j]; box_ub[j] = P.ub[j]; }
#if defined(T2_PHASE_STAMPS)  // diagnostic build only: where do a wave's cycles go?  (per-wave counters in LDS)
  unsigned long long* diag = reinterpret_cast<unsigned long long*>(queue - wave * kQueueCap + (kBlock / 64) * kQueueCap) +
                             wave * kDiagWords;
  if (lane < kDiagWords) diag[lane] = 0ull;
  c.diag = diag;
  const unsigned long long st_all = __builtin_amdgcn_s_memtime();
#endif
#if defined(T2_WG_SHAPE_DIAG)  // where did the dispatcher put this wave?
  {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned w = blockIdx.x * (kWg / 64) + wave;
    if (lane == 0 && w < (unsigned)kPlaceWords) next_chunk[16 + kDiagWords + w] = 0x100000000ull | ((unsigned long long)(xcc & 0xf) << 20) | (hw & 0xfffff);
  }
#endif
  bool parked = false;  // split solvers: digest() done, begin() pending (see park_min below)
  int pend = 0;
  // kWaveWg: the chunk queue is read two steps ahead, so that taking a chunk never waits for memory -- chunk A has
  // its base and its mask bytes (loaded when the chunk before it was taken), chunk B its counter value (lane 0)
  // one counter increment hands a wave `take` 64-voxel chunks in a row (launch_persistent: 2)
  const int kTake = take;
  int64_t base_a = 0;
  int sub_a = 0;
  uint32_t idx_b = 0;
  uint8_t mask_a = 0;
  auto load_mask = [&](int64_t base) -> uint8_t {
    const int64_t vv = base + lane;
    return vv < n_vox ? (mask ? mask[vv] : (uint8_t)1) : (uint8_t)0;
  };
  if constexpr (kWaveWg) {
    uint32_t c0 = 0;
    if (lane == 0) c0 = (uint32_t)atomicAdd(next_chunk, 1ull);
    base_a = (int64_t)__shfl(c0, 0, 64) * (64 * kTake);
    mask_a = load_mask(base_a);
    if (lane == 0) idx_b = (uint32_t)atomicAdd(next_chunk, 1ull);
  }
  for (;;) {
    // Refill in batches: the refill path (sample loads, seed / set-up) runs with only the idle lanes
    // active, so it is entered when at least refill_min lanes are idle (or nothing is running).
    unsigned long long need = __ballot(!busy);
    if (__popcll(need) >= refill_min || need == ~0ull) {
      // (a loop of its own for the rare wave whose new voxels all ended at once -- samples that cannot be fitted -- so
      // that the round below has ONE way in and the solver state one version per trip of the main loop: with a
      // `continue` around the round the compiler kept the state in two register sets and copied it across, ~100 moves
      // per round)
      do {
      T2_BLK_T0(t_rf)
      // lanes that finished since the last refill hand in their results here, together, rather than one
      // or two at a time in the round they finished (the conversion and the stores are divergent code)
      if (done) {
        LaneResult r;
        A::result(s, c, r);
        store_fit<kExtras>(m, v, r);
        done = false;
      }
      bool fresh = false;
      if constexpr (kWaveWg) {
        // The queue is one register: a chunk's active voxels are compacted into lanes 0.. (ds_permute: lane j sends its
        // voxel to the lane of its rank among the active ones) and idle lane number r reads entry q_head + r back
        // (ds_bpermute).  Serve from what is left of the last chunk, then take chunks until every idle lane has a voxel.
        unsigned long long want = need;
        for (;;) {
          if (q_count == 0) {
            if (!chunks_left) break;
            const int64_t base = base_a;
            if (base >= n_vox) { chunks_left = false; break; }
            const int64_t vv = base + lane;
            const bool act = mask_a != 0;
            // step the read-ahead: B's counter value has been back for a while; its mask bytes and the next counter
            // value are needed when the next chunk is taken, many rounds from now
            if (++sub_a < kTake) {
              base_a += 64;
              mask_a = load_mask(base_a);
            } else {
              sub_a = 0;
              base_a = (int64_t)__shfl(idx_b, 0, 64) * (64 * kTake);
              mask_a = load_mask(base_a);
              if (lane == 0) idx_b = (uint32_t)atomicAdd(next_chunk, 1ull);
            }
            const unsigned long long b = __ballot(act);
            q_count = __popcll(b);
            q_head = 0;
            // a full permutation: active lanes to the front in order, the others behind them
            const int dst = act ? __popcll(b & lt_mask) : q_count + __popcll(~b & lt_mask);
            qv = (uint32_t)__builtin_amdgcn_ds_permute(dst << 2, (int)(uint32_t)vv);
            if (vv < n_vox && !act) store_masked<kExtras, true>(m, vv);
            if (q_count == 0) continue;
          }
          const int rank = __popcll(want & lt_mask);
          const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute(((q_head + rank) & 63) << 2, (int)qv);
          if (!busy && rank < q_count) {
            v = (int64_t)got;
            busy = true;
            fresh = true;
          }
          const int n_want = __popcll(want);
          const int taken = n_want < q_count ? n_want : q_count;
          q_head += taken;
          q_count -= taken;
          want = __ballot(!busy);
          if (want == 0ull) break;
        }
      } else {
      const int n_need = __popcll(need);
      while (q_count < n_need && chunks_left) {
        unsigned long long cidx = 0;
        if (lane == 0) cidx = atomicAdd(next_chunk, 1ull);
        cidx = __shfl(cidx, 0, 64);
        const int64_t base = (int64_t)cidx * kChunk;
        if (base >= n_vox) { chunks_left = false; break; }
        if (q_head != 0) {  // fewer than 64 entries are left: move them to the front
          uint32_t tmp = 0;
          if (lane < q_count) tmp = queue[q_head + lane];
          if (lane < q_count) queue[lane] = tmp;
          q_head = 0;
        }
        // all mask bytes of the chunk first (independent loads, one latency), then the queue
        // pushes, then the zero stores of the masked-out voxels
        bool act[kChunk / 64];
#pragma unroll
        for (int q = 0; q < kChunk / 64; ++q) {
          const int64_t vv = base + q * 64 + lane;
          act[q] = vv < n_vox && (mask == nullptr || mask[vv] != 0);
        }
#pragma unroll
        for (int q = 0; q < kChunk / 64; ++q) {
          const unsigned long long b = __ballot(act[q]);
          if (act[q]) queue[q_count + __popcll(b & lt_mask)] = (uint32_t)(base + q * 64 + lane);
          q_count += __popcll(b);
        }
#pragma unroll
        for (int q = 0; q < kChunk / 64; ++q) {
          const int64_t vv = base + q * 64 + lane;
          if (vv < n_vox && !act[q]) store_masked<kExtras, true>(m, vv);
        }
      }
      if (!busy) {
        const int rank = __popcll(need & lt_mask);
        if (rank < q_count) {
          v = (int64_t)queue[q_head + rank];
          busy = true;
          fresh = true;
        }
      }
      const int taken = n_need < q_count ? n_need : q_count;
      q_head += taken;
      q_count -= taken;
      }
      if (fresh) {
        // this lane's samples: up to 8 loads in flight, checked in registers, parked in its LDS
        // column (already divided by the row maximum when cfg.norm, run_t2mapping.py:237-238)
        bool finite = true;
        float ymax = 0.0f, y0 = 0.0f;
        const int n_te = A::kNte > 0 ? A::kNte : P.n_te;
        float first8[8] = {};  // echo-count specialisations: the samples stay in registers (A::take_sample)
        for (int i0 = 0; i0 < n_te; i0 += 8) {
          float tmp[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (i0 + j < n_te)
              // (plain loads.  Non-temporal ones keep more of the half-written map lines of the voxels still being
              // fitted in L2 -- LM float32 writes 313 instead of 425 MB per volume -- but every 128-byte line of samples
              // is then fetched once per refill that touches it: 1.7 x the reads, 4 % slower)
              tmp[j] = T2_SAMPLE_LOAD(layout == T2FIT_LAYOUT_TE_MAJOR ? &echoes[(int64_t)(i0 + j) * n_vox + v]
                                                                      : &echoes[v * n_te + (i0 + j)]);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (i0 + j < n_te) {
              const float sv = tmp[j];
              if constexpr (!kWaveWg) col[(i0 + j) * kStride] = sv;
              if (i0 == 0) first8[j] = sv;
              finite = finite && t2_finite(sv);
              if (i0 + j == 0) { y0 = sv; ymax = sv; }
              ymax = sv > ymax ? sv : ymax;
            }
        }
        if (P.norm) {
          if constexpr (A::kNte > 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (j < n_te) {
                const float sv = first8[j] / ymax;
                first8[j] = sv;
                if constexpr (!kWaveWg) col[j * kStride] = sv;
                finite = finite && t2_finite(sv);
              }
          } else {
            for (int i = 0; i < n_te; ++i) {
              const float sv = col[i * kStride] / ymax;
              col[i * kStride] = sv;
              finite = finite && t2_finite(sv);
            }
          }
        }
        if constexpr (A::kNte > 0)
          static_for<0, 8>([&](auto JC) { A::template take_sample<decltype(JC)::value>(s, first8[decltype(JC)::value]); });
        double lb[3], ub[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { lb[j] = box_lb[j]; ub[j] = box_ub[j]; }
        if (P.no_prior) {  // run_t2mapping.py:243-245
          lb[0] = (double)y0; ub[0] = P.np_k_ub;
          lb[1] = P.np_t2_lb; ub[1] = P.np_t2_ub;
        }
        bool feasible = true;
#pragma unroll
        for (int j = 0; j < NP; ++j) feasible = feasible && !(lb[j] > ub[j]);
        if (!feasible || !finite) {  // nothing to iterate on (fit_lane_t documents both cases)
          LaneResult r;
          r.nit = 0; r.nfev = 0; r.fun = NAN;
          for (int j = 0; j < 3; ++j)
            r.x[j] = !feasible ? (j < NP ? NAN : 0.0) : (j < NP ? t2_clip(box_x0[j], lb[j], ub[j]) : 0.0);
          r.status = !feasible ? T2FIT_ST_INFEASIBLE : T2FIT_ST_NONFINITE;
          store_fit<kExtras>(m, v, r);
          busy = false;
        } else {
          if constexpr (kTrace) {  // per-iteration trace of this voxel (voxel seam only)
            c.trace = m.trace + (size_t)v * 4 * m.trace_cap;
            c.trace_cap = m.trace_cap;
            c.trace_n = m.trace_len + v;
            *c.trace_n = 0;
          }
          A::init(s, c, box_x0, lb, ub, hist, kWg, ghist);
#if defined(T2_PHASE_STAMPS)
          if constexpr (A::kSplit) s.diag = diag;
#endif
        }
      }
      T2_BLK_END(c, 8, t_rf)
      need = __ballot(!busy);
      } while (need == ~0ull && (chunks_left || q_count != 0));
    }
    if (__ballot(busy) == 0ull) break;  // nothing running and nothing left to take
    if constexpr (A::kSplit) {
#if defined(T2_PARK_SWITCH)
      // One round: every lane with a point to evaluate evaluates it (uniform code) and digests the result; lanes
      // whose line search has ended then run begin() (B, Cauchy point, subspace step, line-search set-up).
      // park_min > 1 holds those lanes back until that many of the wave are waiting (or no lane has anything to
      // evaluate), so that begin() runs with more lanes active; results do not depend on it.
      if (busy && !parked) {
        T2_BLK_T0(t_ev)
        s.eval(c);
        T2_BLK_END(c, 7, t_ev)
        pend = s.digest(c);
        if (pend == A::Solver::GO_DONE) { busy = false; done = true; }
        else if (pend != A::Solver::GO_TRIAL) parked = true;
      }
      const unsigned long long pb = __ballot(parked);
      if (pb != 0ull && (__popcll(pb) >= park_min || __ballot(busy && !parked) == 0ull)) {
        if (parked) {
          T2_BLK_T0(t_bg)
          pend = s.begin_pass(c, pend);
          T2_BLK_END(c, 9, t_bg)
          // GO_BEGIN / GO_FAIL: the iteration has to be begun again (memory dropped, line search could not start):
          // the lane stays parked and comes back here in the next round
          if (pend == A::Solver::GO_DONE) { parked = false; busy = false; done = true; }
          else if (pend == A::Solver::GO_TRIAL) parked = false;
        }
      }
#else
      // One round: every lane with a point to evaluate evaluates it (uniform code) and digests the result; lanes whose
      // line search has ended then run begin_pass() (B, Cauchy point, subspace step, line-search set-up).  `parked`: the
      // pass asked to be run again (memory dropped, line search could not start: rare) -- the lane comes back in the next
      // round and skips the evaluation.  One region under `busy`, so that the solver state has one version per round.
      // (-DT2_PARK_SWITCH: the round-2 form with T2FIT_PARK_MIN, which held the lanes that need begin_pass() back until
      // that many of a wave were waiting; measured twice, never paid.)
      (void)park_min;
      if (busy) {
        if (!parked) {
          T2_BLK_T0(t_ev)
          s.eval(c);
          T2_BLK_END(c, 7, t_ev)
          pend = s.digest(c);
        }
        if (pend == A::Solver::GO_BEGIN || pend == A::Solver::GO_FAIL) {
          T2_BLK_T0(t_bg)
          pend = s.begin_pass(c, pend);
          T2_BLK_END(c, 9, t_bg)
        }
        parked = pend == A::Solver::GO_BEGIN || pend == A::Solver::GO_FAIL;
        if (pend == A::Solver::GO_DONE) { busy = false; done = true; }
      }
#endif
    } else {
      if (busy) s.eval(c);
      if (busy && s.advance(c)) {
        busy = false;
        done = true;
      }
    }
  }
  if (done) {  // (every exit passes through the refill block above; kept for safety)
    LaneResult r;
    A::result(s, c, r);
    store_fit<kExtras>(m, v, r);
  }
#if defined(T2_PHASE_STAMPS)
  if (lane == 0) diag[3 * 10] += __builtin_amdgcn_s_memtime() - st_all;  // block 10: the wave's whole life
  if (lane < kDiagWords) atomicAdd(next_chunk + 16 + lane, diag[lane]);
#endif
}

// The kernel proper.  kWavesPerSimd is the occupancy the register allocator is asked to keep: 1 for the
// float64 solvers (the L-BFGS-B lane alone holds ~340 registers), 4 for the float32 LM lane, which sits a few
// registers above the 128-register line of four waves per SIMD without the hint.
template <class A, int kChunk, bool kTrace = false, int kWavesPerSimd = 1, bool kExtras = true, int kWg = kBlock,
          bool kRegs = false>
__global__ __launch_bounds__(kWg, kWavesPerSimd) void fit_persistent_kernel(const LaneParams P,
                                                                    const float* __restrict__ echoes, int layout,
                                                                    const uint8_t* __restrict__ mask, int64_t n_vox,
                                                                    DevMaps m, unsigned long long* next_chunk, int refill_min,
                                                                    int park_min, int take, double* ghist) {
  persistent_fit<A, kChunk, kTrace, kExtras, kWg, kRegs>(P, echoes, layout, mask, n_vox, m, next_chunk, refill_min, park_min, take, ghist);
}

// Residual map (utils/t2map_utils.py:62-89) and optional R^2 from float32 maps already on the device.
// Also the second, fully uniform pass of the reference-trajectory fit: the persistent kernel stores
// t2/k/sigma only, because a lane that evaluated 8 float64 exp() for one finished voxel would hold
// up the other 63 lanes of its wave.
__global__ __launch_bounds__(kBlock) void residuals_kernel(const LaneParams P, const float* __restrict__ echoes,
                                                           int layout, const uint8_t* __restrict__ mask,
                                                           int64_t n_vox, const float* __restrict__ t2,
                                                           const float* __restrict__ k,
                                                           const float* __restrict__ sigma, float* res, float* r2,
                                                           float* se) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * kBlock;
  const int64_t v = base + lane;
  const bool in_range = v < n_vox;
  const bool active = in_range && (mask == nullptr || mask[v] != 0);
  stage_echoes(lds, echoes, layout, P.n_te, n_vox, base, active);
  if (!in_range) return;
  float out = 0.0f, out2 = 0.0f, out3 = 0.0f;
  if (active) {
    bool finite;
    float y0_raw;
    const ObjCtx c = prepare_samples(P, lds + lane, kLdsStride, finite, y0_raw);
    const float kv = k[v], tv = t2[v], sv = sigma ? sigma[v] : 0.0f;
    out = residual_mean(c, kv, tv, sv);
    if (r2) out2 = r_squared(c, (double)kv, (double)tv, (double)sv);
    if (se) out3 = t2_std_error(c, (double)kv, (double)tv, (double)sv);
  }
  res[v] = out;
  if (r2) r2[v] = out2;
  if (se) se[v] = out3;
}

// ---- union mask + ordered flat indices (run_t2mapping.py:383-384,412,421) ----------------------
constexpr int kScanItems = 4;                      // voxels per lane
constexpr int kScanTile = kBlock * kScanItems;     // voxels per workgroup

__device__ __forceinline__ uint8_t union_at(const uint8_t* __restrict__ masks, int n_masks, int64_t n_vox, int64_t v) {
  uint8_t any = 0;
  for (int j = 0; j < n_masks; ++j) any |= masks[(int64_t)j * n_vox + v] != 0;
  return any;
}

__global__ __launch_bounds__(kBlock) void mask_count_kernel(const uint8_t* __restrict__ masks, int n_masks,
                                                            int64_t n_vox, uint8_t* __restrict__ mask_out,
                                                            int64_t* __restrict__ tile_counts) {
  __shared__ int wave_sum[kBlock / 64];
  const int64_t v0 = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int cnt = 0;
  for (int q = 0; q < kScanItems; ++q) {
    const int64_t v = v0 + q;
    if (v < n_vox) {
      const uint8_t u = union_at(masks, n_masks, n_vox, v);
      mask_out[v] = u;
      cnt += u;
    }
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kBlock / 64; ++w) s += wave_sum[w];
    tile_counts[blockIdx.x] = s;
  }
}

// exclusive scan of the per-tile counts by one workgroup (tiles <= N/1024: tens of thousands)
__global__ __launch_bounds__(1024) void tile_scan_kernel(int64_t* tile_counts, int64_t n_tiles, int64_t* total_out) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t per = (n_tiles + 1023) / 1024;
  const int64_t lo = (int64_t)t * per;
  const int64_t hi = lo + per < n_tiles ? lo + per : n_tiles;
  int64_t s = 0;
  for (int64_t i = lo; i < hi; ++i) s += tile_counts[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
    int64_t add = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  int64_t run = t == 0 ? 0 : part[t - 1];
  for (int64_t i = lo; i < hi; ++i) {
    const int64_t c = tile_counts[i];
    tile_counts[i] = run;
    run += c;
  }
  if (t == 1023) *total_out = part[1023];
}

__global__ __launch_bounds__(kBlock) void mask_write_kernel(const uint8_t* __restrict__ mask, int64_t n_vox,
                                                            const int64_t* __restrict__ tile_offsets,
                                                            int64_t* __restrict__ idx_out) {
  __shared__ int wave_sum[kBlock / 64];
  const int64_t v0 = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint8_t f[kScanItems];
  int cnt = 0;
  for (int q = 0; q < kScanItems; ++q) {
    const int64_t v = v0 + q;
    f[q] = v < n_vox ? mask[v] : 0;
    cnt += f[q];
  }
  // exclusive prefix of cnt within the wave, then across the 4 waves
  int incl = cnt;
  const int l = threadIdx.x & 63;
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off, 64);
    if (l >= off) incl += up;
  }
  if (l == 63) wave_sum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += wave_sum[w];
  int64_t pos = tile_offsets[blockIdx.x] + wave_off + (incl - cnt);
  for (int q = 0; q < kScanItems; ++q)
    if (f[q]) idx_out[pos++] = v0 + q;
}

// ---- per-label statistics of a map (utils/t2map_utils.py:43-53: nanmean / nanstd per vial) -------
// Two rounds, numpy's own algorithm: mean first, then the mean of squared deviations from it (a
// constant region gives exactly 0).  Deterministic: each thread tallies its strided share of the
// workgroup's contiguous span into its own LDS column (one slot per label), columns are combined by a
// fixed tree, and the per-workgroup partials are added in workgroup order by one thread per label.
constexpr int kMaxLabels = 32;  // 2 * 32 * 256 doubles of LDS = 128 KiB (the NIST phantom has 14 vials)

template <bool kSecond>
__global__ __launch_bounds__(kBlock) void label_partial_kernel(const float* __restrict__ map,
                                                               const int32_t* __restrict__ label, int64_t n_vox,
                                                               int n_labels, int64_t span, const double* __restrict__ mean,
                                                               double* __restrict__ part_sum, int64_t* __restrict__ part_cnt) {
  extern __shared__ double acc[];  // [n_labels][kBlock] sums, then [n_labels][kBlock] counts (as double)
  double* cnt = acc + (size_t)n_labels * kBlock;
  const int tid = threadIdx.x;
  for (int l = 0; l < n_labels; ++l) { acc[l * kBlock + tid] = 0.0; cnt[l * kBlock + tid] = 0.0; }
  const int64_t lo = (int64_t)blockIdx.x * span;
  const int64_t hi = lo + span < n_vox ? lo + span : n_vox;
  for (int64_t v = lo + tid; v < hi; v += kBlock) {
    const int32_t l = label[v] - 1;
    const float x = map[v];
    if (l >= 0 && l < n_labels && x == x) {  // NaN values are skipped, as nanmean / nanstd do
      double t = (double)x;
      if (kSecond) { t -= mean[l]; t *= t; }
      acc[l * kBlock + tid] += t;
      cnt[l * kBlock + tid] += 1.0;
    }
  }
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if (tid < off)
      for (int l = 0; l < n_labels; ++l) {
        acc[l * kBlock + tid] += acc[l * kBlock + tid + off];
        cnt[l * kBlock + tid] += cnt[l * kBlock + tid + off];
      }
    __syncthreads();
  }
  if (tid < n_labels) {
    part_sum[(size_t)blockIdx.x * n_labels + tid] = acc[tid * kBlock];
    part_cnt[(size_t)blockIdx.x * n_labels + tid] = (int64_t)cnt[tid * kBlock];
  }
}

// kSecond = false: mean_out = sum / count; true: std_out = sqrt(sum of squared deviations / count)
template <bool kSecond>
__global__ void label_final_kernel(const double* __restrict__ part_sum, const int64_t* __restrict__ part_cnt, int n_blocks,
                                   int n_labels, double* __restrict__ out, int64_t* __restrict__ count_out) {
  const int l = threadIdx.x;
  if (l >= n_labels) return;
  double s = 0.0;
  int64_t c = 0;
  for (int b = 0; b < n_blocks; ++b) { s += part_sum[(size_t)b * n_labels + l]; c += part_cnt[(size_t)b * n_labels + l]; }
  const double m = c > 0 ? s / (double)c : (double)NAN;  // numpy: mean of an empty slice is NaN
  out[l] = kSecond ? sqrt(m) : m;
  if (count_out) count_out[l] = c;
}

// ---- launch helpers -----------------------------------------------------------------------------
using FitKernel = void (*)(const LaneParams, const float*, int, const uint8_t*, int64_t, DevMaps);

FitKernel pick_kernel(const t2fit_config& c) {
  if (c.solver == T2FIT_SOLVER_LM) {
    if (c.precision == T2FIT_PREC_F32)
      return c.model == T2FIT_MODEL_GAUSSIAN
                 ? fit_volume_kernel<T2FIT_SOLVER_LM, T2FIT_PREC_F32, T2FIT_MODEL_GAUSSIAN>
                 : fit_volume_kernel<T2FIT_SOLVER_LM, T2FIT_PREC_F32, T2FIT_MODEL_GAUSSIAN_RICIAN>;
    return c.model == T2FIT_MODEL_GAUSSIAN
               ? fit_volume_kernel<T2FIT_SOLVER_LM, T2FIT_PREC_F64, T2FIT_MODEL_GAUSSIAN>
               : fit_volume_kernel<T2FIT_SOLVER_LM, T2FIT_PREC_F64, T2FIT_MODEL_GAUSSIAN_RICIAN>;
  }
  if (c.solver == T2FIT_SOLVER_LOGLIN)  // voxel-major stacks, ragged sizes, unaligned views: one voxel per lane
    return fit_volume_kernel<T2FIT_SOLVER_LOGLIN, T2FIT_PREC_F64, T2FIT_MODEL_GAUSSIAN>;
  return nullptr;  // the L-BFGS-B solver runs in the persistent kernel
}

// kLargeOnly: instantiate the two large-volume kernels only (the echo-count specialisations; small volumes and
// traced voxel batches use the generic lane, where compile time buys nothing)
// kWaveOnly: instantiate the one-wave-workgroup kernels of A only (the less common echo counts: compile time); where
// they do not apply (T2FIT_WAVE_WG=0, diagnostic builds) hipErrorNotSupported tells the caller to use the generic lane
template <class A, bool kLargeOnly = false, bool kWaveOnly = false>
hipError_t launch_persistent(unsigned grid, size_t lds_samples, hipStream_t st, const LaneParams& P,
                             const float* echoes, int layout, const uint8_t* mask, int64_t n_vox, const DevMaps& dm,
                             unsigned long long* counter, bool big) {
  constexpr int W = A::kWavesPerSimd;
  const bool extras = dm.r2 || dm.se || dm.fun || dm.nit || dm.status || dm.xd || dm.fund;
#if !defined(T2_PHASE_STAMPS)
  if constexpr (kLargeOnly && A::kHistDoubles > 0 && A::kWaveWgOk) {
    // One-wave workgroups.  The lane's correction pairs (400 B with three parameters, 240 B with two: s is kept as a
    // direction, t2fit_lbfgsb.h load_s) cap a CU's 160 KiB of LDS at 409 lanes: four waves as one 256-lane workgroup,
    // six as one-wave workgroups (round 2), and EIGHT -- two on every SIMD, what the lane's 256 registers allow -- once
    // one of a pair's five numbers lives in global memory instead (A::WaveWg, Lbfgsb<.., GSPLIT>: 320 B per lane, 16 of
    // a CU's 128 LDS pieces per wave; the 10 MiB of the global part stay in L2).  Measured on one box, maps identical
    // bit for bit: 256^3 x 8 TE 13.62 -> 12.02 ms; with the split ring but capped at six waves 14.05 (the global
    // accesses cost 3 %), at seven 12.96 (profiles/r03_exp3_eight_waves.txt).  Nothing but the pairs is in LDS: the
    // samples are in registers (echo-count specialisation), the voxel queue too.
    if (g_wave_wg) {
      // (T2FIT_WAVE_WG=2 / 3: the same register-queue code in workgroups of 256 / 128 lanes -- diagnostic builds only)
      using AW = typename A::WaveWg;  // three parameters: one number of every pair in global memory, 320 B of LDS per lane
      constexpr int kHint = AW::kWaveWgHint;  // waves per SIMD the register allocator is held to
      auto k64 = extras ? fit_persistent_kernel<AW, kChunkSmall, false, kHint, true, 64, true>
                        : fit_persistent_kernel<AW, kChunkSmall, false, kHint, false, 64, true>;
      unsigned wg = 64;
#if defined(T2_WG_SHAPE_DIAG)
      if (g_wave_wg == 2) { k64 = fit_persistent_kernel<AW, kChunkSmall, false, 1, false, 256, true>; wg = 256; }
      if (g_wave_wg == 3) { k64 = fit_persistent_kernel<AW, kChunkSmall, false, 1, false, 128, true>; wg = 128; }
#endif
      size_t lds64 = (size_t)AW::kHistDoubles * wg * sizeof(double);
      // (LDS is handed out in 1280-byte pieces, 128 to a CU: measured with tools/diag/wave_placement_probe.hip, five
      // workgroups of 32000 bytes are resident together, of 32768 four)
      unsigned per_cu = (unsigned)std::min<size_t>(wg == 64 ? 4 * kHint : 512 / wg, 128 / ((lds64 + 1279) / 1280));
      if (g_waves_per_cu > 0 && (unsigned)g_waves_per_cu < per_cu) {  // A/B switch: pad the allocation so that no more fit
        per_cu = (unsigned)g_waves_per_cu;
        lds64 = (size_t)(128 / per_cu) * 1280;
      }
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k64), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds64);
      if (e != hipSuccess) return e;
      int dev = 0, cus = 0;
      if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
      if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
      grid = std::min<unsigned>(grid, (unsigned)cus);  // every wave that fits is resident; more would only start to leave
      const unsigned n_wg = grid * per_cu;
      double* ghist = nullptr;
      if constexpr (AW::kGlobalPart) {  // M x 64 doubles per wave (5 KiB; 10 MiB for the whole chip: it lives in L2)
        e = hipMallocAsync((void**)&ghist, (size_t)n_wg * (wg / 64) * AW::Solver::M * 64 * sizeof(double), st);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(k64, dim3(n_wg), dim3(wg), lds64, st, P, echoes, layout, mask, n_vox, dm, counter,
                         g_refill_min > 0 ? g_refill_min : AW::kRefillMin, g_park_min,
                         g_take > 0 ? g_take : 2, ghist);
      e = hipGetLastError();
      if (ghist) {
        const hipError_t e2 = hipFreeAsync(ghist, st);
        if (e == hipSuccess) e = e2;
      }
      return e;
    }
  }
#endif
  if constexpr (kWaveOnly) {
    return hipErrorNotSupported;
  } else {
  auto kern = extras ? fit_persistent_kernel<A, kChunkLarge, false, W, true>
                     : fit_persistent_kernel<A, kChunkLarge, false, W, false>;
  if constexpr (!kLargeOnly) {
    if (dm.trace) kern = fit_persistent_kernel<A, kChunkSmall, true, W>;
    else if (!big) kern = fit_persistent_kernel<A, kChunkSmall, false, W>;
  }
  const size_t lds = ((lds_samples / sizeof(float) + 1) & ~(size_t)1) * sizeof(float) +
                     (size_t)A::kHistDoubles * kBlock * sizeof(double) +
                     (size_t)(kBlock / 64) * kQueueCap * sizeof(uint32_t)
#if defined(T2_PHASE_STAMPS)
                     + (size_t)(kBlock / 64) * kDiagWords * sizeof(unsigned long long)
#endif
      ;
  // > 64 KiB of dynamic LDS (the correction pairs of 256 lanes) has to be opted into
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, P, echoes, layout, mask, n_vox, dm, counter,
                     g_refill_min > 0 ? g_refill_min : A::kRefillMin, g_park_min, 1, (double*)nullptr);
  return hipGetLastError();
  }
}

// reference-trajectory lane for `model`, specialised for the common echo-train lengths on large volumes
template <int MODEL>
hipError_t launch_lbfgsb(int n_te, bool large, unsigned grid, size_t lds_samples, hipStream_t st, const LaneParams& P,
                         const float* echoes, int layout, const uint8_t* mask, int64_t n_vox, const DevMaps& dm,
                         unsigned long long* counter) {
  if (large && g_nte_special) {
    if (n_te == 8) return launch_persistent<LbfgsbLane<MODEL, 8>, true>(grid, lds_samples, st, P, echoes, layout, mask, n_vox, dm, counter, large);
    if (n_te == 6) return launch_persistent<LbfgsbLane<MODEL, 6>, true>(grid, lds_samples, st, P, echoes, layout, mask, n_vox, dm, counter, large);
    if (n_te == 3) return launch_persistent<LbfgsbLane<MODEL, 3>, true>(grid, lds_samples, st, P, echoes, layout, mask, n_vox, dm, counter, large);
    // 7 / 5 / 4 echoes: the one-wave-workgroup kernels only (16.2 -> 12.6 ms at 5 echoes against the generic lane)
    {
      hipError_t e = hipErrorNotSupported;
      if (g_wave_wg == 1) {
        if (n_te == 7) e = launch_persistent<LbfgsbLane<MODEL, 7>, true, true>(grid, lds_samples, st, P, echoes, layout, mask, n_vox, dm, counter, large);
        if (n_te == 5) e = launch_persistent<LbfgsbLane<MODEL, 5>, true, true>(grid, lds_samples, st, P, echoes, layout, mask, n_vox, dm, counter, large);
        if (n_te == 4) e = launch_persistent<LbfgsbLane<MODEL, 4>, true, true>(grid, lds_samples, st, P, echoes, layout, mask, n_vox, dm, counter, large);
      }
      if (e != hipErrorNotSupported) return e;
    }
  }
  return launch_persistent<LbfgsbLane<MODEL>>(grid, lds_samples, st, P, echoes, layout, mask, n_vox, dm, counter, large);
}

int check_common(const t2fit_config* cfg, const void* echoes, int layout, int64_t n_vox) {
  const char* why;
  const int rc = config_check(cfg, &why);
  if (rc != T2FIT_OK) return fail(rc, why);
  if (!echoes) return fail(T2FIT_E_INVALID, "echoes is NULL");
  if (layout != T2FIT_LAYOUT_TE_MAJOR && layout != T2FIT_LAYOUT_VOXEL_MAJOR) return fail(T2FIT_E_INVALID, "unknown layout");
  if (n_vox < 0) return fail(T2FIT_E_INVALID, "n_vox is negative");
  if ((n_vox + kBlock - 1) / kBlock > 0x7fffffffLL) return fail(T2FIT_E_INVALID, "n_vox too large for one launch");
  return T2FIT_OK;
}

// the 4-voxels-per-lane form needs 16-byte aligned planes and maps, a 4-byte aligned mask and <= 64 KiB of LDS
bool loglin_vec_ok(const float* echoes, int layout, const uint8_t* mask, int64_t n_vox, const DevMaps& dm, int n_te) {
  auto al = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
  return layout == T2FIT_LAYOUT_TE_MAJOR && n_vox % kLoglinVec == 0 && al(echoes, 16) && al(mask, 4) && al(dm.t2, 16) &&
         al(dm.k, 16) && al(dm.sigma, 16) && al(dm.res, 16) && (size_t)n_te * kLoglinTile * sizeof(float) <= 65536;
}

// part_of_large: this call fits one slab of a large volume (the host seam): the large-volume kernels whatever its size
int launch_fit(const t2fit_config* cfg, const float* echoes, int layout, const uint8_t* mask, int64_t n_vox,
               const DevMaps& dm, hipStream_t st, bool part_of_large = false) {
  if (n_vox == 0) return T2FIT_OK;
  static const bool env_read = [] {  // tuning / A-B switches, read once
    if (const char* e = std::getenv("T2FIT_ONE_SHOT")) g_use_persistent = std::atoi(e) == 0;
    if (const char* e = std::getenv("T2FIT_PERSISTENT_BLOCKS")) g_persistent_blocks = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("T2FIT_REFILL_MIN")) g_refill_min = std::min(64, std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("T2FIT_RESERVE_CUS"); e && !g_reserve_set.load()) g_reserve_cus.store(std::max(0, std::atoi(e)));
    if (const char* e = std::getenv("T2FIT_NTE_SPECIAL")) g_nte_special = std::atoi(e) != 0;
    if (const char* e = std::getenv("T2FIT_TAKE")) g_take = std::max(0, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("T2FIT_SMALL_VOLUME")) kSmallVolume = std::max<int64_t>(0, std::atoll(e));
    if (const char* e = std::getenv("T2FIT_PARK_MIN")) g_park_min = std::min(64, std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("T2FIT_WAVE_WG")) g_wave_wg = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("T2FIT_WAVES_PER_CU")) g_waves_per_cu = std::max(0, std::atoi(e));
    return true;
  }();
  (void)env_read;
  const LaneParams P = make_lane_params(*cfg);
  const unsigned grid = (unsigned)((n_vox + kBlock - 1) / kBlock);
  const size_t lds = (size_t)cfg->n_te * kLdsStride * sizeof(float);
  FitKernel kern = pick_kernel(*cfg);
  const bool loglin = cfg->solver == T2FIT_SOLVER_LOGLIN;
  const bool persistent = !loglin && (g_use_persistent || cfg->solver == T2FIT_SOLVER_LBFGSB);
  if (persistent && n_vox >= 0xffffffffLL) return fail(T2FIT_E_INVALID, "n_vox must be below 2^32 per call");
  unsigned long long* counter = nullptr;
  if (persistent) {
    T2_HIP(hipMallocAsync((void**)&counter, kCounterWords * sizeof(unsigned long long), st));
    T2_HIP(hipMemsetAsync(counter, 0, kCounterWords * sizeof(unsigned long long), st));
  }
  if (g_timing) {
    g_ev_slot = (int)(g_ev_count % kEvRing);
    if (!g_ev0[g_ev_slot]) {
      T2_HIP(hipEventCreate(&g_ev0[g_ev_slot]));
      T2_HIP(hipEventCreate(&g_ev1[g_ev_slot]));
      T2_HIP(hipEventCreate(&g_ev2[g_ev_slot]));
    }
    T2_HIP(hipEventRecord(g_ev0[g_ev_slot], st));
  }
  if (persistent) {
    // one workgroup per CU slot; not required to be co-resident (work comes from an atomic queue)
    const bool big = !dm.trace && (part_of_large || n_vox > kSmallVolume);
    const int kChunk = big ? kChunkLarge : kChunkSmall;
    const int64_t chunks = (n_vox + kChunk - 1) / kChunk;
    unsigned pgrid = (unsigned)std::min<int64_t>((chunks + 3) / 4, g_persistent_blocks);
    const int reserve = g_reserve_cus.load(std::memory_order_relaxed);
    if (reserve > 0 && cfg->solver == T2FIT_SOLVER_LBFGSB) {
      // The resident workgroups of this kernel hold all of a CU's LDS.  Launched `reserve` CUs' worth of workgroups
      // short, the chip keeps that many workgroup slots (LDS and wave slots) free for kernels of other streams (RCCL's
      // all-gather beside the next fit), which otherwise could not become resident before this one drains.  With
      // one-wave workgroups the dispatcher spreads the shortfall over the CUs it likes: free slots, not whole CUs.
      int dev = 0, cus = 0;
      T2_HIP(hipGetDevice(&dev));
      T2_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      pgrid = std::min<unsigned>(pgrid, (unsigned)std::max(1, cus - reserve));
    }
    hipError_t pe;
#define T2_PERSIST(...) pe = launch_persistent<__VA_ARGS__>(pgrid, lds, st, P, echoes, layout, mask, n_vox, dm, counter, big)
    if (cfg->solver == T2FIT_SOLVER_LBFGSB) {
      const bool large = big;
#define T2_LBFGSB(M) pe = launch_lbfgsb<M>(cfg->n_te, large, pgrid, lds, st, P, echoes, layout, mask, n_vox, dm, counter)
      if (cfg->model == T2FIT_MODEL_GAUSSIAN) T2_LBFGSB(T2FIT_MODEL_GAUSSIAN);
      else if (cfg->model == T2FIT_MODEL_GAUSSIAN_RICIAN) T2_LBFGSB(T2FIT_MODEL_GAUSSIAN_RICIAN);
      else T2_LBFGSB(T2FIT_MODEL_RICIAN);
#undef T2_LBFGSB
    } else {
      // converged LM lane; on large volumes with a common echo-train length the echo-count specialisation
      const bool large = big && g_nte_special;
      const bool f32 = cfg->precision == T2FIT_PREC_F32, two = cfg->model == T2FIT_MODEL_GAUSSIAN;
#define T2_LM(T, NPAR)                                                                                                 \
  do {                                                                                                                 \
    if (large && cfg->n_te == 8) pe = launch_persistent<LmLaneAdaptor<T, NPAR, 8>, true>(pgrid, lds, st, P, echoes, layout, mask, n_vox, dm, counter, large); \
    else if (large && cfg->n_te == 6) pe = launch_persistent<LmLaneAdaptor<T, NPAR, 6>, true>(pgrid, lds, st, P, echoes, layout, mask, n_vox, dm, counter, large); \
    else if (large && cfg->n_te == 3) pe = launch_persistent<LmLaneAdaptor<T, NPAR, 3>, true>(pgrid, lds, st, P, echoes, layout, mask, n_vox, dm, counter, large); \
    else T2_PERSIST(LmLaneAdaptor<T, NPAR>);                                                                           \
  } while (0)
      // (float32 stays on the generic lane: the specialised evaluation keeps eight echoes in flight, which does not fit
      // the 128 registers of four waves per SIMD -- 204 B/lane of scratch, 2.0 ms -- and loses at three waves: 1.27 vs
      // 1.20 ms; float64: 2.73 vs 2.96 ms)
      if (f32 && two) T2_PERSIST(LmLaneAdaptor<float, 2>);
      else if (f32) T2_PERSIST(LmLaneAdaptor<float, 3>);
      else if (two) T2_LM(double, 2);
      else T2_LM(double, 3);
#undef T2_LM
    }
#undef T2_PERSIST
    if (pe != hipSuccess) {
      (void)hipFreeAsync(counter, st);
      return fail(T2FIT_E_HIP, std::string("persistent fit launch: ") + hipGetErrorString(pe));
    }
    if (g_timing) T2_HIP(hipEventRecord(g_ev1[g_ev_slot], st));  // the fit kernel ends here; the epilogue pass is a separate, HBM-bound launch
    hipLaunchKernelGGL(residuals_kernel, dim3(grid), dim3(kBlock), lds, st, P, echoes, layout, mask, n_vox,
                       (const float*)dm.t2, (const float*)dm.k, (const float*)dm.sigma, dm.res, dm.r2, dm.se);
    if (g_timing) {
      T2_HIP(hipEventRecord(g_ev2[g_ev_slot], st));
      ++g_ev_count;
    }
  } else if (loglin && loglin_vec_ok(echoes, layout, mask, n_vox, dm, cfg->n_te)) {
    const bool extras = dm.r2 || dm.se || dm.fun || dm.nit || dm.status || dm.xd || dm.fund;
    hipLaunchKernelGGL(extras ? loglin_volume_kernel<true> : loglin_volume_kernel<false>,
                       dim3((unsigned)((n_vox + kLoglinTile - 1) / kLoglinTile)), dim3(kBlock),
                       (size_t)cfg->n_te * kLoglinTile * sizeof(float), st, P, echoes, mask, n_vox, dm);
  } else {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, P, echoes, layout, mask, n_vox, dm);
  }
  T2_HIP(hipGetLastError());
  if (g_timing && !persistent) {  // one-pass kernels: no epilogue (its time reads 0)
    T2_HIP(hipEventRecord(g_ev1[g_ev_slot], st));
    T2_HIP(hipEventRecord(g_ev2[g_ev_slot], st));
    ++g_ev_count;
  }
#if defined(T2_PHASE_STAMPS)
  if (counter && cfg->solver == T2FIT_SOLVER_LBFGSB) {
    unsigned long long h[kCounterWords];
    T2_HIP(hipMemcpyAsync(h, counter, sizeof(h), hipMemcpyDeviceToHost, st));
    T2_HIP(hipStreamSynchronize(st));
    static const char* names[kDiagBlocks] = {"eval + digest", "cauchy: breakpoint taken", "subsm: projected", "begin: build_b",
                                             "begin: cauchy", "begin: subsm", "begin: ls set-up", "eval", "refill",
                                             "begin (all)", "wave life"};
    const double life = (double)h[16 + 3 * 10];
    for (int i = 0; i < kDiagBlocks; ++i) {
      const unsigned long long* d = h + 16 + 3 * i;
      fprintf(stderr, "[t2fit blocks] %-22s %6.2f%% of wave cycles, %5.1f lanes active, %10llu entries, %7.0f cycles each\n",
              names[i], 100.0 * (double)d[0] / life, d[2] ? (double)d[1] / (double)d[2] : 0.0, d[2],
              d[2] ? (double)d[0] / (double)d[2] : 0.0);
    }
  }
#endif
#if defined(T2_WG_SHAPE_DIAG)
  if (counter && cfg->solver == T2FIT_SOLVER_LBFGSB && std::getenv("T2FIT_PLACEMENT")) {
    std::vector<unsigned long long> h(kCounterWords);
    T2_HIP(hipMemcpyAsync(h.data(), counter, h.size() * 8, hipMemcpyDeviceToHost, st));
    T2_HIP(hipStreamSynchronize(st));
    std::map<unsigned, std::array<int, 4>> cu;
    for (int i = 0; i < kPlaceWords; ++i) {
      const unsigned long long v = h[16 + kDiagWords + i];
      if (!(v >> 32)) continue;
      const unsigned hw = (unsigned)v & 0xfffff, xcc = ((unsigned)v >> 20) & 0xf;
      cu[(xcc << 16) | ((hw >> 8) & 0xff)][(hw >> 4) & 3]++;
    }
    std::map<std::string, int> pat;
    for (auto& kv : cu) {
      std::array<int, 4> c = kv.second;
      std::sort(c.begin(), c.end());
      char b[64];
      snprintf(b, sizeof b, "%d,%d,%d,%d", c[3], c[2], c[1], c[0]);
      pat[b]++;
    }
    for (auto& kv : pat) fprintf(stderr, "[t2fit placement] %4d CUs with waves per SIMD %s\n", kv.second, kv.first.c_str());
  }
#endif
  if (counter) T2_HIP(hipFreeAsync(counter, st));
  return T2FIT_OK;
}

}  // namespace

extern "C" {

int t2fit_abi_version(void) { return T2FIT_ABI_VERSION; }

const char* t2fit_last_error(void) { return g_err.c_str(); }

int t2fit_config_default(t2fit_config* cfg, int model, int low_field) {
  const int rc = config_default_impl(cfg, model, low_field);
  return rc == T2FIT_OK ? rc : fail(rc, "t2fit_config_default: cfg is NULL or model unknown");
}

int t2fit_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int t2fit_set_reserve_cus(int cus) {
  g_reserve_set.store(true);
  return g_reserve_cus.exchange(std::max(0, cus));
}

int t2fit_set_timing(int enabled) {
  g_timing = enabled != 0;
  g_ev_count = 0;
  return T2FIT_OK;
}

double t2fit_kernel_ms(int launches_ago) {
  if (launches_ago < 0 || launches_ago >= kEvRing || launches_ago >= g_ev_count) return -1.0;
  const int slot = (int)((g_ev_count - 1 - launches_ago) % kEvRing);
  if (hipEventSynchronize(g_ev1[slot]) != hipSuccess) return -1.0;
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, g_ev0[slot], g_ev1[slot]) != hipSuccess) return -1.0;
  return (double)ms;
}

double t2fit_last_kernel_ms(void) { return t2fit_kernel_ms(0); }

double t2fit_epilogue_ms(int launches_ago) {
  if (launches_ago < 0 || launches_ago >= kEvRing || launches_ago >= g_ev_count) return -1.0;
  const int slot = (int)((g_ev_count - 1 - launches_ago) % kEvRing);
  if (hipEventSynchronize(g_ev2[slot]) != hipSuccess) return -1.0;
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, g_ev1[slot], g_ev2[slot]) != hipSuccess) return -1.0;
  return (double)ms;
}

int t2fit_volume_dev(const t2fit_config* cfg, const float* echoes_dev, int layout, const uint8_t* mask_dev,
                     int64_t n_vox, const t2fit_maps* maps, void* stream) {
  int rc = check_common(cfg, echoes_dev, layout, n_vox);
  if (rc != T2FIT_OK) return rc;
  if (!maps || !maps->t2 || !maps->k || !maps->sigma || !maps->res)
    return fail(T2FIT_E_INVALID, "maps->t2/k/sigma/res must be non-NULL");
  DevMaps dm{maps->t2, maps->k, maps->sigma, maps->res, maps->r2, maps->fun, maps->t2_se, maps->nit, maps->status,
             nullptr, nullptr};
  return launch_fit(cfg, echoes_dev, layout, mask_dev, n_vox, dm, (hipStream_t)stream);
}

// ---- host seam through a context (t2fit_context.h) -------------------------------------------------------------
#define T2_HIP_C(call)                                                          \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) {                                                     \
      cleanup();                                                                \
      return fail(T2FIT_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    }                                                                           \
  } while (0)

int t2fit_create(int device, t2fit_context** out) {
  if (!out) return fail(T2FIT_E_INVALID, "t2fit_create: out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    (void)hipGetLastError();
    return fail(T2FIT_E_HIP, "t2fit_create: no such HIP device");
  }
  T2_HIP(hipSetDevice(device));
  t2fit_context* c = new t2fit_context;
  c->device = device;
  auto cleanup = [&]() {
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_fit) (void)hipStreamDestroy(c->s_fit);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    delete c;
  };
  T2_HIP_C(hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
  T2_HIP_C(hipStreamCreateWithFlags(&c->s_fit, hipStreamNonBlocking));
  T2_HIP_C(hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
  int threads = 8;
  if (const char* e = std::getenv("T2FIT_COPY_THREADS")) threads = std::max(0, std::min(64, std::atoi(e)));
  c->pool = new t2fit::CopyPool(threads);
  *out = c;
  return T2FIT_OK;
}

int t2fit_destroy(t2fit_context* c) {
  if (!c) return T2FIT_OK;
  {
    std::lock_guard<std::mutex> g(c->busy);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->s_in);
    (void)hipStreamSynchronize(c->s_fit);
    (void)hipStreamSynchronize(c->s_out);
    for (auto ev : c->events) (void)hipEventDestroy(ev);
    for (int j = 0; j < 2; ++j) {
      if (c->pin_in[j]) (void)hipHostFree(c->pin_in[j]);
      if (c->pin_out[j]) (void)hipHostFree(c->pin_out[j]);
    }
    if (c->dev) (void)hipFree(c->dev);
    (void)hipStreamDestroy(c->s_in);
    (void)hipStreamDestroy(c->s_fit);
    (void)hipStreamDestroy(c->s_out);
    delete c->pool;
  }
  delete c;
  return T2FIT_OK;
}

int t2fit_context_volume_host(t2fit_context* c, const t2fit_config* cfg, const float* echoes, int layout,
                              const uint8_t* mask, int64_t n_vox, const t2fit_maps* maps) {
  if (!c) return fail(T2FIT_E_INVALID, "context is NULL");
  int rc = check_common(cfg, echoes, layout, n_vox);
  if (rc != T2FIT_OK) return rc;
  if (!maps || !maps->t2 || !maps->k || !maps->sigma || !maps->res)
    return fail(T2FIT_E_INVALID, "maps->t2/k/sigma/res must be non-NULL");
  if (n_vox == 0) return T2FIT_OK;
  std::lock_guard<std::mutex> guard(c->busy);
  T2_HIP(hipSetDevice(c->device));
  const int n_te = cfg->n_te;
  // Slabs of about 2.4 M voxels (multiples of 4096, so that every slab keeps the alignment the vectorised kernels
  // want; every slab of a large volume runs the large-volume kernels, also the short first and last ones): short enough that filling and draining the pipeline costs little,
  // long enough that a slab's fit covers the host-side copies of its neighbours.  The first slab is a quarter of
  // that: the device starts working after a quarter of the copy time.
  int64_t slab = (int64_t)9 << 18;  // 2,359,296
  bool graded = true;
  if (const char* e = std::getenv("T2FIT_HOST_SLABS")) {  // A/B switch and tests: number of (equal) slabs
    const int64_t want = std::max(1, std::min(4096, std::atoi(e)));
    slab = (n_vox + want - 1) / want;
    graded = false;
  }
  slab = std::max<int64_t>(4096, (slab + 4095) & ~(int64_t)4095);
  std::vector<int64_t> bounds{0};
  if (graded && n_vox > slab) bounds.push_back(std::max<int64_t>(4096, (slab / 4) & ~(int64_t)4095));
  while (bounds.back() < n_vox) bounds.push_back(std::min<int64_t>(n_vox, bounds.back() + slab));
  const int n_slabs = (int)bounds.size() - 1;
  // outputs: float maps (t2, k, sigma, res, r2, fun, t2_se), then nit (int32), then status (uint8)
  float* host_f[7] = {maps->t2, maps->k, maps->sigma, maps->res, maps->r2, maps->fun, maps->t2_se};
  const size_t slab_in = (size_t)slab * n_te * 4 + (size_t)slab;          // samples + mask bytes of one slab
  const size_t slab_out = (size_t)slab * (7 * 4 + 4 + 1);                  // every optional map wanted
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(c->s_in);
    (void)hipStreamSynchronize(c->s_fit);
    (void)hipStreamSynchronize(c->s_out);
  };
  if (c->pin_in_cap < slab_in) {
    for (int j = 0; j < 2; ++j) {
      if (c->pin_in[j]) (void)hipHostFree(c->pin_in[j]);
      c->pin_in[j] = nullptr;
    }
    c->pin_in_cap = 0;
    for (int j = 0; j < 2; ++j) T2_HIP_C(hipHostMalloc((void**)&c->pin_in[j], slab_in, hipHostMallocDefault));
    c->pin_in_cap = slab_in;
  }
  if (c->pin_out_cap < slab_out) {
    for (int j = 0; j < 2; ++j) {
      if (c->pin_out[j]) (void)hipHostFree(c->pin_out[j]);
      c->pin_out[j] = nullptr;
    }
    c->pin_out_cap = 0;
    for (int j = 0; j < 2; ++j) T2_HIP_C(hipHostMalloc((void**)&c->pin_out[j], slab_out, hipHostMallocDefault));
    c->pin_out_cap = slab_out;
  }
  // device arena: echoes (slab after slab, each (n_te, len) or (len, n_te)) | 7 float maps | nit | mask | status
  const size_t nb_e = (size_t)n_vox * n_te * sizeof(float);
  const size_t off_maps = (nb_e + 255) & ~(size_t)255;
  const size_t map_b = (((size_t)n_vox * 4) + 255) & ~(size_t)255;
  const size_t off_nit = off_maps + 7 * map_b;
  const size_t off_mask = off_nit + map_b;
  const size_t byte_b = ((size_t)n_vox + 255) & ~(size_t)255;
  const size_t off_status = off_mask + byte_b;
  const size_t total = off_status + byte_b;
  if (c->dev_cap < total) {
    if (c->dev) (void)hipFree(c->dev);
    c->dev = nullptr;
    c->dev_cap = 0;
    T2_HIP_C(hipMalloc((void**)&c->dev, total));
    c->dev_cap = total;
  }
  char* buf = c->dev;
  while (c->events.size() < (size_t)3 * n_slabs) {
    hipEvent_t ev;
    T2_HIP_C(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    c->events.push_back(ev);
  }
  hipEvent_t* ev_in = c->events.data();
  hipEvent_t* ev_fit = ev_in + n_slabs;
  hipEvent_t* ev_out = ev_fit + n_slabs;
  float* fm[7];
  for (int j = 0; j < 7; ++j) fm[j] = (float*)(buf + off_maps + j * map_b);
  const bool want[7] = {true, true, true, true, maps->r2 != nullptr, maps->fun != nullptr, maps->t2_se != nullptr};
  auto span = [&](int k, int64_t& lo, int64_t& len) { lo = bounds[k]; len = bounds[k + 1] - lo; };
  // Blocks of 4096 voxels without a single voxel in the mask are neither copied in (the kernels never read the samples
  // of a masked-out voxel) nor copied out (their maps are zeros: written here, not fetched): on a brain mask that is
  // half of the host-side copy traffic, which is what bounds this entry point.  runs[k]: the [start, end) voxel
  // ranges of slab k, relative to its start, that do hold masked voxels.
  constexpr int64_t kBlockVox = 4096;
  std::vector<std::vector<std::pair<int64_t, int64_t>>> runs(n_slabs);
  auto find_runs = [&](int k) {
    int64_t lo, len;
    span(k, lo, len);
    auto& r = runs[k];
    if (!mask) { r.emplace_back(0, len); return; }
    for (int64_t b = 0; b < len; b += kBlockVox) {
      const int64_t e = std::min(len, b + kBlockVox);
      const uint8_t* p = mask + lo + b;
      bool any = false;
      int64_t i = 0;
      for (; i + 8 <= e - b && !any; i += 8) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        any = w != 0;
      }
      for (; i < e - b && !any; ++i) any = p[i] != 0;
      if (!any) continue;
      if (!r.empty() && r.back().second == b) r.back().second = e;
      else r.emplace_back(b, e);
    }
  };
  // device -> pinned: the maps of slab k, packed one after the other in its staging slot
  auto queue_d2h = [&](int k) -> hipError_t {
    int64_t lo, len;
    span(k, lo, len);
    char* dst = c->pin_out[k & 1];
    hipError_t e = hipStreamWaitEvent(c->s_out, ev_fit[k], 0);
    size_t off = 0;
    for (int j = 0; j < 7 && e == hipSuccess; ++j)
      if (want[j]) { e = hipMemcpyAsync(dst + off, fm[j] + lo, (size_t)len * 4, hipMemcpyDeviceToHost, c->s_out); off += (size_t)len * 4; }
    if (e == hipSuccess && maps->nit) { e = hipMemcpyAsync(dst + off, buf + off_nit + (size_t)lo * 4, (size_t)len * 4, hipMemcpyDeviceToHost, c->s_out); off += (size_t)len * 4; }
    if (e == hipSuccess && maps->status) e = hipMemcpyAsync(dst + off, buf + off_status + lo, (size_t)len, hipMemcpyDeviceToHost, c->s_out);
    if (e == hipSuccess) e = hipEventRecord(ev_out[k], c->s_out);
    return e;
  };
  // pinned -> the caller's arrays (worker threads)
  auto finish_out = [&](int k) -> hipError_t {
    int64_t lo, len;
    span(k, lo, len);
    hipError_t e = hipEventSynchronize(ev_out[k]);
    if (e != hipSuccess) return e;
    const char* src = c->pin_out[k & 1];
    std::vector<t2fit::CopyPool::Row> rows;
    size_t off = 0;
    auto add = [&](char* dst, size_t elem) {  // one map of this slab: copy the runs, zero the gaps (src == nullptr)
      int64_t at = 0;
      for (const auto& r : runs[k]) {
        if (r.first > at) rows.push_back({dst + at * elem, nullptr, (size_t)(r.first - at) * elem});
        rows.push_back({dst + r.first * elem, src + off + r.first * elem, (size_t)(r.second - r.first) * elem});
        at = r.second;
      }
      if (at < len) rows.push_back({dst + at * elem, nullptr, (size_t)(len - at) * elem});
      off += (size_t)len * elem;
    };
    for (int j = 0; j < 7; ++j)
      if (want[j]) add((char*)(host_f[j] + lo), 4);
    if (maps->nit) add((char*)(maps->nit + lo), 4);
    if (maps->status) add((char*)(maps->status + lo), 1);  // T2FIT_ST_MASKED == 0
    c->pool->copy(rows);
    return hipSuccess;
  };
  for (int k = 0; k < n_slabs; ++k) {
    int64_t lo, len;
    span(k, lo, len);
    char* stage = c->pin_in[k & 1];
    if (k >= 2) T2_HIP_C(hipEventSynchronize(ev_in[k - 2]));  // the DMA out of this slot has finished
    find_runs(k);
    std::vector<t2fit::CopyPool::Row> rows;
    for (const auto& r : runs[k]) {
      const size_t nb = (size_t)(r.second - r.first);
      if (layout == T2FIT_LAYOUT_TE_MAJOR) {  // n_te rows of `len` samples out of planes of n_vox
        for (int i = 0; i < n_te; ++i)
          rows.push_back({stage + ((size_t)i * len + r.first) * 4, echoes + (size_t)i * n_vox + lo + r.first, nb * 4});
      } else {
        rows.push_back({stage + (size_t)r.first * n_te * 4, echoes + (size_t)(lo + r.first) * n_te, nb * n_te * 4});
      }
    }
    if (mask) rows.push_back({stage + (size_t)len * n_te * 4, mask + lo, (size_t)len});
    c->pool->copy(rows);
    float* d_e = (float*)buf + (size_t)lo * n_te;  // this slab's block of the device stack
    T2_HIP_C(hipMemcpyAsync(d_e, stage, (size_t)len * n_te * 4, hipMemcpyHostToDevice, c->s_in));
    uint8_t* dmask = nullptr;
    if (mask) {
      dmask = (uint8_t*)(buf + off_mask) + lo;
      T2_HIP_C(hipMemcpyAsync(dmask, stage + (size_t)len * n_te * 4, (size_t)len, hipMemcpyHostToDevice, c->s_in));
    }
    T2_HIP_C(hipEventRecord(ev_in[k], c->s_in));
    T2_HIP_C(hipStreamWaitEvent(c->s_fit, ev_in[k], 0));
    DevMaps dm{fm[0] + lo, fm[1] + lo, fm[2] + lo, fm[3] + lo, maps->r2 ? fm[4] + lo : nullptr,
               maps->fun ? fm[5] + lo : nullptr, maps->t2_se ? fm[6] + lo : nullptr,
               maps->nit ? (int32_t*)(buf + off_nit) + lo : nullptr,
               maps->status ? (uint8_t*)(buf + off_status) + lo : nullptr, nullptr, nullptr};
    rc = launch_fit(cfg, d_e, layout, dmask, len, dm, c->s_fit, n_vox > kSmallVolume);
    if (rc != T2FIT_OK) { cleanup(); return rc; }
    T2_HIP_C(hipEventRecord(ev_fit[k], c->s_fit));
    // the device -> host copy of the previous slab is queued behind this slab's host -> device copy: both directions
    // share one copy queue, and a queued copy that waits for a kernel would hold up every copy behind it
    if (k >= 1) T2_HIP_C(queue_d2h(k - 1));
    if (k >= 2) T2_HIP_C(finish_out(k - 2));
  }
  T2_HIP_C(queue_d2h(n_slabs - 1));
  if (n_slabs >= 2) T2_HIP_C(finish_out(n_slabs - 2));
  T2_HIP_C(finish_out(n_slabs - 1));
  return T2FIT_OK;
}

// The same seam without a context of the caller's: a per-device default context, created on first use and kept
// for the life of the process.
int t2fit_volume_host(const t2fit_config* cfg, const float* echoes, int layout, const uint8_t* mask, int64_t n_vox,
                      const t2fit_maps* maps, int device) {
  static std::mutex m;
  static std::vector<t2fit_context*> ctxs;
  t2fit_context* c = nullptr;
  {
    std::lock_guard<std::mutex> g(m);
    if (device >= 0 && (size_t)device < ctxs.size()) c = ctxs[device];
    if (!c) {
      const int rc = t2fit_create(device, &c);
      if (rc != T2FIT_OK) return rc;
      if ((size_t)device >= ctxs.size()) ctxs.resize(device + 1, nullptr);
      ctxs[device] = c;
    }
  }
  return t2fit_context_volume_host(c, cfg, echoes, layout, mask, n_vox, maps);
}

static int voxels_host_impl(const t2fit_config* cfg, const float* echoes, int layout, int64_t n_vox, const int64_t* idx,
                            int64_t n_idx, double* x, double* fun, int32_t* nit, uint8_t* status, int cap,
                            double* trace_x, int32_t* trace_len, int device) {
  int rc = check_common(cfg, echoes, layout, n_vox);
  if (rc != T2FIT_OK) return rc;
  if (n_idx < 0 || (n_idx > 0 && (!idx || !x))) return fail(T2FIT_E_INVALID, "idx/x must be non-NULL");
  if (cap < 0 || (cap > 0 && (!trace_x || !trace_len))) return fail(T2FIT_E_INVALID, "trace buffers must be non-NULL");
  if (n_idx == 0) return T2FIT_OK;
  const int n_te = cfg->n_te;
  // gather the requested rows into a compact voxel-major block on the host
  std::vector<float> rows((size_t)n_idx * n_te);
  for (int64_t r = 0; r < n_idx; ++r) {
    const int64_t v = idx[r];
    if (v < 0 || v >= n_vox) return fail(T2FIT_E_INVALID, "voxel index out of range");
    for (int i = 0; i < n_te; ++i)
      rows[(size_t)r * n_te + i] =
          layout == T2FIT_LAYOUT_TE_MAJOR ? echoes[(size_t)i * n_vox + v] : echoes[(size_t)v * n_te + i];
  }
  T2_HIP(hipSetDevice(device));
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t nb_e = rows.size() * sizeof(float);
  const size_t off_x = pad(nb_e);
  const size_t off_f = off_x + pad((size_t)n_idx * 24);
  const size_t off_maps = off_f + pad((size_t)n_idx * 8);
  const size_t map_b = pad((size_t)n_idx * 4);
  const size_t off_nit = off_maps + 4 * map_b;
  const size_t off_status = off_nit + map_b;
  const size_t off_tlen = off_status + pad((size_t)n_idx);
  const size_t off_trace = off_tlen + map_b;
  const size_t total = off_trace + pad((size_t)n_idx * cap * 32);
  char* buf = nullptr;
  T2_HIP(hipMalloc((void**)&buf, total));
  auto cleanup = [&]() { (void)hipFree(buf); };
  T2_HIP_C(hipMemcpy(buf, rows.data(), nb_e, hipMemcpyHostToDevice));
  DevMaps dm{(float*)(buf + off_maps), (float*)(buf + off_maps + map_b), (float*)(buf + off_maps + 2 * map_b),
             (float*)(buf + off_maps + 3 * map_b), nullptr, nullptr, nullptr, (int32_t*)(buf + off_nit),
             (uint8_t*)(buf + off_status), (double*)(buf + off_x), (double*)(buf + off_f)};
  if (cap > 0) {
    T2_HIP_C(hipMemset(buf + off_tlen, 0, map_b));
    dm.trace = (double*)(buf + off_trace);
    dm.trace_len = (int32_t*)(buf + off_tlen);
    dm.trace_cap = cap;
  }
  rc = launch_fit(cfg, (const float*)buf, T2FIT_LAYOUT_VOXEL_MAJOR, nullptr, n_idx, dm, nullptr);
  if (rc != T2FIT_OK) { cleanup(); return rc; }
  T2_HIP_C(hipDeviceSynchronize());
  T2_HIP_C(hipMemcpy(x, buf + off_x, (size_t)n_idx * 24, hipMemcpyDeviceToHost));
  if (fun) T2_HIP_C(hipMemcpy(fun, buf + off_f, (size_t)n_idx * 8, hipMemcpyDeviceToHost));
  if (nit) T2_HIP_C(hipMemcpy(nit, buf + off_nit, (size_t)n_idx * 4, hipMemcpyDeviceToHost));
  if (status) T2_HIP_C(hipMemcpy(status, buf + off_status, (size_t)n_idx, hipMemcpyDeviceToHost));
  if (cap > 0) {
    T2_HIP_C(hipMemcpy(trace_x, buf + off_trace, (size_t)n_idx * cap * 32, hipMemcpyDeviceToHost));
    T2_HIP_C(hipMemcpy(trace_len, buf + off_tlen, (size_t)n_idx * 4, hipMemcpyDeviceToHost));
  }
  cleanup();
  return T2FIT_OK;
}

int t2fit_voxels_host(const t2fit_config* cfg, const float* echoes, int layout, int64_t n_vox, const int64_t* idx,
                      int64_t n_idx, double* x, double* fun, int32_t* nit, uint8_t* status, int device) {
  return voxels_host_impl(cfg, echoes, layout, n_vox, idx, n_idx, x, fun, nit, status, 0, nullptr, nullptr, device);
}

int t2fit_voxels_trace_host(const t2fit_config* cfg, const float* echoes, int layout, int64_t n_vox,
                            const int64_t* idx, int64_t n_idx, double* x, double* fun, int32_t* nit, uint8_t* status,
                            int trace_cap, double* trace, int32_t* trace_len, int device) {
  if (trace_cap < 1) return fail(T2FIT_E_INVALID, "trace_cap must be >= 1");
  return voxels_host_impl(cfg, echoes, layout, n_vox, idx, n_idx, x, fun, nit, status, trace_cap, trace, trace_len,
                          device);
}

int t2fit_union_mask_dev(const uint8_t* masks_dev, int n_masks, int64_t n_vox, uint8_t* mask_out, int64_t* idx_out,
                         int64_t* count_out, void* stream) {
  if (!masks_dev || !mask_out || !idx_out || !count_out) return fail(T2FIT_E_INVALID, "NULL pointer");
  if (n_masks < 1 || n_vox < 0) return fail(T2FIT_E_INVALID, "n_masks < 1 or n_vox < 0");
  hipStream_t st = (hipStream_t)stream;
  if (n_vox == 0) {
    T2_HIP(hipMemsetAsync(count_out, 0, sizeof(int64_t), st));
    return T2FIT_OK;
  }
  const int64_t n_tiles = (n_vox + kScanTile - 1) / kScanTile;
  if (n_tiles > 0x7fffffffLL) return fail(T2FIT_E_INVALID, "n_vox too large");
  int64_t* tiles = nullptr;
  T2_HIP(hipMallocAsync((void**)&tiles, (size_t)n_tiles * sizeof(int64_t), st));
  hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)n_tiles), dim3(kBlock), 0, st, masks_dev, n_masks, n_vox, mask_out, tiles);
  hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, tiles, n_tiles, count_out);
  hipLaunchKernelGGL(mask_write_kernel, dim3((unsigned)n_tiles), dim3(kBlock), 0, st, (const uint8_t*)mask_out, n_vox,
                     (const int64_t*)tiles, idx_out);
  T2_HIP(hipGetLastError());
  T2_HIP(hipFreeAsync(tiles, st));
  return T2FIT_OK;
}

int t2fit_residuals_dev(const t2fit_config* cfg, const float* echoes_dev, int layout, const uint8_t* mask_dev,
                        int64_t n_vox, const float* t2, const float* k, const float* sigma, float* res, void* stream) {
  int rc = check_common(cfg, echoes_dev, layout, n_vox);
  if (rc != T2FIT_OK) return rc;
  if (!t2 || !k || !res) return fail(T2FIT_E_INVALID, "t2/k/res must be non-NULL");
  if (n_vox == 0) return T2FIT_OK;
  static const bool env_read = [] {  // tuning / A-B switches, read once
    if (const char* e = std::getenv("T2FIT_ONE_SHOT")) g_use_persistent = std::atoi(e) == 0;
    if (const char* e = std::getenv("T2FIT_PERSISTENT_BLOCKS")) g_persistent_blocks = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("T2FIT_REFILL_MIN")) g_refill_min = std::min(64, std::max(1, std::atoi(e)));
    return true;
  }();
  (void)env_read;
  const LaneParams P = make_lane_params(*cfg);
  const unsigned grid = (unsigned)((n_vox + kBlock - 1) / kBlock);
  const size_t lds = (size_t)cfg->n_te * kLdsStride * sizeof(float);
  hipLaunchKernelGGL(residuals_kernel, dim3(grid), dim3(kBlock), lds, (hipStream_t)stream, P, echoes_dev, layout,
                     mask_dev, n_vox, t2, k, sigma, res, (float*)nullptr, (float*)nullptr);
  T2_HIP(hipGetLastError());
  return T2FIT_OK;
}

int t2fit_label_stats_dev(const float* map_dev, const int32_t* label_dev, int64_t n_vox, int n_labels, double* mean_out,
                          double* std_out, int64_t* count_out, void* stream) {
  if (!map_dev || !label_dev || !mean_out || !std_out) return fail(T2FIT_E_INVALID, "NULL pointer");
  if (n_vox < 0 || n_labels < 1 || n_labels > kMaxLabels) return fail(T2FIT_E_INVALID, "n_vox < 0 or n_labels outside 1..32");
  hipStream_t st = (hipStream_t)stream;
  const int n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n_vox + 8 * kBlock - 1) / (8 * kBlock)));
  const int64_t span = (n_vox + n_blocks - 1) / n_blocks;
  double* part_sum = nullptr;
  int64_t* part_cnt = nullptr;
  T2_HIP(hipMallocAsync((void**)&part_sum, (size_t)n_blocks * n_labels * sizeof(double), st));
  T2_HIP(hipMallocAsync((void**)&part_cnt, (size_t)n_blocks * n_labels * sizeof(int64_t), st));
  const size_t lds = (size_t)2 * n_labels * kBlock * sizeof(double);
  auto k1 = label_partial_kernel<false>;
  auto k2 = label_partial_kernel<true>;
  T2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  T2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k1, dim3(n_blocks), dim3(kBlock), lds, st, map_dev, label_dev, n_vox, n_labels, span,
                     (const double*)nullptr, part_sum, part_cnt);
  hipLaunchKernelGGL(label_final_kernel<false>, dim3(1), dim3(kMaxLabels), 0, st, (const double*)part_sum,
                     (const int64_t*)part_cnt, n_blocks, n_labels, mean_out, count_out);
  hipLaunchKernelGGL(k2, dim3(n_blocks), dim3(kBlock), lds, st, map_dev, label_dev, n_vox, n_labels, span,
                     (const double*)mean_out, part_sum, part_cnt);
  hipLaunchKernelGGL(label_final_kernel<true>, dim3(1), dim3(kMaxLabels), 0, st, (const double*)part_sum,
                     (const int64_t*)part_cnt, n_blocks, n_labels, std_out, (int64_t*)nullptr);
  T2_HIP(hipGetLastError());
  T2_HIP(hipFreeAsync(part_sum, st));
  T2_HIP(hipFreeAsync(part_cnt, st));
  return T2FIT_OK;
}

}  // extern "C"
