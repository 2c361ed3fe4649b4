// Where do the waves of one-wave workgroups land?  Launches G workgroups of 64 lanes with L bytes of LDS and ~228
// VGPRs each (the L-BFGS-B lane's footprint), every wave spins ~2 ms so that all that fit are resident together,
// and records HW_ID / XCC_ID and the start time.  Output: waves per CU and how they spread over the four SIMDs.
//   hipcc -O2 --offload-arch=gfx950 -o wave_placement_probe wave_placement_probe.hip && ./wave_placement_probe 1280 32768
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(64, 2) void probe(unsigned* out, unsigned long long* t0, long long spin) {
  extern __shared__ float lds[];
  asm volatile("v_mov_b32 v227, 0" ::: "v227");
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long st = wall_clock64();  // 100 MHz, one clock for the whole device
  lds[threadIdx.x] = (float)hw;
  while ((long long)(wall_clock64() - st) < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; t0[blockIdx.x] = st; }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 1280;
  const int L = argc > 2 ? atoi(argv[2]) : 32768;
  unsigned* out; unsigned long long* t0;
  hipMalloc(&out, G * 8); hipMalloc(&t0, G * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, L);
  int occ = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, 64, L);
  hipLaunchKernelGGL(probe, dim3(G), dim3(64), L, 0, out, t0, 200000LL);  // 100 MHz counter: 2 ms
  hipError_t e = hipDeviceSynchronize();
  std::vector<unsigned> h(2 * G); std::vector<unsigned long long> t(G);
  hipMemcpy(h.data(), out, G * 8, hipMemcpyDeviceToHost); hipMemcpy(t.data(), t0, G * 8, hipMemcpyDeviceToHost);
  unsigned long long tmin = *std::min_element(t.begin(), t.end());
  std::map<unsigned, std::vector<int>> cu;  // (xcc, se, sh, cu) -> simd ids of the waves that started in the first 1 ms
  int late = 0;
  for (int i = 0; i < G; ++i) {
    if (t[i] - tmin > 100000ULL) { ++late; continue; }
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned simd = (hw >> 4) & 3, cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cu[(xcc << 16) | (se << 8) | (sh << 4) | cuid].push_back((int)simd);
  }
  std::map<std::string, int> pattern;
  for (auto& kv : cu) {
    int c[4] = {0, 0, 0, 0};
    for (int s : kv.second) c[s]++;
    std::sort(c, c + 4);
    char b[64]; snprintf(b, sizeof b, "%d waves: simd load %d,%d,%d,%d", (int)kv.second.size(), c[3], c[2], c[1], c[0]);
    pattern[b]++;
  }
  printf("grid %d lds %d: %s, occupancy API says %d workgroups/CU, %zu CUs seen, %d workgroups started late\n", G, L,
         hipGetErrorString(e), occ, cu.size(), late);
  for (auto& kv : pattern) printf("  %4d CUs with %s\n", kv.second, kv.first.c_str());
  return 0;
}
