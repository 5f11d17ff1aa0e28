// Dependent-load latency of ONE lane on an idle MI355X: a pointer chase through a random cycle of
// 64-byte records (the size of an LBVH node), for footprints that fit the vector L1, the XCD's L2,
// the Infinity Cache, and none of them.  Cycles per load by s_memtime; printed as JSON.
//   hipcc --offload-arch=gfx950 -O2 -o load_latency tools/micro/load_latency.hip && ./load_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

__global__ void k_chase(const uint4 *__restrict__ recs, uint32_t start, int hops, unsigned long long *out) {
  uint32_t at = start;
  // one warm-up lap fraction so that the first loads do not pay TLB misses alone
  for (int i = 0; i < 64; ++i) at = recs[(size_t)at * 4].x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < hops; ++i) at = recs[(size_t)at * 4].x;   // 64-byte records, 16-byte loads
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[0] = t1 - t0;
  out[1] = at;
}

static double chase(size_t bytes, int hops, int laps_warm) {
  const size_t n = bytes / 64;
  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  std::mt19937 rng(12345);
  std::shuffle(order.begin(), order.end(), rng);
  std::vector<uint4> host(n * 4);
  for (size_t i = 0; i < n; ++i) host[(size_t)order[i] * 4].x = order[(i + 1) % n];
  uint4 *d = nullptr;
  unsigned long long *out = nullptr, h[2] = {0, 0};
  if (hipMalloc(&d, n * 64) != hipSuccess || hipMalloc(&out, 16) != hipSuccess) return -1;
  hipMemcpy(d, host.data(), n * 64, hipMemcpyHostToDevice);
  for (int w = 0; w < laps_warm; ++w) {   // small footprints: walk the cycle first so it sits in the cache
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d, order[0], (int)n, out);
  }
  hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d, order[0], hops, out);
  hipDeviceSynchronize();
  hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
  hipFree(d);
  hipFree(out);
  return (double)h[0] / hops;
}

int main() {
  int clock_khz = 0;
  hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0);
  const double ticks_to_cycles = 1.0;   // an s_memtime tick is a shader cycle on gfx950
  struct { const char *name; size_t bytes; int hops; int warm; } rows[] = {
      {"l1_16KB", 16u << 10, 4096, 2}, {"l2_1MB", 1u << 20, 8192, 2}, {"mall_64MB", 64u << 20, 8192, 1},
      {"hbm_2GB", (size_t)2 << 30, 8192, 0}};
  std::printf("{\"clock_khz\": %d, \"unit\": \"shader cycles per dependent 16-byte load of a 64-byte record, one lane, idle chip\"", clock_khz);
  for (auto &r : rows) {
    const double t = chase(r.bytes, r.hops, r.warm);
    std::printf(", \"%s\": %.1f", r.name, t * ticks_to_cycles);
  }
  std::printf("}\n");
  return 0;
}
