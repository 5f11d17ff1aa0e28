// Where the dispatcher puts consecutive workgroups (gfx950): every 64-thread block records its
// XCC_ID and HW_ID registers, then spins ~20 us so that the blocks of the first rounds are
// resident together.  Prints, for the first blocks, XCD / SE / CU / SIMD / wave slot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k_where(unsigned *out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192, show = argc > 2 ? atoi(argv[2]) : 320;
  unsigned *d; hipMalloc(&d, n * 8); hipMemset(d, 0, n * 8);
  hipLaunchKernelGGL(k_where, dim3(n), dim3(64), 0, 0, d, 2000);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  for (int b = 0; b < show && b < n; ++b) {
    const unsigned hw = h[2 * b], x = h[2 * b + 1] & 15u;
    printf("block %4d  xcd %u se %u sh %u cu %2u simd %u wave %2u\n", b, x, (hw >> 13) & 7u, (hw >> 12) & 1u,
           (hw >> 8) & 15u, (hw >> 4) & 3u, hw & 15u);
  }
  // histogram: how many of the first 1024 / 2048 / 4096 blocks share a (xcd, se, cu, simd)
  for (int lim : {256, 1024, 2048, 4096}) {
    if (lim > n) break;
    std::vector<int> cnt(8 * 8 * 2 * 16 * 4, 0); int mx = 0, used = 0;
    for (int b = 0; b < lim; ++b) {
      const unsigned hw = h[2 * b], x = h[2 * b + 1] & 7u;
      const unsigned key = (((x * 8 + ((hw >> 13) & 7u)) * 2 + ((hw >> 12) & 1u)) * 16 + ((hw >> 8) & 15u)) * 4 + ((hw >> 4) & 3u);
      if (++cnt[key] == 1) ++used; if (cnt[key] > mx) mx = cnt[key];
    }
    printf("first %d blocks: %d distinct SIMDs, at most %d blocks on one SIMD\n", lim, used, mx);
  }
  return 0;
}
