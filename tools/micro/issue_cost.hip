// Issue cost of single instructions for ONE wave alone on a SIMD (gfx950): a loop of 32
// copies of the instruction, timed with s_memtime.  "ind" = four independent destination
// registers in rotation, "dep" = each result feeds the next.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP32(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define KERNEL(name, IND, DEP)                                                              \
  __global__ void name##_ind(unsigned long long *out, float seed) {                          \
    float a = seed + threadIdx.x, b = seed * 3.f, c = seed * 5.f, d = seed * 7.f;            \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                            \
    for (int i = 0; i < 256; ++i) { REP4(REP4(IND)) REP4(REP4(IND)) }                        \
    asm volatile("s_nop 0" ::"v"(a), "v"(b), "v"(c), "v"(d));                               \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                            \
    if (threadIdx.x == 0) out[0] = t1 - t0;                                                  \
    if (threadIdx.x == blockDim.x - 64) out[2] = t1 - t0;                                    \
    if (a + b + c + d == 12345.678f) out[1] = 1;                                             \
  }                                                                                          \
  __global__ void name##_dep(unsigned long long *out, float seed) {                          \
    float a = seed + threadIdx.x, b = seed * 3.f, c = seed * 5.f, d = seed * 7.f;            \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                            \
    for (int i = 0; i < 256; ++i) { REP32(DEP) }                                             \
    asm volatile("s_nop 0" ::"v"(a), "v"(b), "v"(c), "v"(d));                               \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                            \
    if (threadIdx.x == 0) out[0] = t1 - t0;                                                  \
    if (threadIdx.x == blockDim.x - 64) out[2] = t1 - t0;                                    \
    if (a + b + c + d == 12345.678f) out[1] = 1;                                             \
  }

#define I4(op)                                                                               \
  asm volatile(op " %0, %0, %4\n" op " %1, %1, %4\n" op " %2, %2, %4\n" op " %3, %3, %4"      \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
#define D1(op) asm volatile(op " %0, %0, %1" : "+v"(a) : "v"(seed));
#define I4u(op)                                                                              \
  asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3"                      \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#define D1u(op) asm volatile(op " %0, %0" : "+v"(a));

// each IND body holds 4 instructions; REP4(REP4()) x2 = 32 per loop trip
#define IND_BIN(op) I4(op)
KERNEL(k_add, I4("v_add_f32"), D1("v_add_f32"))
KERNEL(k_mul_lo, I4("v_mul_lo_u32"), D1("v_mul_lo_u32"))
KERNEL(k_mul_hi, I4("v_mul_hi_u32"), D1("v_mul_hi_u32"))
KERNEL(k_mul24, I4("v_mul_u32_u24"), D1("v_mul_u32_u24"))
KERNEL(k_addu, I4("v_add_u32"), D1("v_add_u32"))
KERNEL(k_xor, I4("v_xor_b32"), D1("v_xor_b32"))
KERNEL(k_lshr, I4("v_lshrrev_b32"), D1("v_lshrrev_b32"))
KERNEL(k_sqrt, I4u("v_sqrt_f32"), D1u("v_sqrt_f32"))
KERNEL(k_log, I4u("v_log_f32"), D1u("v_log_f32"))
KERNEL(k_cos, I4u("v_cos_f32"), D1u("v_cos_f32"))
KERNEL(k_rsq, I4u("v_rsq_f32"), D1u("v_rsq_f32"))
KERNEL(k_cvt, I4u("v_cvt_f32_u32"), D1u("v_cvt_f32_u32"))
KERNEL(k_mov, I4u("v_mov_b32"), D1u("v_mov_b32"))
KERNEL(k_snop, asm volatile("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0");, asm volatile("s_nop 0");)
KERNEL(k_snop1, asm volatile("s_nop 1\ns_nop 1\ns_nop 1\ns_nop 1");, asm volatile("s_nop 1");)
KERNEL(k_smov, asm volatile("s_mov_b32 s20, s21\ns_mov_b32 s22, s21\ns_mov_b32 s23, s21\ns_mov_b32 s24, s21" ::: "s20", "s22", "s23", "s24");,
       asm volatile("s_add_u32 s20, s20, s20" ::: "s20", "scc");)
KERNEL(k_mix, asm volatile("v_add_f32 %0, %0, %4\ns_mov_b32 s20, s21\nv_add_f32 %1, %1, %4\ns_mov_b32 s22, s21" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed) : "s20", "s22");,
       asm volatile("v_add_f32 %0, %0, %1\n" : "+v"(a) : "v"(seed));)

__global__ void k_pk_ind(unsigned long long *out, float seed) {
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 a = {seed, seed + 1.f}, b = {seed * 3.f, 1.f}, c = {seed * 5.f, 2.f}, d = {seed * 7.f, 3.f}, s2 = {seed, seed};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 256; ++i) {
    REP4(REP4(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s2));))
    REP4(REP4(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s2));))
  }
  asm volatile("s_nop 0" ::"v"(a), "v"(b), "v"(c), "v"(d));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if (threadIdx.x == blockDim.x - 64) out[2] = t1 - t0;
  if (a.x + b.x + c.x + d.y == 12345.678f) out[1] = 1;
}
__global__ void k_pk_dep(unsigned long long *out, float seed) {
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 a = {seed, seed + 1.f}, s2 = {seed, seed};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 256; ++i) { REP32(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(s2));) }
  asm volatile("s_nop 0" ::"v"(a));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if (a.x + a.y == 12345.678f) out[1] = 1;
}

// taken branches: 32 forward jumps per trip
__global__ void k_branch_ind(unsigned long long *out, float seed) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 256; ++i) { REP32(asm volatile("s_branch 1f\ns_nop 0\n1:");) }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_branch_dep(unsigned long long *out, float seed) {  // not-taken conditional
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 256; ++i) { REP32(asm volatile("s_cbranch_execz 1f\n1:");) }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

int main(int argc, char **argv) {
  unsigned long long *d;
  (void)hipMalloc(&d, 32);
  if (argc > 1) {
    // waves per SIMD: one block of 64 * 4 * n threads puts n waves on each SIMD of one CU;
    // thread 0 times its own wave's stream while the others run the same one beside it
    for (int n = 1; n <= 4; ++n) {
      unsigned long long a[3], b[3];
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_add_ind, dim3(1), dim3(256 * n), 0, 0, d, 1.5f);
        (void)hipMemcpy(a, d, 24, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_add_dep, dim3(1), dim3(256 * n), 0, 0, d, 1.5f);
        (void)hipMemcpy(b, d, 24, hipMemcpyDeviceToHost);
      }
      printf("%d wave(s) per SIMD: v_add_f32 cycles per instruction per wave: independent %.2f (oldest wave) %.2f (youngest), dependent %.2f / %.2f\n",
             n, a[0] / (256.0 * 128.0), a[2] / (256.0 * 128.0), b[0] / (256.0 * 32.0), b[2] / (256.0 * 32.0));
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_pk_ind, dim3(1), dim3(256 * n), 0, 0, d, 1.5f);
        (void)hipMemcpy(a, d, 24, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_sqrt_ind, dim3(1), dim3(256 * n), 0, 0, d, 1.5f);
        (void)hipMemcpy(b, d, 24, hipMemcpyDeviceToHost);
      }
      printf("                    v_pk_fma_f32 independent %.2f / %.2f, v_sqrt_f32 independent %.2f / %.2f\n",
             a[0] / (256.0 * 128.0), a[2] / (256.0 * 128.0), b[0] / (256.0 * 128.0), b[2] / (256.0 * 128.0));
    }
    return 0;
  }
  struct { const char *name; void (*ind)(unsigned long long *, float); void (*dep)(unsigned long long *, float); } T[] = {
#define E(n) {#n, n##_ind, n##_dep}
      E(k_add), E(k_mul_lo), E(k_mul_hi), E(k_mul24), E(k_addu), E(k_xor), E(k_lshr), E(k_sqrt), E(k_log), E(k_cos),
      E(k_rsq), E(k_cvt), E(k_mov), E(k_snop), E(k_snop1), E(k_smov), E(k_mix), E(k_pk), E(k_branch)};
  for (auto &t : T) {
    unsigned long long r[2][2];
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(t.ind, dim3(1), dim3(64), 0, 0, d, 1.5f);
      (void)hipMemcpy(&r[0], d, 8, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(t.dep, dim3(1), dim3(64), 0, 0, d, 1.5f);
      (void)hipMemcpy(&r[1], d, 8, hipMemcpyDeviceToHost);
    }
    const double n = 256.0 * 32.0;
    // the independent bodies hold four instructions per copy
    const bool four = t.ind != k_branch_ind;
    printf("%-10s independent %.2f cyc/instr   dependent %.2f cyc/instr\n", t.name, r[0][0] / (four ? 4.0 * n : n), r[1][0] / n);
  }
  return 0;
}
