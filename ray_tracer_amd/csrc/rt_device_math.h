// Device-side arithmetic for the path tracer (gfx950).
//
// Two math policies share every kernel:
//   Strict  sin/cos/log/log2/exp2 are fixed fp32 polynomial kernels built only from
//           IEEE add/mul/fma, so a CPU that evaluates the same expressions gets the
//           same bits.
//   Fast    the same calls map to the CDNA transcendental unit (v_sin/v_cos/v_log/
//           v_exp/v_sqrt/v_rsq, ~1 ulp, 8 issue cycles per wave instruction).
// Every accept/reject decision of the closest-hit search equals the IEEE arithmetic's under
// BOTH policies (rt_shading.h, Closest), so the primitive a ray hits does not depend on the
// policy; ray generation without jitter is IEEE too.  The fast policy takes hit distances and
// surface normals from the hardware sqrt / rcp / rsq units.
//
// This translation unit family is compiled with -ffp-contract=off: the compiler
// never fuses on its own; every fused multiply-add is written as fmaf().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct f3 {
  float x, y, z;
};

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
// dot: x product first, y then z folded in with one fused op each
__device__ __forceinline__ float dot(f3 a, f3 b) {
  return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}
// a*s + b per component, fused
__device__ __forceinline__ f3 fma3(f3 a, float s, f3 b) {
  return f3{__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y), __builtin_fmaf(a.z, s, b.z)};
}
// a*b + c per component, fused
__device__ __forceinline__ f3 fma3v(f3 a, f3 b, f3 c) {
  return f3{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y),
            __builtin_fmaf(a.z, b.z, c.z)};
}
// GLSL mix(x, y, a) = x*(1-a) + y*a with the y*a product fused into the sum
__device__ __forceinline__ float mixf(float x, float y, float a) {
  return __builtin_fmaf(y, a, x * (1.0f - a));
}
__device__ __forceinline__ f3 mix3(f3 x, f3 y, float a) {
  return f3{mixf(x.x, y.x, a), mixf(x.y, y.y, a), mixf(x.z, y.z, a)};
}

// IEEE-754 correctly rounded square root and quotient (hipcc's default for
// sqrtf and '/' on fp32; the bit-for-bit property is checked by the GPU tests).
#ifdef RT_MEASURE_APPROX_GEOMETRY
// measurement-only build: what the IEEE requirement costs (results are NOT exact)
__device__ __forceinline__ float ieee_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float ieee_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#else
__device__ __forceinline__ float ieee_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float ieee_div(float a, float b) { return a / b; }
#endif
__device__ __forceinline__ f3 ieee_normalize(f3 v) {
  float inv = ieee_div(1.0f, ieee_sqrt(dot(v, v)));
  return v * inv;
}

// ---- RNG: the shader's PCG hash (ray_tracing.comp:112-119) ------------------
__device__ __forceinline__ uint32_t pcg_next(uint32_t &s) {
  s = s * 747796405u + 2891336453u;
  uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (r >> 22) ^ r;
}
// float(u) / 4294967295.0: the literal is 2^32 in fp32, so this is an exact scale
__device__ __forceinline__ float pcg_float(uint32_t &s) { return (float)pcg_next(s) * 0x1p-32f; }

// The k-th state after s: the LCG step s -> s * A + C composed k times is s * A^k + C_k, so
// the seven states a hit bounce consumes are seven independent multiply-adds instead of a chain
// (same values; the draws no longer wait for one another).
__device__ __forceinline__ uint32_t pcg_jump(uint32_t s, int k) {
  constexpr uint32_t A[8] = {1u, 747796405u, 4201498105u, 3399858189u, 2200120369u, 2975532453u, 1226092713u, 371291773u};
  constexpr uint32_t C[8] = {0u, 2891336453u, 1192405134u, 568162667u, 878960812u, 3598456993u, 3109587930u, 457671207u};
  return s * A[k] + C[k];
}
// randomValue of a state that has already been advanced
__device__ __forceinline__ uint32_t pcg_output_bits(uint32_t s) {
  uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (r >> 22) ^ r;
}
__device__ __forceinline__ float pcg_output_float(uint32_t s) { return (float)pcg_output_bits(s) * 0x1p-32f; }

// ---- strict polynomial kernels ---------------------------------------------
namespace pm {

__device__ __forceinline__ float reduce(float x, int &quadrant) {
  float q = __builtin_rintf(x * 0.636619747f);
  float r = __builtin_fmaf(q, -1.57079601e+00f, x);
  r = __builtin_fmaf(q, -3.13916473e-07f, r);
  r = __builtin_fmaf(q, -5.39030253e-15f, r);
  quadrant = (int)q & 3;
  return r;
}
__device__ __forceinline__ float sin_poly(float r) {
  float z = r * r;
  float p = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  p = __builtin_fmaf(p, z, -1.6666654611e-1f);
  return __builtin_fmaf(p * z, r, r);
}
__device__ __forceinline__ float cos_poly(float r) {
  float z = r * r;
  float p = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  p = __builtin_fmaf(p, z, 4.166664568298827e-2f);
  return __builtin_fmaf(p * z, z, __builtin_fmaf(z, -0.5f, 1.0f));
}
__device__ __forceinline__ float sin(float x) {
  int q;
  float r = reduce(x, q);
  float v = (q & 1) ? cos_poly(r) : sin_poly(r);
  return (q & 2) ? -v : v;
}
__device__ __forceinline__ float cos(float x) {
  int q;
  float r = reduce(x, q);
  float v = (q & 1) ? sin_poly(r) : cos_poly(r);
  return ((q + 1) & 2) ? -v : v;
}
// both at once, one range reduction
__device__ __forceinline__ void sincos(float x, float &s, float &c) {
  int q;
  float r = reduce(x, q);
  float sp = sin_poly(r), cp = cos_poly(r);
  float sv = (q & 1) ? cp : sp;
  float cv = (q & 1) ? sp : cp;
  s = (q & 2) ? -sv : sv;
  c = ((q + 1) & 2) ? -cv : cv;
}

// ln of the mantissa part: x = m * 2^e with m in [sqrt(1/2), sqrt(2)); returns ln(m)
__device__ __forceinline__ float log_core(float x, float &fe) {
  uint32_t u = __float_as_uint(x);
  int e = (int)(u >> 23) - 126;
  float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
  if (m < 0.707106781f) {
    e -= 1;
    m = m + m;
  }
  float t = m - 1.0f;
  float z = t * t;
  float p = __builtin_fmaf(t, 7.0376836292e-2f, -1.1514610310e-1f);
  p = __builtin_fmaf(p, t, 1.1676998740e-1f);
  p = __builtin_fmaf(p, t, -1.2420140846e-1f);
  p = __builtin_fmaf(p, t, 1.4249322787e-1f);
  p = __builtin_fmaf(p, t, -1.6668057665e-1f);
  p = __builtin_fmaf(p, t, 2.0000714765e-1f);
  p = __builtin_fmaf(p, t, -2.4999993993e-1f);
  p = __builtin_fmaf(p, t, 3.3333331174e-1f);
  float y = (p * t) * z;
  y = __builtin_fmaf(z, -0.5f, y);
  fe = (float)e;
  return t + y;
}
__device__ __forceinline__ bool log_special(float x, float &out) {
  if (x != x) { out = x; return true; }
  if (x < 0.0f) { out = __uint_as_float(0x7fc00000u); return true; }
  if (x == 0.0f) { out = -__builtin_inff(); return true; }
  if (x == __builtin_inff()) { out = x; return true; }
  return false;
}
__device__ __forceinline__ float log(float x) {
  float out, fe;
  if (log_special(x, out)) return out;
  float lm;
  if (x < 0x1p-126f) {
    lm = log_core(x * 0x1p24f, fe);
    fe -= 24.0f;
  } else {
    lm = log_core(x, fe);
  }
  return __builtin_fmaf(fe, 0.693359375f, __builtin_fmaf(fe, -2.12194440e-4f, lm));
}
__device__ __forceinline__ float log2(float x) {
  float out, fe;
  if (log_special(x, out)) return out;
  float lm;
  if (x < 0x1p-126f) {
    lm = log_core(x * 0x1p24f, fe);
    fe = fe - 24.0f;
  } else {
    lm = log_core(x, fe);
  }
  return __builtin_fmaf(lm, 1.44269504f, fe);
}
__device__ __forceinline__ float exp2(float x) {
  if (x != x) return x;
  if (x >= 128.0f) return __builtin_inff();
  if (x < -125.0f) return 0.0f;
  float k = __builtin_rintf(x);
  float r = x - k;
  float p = __builtin_fmaf(r, 1.52527338e-5f, 1.54035304e-4f);
  p = __builtin_fmaf(p, r, 1.33335581e-3f);
  p = __builtin_fmaf(p, r, 9.61812911e-3f);
  p = __builtin_fmaf(p, r, 5.55041087e-2f);
  p = __builtin_fmaf(p, r, 2.40226507e-1f);
  p = __builtin_fmaf(p, r, 6.93147181e-1f);
  p = __builtin_fmaf(p, r, 1.0f);
  int ki = (int)k;
  if (ki > 127) {
    ki = 127;
    p = p * 2.0f;
  }
  return p * __uint_as_float((uint32_t)(ki + 127) << 23);
}
__device__ __forceinline__ float pow(float x, float y) { return exp2(y * log2(x)); }

}  // namespace pm

// ---- the policy the sampling code is written against --------------------------
template <bool STRICT>
struct Math;

template <>
struct Math<true> {
  // cos(theta), theta = (2*3.1415926) * u  (ray_tracing.comp:122,124)
  static __device__ __forceinline__ float box_cos(float u) { return pm::cos(6.283185f * u); }
  // -2*log(u)  (ray_tracing.comp:123)
  static __device__ __forceinline__ float neg2log(float u) { return -2.0f * pm::log(u); }
  // the same two for a draw still in its 32 bits (randomValue = bits / 2^32)
  static __device__ __forceinline__ float box_cos_bits(uint32_t r) { return box_cos((float)r * 0x1p-32f); }
  static __device__ __forceinline__ float neg2log_bits(uint32_t r) { return neg2log((float)r * 0x1p-32f); }
  // angle = u*2*PI  (ray_tracing.comp:135-136)
  static __device__ __forceinline__ void circle(float u, float &c, float &s) {
    pm::sincos((u * 2.0f) * 3.1415926f, s, c);
  }
  static __device__ __forceinline__ float sqrt(float x) { return ieee_sqrt(x); }
  static __device__ __forceinline__ f3 normalize(f3 v) { return ieee_normalize(v); }
  static __device__ __forceinline__ float pow(float x, float y) { return pm::pow(x, y); }
  // (p * strength) / width  (ray_tracing.comp:88,91); k = strength / width is unused here
  static __device__ __forceinline__ float jitter(float p, float strength, float width, float) {
    return ieee_div(p * strength, width);
  }
  static __device__ __forceinline__ float div(float a, float b) { return ieee_div(a, b); }
};

template <>
struct Math<false> {
  // v_cos_f32 / v_sin_f32 take their argument in revolutions
  static __device__ __forceinline__ float box_cos(float u) { return __builtin_amdgcn_cosf(u); }
  static __device__ __forceinline__ float neg2log(float u) {
    return -1.38629436f * __builtin_amdgcn_logf(u);  // -2 ln2 * log2(u)
  }
  // For a draw still in its 32 bits the scale by 2^-32 folds away.  Cosine: the cosine unit works
  // in revolutions, so 1 + (bits >> 9) / 2^23, assembled by ONE funnel shift (exponent of 1.0
  // over the top 23 bits), has the cosine of bits / 2^32 to within 2^-23 of a revolution.
  // Logarithm: -2 ln(bits / 2^32) = -2 ln2 (log2(bits) - 32), one multiply-add after the log
  // unit (64 ln2 is 32 x the other constant, exact, so a draw of 1.0 still gives exactly 0);
  // near 1.0 the result is only good to ~4e-6 absolute and may come out a hair below zero, hence
  // the magnitude (it becomes an operand modifier of the square root that follows).
  static __device__ __forceinline__ float box_cos_bits(uint32_t r) {
    return __builtin_amdgcn_cosf(__uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, r, 9u)));
  }
  static __device__ __forceinline__ float neg2log_bits(uint32_t r) {
    return __builtin_fabsf(__builtin_fmaf(__builtin_amdgcn_logf((float)r), -1.38629436f, 32.0f * 1.38629436f));
  }
  static __device__ __forceinline__ void circle(float u, float &c, float &s) {
    c = __builtin_amdgcn_cosf(u);
    s = __builtin_amdgcn_sinf(u);
  }
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ f3 normalize(f3 v) {
    return v * __builtin_amdgcn_rsqf(dot(v, v));
  }
  static __device__ __forceinline__ float pow(float x, float y) {
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
  }
  static __device__ __forceinline__ float jitter(float p, float, float, float k) { return p * k; }
  static __device__ __forceinline__ float div(float a, float b) {
    return a * __builtin_amdgcn_rcpf(b);
  }
};

}  // namespace rt
