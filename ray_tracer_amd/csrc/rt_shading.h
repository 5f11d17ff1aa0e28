// The shader's per-pixel program, piece by piece, as device functions shared by the
// megakernels and the wavefront stages.  Each function cites the lines of
// assets/shaders/ray_tracing.comp it implements.  The order of RNG draws and of
// floating-point operations is part of the contract: both kernel organisations call
// the same functions in the same per-path order and so produce identical bits.
#pragma once

#include "rt_device_math.h"
#include "rt_device_scene.h"

namespace rt {

// ---- camera ---------------------------------------------------------------------
// column-major mat4 * vec4, one x,y,z,w fma chain per output row
__device__ __forceinline__ float mat_row(const float *m, int r, float x, float y, float z, float w) {
  float acc = m[0 + r] * x;
  acc = __builtin_fmaf(m[4 + r], y, acc);
  acc = __builtin_fmaf(m[8 + r], z, acc);
  acc = __builtin_fmaf(m[12 + r], w, acc);
  return acc;
}

// screenToWorldDirection (ray_tracing.comp:224-237).  Depends only on the pixel,
// so it is evaluated once per pixel instead of once per sample.
__device__ __forceinline__ f3 pixel_direction(const Params &P, uint32_t px, uint32_t py) {
  float cx = ieee_div(2.0f * (float)px, P.viewport_w) - 1.0f;
  float cy = ieee_div(2.0f * (float)py, P.viewport_h) - 1.0f;
  float ex = mat_row(P.inv_proj, 0, cx, cy, -1.0f, 1.0f);
  float ey = mat_row(P.inv_proj, 1, cx, cy, -1.0f, 1.0f);
  f3 w = mk3(mat_row(P.inv_view, 0, ex, ey, -1.0f, 0.0f), mat_row(P.inv_view, 1, ex, ey, -1.0f, 0.0f),
             mat_row(P.inv_view, 2, ex, ey, -1.0f, 0.0f));
  return ieee_normalize(w);
}

// rngState seed (ray_tracing.comp:80-82): x*W + y, plus int(frame)*719393
__device__ __forceinline__ uint32_t pixel_seed(const Params &P, uint32_t px, uint32_t py,
                                               uint32_t frame) {
  return px * P.width + py + frame * 719393u;
}

// Primary ray of one sample (ray_tracing.comp:86-93): four RNG draws, defocus first.
template <bool STRICT>
__device__ __forceinline__ void sample_ray(const Params &P, f3 base_dir, uint32_t &rng, f3 &o,
                                           f3 &d) {
  using M = Math<STRICT>;
  float c, s;
  // randomPointInCircle (ray_tracing.comp:134-138): angle draw, then radius draw
  float u_angle = pcg_float(rng);
  float u_rad = pcg_float(rng);
  if (STRICT || P.defocus != 0.0f) {
    M::circle(u_angle, c, s);
    float r = M::sqrt(u_rad);
    float jx = M::jitter(c * r, P.defocus, P.fwidth, P.defocus_over_w);
    float jy = M::jitter(s * r, P.defocus, P.fwidth, P.defocus_over_w);
    o = mk3(P.cam_x + M::div(jx, P.fwidth), P.cam_y + M::div(jy, P.fheight), P.cam_z + 0.0f);
  } else {
    o = mk3(P.cam_x + 0.0f, P.cam_y + 0.0f, P.cam_z + 0.0f);
  }
  u_angle = pcg_float(rng);
  u_rad = pcg_float(rng);
  M::circle(u_angle, c, s);
  float r = M::sqrt(u_rad);
  d = mk3(base_dir.x + M::jitter(c * r, P.diverge, P.fwidth, P.diverge_over_w),
          base_dir.y + M::jitter(s * r, P.diverge, P.fwidth, P.diverge_over_w), base_dir.z + 0.0f);
}

// ---- closest hit, brute force (ray_tracing.comp:145-190) ---------------------------
//
// The shader's quadratic uses b = 2*dot(oc,d), disc = b*b - 4ac, dst = (-b - sqrt(disc))/(2a).
// With bh = dot(oc,d): disc = 4*(bh*bh - a*c) and dst = (-bh - sqrt(bh*bh - a*c))/a, and
// because scaling by 2 and 4 is exact in binary floating point (no overflow or
// underflow in play) every rounded intermediate is the exact half/quarter of the
// shader's, and dst is the same float.  a = dot(d,d) is per ray.
//
// Rejections taken before the expensive IEEE sqrt and divide, none of which changes
// a result:
//   bh > 0           -> -bh - sqrt(..) < 0, so dst < 0: the shader rejects it too
//   n >= best_n      -> n/a >= best_n/a after rounding (a > 0, rounding is monotonic),
//                       so "dst < closest.dst" is false
// A numerator in (-1e-30, 0) is let through to the exact test because its quotient
// may round to -0, which the shader's "dst >= 0" accepts.
struct Hit {
  float dst;
  int idx;  // -1: miss
  // centre of the sphere that was hit, for callers that shade in the same kernel (it saves
  // them a dependent fetch); dead code everywhere else
  float cx, cy, cz;
};

// float4 in the constant address space (memory no kernel of the launch writes)
typedef __attribute__((address_space(4))) const float4 ConstF4;
typedef __attribute__((address_space(4))) const uint32_t ConstU32;
__device__ __forceinline__ float4 load_const(const ConstF4 *p) { return make_float4(p->x, p->y, p->z, p->w); }
// scene arrays are read-only during a launch: read them through the constant address space
__device__ __forceinline__ float4 scene_f4(const float4 *p, size_t i) { return load_const((const ConstF4 *)p + i); }
__device__ __forceinline__ uint32_t scene_u32(const uint32_t *p, size_t i) { return ((const ConstU32 *)p)[i]; }
__device__ __forceinline__ float scene_f32(const float *p, size_t i) {
  return ((__attribute__((address_space(4))) const float *)p)[i];
}

// a x b, the second product of each component fused into the difference
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
  return mk3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
             __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}

// Ray-triangle test (extension: the reference has spheres only).  Moeller-Trumbore with
// the barycentric tests on the un-divided numerators, so only a ray that is inside the
// triangle pays for the IEEE divide:
//   e1 = v1-v0, e2 = v2-v0, p = d x e2, det = e1.p        (det == 0 or NaN: miss)
//   tv = o-v0, un = tv.p, q = tv x e1, vn = d.q, tn = e2.q
//   with s = sign(det): s*un >= 0, s*vn >= 0, s*(un+vn) <= s*det, s*tn >= 0, else miss
//   dst = tn / det.  Both faces are hit.
// Accepts when closer than best, or equally far with a lower primitive id (so the result
// does not depend on the order in which primitives are visited).  best_n is the sphere
// numerator shortcut of sphere_candidate; a triangle win invalidates it.
__device__ __forceinline__ void triangle_candidate(const float4 *__restrict__ T, int idx, f3 o, f3 d,
                                                   float &best_n, Hit &best) {
  const float4 a0 = scene_f4(T, 0), a1 = scene_f4(T, 1), a2 = scene_f4(T, 2);
  const f3 v0 = mk3(a0.x, a0.y, a0.z);
  const f3 e1 = mk3(a1.x - a0.x, a1.y - a0.y, a1.z - a0.z);
  const f3 e2 = mk3(a2.x - a0.x, a2.y - a0.y, a2.z - a0.z);
  const f3 p = cross(d, e2);
  const float det = dot(e1, p);
  if (!(det != 0.0f)) return;
  const float s = det > 0.0f ? 1.0f : -1.0f;
  const f3 tv = o - v0;
  const float un = dot(tv, p);
  if (!(un * s >= 0.0f)) return;
  const f3 q = cross(tv, e1);
  const float vn = dot(d, q);
  if (!(vn * s >= 0.0f) || !((un + vn) * s <= det * s)) return;
  const float tn = dot(e2, q);
  if (!(tn * s >= 0.0f)) return;
  // a ray that runs in the triangle's plane to within rounding: det is residue (tested last, so
  // only a candidate that would otherwise be accepted pays for it)
  if (!(det * det > 0x1p-40f * (dot(e1, e1) * dot(p, p)))) return;
  // the same angle against the plane's own normal: for a ray that also runs nearly along e2, p is
  // a small difference of large products and the residue in det is not small beside |e1| |p|
  {
    const f3 nn = cross(e1, e2);
    if (!(det * det > 0x1p-40f * (dot(d, d) * dot(nn, nn)))) return;
  }
  const float t = ieee_div(tn, det);
  if (t < best.dst || (t == best.dst && idx < best.idx)) {
    best.dst = t;
    best.idx = idx;
    best_n = __builtin_inff();
  }
}

// Exact sphere test with an explicit tie rule, for searches that do not visit primitives in
// index order (the tree): equal distances go to the lower index, which is what the
// shader's in-order loop with its strict "<" returns.  In an in-order loop the extra clause
// never fires, so the brute-force search uses it too.
// FAR (extension: the one dielectric sphere a refracted ray travels inside) takes the other
// root, dst = (-b + sqrt(disc)) / (2a), with the same acceptance test dst >= 0.
template <bool FAR = false>
__device__ __forceinline__ void sphere_candidate_tie(float4 g, int idx, f3 o, f3 d, float a,
                                                     float &best_n, Hit &best) {
  f3 oc = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
  float bh = dot(oc, d);
  float c = dot(oc, oc) - g.w;
  float disc = __builtin_fmaf(bh, bh, -(a * c));
  if (disc >= 0.0f && (FAR || bh <= 0.0f)) {
    float n = FAR ? -bh + ieee_sqrt(disc) : -bh - ieee_sqrt(disc);
    if (n <= best_n && n > -1e-30f) {
      float q = ieee_div(n, a);
      if (q >= 0.0f && (q < best.dst || (q == best.dst && idx < best.idx))) {
        best.dst = q;
        best.idx = idx;
        best.cx = g.x;
        best.cy = g.y;
        best.cz = g.z;
        best_n = n;
      }
    }
  }
}

constexpr int kNoHit = 0x7fffffff;  // "no primitive yet": loses every index tie

// Running closest hit of one search.  APPROX = false: every candidate through the exact
// tests above.  APPROX = true (fast policy, scenes without triangles): the same DECISIONS,
// most of them taken on approximate values.  With s' = hardware sqrt(disc) (within 1 ulp)
// the numerator n' = -bh - s' is within E = 2^-21 (s' + |bh|) of the exactly computed one
// (four times the worst case), and |bh| >= |n| bounds the numerator itself, so a gap wider
// than the two candidates' E separates their quotients by more than two ulps.  Hence
//   n' below -E                the exact numerator is negative: rejected, as the shader does;
//   n' above best + both E     the exact quotient cannot be smaller: rejected;
//   n' below best - both E, above its own E, not denormal-small: accepted;
//   anything else              decided by the exact arithmetic, tie rule included, on this
//                              candidate and on the current best (kept as bh, disc).
// By induction the winner is the exact winner, so hit ids do not depend on the policy;
// only the distance is approximate (n' * rcp(a), about 2 ulp).  The far root of the dielectric
// sphere a ray travels inside (numerator -bh + s') goes through the same bands, with
// E = 2^-21 (s' + |bh|), and the best remembers which root it is for the exact re-evaluation.
template <bool APPROX>
struct Closest;

template <>
struct Closest<false> {
  Hit best;
  float best_n;
  __device__ __forceinline__ void reset(float) {
    best.dst = 3.402823466e+38f;
    best.idx = kNoHit;
    best.cx = best.cy = best.cz = 0.0f;
    best_n = __builtin_inff();
  }
  template <bool LIKELY = false, bool FAR = false, bool SELECT = false>
  __device__ __forceinline__ void sphere(float4 g, int idx, f3 o, f3 d, float a) {
    sphere_candidate_tie(g, idx, o, d, a, best_n, best);
  }
  __device__ __forceinline__ void sphere_far(float4 g, int idx, f3 o, f3 d, float a) {
    sphere_candidate_tie<true>(g, idx, o, d, a, best_n, best);
  }
  __device__ __forceinline__ void triangle(const float4 *T, int idx, f3 o, f3 d) {
    triangle_candidate(T, idx, o, d, best_n, best);
  }
  __device__ __forceinline__ float dst() const { return best.dst; }
  __device__ __forceinline__ bool any() const { return best.idx != kNoHit; }
  __device__ __forceinline__ Hit result() const {
    Hit h = best;
    if (h.idx == kNoHit) h.idx = -1;
    return h;
  }
};

template <>
struct Closest<true> {
  float n, err;    // approximate numerator of the best so far and its error bound
  float bh, disc;  // its inputs, for an exact re-evaluation
  float sgn;       // which root the best is: -1 near (the shader's), +1 far (inside a dielectric sphere)
  float rcp_a;
  int idx;
  float cx, cy, cz;  // its centre (see Hit)
  __device__ __forceinline__ void reset(float a) {
    cx = cy = cz = 0.0f;
    n = __builtin_inff();
    err = 0.0f;
    bh = 0.0f;
    disc = 0.0f;
    sgn = -1.0f;
    rcp_a = __builtin_amdgcn_rcpf(a);
    idx = kNoHit;
  }
  // FAR: the far root, numerator -bh + sqrt(disc) (no early rejection on the sign of bh).
  // LIKELY: the caller expects most tests to find the sphere in front of the ray (code
  // layout only)
  // SELECT: the update as selects behind one three-way classification instead of nested exec-mask
  // regions (fewer scalar instructions and branches per test; it pays in the plain and mixed forms of
  // k_mega_regen, 1920x1080 1.040 -> 1.024 ms, and costs the register-capped batched kernel 0.6 %)
  template <bool LIKELY = false, bool FAR = false, bool SELECT = false>
  __device__ __forceinline__ void sphere(float4 g, int i, f3 o, f3 d, float a) {
    f3 oc = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
    const float cbh = dot(oc, d);
    const float c = dot(oc, oc) - g.w;
    const float cdisc = __builtin_fmaf(cbh, cbh, -(a * c));
    if (__builtin_expect(cdisc >= 0.0f && (FAR || cbh <= 0.0f), LIKELY)) {
      const float s = __builtin_amdgcn_sqrtf(cdisc);
      const float cn = FAR ? -cbh + s : -cbh - s;
      // E plus a floor that keeps denormal-small numerators out of the shortcuts
      const float cerr = __builtin_fmaf(4.76837158e-7f, FAR ? s + __builtin_fabsf(cbh) : s - cbh, 1e-20f);
      const float band = cerr + err;
      if constexpr (SELECT) {
      // accepted outright / rejected outright / too close to call, then ONE set of selects
      bool accept = cn > cerr && cn < n - band;
      const bool unsure = !accept && !(cn < -cerr || cn > n + band);
      if (__builtin_expect(unsure, 0)) {
        const float ne = FAR ? -cbh + ieee_sqrt(cdisc) : -cbh - ieee_sqrt(cdisc);
        if (ne > -1e-30f) {
          const float qe = ieee_div(ne, a);
          if (qe >= 0.0f) {
            accept = true;
            if (idx != kNoHit) {
              const float qb = ieee_div(-bh + sgn * ieee_sqrt(disc), a);
              accept = qe < qb || (qe == qb && i < idx);
            }
          }
        }
      }
      n = accept ? cn : n;
      err = accept ? cerr : err;
      bh = accept ? cbh : bh;
      disc = accept ? cdisc : disc;
      sgn = accept ? (FAR ? 1.0f : -1.0f) : sgn;
      idx = accept ? i : idx;
      cx = accept ? g.x : cx;
      cy = accept ? g.y : cy;
      cz = accept ? g.z : cz;
      } else {
      if (cn < -cerr || cn > n + band) return;
      bool accept = cn > cerr && cn < n - band;
      if (__builtin_expect(!accept, 0)) {
        // too close to call: the exact rule (rare: the compiler keeps it off the straight path)
        const float ne = FAR ? -cbh + ieee_sqrt(cdisc) : -cbh - ieee_sqrt(cdisc);
        if (ne > -1e-30f) {
          const float qe = ieee_div(ne, a);
          if (qe >= 0.0f) {
            accept = true;
            if (idx != kNoHit) {
              const float qb = ieee_div(-bh + sgn * ieee_sqrt(disc), a);
              accept = qe < qb || (qe == qb && i < idx);
            }
          }
        }
      }
      if (accept) {
        n = cn;
        err = cerr;
        bh = cbh;
        disc = cdisc;
        sgn = FAR ? 1.0f : -1.0f;
        idx = i;
        cx = g.x;
        cy = g.y;
        cz = g.z;
      }
      }
    }
  }
  __device__ __forceinline__ void sphere_far(float4 g, int i, f3 o, f3 d, float a) {
    sphere<false, true>(g, i, o, d, a);
  }
  // APPROX kernels are launched only for scenes without a mesh
  __device__ __forceinline__ void triangle(const float4 *, int, f3, f3) {}
  __device__ __forceinline__ float dst() const { return idx != kNoHit ? n * rcp_a : 3.402823466e+38f; }
  __device__ __forceinline__ bool any() const { return idx != kNoHit; }
  __device__ __forceinline__ Hit result() const {
    Hit h;
    h.idx = idx != kNoHit ? idx : -1;
    h.dst = dst();
    h.cx = cx;
    h.cy = cy;
    h.cz = cz;
    return h;
  }
};

// skip: id of the primitive the ray starts on or in (-1: none).  A triangle is left out: a
// planar primitive cannot be hit again by a ray that leaves it, and the triangle test has no
// epsilon.  A sphere -- only ever a dielectric one the ray was refracted into, and only in
// DIEL kernels -- is tested with its far root.
// Four consecutive geometry records (the array is padded to a multiple of four).
struct SphereGroup {
  float4 g0, g1, g2, g3;
};
__device__ __forceinline__ SphereGroup load_group(const Scene &S, uint32_t i) {
  const ConstF4 *geom = (const ConstF4 *)S.geom + i;
  SphereGroup G;
  G.g0 = load_const(geom);
  G.g1 = load_const(geom + 1);
  G.g2 = load_const(geom + 2);
  G.g3 = load_const(geom + 3);
  return G;
}

// GROUPED: spheres are fetched four at a time, for code whose running time is one wave's
// chain of dependent segments rather than the device's throughput.
// SMALL: scenes of at most four spheres take the straight-line form below (callers ask for it where it
// was measured to pay: the plain and mixed forms of k_mega_regen; 1920x1080 1.065 -> 1.044 ms, whereas the
// batched-throughout launch of 3840x2160 lost 1.2 % with it).
template <bool APPROX, bool GROUPED = false, bool DIEL = false, bool SMALL = false>
__device__ __forceinline__ Hit closest_hit_brute(const Scene &S, f3 o, f3 d, int skip) {
  const float a = dot(d, d);
  Closest<APPROX> C;
  C.reset(a);
  const uint32_t n = S.n_spheres;
  // At most four spheres (the reference's own scene has three): ONE 64-byte scalar fetch and one
  // wave-uniform jump into a straight line of tests, instead of a loop whose bookkeeping (index,
  // bound check, address) is ~10 scalar instructions and a taken branch per sphere.  The tests run
  // from the highest index down; the winner does not depend on the order (ties go to the lower index
  // by the rule in Closest, not by visiting order).
  if (SMALL && !DIEL && n <= 4u) {
    if (n) {
      const SphereGroup G = load_group(S, 0);
      switch (n) {
        case 4: C.template sphere<true, false, true>(G.g3, 3, o, d, a); [[fallthrough]];
        case 3: C.template sphere<true, false, true>(G.g2, 2, o, d, a); [[fallthrough]];
        case 2: C.template sphere<true, false, true>(G.g1, 1, o, d, a); [[fallthrough]];
        default: C.template sphere<true, false, true>(G.g0, 0, o, d, a);
      }
    }
  } else if constexpr (GROUPED) {
    // Each step is one 64-byte scalar fetch and the loop bookkeeping is paid once per four
    // tests: 7 % off the chain of a wave that runs alone, nothing for a device full of waves
    // (which hide the bookkeeping behind each other's arithmetic).
    for (uint32_t i = 0; i < n; i += 4u) {
      const SphereGroup G = load_group(S, i);
      if constexpr (DIEL) {
        const float4 g[4] = {G.g0, G.g1, G.g2, G.g3};
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
          if (i + k >= n) break;
          if ((int)(i + k) == skip) C.sphere_far(g[k], (int)(i + k), o, d, a);
          else C.template sphere<true, false, SMALL>(g[k], (int)(i + k), o, d, a);
        }
        continue;
      }
      C.template sphere<true, false, SMALL>(G.g0, (int)i, o, d, a);
      if (i + 1u < n) C.template sphere<true, false, SMALL>(G.g1, (int)(i + 1u), o, d, a);
      if (i + 2u < n) C.template sphere<true, false, SMALL>(G.g2, (int)(i + 2u), o, d, a);
      if (i + 3u < n) C.template sphere<true, false, SMALL>(G.g3, (int)(i + 3u), o, d, a);
    }
  } else {
    // The scene is read-only for the whole launch.  Reading it through the constant address
    // space tells the compiler so, and a wave-uniform index then becomes a scalar load (one
    // fetch per wave through the scalar cache) instead of a vector load of 64 equal addresses.
    const ConstF4 *geom = (const ConstF4 *)S.geom;
    // the next sphere is fetched while the current one is tested
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n) g = load_const(geom);
    for (uint32_t i = 0; i < n; ++i) {
      const float4 g_next = load_const(geom + (i + 1u < n ? i + 1u : i));
      if (DIEL && (int)i == skip) C.sphere_far(g, (int)i, o, d, a);
      else C.template sphere<false, false, SMALL>(g, (int)i, o, d, a);
      g = g_next;
    }
  }
  for (uint32_t t = 0; t < S.n_tris; ++t) {
    const int id = (int)(n + t);
    if (id != skip) C.triangle(S.tri + 3u * t, id, o, d);
  }
  return C.result();
}

// ---- environment (ray_tracing.comp:239-250) ----------------------------------------
__device__ __forceinline__ float clamp01(float t) { return t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t); }
// GLSL smoothstep(e0, e1, x): t = clamp((x - e0) / (e1 - e0), 0, 1); t*t*(3 - 2t).  Both
// call sites have constant edges, and the quotient is taken as (x - e0) * k with
// k = 1 / (e1 - e0) rounded to fp32 once (2.5 for 0.4, 100 for 0.01): within GLSL's
// 2.5-ulp bound for '/', and the same on CPU and GPU.
__device__ __forceinline__ float smoothstep_rcp(float e0, float k, float x) {
  float t = clamp01((x - e0) * k);
  return (t * t) * (3.0f - 2.0f * t);
}

template <bool STRICT>
__device__ __forceinline__ f3 environment(const Params &P, f3 d) {
  using M = Math<STRICT>;
  float sky_t = M::pow(smoothstep_rcp(0.0f, 2.5f, d.y), 0.35f);          // edges 0, 0.4
  float g2s = smoothstep_rcp(-0.01f, 100.0f, d.y);                       // edges -0.01, 0
  f3 sky = mix3(mk3(P.horizon[0], P.horizon[1], P.horizon[2]),
                mk3(P.zenith[0], P.zenith[1], P.zenith[2]), sky_t);
  // With sunIntensity == 0 (the reference's default, main.cpp:634) and sunFocus > 0 the
  // power is finite for every direction, so the product is exactly +0: skip the pow.
  float sun = 0.0f;
  if (!(P.sun_intensity == 0.0f && P.sun_focus > 0.0f)) {
    float sd = dot(d, mk3(P.sun[0], P.sun[1], P.sun[2]));
    sun = M::pow(sd > 0.0f ? sd : 0.0f, P.sun_focus) * P.sun_intensity;
  }
  f3 comp = mix3(mk3(P.ground[0], P.ground[1], P.ground[2]), sky, g2s);
  float s = sun * (g2s >= 1.0f ? 1.0f : 0.0f);
  return mk3(comp.x + s, comp.y + s, comp.z + s);
}

// ---- scatter (ray_tracing.comp:199-213) ---------------------------------------------
// randomValueNormalDistribution (ray_tracing.comp:121-125) = sqrt(-2 log(u2)) * cos(2 pi u1) with
// the theta draw u1 taken before the rho draw u2; evaluated inside scatter() below.

// ---- albedo texture (extension; sampler state of vulkan_texture.cpp:77-95: LINEAR filter,
// REPEAT addressing, one level) -----------------------------------------------------------
// Coordinates of a hit.  Planar: (point.x, point.z).  Octahedral: the unit normal folded
// onto the square, q = n.xz / (|n.x|+|n.y|+|n.z|), lower half (n.y < 0) mirrored outwards,
// (a, b) = q/2 + 1/2.
__device__ __forceinline__ void texture_coords(const Texture &T, f3 p, f3 n, float &a, float &b) {
  if (T.mapping == 1u) {
    const float s = (__builtin_fabsf(n.x) + __builtin_fabsf(n.y)) + __builtin_fabsf(n.z);
    float qx = ieee_div(n.x, s), qz = ieee_div(n.z, s);
    if (n.y < 0.0f) {
      const float fx = (1.0f - __builtin_fabsf(qz)) * (qx >= 0.0f ? 1.0f : -1.0f);
      const float fz = (1.0f - __builtin_fabsf(qx)) * (qz >= 0.0f ? 1.0f : -1.0f);
      qx = fx;
      qz = fz;
    }
    a = __builtin_fmaf(qx, 0.5f, 0.5f);
    b = __builtin_fmaf(qz, 0.5f, 0.5f);
  } else {
    a = p.x;
    b = p.z;
  }
}

// Bilinear sample, one unit of (u, v) = one repeat: the fractional part of the coordinate
// picks the position, texel centres at half-integers, the two columns and rows wrap,
// bytes became floats through the table when the texture was bound, lerps are
// fma(weight, right - left, left).
__device__ __forceinline__ f3 texture_sample(const Texture &T, float u, float v) {
  const int W = (int)T.width, H = (int)T.height;
  float ur = u - __builtin_floorf(u), vr = v - __builtin_floorf(v);
  if (!(ur >= 0.0f && ur <= 1.0f)) ur = 0.0f;
  if (!(vr >= 0.0f && vr <= 1.0f)) vr = 0.0f;
  const float x = __builtin_fmaf(ur, T.fwidth, -0.5f), y = __builtin_fmaf(vr, T.fheight, -0.5f);
  const float x0 = __builtin_floorf(x), y0 = __builtin_floorf(y);
  const float fx = x - x0, fy = y - y0;
  int ix0 = x0 < 0.0f ? W - 1 : (int)x0, iy0 = y0 < 0.0f ? H - 1 : (int)y0;
  if (ix0 >= W) ix0 = W - 1;
  if (iy0 >= H) iy0 = H - 1;
  const int ix1 = ix0 + 1 == W ? 0 : ix0 + 1, iy1 = iy0 + 1 == H ? 0 : iy0 + 1;
  const float4 c00 = scene_f4(T.texels, (size_t)iy0 * W + ix0), c10 = scene_f4(T.texels, (size_t)iy0 * W + ix1);
  const float4 c01 = scene_f4(T.texels, (size_t)iy1 * W + ix0), c11 = scene_f4(T.texels, (size_t)iy1 * W + ix1);
  float out[3];
  {
    const float top = __builtin_fmaf(fx, c10.x - c00.x, c00.x), bot = __builtin_fmaf(fx, c11.x - c01.x, c01.x);
    out[0] = __builtin_fmaf(fy, bot - top, top);
  }
  {
    const float top = __builtin_fmaf(fx, c10.y - c00.y, c00.y), bot = __builtin_fmaf(fx, c11.y - c01.y, c01.y);
    out[1] = __builtin_fmaf(fy, bot - top, top);
  }
  {
    const float top = __builtin_fmaf(fx, c10.z - c00.z, c00.z), bot = __builtin_fmaf(fx, c11.z - c01.z, c01.z);
    out[2] = __builtin_fmaf(fy, bot - top, top);
  }
  return mk3(out[0], out[1], out[2]);
}

// Hit point, surface normal and material of primitive idx hit at distance dst.
// Sphere: hitPoint = origin + dir*dst, normal = normalize(hitPoint - centre) (:164-165).
// Triangle: normal = normalize(e1 x e2) turned against the ray.  Returns the id the next
// segment must skip (the triangle itself) or -1.
// The normal steers only later segments, never the id of the hit itself, so the fast policy
// normalises it with the hardware reciprocal square root like the other sampling code.
// NO_MESH: the launch has no triangles (their branch is compiled out).  CENTRE: h carries the
// sphere's centre.
template <bool STRICT, bool NO_MESH, bool CENTRE>
__device__ __forceinline__ int hit_surface(const Scene &S, const Hit &h, f3 o, f3 d, f3 &p,
                                           f3 &n, float4 &m_colour, float4 &m_emission,
                                           float4 &m_specular) {
  const int idx = h.idx;
  p = fma3(d, h.dst, o);
  int mat, skip;
  if (NO_MESH || (uint32_t)idx < S.n_spheres) {
    f3 centre;
    if constexpr (CENTRE) {
      centre = mk3(h.cx, h.cy, h.cz);
    } else {
      const float4 g = scene_f4(S.geom, (uint32_t)idx);
      centre = mk3(g.x, g.y, g.z);
    }
    n = Math<STRICT>::normalize(p - centre);
    mat = idx;
    skip = -1;
  } else {
    const float4 *T = S.tri + 3u * ((uint32_t)idx - S.n_spheres);
    const float4 a0 = scene_f4(T, 0), a1 = scene_f4(T, 1), a2 = scene_f4(T, 2);
    n = Math<STRICT>::normalize(cross(mk3(a1.x - a0.x, a1.y - a0.y, a1.z - a0.z),
                             mk3(a2.x - a0.x, a2.y - a0.y, a2.z - a0.z)));
    if (dot(n, d) > 0.0f) n = mk3(-n.x, -n.y, -n.z);
    mat = (int)S.n_spheres;
    skip = idx;
  }
  const ConstF4 *material = (const ConstF4 *)S.material;
  m_colour = load_const(material + 3 * mat + 0);
  m_emission = load_const(material + 3 * mat + 1);
  m_specular = load_const(material + 3 * mat + 2);
  return skip;
}

// On a hit at point p with normal n: new origin and direction, emitted light added
// (:199-211).  Returns isSpecularBounce (0 or 1); the throughput update (:212-213) is
// scatter_throughput, applied by the caller once it knows the surface colour.
template <bool STRICT>
__device__ __forceinline__ float scatter(f3 p, f3 n, float smoothness, float4 m_emission,
                                         float specular_probability, uint32_t &rng, f3 &o, f3 &d,
                                         f3 &light, f3 colour) {
  using M = Math<STRICT>;
  o = p;                                     // (:201)
  // The bounce's seven draws (randomDirection: x(theta, rho) y(theta, rho) z(theta, rho), then the
  // specular test) from seven independently computed states: same values as seven chained steps,
  // but no draw waits for the one before it (1.5 % off the chain of a wave that runs alone).
  const uint32_t s0 = rng;
  rng = pcg_jump(s0, 7);
  float x = M::sqrt(M::neg2log_bits(pcg_output_bits(pcg_jump(s0, 2)))) * M::box_cos_bits(pcg_output_bits(pcg_jump(s0, 1)));
  float y = M::sqrt(M::neg2log_bits(pcg_output_bits(pcg_jump(s0, 4)))) * M::box_cos_bits(pcg_output_bits(pcg_jump(s0, 3)));
  float z = M::sqrt(M::neg2log_bits(pcg_output_bits(pcg_jump(s0, 6)))) * M::box_cos_bits(pcg_output_bits(pcg_jump(s0, 5)));
  const float u_spec = pcg_output_float(rng);
  f3 rd = M::normalize(mk3(x, y, z));
  f3 diffuse = M::normalize(n + rd);         // (:202)
  float dn2 = 2.0f * dot(n, d);              // reflect(I, N) = I - 2 dot(N, I) N (:203)
  f3 spec = mk3(__builtin_fmaf(-dn2, n.x, d.x), __builtin_fmaf(-dn2, n.y, d.y),
                __builtin_fmaf(-dn2, n.z, d.z));
  float is_spec = (specular_probability >= u_spec) ? 1.0f : 0.0f;  // (:204-205)
  d = mix3(diffuse, spec, smoothness * is_spec);                          // (:206-207)
  f3 emitted = mk3(m_emission.x * m_emission.w, m_emission.y * m_emission.w,
                   m_emission.z * m_emission.w);                  // (:209-210)
  light = fma3v(emitted, colour, light);                           // (:211)
  return is_spec;
}

__device__ __forceinline__ f3 scatter_throughput(f3 colour, f3 albedo, float4 m_specular, float is_spec) {
  return colour * mix3(albedo, mk3(m_specular.x, m_specular.y, m_specular.z), is_spec);  // (:212-213)
}

// ---- dielectric spheres (extension: the reference's material, ray_tracing.comp:13-20, has no
// refraction term and its sphere test takes the near root only, :157-161) ----------------
// A sphere with an index of refraction scatters by Snell's law with Schlick's reflectance
// instead of :199-207, drawing ONE random number:
//   ud = normalize(dir); exiting = the ray travels inside this sphere; nl = exiting ? -n : n
//   eta = exiting ? ior : 1/ior; cos_i = min(-dot(ud, nl), 1); sin2 = max(1 - cos_i^2, 0)
//   r0 = ((1 - ior) / (1 + ior))^2; reflectance = r0 + (1 - r0) (1 - cos_i)^5; u = randomValue
//   reflect when eta^2 sin2 > 1 or reflectance > u: dir' = ud - 2 dot(nl, ud) nl
//   else refract: perp = eta (ud + cos_i nl); dir' = perp - sqrt(|1 - perp.perp|) nl
// Returns true when the ray was refracted.
template <bool STRICT>
__device__ __forceinline__ bool dielectric_scatter(f3 n, float ior, bool exiting, uint32_t &rng, f3 &d) {
  using M = Math<STRICT>;
  const f3 ud = M::normalize(d);
  const f3 nl = exiting ? mk3(-n.x, -n.y, -n.z) : n;
  const float eta = exiting ? ior : ieee_div(1.0f, ior);
  float cos_i = -dot(ud, nl);
  if (cos_i > 1.0f) cos_i = 1.0f;
  float sin2 = __builtin_fmaf(-cos_i, cos_i, 1.0f);
  if (sin2 < 0.0f) sin2 = 0.0f;
  const bool cannot_refract = (eta * eta) * sin2 > 1.0f;
  float r0 = ieee_div(1.0f - ior, 1.0f + ior);
  r0 = r0 * r0;
  const float x = 1.0f - cos_i;
  const float x2 = x * x;
  const float reflectance = __builtin_fmaf(1.0f - r0, (x2 * x2) * x, r0);
  const float u = pcg_float(rng);
  if (cannot_refract || reflectance > u) {
    const float dn2 = 2.0f * dot(nl, ud);
    d = mk3(__builtin_fmaf(-dn2, nl.x, ud.x), __builtin_fmaf(-dn2, nl.y, ud.y),
            __builtin_fmaf(-dn2, nl.z, ud.z));
    return false;
  }
  const f3 perp = fma3(nl, cos_i, ud) * eta;
  const float par = -M::sqrt(__builtin_fabsf(1.0f - dot(perp, perp)));
  d = fma3(nl, par, perp);
  return true;
}

// ---- per-frame pixel epilogue (ray_tracing.comp:98-109) ------------------------------
__device__ __forceinline__ uint32_t unorm8(float v) {
  if (!(v > 0.0f)) return 0u;  // NaN and negatives store 0
  if (v > 1.0f) v = 1.0f;
  return (uint32_t)__builtin_rintf(v * 255.0f);
}

// mean over samples, gamma, float accumulation, and the shader's progressive blend
// carried in registers: img holds the 8-bit texel as the shader would load it.
__device__ __forceinline__ void frame_epilogue(const Params &P, f3 total, float frame_f,
                                               uint32_t linear_accum, float4 &acc, uchar4 &img) {
  f3 mean = mk3(ieee_div(total.x, P.spp_f), ieee_div(total.y, P.spp_f), ieee_div(total.z, P.spp_f));
  f3 g = mk3(ieee_sqrt(mean.x), ieee_sqrt(mean.y), ieee_sqrt(mean.z));
  f3 c = linear_accum ? mean : g;
  acc.x += c.x;
  acc.y += c.y;
  acc.z += c.z;
  acc.w += 1.0f;
  float w = ieee_div(1.0f, frame_f + 1.0f);
  float omw = 1.0f - w;
  img.x = (unsigned char)unorm8(__builtin_fmaf(g.x, w, ieee_div((float)img.x, 255.0f) * omw));
  img.y = (unsigned char)unorm8(__builtin_fmaf(g.y, w, ieee_div((float)img.y, 255.0f) * omw));
  img.z = (unsigned char)unorm8(__builtin_fmaf(g.z, w, ieee_div((float)img.z, 255.0f) * omw));
  img.w = (unsigned char)unorm8(__builtin_fmaf(1.0f, w, 1.0f * omw));
}

// 16x16 pixel tiles per 256-thread block; each wave64 owns an 8x8 sub-tile so the
// lanes of a wave see neighbouring pixels (similar path lengths, shared cache lines).
__device__ __forceinline__ void pixel_of_thread(uint32_t &px, uint32_t &py) {
  uint32_t t = threadIdx.x;
  uint32_t wave = t >> 6, lane = t & 63u;
  px = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
  py = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
}

// the same layout for a tile picked from a launch-order table
// t: index of the thread within the tile's 256 (threadIdx.x when a block is a tile)
__device__ __forceinline__ void pixel_of_tile(uint32_t tile, uint32_t tiles_x, uint32_t t, uint32_t &px, uint32_t &py) {
  const uint32_t wave = t >> 6, lane = t & 63u;
  px = (tile % tiles_x) * 16u + (wave & 1u) * 8u + (lane & 7u);
  py = (tile / tiles_x) * 16u + (wave >> 1) * 8u + (lane >> 3);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t other = __shfl_xor(v, off, 64);
    v = other > v ? other : v;
  }
  return v;
}

// wave-level sum of a 32-bit count, result valid in every lane
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// a wave's (or block's) path segments into the launch's count; index: any number that differs
// between the waves that finish at about the same time
__device__ __forceinline__ void add_rays(const Outputs &O, uint32_t index, uint32_t n) {
  unsigned long long *p = O.ray_slots ? O.ray_slots + (size_t)(index & (kRaySlots - 1u)) * (kRaySlotStride / 8u)
                                      : O.rays;
  atomicAdd(p, (unsigned long long)n);
}
__device__ __forceinline__ uint32_t wave_index() {
  return (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
}

}  // namespace rt
