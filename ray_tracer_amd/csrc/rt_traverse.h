// Closest-hit search through the linear BVH.
//
// The result must be the one the shader's brute-force loop (ray_tracing.comp:172-190)
// gives: the smallest dst, and among equal dst the LOWEST sphere index (its strict
// "<" keeps the first one found in index order).  Traversal order is arbitrary, so
// ties are resolved explicitly on the original index.
//
// Each lane keeps its traversal stack in LDS, entry-major ([entry][thread]): lane l
// always touches bank l mod 32 whatever its depth, and lanes l / l+32 belong to
// different LDS lane groups, so pushes and pops never conflict.
#pragma once

#include "rt_shading.h"

namespace rt {

constexpr int kBlockThreads = 256;
#ifndef RT_LEAF_SELECT
#define RT_LEAF_SELECT 1   // the sphere tests of the tree search update the best hit by selects (see Closest<true>::sphere): 4K glass batch 128.3 -> 127.2 ms
#endif

// Entry distance of the ray into a box, +inf when it misses or the box is beyond limit.
// Plane distances are taken as fma(plane, 1/d, -(o/d)) with o/d rounded once per ray: one
// instruction per plane instead of two.  Against (plane - o) * (1/d) this adds an absolute
// error of half an ulp of o/d, i.e. at most 6e-8 |o| in space, which is 1.5 % of the
// smallest margin the leaf boxes carry (4e-6 D for triangles, with |o| <= D; the sphere
// margin 4.8e-7 D^2 / r is larger still).
// 1/d is kept finite (slab_rcp): with an infinite one, a zero direction component turns the plane
// terms of a box that straddles the coordinate's origin into -inf and NaN (plane * inf - o * inf),
// the slab's far distance into -inf, and the box -- typically the root -- into a false miss.
// Clamped to +-2^64 the component behaves like a direction of 2^-64: the slab of a box the ray
// runs inside of comes out as (-huge, +huge) and stops constraining the test, and a box the ray
// runs beside is missed, as it should be.
__device__ __forceinline__ float slab_rcp(float d) {
  return __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d), -0x1p64f, 0x1p64f);
}
__device__ __forceinline__ float slab_entry(const float lo[3], const float hi[3], f3 oi, f3 inv_d,
                                            float limit) {
  float x0 = __builtin_fmaf(lo[0], inv_d.x, -oi.x), x1 = __builtin_fmaf(hi[0], inv_d.x, -oi.x);
  float y0 = __builtin_fmaf(lo[1], inv_d.y, -oi.y), y1 = __builtin_fmaf(hi[1], inv_d.y, -oi.y);
  float z0 = __builtin_fmaf(lo[2], inv_d.z, -oi.z), z1 = __builtin_fmaf(hi[2], inv_d.z, -oi.z);
  float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
  float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), limit));
  return tn <= tf ? tn : __builtin_inff();
}

// one 64-byte node, read through the constant address space like the rest of the scene
__device__ __forceinline__ BvhNode load_node(const BvhNode *nodes, int node) {
  const float4 *p = (const float4 *)(nodes + node);
  const float4 a = scene_f4(p, 0), b = scene_f4(p, 1), c = scene_f4(p, 2), e = scene_f4(p, 3);
  BvhNode nd;
  nd.lo0[0] = a.x; nd.lo0[1] = a.y; nd.lo0[2] = a.z; nd.child0 = __float_as_int(a.w);
  nd.hi0[0] = b.x; nd.hi0[1] = b.y; nd.hi0[2] = b.z; nd.count0 = __float_as_int(b.w);
  nd.lo1[0] = c.x; nd.lo1[1] = c.y; nd.lo1[2] = c.z; nd.child1 = __float_as_int(c.w);
  nd.hi1[0] = e.x; nd.hi1[1] = e.y; nd.hi1[2] = e.z; nd.count1 = __float_as_int(e.w);
  return nd;
}

// leaf at sorted position pos: a sphere (geometry in leaf order) or a triangle of the mesh.
// skip: the triangle the ray starts on is left out; in DIEL kernels the dielectric sphere it
// travels inside is tested with its far root (see closest_hit_brute).
template <bool DIEL = false, bool APPROX>
__device__ __forceinline__ void leaf_candidate(const Scene &S, int pos, f3 o, f3 d, float a, int skip,
                                               Closest<APPROX> &C) {
  const int prim = (int)scene_u32(S.sorted_prim, (uint32_t)pos);
  if ((uint32_t)prim < S.n_spheres) {
    if (DIEL && prim == skip) C.sphere_far(scene_f4(S.sorted_geom, (uint32_t)pos), prim, o, d, a);
    else C.template sphere<false, false, RT_LEAF_SELECT != 0>(scene_f4(S.sorted_geom, (uint32_t)pos), prim, o, d, a);
  } else if (prim != skip) {
    C.triangle(S.tri + 3u * ((uint32_t)prim - S.n_spheres), prim, o, d);
  }
}

// every primitive of a leaf reference (a run of sorted positions, see make_leaf_ref)
template <bool DIEL = false, bool APPROX>
__device__ __forceinline__ uint32_t leaf_range(const Scene &S, int ref, f3 o, f3 d, float a, int skip,
                                               Closest<APPROX> &C) {
  const uint32_t first = leaf_ref_first(ref), count = leaf_ref_count(ref);
  for (uint32_t k = 0; k < count; ++k) leaf_candidate<DIEL>(S, (int)(first + k), o, d, a, skip, C);
  return count;
}

// the oversized spheres kept out of the tree, tested before it
template <bool DIEL, bool APPROX>
__device__ __forceinline__ void large_candidates(const Scene &S, f3 o, f3 d, float a, int skip,
                                                 Closest<APPROX> &C) {
  for (uint32_t i = 0; i < S.n_large; ++i) {
    const int prim = (int)scene_u32(S.large_prim, i);
    if (DIEL && prim == skip) C.sphere_far(scene_f4(S.large_geom, i), prim, o, d, a);
    else C.template sphere<false, false, RT_LEAF_SELECT != 0>(scene_f4(S.large_geom, i), prim, o, d, a);
  }
}

// COUNT adds two per-lane tallies (nodes fetched, primitives tested) for the roofline's
// bytes-per-segment figure; the search itself is identical.
// `stack` points at S.stack_entries * kBlockThreads ints of LDS owned by the calling kernel
// (entry-major, sized by the launch from the height of the tree); it is scratch, so a kernel
// may reuse that memory between searches.
template <bool COUNT, bool APPROX, bool DIEL = false>
__device__ __forceinline__ Hit closest_hit_bvh(const Scene &S, f3 o, f3 d, int *stack, int skip,
                                               uint32_t *n_nodes = nullptr,
                                               uint32_t *n_prims = nullptr) {
  const int tid = threadIdx.x;
  const float a = dot(d, d);
  Closest<APPROX> C;
  C.reset(a);
  const f3 inv_d = mk3(slab_rcp(d.x), slab_rcp(d.y),
                       slab_rcp(d.z));
  const f3 oi = mk3(o.x * inv_d.x, o.y * inv_d.y, o.z * inv_d.z);
  large_candidates<DIEL>(S, o, d, a, skip, C);
  if constexpr (COUNT) *n_prims += S.n_large;
  if (S.n_spheres + S.n_tris == 1u) {
    if (S.n_large == 0u) {
      leaf_candidate<DIEL>(S, 0, o, d, a, skip, C);
      if constexpr (COUNT) ++*n_prims;
    }
  } else {
    int sp = 0;
    int node = 0;
    while (true) {
      const BvhNode nd = load_node(S.nodes, node);
      if constexpr (COUNT) ++*n_nodes;
      // the distance limit carries slack: box arithmetic rounds, the primitive test decides
      const float best_dst = C.dst();
      const float limit = __builtin_fmaf(best_dst, 1.0e-4f, best_dst);
      float t0 = slab_entry(nd.lo0, nd.hi0, oi, inv_d, limit);
      float t1 = slab_entry(nd.lo1, nd.hi1, oi, inv_d, limit);
      int c0 = nd.child0, c1 = nd.child1;
      if (t0 < __builtin_inff() && c0 < 0) {
        if (c0 != kNullLeaf) {
          const uint32_t tested = leaf_range<DIEL>(S, c0, o, d, a, skip, C);
          if constexpr (COUNT) *n_prims += tested;
        }
        t0 = __builtin_inff();
      }
      if (t1 < __builtin_inff() && c1 < 0) {
        if (c1 != kNullLeaf) {
          const uint32_t tested = leaf_range<DIEL>(S, c1, o, d, a, skip, C);
          if constexpr (COUNT) *n_prims += tested;
        }
        t1 = __builtin_inff();
      }
      const bool go0 = t0 < __builtin_inff(), go1 = t1 < __builtin_inff();
      if (go0 && go1) {
        const bool first0 = t0 <= t1;
        stack[(sp++) * kBlockThreads + tid] = first0 ? c1 : c0;
        node = first0 ? c0 : c1;
      } else if (go0 || go1) {
        node = go0 ? c0 : c1;
      } else {
        if (sp == 0) break;
        node = stack[(--sp) * kBlockThreads + tid];
      }
    }
  }
  return C.result();
}

// closest hit through whichever structure the kernel was built for
template <bool BVH, bool APPROX, bool GROUPED = false, bool DIEL = false, bool SMALL = false>
__device__ __forceinline__ Hit closest_hit(const Scene &S, f3 o, f3 d, int *stack, int skip) {
  if constexpr (BVH) return closest_hit_bvh<false, APPROX, DIEL>(S, o, d, stack, skip);
  else return closest_hit_brute<APPROX, GROUPED, DIEL, SMALL>(S, o, d, skip);
}

// ---- a search whose state outlives the loop iteration ----------------------------------
// Kernels that let the WAVE choose, every iteration, which step to run for the lanes that
// need it (k_mega_sched, and the scheduled extend and drain stages of the wavefront
// pipeline) keep one Search per lane: current node, stack depth, best hit so far.  Steps:
//   inner  fetch one node, test both child boxes, descend / push / pop
//   leaf   test one primitive, pop
// and a kernel-specific step for lanes whose search is over.  The step with the largest
// weight x waiting lanes runs; a descending lane takes up to RT_SCHED_INNER_REPS inner steps
// per decision.
#ifndef RT_SCHED_W_INNER
#define RT_SCHED_W_INNER 2
#endif
#ifndef RT_SCHED_W_LEAF
#define RT_SCHED_W_LEAF 5
#endif
#ifndef RT_SCHED_W_SHADE
#define RT_SCHED_W_SHADE 1   // 2:5:1 against 2:5:2 at 3840x2160: 10k spheres 5.54 / 5.65 ms, terrain 15.25 / 15.36
#endif
#ifndef RT_SCHED_INNER_REPS
#define RT_SCHED_INNER_REPS 4   // inner steps per scheduling decision (2: 5.24, 4: 5.08, 8: 5.26 ms)
#endif

constexpr int kSearchOver = (int)kNullLeaf;  // never a real child reference
// a lane that has no search at all (k_mega_sched: its pixel is finished or outside the image);
// leaf references are ~x with x < 2^30, i.e. the unsigned words from kLeafRefLowest up
constexpr int kSearchIdle = (int)kNullLeaf + 1;
constexpr uint32_t kLeafRefLowest = 0xC0000000u;
static_assert(kLeafCountShift + 3 <= 30 && RT_LEAF_MAX_PRIMS <= 8, "leaf references must stay above kLeafRefLowest");

template <bool APPROX>
struct Search {
  int node;  // >= 0 inner node, kSearchOver, otherwise a leaf reference
  int sp;
  float a;
  f3 inv_d, oi;  // 1 / d and o / d
  Closest<APPROX> best;
};

template <bool DIEL = false, bool APPROX>
__device__ __forceinline__ void search_begin(const Scene &S, f3 o, f3 d, int skip, Search<APPROX> &Q) {
  Q.a = dot(d, d);
  Q.best.reset(Q.a);
  Q.inv_d = mk3(slab_rcp(d.x), slab_rcp(d.y), slab_rcp(d.z));
  Q.oi = mk3(o.x * Q.inv_d.x, o.y * Q.inv_d.y, o.z * Q.inv_d.z);
  large_candidates<DIEL>(S, o, d, Q.a, skip, Q.best);
  Q.sp = 0;
  if (S.n_spheres + S.n_tris == 1u) Q.node = S.n_large ? kSearchOver : ~0;
  else Q.node = 0;
}

// STRIDE: threads per block of the calling kernel (the stack is [entry][thread])
template <int STRIDE = kBlockThreads, bool APPROX>
__device__ __forceinline__ int search_pop(Search<APPROX> &Q, const int *stack) {
  if (Q.sp == 0) return kSearchOver;
  return stack[(--Q.sp) * STRIDE + threadIdx.x];
}

template <int STRIDE = kBlockThreads, bool APPROX>
__device__ __forceinline__ void search_inner_step(const Scene &S, Search<APPROX> &Q, int *stack) {
  const BvhNode nd = load_node(S.nodes, Q.node);
  const float best_dst = Q.best.dst();
  const float limit = __builtin_fmaf(best_dst, 1.0e-4f, best_dst);
  const float t0 = slab_entry(nd.lo0, nd.hi0, Q.oi, Q.inv_d, limit);
  const float t1 = slab_entry(nd.lo1, nd.hi1, Q.oi, Q.inv_d, limit);
  const int c0 = nd.child0, c1 = nd.child1;
  const bool go0 = t0 < __builtin_inff() && c0 != (int)kNullLeaf;
  const bool go1 = t1 < __builtin_inff() && c1 != (int)kNullLeaf;
  if (go0 && go1) {
    const bool first0 = t0 <= t1;
    stack[(Q.sp++) * STRIDE + threadIdx.x] = first0 ? c1 : c0;
    Q.node = first0 ? c0 : c1;
  } else if (go0 || go1) {
    Q.node = go0 ? c0 : c1;
  } else {
    Q.node = search_pop<STRIDE>(Q, stack);
  }
}

// static LDS of kernels that are only ever built without a tree (1 word keeps the array legal)
template <bool BVH>
constexpr int bvh_stack_words() {
  static_assert(!BVH, "tree kernels size their stack at launch (dynamic LDS)");
  return 1;
}

// One hit bounce: surface at the hit, then the shader's scatter.  Returns the primitive the
// next segment must treat specially (see closest_hit_brute's skip).  EXT: which extensions the
// kernel carries.  kExtTexture (launched while a texture is bound): the material colour
// replaced by a texture sample -- its coordinates are taken at the hit and the texels are
// fetched last, when little else is live.  kExtDielectric (launched while the scene has
// dielectric spheres): those, and the texture if one is bound as well.  kExtNone: neither
// (the 8K textured frame runs 5 % slower through a kernel that also carries the dielectric code).
// from: the skip value the segment that found this hit was traced with.
// NO_MESH and CENTRE as in hit_surface.
constexpr int kExtNone = 0, kExtTexture = 1, kExtDielectric = 2;
template <bool STRICT, int EXT, bool APPROX = false, bool CENTRE = false>
__device__ __forceinline__ int shade_hit(const Scene &S, const Hit &h, int from, uint32_t &rng, f3 &o,
                                         f3 &d, f3 &light, f3 &colour) {
  f3 p, n;
  float4 mc, me, ms;
  const int skip = hit_surface<STRICT, APPROX, CENTRE>(S, h, o, d, p, n, mc, me, ms);
  if constexpr (EXT == kExtDielectric) {
    if (S.ior && (uint32_t)h.idx < S.n_spheres) {
      const float ior = scene_f32(S.ior, (uint32_t)h.idx);
      if (ior > 0.0f) {
        const bool exiting = from == h.idx;
        o = p;
        const bool refracted = dielectric_scatter<STRICT>(n, ior, exiting, rng, d);
        light = fma3v(mk3(me.x * me.w, me.y * me.w, me.z * me.w), colour, light);
        colour = colour * mk3(mc.x, mc.y, mc.z);
        // inside the sphere afterwards: refracted into it, or reflected back while inside
        return refracted != exiting ? h.idx : -1;
      }
    }
  }
  bool textured = false;
  float tu = 0.0f, tv = 0.0f;
  if constexpr (EXT != kExtNone) {
    textured = S.tex && (uint32_t)h.idx - S.tex->first_prim < S.tex->n_prims;
    if (textured) {
      texture_coords(*S.tex, p, n, tu, tv);
      tu *= S.tex->scale_u;
      tv *= S.tex->scale_v;
    }
  }
  const float is_spec = scatter<STRICT>(p, n, mc.w, me, ms.w, rng, o, d, light, colour);
  f3 albedo = mk3(mc.x, mc.y, mc.z);
  if constexpr (EXT != kExtNone) {
    // a specular bounce gives the colour weight 0 and every texel is finite: skip the fetch
    if (textured && is_spec == 0.0f) albedo = texture_sample(*S.tex, tu, tv);
  }
  colour = scatter_throughput(colour, albedo, ms, is_spec);
  return skip;
}

}  // namespace rt
