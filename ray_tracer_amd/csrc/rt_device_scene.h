// Device-side views of parameters, scene and outputs (plain structs passed by value
// as kernel arguments, so uniform fields live in scalar registers).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

// Everything the shader reads from its uniform block (ray_tracing.comp:38-54),
// with the loop-invariant work hoisted by the host: both matrix inverses
// (ray_tracing.comp:230,233 recompute them per sample) and integer trip counts.
struct Params {
  float inv_view[16];  // column-major
  float inv_proj[16];
  float viewport_w, viewport_h;
  float cam_x, cam_y, cam_z;
  float spp_f;     // renderSettings.x as stored (the divisor of the mean)
  int n_spp;       // iterations of "for (i = 0; i < renderSettings.x; i++)"
  int n_bounce;    // iterations of "for (i = 0; i < renderSettings.y; i++)"
  float ground[3], horizon[3], zenith[3], sun[3];
  float sun_focus, sun_intensity;
  float defocus, diverge;
  uint32_t width, height;
  float fwidth, fheight;    // float(imageSize)
  float defocus_over_w;     // defocus / fwidth   (fast-math jitter scale)
  float diverge_over_w;     // diverge / fwidth
  float rcp_w, rcp_h;       // 1 / fwidth, 1 / fheight
};

// Node of the linear BVH: an internal node stores the bounds of its two children,
// so one 64-byte fetch decides both.  child < 0 encodes a leaf: ~child is the
// position in the Morton-sorted primitive order.
struct BvhNode {
  float lo0[3];
  int32_t child0;
  float hi0[3];
  int32_t count0;   // primitives under child0 (unused by traversal, kept for stats)
  float lo1[3];
  int32_t child1;
  float hi1[3];
  int32_t count1;
};

// Most traversal stack entries a launch can provide per lane: 64 KB of LDS for a 256-thread
// block.  A Karras tree over 30-bit codes is at most 30 + log2(primitives sharing one cell)
// levels high, so taller trees need > 2^34 primitives in one cell; such a scene is traced by
// brute force.
constexpr int kBvhStackEntries = 64;
constexpr int kMaxLargeSpheres = 16;  // oversized spheres tested outside the tree
constexpr int32_t kNullLeaf = (int32_t)0x80000000;  // leaf slot of a sphere kept out of the tree

// A leaf reference (child < 0, not kNullLeaf) names a run of consecutive positions of the
// Morton-sorted primitive order: every subtree of at most kLeafMaxPrims primitives is folded
// into one leaf, so the search tests up to that many primitives where it would otherwise walk
// one or two more levels of nodes.  ~child = first position | (count - 1) << kLeafCountShift.
#ifndef RT_LEAF_MAX_PRIMS
#define RT_LEAF_MAX_PRIMS 2  // 1: 4.76 / 143 ms, 2: 4.52 / 142, 4: 4.86 / 145, 8: 5.95 / 150 (terrain 1080p frame / 10k glass spheres 4K batch)
#endif
constexpr int kLeafMaxPrims = RT_LEAF_MAX_PRIMS;   // 1 .. 8
constexpr int kLeafCountShift = 27;                // positions < 2^27 (2^26 spheres + 2^26 triangles)
__host__ __device__ inline int32_t make_leaf_ref(uint32_t first, uint32_t count) {
  return ~(int32_t)(first | ((count - 1u) << kLeafCountShift));
}
__host__ __device__ inline uint32_t leaf_ref_first(int32_t ref) { return (uint32_t)~ref & ((1u << kLeafCountShift) - 1u); }
__host__ __device__ inline uint32_t leaf_ref_count(int32_t ref) { return ((uint32_t)~ref >> kLeafCountShift) + 1u; }

// Albedo texture bound to a range of primitive ids (texels == null: none).
struct Texture {
  // width * height texels, row 0 = v 0, decoded to linear rgb floats when the texture is bound
  // (through the 256-entry byte -> linear table): a sample is then four independent 16-byte
  // fetches instead of four texel words followed by twelve table look-ups that wait for them
  const float4 *texels;
  uint32_t width, height;
  float fwidth, fheight;
  float scale_u, scale_v;
  uint32_t first_prim, n_prims;
  uint32_t mapping;         // 0: planar (point.x, point.z), 1: octahedral from the normal
};

// Primitive ids: spheres 0 .. n_spheres-1, then the triangles of the mesh.
struct Scene {
  const float4 *geom;       // per sphere: centre.xyz, radius*radius   (original order)
  const float4 *material;   // colour, emission, specular (3 x float4) per sphere, then one
                            // entry for the whole mesh at index n_spheres
  uint32_t n_spheres;
  const float4 *tri;        // per triangle: v0, v1, v2 (3 x float4); null without a mesh
  uint32_t n_tris;
  // LBVH (null when brute force is used)
  const BvhNode *nodes;     // n_prims-1 internal nodes, root = 0
  const uint32_t *sorted_prim;  // leaf position -> primitive id
  const float4 *sorted_geom;    // geom[] permuted into leaf order (sphere leaves only)
  const float4 *large_geom;     // oversized spheres, tested before the tree
  const uint32_t *large_prim;
  uint32_t n_large;
  uint32_t stack_entries;       // traversal stack entries a search can need (tree height, at most kBvhStackEntries)
  const Texture *tex;       // device copy of the texture binding; null: none (kept behind a
                            // pointer so untextured launches carry two scalar registers for it)
  const float *ior;         // per sphere: index of refraction of a dielectric sphere, <= 0 for the
                            // reference's material; null: no dielectric spheres
};

struct FrameSeq {
  uint32_t first, count, stride;
};

constexpr uint32_t kRaySlots = 1024, kRaySlotStride = 128;
struct Outputs {
  float4 *accum;            // rgb += frame colour, a += 1, per frame
  uchar4 *rgba8;            // the shader's own 8-bit progressive image
  int32_t *ids;             // first-segment hit of sample 0 (last frame wins)
  unsigned long long *rays; // += closest-hit searches ([1], [2]: traversal tallies of the counting kernels)
  // Optional: kRaySlots counters, kRaySlotStride bytes apart, that the waves add their segment
  // counts to instead of rays[0]; whoever reads the count adds them up.  Atomics on ONE address
  // retire at ~80 per microsecond on the whole device, which bounded every launch of many short
  // waves (1280x720 at 1 spp: 14,400 waves, 0.23 ms with the single counter, 0.075 ms with these).
  unsigned long long *ray_slots;
  uint32_t linear_accum;    // 1: accumulate the linear mean, 0: the gamma-encoded colour
  // Launch order of the 16x16-pixel tiles (k_mega_regen, k_mega_sched): block b renders tile
  // tile_order[b].  The order is heaviest-first by the cost the previous launch of this
  // context measured (tile_cost: the most path segments any pixel of the tile traced), so
  // the few tiles whose pixels run long serial chains of samples start first instead of
  // last.  Both null: row-major order, nothing recorded.
  const uint32_t *tile_order;
  uint32_t *tile_cost;
  uint32_t tiles_x, n_tiles;
  uint32_t chain_ranks;     // k_mega_regen, mixed form: blocks below this rank run the plain loop
  uint32_t sparse_ranks;    // k_mega_regen, mixed form: tiles below this rank are spread over more waves,
  uint32_t sparse_shift;    //   2^sparse_shift (2, 4 or 8) per wave of the tile
  uint32_t probe;           // 1: a cost probe, the tile kernels write tile_cost and nothing else
};

}  // namespace rt
