// Host-callable launchers of the device stages (one .hip file each).
#pragma once

#include <hip/hip_runtime.h>

#include "rt_device_scene.h"

namespace rt {

// rt_megakernel.hip
hipError_t launch_megakernel(bool nested, bool strict, bool bvh, const Params &P, const Scene &S,
                             const FrameSeq &F, const Outputs &O, hipStream_t stream);

// fast math, LBVH, nested loops, plus node/sphere tallies in O.rays[1], O.rays[2]
hipError_t launch_megakernel_counting(const Params &P, const Scene &S, const FrameSeq &F,
                                      const Outputs &O, hipStream_t stream);

// rt_util.hip
hipError_t launch_resolve(const float4 *accum, uchar4 *rgba8, uint32_t n_pixels,
                          uint32_t linear_accum, hipStream_t stream);
// expands an indexed mesh to per-triangle vertices; report (7 device words) receives the
// position of the first out-of-range index (or ~0) and the ordered-bit bounds of the vertices
hipError_t launch_expand_triangles(const float *vertices, uint32_t n_vertices, const uint32_t *indices,
                                   uint32_t n_tris, float4 *tri, uint32_t *report, hipStream_t stream);
// 64-byte sphere records resident on the device -> geometry (n_geom > n entries, zero-padded)
// and material arrays
hipError_t launch_repack_spheres(const void *spheres64, uint32_t n, uint32_t n_geom, float4 *geom,
                                 float4 *material, hipStream_t stream);
// texel bytes -> linear rgb floats through a 256-entry table (w = 0)
hipError_t launch_decode_texels(const uchar4 *texels, const float *lut, float4 *out, size_t n, hipStream_t stream);
// adds the kRaySlots spread counters up into *total and clears them
hipError_t launch_fold_rays(unsigned long long *slots, unsigned long long *total, hipStream_t stream);
// tile costs of the last launch -> heaviest-first launch order; clears the costs
// order_near (optional, needs tiles_x): the same with every tile ranked by the largest cost among
// itself and its eight neighbours
hipError_t launch_tile_order(uint32_t *cost, uint32_t *order, uint32_t n_tiles, hipStream_t stream,
                             uint32_t *order_near = nullptr, uint32_t tiles_x = 0);
hipError_t launch_debug_eval(int op, const float *a, const float *b, float *out, size_t n,
                             hipStream_t stream);

}  // namespace rt
