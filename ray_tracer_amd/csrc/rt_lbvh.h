// Linear BVH over spheres and mesh triangles, built on the GPU (Morton codes, radix sort, Karras hierarchy).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "rt_device_scene.h"

namespace rt {

struct Lbvh {
  // device arrays read by traversal
  BvhNode *nodes = nullptr;        // n-1 internal nodes, root = 0
  uint32_t *sorted_prim = nullptr; // leaf position -> original sphere index
  float4 *sorted_geom = nullptr;   // (centre, r*r) in leaf order
  float4 *large_geom = nullptr;    // oversized spheres kept out of the tree (centre, r*r)
  uint32_t *large_prim = nullptr;  // their original indices, ascending
  uint32_t n_large = 0;
  uint32_t n_nodes = 0;
  uint32_t capacity = 0;           // spheres the arrays above can hold
  uint32_t height = 0;             // longest root-to-leaf path in edges = traversal stack bound
  float scene_lo[3] = {0.f, 0.f, 0.f}, scene_hi[3] = {0.f, 0.f, 0.f};  // bounds of sphere centres and mesh vertices
  uint32_t refit_listed = 0;       // last build: nodes that straddle the tiles of the first refit launch,
  uint32_t refit_rounds = 0;       // and the rounds on global memory the last launch needed for them (normally 0)
  float last_build_ms = 0.f;     // use build_ms()
  bool timing_pending = false;
  float build_ms();              // device time of the last build (waits for it if it is still running)

  struct Work;
  Work *work = nullptr;            // build workspace, kept between builds

  // d_geom: device (centre, r*r) per sphere; h_spheres64: the 64-byte host records;
  // d_tri: device v0,v1,v2 (float4 each) per triangle; mesh_lo/hi: bounds of its vertices;
  // cam: camera position (ray origins are assumed to stay near the camera and the scene).
  int build(const float4 *d_geom, const float *h_spheres64, uint32_t n_spheres, const float4 *d_tri,
            const float mesh_lo[3], const float mesh_hi[3], uint32_t n_tris, const float cam[3],
            hipStream_t stream, std::string *err);
  int download(std::vector<BvhNode> *out_nodes, std::vector<uint32_t> *out_prims) const;
  void release();
};

// The build's radix sort for other users (the wavefront pipeline's ray regrouping experiment)
hipError_t radix_sort_pairs(uint32_t *const keys[2], uint32_t *const vals[2], uint32_t n, int passes,
                            uint32_t *hist, uint32_t *scan_sums, hipStream_t stream);
void radix_sort_scratch(uint32_t n, size_t *hist_bytes, size_t *sums_bytes);

}  // namespace rt
