// Linear BVH over spheres, built on the GPU (Morton codes, radix sort, Karras hierarchy).
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "rt_device_scene.h"

namespace rt {

struct Lbvh {
  BvhNode *nodes = nullptr;
  uint32_t *sorted_prim = nullptr;
  float4 *sorted_geom = nullptr;
  uint32_t n_nodes = 0;
  float last_build_ms = 0.f;

  int build(const float4 *d_geom, const float *h_spheres64, uint32_t n, hipStream_t stream,
            std::string *err);
  void release();
};

}  // namespace rt
