// Wavefront path tracer: raygen / extend / shade stages over compacted queues.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "rt_device_scene.h"

namespace rt {

struct WavefrontState {
  struct Impl;
  Impl *impl = nullptr;  // path-state arrays and queues, sized for the image on first use
  // once at most this many pixels are still active, one drain launch finishes them in
  // registers (0: stay wavefront to the end)
  uint32_t drain_threshold = 0;
  // Experiment (RT_WF_REGROUP=1, LBVH scenes): before every extend stage the live pixels are sorted by
  // (cell of the ray origin on a 16^3 grid over the scene, direction octant), and the stage walks them in
  // that order instead of pixel order.  The path state stays indexed by pixel, so results do not change.
  bool regroup = false;
  float regroup_lo[3] = {0.f, 0.f, 0.f}, regroup_hi[3] = {1.f, 1.f, 1.f};

  int render(bool strict, bool bvh, const Params &P, const Scene &S, const FrameSeq &F,
             const Outputs &O, hipStream_t stream, uint32_t *launches, std::string *err);
  void release();
};

}  // namespace rt
