// Wavefront path tracer: raygen / extend / shade stages over compacted queues.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "rt_device_scene.h"

namespace rt {

struct WavefrontState {
  int render(bool strict, bool bvh, const Params &P, const Scene &S, const FrameSeq &F,
             const Outputs &O, hipStream_t stream, uint32_t *launches, std::string *err);
  void release();
};

}  // namespace rt
