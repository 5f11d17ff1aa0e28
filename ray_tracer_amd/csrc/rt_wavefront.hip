#include "rt_wavefront.h"
namespace rt {
int WavefrontState::render(bool, bool, const Params &, const Scene &, const FrameSeq &, const Outputs &,
                           hipStream_t, uint32_t *, std::string *err) {
  if (err) *err = "wavefront mode is not available in this build";
  return -3;
}
void WavefrontState::release() {}
}  // namespace rt
