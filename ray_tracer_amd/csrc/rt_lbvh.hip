#include "rt_lbvh.h"
namespace rt {
int Lbvh::build(const float4 *, const float *, uint32_t, hipStream_t, std::string *err) {
  if (err) *err = "LBVH build is not available in this build";
  return -3;
}
void Lbvh::release() {}
}  // namespace rt
