// Megakernels: one lane per pixel, whole frame batch in one launch.
//
//   k_mega_nested  the shader's own loop nest (frames > samples > bounces), kept as the
//                  structural baseline and for degenerate trip counts;
//   k_mega_regen   the same per-lane program flattened into ONE loop whose body is
//                  "trace one path segment": a lane whose path ended draws its next
//                  sample in the same iteration, so every active lane does an
//                  intersection every iteration instead of idling until the
//                  longest path of the wave finishes.  The per-lane order of RNG
//                  draws and arithmetic is unchanged, hence identical results.
#include "rt_kernels.h"
#include "rt_shading.h"
#include "rt_traverse.h"

#ifndef RT_MEGA_WAVES
#define RT_MEGA_WAVES 1
#endif

namespace rt {

// COUNT (only with BVH): also tallies LBVH nodes fetched and spheres tested into
// O.rays[1] and O.rays[2].
template <bool STRICT, bool BVH, bool COUNT = false>
__global__ __launch_bounds__(256) void k_mega_nested(Params P, Scene S, FrameSeq F, Outputs O) {
  __shared__ int stack[bvh_stack_words<BVH>()];
  uint32_t px, py;
  pixel_of_thread(px, py);
  uint32_t nrays = 0, cnt_nodes = 0, cnt_prims = 0;
  if (px < P.width && py < P.height) {
    const size_t pix = (size_t)py * P.width + px;
    const f3 base = pixel_direction(P, px, py);
    float4 acc = O.accum[pix];
    uchar4 img = O.rgba8[pix];
    int id0 = -1;
    for (uint32_t fi = 0; fi < F.count; ++fi) {
      const uint32_t frame = F.first + fi * F.stride;
      uint32_t rng = pixel_seed(P, px, py, frame);
      f3 total = mk3(0.f, 0.f, 0.f);
      id0 = -1;
      for (int s = 0; s < P.n_spp; ++s) {
        f3 o, d;
        sample_ray<STRICT>(P, base, rng, o, d);
        f3 light = mk3(0.f, 0.f, 0.f), colour = mk3(1.f, 1.f, 1.f);
        int skip = -1;
        for (int b = 0; b < P.n_bounce; ++b) {
          Hit h;
          if constexpr (COUNT) h = closest_hit_bvh<true>(S, o, d, stack, skip, &cnt_nodes, &cnt_prims);
          else h = closest_hit<BVH>(S, o, d, stack, skip);
          ++nrays;
          if (s == 0 && b == 0) id0 = h.idx;
          if (h.idx >= 0) {
            skip = shade_hit<STRICT>(S, h, rng, o, d, light, colour);
          } else {
            light = fma3v(environment<STRICT>(P, d), colour, light);
            break;
          }
        }
        total = total + light;
      }
      frame_epilogue(P, total, (float)frame, O.linear_accum, acc, img);
    }
    if (F.count) {
      O.accum[pix] = acc;
      O.rgba8[pix] = img;
      O.ids[pix] = id0;
    }
  }
  uint32_t wsum = wave_sum_u32(nrays);
  if ((threadIdx.x & 63u) == 0 && wsum) atomicAdd(O.rays, (unsigned long long)wsum);
  if constexpr (COUNT) {
    const uint32_t a = wave_sum_u32(cnt_nodes), b = wave_sum_u32(cnt_prims);
    if ((threadIdx.x & 63u) == 0) {
      atomicAdd(O.rays + 1, (unsigned long long)a);
      atomicAdd(O.rays + 2, (unsigned long long)b);
    }
  }
}

// Requires n_spp >= 1 and n_bounce >= 1 (the host routes other cases to k_mega_nested).
template <bool STRICT, bool BVH>
__global__ __launch_bounds__(256, RT_MEGA_WAVES) void k_mega_regen(Params P, Scene S, FrameSeq F, Outputs O) {
  __shared__ int stack[bvh_stack_words<BVH>()];
  uint32_t px, py;
  pixel_of_thread(px, py);
  uint32_t nrays = 0;
  if (px < P.width && py < P.height && F.count) {
    const size_t pix = (size_t)py * P.width + px;
    const f3 base = pixel_direction(P, px, py);
    float4 acc = O.accum[pix];
    uchar4 img = O.rgba8[pix];
    int id0 = -1;

    uint32_t fi = 0;
    uint32_t frame = F.first;
    uint32_t rng = pixel_seed(P, px, py, frame);
    f3 total = mk3(0.f, 0.f, 0.f);
    int s = 0, b = 0;
    f3 o, d;
    sample_ray<STRICT>(P, base, rng, o, d);
    f3 light = mk3(0.f, 0.f, 0.f), colour = mk3(1.f, 1.f, 1.f);
    int skip = -1;

    bool active = true;
    while (active) {
      Hit h = closest_hit<BVH>(S, o, d, stack, skip);
      ++nrays;
      if (s == 0 && b == 0) id0 = h.idx;
      bool path_done;
      if (h.idx >= 0) {
        skip = shade_hit<STRICT>(S, h, rng, o, d, light, colour);
        ++b;
        path_done = b >= P.n_bounce;
      } else {
        light = fma3v(environment<STRICT>(P, d), colour, light);
        path_done = true;
      }
      if (path_done) {
        total = total + light;
        ++s;
        if (s >= P.n_spp) {
          frame_epilogue(P, total, (float)frame, O.linear_accum, acc, img);
          ++fi;
          if (fi >= F.count) {
            active = false;
          } else {
            frame += F.stride;
            rng = pixel_seed(P, px, py, frame);
            total = mk3(0.f, 0.f, 0.f);
            s = 0;
          }
        }
        if (active) {
          sample_ray<STRICT>(P, base, rng, o, d);
          light = mk3(0.f, 0.f, 0.f);
          colour = mk3(1.f, 1.f, 1.f);
          b = 0;
          skip = -1;
        }
      }
    }
    O.accum[pix] = acc;
    O.rgba8[pix] = img;
    O.ids[pix] = id0;
  }
  uint32_t wsum = wave_sum_u32(nrays);
  if ((threadIdx.x & 63u) == 0 && wsum) atomicAdd(O.rays, (unsigned long long)wsum);
}

template <bool STRICT, bool BVH>
static hipError_t launch_mega_t(bool nested, const Params &P, const Scene &S, const FrameSeq &F,
                                const Outputs &O, hipStream_t stream) {
  dim3 grid((P.width + 15u) / 16u, (P.height + 15u) / 16u, 1);
  dim3 block(256, 1, 1);
  if (nested)
    hipLaunchKernelGGL((k_mega_nested<STRICT, BVH>), grid, block, 0, stream, P, S, F, O);
  else
    hipLaunchKernelGGL((k_mega_regen<STRICT, BVH>), grid, block, 0, stream, P, S, F, O);
  return hipGetLastError();
}

hipError_t launch_megakernel_counting(const Params &P, const Scene &S, const FrameSeq &F,
                                      const Outputs &O, hipStream_t stream) {
  dim3 grid((P.width + 15u) / 16u, (P.height + 15u) / 16u, 1);
  hipLaunchKernelGGL((k_mega_nested<false, true, true>), grid, dim3(256), 0, stream, P, S, F, O);
  return hipGetLastError();
}

hipError_t launch_megakernel(bool nested, bool strict, bool bvh, const Params &P, const Scene &S,
                             const FrameSeq &F, const Outputs &O, hipStream_t stream) {
  if (strict) {
    return bvh ? launch_mega_t<true, true>(nested, P, S, F, O, stream)
               : launch_mega_t<true, false>(nested, P, S, F, O, stream);
  }
  return bvh ? launch_mega_t<false, true>(nested, P, S, F, O, stream)
             : launch_mega_t<false, false>(nested, P, S, F, O, stream);
}

}  // namespace rt
