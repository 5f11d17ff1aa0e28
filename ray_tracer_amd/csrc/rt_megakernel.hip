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

#include <cstdlib>

#ifndef RT_MEGA_WAVES
#define RT_MEGA_WAVES 1
#endif
#ifndef RT_REGEN_SHADE_BATCH
#define RT_REGEN_SHADE_BATCH 48  // hits a wave gathers before it runs the surface code (1: none; 3840x2160: 1 4.36, 12 4.18, 28 4.08, 48 4.03 ms)
#endif
#ifndef RT_REGEN_END_BATCH
#define RT_REGEN_END_BATCH 24  // ended paths a wave gathers before it runs the path-end code (8 and 24 within 1 % once hits are gathered too)
#endif
#ifndef RT_SCHED_WAVES
#define RT_SCHED_WAVES 6  // waves per SIMD the untextured fast k_mega_sched is compiled for (80 VGPR; uncapped it wants 83-87 and runs 5: no scratch instead of 20-36 bytes per lane, 10k spheres 1920x1080 2.11 -> 2.02 ms, but the 3840x2160 batch 110.3 -> 115.4 ms; terrain the same)
#endif
#ifndef RT_XCD_TILES
#define RT_XCD_TILES 1  // k_mega_sched: the four waves of a tile on one XCD (0: consecutive blocks, i.e. four XCDs)
#endif
#ifndef RT_SCHED_EXT_WAVES
#define RT_SCHED_EXT_WAVES 5  // the same for the fast k_mega_sched that carries the extensions (96 VGPR; 10k glass spheres 4K batch: uncapped (97 VGPR, 4 waves) 149.7 ms, 5 waves 135.1, 6 waves 143.1)
#endif
#ifndef RT_REGEN_BATCHED_TEX_WAVES
#define RT_REGEN_BATCHED_TEX_WAVES 6  // the textured batched loop (80 VGPR; 8K frame pair 30.1 ms, uncapped 32.3, 7 waves 31.3)
#endif
#ifndef RT_REGEN_BATCHED_WAVES
#define RT_REGEN_BATCHED_WAVES 7  // waves per SIMD the untextured batched loop is compiled for (72 VGPR; 6: +2 %, 8: +4 % time at 3840x2160)
#endif
#ifndef RT_REGEN_CHAIN_DIV
#define RT_REGEN_CHAIN_DIV 128u  // a mixed launch runs its first n_tiles / 128 blocks through the plain loop (1920x1080: 16: 1.127, 32: 1.113, 64: 1.108, 128: 1.105, 256: 1.106 ms)
#endif
#ifndef RT_REGEN_BATCH_TILES
#define RT_REGEN_BATCH_TILES 12288u  // launches with at least this many tiles (8 x what the device holds) use the batched loop
#endif

namespace rt {

// COUNT (only with BVH): also tallies LBVH nodes fetched and spheres tested into
// O.rays[1] and O.rays[2].
template <bool STRICT, bool BVH, int EXT, bool APPROX, bool COUNT = false>
__global__ __launch_bounds__(256) void k_mega_nested(Params P, Scene S, FrameSeq F, Outputs O) {
  constexpr bool DIEL = EXT == kExtDielectric;
  extern __shared__ int stack[];  // BVH: [entry][thread], S.stack_entries entries (sized by the launch)
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
          if constexpr (COUNT) h = closest_hit_bvh<true, APPROX, DIEL>(S, o, d, stack, skip, &cnt_nodes, &cnt_prims);
          else h = closest_hit<BVH, APPROX, false, DIEL>(S, o, d, stack, skip);
          ++nrays;
          if (s == 0 && b == 0) id0 = h.idx;
          if (h.idx >= 0) {
            skip = shade_hit<STRICT, EXT, APPROX>(S, h, skip, rng, o, d, light, colour);
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
  if ((threadIdx.x & 63u) == 0 && wsum) add_rays(O, wave_index(), wsum);
  if constexpr (COUNT) {
    const uint32_t a = wave_sum_u32(cnt_nodes), b = wave_sum_u32(cnt_prims);
    if ((threadIdx.x & 63u) == 0) {
      atomicAdd(O.rays + 1, (unsigned long long)a);
      atomicAdd(O.rays + 2, (unsigned long long)b);
    }
  }
}

// Tile of this block under the launch-order table, and the issue priority that goes with its
// rank: the first blocks of an ordered launch are the tiles with the longest per-pixel
// chains of samples; they are on the critical path of the launch, so their waves issue
// ahead of the waves they share a SIMD with.
#ifndef RT_PRIO_TOP
#define RT_PRIO_TOP 16   // ranks below n_tiles / RT_PRIO_TOP run at priority 3
#endif
__device__ __forceinline__ uint32_t ordered_tile(const Outputs &O, uint32_t rank) {
  if (!O.tile_order) return rank;
  if (rank * RT_PRIO_TOP < O.n_tiles) __builtin_amdgcn_s_setprio(3);
  else if (rank * (RT_PRIO_TOP / 4) < O.n_tiles) __builtin_amdgcn_s_setprio(2);
  else if (rank * (RT_PRIO_TOP / 16) < O.n_tiles) __builtin_amdgcn_s_setprio(1);
  return O.tile_order[rank];
}

// Threads per block of k_mega_sched and of the brute-force k_mega_regen.  A tile is 16x16
// pixels = four waves.  With kTileBlock = 64 each wave is a block of its own (four consecutive
// blocks per tile), so a wave that finishes frees its slot, registers and stack at once
// instead of holding them until the slowest wave of its tile is done.
#ifndef RT_WAVE_BLOCKS
#define RT_WAVE_BLOCKS 1
#endif
constexpr int kTileBlock = RT_WAVE_BLOCKS ? 64 : 256;
template <int BLOCK>
__device__ __forceinline__ uint32_t tile_rank() { return BLOCK == 64 ? blockIdx.x >> 2 : blockIdx.x; }
template <int BLOCK>
__device__ __forceinline__ uint32_t tile_thread() {
  return BLOCK == 64 ? ((blockIdx.x & 3u) << 6) | threadIdx.x : threadIdx.x;
}

// per-wave epilogue: path segments into the ray counter, slowest pixel into the tile's cost
// (a probe launch records costs and nothing else)
// "This launch is a cost probe."  Probe launches are their own instantiations (PROBE), in which
// the test folds away; the frame kernels keep the run-time flag of the launch in the test, which
// is never set for them: with the test folded away there too, the 1920x1080 frame of the default
// scene came out 3 % slower (1.245 against 1.209 ms; the chain of the slowest wave is that
// sensitive to how the compiler lays out the loop).
#define RT_IS_PROBE (PROBE || O.probe)
// cost: what the wave reports as its tile's cost (the largest report of a tile counts); 0: the
// segments of its slowest pixel
template <bool PROBE>
__device__ __forceinline__ void tally_wave(const Outputs &O, uint32_t tile, uint32_t nrays, uint32_t cost = 0) {
  const uint32_t wsum = wave_sum_u32(nrays);
  const uint32_t wmax = cost ? cost : wave_max_u32(nrays);
  if ((threadIdx.x & 63u) == 0) {
    if (wsum && !RT_IS_PROBE) add_rays(O, wave_index(), wsum);
    if (O.tile_cost) atomicMax(O.tile_cost + tile, wmax);
  }
}

// End of one frame of one pixel in the tile kernels: the accumulation value and the 8-bit
// pixel are read here, blended (frame_epilogue) and written back, instead of being carried in
// five registers through a loop that needs them once per frame.  A probe launch writes nothing.
template <bool PROBE>
__device__ __forceinline__ void end_frame(const Params &P, const Outputs &O, size_t pix, f3 total,
                                          float frame_f) {
  if (RT_IS_PROBE) return;
  float4 acc = O.accum[pix];
  uchar4 img = O.rgba8[pix];
  frame_epilogue(P, total, frame_f, O.linear_accum, acc, img);
  O.accum[pix] = acc;
  O.rgba8[pix] = img;
}

// Path-end code of k_mega_regen's batched loops, for the lanes whose path has ended: light of
// the environment for a path that escaped, sample sum, frame epilogue, next primary ray.
// (A macro over the kernel's locals rather than a lambda: with a lambda the batched kernel,
// which is compiled against a register cap, came out 4 % slower.)
// The batched launch and k_mega_sched (kIdsEarly) store a pixel's primary-hit id where it is
// found (first segment of sample 0 of each frame, the last frame's store stays) instead of
// carrying it to the end of the launch in a register: that register was one of three the
// batched kernel spilled at its 72-VGPR cap.  The plain and mixed launches, whose time is one
// wave's chain, keep the register and store once at the end (2 % faster there).  The timeline
// build keeps its records in that array.
#ifdef RT_MEASURE_TIMELINE
#define RT_STORE_FIRST_ID(idx)
#else
#define RT_STORE_FIRST_ID(idx) \
        if (s == 0 && b == 0) { \
          if (kIdsEarly) { if (!RT_IS_PROBE) O.ids[(size_t)py * P.width + px] = (idx); } \
          else id0 = (idx); \
        }
#endif

#define RT_END_PATHS() \
        if (active && ended != 0) { \
          if (ended == 1) light = fma3v(environment<STRICT>(P, d), colour, light); \
          ended = 0; \
          total = total + light; \
          ++s; \
          if (s >= P.n_spp) { \
            end_frame<PROBE>(P, O, (size_t)py * P.width + px, total, (float)(F.first + fi * F.stride)); \
            ++fi; \
            if (fi >= F.count) { \
              active = false; \
            } else { \
              rng = pixel_seed(P, px, py, F.first + fi * F.stride); \
              total = mk3(0.f, 0.f, 0.f); \
              s = 0; \
            } \
          } \
          if (active) { \
            sample_ray<STRICT>(P, base, rng, o, d); \
            light = mk3(0.f, 0.f, 0.f); \
            colour = mk3(1.f, 1.f, 1.f); \
            b = 0; \
            skip = -1; \
          } \
        }

// Requires n_spp >= 1 and n_bounce >= 1 (the host routes other cases to k_mega_nested).
// FORM: which of the two loops below the blocks run: kRegenPlain, kRegenBatched, or
// kRegenMixed = the plain loop for the first O.chain_ranks blocks of an ordered launch (the
// tiles with the longest chains, which set the launch time) and the batched loop for the rest
// (whose job is throughput).
constexpr int kRegenPlain = 0, kRegenBatched = 1, kRegenMixed = 2;
// Lanes of the wave for which p holds, as a 32-bit scalar.  (The count is taken by the instruction
// itself: from __builtin_popcountll the compiler carried a 64-bit count into the comparisons that
// follow, and 64-bit ordered compares exist only in the vector unit.)
__device__ __forceinline__ int wave_count(bool p) {
  const unsigned long long m = __builtin_amdgcn_ballot_w64(p);
  int n;
  asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc");
  return n;
}
// Spheres fetched four per 64-byte scalar load in the batched loop of launches that are batched
// throughout (the loop bookkeeping and the address arithmetic of the one-sphere-ahead fetch are
// ~12 scalar instructions per sphere, and a wave's scalar instructions cost what its vector ones
// do): 3840x2160 3.80 -> 3.69 ms, 2560x1440 1.722 -> 1.673 ms, 8K textured batch 219.7 -> 212.5 ms.
// Not in the mixed kernel: there the larger loop costs the chain tiles 4 % (1920x1080 1.063 ->
// 1.106 ms).
#ifndef RT_BATCH_GROUPED
#define RT_BATCH_GROUPED 1
#endif
#ifndef RT_BATCH_VOTES
#define RT_BATCH_VOTES 1
#endif
#ifndef RT_SPARSE_RANKS
#define RT_SPARSE_RANKS 2
#endif
#ifndef RT_SPARSE_FEW_TILES
#define RT_SPARSE_FEW_TILES 4096  // 1280x720 has 3,600 tiles, 1920x1080 8,160
#endif
// waves per SIMD a form is compiled for (the register budget follows from it)
#ifndef RT_REGEN_MIXED_WAVES
#define RT_REGEN_MIXED_WAVES RT_MEGA_WAVES
#endif
constexpr int regen_waves(bool strict, bool tex, int form) {
  if (form == kRegenMixed && !strict && !tex) return RT_REGEN_MIXED_WAVES;
  if (form != kRegenBatched || strict) return RT_MEGA_WAVES;
  return tex ? RT_REGEN_BATCHED_TEX_WAVES : RT_REGEN_BATCHED_WAVES;
}
// PROBE: a cost probe (its own instantiation, so that profiles list it apart from the frames):
// traces like any launch but writes the per-tile costs and nothing else.
template <bool STRICT, bool BVH, int EXT, bool APPROX, int FORM, bool PROBE = false>
__global__ __launch_bounds__(BVH ? 256 : kTileBlock, regen_waves(STRICT, EXT != kExtNone, FORM)) void k_mega_regen(Params P, Scene S, FrameSeq F, Outputs O) {
  constexpr bool DIEL = EXT == kExtDielectric;
  extern __shared__ int stack[];  // BVH: [entry][thread], S.stack_entries entries (sized by the launch)
  // The plain loop is used where one wave's chain of segments sets the launch time: its
  // brute-force search is the form tuned for a wave running alone, and the hit carries the
  // sphere's centre to the shading code instead of having it fetched again.
  constexpr bool CHAIN = !BVH;
  // batch sizes of the batched loop, in the mixed kernel too.  (While the chain tiles ran as full
  // waves the mixed kernel needed small batches, 8 path ends and no gathering of hits, because the
  // tiles just below the chain ranks still have long chains that large batches stretch; since the
  // heaviest tiles run as sparse waves, large batches for everything else win: 1920x1080 1.156 ->
  // 1.117 ms, and 1.105 with a chain share of 1/128 instead of 1/32.)
  constexpr bool kGroupedFetch = RT_BATCH_GROUPED != 0 && FORM == kRegenBatched && !BVH;
  constexpr bool kSmall = FORM != kRegenBatched;  // straight-line search for scenes of <= 4 spheres (see closest_hit_brute)
  constexpr int SHADE_BATCH = FORM != kRegenPlain ? RT_REGEN_SHADE_BATCH : 1;
  constexpr int END_BATCH = FORM != kRegenPlain ? RT_REGEN_END_BATCH : 8;
  constexpr int BLOCK = BVH ? 256 : kTileBlock;
  // Mixed form: the tiles of the first O.sparse_ranks ranks, whose chains bound the launch, are
  // spread over 2^O.sparse_shift times the waves, each with that fraction of its lanes switched
  // on (whole rows of the wave's 8x8 pixels).  What stretches the chain of the slowest lane is
  // the path-end code that runs, in nearly every iteration, for whichever OTHER lanes of its wave
  // need it; with a quarter of the lanes it runs a quarter as often.  The extra waves cost
  // issue slots for the whole launch, so only a few tiles pay: 1920x1080 (small batches, at the
  // time) 1.207 -> 1.160 ms with 3 tiles of 4 x 16 lanes; 16 tiles 1.187, 64 tiles 1.264; with
  // large batches 2 tiles of 8 x 8 lanes are the optimum.  1280x720, where the device is never
  // full: 1.25 -> 1.09 ms with 8 x 8 lanes, 3 to 64 tiles alike.
  uint32_t rank = tile_rank<BLOCK>(), tile_t = tile_thread<BLOCK>();
  bool lane_on = true;
  if constexpr (FORM == kRegenMixed) {
    const uint32_t k = O.sparse_shift;  // 1, 2 or 3: each wave becomes 2, 4 or 8
    const uint32_t nb = (O.sparse_ranks * 4u) << k;
    if (blockIdx.x < nb) {
      rank = blockIdx.x >> (2u + k);
      const uint32_t sub = blockIdx.x & ((4u << k) - 1u);
      tile_t = ((sub >> k) << 6) | threadIdx.x;
      lane_on = (threadIdx.x >> (6u - k)) == (sub & ((1u << k) - 1u));
    } else {
      const uint32_t bb = blockIdx.x - O.sparse_ranks * 4u * ((1u << k) - 1u);
      rank = bb >> 2;
      tile_t = ((bb & 3u) << 6) | threadIdx.x;
    }
  }
  const bool batched = FORM == kRegenMixed ? rank >= O.chain_ranks : FORM == kRegenBatched;
  const uint32_t tile = ordered_tile(O, rank);
#ifdef RT_DEBUG_SKIP_RANKS
  // measurement build: the heaviest tiles of an ordered launch are not traced at all (they keep their
  // place in the order), which shows what the launch would take if their chains cost nothing
  if (O.tile_order && rank < RT_DEBUG_SKIP_RANKS) {
    if (threadIdx.x == 0 && O.tile_cost) atomicMax(O.tile_cost + tile, 0x40000000u - rank);
    return;
  }
#endif
  uint32_t px, py;
  pixel_of_tile(tile, O.tiles_x, tile_t, px, py);
  uint32_t nrays = 0;
#ifdef RT_MEASURE_TIMELINE
  const unsigned long long t_begin = wall_clock64();
#endif
  if (px < P.width && py < P.height && F.count && lane_on) {
    const f3 base = pixel_direction(P, px, py);
    constexpr bool kIdsEarly = FORM == kRegenBatched;
    int id0 = -1;

    uint32_t fi = 0;
    uint32_t rng = pixel_seed(P, px, py, F.first);
    f3 total = mk3(0.f, 0.f, 0.f);
    int s = 0, b = 0;
    f3 o, d;
    sample_ray<STRICT>(P, base, rng, o, d);
    f3 light = mk3(0.f, 0.f, 0.f), colour = mk3(1.f, 1.f, 1.f);
    int skip = -1;

    if (batched) {
      // Throughput form, for launches with many more tiles than the device holds at once.
      // A path that ended (missed everything, or reached the bounce limit) waits until
      // END_BATCH lanes of the wave have one to end; the path-end code (environment, sample
      // sum, next primary ray) then runs for all of them at once instead of in every
      // iteration for the two or three lanes that happen to need it.  It does not shorten the
      // wave that bounds a small launch (its lanes now also wait for the batch), which is why
      // the plain loop below stays for those.
      bool active = true;
      int ended = 0;  // 0: tracing, 1: missed (environment still to add), 2: bounce limit
      if constexpr (SHADE_BATCH > 1) {
        // Hits wait too: the surface code runs once SHADE_BATCH lanes have one pending,
        // instead of after every search for the third of the lanes that hit.
#if RT_BATCH_VOTES
        // Every vote of this loop is ONE vector compare of a per-lane word: "has a hit pending"
        // is pending.idx >= 0 and "has a path to end" is ended > 0, a lane that is finished
        // holding ended = -1 (with bool flags and-ed together the compiler kept them as masks in
        // scalar registers, updated by three scalar instructions at every divergent assignment,
        // and materialised each vote as a select plus a second compare).
        Hit pending;
        pending.idx = -1;
        pending.dst = 0.0f;
        pending.cx = pending.cy = pending.cz = 0.0f;
        while (true) {
          if (ended == 0 && pending.idx < 0) {
            const Hit h = closest_hit<BVH, APPROX, kGroupedFetch, DIEL, kSmall>(S, o, d, stack, skip);
            ++nrays;
            RT_STORE_FIRST_ID(h.idx);
            if (h.idx >= 0) pending = h;
            else ended = 1;
          }
          const int n_hit = wave_count(pending.idx >= 0);
          const int n_ended = wave_count(ended > 0);
          if (n_hit + n_ended == 0) break;
          // a step runs when its batch is together; when neither is, the one more lanes wait for
          // (the two conditions are written out as scalar tests: as bool variables they went
          // through vector registers)
          if (n_hit >= SHADE_BATCH || (n_ended < END_BATCH && n_hit >= n_ended)) {
            if (pending.idx >= 0) {
              skip = shade_hit<STRICT, EXT, APPROX>(S, pending, skip, rng, o, d, light, colour);
              pending.idx = -1;
              ++b;
              if (b >= P.n_bounce) ended = 2;
            }
          }
          if (!(n_ended >= END_BATCH || (n_hit < SHADE_BATCH && n_hit < n_ended))) continue;
          if (ended > 0) {
            if (ended == 1) light = fma3v(environment<STRICT>(P, d), colour, light);
            ended = 0;
            total = total + light;
            ++s;
            if (s >= P.n_spp) {
              end_frame<PROBE>(P, O, (size_t)py * P.width + px, total, (float)(F.first + fi * F.stride));
              ++fi;
              if (fi >= F.count) {
                ended = -1;
              } else {
                rng = pixel_seed(P, px, py, F.first + fi * F.stride);
                total = mk3(0.f, 0.f, 0.f);
                s = 0;
              }
            }
            if (ended == 0) {
              sample_ray<STRICT>(P, base, rng, o, d);
              light = mk3(0.f, 0.f, 0.f);
              colour = mk3(1.f, 1.f, 1.f);
              b = 0;
              skip = -1;
            }
          }
        }
#else
        Hit pending;
        pending.idx = -1;
        pending.dst = 0.0f;
        pending.cx = pending.cy = pending.cz = 0.0f;
        bool has_hit = false;
        while (true) {
          if (active && ended == 0 && !has_hit) {
            const Hit h = closest_hit<BVH, APPROX, false, DIEL>(S, o, d, stack, skip);
            ++nrays;
            RT_STORE_FIRST_ID(h.idx);
            if (h.idx >= 0) {
              pending = h;
              has_hit = true;
            } else {
              ended = 1;
            }
          }
          // every lane that could trace has; now at least one of the other two steps must
          // run: those that have their batch together, else the one more lanes wait for
          const int n_hit = __builtin_popcountll(__ballot(has_hit));
          const int n_ended = __builtin_popcountll(__ballot(active && ended != 0));
          if (n_hit + n_ended == 0) break;
          bool run_shade = n_hit >= SHADE_BATCH, run_end = n_ended >= END_BATCH;
          if (!run_shade && !run_end) {
            run_shade = n_hit >= n_ended;
            run_end = !run_shade;
          }
          if (run_shade && has_hit) {
            skip = shade_hit<STRICT, EXT, APPROX>(S, pending, skip, rng, o, d, light, colour);
            has_hit = false;
            ++b;
            if (b >= P.n_bounce) ended = 2;
          }
          if (!run_end) continue;
          RT_END_PATHS();
        }
#endif
      } else {
        while (true) {
          if (active && ended == 0) {
            const Hit h = closest_hit<BVH, APPROX, false, DIEL>(S, o, d, stack, skip);
            ++nrays;
            RT_STORE_FIRST_ID(h.idx);
            if (h.idx >= 0) {
              skip = shade_hit<STRICT, EXT, APPROX>(S, h, skip, rng, o, d, light, colour);
              ++b;
              if (b >= P.n_bounce) ended = 2;
            } else {
              ended = 1;
            }
          }
          const int n_ended = __builtin_popcountll(__ballot(active && ended != 0));
          const int n_tracing = __builtin_popcountll(__ballot(active && ended == 0));
          if (n_ended + n_tracing == 0) break;
          if (n_ended >= END_BATCH || n_tracing == 0) RT_END_PATHS();
        }
      }
    } else {
      // (this loop keeps the pixel's accumulation value and 8-bit colour in registers: with
      // the read-modify-write of end_frame inside it the chain of a lone wave is 12 % longer)
      const size_t pix = (size_t)py * P.width + px;
      uint32_t frame = F.first;
      float4 acc = O.accum[pix];
      uchar4 img = O.rgba8[pix];
      bool active = true;
      while (active) {
        Hit h = closest_hit<BVH, APPROX, CHAIN, DIEL, kSmall>(S, o, d, stack, skip);
        ++nrays;
        if (s == 0 && b == 0) id0 = h.idx;
        bool path_done;
        if (h.idx >= 0) {
          skip = shade_hit<STRICT, EXT, APPROX, CHAIN>(S, h, skip, rng, o, d, light, colour);
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
      if (!RT_IS_PROBE) {
        O.accum[pix] = acc;
        O.rgba8[pix] = img;
      }
    }
#ifndef RT_MEASURE_TIMELINE
    if (!kIdsEarly && !RT_IS_PROBE) O.ids[(size_t)py * P.width + px] = id0;
#endif
  }
  tally_wave<PROBE>(O, tile, nrays);
#ifdef RT_MEASURE_TIMELINE
  // measurement build (tools/wave_timeline.py): per-wave begin and end in 100 MHz ticks,
  // segments traced and the slowest lane's segments go where the primary ids would
  const uint32_t wsum = wave_sum_u32(nrays), wmax = wave_max_u32(nrays);
  if ((threadIdx.x & 63u) == 0) {
    const unsigned long long t_end = wall_clock64();
    int *rec = O.ids + 6u * (size_t)(tile * 4u + (tile_t >> 6));
    rec[0] = (int)(uint32_t)t_begin;
    rec[1] = (int)(uint32_t)(t_begin >> 32);
    rec[2] = (int)(uint32_t)t_end;
    rec[3] = (int)(uint32_t)(t_end >> 32);
    rec[4] = (int)wsum;
    rec[5] = (int)wmax;
  }
#endif
}

// BVH scenes: the lanes of a wave need very different numbers of tree steps per path
// segment, and under the one-loop-per-search structure above every lane waits for the
// slowest search of its wave before anyone shades.  k_mega_sched keeps each lane's search
// state (current node, stack depth, best hit) alive across loop iterations and lets the
// WAVE choose, each iteration, which of three steps to run for the lanes that need it:
//   inner  fetch one node, test both child boxes, descend / push / pop
//   leaf   test one primitive, pop
//   shade  finish the segment (surface + scatter, or environment and the next sample)
//          and start the next search
// The step with the most waiting lanes runs (weights below), so inner steps run with most
// lanes busy and the expensive shade step runs for many lanes at once.  Each lane still
// executes its own operations in the shader's order, and the closest hit does not depend on
// visiting order (ties go to the lowest primitive id), so the results are those of
// k_mega_nested bit for bit.
// measurement builds only: -DRT_SCHED_STATS=<step> tallies, for that step, the lanes served
// and the times it ran (x64) in the traversal counters
#ifdef RT_SCHED_STATS
#define RT_SCHED_TALLY(step, n)                                               \
  if ((step) == RT_SCHED_STATS && (threadIdx.x & 63u) == 0) {                 \
    atomicAdd(O.rays + 1, (unsigned long long)(n));                           \
    atomicAdd(O.rays + 2, 64ull);                                             \
  }
#else
#define RT_SCHED_TALLY(step, n)
#endif

template <bool STRICT, int EXT, bool APPROX, bool PROBE = false>
__global__ __launch_bounds__(kTileBlock, STRICT ? 1 : (EXT ? RT_SCHED_EXT_WAVES : RT_SCHED_WAVES)) void k_mega_sched(Params P, Scene S, FrameSeq F, Outputs O) {
  // [entry][thread], S.stack_entries entries: the launch sizes it by the height of the tree
  // (a 16-level tree leaves room for twice the waves a fixed 32-entry stack would)
  constexpr bool DIEL = EXT == kExtDielectric;
  extern __shared__ int stack[];
#if RT_XCD_TILES
  // Workgroups go to the 8 XCDs round-robin by index, and each XCD has its own L2: the four
  // waves of a tile are blocks b, b+8, b+16, b+24, so ONE XCD fetches the part of the tree the
  // tile's rays walk instead of four (placement is a speed matter only; terrain frame and 10k
  // glass spheres 4K batch 1.5 % faster).  The grid is padded to whole groups of 8 tiles.
  const uint32_t xrank = (blockIdx.x >> 5) * 8u + (blockIdx.x & 7u);
  if (xrank >= O.n_tiles) return;
  const uint32_t tile = ordered_tile(O, xrank);
  uint32_t px, py;
  pixel_of_tile(tile, O.tiles_x, (((blockIdx.x >> 3) & 3u) << 6) | threadIdx.x, px, py);
#else
  const uint32_t tile = ordered_tile(O, tile_rank<kTileBlock>());
  uint32_t px, py;
  pixel_of_tile(tile, O.tiles_x, tile_thread<kTileBlock>(), px, py);
#endif
  uint32_t nrays = 0;
#ifdef RT_MEASURE_TIMELINE
  const unsigned long long t_begin_sched = wall_clock64();
#endif
  const bool active = px < P.width && py < P.height && F.count;
  const f3 base = pixel_direction(P, px, py);
  constexpr bool kIdsEarly = true;
  int id0 = -1;  // (unused here: RT_STORE_FIRST_ID stores at once)
  (void)id0;
  uint32_t fi = 0;
  uint32_t rng = pixel_seed(P, px, py, F.first);
  f3 total = mk3(0.f, 0.f, 0.f);
  int s = 0, b = 0;
  f3 o, d;
  sample_ray<STRICT>(P, base, rng, o, d);
  f3 light = mk3(0.f, 0.f, 0.f), colour = mk3(1.f, 1.f, 1.f);
  int skip = -1;
  Search<APPROX> Q;
  search_begin<DIEL>(S, o, d, skip, Q);
  if (!active) Q.node = kSearchIdle;

  // The tile's cost for the next launch's order is the number of steps the wave took, not the
  // segments of its slowest pixel: with a tree the time of a wave follows the nodes its rays
  // visit (on the terrain the waves at the horizon take 2 ms for 20 segments per pixel, the
  // median wave 0.07 ms), and ordered by segments those waves started 2.5 ms into a 4.4 ms
  // launch and ran on alone for its last quarter.
  uint32_t steps = 0;
  while (true) {
    steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)steps) + 1u;  // (wave-uniform: kept in a scalar register)
    // A lane without work holds kSearchIdle as its node, so each of the three votes is ONE
    // vector compare of the node word: with a separate "active" flag and-ed into them the
    // compiler materialised every vote as a select plus a second compare (12 instructions of
    // ~30 at the head of every iteration; 4K glass batch 132.4 -> 129.6 ms, terrain frame
    // 3.73 -> 3.68 ms).
    const bool want_inner = Q.node >= 0;
    const bool want_shade = Q.node == kSearchOver;
    const bool want_leaf = (uint32_t)Q.node >= kLeafRefLowest;
    const int n_inner = __builtin_popcountll(__ballot(want_inner));
    const int n_leaf = __builtin_popcountll(__ballot(want_leaf));
    const int n_shade = __builtin_popcountll(__ballot(want_shade));
    if (n_inner + n_leaf + n_shade == 0) break;
    const int w_inner = n_inner * RT_SCHED_W_INNER, w_leaf = n_leaf * RT_SCHED_W_LEAF,
              w_shade = n_shade * RT_SCHED_W_SHADE;
    if (w_inner >= w_leaf && w_inner >= w_shade) {
      RT_SCHED_TALLY(1, n_inner);
      // A lane that is still descending takes up to RT_SCHED_INNER_REPS inner steps before the
      // wave goes back to the scheduler: the three ballots and the weighting cost about a
      // third of an inner step.  (Re-counting the lanes after every step to stop early
      // costs more than the empty lanes it saves: 5.4 ms against 5.1 ms on the terrain.)
      if (want_inner) {
        search_inner_step<kTileBlock>(S, Q, stack);
#pragma unroll
        for (int rep = 1; rep < RT_SCHED_INNER_REPS; ++rep)
          if (Q.node >= 0) search_inner_step<kTileBlock>(S, Q, stack);
      }
    } else if (w_leaf >= w_shade) {
      RT_SCHED_TALLY(2, n_leaf);
      if (want_leaf) {
        leaf_range<DIEL>(S, Q.node, o, d, Q.a, skip, Q.best);
        Q.node = search_pop<kTileBlock>(Q, stack);
      }
    } else {
      RT_SCHED_TALLY(3, n_shade);
      if (!want_shade) continue;
      const Hit h = Q.best.result();
      ++nrays;
      RT_STORE_FIRST_ID(h.idx);
      bool path_done;
      if (h.idx >= 0) {
        skip = shade_hit<STRICT, EXT, APPROX>(S, h, skip, rng, o, d, light, colour);
        ++b;
        path_done = b >= P.n_bounce;
      } else {
        light = fma3v(environment<STRICT>(P, d), colour, light);
        path_done = true;
      }
      // (whether the pixel goes on is a flag of this step only; between steps a finished lane is
      // known by its node word, kSearchIdle: a flag carried around the loop is a scalar mask that
      // costs three scalar instructions in every iteration)
      bool more = true;
      if (path_done) {
        total = total + light;
        ++s;
        if (s >= P.n_spp) {
          end_frame<PROBE>(P, O, (size_t)py * P.width + px, total, (float)(F.first + fi * F.stride));
          ++fi;
          if (fi >= F.count) {
            more = false;
          } else {
            rng = pixel_seed(P, px, py, F.first + fi * F.stride);
            total = mk3(0.f, 0.f, 0.f);
            s = 0;
          }
        }
        if (more) {
          sample_ray<STRICT>(P, base, rng, o, d);
          light = mk3(0.f, 0.f, 0.f);
          colour = mk3(1.f, 1.f, 1.f);
          b = 0;
          skip = -1;
        }
      }
      if (more) search_begin<DIEL>(S, o, d, skip, Q);
      else Q.node = kSearchIdle;
    }
  }
  tally_wave<PROBE>(O, tile, nrays, steps);
#ifdef RT_MEASURE_TIMELINE
  {  // as in k_mega_regen: the wave's begin / end / segments / slowest lane where the ids would go
    const uint32_t wsum = wave_sum_u32(nrays), wmax = wave_max_u32(nrays);
    if ((threadIdx.x & 63u) == 0) {
      const unsigned long long t_end = wall_clock64();
      int *rec = O.ids + 6u * (size_t)(tile * 4u + ((blockIdx.x >> 3) & 3u));
      rec[0] = (int)(uint32_t)t_begin_sched;
      rec[1] = (int)(uint32_t)(t_begin_sched >> 32);
      rec[2] = (int)(uint32_t)t_end;
      rec[3] = (int)(uint32_t)(t_end >> 32);
      rec[4] = (int)wsum;
      rec[5] = (int)wmax;
    }
  }
#endif
}

template <bool STRICT, bool BVH, int EXT, bool APPROX>
static hipError_t launch_mega_t(bool nested, const Params &P, const Scene &S, const FrameSeq &F,
                                const Outputs &O, hipStream_t stream) {
  dim3 grid((P.width + 15u) / 16u, (P.height + 15u) / 16u, 1);
  dim3 tiles(O.n_tiles, 1, 1);  // k_mega_regen and k_mega_sched take tiles from a 1-D grid
#ifdef RT_DEBUG_TILE_LIMIT
  // measurement build (tools/lone_tile.py): only the first N tiles of the order are launched
  if (const char *e = std::getenv("RT_DEBUG_TILE_LIMIT")) {
    const unsigned long v = std::strtoul(e, nullptr, 10);
    if (v && v < O.n_tiles) tiles.x = (uint32_t)v;
  }
#endif
  dim3 block(256, 1, 1);
  // kernels whose blocks are single waves: four blocks per tile
  const dim3 wave_tiles(tiles.x * (256u / kTileBlock), 1, 1), wave_block(kTileBlock, 1, 1);
#if RT_XCD_TILES
  const dim3 sched_tiles((tiles.x + 7u) / 8u * 32u, 1, 1);
#else
  const dim3 sched_tiles = wave_tiles;
#endif
  // k_mega_regen and k_mega_sched need at least one sample and one bounce
  static const bool sched = [] {
    const char *e = getenv("RT_MEGA_SCHED");
    return !(e && e[0] == '0');
  }();
  static const uint32_t batch_tiles = [] {
    const char *e = getenv("RT_REGEN_BATCH_TILES");
    return e ? (uint32_t)strtoul(e, nullptr, 10) : RT_REGEN_BATCH_TILES;
  }();
  static const uint32_t chain_div = [] {
    const char *e = getenv("RT_REGEN_CHAIN_DIV");
    const uint32_t v = e ? (uint32_t)strtoul(e, nullptr, 10) : RT_REGEN_CHAIN_DIV;
    return v ? v : 1u;
  }();
  static const int forced_form = [] {  // RT_REGEN_FORM=plain|batched|mixed pins the loop (tests, measurements)
    const char *e = getenv("RT_REGEN_FORM");
    if (!e) return -1;
    return e[0] == 'p' ? kRegenPlain : e[0] == 'b' ? kRegenBatched : e[0] == 'm' ? kRegenMixed : -1;
  }();
  // LDS of the kernels that run one whole tree search per segment: a stack per thread, as deep as the tree
  const size_t block_stack = BVH ? (size_t)S.stack_entries * 256u * sizeof(int) : 0;
  if (nested)
    hipLaunchKernelGGL((k_mega_nested<STRICT, BVH, EXT, APPROX>), grid, block, block_stack, stream, P, S, F, O);
  else if (BVH && sched && O.probe)
    hipLaunchKernelGGL((k_mega_sched<STRICT, EXT, APPROX, true>), sched_tiles, wave_block,
                       (size_t)S.stack_entries * kTileBlock * sizeof(int), stream, P, S, F, O);
  else if (BVH && sched)
    hipLaunchKernelGGL((k_mega_sched<STRICT, EXT, APPROX>), sched_tiles, wave_block,
                       (size_t)S.stack_entries * kTileBlock * sizeof(int), stream, P, S, F, O);
  else if (BVH && O.probe)
    hipLaunchKernelGGL((k_mega_regen<STRICT, BVH, EXT, APPROX, kRegenPlain, true>), tiles, block, block_stack, stream, P, S, F, O);
  else if (BVH)
    hipLaunchKernelGGL((k_mega_regen<STRICT, BVH, EXT, APPROX, kRegenPlain>), tiles, block, block_stack, stream, P, S, F, O);
  else {
    // Brute-force scenes.  Many tiles: throughput is everything, the batched loop.  Fewer: the
    // longest chain sets the time; with a measured order its tiles (and only they) get the
    // plain loop, without one every tile does.  1920x1080: plain 1.36, batched 1.61, mixed
    // 1.27 ms; 3840x2160: 5.00, 4.02, 4.35 (the mixed kernel holds both loops' registers).
    int form = O.n_tiles >= batch_tiles ? kRegenBatched : (O.tile_order ? kRegenMixed : kRegenPlain);
    if (forced_form >= 0 && (forced_form != kRegenMixed || O.tile_order)) form = forced_form;
    Outputs Q = O;
    Q.chain_ranks = (O.n_tiles + chain_div - 1u) / chain_div;
    constexpr bool B = !BVH;  // keeps the BVH instantiations of the brute-force forms out of the library
    if (O.probe && form == kRegenBatched)
      hipLaunchKernelGGL((k_mega_regen<STRICT, BVH, EXT, APPROX, B ? kRegenBatched : kRegenPlain, true>), wave_tiles, wave_block, 0, stream, P, S, F, Q);
    else if (O.probe)
      hipLaunchKernelGGL((k_mega_regen<STRICT, BVH, EXT, APPROX, kRegenPlain, true>), wave_tiles, wave_block, 0, stream, P, S, F, Q);
    else if (form == kRegenMixed) {
      // The tiles of the first chain_ranks ranks run the plain loop, the first sparse_ranks of
      // them in waves of 64 >> shift lanes (see the kernel); the rest run the batched loop.
      // (Two kernels side by side on two streams, one per loop, were measured too: 1.13 ms at
      // 1920x1080 against 1.105 for this one kernel, whose fork and join cost nothing.)
      // RT_SPARSE_RANKS / RT_SPARSE_SHIFT override (measurements).
      static const int sparse_env = [] {
        const char *e = getenv("RT_SPARSE_RANKS");
        return e ? (int)strtoul(e, nullptr, 10) : -1;
      }();
      static const int shift_env = [] {
        const char *e = getenv("RT_SPARSE_SHIFT");
        const int v = e ? (int)strtoul(e, nullptr, 10) : -1;
        return v < 0 ? -1 : v < 1 ? 1 : v > 3 ? 3 : v;
      }();
      // launches that never fill the device: 16 tiles; others: 2 (RT_SPARSE_RANKS); 8 waves of 8 lanes each
      const bool few = O.n_tiles <= RT_SPARSE_FEW_TILES;
      const uint32_t sparse = sparse_env >= 0 ? (uint32_t)sparse_env : few ? 16u : (uint32_t)RT_SPARSE_RANKS;
      const uint32_t shift = shift_env >= 0 ? (uint32_t)shift_env : 3u;
      Q.sparse_ranks = sparse < Q.chain_ranks ? sparse : Q.chain_ranks;
      if (Q.sparse_ranks > tiles.x) Q.sparse_ranks = tiles.x;
      Q.sparse_shift = shift;
      const uint32_t extra = Q.sparse_ranks * 4u * ((1u << shift) - 1u);
      const dim3 mixed_tiles(wave_tiles.x + extra, 1, 1);
      hipLaunchKernelGGL((k_mega_regen<STRICT, BVH, EXT, APPROX, B ? kRegenMixed : kRegenPlain>), mixed_tiles, wave_block, 0, stream, P, S, F, Q);
    }
    else if (form == kRegenBatched)
      hipLaunchKernelGGL((k_mega_regen<STRICT, BVH, EXT, APPROX, B ? kRegenBatched : kRegenPlain>), wave_tiles, wave_block, 0, stream, P, S, F, Q);
    else
      hipLaunchKernelGGL((k_mega_regen<STRICT, BVH, EXT, APPROX, kRegenPlain>), wave_tiles, wave_block, 0, stream, P, S, F, Q);
  }
  return hipGetLastError();
}

hipError_t launch_megakernel_counting(const Params &P, const Scene &S, const FrameSeq &F,
                                      const Outputs &O, hipStream_t stream) {
  dim3 grid((P.width + 15u) / 16u, (P.height + 15u) / 16u, 1);
  const size_t block_stack = (size_t)S.stack_entries * 256u * sizeof(int);
  if (S.ior)
    hipLaunchKernelGGL((k_mega_nested<false, true, kExtDielectric, false, true>), grid, dim3(256), block_stack, stream, P, S, F, O);
  else if (S.tex)
    hipLaunchKernelGGL((k_mega_nested<false, true, kExtTexture, false, true>), grid, dim3(256), block_stack, stream, P, S, F, O);
  else
    hipLaunchKernelGGL((k_mega_nested<false, true, kExtNone, false, true>), grid, dim3(256), block_stack, stream, P, S, F, O);
  return hipGetLastError();
}

template <bool STRICT, bool BVH, bool APPROX>
static hipError_t launch_mega_e(bool nested, const Params &P, const Scene &S, const FrameSeq &F, const Outputs &O,
                                hipStream_t stream) {
  if (S.ior) return launch_mega_t<STRICT, BVH, kExtDielectric, APPROX>(nested, P, S, F, O, stream);
  if (S.tex) return launch_mega_t<STRICT, BVH, kExtTexture, APPROX>(nested, P, S, F, O, stream);
  return launch_mega_t<STRICT, BVH, kExtNone, APPROX>(nested, P, S, F, O, stream);
}

hipError_t launch_megakernel(bool nested, bool strict, bool bvh, const Params &P, const Scene &S,
                             const FrameSeq &F, const Outputs &O, hipStream_t stream) {
  // fast policy without a mesh: approximate sphere distances behind exact decisions
  const bool approx = !strict && S.n_tris == 0u;
  if (strict)
    return bvh ? launch_mega_e<true, true, false>(nested, P, S, F, O, stream)
               : launch_mega_e<true, false, false>(nested, P, S, F, O, stream);
  if (approx)
    return bvh ? launch_mega_e<false, true, true>(nested, P, S, F, O, stream)
               : launch_mega_e<false, false, true>(nested, P, S, F, O, stream);
  return bvh ? launch_mega_e<false, true, false>(nested, P, S, F, O, stream)
             : launch_mega_e<false, false, false>(nested, P, S, F, O, stream);
}

}  // namespace rt
