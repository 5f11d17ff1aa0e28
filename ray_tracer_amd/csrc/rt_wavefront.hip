// Wavefront path tracer: the per-pixel program of rt_shading.h split into stages that
// run over compacted queues, so that every wave of a stage executes one branch of the
// shader (all "hit" or all "miss") instead of both.
//
// One path per pixel is in flight (a pixel's RNG state advances sequentially through
// its samples, ray_tracing.comp:82-96, so the samples of one pixel cannot run side by
// side).  Path state lives in HBM as structure-of-arrays of float4, indexed by pixel:
//   P0 = origin.xyz, rng            P1 = dir.xyz, (sample << 16 | bounce)
//   P2 = throughput.xyz, skip id    P3 = light.xyz, -
//   T  = sum over samples.xyz, -    H  = (dst, sphere index) of the last extend
//   alive = 1 while the pixel still has samples to trace in this frame
// One iteration = two launches:
//   extend  walks the PIXELS in order (2048 consecutive pixels per block), traces the
//           live ones, stores H, and sorts the block's live pixels into ITS OWN slice of
//           the queue: hits packed from the front, misses from the back (ballot ranks and
//           an LDS scan; no global atomic decides a position)
//   shade   block b reads slice b: its leading waves scatter the hits, its trailing
//           waves light the misses and start the pixel's next sample; at most two waves
//           of a block see both kinds
// The totals the host polls rotate through four slots so no launch ever resets a counter
// another launch of the same iteration still reads.
//
// Three measured facts shape the kernels.  (1) Returning atomics on ONE counter retire
// at ~90 per microsecond chip-wide, so a push per wave (32k waves per launch at 1080p)
// costs more than the stage's arithmetic; even one push per block and queue (4,050 per
// launch) held the extend stage at 54 us.  Hence the per-block queue slices above: the only
// atomics left are two fire-and-forget additions per block for the totals.  (2) A queue that is itself
// compacted from queue order drifts into a random permutation of the pixels after a few
// dozen iterations, and 16-byte gathers from random lines cost 3x the time of the
// first, ordered iteration; walking pixels in order keeps every queue chunk inside a
// 1024-pixel window of the state arrays.  (3) A few pixels need ten times more path
// segments than the median, so the tail of the frame is hundreds of launches that do
// almost nothing: once the live count drops below a threshold the remaining pixels are
// compacted once and finished by one "drain" launch that keeps each path in registers
// until its pixel is done (the megakernel's loop, resumed from the stored state).
#include "rt_wavefront.h"

#include "rt_lbvh.h"
#include "rt_shading.h"
#include "rt_traverse.h"

#ifndef RT_WF_TREE_WAVES
#define RT_WF_TREE_WAVES 1  // waves per SIMD the scheduled tree kernels are compiled for
#endif

namespace rt {

struct WfView {
  float4 *P0, *P1, *P2, *P3, *T;
  float2 *H;
  unsigned char *alive;
  uint32_t *q;       // one slice of kWfChunk entries per extend block; also the drain queue
  uint2 *slice;      // per extend block: hits at the front, misses at the back of its slice
  uint32_t *counts;  // [4 slots][4]: -, hits, misses, - (totals); [16]: length of the drain queue
  const uint32_t *order;  // regrouping experiment: the extend stage's k-th entry is pixel order[k]; null: pixel k
};

struct RegroupGrid {
  float lo[3], scale[3];  // cell = clamp((origin - lo) * scale, 0, 15) per axis
};
// Sort key of every pixel for the regrouping experiment: 12 bits of Morton-interleaved cell of the ray's
// origin, 3 bits of direction octant; pixels without a live path sort to the end.
__global__ __launch_bounds__(256) void k_wf_keys(WfView V, uint32_t n_pixels, RegroupGrid G, uint32_t *keys,
                                                 uint32_t *vals) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_pixels) return;
  uint32_t key = 0xffffu;
  if (V.alive[i]) {
    const float4 p0 = V.P0[i], p1 = V.P1[i];
    const float o[3] = {p0.x, p0.y, p0.z};
    uint32_t cell[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float t = (o[k] - G.lo[k]) * G.scale[k];
      t = t < 0.f ? 0.f : (t > 15.f ? 15.f : t);
      cell[k] = (uint32_t)t;
    }
    uint32_t m = 0;
#pragma unroll
    for (int b = 3; b >= 0; --b)
      m = (m << 3) | (((cell[0] >> b) & 1u) << 2) | (((cell[1] >> b) & 1u) << 1) | ((cell[2] >> b) & 1u);
    const uint32_t oct = (p1.x < 0.f ? 4u : 0u) | (p1.y < 0.f ? 2u : 0u) | (p1.z < 0.f ? 1u : 0u);
    key = (m << 3) | oct;
  }
  keys[i] = key;
  vals[i] = i;
}

// Queue entries per thread of the extend and shade stages (a block handles 256 x E pixels,
// and that is the size of its queue slice).  Brute-force scenes want few, long blocks (8:
// 8.7 ms per frame against 9.7 with 4 and 11.5 with 2 on the 3-sphere scene); a tree search
// is long and ragged, so LBVH scenes want many waves instead (2: 4.6 ms against 4.9 with 4
// and 6.0 with 8 on the 10k-sphere scene).
constexpr int kWfEBrute = 8, kWfETree = 2;
constexpr int kWfE = kWfEBrute;             // k_wf_compact's own chunking
constexpr uint32_t kWfChunk = 256u * kWfE;

// Ranks the entries (lane, k) with take[k] set in (k, wave, lane) order: off[k] is the
// position of this thread's k-th entry among them, the return value their number.
// Ballots rank the lanes of each wave, a scan over the kWfE x 4 (k, wave) groups in LDS
// ranks those.  Must be reached by every thread of the block; lds holds kWfE * 4 + 1 words.
template <int E>
__device__ __forceinline__ uint32_t block_rank(const bool (&take)[E], uint32_t (&off)[E], uint32_t *lds) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t rank[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const unsigned long long m = __ballot(take[k]);
    rank[k] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) lds[k * 4 + wave] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < E * 4; ++j) {
      const uint32_t c = lds[j];
      lds[j] = run;
      run += c;
    }
    lds[E * 4] = run;
  }
  __syncthreads();
  const uint32_t total = lds[E * 4];
#pragma unroll
  for (int k = 0; k < E; ++k) off[k] = lds[k * 4 + wave] + rank[k];
  __syncthreads();
  return total;
}

// Appends v[k] to the global queue q for every entry with take[k] set: block_rank plus ONE
// returning atomic per block for the output range (used once per frame, by k_wf_compact).
template <int E>
__device__ __forceinline__ void block_push(const bool (&take)[E], const uint32_t (&v)[E], uint32_t *q,
                                           uint32_t *count, uint32_t *lds) {
  __shared__ uint32_t base_slot;
  uint32_t off[E];
  const uint32_t total = block_rank(take, off, lds);
  if (threadIdx.x == 0) base_slot = total ? atomicAdd(count, total) : 0u;
  __syncthreads();
  const uint32_t base = base_slot;
#pragma unroll
  for (int k = 0; k < E; ++k)
    if (take[k]) q[base + off[k]] = v[k];
}

template <bool STRICT>
__global__ __launch_bounds__(256) void k_wf_init(Params P, uint32_t frame, WfView V) {
  const uint32_t n = P.width * P.height;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t px = i % P.width, py = i / P.width;
  uint32_t rng = pixel_seed(P, px, py, frame);
  f3 o, d;
  sample_ray<STRICT>(P, pixel_direction(P, px, py), rng, o, d);
  V.P0[i] = make_float4(o.x, o.y, o.z, __uint_as_float(rng));
  V.P1[i] = make_float4(d.x, d.y, d.z, __uint_as_float(0u));
  V.P2[i] = make_float4(1.f, 1.f, 1.f, __int_as_float(-1));
  V.P3[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  V.T[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  V.alive[i] = 1;
}

template <bool BVH, bool APPROX, bool DIEL>
__global__ __launch_bounds__(256) void k_wf_extend(Scene S, WfView V, uint32_t slot, uint32_t n_pixels,
                                                   int32_t *ids) {
  __shared__ uint32_t lds[kWfEBrute * 4 + 1];
  __shared__ int stack[bvh_stack_words<BVH>()];
  uint32_t *cnt = V.counts + 4u * (slot & 3u);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    uint32_t *z = V.counts + 4u * ((slot + 1u) & 3u);
    z[1] = 0u;
    z[2] = 0u;
  }
  const uint32_t first = blockIdx.x * (256u * kWfEBrute);
  bool is_hit[kWfEBrute], is_miss[kWfEBrute];
  uint32_t pix[kWfEBrute];
#pragma unroll
  for (int k = 0; k < kWfEBrute; ++k) {
    pix[k] = first + (uint32_t)k * 256u + threadIdx.x;
    is_hit[k] = is_miss[k] = false;
    if (pix[k] < n_pixels && V.alive[pix[k]]) {
      const float4 p0 = V.P0[pix[k]], p1 = V.P1[pix[k]];
      const int skip = (S.n_tris || DIEL) ? __float_as_int(V.P2[pix[k]].w) : -1;
      const Hit h = closest_hit<BVH, APPROX, false, DIEL>(S, mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), stack, skip);
      V.H[pix[k]] = make_float2(h.dst, __int_as_float(h.idx));
      if (__float_as_uint(p1.w) == 0u) ids[pix[k]] = h.idx;  // sample 0, bounce 0
      is_hit[k] = h.idx >= 0;
      is_miss[k] = !is_hit[k];
    }
  }
  // this block's slice of the queue: hits from the front, misses from the back
  uint32_t off[kWfEBrute];
  uint32_t *slice = V.q + first;
  const uint32_t n_hit = block_rank(is_hit, off, lds);
#pragma unroll
  for (int k = 0; k < kWfEBrute; ++k)
    if (is_hit[k]) slice[off[k]] = pix[k];
  const uint32_t n_miss = block_rank(is_miss, off, lds);
#pragma unroll
  for (int k = 0; k < kWfEBrute; ++k)
    if (is_miss[k]) slice[(256u * kWfEBrute) - n_miss + off[k]] = pix[k];
  if (threadIdx.x == 0) {
    V.slice[blockIdx.x] = make_uint2(n_hit, n_miss);
    if (n_hit) atomicAdd(cnt + 1, n_hit);    // totals for the host; nobody waits on them
    if (n_miss) atomicAdd(cnt + 2, n_miss);
  }
}

// Extend through the LBVH.  The plain kernel above would run one whole tree search per lane
// and entry, every lane waiting for the slowest search of its wave (14 % of the lanes busy
// on the 10k-sphere scene).  Here the searches of a thread's kWfE entries are strung into one
// wave-scheduled loop (see rt_traverse.h): inner and leaf steps as in k_mega_sched, and a
// "next" step that stores the finished search and starts the thread's next live pixel.
template <bool APPROX, bool DIEL>
__global__ __launch_bounds__(256, RT_WF_TREE_WAVES) void k_wf_extend_tree(Scene S, WfView V, uint32_t slot,
                                                        uint32_t n_pixels, int32_t *ids) {
  __shared__ uint32_t lds[kWfETree * 4 + 1];
  extern __shared__ int stack[];  // [entry][thread], S.stack_entries entries (sized by the launch)
  uint32_t *cnt = V.counts + 4u * (slot & 3u);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    uint32_t *z = V.counts + 4u * ((slot + 1u) & 3u);
    z[1] = 0u;
    z[2] = 0u;
  }
  const uint32_t first = blockIdx.x * (256u * kWfETree);
  uint32_t hit_mask = 0, miss_mask = 0;  // bit k: entry k of this thread hit / missed
  uint32_t entry_pix[kWfETree] = {};     // the pixel of entry k (pixel order, or the regrouped order)
  int k = -1;                            // entry in flight; kWfETree: none left
  bool sample0 = false;
  uint32_t pix = 0;
  f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
  int skip = -1;
  Search<APPROX> Q;
  Q.node = kSearchOver;
  Q.sp = 0;
  Q.a = 1.f;
  Q.inv_d = Q.oi = mk3(0.f, 0.f, 0.f);
  Q.best.reset(1.f);

  while (true) {
    const bool busy = k >= 0 && k < kWfETree;
    const bool want_inner = busy && Q.node >= 0;
    const bool want_next = k < kWfETree && (!busy || Q.node == kSearchOver);
    const bool want_leaf = busy && !want_inner && Q.node != kSearchOver;
    const int n_inner = __builtin_popcountll(__ballot(want_inner));
    const int n_leaf = __builtin_popcountll(__ballot(want_leaf));
    const int n_next = __builtin_popcountll(__ballot(want_next));
    if (n_inner + n_leaf + n_next == 0) break;
    const int w_inner = n_inner * RT_SCHED_W_INNER, w_leaf = n_leaf * RT_SCHED_W_LEAF,
              w_next = n_next * RT_SCHED_W_SHADE;
    if (w_inner >= w_leaf && w_inner >= w_next) {
      if (want_inner) {
        search_inner_step(S, Q, stack);
#pragma unroll
        for (int rep = 1; rep < RT_SCHED_INNER_REPS; ++rep)
          if (Q.node >= 0) search_inner_step(S, Q, stack);
      }
    } else if (w_leaf >= w_next) {
      if (want_leaf) {
        leaf_range<DIEL>(S, Q.node, o, d, Q.a, skip, Q.best);
        Q.node = search_pop(Q, stack);
      }
    } else if (want_next) {
      if (busy) {  // the search of entry k is over: record it
        const Hit h = Q.best.result();
        V.H[pix] = make_float2(h.dst, __int_as_float(h.idx));
        if (sample0) ids[pix] = h.idx;
        if (h.idx >= 0) hit_mask |= 1u << k;
        else miss_mask |= 1u << k;
      }
      // the thread's next live pixel, if any
      for (++k; k < kWfETree; ++k) {
        const uint32_t e = first + (uint32_t)k * 256u + threadIdx.x;
        if (e >= n_pixels) continue;
        pix = V.order ? V.order[e] : e;
        entry_pix[k] = pix;
        if (V.alive[pix]) break;
      }
      if (k < kWfETree) {
        const float4 p0 = V.P0[pix], p1 = V.P1[pix];
        skip = (S.n_tris || DIEL) ? __float_as_int(V.P2[pix].w) : -1;
        sample0 = __float_as_uint(p1.w) == 0u;  // sample 0, bounce 0
        o = mk3(p0.x, p0.y, p0.z);
        d = mk3(p1.x, p1.y, p1.z);
        search_begin<DIEL>(S, o, d, skip, Q);
      }
    }
  }

  bool is_hit[kWfETree], is_miss[kWfETree];
  uint32_t pixk[kWfETree];
#pragma unroll
  for (int j = 0; j < kWfETree; ++j) {
    is_hit[j] = (hit_mask >> j) & 1u;
    is_miss[j] = (miss_mask >> j) & 1u;
    pixk[j] = entry_pix[j];
  }
  uint32_t off[kWfETree];
  uint32_t *slice = V.q + first;
  const uint32_t n_hit = block_rank(is_hit, off, lds);
#pragma unroll
  for (int j = 0; j < kWfETree; ++j)
    if (is_hit[j]) slice[off[j]] = pixk[j];
  const uint32_t n_miss = block_rank(is_miss, off, lds);
#pragma unroll
  for (int j = 0; j < kWfETree; ++j)
    if (is_miss[j]) slice[(256u * kWfETree) - n_miss + off[j]] = pixk[j];
  if (threadIdx.x == 0) {
    V.slice[blockIdx.x] = make_uint2(n_hit, n_miss);
    if (n_hit) atomicAdd(cnt + 1, n_hit);
    if (n_miss) atomicAdd(cnt + 2, n_miss);
  }
}

// A path ended with radiance `light`: fold it into the pixel's sample sum and either
// start the next sample (returns true, state written) or finish the pixel's frame.
template <bool STRICT>
__device__ __forceinline__ bool path_end(const Params &P, const WfView &V, const Outputs &O,
                                         uint32_t frame, uint32_t pix, uint32_t rng, uint32_t s,
                                         f3 light) {
  float4 t = V.T[pix];
  f3 total = mk3(t.x, t.y, t.z) + light;
  ++s;
  if (s >= (uint32_t)P.n_spp) {
    float4 acc = O.accum[pix];
    uchar4 img = O.rgba8[pix];
    frame_epilogue(P, total, (float)frame, O.linear_accum, acc, img);
    O.accum[pix] = acc;
    O.rgba8[pix] = img;
    V.alive[pix] = 0;
    return false;
  }
  V.T[pix] = make_float4(total.x, total.y, total.z, 0.f);
  const uint32_t px = pix % P.width, py = pix / P.width;
  f3 o, d;
  sample_ray<STRICT>(P, pixel_direction(P, px, py), rng, o, d);
  V.P0[pix] = make_float4(o.x, o.y, o.z, __uint_as_float(rng));
  V.P1[pix] = make_float4(d.x, d.y, d.z, __uint_as_float(s << 16));
  V.P2[pix] = make_float4(1.f, 1.f, 1.f, __int_as_float(-1));
  V.P3[pix] = make_float4(0.f, 0.f, 0.f, 0.f);
  return true;
}

template <bool STRICT, int EXT, int E>
__global__ __launch_bounds__(256) void k_wf_shade(Params P, Scene S, WfView V, uint32_t slot,
                                                  uint32_t frame, Outputs O) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const uint32_t *cnt = V.counts + 4u * (slot & 3u);
    const uint32_t traced = cnt[1] + cnt[2];
    if (traced) add_rays(O, slot, traced);
  }
  const uint2 fill = V.slice[blockIdx.x];
  const uint32_t n_hit = fill.x, first_miss = (256u * E) - fill.y;
  const uint32_t *slice = V.q + blockIdx.x * (256u * E);
#pragma unroll 2
  for (int k = 0; k < E; ++k) {
    const uint32_t e = (uint32_t)k * 256u + threadIdx.x;
    if (e < n_hit) {
      const uint32_t px = slice[e];
      const float4 p0 = V.P0[px], p1 = V.P1[px], p2 = V.P2[px], p3 = V.P3[px];
      const float2 hrec = V.H[px];
      const int idx = __float_as_int(hrec.y);
      uint32_t rng = __float_as_uint(p0.w);
      const uint32_t sb = __float_as_uint(p1.w);
      uint32_t s = sb >> 16, b = sb & 0xffffu;
      f3 o = mk3(p0.x, p0.y, p0.z), d = mk3(p1.x, p1.y, p1.z);
      f3 colour = mk3(p2.x, p2.y, p2.z), light = mk3(p3.x, p3.y, p3.z);
      Hit h{};
      h.dst = hrec.x;
      h.idx = idx;
      const int skip = shade_hit<STRICT, EXT>(S, h, __float_as_int(p2.w), rng, o, d, light, colour);
      ++b;
      if (b >= (uint32_t)P.n_bounce) {
        path_end<STRICT>(P, V, O, frame, px, rng, s, light);
      } else {
        V.P0[px] = make_float4(o.x, o.y, o.z, __uint_as_float(rng));
        V.P1[px] = make_float4(d.x, d.y, d.z, __uint_as_float((s << 16) | b));
        V.P2[px] = make_float4(colour.x, colour.y, colour.z, __int_as_float(skip));
        V.P3[px] = make_float4(light.x, light.y, light.z, 0.f);
      }
    } else if (e >= first_miss) {
      const uint32_t px = slice[e];
      const float4 p0 = V.P0[px], p1 = V.P1[px], p2 = V.P2[px], p3 = V.P3[px];
      f3 light = fma3v(environment<STRICT>(P, mk3(p1.x, p1.y, p1.z)), mk3(p2.x, p2.y, p2.z),
                       mk3(p3.x, p3.y, p3.z));
      path_end<STRICT>(P, V, O, frame, px, __float_as_uint(p0.w), __float_as_uint(p1.w) >> 16, light);
    }
  }
}

// Gathers the pixels that are still alive into the drain queue (q, counts[16]).
__global__ __launch_bounds__(256) void k_wf_compact(WfView V, uint32_t n_pixels) {
  __shared__ uint32_t lds[kWfE * 4 + 1];
  const uint32_t first = blockIdx.x * kWfChunk;
  bool take[kWfE];
  uint32_t pix[kWfE];
#pragma unroll
  for (int k = 0; k < kWfE; ++k) {
    pix[k] = first + (uint32_t)k * 256u + threadIdx.x;
    take[k] = pix[k] < n_pixels && V.alive[pix[k]] != 0;
  }
  block_push(take, pix, V.q, V.counts + 16, lds);
}

// Finishes every pixel of the drain queue in registers: the loop of the regenerating
// megakernel, resumed from the stored path state, for this one frame.
template <bool STRICT, bool BVH, int EXT, bool APPROX>
__global__ __launch_bounds__(256) void k_wf_drain(Params P, Scene S, WfView V, uint32_t frame,
                                                  Outputs O) {
  constexpr bool DIEL = EXT == kExtDielectric;
  __shared__ int stack[bvh_stack_words<BVH>()];
  const uint32_t n = V.counts[16];
  const uint32_t qi = blockIdx.x * 256u + threadIdx.x;
  uint32_t nrays = 0;
  if (qi < n) {
    const uint32_t pix = V.q[qi];
    const float4 p0 = V.P0[pix], p1 = V.P1[pix], p2 = V.P2[pix], p3 = V.P3[pix], t = V.T[pix];
    const uint32_t px = pix % P.width, py = pix / P.width;
    const f3 base = pixel_direction(P, px, py);
    uint32_t rng = __float_as_uint(p0.w);
    const uint32_t sb = __float_as_uint(p1.w);
    int s = (int)(sb >> 16), b = (int)(sb & 0xffffu);
    f3 o = mk3(p0.x, p0.y, p0.z), d = mk3(p1.x, p1.y, p1.z);
    f3 colour = mk3(p2.x, p2.y, p2.z), light = mk3(p3.x, p3.y, p3.z);
    f3 total = mk3(t.x, t.y, t.z);
    int id0 = -2;
    int skip = __float_as_int(p2.w);
    while (true) {
      // the drain is the chain of the slowest pixels: the brute-force search in its form for a
      // wave running alone, the sphere's centre carried with the hit (as k_mega_regen's plain loop)
      const Hit h = closest_hit<BVH, APPROX, !BVH, DIEL>(S, o, d, stack, skip);
      ++nrays;
      if (s == 0 && b == 0) id0 = h.idx;
      bool path_done;
      if (h.idx >= 0) {
        skip = shade_hit<STRICT, EXT, APPROX, !BVH>(S, h, skip, rng, o, d, light, colour);
        ++b;
        path_done = b >= P.n_bounce;
      } else {
        light = fma3v(environment<STRICT>(P, d), colour, light);
        path_done = true;
      }
      if (path_done) {
        total = total + light;
        ++s;
        if (s >= P.n_spp) break;
        sample_ray<STRICT>(P, base, rng, o, d);
        light = mk3(0.f, 0.f, 0.f);
        colour = mk3(1.f, 1.f, 1.f);
        b = 0;
        skip = -1;
      }
    }
    float4 acc = O.accum[pix];
    uchar4 img = O.rgba8[pix];
    frame_epilogue(P, total, (float)frame, O.linear_accum, acc, img);
    O.accum[pix] = acc;
    O.rgba8[pix] = img;
    if (id0 != -2) O.ids[pix] = id0;
  }
  const uint32_t wsum = wave_sum_u32(nrays);
  if ((threadIdx.x & 63u) == 0 && wsum) add_rays(O, wave_index(), wsum);
}

// The drain launch for LBVH scenes: the same resumption, in k_mega_sched's wave-scheduled
// form (inner / leaf / shade steps chosen by waiting-lane count).
template <bool STRICT, int EXT, bool APPROX>
__global__ __launch_bounds__(256, RT_WF_TREE_WAVES) void k_wf_drain_tree(Params P, Scene S, WfView V, uint32_t frame,
                                                       Outputs O) {
  constexpr bool DIEL = EXT == kExtDielectric;
  extern __shared__ int stack[];  // [entry][thread], S.stack_entries entries (sized by the launch)
  const uint32_t n = V.counts[16];
  const uint32_t qi = blockIdx.x * 256u + threadIdx.x;
  uint32_t nrays = 0;
  bool active = qi < n;
  const uint32_t pix = active ? V.q[qi] : 0u;
  const uint32_t px = pix % P.width, py = pix / P.width;
  const f3 base = pixel_direction(P, px, py);
  uint32_t rng = 0;
  int s = 0, b = 0, skip = -1, id0 = -2;
  f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
  f3 colour = mk3(1.f, 1.f, 1.f), light = mk3(0.f, 0.f, 0.f), total = mk3(0.f, 0.f, 0.f);
  if (active) {
    const float4 p0 = V.P0[pix], p1 = V.P1[pix], p2 = V.P2[pix], p3 = V.P3[pix], t = V.T[pix];
    rng = __float_as_uint(p0.w);
    const uint32_t sb = __float_as_uint(p1.w);
    s = (int)(sb >> 16);
    b = (int)(sb & 0xffffu);
    o = mk3(p0.x, p0.y, p0.z);
    d = mk3(p1.x, p1.y, p1.z);
    colour = mk3(p2.x, p2.y, p2.z);
    light = mk3(p3.x, p3.y, p3.z);
    total = mk3(t.x, t.y, t.z);
    skip = __float_as_int(p2.w);
  }
  Search<APPROX> Q;
  search_begin<DIEL>(S, o, d, skip, Q);

  while (true) {
    const bool want_inner = active && Q.node >= 0;
    const bool want_shade = active && Q.node == kSearchOver;
    const bool want_leaf = active && !want_inner && !want_shade;
    const int n_inner = __builtin_popcountll(__ballot(want_inner));
    const int n_leaf = __builtin_popcountll(__ballot(want_leaf));
    const int n_shade = __builtin_popcountll(__ballot(want_shade));
    if (n_inner + n_leaf + n_shade == 0) break;
    const int w_inner = n_inner * RT_SCHED_W_INNER, w_leaf = n_leaf * RT_SCHED_W_LEAF,
              w_shade = n_shade * RT_SCHED_W_SHADE;
    if (w_inner >= w_leaf && w_inner >= w_shade) {
      if (want_inner) {
        search_inner_step(S, Q, stack);
#pragma unroll
        for (int rep = 1; rep < RT_SCHED_INNER_REPS; ++rep)
          if (Q.node >= 0) search_inner_step(S, Q, stack);
      }
    } else if (w_leaf >= w_shade) {
      if (want_leaf) {
        leaf_range<DIEL>(S, Q.node, o, d, Q.a, skip, Q.best);
        Q.node = search_pop(Q, stack);
      }
    } else if (want_shade) {
      const Hit h = Q.best.result();
      ++nrays;
      if (s == 0 && b == 0) id0 = h.idx;
      bool path_done;
      if (h.idx >= 0) {
        skip = shade_hit<STRICT, EXT, APPROX>(S, h, skip, rng, o, d, light, colour);
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
          active = false;
        } else {
          sample_ray<STRICT>(P, base, rng, o, d);
          light = mk3(0.f, 0.f, 0.f);
          colour = mk3(1.f, 1.f, 1.f);
          b = 0;
          skip = -1;
        }
      }
      if (active) {
        search_begin<DIEL>(S, o, d, skip, Q);
      } else {
        float4 acc = O.accum[pix];
        uchar4 img = O.rgba8[pix];
        frame_epilogue(P, total, (float)frame, O.linear_accum, acc, img);
        O.accum[pix] = acc;
        O.rgba8[pix] = img;
        if (id0 != -2) O.ids[pix] = id0;
      }
    }
  }
  const uint32_t wsum = wave_sum_u32(nrays);
  if ((threadIdx.x & 63u) == 0 && wsum) add_rays(O, wave_index(), wsum);
}

// ---- host side ---------------------------------------------------------------------
struct WavefrontState::Impl {
  size_t n_pixels = 0;
  char *slab = nullptr;       // one allocation carved into the arrays below
  WfView view{};
  uint32_t *h_counts = nullptr;  // pinned read-back ring
  hipEvent_t ev[8] = {};
  // regrouping experiment: sort buffers (allocated on first use)
  char *sort_slab = nullptr;
  size_t sort_n = 0;
  uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *hist = nullptr, *sums = nullptr;
};

static constexpr int kCheckEvery = 4;   // iterations between read-backs of the active count
static constexpr int kReadbackLag = 1;  // read-backs the host may run ahead of the device

void WavefrontState::release() {
  if (!impl) return;
  if (impl->slab) hipFree(impl->slab);
  if (impl->sort_slab) hipFree(impl->sort_slab);
  if (impl->h_counts) hipHostFree(impl->h_counts);
  for (auto &e : impl->ev)
    if (e) hipEventDestroy(e);
  delete impl;
  impl = nullptr;
}

#define WF_HIP(call)                                                          \
  do {                                                                        \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess) {                                                   \
      if (err) *err = std::string(#call) + ": " + hipGetErrorString(e_);      \
      return -2;                                                              \
    }                                                                         \
  } while (0)

// On failure the state stays consistent: whatever is missing (read-back ring, events, slab) is
// created by the next call, and n_pixels is 0 until the slab exists.
static int wf_allocate(WavefrontState::Impl *&impl, size_t n, std::string *err) {
  if (impl && impl->slab && impl->n_pixels == n) return 0;
  if (!impl) impl = new WavefrontState::Impl();
  if (impl->slab) hipFree(impl->slab);
  impl->slab = nullptr;
  impl->n_pixels = 0;
  if (!impl->h_counts) WF_HIP(hipHostMalloc((void **)&impl->h_counts, 16 * sizeof(uint32_t)));
  for (auto &e : impl->ev)
    if (!e) WF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const size_t slices = (n + 256 * kWfETree - 1) / (256 * kWfETree);  // the finer slicing
  const size_t f4 = n * sizeof(float4), q = (n + 256 * kWfEBrute) * sizeof(uint32_t);
  const size_t sl = ((slices * sizeof(uint2) + 255) / 256) * 256;
  const size_t h = ((n * sizeof(float2) + 255) / 256) * 256;
  const size_t a = ((n + 255) / 256) * 256;
  const size_t total = 5 * f4 + h + q + sl + a + 256;
  WF_HIP(hipMalloc((void **)&impl->slab, total));
  char *p = impl->slab;
  WfView &V = impl->view;
  V.P0 = (float4 *)p; p += f4;
  V.P1 = (float4 *)p; p += f4;
  V.P2 = (float4 *)p; p += f4;
  V.P3 = (float4 *)p; p += f4;
  V.T = (float4 *)p; p += f4;
  V.H = (float2 *)p; p += h;
  V.alive = (unsigned char *)p; p += a;
  V.q = (uint32_t *)p; p += q;
  V.slice = (uint2 *)p; p += sl;
  V.counts = (uint32_t *)p;
  impl->n_pixels = n;
  return 0;
}

static int regroup_allocate(WavefrontState::Impl *impl, size_t n, std::string *err) {
  if (impl->sort_slab && impl->sort_n == n) return 0;
  if (impl->sort_slab) hipFree(impl->sort_slab);
  impl->sort_slab = nullptr;
  impl->sort_n = 0;
  size_t hist = 0, sums = 0;
  radix_sort_scratch((uint32_t)n, &hist, &sums);
  const size_t u4 = (n * sizeof(uint32_t) + 255) / 256 * 256;
  hist = (hist + 255) / 256 * 256;
  sums = (sums + 255) / 256 * 256;
  WF_HIP(hipMalloc((void **)&impl->sort_slab, 4 * u4 + hist + sums));
  char *p = impl->sort_slab;
  impl->keys[0] = (uint32_t *)p; p += u4;
  impl->keys[1] = (uint32_t *)p; p += u4;
  impl->vals[0] = (uint32_t *)p; p += u4;
  impl->vals[1] = (uint32_t *)p; p += u4;
  impl->hist = (uint32_t *)p; p += hist;
  impl->sums = (uint32_t *)p;
  impl->sort_n = n;
  return 0;
}

template <bool STRICT, bool BVH, int EXT, bool APPROX>
static int wf_frame(WavefrontState::Impl *impl, uint32_t drain_threshold, const RegroupGrid *regroup, const Params &P,
                    const Scene &S, uint32_t frame, const Outputs &O, hipStream_t stream,
                    uint32_t *launches, std::string *err) {
  WfView V = impl->view;
  V.order = nullptr;
  const uint32_t n = (uint32_t)impl->n_pixels;
  const uint32_t full_blocks = (n + 255u) / 256u;
  constexpr int E = BVH ? kWfETree : kWfEBrute;
  const uint32_t window_blocks = (n + 256u * E - 1u) / (256u * E);
  const uint32_t compact_blocks = (n + kWfChunk - 1u) / kWfChunk;
  // LDS of the scheduled tree kernels: one traversal stack per thread, as deep as the tree
  const size_t tree_stack_bytes = (size_t)(S.stack_entries ? S.stack_entries : 1u) * 256u * sizeof(int);
  WF_HIP(hipMemsetAsync(V.counts, 0, 128, stream));
  hipLaunchKernelGGL((k_wf_init<STRICT>), dim3(full_blocks), dim3(256), 0, stream, P, frame, V);
  ++*launches;
  uint32_t bound = n;  // upper bound of the live-pixel count (it never grows within a frame)
  // worst case: every sample of the slowest pixel runs to the bounce limit
  const uint64_t max_iters = (uint64_t)P.n_spp * (uint64_t)P.n_bounce;
  int pending = 0;     // read-backs issued and not yet consumed
  uint32_t issue = 0, consume = 0;
  for (uint64_t it = 0; it < max_iters; ++it) {
    const uint32_t slot = (uint32_t)(it & 3u);
    if (bound <= drain_threshold) {
      hipLaunchKernelGGL(k_wf_compact, dim3(compact_blocks), dim3(256), 0, stream, V, n);
      if constexpr (BVH)
        hipLaunchKernelGGL((k_wf_drain_tree<STRICT, EXT, APPROX>), dim3((bound + 255u) / 256u), dim3(256), tree_stack_bytes,
                           stream, P, S, V, frame, O);
      else
        hipLaunchKernelGGL((k_wf_drain<STRICT, BVH, EXT, APPROX>), dim3((bound + 255u) / 256u), dim3(256), 0,
                           stream, P, S, V, frame, O);
      *launches += 2;
      break;
    }
    if constexpr (BVH) {
      if (regroup) {
        hipLaunchKernelGGL(k_wf_keys, dim3(full_blocks), dim3(256), 0, stream, V, n, *regroup, impl->keys[0], impl->vals[0]);
        WF_HIP(radix_sort_pairs(impl->keys, impl->vals, n, 2, impl->hist, impl->sums, stream));
        V.order = impl->vals[0];  // two passes: back in the first pair of buffers
        *launches += 11;
      }
      hipLaunchKernelGGL((k_wf_extend_tree<APPROX, EXT == kExtDielectric>), dim3(window_blocks), dim3(256), tree_stack_bytes, stream, S, V, slot, n,
                         O.ids);
    }
    else
      hipLaunchKernelGGL((k_wf_extend<BVH, APPROX, EXT == kExtDielectric>), dim3(window_blocks), dim3(256), 0, stream, S, V, slot, n,
                         O.ids);
    hipLaunchKernelGGL((k_wf_shade<STRICT, EXT, E>), dim3(window_blocks), dim3(256),
                       0, stream, P, S, V, slot, frame, O);
    *launches += 2;
    if ((it + 1) % kCheckEvery == 0) {
      // hit and miss counts of this iteration (their sum = live pixels) -> pinned ring,
      // consumed one read-back later so the host runs ahead of the device
      const uint32_t r = issue++ & 7u;
      WF_HIP(hipMemcpyAsync(&impl->h_counts[2 * r], V.counts + 4u * slot + 1u, 2 * sizeof(uint32_t),
                            hipMemcpyDeviceToHost, stream));
      WF_HIP(hipEventRecord(impl->ev[r], stream));
      ++pending;
      if (pending > kReadbackLag) {
        const uint32_t c = consume++ & 7u;
        WF_HIP(hipEventSynchronize(impl->ev[c]));
        --pending;
        const uint32_t live = impl->h_counts[2 * c] + impl->h_counts[2 * c + 1];
        if (live == 0u) break;
        if (live < bound) bound = live;
      }
    }
  }
  WF_HIP(hipGetLastError());
  return 0;
}

int WavefrontState::render(bool strict, bool bvh, const Params &P, const Scene &S, const FrameSeq &F,
                           const Outputs &O, hipStream_t stream, uint32_t *launches,
                           std::string *err) {
  if (P.n_spp > 65535 || P.n_bounce > 65535) {
    if (err) *err = "wavefront mode packs sample and bounce counters into 16 bits each";
    return -1;
  }
  int rc = wf_allocate(impl, (size_t)P.width * P.height, err);
  if (rc) return rc;
  RegroupGrid grid{};
  const RegroupGrid *use_grid = nullptr;
  if (regroup && bvh) {
    rc = regroup_allocate(impl, (size_t)P.width * P.height, err);
    if (rc) return rc;
    for (int k = 0; k < 3; ++k) {
      const float ext = regroup_hi[k] - regroup_lo[k];
      grid.lo[k] = regroup_lo[k];
      grid.scale[k] = ext > 0.f ? 16.0f / ext : 0.f;
    }
    use_grid = &grid;
  }
  *launches = 0;
  for (uint32_t fi = 0; fi < F.count; ++fi) {
    const uint32_t frame = F.first + fi * F.stride;
    const bool approx = !strict && S.n_tris == 0u;  // see launch_megakernel
    const int ext = S.ior ? kExtDielectric : (S.tex ? kExtTexture : kExtNone);
    const int variant = (approx ? 8 : 0) | (strict ? 4 : 0) | (bvh ? 2 : 0);
#define RT_WF_CASE(n, a, b, e)                                                                              \
  case n:                                                                                                   \
    rc = ext == kExtDielectric                                                                              \
             ? wf_frame<a, b, kExtDielectric, e>(impl, drain_threshold, use_grid, P, S, frame, O, stream, launches, err) \
         : ext == kExtTexture                                                                               \
             ? wf_frame<a, b, kExtTexture, e>(impl, drain_threshold, use_grid, P, S, frame, O, stream, launches, err)  \
             : wf_frame<a, b, kExtNone, e>(impl, drain_threshold, use_grid, P, S, frame, O, stream, launches, err);    \
    break;
    switch (variant) {
      RT_WF_CASE(0, false, false, false)
      RT_WF_CASE(2, false, true, false)
      RT_WF_CASE(4, true, false, false)
      RT_WF_CASE(6, true, true, false)
      RT_WF_CASE(8, false, false, true)
      RT_WF_CASE(10, false, true, true)
    }
#undef RT_WF_CASE
    if (rc) return rc;
  }
  return 0;
}

}  // namespace rt
