// Linear BVH over spheres and mesh triangles, built on the GPU.
//
//   1. leaf boxes (inflated, see below) + bounds of the centres       k_leaf_boxes
//   2. 30-bit Morton code of each centre (bits handed to the axes by   k_morton
//      the shape of the scene)
//   3. stable LSD radix sort of (code, index), 4 passes of 8 bits     k_rs_hist / k_scan_rows / k_rs_scatter
//   4. Karras (2012) radix-tree hierarchy over the sorted codes,      k_hierarchy
//      ties broken by sorted position
//   5. bottom-up bounds in two or three launches                      k_refit_local / k_refit_upper
//   6. nodes with both child boxes inline (one 64-byte fetch per      (written by the refit kernels)
//      traversal step), geometry permuted into leaf order             k_permute
//
// Why the leaf boxes are inflated.  The closest-hit search must return exactly what the
// shader's brute-force loop returns, and that loop runs the quadratic in fp32: for a ray
// whose origin is |oc| away from a sphere of radius r the discriminant carries an
// absolute error of a few ulp of |oc|^2, so the float test reports hits for rays that
// pass up to ~3 eps |oc|^2 / r OUTSIDE the sphere.  A tight box would cull those.  Each
// leaf box is therefore a cube of half-width sqrt(r^2 + 9.6e-7 D^2) + r / 1024, where D bounds
// the distance from the sphere to any ray origin (camera and the region the ordinary-
// sized spheres occupy).
//
// Oversized spheres (radius above a tenth of the extent of all centres, at most 16 of them:
// a ground sphere, a sky dome) stay out of the tree: their box would cover every ancestor
// on their root path and stretch the Morton grid.  Their leaves become null leaves with an
// empty box and the traversal tests them first from a short list.
//
// Refit without hand-offs between workgroups.  A node of the radix tree covers a contiguous run
// of sorted positions, and its index is one end of that run.  k_refit_local gives every block a
// tile of kRefitTile consecutive positions and lets it finish every node whose run lies inside the
// tile (its children's runs then do too, so all it needs comes from its own block): walks climb
// from the leaves and meet at the nodes through tokens in the block's LDS (see climb).  The nodes
// that straddle a tile boundary -- about a dozen per boundary -- go onto a list with their runs.
// k_refit_upper finishes the listed nodes the same way, first (when there are several) inside tiles
// of 2^RT_REFIT_MID_SHIFT positions, one block per tile, then, in ONE block, whatever is left: their
// children are either listed too and lie in the same tile (same block) or were finished by an
// earlier launch.  No fences, no flags polled across CUs, no dependence on dispatch order or on
// which XCD a block runs, and ONE look at the device from the host, when the tree is complete.
// (It was 16 launches of a kernel over all nodes and a read-back per round of them: 224 of the 460
// microseconds of a million-triangle build; then rounds separated by barriers in LDS, 140; the
// walks take 83, and the build 270.)
#include "rt_lbvh.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace rt {

namespace {

constexpr int kThreads = 256;
constexpr int kRsE = 8;                         // keys per thread per radix pass
constexpr uint32_t kRsTile = kThreads * kRsE;   // keys per block

struct BoxParams {
  float blo[3], bhi[3];  // region ray origins are assumed to stay in
  float large_radius;    // spheres above this radius are kept out of the tree
};

// order-preserving map float -> uint for atomicMin / atomicMax
__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t u) {
  uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __uint_as_float(b);
#else
  std::memcpy(&f, &b, 4);
#endif
  return f;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Primitive i < n_spheres is sphere i (cr = centre.xyz, radius); the others are the mesh
// triangles (tri = v0, v1, v2 per triangle).  cbounds = 6 ordered words: min xyz, max xyz of
// the box centres of the primitives that go into the tree.
__global__ __launch_bounds__(kThreads) void k_leaf_boxes(const float4 *__restrict__ cr,
                                                         uint32_t n_spheres,
                                                         const float4 *__restrict__ tri, uint32_t n,
                                                         BoxParams bp, float4 *__restrict__ leaf_lo,
                                                         float4 *__restrict__ leaf_hi,
                                                         uint32_t *cbounds) {
  float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    float lo[3], hi[3], c[3];
    if (i < n_spheres) {
      const float4 s = cr[i];
      const float r = fabsf(s.w);
      if (r > bp.large_radius) {  // null leaf: empty box, flagged in .w
        leaf_lo[i] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 1.f);
        leaf_hi[i] = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), 0.f);
        continue;
      }
      c[0] = s.x; c[1] = s.y; c[2] = s.z;
      float d2 = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float far = fmaxf(fabsf(c[k] - bp.blo[k]), fabsf(c[k] - bp.bhi[k]));
        d2 += far * far;
      }
      // The float test accepts a ray when its computed discriminant a (r^2 - dist^2) + error is >= 0,
      // with |error| up to a few ulp of a |oc|^2: rays pass that run up to sqrt(r^2 + 9.6e-7 D^2) from
      // the centre.  For an ordinary sphere that is r (1 + 4.8e-7 D^2 / r^2); for a tiny one (down to
      // radius 0, which the in-order loop does hit with rays aimed at its centre) it is ~1e-3 D,
      // however small r is.  Plus r / 1024 for the slab test's own rounding.
      const float h = sqrtf(__builtin_fmaf(r, r, 9.6e-7f * d2)) + r * 0x1p-10f + 1e-30f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        lo[k] = c[k] - h;
        hi[k] = c[k] + h;
      }
    } else {
      // a triangle's float test errs by ~eps * (distance to the ray origin) at its edges
      const float4 *T = tri + 3u * (i - n_spheres);
      const float4 a = T[0], b = T[1], cc = T[2];
      const float v[3][3] = {{a.x, a.y, a.z}, {b.x, b.y, b.z}, {cc.x, cc.y, cc.z}};
      float d2 = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        lo[k] = fminf(fminf(v[0][k], v[1][k]), v[2][k]);
        hi[k] = fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]);
        c[k] = 0.5f * (lo[k] + hi[k]);
        const float far = fmaxf(fabsf(c[k] - bp.blo[k]), fabsf(c[k] - bp.bhi[k])) + (hi[k] - lo[k]);
        d2 += far * far;
      }
      // ... and by more at a corner with a flat angle: the barycentric tests divide rounding
      // residue by |e1 x e2|, so the margin grows with |e1| |e2| / |e1 x e2| (1 for a right angle
      // at the first vertex; capped, which makes the box of a needle cover the scene)
      const float e1[3] = {b.x - a.x, b.y - a.y, b.z - a.z}, e2[3] = {cc.x - a.x, cc.y - a.y, cc.z - a.z};
      const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2],
                  cz = e1[0] * e2[1] - e1[1] * e2[0];
      const float l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
      const float area2 = cx * cx + cy * cy + cz * cz;
      float aspect = area2 > 0.f ? sqrtf(l1 * l2 / area2) : 1.f;   // (no area at all: never hit, see k_expand_triangles)
      aspect = fminf(fmaxf(aspect, 1.f), 1048576.f);
      const float m = 4.0e-6f * sqrtf(d2) * aspect + 1e-30f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        lo[k] -= m;
        hi[k] += m;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      mn[k] = fminf(mn[k], c[k]);
      mx[k] = fmaxf(mx[k], c[k]);
    }
    leaf_lo[i] = make_float4(lo[0], lo[1], lo[2], 0.f);
    leaf_hi[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
  }
  // one atomic per block and bound: same-address atomics retire at ~90 per microsecond, so
  // the grid is kept small (256 blocks) and the four waves of a block combine in LDS first
  __shared__ float red[2][3][kThreads / 64];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float a = wave_min(mn[k]), b = wave_max(mx[k]);
    if ((threadIdx.x & 63) == 0) {
      red[0][k][threadIdx.x >> 6] = a;
      red[1][k][threadIdx.x >> 6] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    float a = red[0][k][0], b = red[1][k][0];
#pragma unroll
    for (int wv = 1; wv < kThreads / 64; ++wv) {
      a = fminf(a, red[0][k][wv]);
      b = fmaxf(b, red[1][k][wv]);
    }
    atomicMin(&cbounds[k], f2ord(a));
    atomicMax(&cbounds[3 + k], f2ord(b));
  }
}

// cbounds: [0..2] running minima (ordered-uint form, start at the top), [3..5] running maxima,
// [64..127] the refit's words of meta
__global__ void k_build_init(uint32_t *cbounds) {
  cbounds[threadIdx.x] = threadIdx.x < 3u ? 0xffffffffu : 0u;
}

#ifndef RT_MORTON_ADAPTIVE
#define RT_MORTON_ADAPTIVE 1  // 0: ten bits per axis on cubic cells
#endif

__global__ __launch_bounds__(kThreads) void k_morton(const float4 *__restrict__ leaf_lo,
                                                     const float4 *__restrict__ leaf_hi, uint32_t n,
                                                     const uint32_t *__restrict__ cbounds,
                                                     uint32_t *__restrict__ keys,
                                                     uint32_t *__restrict__ vals) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const float4 bl = leaf_lo[i], bh = leaf_hi[i];
  float c[3] = {0.5f * (bl.x + bh.x), 0.5f * (bl.y + bh.y), 0.5f * (bl.z + bh.z)};
  if (bl.w != 0.f) c[0] = c[1] = c[2] = 0.f;  // null leaf (empty box): any code will do
  // The 30 code bits go, one at a time from the top, to the axis whose cells are the longest at
  // that point (ties: x, then y, then z), each bit halving that axis's cells.  A cube-shaped scene
  // gets the usual x y z x y z ... interleave, 10 bits per axis; a flat one -- a terrain -- spends
  // its bits on the two axes it extends along (13 + 13 + 4 instead of 10 + 10 + 10 of which the
  // thin axis uses 5), so the lowest levels of the tree still follow where the primitives are.
  // A bit of an axis is the next binary digit of the centre's position along it (doubling a
  // float in [0, 1) and taking off the 1 are exact).
  float ex = ord2f(cbounds[3]) - ord2f(cbounds[0]), ey = ord2f(cbounds[4]) - ord2f(cbounds[1]),
        ez = ord2f(cbounds[5]) - ord2f(cbounds[2]);
#if !RT_MORTON_ADAPTIVE
  ex = ey = ez = fmaxf(fmaxf(ex, ey), ez);  // cubic cells, 10 bits per axis
#endif
  const float top = 0x1.fffffep-1f;
  float tx = ex > 0.f ? fminf(fmaxf((c[0] - ord2f(cbounds[0])) / ex, 0.f), top) : 0.f;
  float ty = ey > 0.f ? fminf(fmaxf((c[1] - ord2f(cbounds[1])) / ey, 0.f), top) : 0.f;
  float tz = ez > 0.f ? fminf(fmaxf((c[2] - ord2f(cbounds[2])) / ez, 0.f), top) : 0.f;
  uint32_t key = 0;
  for (int bit = 0; bit < 30; ++bit) {
    uint32_t digit;
    if (ex >= ey && ex >= ez) {
      ex *= 0.5f; tx *= 2.f; digit = tx >= 1.f; tx -= (float)digit;
    } else if (ey >= ez) {
      ey *= 0.5f; ty *= 2.f; digit = ty >= 1.f; ty -= (float)digit;
    } else {
      ez *= 0.5f; tz *= 2.f; digit = tz >= 1.f; tz -= (float)digit;
    }
    key = (key << 1) | digit;
  }
  keys[i] = key;
  vals[i] = i;
}

// ---- radix sort, one 8-bit digit per pass ---------------------------------------------
__global__ __launch_bounds__(kThreads) void k_rs_hist(const uint32_t *__restrict__ keys, uint32_t n,
                                                      uint32_t shift, uint32_t *__restrict__ hist,
                                                      uint32_t nblk) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t first = blockIdx.x * kRsTile;
#pragma unroll
  for (int k = 0; k < kRsE; ++k) {
    const uint32_t i = first + k * kThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];  // digit-major: scan order
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(v, off, 64);
    if (lane >= (uint32_t)off) v += up;
  }
  return v;
}

// The radix passes' own scan, in one launch: block d turns row d of the digit-major histogram (the
// counts of digit d in blocks 0, 1, ...) into its exclusive prefix, in place, and leaves the row's total
// in totals[d]; k_rs_scatter adds the 256 totals up by itself.  (A general exclusive scan of all
// 256 x blocks counters -- tile sums, a scan of those, tiles again: three launches -- cost a radix pass
// 10 of its 29 microseconds at a million keys.)
__global__ __launch_bounds__(kThreads) void k_scan_rows(uint32_t *__restrict__ hist, uint32_t nblk,
                                                        uint32_t *__restrict__ totals) {
  __shared__ uint32_t wsum[kThreads / 64];
  uint32_t *row = hist + (size_t)blockIdx.x * nblk;
  const uint32_t per = (nblk + kThreads - 1u) / kThreads;
  const uint32_t b = min(threadIdx.x * per, nblk), e = min(b + per, nblk);
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; ++i) sum += row[i];
  const uint32_t inc = wave_inclusive_scan(sum);
  if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t run = inc - sum;
  for (uint32_t wv = 0; wv < (threadIdx.x >> 6); ++wv) run += wsum[wv];
  for (uint32_t i = b; i < e; ++i) {
    const uint32_t c = row[i];
    row[i] = run;
    run += c;
  }
  if (threadIdx.x == kThreads - 1) totals[blockIdx.x] = run;
}

__global__ __launch_bounds__(kThreads) void k_rs_scatter(const uint32_t *__restrict__ keys_in,
                                                         const uint32_t *__restrict__ vals_in,
                                                         uint32_t *__restrict__ keys_out,
                                                         uint32_t *__restrict__ vals_out, uint32_t n,
                                                         uint32_t shift,
                                                         const uint32_t *__restrict__ hist,
                                                         const uint32_t *__restrict__ totals,
                                                         uint32_t nblk) {
  __shared__ uint32_t base[256];       // next output slot of each digit for this block
  __shared__ uint32_t wcount[4][256];  // per wave, keys of each digit in the current round
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  {  // keys of smaller digits anywhere (exclusive scan of the 256 totals) + keys of this digit in earlier blocks
    const uint32_t mine = totals[t];
    const uint32_t inc = wave_inclusive_scan(mine);
    if (lane == 63u) wcount[0][wave] = inc;
    __syncthreads();
    uint32_t before = inc - mine;
    for (uint32_t w = 0; w < wave; ++w) before += wcount[0][w];
    __syncthreads();
    base[t] = before + hist[t * nblk + blockIdx.x];
  }
  const uint32_t first = blockIdx.x * kRsTile;
  for (int k = 0; k < kRsE; ++k) {
    wcount[0][t] = 0; wcount[1][t] = 0; wcount[2][t] = 0; wcount[3][t] = 0;
    __syncthreads();
    const uint32_t i = first + k * kThreads + t;
    const bool valid = i < n;
    const uint32_t key = valid ? keys_in[i] : 0xffffffffu;
    const uint32_t val = valid ? vals_in[i] : 0u;
    const uint32_t digit = (key >> shift) & 255u;
    // lanes of this wave holding the same digit (stable: rank = lower lanes first)
    unsigned long long mask = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (digit >> b) & 1u;
      const unsigned long long bm = __ballot(bit);
      mask &= bit ? bm : ~bm;
    }
    const unsigned long long lower = mask & ((1ull << lane) - 1ull);
    const uint32_t rank = __popcll(lower);
    if (valid && lower == 0ull) wcount[wave][digit] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (valid) {
      uint32_t off = base[digit];
      for (uint32_t w = 0; w < wave; ++w) off += wcount[w][digit];
      keys_out[off + rank] = key;
      vals_out[off + rank] = val;
    }
    __syncthreads();
    base[t] += wcount[0][t] + wcount[1][t] + wcount[2][t] + wcount[3][t];
    __syncthreads();
  }
}

// ---- hierarchy (Karras 2012) ------------------------------------------------------------
// common-prefix length of sorted keys i and j; equal codes fall back to the position
__device__ __forceinline__ int delta(const uint32_t *__restrict__ codes, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint32_t a = codes[i], b = codes[j];
  if (a != b) return __clz(a ^ b);
  return 32 + __clz((uint32_t)i ^ (uint32_t)j);
}

// child encoding: >= 0 internal node index, < 0 leaf at sorted position ~child
__global__ __launch_bounds__(kThreads) void k_hierarchy(const uint32_t *__restrict__ codes, int n,
                                                        int *__restrict__ left, int *__restrict__ right,
                                                        int *__restrict__ range_first,
                                                        int *__restrict__ range_count) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(codes, n, i, i + 1) - delta(codes, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(codes, n, i, i - d);
  int lmax = 2;
  while (delta(codes, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(codes, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(codes, n, i, j);
  int s = 0;
  int t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(codes, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  const int lc = (lo == gamma) ? ~gamma : gamma;
  const int rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
  left[i] = lc;
  right[i] = rc;
  range_first[i] = lo;  // the node covers sorted positions lo .. hi
  range_count[i] = hi - lo + 1;
}

struct RefitView {
  const int *left, *right;
  const int *range_first, *range_count;  // sorted positions under each internal node
  const uint32_t *sorted_prim;        // sorted position -> original index
  const float4 *leaf_lo, *leaf_hi;    // by sorted position (k_permute)
  uint32_t *stamp;                    // pass in which a node was finished, 0 = not yet
  float4 *box_lo, *box_hi;            // bounds of finished nodes; .w of box_lo = subtree height
  BvhNode *nodes;
  int n_internal;
};

__device__ __forceinline__ bool child_ready(const RefitView &v, int c, uint32_t pass) {
  if (c < 0) return true;
  const uint32_t s = v.stamp[c];
  return s != 0u && s < pass;
}
// an internal node over at most kLeafMaxPrims primitives is referenced as one leaf (its own
// record is then never visited); the root stays a node
__device__ __forceinline__ void fold_small_subtree(const RefitView &v, int &c, float &height) {
  if (c > 0 && v.range_count[c] <= kLeafMaxPrims) {
    c = make_leaf_ref((uint32_t)v.range_first[c], (uint32_t)v.range_count[c]);
    height = 0.f;
  }
}
// For a leaf of an oversized sphere, c is rewritten to kNullLeaf.
__device__ __forceinline__ void child_box(const RefitView &v, int &c, float4 &lo, float4 &hi,
                                          float &height) {
  if (c < 0) {
    const uint32_t pos = (uint32_t)~c;
    lo = v.leaf_lo[pos];
    hi = v.leaf_hi[pos];
    height = 0.f;
    c = lo.w != 0.f ? kNullLeaf : make_leaf_ref(pos, 1u);
  } else {
    lo = v.box_lo[c];
    hi = v.box_hi[c];
    height = lo.w;
    fold_small_subtree(v, c, height);
  }
}
// writes node i from its two children's boxes; returns its own box in (lo, hi, height)
__device__ __forceinline__ void finish_node(const RefitView &v, int i, int c0, int c1, float4 l0,
                                            float4 h0, float ht0, float4 l1, float4 h1, float ht1,
                                            uint32_t pass, float4 &lo, float4 &hi, float &height) {
  BvhNode nd;
  nd.lo0[0] = l0.x; nd.lo0[1] = l0.y; nd.lo0[2] = l0.z; nd.child0 = c0;
  nd.hi0[0] = h0.x; nd.hi0[1] = h0.y; nd.hi0[2] = h0.z; nd.count0 = 0;
  nd.lo1[0] = l1.x; nd.lo1[1] = l1.y; nd.lo1[2] = l1.z; nd.child1 = c1;
  nd.hi1[0] = h1.x; nd.hi1[1] = h1.y; nd.hi1[2] = h1.z; nd.count1 = 0;
  v.nodes[i] = nd;
  height = fmaxf(ht0, ht1) + 1.f;
  lo = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), height);
  hi = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.f);
  v.box_lo[i] = lo;
  v.box_hi[i] = hi;
  v.stamp[i] = pass;
}

#ifndef RT_REFIT_MID_SHIFT
#define RT_REFIT_MID_SHIFT 16   // the second refit launch works on tiles of 2^16 sorted positions
#endif
constexpr int kRefitTile = 2048;                     // sorted positions (and node indices) per block of k_refit_local
constexpr int kRefitThreads = 1024;
constexpr int kRefitE = kRefitTile / kRefitThreads;  // nodes per thread
constexpr int kTopCap = 3072;                        // listed nodes a block of k_refit_upper has LDS slots for
constexpr int kTopE = kTopCap / 1024;
constexpr uint32_t kTopPassBase = 0x01000000u;       // the last launch's rounds on global memory are numbered from here (above the stamp of a finished node)
constexpr uint32_t kListedBit = 0x80000000u;         // stamp of a listed node until it is finished: this bit | its slot

// ---- bounds, bottom-up, without rounds -------------------------------------------------------
// Inside one block the classic scheme works on LDS alone: a walk starts at a leaf and climbs; at every
// node it publishes what it carries (the box, height and reference of the child it comes from) in the
// walk's own LDS slot and then swaps a token naming that slot into the node's word.  The walk that finds
// the word empty stops there; the one that finds a token is the second to arrive, reads the other child
// from the slot the token names, writes the node's record and climbs on with the union.  Nobody waits
// for anybody, and the only memory the hand-off goes through is the block's own LDS (DS operations of a
// wave execute in order, so a slot is complete before its token can be seen).  Blocks never hand
// anything to each other inside a launch: what crosses a block's boundary is left to the next launch.
struct Carried {
  float lo[3], hi[3];
  float height;   // as the parent counts it: 0 for a leaf and for a subtree referenced as one leaf
  int ref;        // what the parent's record holds: node index, leaf reference or kNullLeaf
};
constexpr uint32_t kTokenSide = 0x80000000u;

template <int CAP>
__device__ __forceinline__ void publish_slot(float (&pub)[8][CAP], uint32_t slot, const Carried &c) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    pub[q][slot] = c.lo[q];
    pub[3 + q][slot] = c.hi[q];
  }
  pub[6][slot] = c.height;
  pub[7][slot] = __int_as_float(c.ref);
}
template <int CAP>
__device__ __forceinline__ Carried read_slot(const float (&pub)[8][CAP], uint32_t slot) {
  Carried c;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    c.lo[q] = pub[q][slot];
    c.hi[q] = pub[3 + q][slot];
  }
  c.height = pub[6][slot];
  c.ref = __float_as_int(pub[7][slot]);
  return c;
}
// record of node i from its two children (a = child 0); returns the node's box and true height
__device__ __forceinline__ void write_record(const RefitView &v, int i, const Carried &a, const Carried &b,
                                             float box[7]) {
  float4 *out = (float4 *)(v.nodes + i);
  out[0] = make_float4(a.lo[0], a.lo[1], a.lo[2], __int_as_float(a.ref));
  out[1] = make_float4(a.hi[0], a.hi[1], a.hi[2], __int_as_float(0));
  out[2] = make_float4(b.lo[0], b.lo[1], b.lo[2], __int_as_float(b.ref));
  out[3] = make_float4(b.hi[0], b.hi[1], b.hi[2], __int_as_float(0));
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    box[k] = fminf(a.lo[k], b.lo[k]);
    box[3 + k] = fmaxf(a.hi[k], b.hi[k]);
  }
  box[6] = fmaxf(a.height, b.height) + 1.f;
}

// One walk: `c` arrives at the node of local index `cur` from side `side`.  node_of(cur) is the node's
// global index, ref_self[cur] the reference its parent holds for it (itself, or a leaf reference when
// it is folded), parent[cur] its parent's local index | side << 15, or 0xffff when the parent is not
// this block's.  Every node a walk finishes gets its record; those without a parent here (all of them
// with all_boxes) also get their box and a stamp, which the next launch reads.
// The walk takes at most max_steps nodes; returns true when it is still going then (cur, side and c say
// where it stands: arriving, nothing published yet).  A wave keeps looping while any of its lanes
// climbs, and the walks thin out by half per level, so the callers cut the climbs short, gather the
// survivors into the block's first waves and let those finish: a third of the instructions.
template <int CAP, typename NodeOf>
__device__ __forceinline__ bool climb(const RefitView &v, float (&pub)[8][CAP], uint32_t *token,
                                      const unsigned short *parent, const int *ref_self, NodeOf node_of,
                                      uint32_t my_slot, uint32_t &cur, uint32_t &side, Carried &c, bool all_boxes,
                                      uint32_t max_steps, uint32_t *root_out = nullptr) {
  for (uint32_t step = 0; step < max_steps; ++step) {
    // (The LDS operations of a wave execute in the order they are issued, so the slot is written before
    // the swap and read after it; the compiler must keep that order too.  A fence proper would also wait
    // for the wave's global stores, a memory round trip per level of the climb.)
    publish_slot(pub, my_slot, c);
    const uint32_t up = parent[cur];     // (read here, so that they travel with the swap instead of after it)
    const int self = ref_self[cur];
    asm volatile("" ::: "memory");
    const uint32_t mine = (my_slot + 1u) | (side ? kTokenSide : 0u);
    const uint32_t old = atomicCAS(&token[cur], 0u, mine);
    asm volatile("" ::: "memory");
    if (old == 0u) return false;   // first to arrive: the other child's walk finishes this node
    const Carried other = read_slot(pub, (old & ~kTokenSide) - 1u);
    const int i = node_of(cur);
    float box[7];
    if (side) write_record(v, i, other, c, box);
    else write_record(v, i, c, other, box);
    if (all_boxes || up == 0xffffu) {
      v.box_lo[i] = make_float4(box[0], box[1], box[2], box[6]);
      v.box_hi[i] = make_float4(box[3], box[4], box[5], 0.f);
      v.stamp[i] = 1u;
    }
    if (i == 0 && root_out) {
      root_out[1] = (uint32_t)box[6];
      root_out[0] = 1u;
    }
    if (up == 0xffffu) return false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      c.lo[k] = box[k];
      c.hi[k] = box[3 + k];
    }
    c.ref = self;
    c.height = c.ref < 0 ? 0.f : box[6];
    cur = up & 0x7fffu;
    side = up >> 15;
  }
  return true;
}
#ifndef RT_REFIT_FIRST_STEPS
#define RT_REFIT_FIRST_STEPS 5   // nodes a leaf's walk takes before the survivors are gathered
#endif

// meta: [0] number of listed nodes, [1] height of the tree, [2] 1 once the root is finished, [3] rounds of the slow path
// A listed node: its index and its run of sorted positions, so that the launches that pick nodes off the
// list by their runs read the list and nothing else.
__global__ __launch_bounds__(kRefitThreads) void k_refit_local(RefitView v, uint4 *__restrict__ top_list,
                                                               uint32_t *meta) {
  __shared__ float pub[8][kRefitTile];                // one slot per leaf = per walk
  __shared__ uint32_t token[kRefitTile];              // per node: 0, or the slot + 1 (| side) of the child that arrived first
  __shared__ unsigned short parent[kRefitTile];       // per node: parent's local index | side << 15, 0xffff: not in this tile
  __shared__ unsigned short leaf_parent[kRefitTile];  // the same per leaf
  __shared__ int ref_self[kRefitTile];                // per node: the reference its parent's record holds
  __shared__ uint32_t queue[kRefitTile];              // walks that go on after their first steps: slot | node << 11 | side << 22
  __shared__ uint32_t n_listed, list_base, n_queued;
  static_assert(kRefitTile == 2048, "queue entries pack two 11-bit indices");
  const int base = blockIdx.x * kRefitTile;
  const int n_leaves = v.n_internal + 1;
  if (threadIdx.x == 0) n_listed = n_queued = 0;
  for (int p = threadIdx.x; p < kRefitTile; p += kRefitThreads) {
    token[p] = 0u;
    parent[p] = 0xffffu;
    leaf_parent[p] = 0xffffu;
  }
  __syncthreads();
  // this thread's nodes: inside the tile (their whole subtree is) or across its boundary (listed)
  uint32_t outside = 0;
  int run_first[kRefitE], run_count[kRefitE];
#pragma unroll
  for (int k = 0; k < kRefitE; ++k) {
    const int local = k * kRefitThreads + (int)threadIdx.x;
    const int i = base + local;
    run_first[k] = run_count[k] = 0;
    if (i >= v.n_internal) continue;
    const int first = v.range_first[i], count = v.range_count[i];
    run_first[k] = first;
    run_count[k] = count;
    if (!(first >= base && first + count <= base + kRefitTile)) {
      outside |= 1u << k;
      continue;
    }
    // a small subtree is referenced as one leaf (the root stays a node)
    ref_self[local] = (i > 0 && count <= kLeafMaxPrims) ? make_leaf_ref((uint32_t)first, (uint32_t)count) : i;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int c = side ? v.right[i] : v.left[i];
      const unsigned short link = (unsigned short)(local | (side << 15));
      if (c >= 0) parent[c - base] = link;
      else leaf_parent[~c - base] = link;
    }
  }
  __syncthreads();
  auto node_of = [&](uint32_t local) { return base + (int)local; };
#pragma unroll
  for (int k = 0; k < kRefitE; ++k) {
    const int p = k * kRefitThreads + (int)threadIdx.x;
    if (base + p >= n_leaves) continue;
    const uint32_t up = leaf_parent[p];
    if (up == 0xffffu) continue;   // the leaf's parent straddles the tile: a later launch reads the leaf itself
    const float4 lo = v.leaf_lo[base + p], hi = v.leaf_hi[base + p];
    Carried c;
    c.lo[0] = lo.x; c.lo[1] = lo.y; c.lo[2] = lo.z;
    c.hi[0] = hi.x; c.hi[1] = hi.y; c.hi[2] = hi.z;
    c.height = 0.f;
    c.ref = lo.w != 0.f ? kNullLeaf : make_leaf_ref((uint32_t)(base + p), 1u);
    uint32_t cur = up & 0x7fffu, side = up >> 15;
    if (climb(v, pub, token, parent, ref_self, node_of, (uint32_t)p, cur, side, c, false, RT_REFIT_FIRST_STEPS)) {
      publish_slot(pub, (uint32_t)p, c);   // (nobody reads the slot before a token names it, and none does yet)
      queue[atomicAdd(&n_queued, 1u)] = (uint32_t)p | (cur << 11) | (side << 22);
    }
  }
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < n_queued; e += kRefitThreads) {
    const uint32_t entry = queue[e], p = entry & 2047u;
    uint32_t cur = (entry >> 11) & 2047u, side = entry >> 22;
    Carried c = read_slot(pub, p);
    climb(v, pub, token, parent, ref_self, node_of, p, cur, side, c, false, 0xffffffffu);
  }
  // the nodes across the tile's boundary go onto the list
  uint32_t slot = 0;
  const uint32_t mine = (uint32_t)__builtin_popcount(outside);
  if (mine) slot = atomicAdd(&n_listed, mine);
  __syncthreads();
  if (threadIdx.x == 0 && n_listed) list_base = atomicAdd(meta, n_listed);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kRefitE; ++k)
    if (outside & (1u << k)) {
      const uint32_t i = (uint32_t)(base + k * kRefitThreads + (int)threadIdx.x);
      top_list[list_base + slot] = make_uint4(i, (uint32_t)run_first[k], (uint32_t)run_count[k], 0u);
      v.stamp[i] = kListedBit | (list_base + slot);
      ++slot;
    }
}

// k_refit_upper's last launch with more nodes than slots: one attempt at node i in round `pass` on global memory
__device__ __forceinline__ bool try_finish(const RefitView &v, int i, uint32_t pass) {
  int c0 = v.left[i], c1 = v.right[i];
  if (!child_ready(v, c0, pass) || !child_ready(v, c1, pass)) return false;
  float4 l0, h0, l1, h1, lo, hi;
  float ht0, ht1, height;
  child_box(v, c0, l0, h0, ht0);
  child_box(v, c1, l1, h1, ht1);
  finish_node(v, i, c0, c1, l0, h0, ht0, l1, h1, ht1, pass, lo, hi, height);
  return true;
}

// The listed nodes.  A launch with tile_shift = t finishes every listed node whose run of sorted
// positions lies inside one tile of 2^t positions (block b takes tile b); the last launch, one block,
// takes whatever is left.  A block picks its nodes off the list, gives them LDS slots (the stamp of a
// picked node becomes kListedBit | its slot, which is how a parent finds a child that the same block
// finishes), fetches the boxes of children that an earlier launch finished, and lets walks climb from
// the nodes that have both children (see climb).  A block that finds more nodes than it has slots
// leaves them to the last launch, which falls back to rounds on global memory when it, too, has too many.
__global__ __launch_bounds__(1024) void k_refit_upper(RefitView v, const uint4 *__restrict__ top_list,
                                                      uint32_t *meta, uint32_t tile_shift, uint32_t last,
                                                      uint32_t earlier_shift, uint32_t cap) {
  __shared__ float pub[8][kTopCap];            // one slot per picked node (see climb)
  __shared__ uint32_t token[kTopCap];
  __shared__ unsigned short parent[kTopCap];   // parent's slot | side << 15, 0xffff: not picked by this block
  __shared__ int ref_self[kTopCap];
  __shared__ int slot_node[kTopCap];
  __shared__ unsigned short queue[kTopCap];    // slots of the nodes that start walks
  __shared__ uint32_t wave_base[16], n_picked, root_seen[2], n_queued;
  const uint32_t n_listed = meta[0];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) root_seen[0] = root_seen[1] = n_queued = 0u;
  // ---- pick: unfinished listed nodes whose run lies in this block's tile, counted, then numbered.
  // A thread looks at kPickE list entries at a time with the loads of one kind issued together (entry,
  // stamp, run): three memory round trips per batch instead of three per entry.
  constexpr int kPickE = 8;
  // The last launch takes what the launch before it (tiles of 2^earlier_shift positions; 32: there was
  // none) did not finish: the nodes whose run crosses a boundary of those tiles -- or, when a block of
  // that launch had more nodes than slots and left them all (meta[4]), whatever still carries a listed stamp.
  const bool by_stamp = last && earlier_shift < 32u && meta[4] != 0u;
  auto pick_batch = [&](uint32_t j0, int out[kPickE]) -> uint32_t {
    uint32_t valid = 0;
    uint4 e[kPickE];
#pragma unroll
    for (int u = 0; u < kPickE; ++u) {
      const uint32_t j = j0 + (uint32_t)u * 1024u + threadIdx.x;
      e[u] = j < n_listed ? top_list[j] : make_uint4(0u, 0u, 0u, 1u);
    }
#pragma unroll
    for (int u = 0; u < kPickE; ++u) {
      out[u] = (int)e[u].x;
      if (e[u].w) continue;   // beyond the list
      const uint32_t first = e[u].y, end = e[u].y + e[u].z - 1u;
      bool take;
      if (!last) take = (first >> tile_shift) == blockIdx.x && (end >> tile_shift) == blockIdx.x;
      else if (by_stamp) take = true;
      else take = earlier_shift >= 32u || (first >> earlier_shift) != (end >> earlier_shift);
      if (take) valid |= 1u << u;
    }
    if (by_stamp) {
      uint32_t st[kPickE];
#pragma unroll
      for (int u = 0; u < kPickE; ++u) st[u] = (valid >> u) & 1u ? v.stamp[out[u]] : 0u;
#pragma unroll
      for (int u = 0; u < kPickE; ++u)
        if (!(st[u] & kListedBit)) valid &= ~(1u << u);   // finished by the earlier launch
    }
    return valid;
  };
  int first_batch[kPickE];
  const uint32_t first_valid = pick_batch(0u, first_batch);   // kept: most builds have no second batch
  uint32_t count = (uint32_t)__builtin_popcount(first_valid);
  for (uint32_t j0 = kPickE * 1024u; j0 < n_listed; j0 += kPickE * 1024u) {
    int tmp[kPickE];
    count += (uint32_t)__builtin_popcount(pick_batch(j0, tmp));
  }
  // exclusive prefix over the block's threads: within the wave by shuffles, across waves through LDS
  uint32_t inc = count;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(inc, off, 64);
    if (lane >= (uint32_t)off) inc += up;
  }
  if (lane == 63u) wave_base[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int w = 0; w < 16; ++w) {
      const uint32_t c = wave_base[w];
      wave_base[w] = run;
      run += c;
    }
    n_picked = run;
  }
  __syncthreads();
  const uint32_t picked = n_picked;
  uint32_t rounds = 0;
  if (picked > cap) {   // (cap: kTopCap, or less when a test wants to see the paths below)
    if (!last) {   // too many for this block's LDS: the last launch takes them
      if (threadIdx.x == 0) atomicOr(&meta[4], 1u);
      return;
    }
    // the last launch, with more nodes than LDS holds (tens of millions of primitives): rounds on global memory
    for (uint32_t pass = kTopPassBase;; ++pass) {
      bool progress = false;
      for (uint32_t j = threadIdx.x; j < n_listed; j += 1024u) {
        const int i = (int)top_list[j].x;
        if (!(v.stamp[i] & kListedBit)) continue;
        if (try_finish(v, i, pass)) progress = true;
      }
      ++rounds;
      if (!__syncthreads_or(progress)) break;
    }
  } else if (picked) {
    uint32_t slot = wave_base[wave] + inc - count;
#pragma unroll
    for (int u = 0; u < kPickE; ++u)
      if ((first_valid >> u) & 1u) slot_node[slot++] = first_batch[u];
    for (uint32_t j0 = kPickE * 1024u; j0 < n_listed; j0 += kPickE * 1024u) {
      int tmp[kPickE];
      const uint32_t valid = pick_batch(j0, tmp);
#pragma unroll
      for (int u = 0; u < kPickE; ++u)
        if ((valid >> u) & 1u) slot_node[slot++] = tmp[u];
    }
    for (uint32_t q = threadIdx.x; q < (uint32_t)kTopCap; q += 1024u) {
      token[q] = 0u;
      parent[q] = 0xffffu;
    }
    __syncthreads();
    // (stamps are rewritten only now, after every thread has done its second look at them)
    for (uint32_t q = threadIdx.x; q < picked; q += 1024u) v.stamp[slot_node[q]] = kListedBit | q;
    __syncthreads();
    // ---- what each picked node already has: children finished by earlier launches, and leaves
    Carried ready[kTopE][2];
    uint32_t n_ready[kTopE];
#pragma unroll
    for (int k = 0; k < kTopE; ++k) {
      const uint32_t q = (uint32_t)k * 1024u + threadIdx.x;
      n_ready[k] = 0xffu;   // no node
      if (q >= picked) continue;
      const int i = slot_node[q];
      const int under = v.range_count[i];
      ref_self[q] = (i > 0 && under <= kLeafMaxPrims) ? make_leaf_ref((uint32_t)v.range_first[i], (uint32_t)under) : i;
      uint32_t have = 0;
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        Carried &c = ready[k][side];
        const int cc = side ? v.right[i] : v.left[i];
        float4 lo, hi;
        if (cc < 0) {
          lo = v.leaf_lo[~cc];
          hi = v.leaf_hi[~cc];
          c.ref = lo.w != 0.f ? kNullLeaf : make_leaf_ref((uint32_t)~cc, 1u);
          c.height = 0.f;
          have |= 1u << side;
        } else {
          const uint32_t st = v.stamp[cc];
          if (st & kListedBit) {   // a child of a picked node that is still listed was picked too: its walk comes by
            parent[st & ~kListedBit] = (unsigned short)(q | ((uint32_t)side << 15));
            lo = hi = make_float4(0.f, 0.f, 0.f, 0.f);
            c.ref = 0;
            c.height = 0.f;
          } else {
            const int n_under = v.range_count[cc];
            const bool folded = n_under <= kLeafMaxPrims;
            c.ref = folded ? make_leaf_ref((uint32_t)v.range_first[cc], (uint32_t)n_under) : cc;
            lo = v.box_lo[cc];
            hi = v.box_hi[cc];
            c.height = folded ? 0.f : lo.w;
            have |= 1u << side;
          }
        }
        c.lo[0] = lo.x; c.lo[1] = lo.y; c.lo[2] = lo.z;
        c.hi[0] = hi.x; c.hi[1] = hi.y; c.hi[2] = hi.z;
      }
      n_ready[k] = have;
      if (have == 1u || have == 2u) {   // one child here already: it counts as the first arrival
        const uint32_t side = have >> 1;
        if (side) publish_slot(pub, q, ready[k][1]);
        else publish_slot(pub, q, ready[k][0]);
        token[q] = (q + 1u) | (side ? kTokenSide : 0u);
      }
    }
    __syncthreads();
    auto node_of = [&](uint32_t q) { return slot_node[q]; };
    // ---- the nodes with both children here start the walks: there are few of them (one per chain of
    // nodes along a tile boundary), scattered over the block's threads, so they are gathered first
#pragma unroll
    for (int k = 0; k < kTopE; ++k) {
      if (n_ready[k] != 3u) continue;
      const uint32_t q = (uint32_t)k * 1024u + threadIdx.x;
      const int i = slot_node[q];
      float box[7];
      write_record(v, i, ready[k][0], ready[k][1], box);
      v.box_lo[i] = make_float4(box[0], box[1], box[2], box[6]);
      v.box_hi[i] = make_float4(box[3], box[4], box[5], 0.f);
      v.stamp[i] = 1u;
      if (i == 0) {
        root_seen[1] = (uint32_t)box[6];
        root_seen[0] = 1u;
      }
      if (parent[q] == 0xffffu) continue;
      Carried c;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        c.lo[e] = box[e];
        c.hi[e] = box[3 + e];
      }
      c.ref = ref_self[q];
      c.height = c.ref < 0 ? 0.f : box[6];
      publish_slot(pub, q, c);
      queue[atomicAdd(&n_queued, 1u)] = (unsigned short)q;
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < n_queued; e += 1024u) {
      const uint32_t q = queue[e], up = parent[q];
      uint32_t cur = up & 0x7fffu, side = up >> 15;
      Carried c = read_slot(pub, q);
      climb(v, pub, token, parent, ref_self, node_of, q, cur, side, c, true, 0xffffffffu, root_seen);
    }
    __syncthreads();
  }
  if (last && threadIdx.x == 0) {
    // the root: finished by this block (LDS says so), or by an earlier launch (its stamp does)
    bool done = root_seen[0] != 0u;
    uint32_t height = root_seen[1];
    if (!done && v.n_internal > 0) {
      const uint32_t st = v.stamp[0];
      done = st != 0u && !(st & kListedBit);
      if (done) height = (uint32_t)v.box_lo[0].w;
    }
    meta[1] = done ? height : 0u;
    meta[2] = done ? 1u : 0u;
    meta[3] = rounds;
  }
}

// geometry and leaf boxes into sorted order: the one gather of the build (the refit kernels then read
// their leaves' boxes coalesced, from blocks that have few waves to hide a gather behind)
__global__ __launch_bounds__(kThreads) void k_permute(const float4 *__restrict__ geom,
                                                      uint32_t n_spheres,
                                                      const uint32_t *__restrict__ sorted_prim,
                                                      float4 *__restrict__ sorted_geom,
                                                      const float4 *__restrict__ leaf_lo,
                                                      const float4 *__restrict__ leaf_hi,
                                                      float4 *__restrict__ sorted_lo,
                                                      float4 *__restrict__ sorted_hi, uint32_t n) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint32_t prim = sorted_prim[i];
  sorted_geom[i] = prim < n_spheres ? geom[prim] : make_float4(0.f, 0.f, 0.f, 0.f);
  sorted_lo[i] = leaf_lo[prim];
  sorted_hi[i] = leaf_hi[prim];
}

}  // namespace

// Stable LSD radix sort of n (key, value) pairs by the low 8 * passes bits of the key, with the build's
// own kernels: keys[0] / vals[0] hold the input, the result is in keys[r] / vals[r] with r = passes & 1.
hipError_t radix_sort_pairs(uint32_t *const keys[2], uint32_t *const vals[2], uint32_t n, int passes,
                            uint32_t *hist, uint32_t *scan_sums, hipStream_t stream) {
  const uint32_t nblk = (n + kRsTile - 1) / kRsTile;
  int cur = 0;
  for (int pass = 0; pass < passes; ++pass) {
    const uint32_t shift = 8u * (uint32_t)pass;
    hipLaunchKernelGGL(k_rs_hist, dim3(nblk), dim3(kThreads), 0, stream, keys[cur], n, shift, hist, nblk);
    hipLaunchKernelGGL(k_scan_rows, dim3(256), dim3(kThreads), 0, stream, hist, nblk, scan_sums);
    hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(kThreads), 0, stream, keys[cur], vals[cur], keys[cur ^ 1],
                       vals[cur ^ 1], n, shift, hist, scan_sums, nblk);
    cur ^= 1;
  }
  return hipGetLastError();
}
void radix_sort_scratch(uint32_t n, size_t *hist_bytes, size_t *sums_bytes) {
  const uint32_t nblk = (n + kRsTile - 1) / kRsTile;
  *hist_bytes = (size_t)256 * nblk * sizeof(uint32_t);
  *sums_bytes = 256 * sizeof(uint32_t);   // the digit totals of one pass
}

struct Lbvh::Work {
  size_t capacity = 0;
  char *slab = nullptr;
  float4 *cr = nullptr, *leaf_lo = nullptr, *leaf_hi = nullptr, *box_lo = nullptr, *box_hi = nullptr;
  float4 *sorted_lo = nullptr, *sorted_hi = nullptr;  // the leaf boxes by sorted position
  uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr};
  uint32_t *hist = nullptr, *scan_sums = nullptr, *stamp = nullptr, *cbounds = nullptr;
  int *left = nullptr, *right = nullptr;
  uint4 *top_list = nullptr;     // nodes that straddle a tile boundary of k_refit_local, with their runs
  int *range_first = nullptr, *range_count = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  uint32_t *h_meta = nullptr;    // pinned: the one read-back of a build lands here without a staging copy
  // what the device holds of the host-made inputs (uploads of unchanged content are skipped: a
  // blocking copy per rebuild is a tenth of a small rebuild)
  std::vector<float> up_cr, up_large_geom;
  std::vector<uint32_t> up_large_prim;
};

void Lbvh::release() {
  if (work) {
    if (work->slab) hipFree(work->slab);
    if (work->ev0) hipEventDestroy(work->ev0);
    if (work->ev1) hipEventDestroy(work->ev1);
    if (work->h_meta) hipHostFree(work->h_meta);
    delete work;
    work = nullptr;
  }
  if (nodes) hipFree(nodes);
  if (sorted_prim) hipFree(sorted_prim);
  if (sorted_geom) hipFree(sorted_geom);
  if (large_geom) hipFree(large_geom);
  if (large_prim) hipFree(large_prim);
  nodes = nullptr;
  sorted_prim = nullptr;
  sorted_geom = nullptr;
  large_geom = nullptr;
  large_prim = nullptr;
  n_large = 0;
  n_nodes = 0;
  capacity = 0;
}

#define BVH_HIP(call)                                                       \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) {                                                 \
      if (err) *err = std::string(#call) + ": " + hipGetErrorString(e_);    \
      return -2;                                                            \
    }                                                                       \
  } while (0)

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }
static bool same_bits(const std::vector<float> &a, const std::vector<float> &b) {
  return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0);
}

int Lbvh::build(const float4 *d_geom, const float *h_spheres64, uint32_t n_spheres,
                const float4 *d_tri, const float mesh_lo[3], const float mesh_hi[3], uint32_t n_tris,
                const float cam[3], hipStream_t stream, std::string *err) {
  const uint32_t n = n_spheres + n_tris;
  if (n == 0) {
    if (err) *err = "empty scene";
    return -1;
  }
  // ---- storage -----------------------------------------------------------------------
  if (n > capacity) {
    if (nodes) hipFree(nodes);
    if (sorted_prim) hipFree(sorted_prim);
    if (sorted_geom) hipFree(sorted_geom);
    nodes = nullptr; sorted_prim = nullptr; sorted_geom = nullptr;
    capacity = 0;  // stays 0 if an allocation below fails
    BVH_HIP(hipMalloc((void **)&nodes, (size_t)std::max(n - 1, 1u) * sizeof(BvhNode)));
    BVH_HIP(hipMalloc((void **)&sorted_prim, (size_t)n * sizeof(uint32_t)));
    BVH_HIP(hipMalloc((void **)&sorted_geom, (size_t)n * sizeof(float4)));
    capacity = n;
  }
  if (!work) {
    work = new Work();
    BVH_HIP(hipEventCreate(&work->ev0));
    BVH_HIP(hipEventCreate(&work->ev1));
    BVH_HIP(hipHostMalloc((void **)&work->h_meta, 8 * sizeof(uint32_t), hipHostMallocDefault));
  }
  const uint32_t nblk = (n + kRsTile - 1) / kRsTile;
  if (n > work->capacity) {
    if (work->slab) hipFree(work->slab);
    work->slab = nullptr;
    work->capacity = 0;
    work->up_cr.clear();  // (lived in the slab)
    const size_t f4 = align256((size_t)n * sizeof(float4)), u4 = align256((size_t)n * sizeof(uint32_t));
    const size_t hist = align256((size_t)256 * nblk * sizeof(uint32_t));
    const size_t sums = align256(256 * sizeof(uint32_t));   // the digit totals of one radix pass
    const size_t total = 8 * f4 + 9 * u4 + hist + sums + 512;
    BVH_HIP(hipMalloc((void **)&work->slab, total));
    char *p = work->slab;
    auto take = [&](size_t bytes) { char *q = p; p += bytes; return q; };
    work->cr = (float4 *)take(f4);
    work->leaf_lo = (float4 *)take(f4);
    work->leaf_hi = (float4 *)take(f4);
    work->sorted_lo = (float4 *)take(f4);
    work->sorted_hi = (float4 *)take(f4);
    work->box_lo = (float4 *)take(f4);
    work->box_hi = (float4 *)take(f4);
    work->keys[0] = (uint32_t *)take(u4);
    work->keys[1] = (uint32_t *)take(u4);
    work->vals[0] = (uint32_t *)take(u4);
    work->vals[1] = (uint32_t *)take(u4);
    work->stamp = (uint32_t *)take(u4);
    work->left = (int *)take(u4);
    work->right = (int *)take(u4);
    work->top_list = (uint4 *)take(f4);
    work->range_first = (int *)take(u4);
    work->range_count = (int *)take(u4);
    work->hist = (uint32_t *)take(hist);
    work->scan_sums = (uint32_t *)take(sums);
    work->cbounds = (uint32_t *)take(512);  // six bounds words, and the refit's four words of meta in a 256-byte block of their own
    work->capacity = n;
  }
  Work &w = *work;

  // ---- host prologue: (centre, radius) records and the region ray origins stay in ----------
  std::vector<float> cr((size_t)std::max(n_spheres, 1u) * 4);
  double rsum = 0.0;
  for (uint32_t i = 0; i < n_spheres; ++i) {
    const float *s = h_spheres64 + (size_t)i * 16;
    cr[4 * (size_t)i + 0] = s[0];
    cr[4 * (size_t)i + 1] = s[1];
    cr[4 * (size_t)i + 2] = s[2];
    cr[4 * (size_t)i + 3] = s[3];
    rsum += std::fabs((double)s[3]);
  }
  const float r_ordinary = n_spheres ? (float)(8.0 * rsum / n_spheres) : 0.f;
  BoxParams bp;
  {
    // oversized spheres: radius above a tenth of the diagonal of the bounds of all sphere
    // centres and mesh vertices
    float clo[3] = {3e38f, 3e38f, 3e38f}, chi[3] = {-3e38f, -3e38f, -3e38f};
    for (uint32_t i = 0; i < n_spheres; ++i)
      for (int k = 0; k < 3; ++k) {
        clo[k] = std::min(clo[k], cr[4 * (size_t)i + k]);
        chi[k] = std::max(chi[k], cr[4 * (size_t)i + k]);
      }
    if (n_tris)
      for (int k = 0; k < 3; ++k) {
        clo[k] = std::min(clo[k], mesh_lo[k]);
        chi[k] = std::max(chi[k], mesh_hi[k]);
      }
    for (int k = 0; k < 3; ++k) {
      scene_lo[k] = clo[k];
      scene_hi[k] = chi[k];
    }
    double diag2 = 0.0;
    for (int k = 0; k < 3; ++k) diag2 += ((double)chi[k] - clo[k]) * ((double)chi[k] - clo[k]);
    bp.large_radius = n >= 3 ? (float)(0.1 * std::sqrt(diag2)) : 3.0e38f;
    std::vector<uint32_t> big;
    for (uint32_t i = 0; i < n_spheres; ++i)
      if (std::fabs(cr[4 * (size_t)i + 3]) > bp.large_radius) big.push_back(i);
    if (big.size() > (size_t)kMaxLargeSpheres || big.size() == n) {
      bp.large_radius = 3.0e38f;  // too many to list: everything goes into the tree
      big.clear();
    }
    n_large = (uint32_t)big.size();
    std::vector<float> lg((size_t)kMaxLargeSpheres * 4, 0.f);
    std::vector<uint32_t> li((size_t)kMaxLargeSpheres, 0u);
    for (size_t j = 0; j < big.size(); ++j) {
      const float *s = h_spheres64 + (size_t)big[j] * 16;
      lg[4 * j + 0] = s[0]; lg[4 * j + 1] = s[1]; lg[4 * j + 2] = s[2];
      lg[4 * j + 3] = s[3] * s[3];
      li[j] = big[j];
    }
    if (!large_geom) {
      BVH_HIP(hipMalloc((void **)&large_geom, kMaxLargeSpheres * sizeof(float4)));
      BVH_HIP(hipMalloc((void **)&large_prim, kMaxLargeSpheres * sizeof(uint32_t)));
    }
    // (uploads are skipped when the device already holds the same BITS: -0 is not +0 here.
    // The copies go through the launch stream, behind any frame that still reads the old list.)
    if (!same_bits(lg, w.up_large_geom) || li != w.up_large_prim) {
      BVH_HIP(hipMemcpyAsync(large_geom, lg.data(), lg.size() * sizeof(float), hipMemcpyHostToDevice, stream));
      BVH_HIP(hipMemcpyAsync(large_prim, li.data(), li.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
      BVH_HIP(hipStreamSynchronize(stream));  // lg, li (pageable host memory) must outlive the copies
      w.up_large_geom = lg;
      w.up_large_prim = li;
    }
  }
  float rmax_small = 0.f;
  for (int k = 0; k < 3; ++k) bp.blo[k] = bp.bhi[k] = cam[k];
  if (n_tris)
    for (int k = 0; k < 3; ++k) {
      bp.blo[k] = std::min(bp.blo[k], mesh_lo[k]);
      bp.bhi[k] = std::max(bp.bhi[k], mesh_hi[k]);
    }
  for (uint32_t i = 0; i < n_spheres; ++i) {
    const float *s = &cr[4 * (size_t)i];
    if (std::fabs(s[3]) > r_ordinary) continue;
    rmax_small = std::max(rmax_small, std::fabs(s[3]));
    for (int k = 0; k < 3; ++k) {
      bp.blo[k] = std::min(bp.blo[k], s[k]);
      bp.bhi[k] = std::max(bp.bhi[k], s[k]);
    }
  }
  for (int k = 0; k < 3; ++k) {
    bp.blo[k] -= rmax_small;
    bp.bhi[k] += rmax_small;
  }

  // ---- device build ----------------------------------------------------------------------
  BVH_HIP(hipEventRecord(w.ev0, stream));
  if (n_spheres && !same_bits(cr, w.up_cr)) {
    BVH_HIP(hipMemcpyAsync(w.cr, cr.data(), (size_t)n_spheres * 4 * sizeof(float),
                           hipMemcpyHostToDevice, stream));
    BVH_HIP(hipStreamSynchronize(stream));  // cr (pageable host memory) must outlive the copy
    w.up_cr = cr;
  }
  // running minima and maxima of the centres, and the refit's words of meta (one launch instead of three fills)
  hipLaunchKernelGGL(k_build_init, dim3(1), dim3(128), 0, stream, w.cbounds);
  const uint32_t blocks_n = (n + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(k_leaf_boxes, dim3(std::min(blocks_n, 256u)), dim3(kThreads), 0, stream, w.cr,
                     n_spheres, d_tri, n, bp, w.leaf_lo, w.leaf_hi, w.cbounds);
  hipLaunchKernelGGL(k_morton, dim3(blocks_n), dim3(kThreads), 0, stream, w.leaf_lo, w.leaf_hi, n,
                     w.cbounds, w.keys[0], w.vals[0]);
  int cur = 0;
  for (uint32_t shift = 0; shift < 32; shift += 8) {
    hipLaunchKernelGGL(k_rs_hist, dim3(nblk), dim3(kThreads), 0, stream, w.keys[cur], n, shift, w.hist,
                       nblk);
    hipLaunchKernelGGL(k_scan_rows, dim3(256), dim3(kThreads), 0, stream, w.hist, nblk, w.scan_sums);
    // the last pass writes the sorted order where the traversal reads it
    hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(kThreads), 0, stream, w.keys[cur], w.vals[cur],
                       w.keys[cur ^ 1], shift == 24u ? sorted_prim : w.vals[cur ^ 1], n, shift, w.hist, w.scan_sums, nblk);
    cur ^= 1;
  }
  // sorted codes in keys[cur], sorted order in sorted_prim
  hipLaunchKernelGGL(k_permute, dim3(blocks_n), dim3(kThreads), 0, stream, d_geom, n_spheres,
                     sorted_prim, sorted_geom, w.leaf_lo, w.leaf_hi, w.sorted_lo, w.sorted_hi, n);
  n_nodes = n - 1;
  height = 0;
  if (n >= 2) {
    const uint32_t blocks_i = (n - 1 + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_hierarchy, dim3(blocks_i), dim3(kThreads), 0, stream, w.keys[cur], (int)n,
                       w.left, w.right, w.range_first, w.range_count);
    // (no stamp is read that this build's first refit launch has not written: the roots of the tiles'
    // subtrees, the listed nodes, and through them the root of the tree)
    uint32_t *meta = w.cbounds + 64;
    RefitView rv;
    rv.left = w.left; rv.right = w.right;
    rv.range_first = w.range_first; rv.range_count = w.range_count;
    rv.sorted_prim = sorted_prim;
    rv.leaf_lo = w.sorted_lo; rv.leaf_hi = w.sorted_hi;
    rv.stamp = w.stamp; rv.box_lo = w.box_lo; rv.box_hi = w.box_hi;
    rv.nodes = nodes;
    rv.n_internal = (int)(n - 1);
    const uint32_t tiles = (n - 1 + kRefitTile - 1) / kRefitTile;
    hipLaunchKernelGGL(k_refit_local, dim3(tiles), dim3(kRefitThreads), 0, stream, rv, w.top_list, meta);
    // the nodes that straddle boundaries of those tiles: first the ones inside tiles of 64 x 2048 positions
    // (when there are several such tiles), then, in one block, the rest
    constexpr uint32_t kMidShift = RT_REFIT_MID_SHIFT;
    const uint32_t mid_tiles = (n + (1u << kMidShift) - 1u) >> kMidShift;
    // test hook: fewer LDS slots than the kernel has, so that a small scene reaches the paths of a huge one
    uint32_t top_cap = (uint32_t)kTopCap;
    if (const char *e = std::getenv("RT_REFIT_TOP_CAP")) top_cap = std::min(top_cap, (uint32_t)std::strtoul(e, nullptr, 10));
    if (mid_tiles > 1u)
      hipLaunchKernelGGL(k_refit_upper, dim3(mid_tiles), dim3(1024), 0, stream, rv, w.top_list, meta, kMidShift, 0u, 32u,
                         top_cap);
    hipLaunchKernelGGL(k_refit_upper, dim3(1), dim3(1024), 0, stream, rv, w.top_list, meta, 32u, 1u,
                       mid_tiles > 1u ? kMidShift : 32u, top_cap);
    // ONE look at the device per build: the height sizes the traversal stack of the launches that follow
    uint32_t *h_meta = w.h_meta;
    h_meta[0] = h_meta[1] = h_meta[2] = h_meta[3] = 0u;
    BVH_HIP(hipMemcpyAsync(h_meta, meta, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    BVH_HIP(hipStreamSynchronize(stream));
    if (h_meta[2] != 1u) {
      if (err) *err = "bounds propagation did not reach the root";
      return -2;
    }
    height = h_meta[1];
    refit_listed = h_meta[0];
    refit_rounds = h_meta[3];
  }
  BVH_HIP(hipEventRecord(w.ev1, stream));
  timing_pending = true;  // read by build_ms(), so that no build waits for its own end
  BVH_HIP(hipGetLastError());
  return 0;
}

float Lbvh::build_ms() {
  if (timing_pending && work) {
    float ms = 0.f;
    if (hipEventSynchronize(work->ev1) == hipSuccess &&
        hipEventElapsedTime(&ms, work->ev0, work->ev1) == hipSuccess)
      last_build_ms = ms;
    timing_pending = false;
  }
  return last_build_ms;
}

int Lbvh::download(std::vector<BvhNode> *out_nodes, std::vector<uint32_t> *out_prims) const {
  out_nodes->resize(n_nodes);
  out_prims->resize(n_nodes + 1);
  if (n_nodes &&
      hipMemcpy(out_nodes->data(), nodes, (size_t)n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost) !=
          hipSuccess)
    return -2;
  if (hipMemcpy(out_prims->data(), sorted_prim, (size_t)(n_nodes + 1) * sizeof(uint32_t),
                hipMemcpyDeviceToHost) != hipSuccess)
    return -2;
  return 0;
}

}  // namespace rt
