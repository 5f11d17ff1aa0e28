// Linear BVH over spheres and mesh triangles, built on the GPU.
//
//   1. leaf boxes (inflated, see below) + bounds of the centres       k_leaf_boxes
//   2. 30-bit Morton code of each centre (bits handed to the axes by   k_morton
//      the shape of the scene)
//   3. stable LSD radix sort of (code, index), 4 passes of 8 bits     k_rs_hist / k_rs_scan / k_rs_scatter
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
// tile of kRefitTile consecutive positions and lets it finish, in rounds separated by workgroup
// barriers, every node whose run lies inside the tile (its children's runs then do too, so all it
// reads was written by its own block, and a barrier orders that); the nodes that straddle a tile
// boundary -- about a dozen per boundary -- go onto a list.  k_refit_upper finishes the listed
// nodes the same way, first (when there are several) inside tiles of 2^17 positions, one block
// per tile, then, in ONE block, whatever is left: their children are either listed too and lie in
// the same tile (same block) or were finished by an earlier launch.  A node is finished in round r
// only from children finished in a round before r, so it never reads a box that is being written;
// the rounds run on LDS and registers, global memory is read before them and written after them.
// No fences, no flags polled across CUs, no dependence on dispatch order or on which XCD a block
// runs, and ONE look at the device from the host, when the tree is complete (it was 16 launches
// of a kernel over all nodes and a read-back per round of them: 224 of the 460 microseconds of a
// million-triangle build; now 140 in three launches, the build 380).
#include "rt_lbvh.h"

#include <algorithm>
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

// Exclusive scan of `total` counters in place.  Three launches: every block sums its tile of
// 2048 counters (coalesced), one block scans the tile sums, every block scans its tile on
// top of its sum.  (One block walking all 256 * blocks counters took 0.2 ms per radix pass
// for a million keys, half of the whole build.)
constexpr uint32_t kScanE = 8;
constexpr uint32_t kScanTile = kThreads * kScanE;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(v, off, 64);
    if (lane >= (uint32_t)off) v += up;
  }
  return v;
}

__global__ __launch_bounds__(kThreads) void k_scan_reduce(const uint32_t *__restrict__ data, uint32_t total,
                                                          uint32_t *__restrict__ tile_sum) {
  __shared__ uint32_t wsum[kThreads / 64];
  const uint32_t first = blockIdx.x * kScanTile + threadIdx.x * kScanE;
  uint32_t sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < kScanE; ++k) sum += first + k < total ? data[first + k] : 0u;
  const uint32_t inc = wave_inclusive_scan(sum);
  if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int wv = 0; wv < kThreads / 64; ++wv) t += wsum[wv];
    tile_sum[blockIdx.x] = t;
  }
}

// exclusive scan of a short array in place, one block of 1024 threads
__global__ __launch_bounds__(1024) void k_scan_top(uint32_t *data, uint32_t total) {
  __shared__ uint32_t part[1024];
  const uint32_t per = (total + 1023u) / 1024u;
  const uint32_t b = threadIdx.x * per, e = min(b + per, total);
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; ++i) sum += data[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024u; off <<= 1) {
    uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - sum;  // exclusive prefix of this thread's chunk
  for (uint32_t i = b; i < e; ++i) {
    const uint32_t c = data[i];
    data[i] = run;
    run += c;
  }
}

__global__ __launch_bounds__(kThreads) void k_scan_apply(uint32_t *__restrict__ data, uint32_t total,
                                                         const uint32_t *__restrict__ tile_sum) {
  __shared__ uint32_t wsum[kThreads / 64];
  const uint32_t first = blockIdx.x * kScanTile + threadIdx.x * kScanE;
  uint32_t c[kScanE], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < kScanE; ++k) {
    c[k] = first + k < total ? data[first + k] : 0u;
    sum += c[k];
  }
  const uint32_t inc = wave_inclusive_scan(sum);
  if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t run = tile_sum[blockIdx.x] + inc - sum;
  for (uint32_t wv = 0; wv < (threadIdx.x >> 6); ++wv) run += wsum[wv];
#pragma unroll
  for (uint32_t k = 0; k < kScanE; ++k) {
    if (first + k < total) data[first + k] = run;
    run += c[k];
  }
}

__global__ __launch_bounds__(kThreads) void k_rs_scatter(const uint32_t *__restrict__ keys_in,
                                                         const uint32_t *__restrict__ vals_in,
                                                         uint32_t *__restrict__ keys_out,
                                                         uint32_t *__restrict__ vals_out, uint32_t n,
                                                         uint32_t shift,
                                                         const uint32_t *__restrict__ hist,
                                                         uint32_t nblk) {
  __shared__ uint32_t base[256];       // next output slot of each digit for this block
  __shared__ uint32_t wcount[4][256];  // per wave, keys of each digit in the current round
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  base[t] = hist[t * nblk + blockIdx.x];
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

constexpr int kRefitTile = 2048;                     // sorted positions (and node indices) per block of k_refit_local
constexpr int kRefitThreads = 1024;
constexpr int kRefitE = kRefitTile / kRefitThreads;  // nodes per thread
constexpr int kTopCap = 3072;                        // listed nodes k_refit_top keeps in LDS (more: its slow path)
constexpr int kTopE = kTopCap / 1024;
constexpr uint32_t kTopPassBase = 0x01000000u;       // slow path: its rounds are numbered from here (above any local round)
constexpr uint32_t kListedBit = 0x80000000u;         // stamp of a listed node until it is finished: this bit | its slot

// A child as its parent sees it while the parent waits: `wait` >= 0 is the LDS slot of a child that the
// same kernel has yet to finish (its box is read from LDS then), -1 a child whose box is already here.
struct PendingChild {
  int ref;        // what the parent's record will hold: node index, leaf reference or kNullLeaf
  int wait;
  bool folded;    // an internal child of at most kLeafMaxPrims primitives: referenced as a leaf, height 0
  float lo[3], hi[3], height;
};

__device__ __forceinline__ void write_node(const RefitView &v, int i, const PendingChild &a, const PendingChild &b,
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

// meta: [0] number of listed nodes, [1] height of the tree, [2] 1 once the root is finished, [3] rounds of k_refit_top
__global__ __launch_bounds__(kRefitThreads) void k_refit_local(RefitView v, uint32_t *__restrict__ top_list,
                                                               uint32_t *meta) {
  __shared__ float leaf_b[6][kRefitTile];           // boxes of the tile's leaves
  __shared__ unsigned char leaf_null[kRefitTile];   // leaf of an oversized sphere (kept out of the tree)
  __shared__ float node_b[7][kRefitTile];           // boxes and heights of the tile's finished nodes
  __shared__ unsigned short node_round[kRefitTile]; // round in which a node was finished, 0: not yet
  __shared__ unsigned char has_parent[kRefitTile];  // its parent lies inside the tile too
  __shared__ uint32_t n_listed, list_base;
  __shared__ int moved[3];                          // "a node was finished in round r" at [r % 3]
  const int base = blockIdx.x * kRefitTile;
  const int n_leaves = v.n_internal + 1;
  if (threadIdx.x == 0) {
    n_listed = 0;
    moved[0] = moved[1] = moved[2] = 0;
  }
  for (int p = threadIdx.x; p < kRefitTile; p += kRefitThreads) {
    node_round[p] = 0;
    has_parent[p] = 0;
    if (base + p < n_leaves) {
      const float4 lo = v.leaf_lo[base + p], hi = v.leaf_hi[base + p];
      leaf_b[0][p] = lo.x; leaf_b[1][p] = lo.y; leaf_b[2][p] = lo.z;
      leaf_b[3][p] = hi.x; leaf_b[4][p] = hi.y; leaf_b[5][p] = hi.z;
      leaf_null[p] = lo.w != 0.f;
    }
  }
  // this thread's nodes: inside the tile (todo) or across its boundary (outside)
  uint32_t todo = 0, outside = 0;
  int child[kRefitE][2], ref[kRefitE][2];
#pragma unroll
  for (int k = 0; k < kRefitE; ++k) {
    const int i = base + k * kRefitThreads + (int)threadIdx.x;
    child[k][0] = child[k][1] = ref[k][0] = ref[k][1] = 0;
    if (i >= v.n_internal) continue;
    const int first = v.range_first[i], count = v.range_count[i];
    if (!(first >= base && first + count <= base + kRefitTile)) {
      outside |= 1u << k;
      continue;
    }
    todo |= 1u << k;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int c = side ? v.right[i] : v.left[i];
      child[k][side] = c;
      ref[k][side] = c;
      if (c >= 0) {   // a small subtree is referenced as one leaf
        const int cc = v.range_count[c];
        if (cc <= kLeafMaxPrims) ref[k][side] = make_leaf_ref((uint32_t)v.range_first[c], (uint32_t)cc);
      }
    }
  }
  __syncthreads();
  const uint32_t inside = todo;
#pragma unroll
  for (int k = 0; k < kRefitE; ++k)
    if (inside & (1u << k)) {
      if (child[k][0] >= 0) has_parent[child[k][0] - base] = 1;
      if (child[k][1] >= 0) has_parent[child[k][1] - base] = 1;
    }
  // children of node k of this thread, as its record will hold them, from the tile's LDS
  auto gather = [&](int k, PendingChild pc[2]) {
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int c = child[k][side];
      if (c < 0) {
        const int p = ~c - base;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          pc[side].lo[q] = leaf_b[q][p];
          pc[side].hi[q] = leaf_b[3 + q][p];
        }
        pc[side].height = 0.f;
        pc[side].ref = leaf_null[p] ? kNullLeaf : make_leaf_ref((uint32_t)~c, 1u);
      } else {
        const int p = c - base;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          pc[side].lo[q] = node_b[q][p];
          pc[side].hi[q] = node_b[3 + q][p];
        }
        pc[side].ref = ref[k][side];
        pc[side].height = ref[k][side] < 0 ? 0.f : node_b[6][p];
      }
    }
  };
  // The rounds touch LDS only (a barrier waits for the wave's outstanding global stores too, and with a
  // store per finished node every one of ~25 rounds waited a memory round trip); the records go out
  // to global memory after the last round.
  for (uint32_t pass = 1;; ++pass) {
    bool progress = false;
#pragma unroll
    for (int k = 0; k < kRefitE; ++k) {
      if (!(todo & (1u << k))) continue;
      bool ready = true;
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const int c = child[k][side];
        if (c >= 0) {
          const uint32_t r = node_round[c - base];
          ready = ready && r != 0u && r < pass;
        }
      }
      if (!ready) continue;
      PendingChild pc[2];
      gather(k, pc);
      const int p = k * kRefitThreads + (int)threadIdx.x;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        node_b[q][p] = fminf(pc[0].lo[q], pc[1].lo[q]);
        node_b[3 + q][p] = fmaxf(pc[0].hi[q], pc[1].hi[q]);
      }
      node_b[6][p] = fmaxf(pc[0].height, pc[1].height) + 1.f;
      node_round[p] = (unsigned short)pass;
      todo &= ~(1u << k);
      progress = true;
    }
    // One barrier per round: it orders this round's LDS stores before the next round's loads, and
    // behind it everyone reads whether anyone moved.  (Three flags in rotation: the one for the next
    // round is cleared now, while nobody can still be reading it.)
    if (progress) moved[pass % 3u] = 1;
    if (threadIdx.x == 0) moved[(pass + 1u) % 3u] = 0;
    __syncthreads();
    if (!moved[pass % 3u]) break;
  }
  // every node inside the tile is finished now (its whole subtree is inside the tile): the records, and
  // for the nodes whose parent is not in the tile the box and a stamp, which a later launch reads
#pragma unroll
  for (int k = 0; k < kRefitE; ++k) {
    if (!(inside & (1u << k))) continue;
    PendingChild pc[2];
    gather(k, pc);
    const int i = base + k * kRefitThreads + (int)threadIdx.x;
    float box[7];
    write_node(v, i, pc[0], pc[1], box);
    if (!has_parent[i - base]) {   // the root of a subtree of this tile: a later launch reads its box
      v.box_lo[i] = make_float4(box[0], box[1], box[2], box[6]);
      v.box_hi[i] = make_float4(box[3], box[4], box[5], 0.f);
      v.stamp[i] = node_round[i - base];
    }
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
      top_list[list_base + slot] = i;
      v.stamp[i] = kListedBit | (list_base + slot);
      ++slot;
    }
}

// slow path of k_refit_top: one attempt at node i in round `pass` on global memory
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

// The listed nodes, level by level.  A launch with tile_shift = t finishes every listed node whose run of
// sorted positions lies inside one tile of 2^t positions (block b takes tile b); the last launch, one
// block with t = 32, takes whatever is left.  A block picks its nodes off the list, gives them LDS slots
// (the stamp of a picked node becomes kListedBit | its slot, which is how a parent finds a child that
// the same block finishes), fetches the boxes of children that an earlier launch finished, and then
// runs the rounds on registers and LDS alone.  A block that finds more nodes than it has slots leaves
// them to the last launch, which falls back to rounds on global memory when it, too, has too many.
__global__ __launch_bounds__(1024) void k_refit_upper(RefitView v, const uint32_t *__restrict__ top_list,
                                                      uint32_t *meta, uint32_t tile_shift, uint32_t last) {
  __shared__ float nb[7][kTopCap];
  __shared__ unsigned short rnd[kTopCap];
  __shared__ int slot_node[kTopCap];
  __shared__ uint32_t wave_base[16], n_picked;
  __shared__ int moved[3];
  const uint32_t n_listed = meta[0];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) moved[0] = moved[1] = moved[2] = 0;
  // ---- pick: unfinished listed nodes whose run lies in this block's tile, counted, then numbered
  auto mine = [&](uint32_t j) -> int {
    if (j >= n_listed) return -1;
    const int i = (int)top_list[j];
    if (!(v.stamp[i] & kListedBit)) return -1;   // finished by an earlier launch
    if (tile_shift < 32u) {
      const uint32_t first = (uint32_t)v.range_first[i], end = first + (uint32_t)v.range_count[i] - 1u;
      if ((first >> tile_shift) != blockIdx.x || (end >> tile_shift) != blockIdx.x) return -1;
    }
    return i;
  };
  uint32_t count = 0;
  for (uint32_t j0 = 0; j0 < n_listed; j0 += 1024u) count += mine(j0 + threadIdx.x) >= 0 ? 1u : 0u;
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
  bool done = false;
  float height = 0.f;
  if (picked > (uint32_t)kTopCap) {
    if (!last) return;   // too many for this block's LDS: the last launch takes them
    // the last launch, with more nodes than LDS holds (tens of millions of primitives): rounds on global memory
    for (uint32_t pass = kTopPassBase;; ++pass) {
      bool progress = false;
      for (uint32_t j = threadIdx.x; j < n_listed; j += 1024u) {
        const int i = (int)top_list[j];
        if (!(v.stamp[i] & kListedBit)) continue;
        if (try_finish(v, i, pass)) progress = true;
      }
      ++rounds;
      if (!__syncthreads_or(progress)) break;
    }
  } else if (picked) {
    uint32_t slot = wave_base[wave] + inc - count;
    for (uint32_t j0 = 0; j0 < n_listed; j0 += 1024u) {
      const int i = mine(j0 + threadIdx.x);
      if (i < 0) continue;
      slot_node[slot] = i;
      rnd[slot] = 0;
      ++slot;
    }
    __syncthreads();
    // (stamps are rewritten only now, after every thread has done its second look at them)
    for (uint32_t q = threadIdx.x; q < picked; q += 1024u) v.stamp[slot_node[q]] = kListedBit | q;
    __syncthreads();
    int node[kTopE];
    PendingChild pc[kTopE][2];
    uint32_t todo = 0;
#pragma unroll
    for (int k = 0; k < kTopE; ++k) {
      const uint32_t q = (uint32_t)k * 1024u + threadIdx.x;
      node[k] = -1;
      if (q >= picked) continue;
      const int i = slot_node[q];
      node[k] = i;
      todo |= 1u << k;
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        PendingChild &c = pc[k][side];
        const int cc = side ? v.right[i] : v.left[i];
        c.wait = -1;
        c.folded = false;
        float4 lo, hi;
        if (cc < 0) {
          lo = v.leaf_lo[~cc];
          hi = v.leaf_hi[~cc];
          c.ref = lo.w != 0.f ? kNullLeaf : make_leaf_ref((uint32_t)~cc, 1u);
          c.height = 0.f;
        } else {
          const uint32_t st = v.stamp[cc];
          const int n_under = v.range_count[cc];
          c.folded = n_under <= kLeafMaxPrims;
          c.ref = c.folded ? make_leaf_ref((uint32_t)v.range_first[cc], (uint32_t)n_under) : cc;
          if (st & kListedBit) {   // a child of a picked node that is still listed was picked too
            c.wait = (int)(st & ~kListedBit);
            lo = hi = make_float4(0.f, 0.f, 0.f, 0.f);
            c.height = 0.f;
          } else {
            lo = v.box_lo[cc];
            hi = v.box_hi[cc];
            c.height = c.folded ? 0.f : lo.w;
          }
        }
        c.lo[0] = lo.x; c.lo[1] = lo.y; c.lo[2] = lo.z;
        c.hi[0] = hi.x; c.hi[1] = hi.y; c.hi[2] = hi.z;
      }
    }
    __syncthreads();
    for (uint32_t pass = 1;; ++pass) {
      bool progress = false;
#pragma unroll
      for (int k = 0; k < kTopE; ++k) {
        if (!(todo & (1u << k))) continue;
        bool ready = true;
#pragma unroll
        for (int side = 0; side < 2; ++side)
          if (pc[k][side].wait >= 0) {
            const uint32_t r = rnd[pc[k][side].wait];
            ready = ready && r != 0u && r < pass;
          }
        if (!ready) continue;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          PendingChild &c = pc[k][side];
          if (c.wait >= 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              c.lo[q] = nb[q][c.wait];
              c.hi[q] = nb[3 + q][c.wait];
            }
            c.height = c.folded ? 0.f : nb[6][c.wait];
          }
        }
        const uint32_t q = (uint32_t)k * 1024u + threadIdx.x;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          nb[e][q] = fminf(pc[k][0].lo[e], pc[k][1].lo[e]);
          nb[3 + e][q] = fmaxf(pc[k][0].hi[e], pc[k][1].hi[e]);
        }
        nb[6][q] = fmaxf(pc[k][0].height, pc[k][1].height) + 1.f;
        rnd[q] = (unsigned short)pass;
        todo &= ~(1u << k);
        progress = true;
      }
      ++rounds;
      if (progress) moved[pass % 3u] = 1;
      if (threadIdx.x == 0) moved[(pass + 1u) % 3u] = 0;
      __syncthreads();
      if (!moved[pass % 3u]) break;
    }
    // (as in k_refit_local, global memory is written after the rounds: the children are in registers by now)
#pragma unroll
    for (int k = 0; k < kTopE; ++k) {
      if (node[k] < 0 || (todo & (1u << k))) continue;
      float box[7];
      write_node(v, node[k], pc[k][0], pc[k][1], box);
      v.box_lo[node[k]] = make_float4(box[0], box[1], box[2], box[6]);
      v.box_hi[node[k]] = make_float4(box[3], box[4], box[5], 0.f);
      v.stamp[node[k]] = rnd[(uint32_t)k * 1024u + threadIdx.x];
    }
    __syncthreads();
  }
  if (last && threadIdx.x == 0) {
    // (no progress with nodes left cannot happen in a tree; the host would see meta[2] == 0)
    done = v.n_internal > 0 && v.stamp[0] != 0u && !(v.stamp[0] & kListedBit);
    if (done) height = v.box_lo[0].w;
    meta[1] = done ? (uint32_t)height : 0u;
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
    const uint32_t n_counters = 256u * nblk, scan_tiles = (n_counters + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_scan_reduce, dim3(scan_tiles), dim3(kThreads), 0, stream, hist, n_counters, scan_sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, stream, scan_sums, scan_tiles);
    hipLaunchKernelGGL(k_scan_apply, dim3(scan_tiles), dim3(kThreads), 0, stream, hist, n_counters, scan_sums);
    hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(kThreads), 0, stream, keys[cur], vals[cur], keys[cur ^ 1],
                       vals[cur ^ 1], n, shift, hist, nblk);
    cur ^= 1;
  }
  return hipGetLastError();
}
void radix_sort_scratch(uint32_t n, size_t *hist_bytes, size_t *sums_bytes) {
  const uint32_t nblk = (n + kRsTile - 1) / kRsTile;
  *hist_bytes = (size_t)256 * nblk * sizeof(uint32_t);
  *sums_bytes = ((size_t)256 * nblk / kScanTile + 2) * sizeof(uint32_t);
}

struct Lbvh::Work {
  size_t capacity = 0;
  char *slab = nullptr;
  float4 *cr = nullptr, *leaf_lo = nullptr, *leaf_hi = nullptr, *box_lo = nullptr, *box_hi = nullptr;
  float4 *sorted_lo = nullptr, *sorted_hi = nullptr;  // the leaf boxes by sorted position
  uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr};
  uint32_t *hist = nullptr, *scan_sums = nullptr, *stamp = nullptr, *cbounds = nullptr;
  int *left = nullptr, *right = nullptr;
  uint32_t *top_list = nullptr;  // nodes that straddle a tile boundary of k_refit_local
  int *range_first = nullptr, *range_count = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
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
  }
  const uint32_t nblk = (n + kRsTile - 1) / kRsTile;
  if (n > work->capacity) {
    if (work->slab) hipFree(work->slab);
    work->slab = nullptr;
    work->capacity = 0;
    work->up_cr.clear();  // (lived in the slab)
    const size_t f4 = align256((size_t)n * sizeof(float4)), u4 = align256((size_t)n * sizeof(uint32_t));
    const size_t hist = align256((size_t)256 * nblk * sizeof(uint32_t));
    const size_t sums = align256(((size_t)256 * nblk / kScanTile + 2) * sizeof(uint32_t));
    const size_t total = 7 * f4 + 10 * u4 + hist + sums + 512;
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
    work->top_list = (uint32_t *)take(u4);
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
  BVH_HIP(hipMemsetAsync(w.cbounds, 0xff, 3 * sizeof(uint32_t), stream));      // running minima
  BVH_HIP(hipMemsetAsync(w.cbounds + 3, 0x00, 3 * sizeof(uint32_t), stream));  // running maxima
  const uint32_t blocks_n = (n + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(k_leaf_boxes, dim3(std::min(blocks_n, 256u)), dim3(kThreads), 0, stream, w.cr,
                     n_spheres, d_tri, n, bp, w.leaf_lo, w.leaf_hi, w.cbounds);
  hipLaunchKernelGGL(k_morton, dim3(blocks_n), dim3(kThreads), 0, stream, w.leaf_lo, w.leaf_hi, n,
                     w.cbounds, w.keys[0], w.vals[0]);
  int cur = 0;
  for (uint32_t shift = 0; shift < 32; shift += 8) {
    hipLaunchKernelGGL(k_rs_hist, dim3(nblk), dim3(kThreads), 0, stream, w.keys[cur], n, shift, w.hist,
                       nblk);
    const uint32_t n_counters = 256u * nblk, scan_tiles = (n_counters + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_scan_reduce, dim3(scan_tiles), dim3(kThreads), 0, stream, w.hist, n_counters,
                       w.scan_sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, stream, w.scan_sums, scan_tiles);
    hipLaunchKernelGGL(k_scan_apply, dim3(scan_tiles), dim3(kThreads), 0, stream, w.hist, n_counters,
                       w.scan_sums);
    // the last pass writes the sorted order where the traversal reads it
    hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(kThreads), 0, stream, w.keys[cur], w.vals[cur],
                       w.keys[cur ^ 1], shift == 24u ? sorted_prim : w.vals[cur ^ 1], n, shift, w.hist, nblk);
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
    BVH_HIP(hipMemsetAsync(w.stamp, 0, (size_t)(n - 1) * sizeof(uint32_t), stream));
    uint32_t *meta = w.cbounds + 64;
    BVH_HIP(hipMemsetAsync(meta, 0, 256, stream));
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
    constexpr uint32_t kMidShift = 17;
    const uint32_t mid_tiles = (n + (1u << kMidShift) - 1u) >> kMidShift;
    if (mid_tiles > 1u)
      hipLaunchKernelGGL(k_refit_upper, dim3(mid_tiles), dim3(1024), 0, stream, rv, w.top_list, meta, kMidShift, 0u);
    hipLaunchKernelGGL(k_refit_upper, dim3(1), dim3(1024), 0, stream, rv, w.top_list, meta, 32u, 1u);
    // ONE look at the device per build: the height sizes the traversal stack of the launches that follow
    uint32_t h_meta[4] = {0, 0, 0, 0};
    BVH_HIP(hipMemcpyAsync(h_meta, meta, sizeof h_meta, hipMemcpyDeviceToHost, stream));
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
