// C ABI of librt_hip.so (include/rt_hip.h): context management, uploads, launches.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_hostmath.h"
#include "rt_kernels.h"
#include "rt_lbvh.h"
#include "rt_wavefront.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define RT_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(RT_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
                  __FILE__, __LINE__);                                                    \
  } while (0)

// trip count of "for (int i = 0; i < limit; i++)" with a float limit
int trip_count(float limit) {
  if (!(limit > 0.0f)) return 0;
  float c = std::ceil(limit);
  if (c > 16777216.0f) c = 16777216.0f;
  return (int)c;
}

// No exception crosses the C ABI: the bodies of the entry points that touch containers or
// strings run inside this.
template <class F>
int guarded(const char *who, F body) {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    return fail(RT_ERR_NOMEM, "%s: out of host memory", who);
  } catch (const std::length_error &) {
    return fail(RT_ERR_NOMEM, "%s: size exceeds what a host container can hold", who);
  } catch (const std::exception &e) {
    return fail(RT_ERR_INVALID, "%s: %s", who, e.what());
  } catch (...) {
    return fail(RT_ERR_INVALID, "%s: unknown exception", who);
  }
}

}  // namespace

struct rt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  uint32_t width = 0, height = 0, mode = 0, flags = 0;
  size_t n_pixels = 0;

  rt::Params P{};
  bool have_params = false;

  float4 *d_geom = nullptr;      // centre.xyz, r*r
  float4 *d_material = nullptr;  // 3 per sphere
  uint32_t n_spheres = 0;
  bool have_scene = false;
  std::vector<float> h_spheres;  // the 64-byte records as uploaded (for rebuilds)

  float4 *d_tri = nullptr;       // v0, v1, v2 per mesh triangle
  uint32_t n_tris = 0;
  uint32_t n_tris_alloc = 0;     // triangles d_tri was allocated for
  float4 *d_tri_spare = nullptr; // the previous mesh's array, reused by an upload of the same size
  bool bvh_fallback = false;     // an LBVH was built but is too tall to traverse: brute force instead
  uint32_t bvh_height_limit = (uint32_t)rt::kBvhStackEntries;  // RT_BVH_MAX_HEIGHT lowers it (test hook for the fallback)
  uint32_t tri_spare_n = 0;
  float mesh_lo[3] = {0, 0, 0}, mesh_hi[3] = {0, 0, 0};  // bounds of the mesh vertices
  float mesh_material[12] = {0};

  uint32_t *d_mesh_report = nullptr;  // 7 words an upload's device checks report in (kept between uploads)
  uint32_t *h_mesh_report = nullptr;  // where they are read back to (pinned: no staging copy on the way)
  // the mesh's material record as the device holds it (slot = the sphere count it was written behind):
  // an upload that changes neither skips the blocking 48-byte copy
  float mesh_material_dev[12] = {0};
  int64_t mesh_material_slot = -1;
  float *d_ior = nullptr;        // per sphere, > 0: dielectric (optional; null: none)
  bool have_ior = false;

  float4 *d_texels = nullptr;    // albedo texture (optional), decoded to linear rgb floats
  float *d_tex_lut = nullptr;    // byte -> linear value (used while a texture is being bound)
  rt::Texture *d_tex = nullptr;  // device copy of the binding
  bool have_tex = false;

  rt::Lbvh bvh;                  // device LBVH + build workspace
  bool use_bvh = false;
  float bvh_cam[3] = {0.f, 0.f, 0.f};  // camera position the leaf boxes were sized for

  rt::WavefrontState wf;         // queues of the wavefront pipeline

  float4 *d_accum = nullptr;
  bool own_accum = false;
  uchar4 *d_rgba8 = nullptr;
  int32_t *d_ids = nullptr;
  unsigned long long *d_rays = nullptr;
  // tile scheduling of the megakernels: cost measured by the last launch, order for the next
  uint32_t *d_tile_cost = nullptr, *d_tile_order = nullptr;
  uint32_t tiles_x = 0, n_tiles = 0;
  bool order_valid = false, use_tile_order = true, use_probe = true;
  // d_tile_order holds two tables of n_tiles: the order by the tiles' own costs, and the order by
  // the largest cost in each tile's 3x3 neighbourhood, which LBVH scenes use while the geometry
  // has changed since the costs were measured (a mesh re-uploaded every frame)
  bool order_moved = false;
  // the order is refreshed after launches 1, 2, 4, 8, then every 16th since the last change
  // of scene or parameters; in between the costs keep accumulating (atomicMax)
  uint32_t order_interval = 1, launches_since_order = 0;
  // The refresh of the order after a launch runs on a stream of its own, tied to the context's by two
  // events: it needs the launch's costs and the NEXT launch needs its order, but nothing else queued in
  // between (an upload, an LBVH rebuild) has to wait for it.
  hipStream_t order_stream = nullptr;
  hipEvent_t order_in = nullptr, order_out = nullptr;
  bool order_pending = false;

  // HIP event pairs around each rt_render; resolved lazily so rendering never waits
  struct EventPair {
    hipEvent_t begin = nullptr, end = nullptr;
  };
  std::vector<EventPair> ev_free, ev_pending;
  uint64_t frames = 0;
  float last_ms = 0.f, total_ms = 0.f;
  uint32_t last_launches = 0;
};

namespace {

constexpr size_t kMaxPendingEvents = 512;
// the context's counters: 256 bytes (segments, traversal tallies), then the spread segment counters
constexpr size_t kRaySlotsOffset = 256;
constexpr size_t kRayBytes = kRaySlotsOffset + (size_t)rt::kRaySlots * rt::kRaySlotStride;

// folds finished event pairs into the statistics; keep_newest pairs may stay pending
int collect_timing(rt_ctx *c, size_t keep_newest) {
  while (c->ev_pending.size() > keep_newest) {
    rt_ctx::EventPair p = c->ev_pending.front();
    float ms = 0.f;
    RT_HIP(hipEventSynchronize(p.end));
    RT_HIP(hipEventElapsedTime(&ms, p.begin, p.end));
    c->last_ms = ms;
    c->total_ms += ms;
    c->ev_pending.erase(c->ev_pending.begin());
    c->ev_free.push_back(p);
  }
  return RT_OK;
}

int acquire_events(rt_ctx *c, rt_ctx::EventPair *out) {
  if (c->ev_free.empty()) {
    if (c->ev_pending.size() >= kMaxPendingEvents) {
      if (collect_timing(c, kMaxPendingEvents / 2) != RT_OK) return RT_ERR_DEVICE;
    } else {
      rt_ctx::EventPair p;
      RT_HIP(hipEventCreate(&p.begin));
      RT_HIP(hipEventCreate(&p.end));
      c->ev_free.push_back(p);
    }
  }
  *out = c->ev_free.back();
  c->ev_free.pop_back();
  return RT_OK;
}

rt::Scene scene_view(const rt_ctx *c) {
  rt::Scene S{};
  S.geom = c->d_geom;
  S.material = c->d_material;
  S.n_spheres = c->n_spheres;
  S.tri = c->d_tri;
  S.n_tris = c->n_tris;
  S.tex = c->have_tex ? c->d_tex : nullptr;
  S.ior = c->have_ior ? c->d_ior : nullptr;
  if (c->use_bvh) {
    S.nodes = c->bvh.nodes;
    S.sorted_prim = c->bvh.sorted_prim;
    S.sorted_geom = c->bvh.sorted_geom;
    S.large_geom = c->bvh.large_geom;
    S.large_prim = c->bvh.large_prim;
    S.n_large = c->bvh.n_large;
    S.stack_entries = std::max(c->bvh.height, 1u);
  }
  return S;
}

// (Re)builds the LBVH over spheres + mesh when the scene is large enough or forced.
int rebuild_accel(rt_ctx *c) {
  const uint32_t n_prims = c->n_spheres + c->n_tris;
  c->use_bvh = false;
  c->bvh_fallback = false;
  const bool want = (c->flags & RT_FLAG_FORCE_BVH) != 0 ||
                    (!(c->flags & RT_FLAG_FORCE_BRUTE) && n_prims > 16u);
  if (!want || n_prims == 0) return RT_OK;
  std::string err;
  const float cam[3] = {c->have_params ? c->P.cam_x : 0.f, c->have_params ? c->P.cam_y : 0.f,
                        c->have_params ? c->P.cam_z : 0.f};
  int rc = c->bvh.build(c->d_geom, c->h_spheres.data(), c->n_spheres, c->d_tri, c->mesh_lo,
                        c->mesh_hi, c->n_tris, cam, c->stream, &err);
  if (rc != RT_OK) return fail(rc, "LBVH build: %s", err.c_str());
  std::memcpy(c->bvh_cam, cam, sizeof cam);
  if (std::getenv("RT_BVH_DEBUG"))
    std::fprintf(stderr, "rt_hip: LBVH over %u primitives: height %u, %u nodes across refit tiles, %u rounds on global memory\n",
                 n_prims, c->bvh.height, c->bvh.refit_listed, c->bvh.refit_rounds);
  if (c->bvh.height > c->bvh_height_limit) {
    // taller than any launch can give a stack for: trace by brute force (correct, slow) and
    // say so in rt_stats.bvh_fallback; the upload itself succeeds and rt_last_error is left alone
    c->bvh_fallback = true;
    return RT_OK;
  }
  c->use_bvh = true;
  return RT_OK;
}

}  // namespace

extern "C" {

int rt_destroy(rt_ctx *c);
int rt_clear(rt_ctx *c);

const char *rt_last_error(void) { return g_last_error.c_str(); }
const char *rt_version(void) { return "rt_hip 0.1 (gfx950)"; }

static int rt_create_impl(const rt_config *cfg, rt_ctx **out_ctx) {
  if (!cfg || !out_ctx) return fail(RT_ERR_INVALID, "rt_create: null argument");
  if (cfg->struct_size != sizeof(rt_config))
    return fail(RT_ERR_INVALID, "rt_create: struct_size %u, expected %zu", cfg->struct_size,
                sizeof(rt_config));
  if (cfg->width == 0 || cfg->height == 0 || cfg->width > 65536u || cfg->height > 65536u ||
      (uint64_t)cfg->width * cfg->height > (1ull << 28))
    return fail(RT_ERR_INVALID, "rt_create: bad image size %ux%u (at most 2^28 pixels)", cfg->width,
                cfg->height);
  if (cfg->mode > RT_MODE_MEGAKERNEL_NESTED)
    return fail(RT_ERR_INVALID, "rt_create: unknown mode %u", cfg->mode);
  if ((cfg->flags & RT_FLAG_FORCE_BRUTE) && (cfg->flags & RT_FLAG_FORCE_BVH))
    return fail(RT_ERR_INVALID, "rt_create: FORCE_BRUTE and FORCE_BVH are exclusive");
  int n_dev = 0;
  RT_HIP(hipGetDeviceCount(&n_dev));
  if (cfg->device < 0 || cfg->device >= n_dev)
    return fail(RT_ERR_INVALID, "rt_create: device %d of %d", cfg->device, n_dev);
  RT_HIP(hipSetDevice(cfg->device));

  rt_ctx *c = new (std::nothrow) rt_ctx();
  if (!c) return fail(RT_ERR_NOMEM, "rt_create: out of host memory");
  c->device = cfg->device;
  c->stream = (hipStream_t)cfg->stream;
  c->width = cfg->width;
  c->height = cfg->height;
  c->mode = cfg->mode;
  c->flags = cfg->flags;
  c->n_pixels = (size_t)cfg->width * cfg->height;
  // wavefront mode hands the thin tail of the active queue to one drain launch; the
  // switch point can be overridden for experiments
  c->wf.drain_threshold = (uint32_t)(c->n_pixels / 2);
  if (const char *e = std::getenv("RT_WF_DRAIN_THRESHOLD")) c->wf.drain_threshold = (uint32_t)std::strtoul(e, nullptr, 10);
  if (const char *e = std::getenv("RT_WF_REGROUP")) c->wf.regroup = e[0] != '0';

  auto bail = [&](hipError_t e, const char *what) {
    int rc = fail(RT_ERR_DEVICE, "rt_create: %s: %s", what, hipGetErrorString(e));
    rt_destroy(c);
    return rc;
  };
  hipError_t e;
  if (cfg->accum_buffer) {
    c->d_accum = (float4 *)cfg->accum_buffer;
  } else {
    if ((e = hipMalloc(&c->d_accum, c->n_pixels * sizeof(float4))) != hipSuccess)
      return bail(e, "hipMalloc(accum)");
    c->own_accum = true;
  }
  if ((e = hipMalloc(&c->d_rgba8, c->n_pixels * sizeof(uchar4))) != hipSuccess)
    return bail(e, "hipMalloc(rgba8)");
  if ((e = hipMalloc(&c->d_ids, c->n_pixels * sizeof(int32_t))) != hipSuccess)
    return bail(e, "hipMalloc(ids)");
  if ((e = hipMalloc(&c->d_rays, kRayBytes)) != hipSuccess) return bail(e, "hipMalloc(rays)");
  c->tiles_x = (cfg->width + 15u) / 16u;
  c->n_tiles = c->tiles_x * ((cfg->height + 15u) / 16u);
  if ((e = hipMalloc(&c->d_tile_cost, c->n_tiles * sizeof(uint32_t))) != hipSuccess)
    return bail(e, "hipMalloc(tile cost)");
  if ((e = hipMalloc(&c->d_tile_order, 2 * (size_t)c->n_tiles * sizeof(uint32_t))) != hipSuccess)
    return bail(e, "hipMalloc(tile order)");
  if ((e = hipMemsetAsync(c->d_tile_cost, 0, c->n_tiles * sizeof(uint32_t), c->stream)) != hipSuccess)
    return bail(e, "hipMemsetAsync(tile cost)");
  if (const char *env = std::getenv("RT_TILE_ORDER")) c->use_tile_order = env[0] != '0';
  if (const char *env = std::getenv("RT_BVH_MAX_HEIGHT"))
    c->bvh_height_limit = std::min<uint32_t>((uint32_t)std::strtoul(env, nullptr, 10), (uint32_t)rt::kBvhStackEntries);
  if (const char *env = std::getenv("RT_TILE_PROBE")) c->use_probe = env[0] != '0';
  int rc = rt_clear(c);
  if (rc != RT_OK) {
    rt_destroy(c);
    return rc;
  }
  *out_ctx = c;
  return RT_OK;
}

int rt_destroy(rt_ctx *c) {
  if (!c) return RT_OK;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  if (c->order_stream) {
    hipStreamSynchronize(c->order_stream);
    hipStreamDestroy(c->order_stream);
  }
  if (c->order_in) hipEventDestroy(c->order_in);
  if (c->order_out) hipEventDestroy(c->order_out);
  if (c->own_accum && c->d_accum) hipFree(c->d_accum);
  if (c->d_rgba8) hipFree(c->d_rgba8);
  if (c->d_ids) hipFree(c->d_ids);
  if (c->d_rays) hipFree(c->d_rays);
  if (c->d_tile_cost) hipFree(c->d_tile_cost);
  if (c->d_tile_order) hipFree(c->d_tile_order);
  if (c->d_geom) hipFree(c->d_geom);
  if (c->d_material) hipFree(c->d_material);
  if (c->d_tri) hipFree(c->d_tri);
  if (c->d_tri_spare) hipFree(c->d_tri_spare);
  if (c->d_ior) hipFree(c->d_ior);
  if (c->d_mesh_report) hipFree(c->d_mesh_report);
  if (c->h_mesh_report) hipHostFree(c->h_mesh_report);
  if (c->d_texels) hipFree(c->d_texels);
  if (c->d_tex_lut) hipFree(c->d_tex_lut);
  if (c->d_tex) hipFree(c->d_tex);
  c->bvh.release();
  c->wf.release();
  for (auto &p : c->ev_pending) c->ev_free.push_back(p);
  for (auto &p : c->ev_free) {
    hipEventDestroy(p.begin);
    hipEventDestroy(p.end);
  }
  delete c;
  return RT_OK;
}

int rt_clear(rt_ctx *c) {
  if (!c) return fail(RT_ERR_INVALID, "rt_clear: null context");
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipMemsetAsync(c->d_accum, 0, c->n_pixels * sizeof(float4), c->stream));
  RT_HIP(hipMemsetAsync(c->d_rgba8, 0, c->n_pixels * sizeof(uchar4), c->stream));
  RT_HIP(hipMemsetAsync(c->d_ids, 0xff, c->n_pixels * sizeof(int32_t), c->stream));
  RT_HIP(hipMemsetAsync(c->d_rays, 0, kRayBytes, c->stream));
  c->frames = 0;
  c->last_ms = c->total_ms = 0.f;
  c->last_launches = 0;
  for (auto &p : c->ev_pending) c->ev_free.push_back(p);
  c->ev_pending.clear();
  return RT_OK;
}

// The kernels' parameter block from the shader's 272-byte uniform block, for a width x height image
static void fill_params(const void *ubo272, uint32_t width, uint32_t height, rt::Params &P) {
  rt_ubo u;
  std::memcpy(&u, ubo272, sizeof u);
  rt::mat4_inverse(u.view, P.inv_view);
  rt::mat4_inverse(u.projection, P.inv_proj);
  P.viewport_w = u.viewport_size[0];
  P.viewport_h = u.viewport_size[1];
  P.cam_x = u.camera_position[0];
  P.cam_y = u.camera_position[1];
  P.cam_z = u.camera_position[2];
  P.spp_f = u.render_settings[0];
  P.n_spp = trip_count(u.render_settings[0]);
  P.n_bounce = trip_count(u.render_settings[1]);
  for (int k = 0; k < 3; ++k) {
    P.ground[k] = u.ground_colour[k];
    P.horizon[k] = u.sky_colour_horizon[k];
    P.zenith[k] = u.sky_colour_zenith[k];
    P.sun[k] = u.sun_position[k];
  }
  P.sun_focus = u.sun_focus;
  P.sun_intensity = u.sun_intensity;
  P.defocus = u.defocus_strength;
  P.diverge = u.diverge_strength;
  P.width = width;
  P.height = height;
  P.fwidth = (float)width;
  P.fheight = (float)height;
  P.defocus_over_w = P.defocus / P.fwidth;
  P.diverge_over_w = P.diverge / P.fwidth;
  P.rcp_w = 1.0f / P.fwidth;
  P.rcp_h = 1.0f / P.fheight;
}

static int rt_set_params_impl(rt_ctx *c, const void *ubo272) {
  if (!c || !ubo272) return fail(RT_ERR_INVALID, "rt_set_params: null argument");
  const rt::Params before = c->P;
  rt::Params &P = c->P;
  fill_params(ubo272, c->width, c->height, P);
  if (!c->have_params || std::memcmp(&before, &P, sizeof P) != 0) {
    c->order_interval = 1;  // a new view: measure tile costs afresh
    c->launches_since_order = 0;
  }
  c->have_params = true;
  if (c->use_bvh && (P.cam_x != c->bvh_cam[0] || P.cam_y != c->bvh_cam[1] || P.cam_z != c->bvh_cam[2])) {
    // leaf boxes are sized from the distance to the camera: rebuild for the new position
    RT_HIP(hipSetDevice(c->device));
    RT_HIP(hipStreamSynchronize(c->stream));
    int rc = rebuild_accel(c);
    if (rc != RT_OK) return rc;
    const float cam[3] = {P.cam_x, P.cam_y, P.cam_z};
    std::memcpy(c->bvh_cam, cam, sizeof cam);
  }
  return RT_OK;
}

static int rt_set_spheres_impl(rt_ctx *c, const void *spheres64, uint32_t count) {
  if (!c) return fail(RT_ERR_INVALID, "rt_set_spheres: null context");
  if (count && !spheres64) return fail(RT_ERR_INVALID, "rt_set_spheres: null data");
  if (count > (1u << 26)) return fail(RT_ERR_INVALID, "rt_set_spheres: %u spheres", count);
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipStreamSynchronize(c->stream));  // previous launches may still read the old scene
  // until everything below has succeeded the context holds no scene: a failed upload must
  // not leave counts that point past freed or unallocated buffers
  c->have_scene = false;
  c->use_bvh = false;
  c->n_spheres = 0;
  c->have_ior = false;  // indices of refraction belong to the sphere set they were given for
  if (c->d_geom) RT_HIP(hipFree(c->d_geom));
  c->d_geom = nullptr;
  if (c->d_material) RT_HIP(hipFree(c->d_material));
  c->d_material = nullptr;
  c->mesh_material_slot = -1;
  // the brute-force search fetches geometry four records at a time: pad, and clear the padding
  const size_t alloc_n = ((size_t)count + 3u) / 4u * 4u + 4u;
  RT_HIP(hipMalloc(&c->d_geom, alloc_n * sizeof(float4)));
  RT_HIP(hipMemset(c->d_geom, 0, alloc_n * sizeof(float4)));
  RT_HIP(hipMalloc(&c->d_material, ((size_t)count + 1) * 3 * sizeof(float4)));  // + the mesh's
  RT_HIP(hipMemcpy(c->d_material + 3 * (size_t)count, c->mesh_material, sizeof c->mesh_material,
                   hipMemcpyHostToDevice));
  std::memcpy(c->mesh_material_dev, c->mesh_material, sizeof c->mesh_material);
  c->mesh_material_slot = (int64_t)count;
  c->h_spheres.assign((const float *)spheres64, (const float *)spheres64 + (size_t)count * 16);
  if (count) {
    // repack the 64-byte records: geometry (centre, r*r) apart from the material,
    // so the closest-hit loop streams 16 bytes per sphere instead of 64
    std::vector<float> geom((size_t)count * 4), mat((size_t)count * 12);
    const float *s = c->h_spheres.data();
    for (uint32_t i = 0; i < count; ++i) {
      const float *r = s + (size_t)i * 16;
      geom[4 * (size_t)i + 0] = r[0];
      geom[4 * (size_t)i + 1] = r[1];
      geom[4 * (size_t)i + 2] = r[2];
      geom[4 * (size_t)i + 3] = r[3] * r[3];
      std::memcpy(&mat[12 * (size_t)i], r + 4, 48);
    }
    RT_HIP(hipMemcpy(c->d_geom, geom.data(), geom.size() * sizeof(float), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(c->d_material, mat.data(), mat.size() * sizeof(float),
                     hipMemcpyHostToDevice));
  }
  c->n_spheres = count;
  c->order_moved = true;  // the costs on record belong to the geometry that is being replaced
  c->order_interval = 1;
  c->launches_since_order = 0;
  int rc = rebuild_accel(c);
  if (rc != RT_OK) return rc;
  c->have_scene = true;
  return RT_OK;
}

// on_device: vertices and indices are device addresses (rt_set_mesh_device), else host memory
static int set_mesh(rt_ctx *c, const float *vertices, uint32_t n_vertices, const uint32_t *indices,
                    uint32_t n_triangles, const rt_material *material, bool on_device) {
  if (!c) return fail(RT_ERR_INVALID, "rt_set_mesh: null context");
  if (n_triangles && (!vertices || !indices || !material))
    return fail(RT_ERR_INVALID, "rt_set_mesh: null data");
  if (n_triangles > (1u << 26)) return fail(RT_ERR_INVALID, "rt_set_mesh: %u triangles", n_triangles);
  if (n_triangles && n_vertices == 0) return fail(RT_ERR_INVALID, "rt_set_mesh: triangles without vertices");
  RT_HIP(hipSetDevice(c->device));
  // A mesh re-uploaded every frame from device memory into the spare array of its size touches nothing a
  // launch in flight reads (the spare was last read before the previous upload, which waited for its own
  // report), and everything it enqueues is ordered behind those launches on the context's stream: no need
  // to drain the stream first.  Every other case may allocate, free or overwrite what is in use.
  const bool in_place = on_device && n_triangles && c->d_tri_spare && c->tri_spare_n == n_triangles &&
                        c->d_mesh_report && c->h_mesh_report;
  if (!in_place) RT_HIP(hipStreamSynchronize(c->stream));

  // The new mesh is expanded, checked and measured on the device BEFORE the old one is
  // dropped, so a rejected upload (index out of range) leaves the context as it was.
  float4 *new_tri = nullptr;
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  if (n_triangles) {
    float *d_v = nullptr;
    uint32_t *d_i = nullptr;
    // (no allocation or release per upload where it can be helped: hipFree waits for the device)
    hipError_t e = c->d_mesh_report ? hipSuccess : hipMalloc(&c->d_mesh_report, 7 * sizeof(uint32_t));
    if (e == hipSuccess && !c->h_mesh_report)
      e = hipHostMalloc((void **)&c->h_mesh_report, 8 * sizeof(uint32_t), hipHostMallocDefault);
    uint32_t *d_report = c->d_mesh_report;
    uint32_t *report = c->h_mesh_report;
    // the triangle array of the previous mesh is reused when the new one has as many triangles
    // and a spare exists (a mesh re-uploaded every frame then allocates nothing here)
    if (e == hipSuccess) {
      if (c->d_tri_spare && c->tri_spare_n == n_triangles) {
        new_tri = c->d_tri_spare;
        c->d_tri_spare = nullptr;
      } else {
        e = hipMalloc(&new_tri, (size_t)n_triangles * 3 * sizeof(float4));
      }
    }
    if (on_device) {
      d_v = const_cast<float *>(vertices);
      d_i = const_cast<uint32_t *>(indices);
    } else {
      if (e == hipSuccess) e = hipMalloc(&d_v, (size_t)n_vertices * 3 * sizeof(float));
      if (e == hipSuccess) e = hipMalloc(&d_i, (size_t)n_triangles * 3 * sizeof(uint32_t));
      if (e == hipSuccess)
        e = hipMemcpyAsync(d_v, vertices, (size_t)n_vertices * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(d_i, indices, (size_t)n_triangles * 3 * sizeof(uint32_t), hipMemcpyHostToDevice,
                           c->stream);
    }
    if (e == hipSuccess)
      e = rt::launch_expand_triangles(d_v, n_vertices, d_i, n_triangles, new_tri, d_report, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(report, d_report, 7 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    uint32_t bad_index = 0;
    if (e == hipSuccess && report[0] != 0xffffffffu) {
      if (on_device) e = hipMemcpy(&bad_index, d_i + report[0], sizeof bad_index, hipMemcpyDeviceToHost);
      else bad_index = indices[report[0]];
    }
    if (!on_device) {
      hipFree(d_v);
      hipFree(d_i);
    }
    if (e != hipSuccess) {
      hipFree(new_tri);
      return fail(RT_ERR_DEVICE, "rt_set_mesh: %s", hipGetErrorString(e));
    }
    if (report[0] != 0xffffffffu) {
      hipFree(new_tri);
      return fail(RT_ERR_INVALID, "rt_set_mesh: index %u at position %u, %u vertices", bad_index, report[0],
                  n_vertices);
    }
    for (int k = 0; k < 3; ++k) {  // ordered bits back to floats
      auto back = [](uint32_t u) {
        u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
        float f;
        std::memcpy(&f, &u, sizeof f);
        return f;
      };
      lo[k] = back(report[1 + k]);
      hi[k] = back(report[4 + k]);
    }
  }

  c->have_scene = false;  // restored once everything below has succeeded (see rt_set_spheres)
  c->use_bvh = false;
  c->n_tris = 0;
  // the outgoing triangle array becomes the spare for the next upload -- when it has the size of the
  // incoming one, the sign of a mesh re-uploaded every frame; anything else is freed now instead of
  // sitting dead (48 bytes per triangle) until the context goes
  if (c->d_tri_spare) RT_HIP(hipFree(c->d_tri_spare));
  c->d_tri_spare = nullptr;
  c->tri_spare_n = 0;
  if (c->d_tri && c->n_tris_alloc == n_triangles && n_triangles) {
    c->d_tri_spare = c->d_tri;
    c->tri_spare_n = c->n_tris_alloc;
  } else if (c->d_tri) {
    RT_HIP(hipFree(c->d_tri));
  }
  c->d_tri = new_tri;
  c->n_tris_alloc = n_triangles;
  if (material) std::memcpy(c->mesh_material, material, sizeof c->mesh_material);
  std::memcpy(c->mesh_lo, lo, sizeof lo);
  std::memcpy(c->mesh_hi, hi, sizeof hi);
  c->n_tris = n_triangles;
  c->order_moved = true;  // the costs on record belong to the geometry that is being replaced
  c->order_interval = 1;
  c->launches_since_order = 0;
  if (!c->d_material) {  // no spheres uploaded yet (or an upload failed half way): an empty sphere set
    if (c->d_geom) RT_HIP(hipFree(c->d_geom));
    c->d_geom = nullptr;
    RT_HIP(hipMalloc(&c->d_geom, 4 * sizeof(float4)));
    RT_HIP(hipMemset(c->d_geom, 0, 4 * sizeof(float4)));
    RT_HIP(hipMalloc(&c->d_material, 3 * sizeof(float4)));
    c->mesh_material_slot = -1;
    c->n_spheres = 0;
  }
  if (c->mesh_material_slot != (int64_t)c->n_spheres ||
      std::memcmp(c->mesh_material_dev, c->mesh_material, sizeof c->mesh_material) != 0) {
    c->mesh_material_slot = -1;
    RT_HIP(hipMemcpy(c->d_material + 3 * (size_t)c->n_spheres, c->mesh_material, sizeof c->mesh_material,
                     hipMemcpyHostToDevice));
    std::memcpy(c->mesh_material_dev, c->mesh_material, sizeof c->mesh_material);
    c->mesh_material_slot = (int64_t)c->n_spheres;
  }
  int rc = rebuild_accel(c);
  if (rc != RT_OK) return rc;
  c->have_scene = true;
  return RT_OK;
}

static int rt_set_dielectric_impl(rt_ctx *c, const float *ior, uint32_t count) {
  if (!c) return fail(RT_ERR_INVALID, "rt_set_dielectric: null context");
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipStreamSynchronize(c->stream));  // launches in flight may still read the old table
  c->have_ior = false;
  if (!ior || count == 0) return RT_OK;
  if (count != c->n_spheres)
    return fail(RT_ERR_INVALID, "rt_set_dielectric: %u values for %u spheres", count, c->n_spheres);
  bool any = false;
  for (uint32_t i = 0; i < count; ++i) {
    if (ior[i] != ior[i] || ior[i] > 1.0e6f)
      return fail(RT_ERR_INVALID, "rt_set_dielectric: index of refraction %g for sphere %u", (double)ior[i], i);
    any = any || ior[i] > 0.0f;
  }
  if (!any) return RT_OK;
  if (c->d_ior) RT_HIP(hipFree(c->d_ior));
  c->d_ior = nullptr;
  RT_HIP(hipMalloc(&c->d_ior, (size_t)count * sizeof(float)));
  RT_HIP(hipMemcpy(c->d_ior, ior, (size_t)count * sizeof(float), hipMemcpyHostToDevice));
  c->have_ior = true;
  c->order_interval = 1;
  c->launches_since_order = 0;
  return RT_OK;
}

static int rt_set_texture_impl(rt_ctx *c, const rt_texture_desc *desc, const void *rgba8) {
  if (!c) return fail(RT_ERR_INVALID, "rt_set_texture: null context");
  if (desc) {
    if (desc->struct_size != sizeof(rt_texture_desc))
      return fail(RT_ERR_INVALID, "rt_set_texture: struct_size %u, expected %zu", desc->struct_size,
                  sizeof(rt_texture_desc));
    if (!rgba8) return fail(RT_ERR_INVALID, "rt_set_texture: null texels");
    if (desc->width == 0 || desc->height == 0 || desc->width > 32768u || desc->height > 32768u)
      return fail(RT_ERR_INVALID, "rt_set_texture: bad size %ux%u", desc->width, desc->height);
    if (desc->mapping > RT_TEX_MAP_OCTAHEDRAL)
      return fail(RT_ERR_INVALID, "rt_set_texture: unknown mapping %u", desc->mapping);
    if (desc->flags & ~RT_TEX_SRGB) return fail(RT_ERR_INVALID, "rt_set_texture: unknown flags 0x%x", desc->flags);
  }
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipStreamSynchronize(c->stream));  // launches in flight may still sample the old one
  if (c->d_texels) RT_HIP(hipFree(c->d_texels));
  c->d_texels = nullptr;
  c->have_tex = false;
  if (!desc) return RT_OK;
  const size_t n = (size_t)desc->width * desc->height;
  RT_HIP(hipMalloc(&c->d_texels, n * sizeof(float4)));
  if (!c->d_tex_lut) RT_HIP(hipMalloc(&c->d_tex_lut, 256 * sizeof(float)));
  float lut[256];
  for (int i = 0; i < 256; ++i) {
    if (desc->flags & RT_TEX_SRGB) {
      // IEC 61966-2-1 decoding, evaluated in double and rounded once
      const double v = (double)i / 255.0;
      lut[i] = (float)(v <= 0.04045 ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4));
    } else {
      lut[i] = (float)i / 255.0f;
    }
  }
  RT_HIP(hipMemcpy(c->d_tex_lut, lut, sizeof lut, hipMemcpyHostToDevice));
  {
    // the bytes go up once and are decoded on the device; the staging copy is dropped again
    uchar4 *d_bytes = nullptr;
    RT_HIP(hipMalloc(&d_bytes, n * sizeof(uchar4)));
    hipError_t e = hipMemcpy(d_bytes, rgba8, n * sizeof(uchar4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rt::launch_decode_texels(d_bytes, c->d_tex_lut, c->d_texels, n, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_bytes);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_set_texture: %s", hipGetErrorString(e));
  }
  rt::Texture t{};
  t.texels = c->d_texels;
  t.width = desc->width;
  t.height = desc->height;
  t.fwidth = (float)desc->width;
  t.fheight = (float)desc->height;
  t.scale_u = desc->scale_u;
  t.scale_v = desc->scale_v;
  t.first_prim = desc->first_prim;
  t.n_prims = desc->n_prims;
  t.mapping = desc->mapping;
  if (!c->d_tex) RT_HIP(hipMalloc(&c->d_tex, sizeof t));
  RT_HIP(hipMemcpy(c->d_tex, &t, sizeof t, hipMemcpyHostToDevice));
  c->have_tex = true;
  return RT_OK;
}

static int rt_render_impl(rt_ctx *c, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride) {
  if (!c) return fail(RT_ERR_INVALID, "rt_render: null context");
  if (!c->have_params) return fail(RT_ERR_STATE, "rt_render: rt_set_params has not been called");
  if (!c->have_scene) return fail(RT_ERR_STATE, "rt_render: rt_set_spheres has not been called");
  if (n_frames == 0) return RT_OK;
  if (frame_stride == 0) return fail(RT_ERR_INVALID, "rt_render: frame_stride must be >= 1");
  RT_HIP(hipSetDevice(c->device));
  rt_ctx::EventPair ev;
  if (acquire_events(c, &ev) != RT_OK) return RT_ERR_DEVICE;

  rt::FrameSeq F{first_frame, n_frames, frame_stride};
  rt::Outputs O{};
  O.accum = c->d_accum;
  O.rgba8 = c->d_rgba8;
  O.ids = c->d_ids;
  O.rays = c->d_rays;
  O.ray_slots = c->d_rays + kRaySlotsOffset / sizeof(unsigned long long);
  O.linear_accum = (c->flags & RT_FLAG_ACCUM_LINEAR) ? 1u : 0u;
  O.tiles_x = c->tiles_x;
  O.n_tiles = c->n_tiles;
  if (c->use_tile_order) {
    O.tile_cost = c->d_tile_cost;
    O.tile_order = !c->order_valid ? nullptr
                   : (c->order_moved && c->use_bvh) ? c->d_tile_order + c->n_tiles : c->d_tile_order;
  }
  const rt::Scene S = scene_view(c);
  const bool strict = (c->flags & RT_FLAG_STRICT_MATH) != 0;
  const bool degenerate = c->P.n_spp < 1 || c->P.n_bounce < 1;

  hipError_t he = hipSuccess;
  if (c->order_pending) {   // the order this launch runs in (and the cleared costs it adds to) must be there
    he = hipStreamWaitEvent(c->stream, c->order_out, 0);
    c->order_pending = false;
  }
  if (he == hipSuccess) he = hipEventRecord(ev.begin, c->stream);
  uint32_t launches = 0;
  int rc = RT_OK;
  bool ordered = false;
  std::string err;
  const bool tile_kernels = c->mode == RT_MODE_MEGAKERNEL && !degenerate &&
                            !((c->flags & RT_FLAG_COUNT_TRAVERSAL) && c->use_bvh && !strict);
  if (he == hipSuccess && tile_kernels && c->use_tile_order && c->use_probe && !c->order_valid &&
      (uint64_t)c->P.n_spp * n_frames >= 32u) {
    // First launch of the context: no costs measured yet.  Two samples per pixel of the first
    // frame, traced without writing anything but the per-tile costs, are enough to tell the
    // tiles with long chains from the rest, and this launch is then already heaviest-first
    // (default scene, 1920x1080, 64 spp: 1.41 instead of 2.1 ms).  Launches of fewer than 32
    // samples per pixel are not worth the 2.
    rt::Params probe_params = c->P;
    probe_params.n_spp = std::min(probe_params.n_spp, 2);
    rt::Outputs probe_out = O;
    probe_out.tile_order = nullptr;
    probe_out.probe = 1u;
    he = rt::launch_megakernel(false, strict, c->use_bvh, probe_params, S, rt::FrameSeq{first_frame, 1u, 1u},
                               probe_out, c->stream);
    if (he == hipSuccess)
      he = rt::launch_tile_order(c->d_tile_cost, c->d_tile_order, c->n_tiles, c->stream,
                                 c->d_tile_order + c->n_tiles, c->tiles_x);
    if (he == hipSuccess) {
      c->order_valid = true;
      c->order_moved = false;
      O.tile_order = c->d_tile_order;
    }
  }
  if (he == hipSuccess) {
    if (c->mode == RT_MODE_WAVEFRONT && !degenerate) {
      if (c->wf.regroup && c->use_bvh)
        for (int k = 0; k < 3; ++k) {
          c->wf.regroup_lo[k] = c->bvh.scene_lo[k];
          c->wf.regroup_hi[k] = c->bvh.scene_hi[k];
        }
      rc = c->wf.render(strict, c->use_bvh, c->P, S, F, O, c->stream, &launches, &err);
    } else if ((c->flags & RT_FLAG_COUNT_TRAVERSAL) && c->use_bvh && !strict && !degenerate) {
      he = rt::launch_megakernel_counting(c->P, S, F, O, c->stream);
      launches = 1;
    } else {
      const bool nested = degenerate || c->mode == RT_MODE_MEGAKERNEL_NESTED;
      he = rt::launch_megakernel(nested, strict, c->use_bvh, c->P, S, F, O, c->stream);
      launches = 1;
      ordered = !nested && c->use_tile_order;
    }
  }
  if (he == hipSuccess && rc == RT_OK) he = hipEventRecord(ev.end, c->stream);
  if (he == hipSuccess && rc == RT_OK && ordered && ++c->launches_since_order >= c->order_interval) {
    // what the launches since the last refresh measured becomes the next launch's order (outside the
    // timed pair).  While every launch refreshes it -- the scene or the view has just changed, e.g. a
    // mesh re-uploaded per frame -- the refresh runs beside whatever the caller queues before the next
    // launch; a settled view refreshes every 16th launch, in line (four stream calls cost more there).
    const bool beside = c->order_interval == 1u;
    if (beside && !c->order_stream) {
      he = hipStreamCreateWithFlags(&c->order_stream, hipStreamNonBlocking);
      if (he == hipSuccess) he = hipEventCreateWithFlags(&c->order_in, hipEventDisableTiming);
      if (he == hipSuccess) he = hipEventCreateWithFlags(&c->order_out, hipEventDisableTiming);
    }
    if (beside) {
      if (he == hipSuccess) he = hipEventRecord(c->order_in, c->stream);
      if (he == hipSuccess) he = hipStreamWaitEvent(c->order_stream, c->order_in, 0);
    }
    if (he == hipSuccess)
      he = rt::launch_tile_order(c->d_tile_cost, c->d_tile_order, c->n_tiles, beside ? c->order_stream : c->stream,
                                 c->d_tile_order + c->n_tiles, c->tiles_x);
    if (beside && he == hipSuccess) he = hipEventRecord(c->order_out, c->order_stream);
    if (he == hipSuccess) {
      c->order_pending = beside;
      c->order_valid = true;
      c->order_moved = false;
    }
    c->launches_since_order = 0;
    if (c->order_interval < 16u) c->order_interval *= 2u;
  }
  if (he != hipSuccess || rc != RT_OK) {
    c->ev_free.push_back(ev);
    if (rc != RT_OK) return fail(rc, "rt_render: wavefront: %s", err.c_str());
    return fail(RT_ERR_DEVICE, "rt_render: %s", hipGetErrorString(he));
  }
  c->ev_pending.push_back(ev);
  c->frames += n_frames;
  c->last_launches = launches;
  return RT_OK;
}

int rt_sync(rt_ctx *c) {
  if (!c) return fail(RT_ERR_INVALID, "rt_sync: null context");
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipStreamSynchronize(c->stream));
  if (c->order_pending) RT_HIP(hipStreamSynchronize(c->order_stream));
  return RT_OK;
}

static int read_back(rt_ctx *c, const void *dptr, size_t expect, void *host, size_t bytes,
                     const char *who) {
  if (!c || !host) return fail(RT_ERR_INVALID, "%s: null argument", who);
  if (bytes != expect) return fail(RT_ERR_INVALID, "%s: %zu bytes given, %zu needed", who, bytes, expect);
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, c->stream));
  RT_HIP(hipStreamSynchronize(c->stream));
  return RT_OK;
}

int rt_read_rgba8(rt_ctx *c, void *host, size_t bytes) {
  return read_back(c, c ? c->d_rgba8 : nullptr, c ? c->n_pixels * 4 : 0, host, bytes, "rt_read_rgba8");
}
int rt_read_accum_f32(rt_ctx *c, void *host, size_t bytes) {
  return read_back(c, c ? c->d_accum : nullptr, c ? c->n_pixels * 16 : 0, host, bytes,
                   "rt_read_accum_f32");
}
int rt_read_primary_ids(rt_ctx *c, void *host, size_t bytes) {
  return read_back(c, c ? c->d_ids : nullptr, c ? c->n_pixels * 4 : 0, host, bytes,
                   "rt_read_primary_ids");
}

int rt_resolve(rt_ctx *c) {
  if (!c) return fail(RT_ERR_INVALID, "rt_resolve: null context");
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(rt::launch_resolve(c->d_accum, c->d_rgba8, (uint32_t)c->n_pixels,
                            (c->flags & RT_FLAG_ACCUM_LINEAR) ? 1u : 0u, c->stream));
  return RT_OK;
}

int rt_accum_device_ptr(rt_ctx *c, void **out_dptr, size_t *out_bytes) {
  if (!c || !out_dptr) return fail(RT_ERR_INVALID, "rt_accum_device_ptr: null argument");
  *out_dptr = c->d_accum;
  if (out_bytes) *out_bytes = c->n_pixels * sizeof(float4);
  return RT_OK;
}

static int rt_get_stats_impl(rt_ctx *c, rt_stats *out) {
  if (!c || !out) return fail(RT_ERR_INVALID, "rt_get_stats: null argument");
  RT_HIP(hipSetDevice(c->device));
  RT_HIP(hipStreamSynchronize(c->stream));
  if (collect_timing(c, 0) != RT_OK) return RT_ERR_DEVICE;
  // the segment count is rays[0] plus the spread counters the waves add to
  std::vector<unsigned long long> counters(kRayBytes / sizeof(unsigned long long));
  RT_HIP(hipMemcpy(counters.data(), c->d_rays, kRayBytes, hipMemcpyDeviceToHost));
  unsigned long long rays = counters[0];
  for (uint32_t i = 0; i < rt::kRaySlots; ++i)
    rays += counters[(kRaySlotsOffset + (size_t)i * rt::kRaySlotStride) / sizeof(unsigned long long)];
  std::memset(out, 0, sizeof *out);
  out->rays = rays;
  out->frames = c->frames;
  out->last_render_ms = c->last_ms;
  out->total_render_ms = c->total_ms;
  out->last_launches = c->last_launches;
  out->bvh_nodes = c->use_bvh ? c->bvh.n_nodes : 0u;
  out->last_build_ms = c->bvh.build_ms();
  out->bvh_height = c->use_bvh ? c->bvh.height : 0u;
  out->bvh_fallback = c->bvh_fallback ? 1u : 0u;
  out->reserved = 0u;
  out->bvh_nodes_visited = counters[1];
  out->bvh_spheres_tested = counters[2];
  return RT_OK;
}

int rt_debug_eval(int device, int op, const float *a, const float *b, float *out, size_t n) {
  if (!a || !out) return fail(RT_ERR_INVALID, "rt_debug_eval: null argument");
  if (op < 0 || op > 15) return fail(RT_ERR_INVALID, "rt_debug_eval: unknown op %d", op);
  if (n == 0) return RT_OK;
  RT_HIP(hipSetDevice(device));
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  hipError_t e = hipMalloc(&da, n * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&dout, n * sizeof(float));
  if (e == hipSuccess && b) e = hipMalloc(&db, n * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(da, a, n * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess && b) e = hipMemcpy(db, b, n * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = rt::launch_debug_eval(op, da, db, dout, n, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost);
  hipFree(da);
  hipFree(dout);
  if (db) hipFree(db);
  if (e != hipSuccess) return fail(RT_ERR_DEVICE, "rt_debug_eval: %s", hipGetErrorString(e));
  return RT_OK;
}

// ---- rt_render_static: the entry point that allocates nothing ---------------------------------
// Scratch of rt_render_static: the repacked scene and the tile cost / order tables that
// rt_create and rt_set_spheres otherwise allocate.  One copy per GPU, held until the process
// ends.  Host code never uses the array's name as an address (it is a host-side stand-in):
// the device address comes from hipGetSymbolAddress, once per device.
namespace {
constexpr uint32_t kStaticGeomEntries = RT_STATIC_MAX_SPHERES + 8u;  // padded to whole groups of four, + one group
constexpr size_t kStaticGeomBytes = (size_t)kStaticGeomEntries * sizeof(float4);
constexpr size_t kStaticMaterialBytes = ((size_t)RT_STATIC_MAX_SPHERES + 1u) * 3u * sizeof(float4);
constexpr size_t kStaticTileBytes = (size_t)RT_STATIC_MAX_TILES * sizeof(uint32_t);
constexpr size_t kStaticOffMaterial = kStaticGeomBytes;
constexpr size_t kStaticOffCost = (kStaticOffMaterial + kStaticMaterialBytes + 255u) / 256u * 256u;
constexpr size_t kStaticOffOrder = kStaticOffCost + kStaticTileBytes;
constexpr size_t kStaticOffRays = kStaticOffOrder + kStaticTileBytes;
constexpr size_t kStaticOffRaySlots = kStaticOffRays + 256u;
constexpr size_t kStaticBytes = kStaticOffRaySlots + (size_t)rt::kRaySlots * rt::kRaySlotStride;
}  // namespace

static __device__ __align__(256) unsigned char scratch[kStaticBytes];

namespace {
constexpr int kMaxDevices = 64;
unsigned char *g_scratch_ptr[kMaxDevices];  // per device: where ITS copy of `scratch` lives
std::mutex g_scratch_mutex;

// device address of this device's scratch array (the device must be current)
int static_scratch(int device, unsigned char **out) {
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void *p = nullptr;
    RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(RT_ERR_DEVICE, "rt_render_static: no address for the scratch array on device %d", device);
    g_scratch_ptr[device] = (unsigned char *)p;
  }
  *out = g_scratch_ptr[device];
  return RT_OK;
}
}  // namespace

int rt_render_static(const rt_static_job *job) {
  if (!job) return fail(RT_ERR_INVALID, "rt_render_static: null job");
  if (job->struct_size != sizeof(rt_static_job))
    return fail(RT_ERR_INVALID, "rt_render_static: struct_size %u, expected %zu", job->struct_size,
                sizeof(rt_static_job));
  if (!job->ubo272 || !job->d_accum || !job->d_rgba8 || !job->d_ids || (job->n_spheres && !job->d_spheres64))
    return fail(RT_ERR_INVALID, "rt_render_static: null pointer");
  if (job->width == 0 || job->height == 0 || job->width > 65536u || job->height > 65536u)
    return fail(RT_ERR_INVALID, "rt_render_static: bad image size %ux%u", job->width, job->height);
  const uint32_t tiles_x = (job->width + 15u) / 16u;
  const uint64_t n_tiles64 = (uint64_t)tiles_x * ((job->height + 15u) / 16u);
  if (n_tiles64 > RT_STATIC_MAX_TILES)
    return fail(RT_ERR_INVALID, "rt_render_static: %llu tiles of 16x16 pixels, at most %u fit the static tables",
                (unsigned long long)n_tiles64, (unsigned)RT_STATIC_MAX_TILES);
  if (job->n_spheres > RT_STATIC_MAX_SPHERES)
    return fail(RT_ERR_INVALID, "rt_render_static: %u spheres, at most %u fit the static scene", job->n_spheres,
                (unsigned)RT_STATIC_MAX_SPHERES);
  if (job->flags & ~(uint32_t)(RT_FLAG_STRICT_MATH | RT_FLAG_ACCUM_LINEAR))
    return fail(RT_ERR_INVALID, "rt_render_static: flags 0x%x (strict math and linear accumulation only)", job->flags);
  if (job->frame_stride == 0) return fail(RT_ERR_INVALID, "rt_render_static: frame_stride must be >= 1");
  if (job->reserved != 0) return fail(RT_ERR_INVALID, "rt_render_static: reserved must be 0");
  if (job->device < 0 || job->device >= kMaxDevices)
    return fail(RT_ERR_INVALID, "rt_render_static: device %d", job->device);
  if (job->n_frames == 0) return RT_OK;
  RT_HIP(hipSetDevice(job->device));
  unsigned char *base = nullptr;
  int rc = static_scratch(job->device, &base);
  if (rc != RT_OK) return rc;
  hipStream_t stream = (hipStream_t)job->stream;
  const uint32_t n_tiles = (uint32_t)n_tiles64;
  float4 *geom = (float4 *)base;
  float4 *material = (float4 *)(base + kStaticOffMaterial);
  uint32_t *tile_cost = (uint32_t *)(base + kStaticOffCost);
  uint32_t *tile_order = (uint32_t *)(base + kStaticOffOrder);
  unsigned long long *rays = job->d_rays ? (unsigned long long *)job->d_rays
                                         : (unsigned long long *)(base + kStaticOffRays);

  rt::Params P{};
  fill_params(job->ubo272, job->width, job->height, P);
  // the array keeps the last call's contents: the scene is rewritten (padding included) and the
  // costs restart from zero
  const uint32_t n_geom = (job->n_spheres + 3u) / 4u * 4u + 4u;
  RT_HIP(rt::launch_repack_spheres(job->d_spheres64, job->n_spheres, n_geom, geom, material, stream));
  RT_HIP(hipMemsetAsync(tile_cost, 0, (size_t)n_tiles * sizeof(uint32_t), stream));

  rt::Scene S{};
  S.geom = geom;
  S.material = material;
  S.n_spheres = job->n_spheres;
  rt::FrameSeq F{job->first_frame, job->n_frames, job->frame_stride};
  rt::Outputs O{};
  O.accum = (float4 *)job->d_accum;
  O.rgba8 = (uchar4 *)job->d_rgba8;
  O.ids = (int32_t *)job->d_ids;
  O.rays = rays;
  // the waves count into spread counters (zero at load, cleared by every fold), which the end
  // of the call adds into the caller's word (or the array's own)
  O.ray_slots = (unsigned long long *)(base + kStaticOffRaySlots);
  O.linear_accum = (job->flags & RT_FLAG_ACCUM_LINEAR) ? 1u : 0u;
  O.tiles_x = tiles_x;
  O.n_tiles = n_tiles;
  O.tile_cost = tile_cost;
  const bool strict = (job->flags & RT_FLAG_STRICT_MATH) != 0;
  const bool degenerate = P.n_spp < 1 || P.n_bounce < 1;
  if (!degenerate && (uint64_t)P.n_spp * job->n_frames >= 32u) {
    // as rt_render's first launch on a context: a two-sample cost probe, then heaviest tiles first
    rt::Params probe_params = P;
    probe_params.n_spp = std::min(probe_params.n_spp, 2);
    rt::Outputs probe_out = O;
    probe_out.probe = 1u;
    RT_HIP(rt::launch_megakernel(false, strict, false, probe_params, S, rt::FrameSeq{job->first_frame, 1u, 1u},
                                 probe_out, stream));
    RT_HIP(rt::launch_tile_order(tile_cost, tile_order, n_tiles, stream));
    O.tile_order = tile_order;
  }
  RT_HIP(rt::launch_megakernel(degenerate, strict, false, P, S, F, O, stream));
  // (always, so that no call leaves counts behind for the next one)
  RT_HIP(rt::launch_fold_rays(O.ray_slots, rays, stream));
  return RT_OK;
}

int rt_create(const rt_config *cfg, rt_ctx **out_ctx) {
  return guarded("rt_create", [&] { return rt_create_impl(cfg, out_ctx); });
}

int rt_set_params(rt_ctx *c, const void *ubo272) {
  return guarded("rt_set_params", [&] { return rt_set_params_impl(c, ubo272); });
}

int rt_set_spheres(rt_ctx *c, const void *spheres64, uint32_t count) {
  return guarded("rt_set_spheres", [&] { return rt_set_spheres_impl(c, spheres64, count); });
}

int rt_set_mesh(rt_ctx *c, const float *vertices, uint32_t n_vertices, const uint32_t *indices,
                uint32_t n_triangles, const rt_material *material) {
  return guarded("rt_set_mesh", [&] { return set_mesh(c, vertices, n_vertices, indices, n_triangles, material, false); });
}

int rt_set_mesh_device(rt_ctx *c, const float *d_vertices, uint32_t n_vertices, const uint32_t *d_indices,
                       uint32_t n_triangles, const rt_material *material) {
  return guarded("rt_set_mesh_device",
                 [&] { return set_mesh(c, d_vertices, n_vertices, d_indices, n_triangles, material, true); });
}

int rt_set_dielectric(rt_ctx *c, const float *ior, uint32_t count) {
  return guarded("rt_set_dielectric", [&] { return rt_set_dielectric_impl(c, ior, count); });
}

int rt_set_texture(rt_ctx *c, const rt_texture_desc *desc, const void *rgba8) {
  return guarded("rt_set_texture", [&] { return rt_set_texture_impl(c, desc, rgba8); });
}

int rt_render(rt_ctx *c, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride) {
  return guarded("rt_render", [&] { return rt_render_impl(c, first_frame, n_frames, frame_stride); });
}

int rt_get_stats(rt_ctx *c, rt_stats *out) {
  return guarded("rt_get_stats", [&] { return rt_get_stats_impl(c, out); });
}

}  // extern "C"
