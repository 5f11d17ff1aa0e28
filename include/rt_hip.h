/*
 * rt_hip.h -- C ABI of the MI355X (gfx950) path-tracing module, librt_hip.so.
 *
 * novadinn/ray-tracer has no plugin or FFI interface: its compute shader is reached
 * through Vulkan calls written inline in main().  Each entry point below stands
 * exactly where one group of those calls stands, takes the same bytes the
 * reference hands to Vulkan (rt_contract.h), and is cited against the call site it
 * replaces.  Plain pointers and sizes only; every call returns 0 on success or a
 * negative rt_status and never aborts; rt_last_error() gives the text for the
 * calling thread.  A context is used from one host thread at a time.  Inputs are
 * copied during the call (the reference frees its staging buffer right after the
 * upload, src/vulkan_buffer.cpp:66-89); outputs are copied into caller memory.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "rt_contract.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID = -1,   /* bad argument */
  RT_ERR_DEVICE = -2,    /* a HIP call failed */
  RT_ERR_STATE = -3,     /* call order: e.g. render before set_params/set_spheres */
  RT_ERR_NOMEM = -4
} rt_status;

/* kernel organisation for one frame batch */
typedef enum rt_mode {
  RT_MODE_MEGAKERNEL = 0, /* one launch, one lane per pixel, paths regenerated in-lane */
  RT_MODE_WAVEFRONT = 1,  /* raygen / extend / shade stages over compacted queues */
  RT_MODE_MEGAKERNEL_NESTED = 2 /* the shader's own loop nest, kept as the structural baseline */
} rt_mode;

enum {
  /* sin/cos/log/pow from fixed fp32 polynomial kernels instead of the hardware
   * approximations: every output then equals the CPU restatement bit for bit */
  RT_FLAG_STRICT_MATH = 1u << 0,
  /* accumulate the per-frame linear mean instead of the gamma-encoded colour */
  RT_FLAG_ACCUM_LINEAR = 1u << 1,
  /* closest-hit search: brute force over all spheres (the shader's loop) even for
   * large scenes; default is the LBVH once the scene has more than 16 spheres */
  RT_FLAG_FORCE_BRUTE = 1u << 2,
  RT_FLAG_FORCE_BVH = 1u << 3,
  /* measurement aid: megakernel + LBVH + fast math only; also counts the LBVH nodes
   * fetched and the spheres tested (rt_stats.bvh_nodes_visited / bvh_spheres_tested).
   * Slower.  Hit ids are unchanged; the counting kernel takes exact sphere distances where the
   * default kernel takes them from the hardware sqrt/rcp units, so colours agree with a normal
   * launch to fast-math tolerance, not bit for bit. */
  RT_FLAG_COUNT_TRAVERSAL = 1u << 4
};

typedef struct rt_config {
  uint32_t struct_size;   /* sizeof(rt_config) */
  int32_t device;         /* HIP device ordinal */
  uint32_t width, height; /* image size in pixels */
  uint32_t mode;          /* rt_mode */
  uint32_t flags;         /* RT_FLAG_* */
  void *stream;           /* hipStream_t to launch on, or NULL for the null stream */
  void *accum_buffer;     /* optional caller-owned device buffer, width*height*16 bytes,
                             used instead of an internal one (lets the caller hand the
                             accumulation image to a collective without a copy) */
} rt_config;

typedef struct rt_stats {
  uint64_t rays;            /* path segments traced since rt_clear (one per closest-hit search) */
  uint64_t frames;          /* frames rendered since rt_clear */
  float last_render_ms;     /* device time of the last rt_render, HIP events on the stream */
  float total_render_ms;    /* sum of the above since rt_clear */
  uint32_t last_launches;   /* kernel launches issued by the last rt_render */
  uint32_t bvh_nodes;       /* internal nodes of the current LBVH, 0 when brute force */
  float last_build_ms;      /* device time of the last LBVH build */
  uint32_t bvh_height;      /* longest root-to-leaf path of the LBVH, in edges */
  uint64_t bvh_nodes_visited;   /* with RT_FLAG_COUNT_TRAVERSAL: 64-byte node fetches */
  uint64_t bvh_spheres_tested;  /* with RT_FLAG_COUNT_TRAVERSAL: 16-byte sphere fetches */
  uint32_t bvh_fallback;    /* 1: an LBVH was wanted and built but is taller than the traversal stack
                               (64 levels), so the scene is traced by brute force -- correct, slow;
                               bvh_nodes is 0 then as it is under RT_FLAG_FORCE_BRUTE */
  uint32_t reserved;
} rt_stats;

/* Replaces createTexture(RGBA8_UNORM, W, H, STORAGE) + zero fill, main.cpp:393-448,
 * and createBuffer(UBO)/createBuffer(SSBO), main.cpp:487-495,540-548: allocates the
 * 8-bit image (zeroed), the float accumulation image and the parameter block. */
int rt_create(const rt_config *cfg, rt_ctx **out_ctx);

/* Replaces destroyTexture/destroyBuffer, main.cpp:1044-1051. */
int rt_destroy(rt_ctx *ctx);

/* Replaces loadBufferData(&compute_ubo_buffer, ...), main.cpp:703
 * (vulkan_buffer.cpp:57-64): takes the shader's 272-byte std140 block as is.
 * ubo->frame.x is ignored; rt_render supplies the frame index. */
int rt_set_params(rt_ctx *ctx, const void *ubo272);

/* Replaces loadBufferDataStaging(&compute_ssbo, ...), main.cpp:549-554: count records
 * of 64 bytes (rt_sphere).  Rebuilds the acceleration structure when one is used. */
int rt_set_spheres(rt_ctx *ctx, const void *spheres64, uint32_t count);

/* No counterpart in the reference, which traces spheres only (assimp is linked but never
 * called, CMakeLists.txt:12): an indexed triangle mesh with ONE material, traced together
 * with the spheres.  vertices = n_vertices * xyz (12-byte stride), indices = 3 per
 * triangle.  Primitive ids of triangles follow the spheres: n_spheres + triangle index.
 * Both faces are hit; the shading normal is the geometric normal turned against the ray.
 * A triangle without area (a repeated vertex, collinear vertices whose edge cross product is
 * exactly zero) keeps its id but is never hit.
 * n_triangles = 0 removes the mesh.  Rebuilds the acceleration structure.
 * Coordinate range: the triangle rule compares products of four coordinate differences
 * (det^2 against 2^-40 (e1.e1)(p.p) and (d.d)(n.n)), which leave the float range for edges longer
 * than ~4e9 or shorter than ~1e-10 world units; such triangles are never hit.  Keep meshes within
 * 1e-9 .. 1e9 units.
 * Device memory: 48 bytes per triangle, plus as much again while a replaced mesh of the SAME
 * triangle count is kept for the next upload (a mesh re-uploaded every frame then allocates
 * nothing); a replaced mesh of another size, or of a mesh that is removed, is freed at once. */
int rt_set_mesh(rt_ctx *ctx, const float *vertices, uint32_t n_vertices, const uint32_t *indices,
                uint32_t n_triangles, const rt_material *material);

/* The same with vertices and indices already resident in device memory (same layouts), for a
 * mesh that another kernel animates every frame: nothing crosses the host-device link.  The
 * arrays are read during the call, by kernels on the context's stream (whatever writes them must be
 * ordered before it: the same stream, or an event), and the call returns after the acceleration
 * structure is rebuilt; material is host memory.  A mesh re-uploaded with the triangle count of the
 * one before it costs two host round trips (validity and bounds of the upload; height of the
 * tree) and does not wait for launches already queued on the stream. */
int rt_set_mesh_device(rt_ctx *ctx, const float *d_vertices, uint32_t n_vertices,
                       const uint32_t *d_indices, uint32_t n_triangles, const rt_material *material);

/* Stands in for the sampled image the reference declares but never binds: loadTexture
 * (main.cpp:1384-1409, no call site) would create an RGBA8 image with a LINEAR / REPEAT,
 * single-level sampler (vulkan_texture.cpp:77-95), and the compute shader has no sampler
 * binding.  Here ONE albedo texture can be bound to a range of primitive ids: on a hit of
 * first_prim <= id < first_prim + n_prims the material's colour.rgb is replaced by the
 * bilinear sample (colour.a, emission and specular stay the material's).
 *   rgba8     width * height texels, 4 bytes each, row 0 at v = 0 (copied)
 *   flags     RT_TEX_SRGB: bytes are sRGB-encoded and are linearised before filtering
 *   mapping   RT_TEX_MAP_PLANAR_XZ: (u, v) = (point.x * scale_u, point.z * scale_v)
 *             RT_TEX_MAP_OCTAHEDRAL: the unit normal folded onto a square (spheres),
 *             (u, v) = (a * scale_u, b * scale_v) with a, b in [0, 1]
 * One unit of u or v is one repeat of the image.  desc == NULL removes the texture.
 * The texels are kept on the device decoded to linear floats, 16 bytes each (a 32768 x 32768
 * image takes 16 GiB; RT_ERR_DEVICE if that cannot be allocated). */
#define RT_TEX_SRGB 1u
#define RT_TEX_MAP_PLANAR_XZ 0u
#define RT_TEX_MAP_OCTAHEDRAL 1u
typedef struct rt_texture_desc {
  uint32_t struct_size;          /* sizeof(rt_texture_desc) */
  uint32_t width, height;        /* 1 .. 32768 */
  uint32_t flags;
  uint32_t mapping;
  float scale_u, scale_v;
  uint32_t first_prim, n_prims;
} rt_texture_desc;
int rt_set_texture(rt_ctx *ctx, const rt_texture_desc *desc, const void *rgba8);

/* No counterpart in the reference, whose material (ray_tracing.comp:13-20) has no refraction
 * term and whose sphere test returns the near root only (:157-161), so a ray inside a sphere
 * never hits it: dielectric spheres.  ior = count floats, count == the number of spheres last
 * given to rt_set_spheres; ior[i] > 0 makes sphere i a dielectric with that index of
 * refraction, anything else leaves it with the reference's material model.  A hit on a
 * dielectric sphere draws ONE random number and reflects or refracts (Snell's law, Schlick's
 * reflectance; total internal reflection when there is no refracted ray); the throughput is
 * multiplied by colour.rgb either way and emission is added as for any other hit.  A ray that
 * was refracted into a sphere hits THAT sphere at its far root until it leaves it again; all
 * other spheres keep the reference's near-root test.  The 48-byte material has no spare
 * field, hence this side table.  ior == NULL or count == 0 removes it; rt_set_spheres removes
 * it too (the indices belong to the sphere set).  A bound texture is not sampled on dielectric
 * spheres. */
int rt_set_dielectric(rt_ctx *ctx, const float *ior, uint32_t count);

/* Replaces bind pipeline + descriptor sets + vkCmdDispatch(W/16,H/16,1) + submit,
 * main.cpp:819-868, for n_frames dispatches with ubo.frame.x = first_frame,
 * first_frame + frame_stride, ...  (main.cpp:1028 increments it by one per
 * dispatch).  Every pixel is shaded (ceil-div grid with a bounds guard).
 * Megakernel modes are asynchronous: the call returns after enqueueing on the context's
 * stream.  RT_MODE_WAVEFRONT reads the live-path count back every few iterations to decide
 * how many more stages to launch, so the call blocks for most of the frame. */
int rt_render(rt_ctx *ctx, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride);

/* Replaces vkWaitForFences on the compute fence, main.cpp:872-874. */
int rt_sync(rt_ctx *ctx);

/* The reference restarts accumulation by resetting ubo.frame.x to 0 on any edit,
 * main.cpp:1028-1031, which makes the next dispatch overwrite the image; this
 * zeroes image, accumulation buffer, id map and statistics explicitly. */
int rt_clear(rt_ctx *ctx);

/* Replaces the blit's read of the result image, main.cpp:951-957: the 8-bit image as
 * the shader's own progressive blend leaves it (width*height*4 bytes, row 0 first;
 * the reference shows it bottom-up, main.cpp:934-940). */
int rt_read_rgba8(rt_ctx *ctx, void *host, size_t bytes);

/* No counterpart in the reference (it accumulates only in the 8-bit image): the
 * float accumulation image, width*height float4 = (sum over frames of the frame
 * colour, number of frames). */
int rt_read_accum_f32(rt_ctx *ctx, void *host, size_t bytes);

/* No counterpart in the reference: per pixel, the index of the sphere hit by the
 * first segment of sample 0 of the last frame rendered, -1 for a miss (int32). */
int rt_read_primary_ids(rt_ctx *ctx, void *host, size_t bytes);

/* Rewrites the 8-bit image from the float accumulation image (mean over frames,
 * gamma applied if the sums are linear, then the rgba8 store conversion).  Used
 * after a multi-GPU sum, where no single device ran the progressive blend. */
int rt_resolve(rt_ctx *ctx);

/* Device address and size of the accumulation image, for collectives. */
int rt_accum_device_ptr(rt_ctx *ctx, void **out_dptr, size_t *out_bytes);

/* Waits for the stream, then reports counters. */
int rt_get_stats(rt_ctx *ctx, rt_stats *out);

/* No counterpart in the reference: ONE entry point that allocates nothing, for callers that
 * cannot call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid allocation.  It is rt_render for a brute-force sphere scene without a context: one
 * call repacks the scene, measures tile costs with a two-sample probe and renders n_frames
 * frames with the megakernel, all enqueued on `stream`, and issues no hipMalloc / hipFree, no
 * events and no host-device copies.  What rt_create and rt_set_spheres otherwise allocate (the
 * repacked scene, the tile cost and order tables) lives in a static device array of about
 * 1.2 MB that every GPU the library runs on holds until the process ends.
 *   ubo272       HOST: the shader's 272-byte uniform block (frame.x ignored), read during the call
 *   d_spheres64  DEVICE: n_spheres records of 64 bytes, read by the launches
 *   d_accum      DEVICE: width*height float4, read-modify-written like rt_render's accumulation image
 *   d_rgba8      DEVICE: width*height uchar4, the 8-bit progressive image, read-modify-written
 *   d_ids        DEVICE: width*height int32, primary-hit ids, written
 *   d_rays       DEVICE or NULL: one uint64, += path segments traced
 *   flags        RT_FLAG_STRICT_MATH and RT_FLAG_ACCUM_LINEAR only
 * Limits (RT_ERR_INVALID beyond them): n_spheres <= RT_STATIC_MAX_SPHERES; at most
 * RT_STATIC_MAX_TILES tiles of 16x16 pixels (a 7680x4320 image has 129,600).  Results equal
 * rt_render's on a context created with RT_FLAG_FORCE_BRUTE, bit for bit.  Every call on a
 * GPU uses that GPU's one array: calls whose launches overlap on one GPU (different streams or
 * host threads) race; calls on one stream are ordered and safe.
 * Under stream capture: the FIRST call on each device looks the array's address up
 * (hipGetSymbolAddress, which loads the code object) and must therefore happen outside capture --
 * one ordinary call per device warms it up; later calls only enqueue kernels and one memset on
 * `stream` and can be captured into a graph and replayed (ubo272 is read at capture time; the
 * device pointers are baked into the graph).  Every call makes `device` the calling thread's
 * current device (hipSetDevice) and leaves it so.  `reserved` must be 0. */
#define RT_STATIC_MAX_SPHERES 1024u
#define RT_STATIC_MAX_TILES 139264u
typedef struct rt_static_job {
  uint32_t struct_size;   /* sizeof(rt_static_job) */
  int32_t device;
  void *stream;           /* hipStream_t, or NULL for the null stream */
  uint32_t width, height;
  uint32_t flags;
  uint32_t n_spheres;
  const void *ubo272;
  const void *d_spheres64;
  uint32_t first_frame, n_frames, frame_stride;
  uint32_t reserved;      /* 0 */
  void *d_accum, *d_rgba8, *d_ids, *d_rays;
} rt_static_job;
int rt_render_static(const rt_static_job *job);

/* Test hook, no counterpart in the reference: evaluates one device math routine
 * elementwise on host arrays so tests can compare it bit for bit with the CPU.
 * op: 0 sqrt(a) 1 a/b 2 sin(a) 3 cos(a) 4 log(a) 5 log2(a) 6 exp2(a) 7 pow(a,b)
 * (2..7 are the strict-math polynomial kernels), 8..13 the fast counterparts of
 * 2..7 (hardware approximations). */
int rt_debug_eval(int device, int op, const float *a, const float *b, float *out, size_t n);

const char *rt_last_error(void);
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif
