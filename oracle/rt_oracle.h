/*
 * rt_oracle.h -- CPU oracle for the per-pixel path-tracing loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (ray_tracer_amd/, include/)
 * may include, link or call this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * PARITY UNPINNED: the reference (novadinn/ray-tracer) ships no tests, golden
 * vectors or fixtures for this path, and its shader cannot be compiled or run in
 * this image (no glslang, no Vulkan loader, no lavapipe).  This file is a
 * restatement of assets/shaders/ray_tracing.comp and src/camera.cpp written from
 * the source text; it is pinned only by closed-form known answers that follow
 * from that text (PCG streams, primary-hit census, centre-pixel distance).
 *
 * Numerics policy.  GLSL leaves FMA contraction and the precision of
 * inverse/normalize/sin/cos/log/pow to the driver, so the oracle fixes ONE valid
 * instance of the shader and documents it:
 *   - add/mul/fma exactly as written here (compiled with -ffp-contract=off, every
 *     fused operation spelled fmaf);
 *   - sqrt and division IEEE-754 correctly rounded;
 *   - RTO_MATH_PORTABLE: sin/cos/log/log2/exp2 are fixed fp32 polynomial kernels
 *     (this file), reproducible bit for bit on any IEEE machine, GPU included;
 *   - RTO_MATH_LIBM: the same program with the host libm's sinf/cosf/logf/powf,
 *     used to show the polynomial kernels do not bias the image.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RTO_MATH_PORTABLE = 0, RTO_MATH_LIBM = 1 };
enum { RTO_ACCUM_GAMMA = 0, RTO_ACCUM_LINEAR = 1 };

/* --- RNG (ray_tracing.comp:112-119) --- */
uint32_t rto_next_random(uint32_t *state);
float rto_random_value(uint32_t *state);

/* --- portable transcendental kernels (exported so tests can bound their error) --- */
float rto_pm_sin(float x);
float rto_pm_cos(float x);
float rto_pm_log(float x);
float rto_pm_log2(float x);
float rto_pm_exp2(float x);
float rto_pm_pow(float x, float y);

/* --- camera.cpp:5-50 restated without glm; matrices are column-major float[16] --- */
typedef struct rto_camera {
  float fov, near_, far_;
  float pitch, yaw;
  float aspect_ratio;
  float distance;
  float viewport_width, viewport_height;
} rto_camera;

void rto_camera_default(rto_camera *cam);                       /* camera.h:7-13 */
void rto_camera_create(float fov, float aspect, float near_, float far_,
                       rto_camera *cam);                        /* camera.cpp:5-13 */
void rto_camera_rotate(rto_camera *cam, float dx, float dy);    /* camera.cpp:15-19 */
void rto_camera_projection(rto_camera *cam, float out16[16]);   /* camera.cpp:21-25 */
void rto_camera_view(const rto_camera *cam, float out16[16]);   /* camera.cpp:27-34 */
void rto_camera_orientation(const rto_camera *cam, float out_wxyz[4]); /* camera.cpp:48-50 */
void rto_camera_up(const rto_camera *cam, float out3[3]);       /* camera.cpp:36-38 */
void rto_camera_right(const rto_camera *cam, float out3[3]);    /* camera.cpp:40-42 */
void rto_camera_forward(const rto_camera *cam, float out3[3]);  /* camera.cpp:44-46 */

/* general 4x4 inverse (GLSL inverse(), ray_tracing.comp:230,233), column-major */
void rto_mat4_inverse(const float m[16], float out[16]);

/* default UBO (main.cpp:626-636,697-701) and default scene (main.cpp:517-538) */
void rto_default_ubo(const rto_camera *cam_or_null, void *ubo272);
uint32_t rto_default_spheres(void *spheres64x3);

/*
 * Render frames first_frame, first_frame+stride, ... (n_frames of them) of a
 * W x H image exactly as ray_tracing.comp:74-110 would, one invocation per pixel
 * (every pixel, i.e. a ceil-div dispatch with a bounds guard).
 *
 *   ubo272     the shader's 272-byte std140 uniform block; ubo.frame.x is
 *              overwritten per frame with the frame index
 *   spheres64  n_spheres records of 64 bytes (std140 Sphere)
 *   accum      optional W*H*4 floats, ADDED to: rgb += per-frame colour (gamma
 *              encoded for RTO_ACCUM_GAMMA, linear mean for RTO_ACCUM_LINEAR),
 *              a += 1 per frame
 *   rgba8      optional W*H*4 bytes, read-modify-written with the shader's own
 *              progressive blend in an 8-bit UNORM image (ray_tracing.comp:103-109)
 *   ids        optional W*H int32: index of the sphere hit by the first path
 *              segment of sample 0 of the last frame rendered, -1 on a miss
 *   rays       optional: number of calculateRayCollision calls is ADDED to *rays
 *   threads    host threads (rows are handed out dynamically); <=0 means 1
 * Returns 0, or -1 on bad arguments.
 */
int rto_render(const void *ubo272, const void *spheres64, uint32_t n_spheres,
               uint32_t width, uint32_t height, uint32_t first_frame,
               uint32_t n_frames, uint32_t frame_stride, int math_mode,
               int accum_mode, float *accum, uint8_t *rgba8, int32_t *ids,
               uint64_t *rays, int threads);

/* Triangle mesh (EXTENSION: the reference traces spheres only; BASELINE.json asks for
 * triangles).  One material for the whole mesh.  Primitive ids of triangles follow the
 * spheres: n_spheres + triangle index.  Intersection rule: see ray_triangle() in
 * rt_oracle.c; a triangle without area (e1 x e2 == 0) is never hit. */
typedef struct rto_mesh {
  const float *vertices;      /* xyz, 12-byte stride */
  uint32_t n_vertices;
  const uint32_t *indices;    /* three per triangle */
  uint32_t n_triangles;
  const float *material;      /* 12 floats: colour, emissionColour, specularColour */
} rto_mesh;

int rto_render_mesh(const void *ubo272, const void *spheres64, uint32_t n_spheres,
                    const rto_mesh *mesh, uint32_t width, uint32_t height, uint32_t first_frame,
                    uint32_t n_frames, uint32_t frame_stride, int math_mode, int accum_mode,
                    float *accum, uint8_t *rgba8, int32_t *ids, uint64_t *rays, int threads);

/* Albedo texture (EXTENSION: the reference's shader samples no texture; its sampler state is
 * vulkan_texture.cpp:77-95 -- LINEAR filter, REPEAT addressing, one mip level).  On a hit of
 * primitive first_prim <= id < first_prim + n_prims the material's colour.rgb is replaced by
 * the bilinear sample; see texture_sample() in rt_oracle.c for the arithmetic. */
enum { RTO_TEX_MAP_PLANAR_XZ = 0, RTO_TEX_MAP_OCTAHEDRAL = 1 };
typedef struct rto_texture {
  const uint8_t *rgba8;       /* width*height texels, row 0 = v 0 */
  uint32_t width, height;
  int srgb;                   /* 1: texel bytes are sRGB-encoded, 0: linear (byte/255) */
  int mapping;                /* RTO_TEX_MAP_* */
  float scale_u, scale_v;
  uint32_t first_prim, n_prims;
} rto_texture;

int rto_render_scene(const void *ubo272, const void *spheres64, uint32_t n_spheres,
                     const rto_mesh *mesh, const rto_texture *texture, uint32_t width,
                     uint32_t height, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride,
                     int math_mode, int accum_mode, float *accum, uint8_t *rgba8, int32_t *ids,
                     uint64_t *rays, int threads);

/* Everything above in one call, plus two more options.
 *   ior      EXTENSION (the reference has no refraction, ray_tracing.comp:13-20,157-161):
 *            optional array of n_spheres floats; ior[i] > 0 makes sphere i a dielectric with
 *            that index of refraction.  Rule: dielectric_scatter() and closest_hit() in
 *            rt_oracle.c (Snell + Schlick, one random draw; a ray refracted into a sphere hits
 *            that sphere at its far root).
 *   pixels   optional list of n_pixels pixel indices (y * width + x).  Only those pixels are
 *            shaded -- each exactly as in a whole-image render, since pixels are independent
 *            (ray_tracing.comp:80-82) -- and accum / rgba8 / ids are then COMPACT arrays
 *            with one entry per listed pixel, in list order.  Lets a test check a sample of a
 *            scene whose full brute-force render would take hours on a CPU. */
typedef struct rto_job {
  const void *ubo272;
  const void *spheres64;
  uint32_t n_spheres;
  const rto_mesh *mesh;
  const rto_texture *texture;
  const float *ior;
  uint32_t width, height, first_frame, n_frames, frame_stride;
  int math_mode, accum_mode;
  const uint32_t *pixels;
  uint32_t n_pixels;
  float *accum;
  uint8_t *rgba8;
  int32_t *ids;
  uint64_t *rays;
  int threads;
} rto_job;
int rto_render_job(const rto_job *job);

/* Debugging aid: while records8 is non-NULL every closest-hit search of a render appends 8 floats
 * (origin, direction, hit index or -1, distance) up to max_records; for a one-pixel list on one
 * thread.  rto_debug_log(NULL, 0) switches it off; rto_debug_log_count() = records written. */
void rto_debug_log(float *records8, uint32_t max_records);
uint32_t rto_debug_log_count(void);

/* one bilinear sample of the texture at (u, v), for unit tests of the sampling rule */
void rto_texture_sample(const rto_texture *texture, float u, float v, float out_rgb[3]);

/* one dielectric scatter (see dielectric_scatter() in rt_oracle.c), for unit tests of the rule:
 * the outward unit normal of the sphere at the hit, exiting != 0 when the ray travels inside
 * it; advances *rng by one draw */
void rto_dielectric_scatter(const float dir[3], const float normal[3], float ior, int exiting,
                            uint32_t *rng, float out_dir[3], int *out_refracted);

/* byte -> linear float of the sRGB transfer function (IEC 61966-2-1), evaluated in double */
float rto_srgb_to_linear(uint8_t v);

/* accum (rgb sum, a = frame count) -> rgba8 with the image-store conversion */
void rto_resolve_rgba8(const float *accum, uint32_t n_pixels, int accum_mode,
                       uint8_t *rgba8);

/* float -> UNORM8 as imageStore on an rgba8 image does it */
uint8_t rto_unorm8(float v);

#ifdef __cplusplus
}
#endif
#endif
