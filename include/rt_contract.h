/*
 * rt_contract.h -- the data contract of the path-tracing hot path, byte for byte
 * what novadinn/ray-tracer's host writes and its compute shader reads.
 *
 * Plain C, no dependencies.  Matrices are column-major float[16] (glm::mat4 /
 * GLSL mat4 memory order): element (row r, column c) lives at [c*4 + r].
 */
#ifndef RT_CONTRACT_H
#define RT_CONTRACT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* std140 uniform block of the compute shader: assets/shaders/ray_tracing.comp:38-54,
 * host mirror src/main.cpp:39-54.  272 bytes. */
typedef struct rt_ubo {
  float view[16];            /*   0 */
  float projection[16];      /*  64 */
  float viewport_size[4];    /* 128  (width, height, 0, 0) */
  float camera_position[4];  /* 144 */
  float render_settings[4];  /* 160  x = samples per pixel per frame, y = max bounces */
  float frame[4];            /* 176  x = progressive frame index */
  float ground_colour[4];    /* 192 */
  float sky_colour_horizon[4]; /* 208 */
  float sky_colour_zenith[4];  /* 224 */
  float sun_position[4];     /* 240 */
  float sun_focus;           /* 256 */
  float sun_intensity;       /* 260 */
  float defocus_strength;    /* 264 */
  float diverge_strength;    /* 268 */
} rt_ubo;

/* ray_tracing.comp:13-20, main.cpp:56-60.  48 bytes. */
typedef struct rt_material {
  float colour[4];           /* rgb, a = smoothness */
  float emission_colour[4];  /* rgb, a = emission strength */
  float specular_colour[4];  /* rgb, a = specular probability */
} rt_material;

/* ray_tracing.comp:30-34, main.cpp:62-66.  std140 array stride 64 bytes. */
typedef struct rt_sphere {
  float position[3];
  float radius;
  rt_material material;
} rt_sphere;

/* src/camera.h:7-13, field for field (near/far renamed: they are macros on some
 * platforms; offsets are unchanged). */
typedef struct rt_camera {
  float fov, near_plane, far_plane;
  float pitch, yaw;
  float aspect_ratio;
  float distance;
  float viewport_width, viewport_height;
} rt_camera;

#ifdef __cplusplus
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
RT_STATIC_ASSERT(sizeof(rt_ubo) == 272, "UBO must be 272 bytes");
RT_STATIC_ASSERT(offsetof(rt_ubo, projection) == 64, "ubo.projection");
RT_STATIC_ASSERT(offsetof(rt_ubo, viewport_size) == 128, "ubo.viewportSize");
RT_STATIC_ASSERT(offsetof(rt_ubo, camera_position) == 144, "ubo.cameraPosition");
RT_STATIC_ASSERT(offsetof(rt_ubo, render_settings) == 160, "ubo.renderSettings");
RT_STATIC_ASSERT(offsetof(rt_ubo, frame) == 176, "ubo.frame");
RT_STATIC_ASSERT(offsetof(rt_ubo, ground_colour) == 192, "ubo.groundColour");
RT_STATIC_ASSERT(offsetof(rt_ubo, sky_colour_horizon) == 208, "ubo.skyColourHorizon");
RT_STATIC_ASSERT(offsetof(rt_ubo, sky_colour_zenith) == 224, "ubo.skyColourZenith");
RT_STATIC_ASSERT(offsetof(rt_ubo, sun_position) == 240, "ubo.sunPosition");
RT_STATIC_ASSERT(offsetof(rt_ubo, sun_focus) == 256, "ubo.sunFocus");
RT_STATIC_ASSERT(offsetof(rt_ubo, diverge_strength) == 268, "ubo.divergeStrength");
RT_STATIC_ASSERT(sizeof(rt_material) == 48, "material must be 48 bytes");
RT_STATIC_ASSERT(sizeof(rt_sphere) == 64, "sphere stride must be 64 bytes");
RT_STATIC_ASSERT(offsetof(rt_sphere, radius) == 12, "sphere.radius");
RT_STATIC_ASSERT(offsetof(rt_sphere, material) == 16, "sphere.material");
RT_STATIC_ASSERT(sizeof(rt_camera) == 36, "camera is nine floats");

#ifdef __cplusplus
}
#endif
#endif
