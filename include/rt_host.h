/*
 * rt_host.h -- host-side half of the drop-in contract: the reference's camera
 * (src/camera.h, src/camera.cpp) and its default scene and render settings
 * (src/main.cpp), restated without glm so that a caller can fill the 272-byte
 * uniform block and the 64-byte sphere records exactly as the reference's main()
 * does.  Exported from librt_hip.so next to the device entry points (rt_hip.h).
 *
 * Matrices are column-major float[16].
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_contract.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Camera's in-class initialisers, camera.h:7-13 */
void rt_camera_init(rt_camera *cam);
/* createCamera, camera.cpp:5-13 */
void rt_camera_create(float start_fov, float start_aspect_ratio, float start_near,
                      float start_far, rt_camera *out_camera);
/* cameraRotate, camera.cpp:15-19 (delta = mouse delta * 0.01, main.cpp:682-686) */
void rt_camera_rotate(rt_camera *cam, float delta_x, float delta_y);
/* cameraGetProjectionMatrix, camera.cpp:21-25: left-handed, depth -1..1
 * (CMakeLists.txt:28-30); also refreshes cam->aspect_ratio like the reference */
void rt_camera_projection_matrix(rt_camera *cam, float out16[16]);
/* cameraGetViewMatrix, camera.cpp:27-34 */
void rt_camera_view_matrix(const rt_camera *cam, float out16[16]);
/* cameraGetOrientation, camera.cpp:48-50; out = (w, x, y, z) */
void rt_camera_orientation(const rt_camera *cam, float out_wxyz[4]);
/* cameraGetUp / Right / Forward, camera.cpp:36-46 */
void rt_camera_up(const rt_camera *cam, float out3[3]);
void rt_camera_right(const rt_camera *cam, float out3[3]);
void rt_camera_forward(const rt_camera *cam, float out3[3]);
/* cameraGetPanSpeed / RotationSpeed / ZoomSpeed, camera.cpp:52-71 */
void rt_camera_pan_speed(const rt_camera *cam, float out2[2]);
float rt_camera_rotation_speed(const rt_camera *cam);
float rt_camera_zoom_speed(const rt_camera *cam);

/* The uniform block as main() fills it: render settings main.cpp:626-636, camera
 * part main.cpp:697-701.  cam == NULL uses createCamera(90, 1, 0.01, 10000), i.e.
 * main.cpp:624 where 800/608 is an unsigned division. */
void rt_default_ubo(rt_camera *cam, rt_ubo *out);
/* Only the camera-dependent fields (view, projection, viewportSize, cameraPosition),
 * as the frame loop rewrites them every frame, main.cpp:697-701. */
void rt_ubo_set_camera(rt_camera *cam, rt_ubo *ubo);
/* The three spheres of main.cpp:517-538; returns 3. */
uint32_t rt_default_scene(rt_sphere out3[3]);

/* No counterpart in the reference (it has one hard-coded scene): a deterministic synthetic
 * scene of n spheres in the reference's record format.  Sphere 0 is the ground, centre
 * (0,-1002,-60), radius 1000, diffuse grey; the others rest on it on a jittered square
 * grid over x in [-50,50], z in [-110,-10], radius 0.2..0.5.  All random choices are drawn
 * with the shader's own PCG hash (ray_tracing.comp:112-119) from `seed`, in this order
 * per sphere: radius, x jitter, z jitter, material class, r, g, b, [class parameter].
 * Classes: < 0.60 diffuse; < 0.85 metal (smoothness 0.5..1, specular probability 1,
 * specular colour 0.5 + 0.5 rgb); else glossy coat (smoothness 1, specular probability
 * 0.15..0.5).  Every 50th sphere also emits its colour with strength 4.  The reference's
 * material has no refraction term, so there is no dielectric class.  Returns n. */
uint32_t rt_grid_scene(uint32_t n, uint32_t seed, rt_sphere *out);

/* The same scene (same seed, same draws, same centres and radii) with the third material
 * class made of glass instead of a glossy coat: for those spheres ior_out[i] = 1.5 and
 * colour.rgb = 0.75 + 0.25 * (the drawn r, g, b) is the tint; every other sphere gets
 * ior_out[i] = 0 and the record rt_grid_scene writes.  ior_out (n floats) is what
 * rt_set_dielectric takes.  Returns n. */
uint32_t rt_grid_scene_glass(uint32_t n, uint32_t seed, rt_sphere *out, float *ior_out);

/* No counterpart in the reference, whose only texture is a photograph
 * (assets/textures/brickwall.jpg, never sampled): a deterministic brick-wall pattern for
 * tests and benchmarks.  width x height sRGB texels, 4 bytes each, written row by row.
 * Rows of 8 courses per image height, bricks two courses long, every other course offset
 * by half a brick, mortar joints 1/16 of a course wide; each brick's tint comes from the
 * shader's PCG hash (ray_tracing.comp:112-119) of its course and column, and a hash of the
 * texel adds +-4 levels of grain.  The pattern tiles seamlessly. */
void rt_brick_texture(uint32_t width, uint32_t height, uint8_t *rgba8_out);

/* Writes a binary PPM (P6) of an rgba8 image; flip_y != 0 writes the last buffer row
 * first, which is how the reference's window shows the image (negative viewport
 * height, main.cpp:934-940).  Returns 0, or -1 with errno set. */
int rt_write_ppm(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height,
                 int flip_y);

/* Same image as an 8-bit RGB PNG (stored, i.e. uncompressed, deflate blocks; no external
 * library: the reference links stb only for reading, main.cpp:29-33). */
int rt_write_png(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height,
                 int flip_y);

/* ---- data formats either side of the path (host only) --------------------------------
 * The reference links assimp (CMakeLists.txt:12) and stb_image (main.cpp:29-33) for these
 * and copies the .obj files of assets/models and everything in assets/textures next to the binary
 * (CMakeLists.txt:78-89), but calls neither on the compute path.  Buffers returned through
 * out-pointers are malloc'ed; release them with rt_free.  On failure the functions return
 * -1 and rt_ingest_error() describes why. */
const char *rt_ingest_error(void);
void rt_free(void *p);

/* Wavefront OBJ: "v x y z" and "f ..." records (corners "v", "v/vt", "v//vn", "v/vt/vn",
 * negative = relative indices; polygons are fan-triangulated; everything else is ignored).
 * Result in the layout rt_set_mesh takes. */
int rt_load_obj(const char *path, float **out_vertices, uint32_t *out_n_vertices,
                uint32_t **out_indices, uint32_t *out_n_triangles);

/* Baseline (and extended-sequential Huffman) JPEG, 8 bits, grey or YCbCr with sampling
 * factors 1 or 2, restart intervals; progressive files are rejected.  Output RGBA8 (alpha
 * 255), row 0 = top of the picture, or bottom when flip_y != 0 -- what loadTexture asks of
 * stb with stbi_set_flip_vertically_on_load (main.cpp:1384-1409).  Inverse DCT and chroma
 * interpolation are done in double precision, so values may differ from libjpeg's integer
 * pipeline by a level or two. */
int rt_decode_jpeg(const void *bytes, size_t size, int flip_y, uint8_t **out_rgba8,
                   uint32_t *out_width, uint32_t *out_height);
int rt_load_jpeg(const char *path, int flip_y, uint8_t **out_rgba8, uint32_t *out_width,
                 uint32_t *out_height);

#ifdef __cplusplus
}
#endif
#endif
