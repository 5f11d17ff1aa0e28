/*
 * rt_oracle.c -- CPU oracle for the per-pixel path-tracing loop of
 * novadinn/ray-tracer (assets/shaders/ray_tracing.comp, src/camera.cpp,
 * src/main.cpp).  TEST INFRASTRUCTURE ONLY; see rt_oracle.h for the rules and the
 * "parity unpinned" statement.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (see Makefile).  Every
 * fused multiply-add in this file is an explicit fmaf(); everything else rounds
 * once per operator.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* small vector helpers                                                        */
/* ------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
/* dot(): x*x first, then y and z folded in with one fused op each */
static inline float v_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
/* a*s + b, fused per component */
static inline v3 v_fma(v3 a, float s, v3 b) {
  return V3(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z));
}
/* normalize(): v * (1 / sqrt(dot(v,v))) with IEEE sqrt and divide */
static inline v3 v_normalize(v3 a) {
  float inv = 1.0f / sqrtf(v_dot(a, a));
  return v_scale(a, inv);
}
/* GLSL mix(x, y, a) = x*(1-a) + y*a, the y*a product fused into the sum */
static inline float mixf(float x, float y, float a) { return fmaf(y, a, x * (1.0f - a)); }
static inline v3 v_mix(v3 x, v3 y, float a) {
  return V3(mixf(x.x, y.x, a), mixf(x.y, y.y, a), mixf(x.z, y.z, a));
}

/* ------------------------------------------------------------------------- */
/* RNG: ray_tracing.comp:112-119                                               */
/* ------------------------------------------------------------------------- */
uint32_t rto_next_random(uint32_t *state) {
  uint32_t s = *state * 747796405u + 2891336453u;
  *state = s;
  uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (r >> 22) ^ r;
}

/* the shader divides by the literal 4294967295.0, which is 2^32 in fp32 */
float rto_random_value(uint32_t *state) {
  return (float)rto_next_random(state) * 0x1p-32f;
}

/* ------------------------------------------------------------------------- */
/* portable fp32 transcendental kernels                                        */
/* ------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* reduce x to r in [-pi/4, pi/4] and a quadrant; valid for |x| < ~1e4 */
static inline float pm_reduce(float x, int *quadrant) {
  float q = rintf(x * 0.636619747f); /* 2/pi */
  float r = fmaf(q, -1.57079601e+00f, x);
  r = fmaf(q, -3.13916473e-07f, r);
  r = fmaf(q, -5.39030253e-15f, r);
  *quadrant = (int)q & 3;
  return r;
}
static inline float pm_sin_poly(float r) {
  float z = r * r;
  float p = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  p = fmaf(p, z, -1.6666654611e-1f);
  return fmaf(p * z, r, r);
}
static inline float pm_cos_poly(float r) {
  float z = r * r;
  float p = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  p = fmaf(p, z, 4.166664568298827e-2f);
  return fmaf(p * z, z, fmaf(z, -0.5f, 1.0f));
}
float rto_pm_sin(float x) {
  int q;
  float r = pm_reduce(x, &q);
  float v = (q & 1) ? pm_cos_poly(r) : pm_sin_poly(r);
  return (q & 2) ? -v : v;
}
float rto_pm_cos(float x) {
  int q;
  float r = pm_reduce(x, &q);
  float v = (q & 1) ? pm_sin_poly(r) : pm_cos_poly(r);
  return ((q + 1) & 2) ? -v : v;
}

/* split a positive normal float into m in [sqrt(1/2), sqrt(2)) - 1 and exponent */
static inline float pm_log_core(float x, float *fe) {
  uint32_t u = f2u(x);
  int e = (int)(u >> 23) - 126;                     /* x = m * 2^e, m in [0.5,1) */
  float m = u2f((u & 0x007fffffu) | 0x3f000000u);
  if (m < 0.707106781f) { e -= 1; m = m + m; }
  float t = m - 1.0f;
  float z = t * t;
  float p = fmaf(t, 7.0376836292e-2f, -1.1514610310e-1f);
  p = fmaf(p, t, 1.1676998740e-1f);
  p = fmaf(p, t, -1.2420140846e-1f);
  p = fmaf(p, t, 1.4249322787e-1f);
  p = fmaf(p, t, -1.6668057665e-1f);
  p = fmaf(p, t, 2.0000714765e-1f);
  p = fmaf(p, t, -2.4999993993e-1f);
  p = fmaf(p, t, 3.3333331174e-1f);
  float y = (p * t) * z;
  y = fmaf(z, -0.5f, y);
  *fe = (float)e;
  return t + y;                                     /* ln(m) */
}
/* domain handling shared by log and log2: returns 1 when *out is final */
static inline int pm_log_special(float x, float *out) {
  if (x != x) { *out = x; return 1; }
  if (x < 0.0f) { *out = u2f(0x7fc00000u); return 1; }
  if (x == 0.0f) { *out = -INFINITY; return 1; }
  if (x == INFINITY) { *out = x; return 1; }
  return 0;
}
float rto_pm_log(float x) {
  float out, fe;
  if (pm_log_special(x, &out)) return out;
  if (x < 0x1p-126f) {                              /* subnormal: rescale exactly */
    float lm = pm_log_core(x * 0x1p24f, &fe);
    fe -= 24.0f;
    return fmaf(fe, 0.693359375f, fmaf(fe, -2.12194440e-4f, lm));
  }
  float lm = pm_log_core(x, &fe);
  return fmaf(fe, 0.693359375f, fmaf(fe, -2.12194440e-4f, lm));
}
float rto_pm_log2(float x) {
  float out, fe;
  if (pm_log_special(x, &out)) return out;
  if (x < 0x1p-126f) {
    float lm = pm_log_core(x * 0x1p24f, &fe);
    return fmaf(lm, 1.44269504f, fe - 24.0f);
  }
  float lm = pm_log_core(x, &fe);
  return fmaf(lm, 1.44269504f, fe);
}
float rto_pm_exp2(float x) {
  if (x != x) return x;
  if (x >= 128.0f) return INFINITY;
  if (x < -125.0f) return 0.0f;                     /* results below ~2^-125 flush to 0 */
  float k = rintf(x);
  float r = x - k;
  float p = fmaf(r, 1.52527338e-5f, 1.54035304e-4f);
  p = fmaf(p, r, 1.33335581e-3f);
  p = fmaf(p, r, 9.61812911e-3f);
  p = fmaf(p, r, 5.55041087e-2f);
  p = fmaf(p, r, 2.40226507e-1f);
  p = fmaf(p, r, 6.93147181e-1f);
  p = fmaf(p, r, 1.0f);
  int ki = (int)k;                                  /* -125 .. 128 */
  if (ki > 127) { ki = 127; p = p * 2.0f; }         /* x in [127.5, 128): 2^128 is not a float */
  return p * u2f((uint32_t)(ki + 127) << 23);
}
/* GLSL pow(x,y) = exp2(y * log2(x)) */
float rto_pm_pow(float x, float y) { return rto_pm_exp2(y * rto_pm_log2(x)); }

/* dispatch on the oracle's math mode */
typedef struct {
  float (*sin_)(float);
  float (*cos_)(float);
  float (*log_)(float);
  float (*pow_)(float, float);
} math_ops;
static float libm_sin(float x) { return sinf(x); }
static float libm_cos(float x) { return cosf(x); }
static float libm_log(float x) { return logf(x); }
static float libm_pow(float x, float y) { return powf(x, y); }
static const math_ops k_portable = {rto_pm_sin, rto_pm_cos, rto_pm_log, rto_pm_pow};
static const math_ops k_libm = {libm_sin, libm_cos, libm_log, libm_pow};

/* ------------------------------------------------------------------------- */
/* matrices, camera (camera.cpp), defaults (main.cpp)                          */
/* ------------------------------------------------------------------------- */
#define A(r, c) m[(c) * 4 + (r)]
void rto_mat4_inverse(const float m[16], float out[16]) {
  float s0 = A(0,0) * A(1,1) - A(1,0) * A(0,1);
  float s1 = A(0,0) * A(1,2) - A(1,0) * A(0,2);
  float s2 = A(0,0) * A(1,3) - A(1,0) * A(0,3);
  float s3 = A(0,1) * A(1,2) - A(1,1) * A(0,2);
  float s4 = A(0,1) * A(1,3) - A(1,1) * A(0,3);
  float s5 = A(0,2) * A(1,3) - A(1,2) * A(0,3);
  float c5 = A(2,2) * A(3,3) - A(3,2) * A(2,3);
  float c4 = A(2,1) * A(3,3) - A(3,1) * A(2,3);
  float c3 = A(2,1) * A(3,2) - A(3,1) * A(2,2);
  float c2 = A(2,0) * A(3,3) - A(3,0) * A(2,3);
  float c1 = A(2,0) * A(3,2) - A(3,0) * A(2,2);
  float c0 = A(2,0) * A(3,1) - A(3,0) * A(2,1);
  float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
  float id = 1.0f / det;
  float b[4][4];
  b[0][0] = ( A(1,1) * c5 - A(1,2) * c4 + A(1,3) * c3) * id;
  b[0][1] = (-A(0,1) * c5 + A(0,2) * c4 - A(0,3) * c3) * id;
  b[0][2] = ( A(3,1) * s5 - A(3,2) * s4 + A(3,3) * s3) * id;
  b[0][3] = (-A(2,1) * s5 + A(2,2) * s4 - A(2,3) * s3) * id;
  b[1][0] = (-A(1,0) * c5 + A(1,2) * c2 - A(1,3) * c1) * id;
  b[1][1] = ( A(0,0) * c5 - A(0,2) * c2 + A(0,3) * c1) * id;
  b[1][2] = (-A(3,0) * s5 + A(3,2) * s2 - A(3,3) * s1) * id;
  b[1][3] = ( A(2,0) * s5 - A(2,2) * s2 + A(2,3) * s1) * id;
  b[2][0] = ( A(1,0) * c4 - A(1,1) * c2 + A(1,3) * c0) * id;
  b[2][1] = (-A(0,0) * c4 + A(0,1) * c2 - A(0,3) * c0) * id;
  b[2][2] = ( A(3,0) * s4 - A(3,1) * s2 + A(3,3) * s0) * id;
  b[2][3] = (-A(2,0) * s4 + A(2,1) * s2 - A(2,3) * s0) * id;
  b[3][0] = (-A(1,0) * c3 + A(1,1) * c1 - A(1,2) * c0) * id;
  b[3][1] = ( A(0,0) * c3 - A(0,1) * c1 + A(0,2) * c0) * id;
  b[3][2] = (-A(3,0) * s3 + A(3,1) * s1 - A(3,2) * s0) * id;
  b[3][3] = ( A(2,0) * s3 - A(2,1) * s1 + A(2,2) * s0) * id;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) out[c * 4 + r] = b[r][c];
}
#undef A

/* column-major mat4 * vec4, each component an x,y,z,w fma chain */
static inline void mat4_mul_vec4(const float m[16], const float v[4], float out[4]) {
  for (int r = 0; r < 4; ++r) {
    float acc = m[0 + r] * v[0];
    acc = fmaf(m[4 + r], v[1], acc);
    acc = fmaf(m[8 + r], v[2], acc);
    acc = fmaf(m[12 + r], v[3], acc);
    out[r] = acc;
  }
}

void rto_camera_default(rto_camera *cam) { /* camera.h:7-13 */
  cam->fov = 45.0f; cam->near_ = 0.1f; cam->far_ = 1000.0f;
  cam->pitch = 0.0f; cam->yaw = 0.0f;
  cam->aspect_ratio = 1.778f;
  cam->distance = 10.0f;
  cam->viewport_width = 800.0f; cam->viewport_height = 600.0f;
}

void rto_camera_create(float fov, float aspect, float near_, float far_, rto_camera *cam) {
  rto_camera_default(cam);
  cam->fov = fov; cam->aspect_ratio = aspect; cam->near_ = near_; cam->far_ = far_;
}

/* glm::quat(vec3(-pitch, -yaw, 0)): camera.cpp:48-50 */
void rto_camera_orientation(const rto_camera *cam, float q[4]) {
  float ex = -cam->pitch, ey = -cam->yaw, ez = 0.0f;
  float cx = cosf(ex * 0.5f), cy = cosf(ey * 0.5f), cz = cosf(ez * 0.5f);
  float sx = sinf(ex * 0.5f), sy = sinf(ey * 0.5f), sz = sinf(ez * 0.5f);
  q[0] = cx * cy * cz + sx * sy * sz; /* w */
  q[1] = sx * cy * cz - cx * sy * sz; /* x */
  q[2] = cx * sy * cz + sx * cy * sz; /* y */
  q[3] = cx * cy * sz - sx * sy * cz; /* z */
}

/* glm::rotate(quat, vec3) = v + 2*(w*(qv x v) + qv x (qv x v)) */
static void quat_rotate(const float q[4], v3 v, float out[3]) {
  v3 qv = V3(q[1], q[2], q[3]);
  v3 uv = V3(qv.y * v.z - v.y * qv.z, qv.z * v.x - v.z * qv.x, qv.x * v.y - v.x * qv.y);
  v3 uuv = V3(qv.y * uv.z - uv.y * qv.z, qv.z * uv.x - uv.z * qv.x, qv.x * uv.y - uv.x * qv.y);
  out[0] = v.x + ((uv.x * q[0]) + uuv.x) * 2.0f;
  out[1] = v.y + ((uv.y * q[0]) + uuv.y) * 2.0f;
  out[2] = v.z + ((uv.z * q[0]) + uuv.z) * 2.0f;
}
void rto_camera_up(const rto_camera *cam, float out[3]) {
  float q[4]; rto_camera_orientation(cam, q); quat_rotate(q, V3(0, 1, 0), out);
}
void rto_camera_right(const rto_camera *cam, float out[3]) {
  float q[4]; rto_camera_orientation(cam, q); quat_rotate(q, V3(1, 0, 0), out);
}
void rto_camera_forward(const rto_camera *cam, float out[3]) {
  float q[4]; rto_camera_orientation(cam, q); quat_rotate(q, V3(0, 0, 1), out);
}

/* camera.cpp:15-19; rotation speed is the constant 0.8 (camera.cpp:62) */
void rto_camera_rotate(rto_camera *cam, float dx, float dy) {
  float up[3];
  rto_camera_up(cam, up);
  float yaw_sign = up[1] < 0.0f ? 1.0f : -1.0f;
  cam->yaw += yaw_sign * dx * 0.8f;
  cam->pitch += dy * 0.8f;
}

/* camera.cpp:21-25 with GLM_FORCE_LEFT_HANDED and depth -1..1 (CMakeLists.txt:28-30) */
void rto_camera_projection(rto_camera *cam, float p[16]) {
  cam->aspect_ratio = cam->viewport_width / cam->viewport_height;
  float fovy = cam->fov * 0.01745329251994329576923690768489f;
  float t = tanf(fovy / 2.0f);
  memset(p, 0, 16 * sizeof(float));
  p[0] = 1.0f / (cam->aspect_ratio * t);
  p[5] = 1.0f / t;
  p[10] = (cam->far_ + cam->near_) / (cam->far_ - cam->near_);
  p[11] = 1.0f;
  p[14] = -(2.0f * cam->far_ * cam->near_) / (cam->far_ - cam->near_);
}

/* camera.cpp:27-34: inverse(translate(I, 0) * toMat4(orientation)) */
void rto_camera_view(const rto_camera *cam, float view[16]) {
  float q[4];
  rto_camera_orientation(cam, q);
  float w = q[0], x = q[1], y = q[2], z = q[3];
  float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z;
  float wx = w * x, wy = w * y, wz = w * z;
  float r[16];
  memset(r, 0, sizeof r);
  r[0] = 1.0f - 2.0f * (yy + zz); r[1] = 2.0f * (xy + wz);        r[2] = 2.0f * (xz - wy);
  r[4] = 2.0f * (xy - wz);        r[5] = 1.0f - 2.0f * (xx + zz); r[6] = 2.0f * (yz + wx);
  r[8] = 2.0f * (xz + wy);        r[9] = 2.0f * (yz - wx);        r[10] = 1.0f - 2.0f * (xx + yy);
  r[15] = 1.0f;
  rto_mat4_inverse(r, view);
}

/* main.cpp:626-636 (settings) and :697-701 (camera part) */
void rto_default_ubo(const rto_camera *cam_in, void *ubo272) {
  float u[68];
  memset(u, 0, sizeof u);
  rto_camera cam;
  if (cam_in) cam = *cam_in;
  else rto_camera_create(90.0f, 1.0f, 0.01f, 10000.0f, &cam); /* main.cpp:624: 800/608 as uint */
  rto_camera_view(&cam, &u[0]);
  rto_camera_projection(&cam, &u[16]);
  u[32] = cam.viewport_width; u[33] = cam.viewport_height;       /* viewportSize */
  /* cameraPosition u[36..39] = 0 */
  u[40] = 50.0f; u[41] = 25.0f;                                   /* renderSettings */
  /* frame u[44..47] = 0 */
  u[48] = 0.35f; u[49] = 0.3f; u[50] = 0.35f; u[51] = 1.0f;       /* groundColour */
  u[52] = u[53] = u[54] = u[55] = 1.0f;                           /* skyColourHorizon */
  u[56] = 0.078f; u[57] = 0.36f; u[58] = 0.72f; u[59] = 1.0f;     /* skyColourZenith */
  u[60] = u[61] = u[62] = u[63] = 0.5f;                           /* normalize(vec4(1)) */
  u[64] = 1.0f;                                                   /* sunFocus */
  u[65] = 0.0f;                                                   /* sunInternsity */
  u[66] = 0.0f;                                                   /* defocusStrength */
  u[67] = 1.0f;                                                   /* divergeStrength */
  memcpy(ubo272, u, 272);
}

/* main.cpp:517-538 */
uint32_t rto_default_spheres(void *dst) {
  static const float s[3][16] = {
      {0, 0, -5, 1,      0.5f, 0.5f, 0.5f, 1.0f,   0, 0, 0, 0,  1, 1, 1, 0.5f},
      {3, 0, -5, 1,      0.8f, 0.2f, 0.2f, 0.5f,   0, 0, 0, 0,  1, 1, 1, 0.0f},
      {0, -101, -5, 100, 0.2f, 0.8f, 0.05f, 0.0f,  0, 0, 0, 0,  1, 1, 1, 0.0f},
  };
  memcpy(dst, s, sizeof s);
  return 3;
}

/* ------------------------------------------------------------------------- */
/* the shader                                                                  */
/* ------------------------------------------------------------------------- */
typedef struct {
  /* UBO fields, ray_tracing.comp:38-54 */
  float inv_view[16], inv_proj[16];
  float viewport_w, viewport_h;
  v3 cam_pos;
  float spp, max_bounce, frame;
  v3 ground, horizon, zenith, sun_pos;
  float sun_focus, sun_intensity, defocus, diverge;
  /* scene */
  const float *spheres; /* 16 floats per sphere */
  uint32_t n_spheres;
  const rto_mesh *mesh; /* optional triangle mesh (extension, see rt_oracle.h) */
  const rto_texture *tex; /* optional albedo texture (extension) */
  const float *ior;       /* optional per-sphere index of refraction, <= 0: not a dielectric (extension) */
  float tex_lut[256];     /* byte -> linear value */
  uint32_t width, height;
  const math_ops *mo;
} frame_ctx;

typedef struct { int hit; float dst; v3 point, normal; int index; } hit_t;

/* raySphere, ray_tracing.comp:145-170.  far_root = 0 is the shader's function: the near root
 * only, so a ray that starts inside a sphere never hits it.  far_root = 1 (extension, used for
 * the one dielectric sphere a refracted ray is travelling inside) takes the other root,
 * dst = (-b + sqrt(disc)) / (2a), with the same acceptance test dst >= 0. */
static inline hit_t ray_sphere(v3 o, v3 d, v3 centre, float radius, int far_root) {
  hit_t h;
  h.hit = 0; h.dst = 0.0f; h.point = V3(0, 0, 0); h.normal = V3(0, 0, 0); h.index = -1;
  v3 oc = v_sub(o, centre);
  float a = v_dot(d, d);
  float b = 2.0f * v_dot(oc, d);
  float c = v_dot(oc, oc) - radius * radius;
  float disc = fmaf(b, b, -((4.0f * a) * c));
  if (disc >= 0.0f) {
    float root = sqrtf(disc);
    float dst = (far_root ? -b + root : -b - root) / (2.0f * a);
    if (dst >= 0.0f) {
      h.hit = 1;
      h.dst = dst;
      h.point = v_fma(d, dst, o);
      h.normal = v_normalize(v_sub(h.point, centre));
    }
  }
  return h;
}

/* a x b with the second product of each component fused into the difference */
static inline v3 v_cross(v3 a, v3 b) {
  return V3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

/* Ray-triangle test (extension: the reference has spheres only).  Moeller-Trumbore with the
 * barycentric tests done on the un-divided numerators, so only an accepted hit divides:
 *   e1 = v1-v0, e2 = v2-v0, p = d x e2, det = e1.p          (det == 0: parallel, miss)
 *   tv = o-v0, un = tv.p, q = tv x e1, vn = d.q, tn = e2.q
 *   with s = sign(det): s*un >= 0, s*vn >= 0, s*(un+vn) <= s*det, s*tn >= 0   else miss
 *   det*det > 2^-40 (e1.e1)(p.p) and det*det > 2^-40 (d.d)(n.n), n = e1 x e2
 *                                  else miss (the ray runs in the triangle's plane)
 *   dst = tn / det;  point = o + d*dst;  normal = normalize(e1 x e2), turned against d
 * Both faces are hit.  A triangle without area -- e1 x e2 == (0, 0, 0) as computed by v_cross, which
 * covers two equal vertices and collinear lattice points -- is never hit: the sign tests above
 * would otherwise decide on rounding residue (closest_hit leaves such triangles out). */
static inline hit_t ray_triangle(v3 o, v3 d, v3 v0, v3 v1, v3 v2) {
  hit_t h;
  h.hit = 0; h.dst = 0.0f; h.point = V3(0, 0, 0); h.normal = V3(0, 0, 0); h.index = -1;
  v3 e1 = v_sub(v1, v0), e2 = v_sub(v2, v0);
  v3 p = v_cross(d, e2);
  float det = v_dot(e1, p);
  if (!(det != 0.0f)) return h;                 /* zero or NaN */
  float s = det > 0.0f ? 1.0f : -1.0f;
  v3 tv = v_sub(o, v0);
  float un = v_dot(tv, p);
  if (!(un * s >= 0.0f)) return h;
  v3 q = v_cross(tv, e1);
  float vn = v_dot(d, q);
  if (!(vn * s >= 0.0f) || !((un + vn) * s <= det * s)) return h;
  float tn = v_dot(e2, q);
  if (!(tn * s >= 0.0f)) return h;
  /* A ray that runs in the triangle's plane to within rounding (|sin| of its angle to the plane
   * below 2^-20): det is then rounding residue, and so is everything divided by it.  Tested last,
   * so only a candidate that would otherwise be accepted pays for it. */
  if (!(det * det > 0x1p-40f * (v_dot(e1, e1) * v_dot(p, p)))) return h;
  /* The same angle measured against the plane's own normal: when the ray also runs nearly along
   * the edge e2 (a camera on the extension of that edge), p = d x e2 is itself a small difference
   * of large products, and the residue in det is no longer small beside |e1| |p|. */
  {
    v3 nn = v_cross(e1, e2);
    if (!(det * det > 0x1p-40f * (v_dot(d, d) * v_dot(nn, nn)))) return h;
  }
  h.hit = 1;
  h.dst = tn / det;
  h.point = v_fma(d, h.dst, o);
  v3 n = v_normalize(v_cross(e1, e2));
  if (v_dot(n, d) > 0.0f) n = V3(-n.x, -n.y, -n.z);
  h.normal = n;
  return h;
}

/* calculateRayCollision, ray_tracing.comp:172-190, extended over the mesh: primitive ids
 * are sphere indices followed by n_spheres + triangle index; strict "<" keeps the lowest
 * id among equal distances.  skip is the primitive the ray starts on or in (-1: none).  A
 * triangle: a planar primitive cannot be hit again by a ray leaving it, and without an epsilon
 * the test above would report it at dst = 0, so it is left out.  A sphere (only ever a
 * dielectric one the ray was refracted into): it is tested with its far root. */
static inline hit_t closest_hit(const frame_ctx *fc, v3 o, v3 d, int skip) {
  hit_t best;
  best.hit = 0; best.dst = 3.402823466e+38f;
  best.point = V3(0, 0, 0); best.normal = V3(0, 0, 0); best.index = -1;
  for (uint32_t i = 0; i < fc->n_spheres; ++i) {
    const float *s = fc->spheres + 16u * i;
    hit_t h = ray_sphere(o, d, V3(s[0], s[1], s[2]), s[3], (int)i == skip);
    if (h.hit && h.dst < best.dst) { best = h; best.index = (int)i; }
  }
  if (fc->mesh) {
    const rto_mesh *m = fc->mesh;
    for (uint32_t t = 0; t < m->n_triangles; ++t) {
      const int id = (int)(fc->n_spheres + t);
      if (id == skip) continue;
      const float *a = m->vertices + 3u * m->indices[3u * t + 0];
      const float *b = m->vertices + 3u * m->indices[3u * t + 1];
      const float *c = m->vertices + 3u * m->indices[3u * t + 2];
      {
        v3 va = V3(a[0], a[1], a[2]);
        v3 n = v_cross(v_sub(V3(b[0], b[1], b[2]), va), v_sub(V3(c[0], c[1], c[2]), va));
        if (n.x == 0.0f && n.y == 0.0f && n.z == 0.0f) continue;   /* no area: never hit */
      }
      hit_t h = ray_triangle(o, d, V3(a[0], a[1], a[2]), V3(b[0], b[1], b[2]), V3(c[0], c[1], c[2]));
      if (h.hit && h.dst < best.dst) { best = h; best.index = id; }
    }
  }
  return best;
}

float rto_srgb_to_linear(uint8_t v) {
  double c = (double)v / 255.0;
  return (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
}

/* Texture coordinates of a hit.  PLANAR_XZ: (point.x, point.z).  OCTAHEDRAL: the unit
 * normal n folded onto the square: q = n.xz / (|n.x|+|n.y|+|n.z|); on the lower half
 * (n.y < 0) q = (1 - |q.zx|) * sign(q.xz); (a, b) = q * 0.5 + 0.5. */
static inline void texture_coords(const rto_texture *t, v3 point, v3 n, float *a, float *b) {
  if (t->mapping == RTO_TEX_MAP_OCTAHEDRAL) {
    float s = (fabsf(n.x) + fabsf(n.y)) + fabsf(n.z);
    float qx = n.x / s, qz = n.z / s;
    if (n.y < 0.0f) {
      float fx = (1.0f - fabsf(qz)) * (qx >= 0.0f ? 1.0f : -1.0f);
      float fz = (1.0f - fabsf(qx)) * (qz >= 0.0f ? 1.0f : -1.0f);
      qx = fx; qz = fz;
    }
    *a = fmaf(qx, 0.5f, 0.5f);
    *b = fmaf(qz, 0.5f, 0.5f);
  } else {
    *a = point.x;
    *b = point.z;
  }
}

/* LINEAR / REPEAT sample at (u, v), one unit = one repeat:
 *   ur = u - floor(u);  x = ur*W - 0.5 (fused);  x0 = floor(x);  fx = x - x0;
 *   columns x0 and x0+1 wrapped into [0, W); same for rows;
 *   texel bytes -> floats through the 256-entry table;
 *   lerp along x then y, each as fma(weight, right - left, left). */
static inline v3 texture_sample(const frame_ctx *fc, float u, float v) {
  const rto_texture *t = fc->tex;
  const int W = (int)t->width, H = (int)t->height;
  float ur = u - floorf(u), vr = v - floorf(v);
  if (!(ur >= 0.0f && ur <= 1.0f)) ur = 0.0f;   /* NaN or infinite coordinate */
  if (!(vr >= 0.0f && vr <= 1.0f)) vr = 0.0f;
  float x = fmaf(ur, (float)W, -0.5f), y = fmaf(vr, (float)H, -0.5f);
  float x0 = floorf(x), y0 = floorf(y);
  float fx = x - x0, fy = y - y0;
  int ix0 = x0 < 0.0f ? W - 1 : (int)x0, iy0 = y0 < 0.0f ? H - 1 : (int)y0;
  if (ix0 >= W) ix0 = W - 1;
  if (iy0 >= H) iy0 = H - 1;
  int ix1 = ix0 + 1 == W ? 0 : ix0 + 1, iy1 = iy0 + 1 == H ? 0 : iy0 + 1;
  const uint8_t *c00 = t->rgba8 + 4u * ((size_t)iy0 * W + ix0), *c10 = t->rgba8 + 4u * ((size_t)iy0 * W + ix1);
  const uint8_t *c01 = t->rgba8 + 4u * ((size_t)iy1 * W + ix0), *c11 = t->rgba8 + 4u * ((size_t)iy1 * W + ix1);
  float out[3];
  for (int k = 0; k < 3; ++k) {
    float a = fc->tex_lut[c00[k]], b = fc->tex_lut[c10[k]];
    float c = fc->tex_lut[c01[k]], d = fc->tex_lut[c11[k]];
    float top = fmaf(fx, b - a, a), bot = fmaf(fx, d - c, c);
    out[k] = fmaf(fy, bot - top, top);
  }
  return V3(out[0], out[1], out[2]);
}

static inline float clamp01(float t) { return t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t); }
/* GLSL smoothstep: t = clamp((x-e0)/(e1-e0),0,1); t*t*(3-2t).  The quotient is taken as
 * (x-e0) * k with k = 1/(e1-e0) rounded to fp32 once -- one of the evaluations GLSL's
 * 2.5-ulp bound on '/' allows, and the one this oracle pins. */
static inline float smoothstepf(float e0, float e1, float x) {
  const float k = 1.0f / (e1 - e0);
  float t = clamp01((x - e0) * k);
  return (t * t) * (3.0f - 2.0f * t);
}

/* getEnvironmentLight, ray_tracing.comp:239-250 */
static inline v3 environment(const frame_ctx *fc, v3 d) {
  float sky_t = fc->mo->pow_(smoothstepf(0.0f, 0.4f, d.y), 0.35f);
  float g2s = smoothstepf(-0.01f, 0.0f, d.y);
  v3 sky = v_mix(fc->horizon, fc->zenith, sky_t);
  float sd = v_dot(d, fc->sun_pos);
  float sun = fc->mo->pow_(sd > 0.0f ? sd : 0.0f, fc->sun_focus) * fc->sun_intensity;
  v3 comp = v_mix(fc->ground, sky, g2s);
  float s = sun * (g2s >= 1.0f ? 1.0f : 0.0f);
  return V3(comp.x + s, comp.y + s, comp.z + s);
}

/* randomValueNormalDistribution, ray_tracing.comp:121-125 */
static inline float normal_draw(const frame_ctx *fc, uint32_t *rng) {
  float theta = 6.283185f * rto_random_value(rng);          /* (2 * 3.1415926) * u */
  float rho = sqrtf(-2.0f * fc->mo->log_(rto_random_value(rng)));
  return rho * fc->mo->cos_(theta);
}
/* randomDirection, ray_tracing.comp:127-132 */
static inline v3 random_direction(const frame_ctx *fc, uint32_t *rng) {
  float x = normal_draw(fc, rng);
  float y = normal_draw(fc, rng);
  float z = normal_draw(fc, rng);
  return v_normalize(V3(x, y, z));
}
/* randomPointInCircle, ray_tracing.comp:134-138 */
static inline void point_in_circle(const frame_ctx *fc, uint32_t *rng, float out[2]) {
  float angle = (rto_random_value(rng) * 2.0f) * 3.1415926f;
  float c = fc->mo->cos_(angle), s = fc->mo->sin_(angle);
  float r = sqrtf(rto_random_value(rng));
  out[0] = c * r; out[1] = s * r;
}

/* Dielectric sphere (EXTENSION: the reference's material, ray_tracing.comp:13-20, has no
 * refraction term and its sphere test takes the near root only, :157-161; BASELINE.json asks
 * for dielectric spheres).  A sphere with ior[index] > 0 scatters by Snell's law with
 * Schlick's reflectance instead of ray_tracing.comp:199-207, drawing ONE random number:
 *   ud = normalize(dir);  exiting = the ray travels inside this sphere (it was refracted into
 *   it, see closest_hit);  nl = exiting ? -normal : normal;  eta = exiting ? ior : 1/ior
 *   cos_i = min(-dot(ud, nl), 1);  sin2 = max(1 - cos_i^2, 0)
 *   r0 = ((1 - ior) / (1 + ior))^2;  reflectance = r0 + (1 - r0) * (1 - cos_i)^5
 *   u = randomValue;  reflect when eta^2 * sin2 > 1 (total internal reflection) or
 *   reflectance > u:  dir' = ud - 2 dot(nl, ud) nl
 *   otherwise refract: perp = eta * (ud + cos_i * nl);  dir' = perp - sqrt(|1 - perp.perp|) nl
 * Emission and throughput follow ray_tracing.comp:209-213 with colour.rgb as the tint of both
 * outcomes (the specular colour and the texture are not used).  *refracted reports which. */
static inline v3 dielectric_scatter(v3 d, v3 n, float ior, int exiting, uint32_t *rng, int *refracted) {
  v3 ud = v_normalize(d);
  v3 nl = exiting ? V3(-n.x, -n.y, -n.z) : n;
  float eta = exiting ? ior : 1.0f / ior;
  float cos_i = -v_dot(ud, nl);
  if (cos_i > 1.0f) cos_i = 1.0f;
  float sin2 = fmaf(-cos_i, cos_i, 1.0f);
  if (sin2 < 0.0f) sin2 = 0.0f;
  int cannot_refract = (eta * eta) * sin2 > 1.0f;
  float r0 = (1.0f - ior) / (1.0f + ior);
  r0 = r0 * r0;
  float x = 1.0f - cos_i;
  float x2 = x * x;
  float reflectance = fmaf(1.0f - r0, (x2 * x2) * x, r0);
  float u = rto_random_value(rng);
  if (cannot_refract || reflectance > u) {
    float dn2 = 2.0f * v_dot(nl, ud);
    *refracted = 0;
    return V3(fmaf(-dn2, nl.x, ud.x), fmaf(-dn2, nl.y, ud.y), fmaf(-dn2, nl.z, ud.z));
  }
  v3 perp = v_scale(v_fma(nl, cos_i, ud), eta);
  float par = -sqrtf(fabsf(1.0f - v_dot(perp, perp)));
  *refracted = 1;
  return v_fma(nl, par, perp);
}

/* Debugging aid for the tests: while a log is set (rto_debug_log), every closest-hit search of
 * trace_path appends 8 floats -- origin, direction, hit index (-1: none), distance.  Meant for a
 * one-pixel list rendered on one thread. */
static float *g_debug_log = NULL;
static uint32_t g_debug_cap = 0, g_debug_n = 0;
void rto_debug_log(float *records8, uint32_t max_records) {
  g_debug_log = records8; g_debug_cap = max_records; g_debug_n = 0;
}
uint32_t rto_debug_log_count(void) { return g_debug_n; }

/* trace, ray_tracing.comp:192-222; first_id receives the first segment's hit index */
static inline v3 trace_path(const frame_ctx *fc, v3 o, v3 d, uint32_t *rng, int *first_id,
                            uint64_t *rays) {
  v3 light = V3(0, 0, 0), colour = V3(1, 1, 1);
  int skip = -1;
  for (int i = 0; (float)i < fc->max_bounce; ++i) {
    hit_t h = closest_hit(fc, o, d, skip);
    ++*rays;
    if (g_debug_log && g_debug_n < g_debug_cap) {
      float *r = g_debug_log + 8u * g_debug_n++;
      r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
      r[6] = h.hit ? (float)h.index : -1.0f; r[7] = h.dst;
    }
    if (i == 0 && first_id) *first_id = h.hit ? h.index : -1;
    if (h.hit) {
      /* colour, emission, specular: the sphere's own, or the mesh's one material */
      const float *m = (uint32_t)h.index < fc->n_spheres ? fc->spheres + 16u * (uint32_t)h.index + 4
                                                         : fc->mesh->material;
      if (fc->ior && (uint32_t)h.index < fc->n_spheres && fc->ior[h.index] > 0.0f) {
        const int exiting = skip == h.index;
        int refracted;
        o = h.point;
        d = dielectric_scatter(d, h.normal, fc->ior[h.index], exiting, rng, &refracted);
        v3 emitted = V3(m[4] * m[7], m[5] * m[7], m[6] * m[7]);
        light = V3(fmaf(emitted.x, colour.x, light.x), fmaf(emitted.y, colour.y, light.y),
                   fmaf(emitted.z, colour.z, light.z));
        colour = v_mul(colour, V3(m[0], m[1], m[2]));
        /* inside the sphere afterwards: refracted into it, or reflected back while inside */
        skip = (refracted != 0) != (exiting != 0) ? h.index : -1;
        continue;
      }
      skip = (uint32_t)h.index < fc->n_spheres ? -1 : h.index;
      v3 albedo = V3(m[0], m[1], m[2]);
      if (fc->tex && (uint32_t)h.index - fc->tex->first_prim < fc->tex->n_prims) {
        float ta, tb;
        texture_coords(fc->tex, h.point, h.normal, &ta, &tb);
        albedo = texture_sample(fc, ta * fc->tex->scale_u, tb * fc->tex->scale_v);
      }
      o = h.point;
      v3 diffuse = v_normalize(v_add(h.normal, random_direction(fc, rng)));
      float dn2 = 2.0f * v_dot(h.normal, d);                      /* reflect(I,N) = I - 2 dot(N,I) N */
      v3 specular = V3(fmaf(-dn2, h.normal.x, d.x), fmaf(-dn2, h.normal.y, d.y),
                       fmaf(-dn2, h.normal.z, d.z));
      float is_spec = (m[11] >= rto_random_value(rng)) ? 1.0f : 0.0f;
      d = v_mix(diffuse, specular, m[3] * is_spec);
      v3 emitted = V3(m[4] * m[7], m[5] * m[7], m[6] * m[7]);
      light = V3(fmaf(emitted.x, colour.x, light.x), fmaf(emitted.y, colour.y, light.y),
                 fmaf(emitted.z, colour.z, light.z));
      colour = v_mul(colour, v_mix(albedo, V3(m[8], m[9], m[10]), is_spec));
    } else {
      v3 env = environment(fc, d);
      light = V3(fmaf(env.x, colour.x, light.x), fmaf(env.y, colour.y, light.y),
                 fmaf(env.z, colour.z, light.z));
      break;
    }
  }
  return light;
}

/* screenToWorldDirection, ray_tracing.comp:224-237 (inverses hoisted: same values) */
static inline v3 screen_to_world(const frame_ctx *fc, float px, float py) {
  float clip[4] = {(2.0f * px) / fc->viewport_w - 1.0f, (2.0f * py) / fc->viewport_h - 1.0f,
                   -1.0f, 1.0f};
  float eye[4], world[4];
  mat4_mul_vec4(fc->inv_proj, clip, eye);
  eye[2] = -1.0f; eye[3] = 0.0f;
  mat4_mul_vec4(fc->inv_view, eye, world);
  return v_normalize(V3(world[0], world[1], world[2]));
}

uint8_t rto_unorm8(float v) {
  if (!(v > 0.0f)) return 0;          /* NaN and negatives */
  if (v > 1.0f) v = 1.0f;
  return (uint8_t)rintf(v * 255.0f);
}

/* main(), ray_tracing.comp:74-110, for one pixel of one frame */
/* p: where the pixel's outputs go (y*W + x for a whole image, the list position for a pixel list) */
static void shade_pixel(const frame_ctx *fc, uint32_t px, uint32_t py, size_t p, int accum_mode,
                        float *accum, uint8_t *rgba8, int32_t *ids, uint64_t *rays) {
  uint32_t pixel_index = px * fc->width + py;                      /* x*W + y, as written */
  uint32_t rng = pixel_index + (uint32_t)((int32_t)fc->frame) * 719393u;
  float fw = (float)fc->width, fh = (float)fc->height;
  v3 total = V3(0, 0, 0);
  int first_id = -1;
  for (int s = 0; (float)s < fc->spp; ++s) {
    float pt[2];
    point_in_circle(fc, &rng, pt);
    float dj[2] = {(pt[0] * fc->defocus) / fw, (pt[1] * fc->defocus) / fw};
    v3 o = v_add(fc->cam_pos, V3(dj[0] / fw, dj[1] / fh, 0.0f));
    point_in_circle(fc, &rng, pt);
    float jx = (pt[0] * fc->diverge) / fw, jy = (pt[1] * fc->diverge) / fw;
    v3 d = v_add(screen_to_world(fc, (float)px, (float)py), V3(jx, jy, 0.0f));
    total = v_add(total, trace_path(fc, o, d, &rng, s == 0 ? &first_id : NULL, rays));
  }
  v3 mean = V3(total.x / fc->spp, total.y / fc->spp, total.z / fc->spp);
  v3 g = V3(sqrtf(mean.x), sqrtf(mean.y), sqrtf(mean.z));        /* linearToGamma */
  if (ids) ids[p] = first_id;
  if (accum) {
    v3 c = accum_mode == RTO_ACCUM_LINEAR ? mean : g;
    accum[4 * p + 0] += c.x; accum[4 * p + 1] += c.y; accum[4 * p + 2] += c.z;
    accum[4 * p + 3] += 1.0f;
  }
  if (rgba8) {
    float w = 1.0f / (fc->frame + 1.0f);
    float newc[4] = {g.x, g.y, g.z, 1.0f};
    for (int k = 0; k < 4; ++k) {
      float old = k < 3 ? (float)rgba8[4 * p + k] / 255.0f : 1.0f;
      rgba8[4 * p + k] = rto_unorm8(fmaf(newc[k], w, old * (1.0f - w)));
    }
  }
}

typedef struct {
  const frame_ctx *fc;
  int accum_mode;
  float *accum; uint8_t *rgba8; int32_t *ids;
  const uint32_t *pixels; uint32_t n_pixels; /* optional pixel list */
  uint32_t *next_row;
  uint64_t rays;
} job_t;

static void *worker(void *arg) {
  job_t *j = (job_t *)arg;
  const frame_ctx *fc = j->fc;
  if (j->pixels) {
    /* listed pixels, handed out 16 at a time */
    for (;;) {
      uint32_t k0 = __atomic_fetch_add(j->next_row, 16u, __ATOMIC_RELAXED);
      if (k0 >= j->n_pixels) break;
      for (uint32_t k = k0; k < k0 + 16u && k < j->n_pixels; ++k)
        shade_pixel(fc, j->pixels[k] % fc->width, j->pixels[k] / fc->width, k, j->accum_mode,
                    j->accum, j->rgba8, j->ids, &j->rays);
    }
    return NULL;
  }
  for (;;) {
    uint32_t y = __atomic_fetch_add(j->next_row, 1u, __ATOMIC_RELAXED);
    if (y >= fc->height) break;
    for (uint32_t x = 0; x < fc->width; ++x)
      shade_pixel(fc, x, y, (size_t)y * fc->width + x, j->accum_mode, j->accum, j->rgba8, j->ids,
                  &j->rays);
  }
  return NULL;
}

int rto_render(const void *ubo272, const void *spheres64, uint32_t n_spheres, uint32_t width,
               uint32_t height, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride,
               int math_mode, int accum_mode, float *accum, uint8_t *rgba8, int32_t *ids,
               uint64_t *rays, int threads) {
  return rto_render_mesh(ubo272, spheres64, n_spheres, NULL, width, height, first_frame, n_frames,
                         frame_stride, math_mode, accum_mode, accum, rgba8, ids, rays, threads);
}

int rto_render_mesh(const void *ubo272, const void *spheres64, uint32_t n_spheres,
                    const rto_mesh *mesh, uint32_t width, uint32_t height, uint32_t first_frame,
                    uint32_t n_frames, uint32_t frame_stride, int math_mode, int accum_mode,
                    float *accum, uint8_t *rgba8, int32_t *ids, uint64_t *rays, int threads) {
  return rto_render_scene(ubo272, spheres64, n_spheres, mesh, NULL, width, height, first_frame,
                          n_frames, frame_stride, math_mode, accum_mode, accum, rgba8, ids, rays,
                          threads);
}

int rto_render_scene(const void *ubo272, const void *spheres64, uint32_t n_spheres,
                     const rto_mesh *mesh, const rto_texture *texture, uint32_t width,
                     uint32_t height, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride,
                     int math_mode, int accum_mode, float *accum, uint8_t *rgba8, int32_t *ids,
                     uint64_t *rays, int threads) {
  rto_job job;
  memset(&job, 0, sizeof job);
  job.ubo272 = ubo272; job.spheres64 = spheres64; job.n_spheres = n_spheres;
  job.mesh = mesh; job.texture = texture;
  job.width = width; job.height = height;
  job.first_frame = first_frame; job.n_frames = n_frames; job.frame_stride = frame_stride;
  job.math_mode = math_mode; job.accum_mode = accum_mode;
  job.accum = accum; job.rgba8 = rgba8; job.ids = ids; job.rays = rays; job.threads = threads;
  return rto_render_job(&job);
}

int rto_render_job(const rto_job *job) {
  if (!job) return -1;
  const void *ubo272 = job->ubo272, *spheres64 = job->spheres64;
  const uint32_t n_spheres = job->n_spheres, width = job->width, height = job->height;
  const rto_mesh *mesh = job->mesh;
  const rto_texture *texture = job->texture;
  const uint32_t first_frame = job->first_frame, n_frames = job->n_frames, frame_stride = job->frame_stride;
  const int math_mode = job->math_mode, accum_mode = job->accum_mode;
  float *accum = job->accum; uint8_t *rgba8 = job->rgba8; int32_t *ids = job->ids;
  uint64_t *rays = job->rays;
  int threads = job->threads;
  if (!ubo272 || (!spheres64 && n_spheres) || !width || !height) return -1;
  if (job->n_pixels && !job->pixels) return -1;
  for (uint32_t k = 0; k < job->n_pixels; ++k)
    if (job->pixels[k] >= (uint64_t)width * height) return -1;
  if (texture && (!texture->rgba8 || !texture->width || !texture->height || texture->width > 32768u ||
                  texture->height > 32768u))
    return -1;
  if (mesh && mesh->n_triangles == 0) mesh = NULL;
  if (mesh && (!mesh->vertices || !mesh->indices || !mesh->material)) return -1;
  if (mesh)
    for (uint32_t k = 0; k < 3u * mesh->n_triangles; ++k)
      if (mesh->indices[k] >= mesh->n_vertices) return -1;
  if (threads <= 0) threads = 1;
  if (threads > 256) threads = 256;
  float u[68];
  memcpy(u, ubo272, 272);
  frame_ctx fc;
  rto_mat4_inverse(&u[0], fc.inv_view);
  rto_mat4_inverse(&u[16], fc.inv_proj);
  fc.viewport_w = u[32]; fc.viewport_h = u[33];
  fc.cam_pos = V3(u[36], u[37], u[38]);
  fc.spp = u[40]; fc.max_bounce = u[41];
  fc.ground = V3(u[48], u[49], u[50]);
  fc.horizon = V3(u[52], u[53], u[54]);
  fc.zenith = V3(u[56], u[57], u[58]);
  fc.sun_pos = V3(u[60], u[61], u[62]);
  fc.sun_focus = u[64]; fc.sun_intensity = u[65]; fc.defocus = u[66]; fc.diverge = u[67];
  fc.spheres = (const float *)spheres64; fc.n_spheres = n_spheres;
  fc.mesh = mesh;
  fc.tex = texture;
  fc.ior = job->ior;
  for (int i = 0; i < 256; ++i)
    fc.tex_lut[i] = texture && texture->srgb ? rto_srgb_to_linear((uint8_t)i) : (float)i / 255.0f;
  fc.width = width; fc.height = height;
  fc.mo = math_mode == RTO_MATH_LIBM ? &k_libm : &k_portable;

  pthread_t tid[256];
  job_t jobs[256];
  for (uint32_t f = 0; f < n_frames; ++f) {
    fc.frame = (float)(first_frame + f * frame_stride);
    uint32_t next_row = 0;
    for (int t = 0; t < threads; ++t) {
      jobs[t].fc = &fc; jobs[t].accum_mode = accum_mode;
      jobs[t].accum = accum; jobs[t].rgba8 = rgba8; jobs[t].ids = ids;
      jobs[t].pixels = job->n_pixels ? job->pixels : NULL; jobs[t].n_pixels = job->n_pixels;
      jobs[t].next_row = &next_row; jobs[t].rays = 0;
    }
    for (int t = 1; t < threads; ++t) pthread_create(&tid[t], NULL, worker, &jobs[t]);
    worker(&jobs[0]);
    for (int t = 1; t < threads; ++t) pthread_join(tid[t], NULL);
    if (rays)
      for (int t = 0; t < threads; ++t) *rays += jobs[t].rays;
  }
  return 0;
}

void rto_texture_sample(const rto_texture *texture, float u, float v, float out_rgb[3]) {
  static _Thread_local frame_ctx fc;
  fc.tex = texture;
  for (int i = 0; i < 256; ++i)
    fc.tex_lut[i] = texture->srgb ? rto_srgb_to_linear((uint8_t)i) : (float)i / 255.0f;
  v3 c = texture_sample(&fc, u, v);
  out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
}

void rto_dielectric_scatter(const float dir[3], const float normal[3], float ior, int exiting,
                            uint32_t *rng, float out_dir[3], int *out_refracted) {
  v3 d = dielectric_scatter(V3(dir[0], dir[1], dir[2]), V3(normal[0], normal[1], normal[2]), ior,
                            exiting, rng, out_refracted);
  out_dir[0] = d.x; out_dir[1] = d.y; out_dir[2] = d.z;
}

void rto_resolve_rgba8(const float *accum, uint32_t n_pixels, int accum_mode, uint8_t *rgba8) {
  for (uint32_t p = 0; p < n_pixels; ++p) {
    float n = accum[4 * p + 3];
    for (int k = 0; k < 3; ++k) {
      float v = accum[4 * p + k] / n;
      if (accum_mode == RTO_ACCUM_LINEAR) v = sqrtf(v);
      rgba8[4 * p + k] = rto_unorm8(v);
    }
    rgba8[4 * p + 3] = 255;
  }
}
