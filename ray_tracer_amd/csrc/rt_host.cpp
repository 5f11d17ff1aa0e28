// Host half of the contract (include/rt_host.h): the reference's camera and default
// scene, restated without glm.  Compiled with -ffp-contract=off; all arithmetic is
// fp32 like glm's, trigonometry comes from the host libm like glm's std::cos/sin/tan.
#include "../../include/rt_host.h"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "rt_hostmath.h"

namespace {

struct Quat {
  float w, x, y, z;
};

// glm::quat(vec3 eulerAngles) with angles (-pitch, -yaw, 0): camera.cpp:48-50
Quat orientation_of(const rt_camera &cam) {
  const float hx = -cam.pitch * 0.5f, hy = -cam.yaw * 0.5f, hz = 0.0f * 0.5f;
  const float cx = std::cos(hx), cy = std::cos(hy), cz = std::cos(hz);
  const float sx = std::sin(hx), sy = std::sin(hy), sz = std::sin(hz);
  Quat q;
  q.w = cx * cy * cz + sx * sy * sz;
  q.x = sx * cy * cz - cx * sy * sz;
  q.y = cx * sy * cz + sx * cy * sz;
  q.z = cx * cy * sz - sx * sy * cz;
  return q;
}

// glm::rotate(quat, vec3): v + 2 * (w * (qv x v) + qv x (qv x v))
void rotate_by(const Quat &q, float vx, float vy, float vz, float out[3]) {
  const float ux = q.y * vz - vy * q.z, uy = q.z * vx - vz * q.x, uz = q.x * vy - vx * q.y;
  const float wx = q.y * uz - uy * q.z, wy = q.z * ux - uz * q.x, wz = q.x * uy - ux * q.y;
  out[0] = vx + ((ux * q.w) + wx) * 2.0f;
  out[1] = vy + ((uy * q.w) + wy) * 2.0f;
  out[2] = vz + ((uz * q.w) + wz) * 2.0f;
}

float pan_factor(float extent) {
  const float v = std::fmin(extent / 1000.0f, 2.4f);
  return 0.0366f * (v * v) - 0.1778f * v + 0.3021f;
}

}  // namespace

extern "C" {

void rt_camera_init(rt_camera *cam) {
  cam->fov = 45.0f;
  cam->near_plane = 0.1f;
  cam->far_plane = 1000.0f;
  cam->pitch = 0.0f;
  cam->yaw = 0.0f;
  cam->aspect_ratio = 1.778f;
  cam->distance = 10.0f;
  cam->viewport_width = 800.0f;
  cam->viewport_height = 600.0f;
}

void rt_camera_create(float start_fov, float start_aspect_ratio, float start_near, float start_far,
                      rt_camera *out_camera) {
  rt_camera_init(out_camera);
  out_camera->fov = start_fov;
  out_camera->aspect_ratio = start_aspect_ratio;
  out_camera->near_plane = start_near;
  out_camera->far_plane = start_far;
}

void rt_camera_orientation(const rt_camera *cam, float out_wxyz[4]) {
  const Quat q = orientation_of(*cam);
  out_wxyz[0] = q.w;
  out_wxyz[1] = q.x;
  out_wxyz[2] = q.y;
  out_wxyz[3] = q.z;
}

void rt_camera_up(const rt_camera *cam, float out3[3]) {
  rotate_by(orientation_of(*cam), 0.0f, 1.0f, 0.0f, out3);
}
void rt_camera_right(const rt_camera *cam, float out3[3]) {
  rotate_by(orientation_of(*cam), 1.0f, 0.0f, 0.0f, out3);
}
void rt_camera_forward(const rt_camera *cam, float out3[3]) {
  rotate_by(orientation_of(*cam), 0.0f, 0.0f, 1.0f, out3);
}

float rt_camera_rotation_speed(const rt_camera *) { return 0.8f; }

float rt_camera_zoom_speed(const rt_camera *cam) {
  float d = std::fmax(cam->distance * 0.2f, 0.0f);
  return std::fmin(d * d, 100.0f);
}

void rt_camera_pan_speed(const rt_camera *cam, float out2[2]) {
  out2[0] = pan_factor(cam->viewport_width);
  out2[1] = pan_factor(cam->viewport_height);
}

void rt_camera_rotate(rt_camera *cam, float delta_x, float delta_y) {
  float up[3];
  rt_camera_up(cam, up);
  const float yaw_sign = up[1] < 0.0f ? 1.0f : -1.0f;
  const float speed = rt_camera_rotation_speed(cam);
  cam->yaw += yaw_sign * delta_x * speed;
  cam->pitch += delta_y * speed;
}

void rt_camera_projection_matrix(rt_camera *cam, float p[16]) {
  cam->aspect_ratio = cam->viewport_width / cam->viewport_height;
  const float fovy = cam->fov * 0.01745329251994329576923690768489f;  // glm::radians
  const float t = std::tan(fovy / 2.0f);
  const float n = cam->near_plane, f = cam->far_plane;
  std::memset(p, 0, 16 * sizeof(float));
  p[0] = 1.0f / (cam->aspect_ratio * t);
  p[5] = 1.0f / t;
  p[10] = (f + n) / (f - n);
  p[11] = 1.0f;
  p[14] = -(2.0f * f * n) / (f - n);
}

void rt_camera_view_matrix(const rt_camera *cam, float view[16]) {
  const Quat q = orientation_of(*cam);
  const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
  const float xz = q.x * q.z, xy = q.x * q.y, yz = q.y * q.z;
  const float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
  float rot[16] = {0};
  rot[0] = 1.0f - 2.0f * (yy + zz);
  rot[1] = 2.0f * (xy + wz);
  rot[2] = 2.0f * (xz - wy);
  rot[4] = 2.0f * (xy - wz);
  rot[5] = 1.0f - 2.0f * (xx + zz);
  rot[6] = 2.0f * (yz + wx);
  rot[8] = 2.0f * (xz + wy);
  rot[9] = 2.0f * (yz - wx);
  rot[10] = 1.0f - 2.0f * (xx + yy);
  rot[15] = 1.0f;
  // translate(I, vec3(0)) * rot == rot; the reference inverts that product
  rt::mat4_inverse(rot, view);
}

void rt_ubo_set_camera(rt_camera *cam, rt_ubo *ubo) {
  rt_camera_view_matrix(cam, ubo->view);
  rt_camera_projection_matrix(cam, ubo->projection);
  ubo->viewport_size[0] = cam->viewport_width;
  ubo->viewport_size[1] = cam->viewport_height;
  ubo->viewport_size[2] = 0.0f;
  ubo->viewport_size[3] = 0.0f;
  for (int k = 0; k < 4; ++k) ubo->camera_position[k] = 0.0f;
}

void rt_default_ubo(rt_camera *cam, rt_ubo *out) {
  std::memset(out, 0, sizeof *out);
  rt_camera local;
  if (!cam) {
    rt_camera_create(90.0f, 1.0f, 0.01f, 10000.0f, &local);
    cam = &local;
  }
  rt_ubo_set_camera(cam, out);
  out->render_settings[0] = 50.0f;
  out->render_settings[1] = 25.0f;
  const float ground[4] = {0.35f, 0.3f, 0.35f, 1.0f};
  const float horizon[4] = {1.0f, 1.0f, 1.0f, 1.0f};
  const float zenith[4] = {0.078f, 0.36f, 0.72f, 1.0f};
  std::memcpy(out->ground_colour, ground, sizeof ground);
  std::memcpy(out->sky_colour_horizon, horizon, sizeof horizon);
  std::memcpy(out->sky_colour_zenith, zenith, sizeof zenith);
  const float inv_len = 1.0f / std::sqrt(4.0f);  // normalize(vec4(1.0))
  for (int k = 0; k < 4; ++k) out->sun_position[k] = 1.0f * inv_len;
  out->sun_focus = 1.0f;
  out->sun_intensity = 0.0f;
  out->defocus_strength = 0.0f;
  out->diverge_strength = 1.0f;
}

uint32_t rt_default_scene(rt_sphere out3[3]) {
  std::memset(out3, 0, 3 * sizeof(rt_sphere));
  auto put = [&](int i, float x, float y, float z, float r, float cr, float cg, float cb,
                 float smooth, float spec_prob) {
    rt_sphere &s = out3[i];
    s.position[0] = x; s.position[1] = y; s.position[2] = z;
    s.radius = r;
    s.material.colour[0] = cr; s.material.colour[1] = cg; s.material.colour[2] = cb;
    s.material.colour[3] = smooth;
    s.material.specular_colour[0] = s.material.specular_colour[1] = 1.0f;
    s.material.specular_colour[2] = 1.0f;
    s.material.specular_colour[3] = spec_prob;
  };
  put(0, 0.0f, 0.0f, -5.0f, 1.0f, 0.5f, 0.5f, 0.5f, 1.0f, 0.5f);
  put(1, 3.0f, 0.0f, -5.0f, 1.0f, 0.8f, 0.2f, 0.2f, 0.5f, 0.0f);
  put(2, 0.0f, -101.0f, -5.0f, 100.0f, 0.2f, 0.8f, 0.05f, 0.0f, 0.0f);
  return 3;
}

static uint32_t pcg_hash(uint32_t v) {  // one nextRandom step from state v
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t r = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
  return (r >> 22) ^ r;
}

void rt_brick_texture(uint32_t width, uint32_t height, uint8_t *out) {
  if (!out || !width || !height) return;
  const uint32_t courses = 8, columns = 4;  // per image: 8 courses of 4 bricks
  for (uint32_t y = 0; y < height; ++y) {
    const uint32_t course = (uint32_t)((uint64_t)y * courses / height);
    // position inside the course, in 1/4096 of its height
    const uint32_t fy = (uint32_t)(((uint64_t)y * courses * 4096u / height) & 4095u);
    for (uint32_t x = 0; x < width; ++x) {
      // horizontal position in 1/4096 of a brick, odd courses shifted by half a brick
      uint64_t hx = (uint64_t)x * columns * 4096u / width + ((course & 1u) ? 2048u : 0u);
      const uint32_t column = (uint32_t)((hx >> 12) % columns);
      const uint32_t fx = (uint32_t)(hx & 4095u);
      // a joint is 1/16 course high; a brick is two courses long, so 1/32 of its length
      const bool mortar = fy < 256u || fx < 128u;
      const uint32_t grain = pcg_hash(y * 73856093u ^ x * 19349663u) & 7u;  // 0..7
      int r, g, b;
      if (mortar) {
        r = g = b = 176;
      } else {
        const uint32_t tint = pcg_hash(course * 131u + column * 7u + 1u);
        r = 150 + (int)(tint & 31u);
        g = 62 + (int)((tint >> 5) & 15u);
        b = 48 + (int)((tint >> 9) & 15u);
      }
      uint8_t *px = out + 4u * ((size_t)y * width + x);
      px[0] = (uint8_t)(r + (int)grain - 4);
      px[1] = (uint8_t)(g + (int)grain - 4);
      px[2] = (uint8_t)(b + (int)grain - 4);
      px[3] = 255;
    }
  }
}

static uint32_t grid_scene(uint32_t n, uint32_t seed, rt_sphere *out, float *ior_out);

uint32_t rt_grid_scene(uint32_t n, uint32_t seed, rt_sphere *out) { return grid_scene(n, seed, out, nullptr); }

uint32_t rt_grid_scene_glass(uint32_t n, uint32_t seed, rt_sphere *out, float *ior_out) {
  if (!ior_out) return 0;
  return grid_scene(n, seed, out, ior_out);
}

// ior_out == null: the third material class is the glossy coat; else glass (see rt_host.h)
static uint32_t grid_scene(uint32_t n, uint32_t seed, rt_sphere *out, float *ior_out) {
  if (!out || n == 0) return 0;
  if (ior_out) std::memset(ior_out, 0, (size_t)n * sizeof(float));
  uint32_t state = seed;
  auto draw = [&state]() -> float {  // the shader's nextRandom / randomValue
    state = state * 747796405u + 2891336453u;
    uint32_t r = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    r = (r >> 22) ^ r;
    return (float)r * 0x1p-32f;
  };
  std::memset(out, 0, (size_t)n * sizeof(rt_sphere));
  out[0].position[0] = 0.0f; out[0].position[1] = -1002.0f; out[0].position[2] = -60.0f;
  out[0].radius = 1000.0f;
  out[0].material.colour[0] = out[0].material.colour[1] = out[0].material.colour[2] = 0.5f;
  out[0].material.specular_colour[0] = out[0].material.specular_colour[1] = 1.0f;
  out[0].material.specular_colour[2] = 1.0f;
  uint32_t side = 1;
  while ((uint64_t)side * side < (uint64_t)(n > 1 ? n - 1 : 1)) ++side;
  const double cell = 100.0 / side;
  for (uint32_t i = 1; i < n; ++i) {
    rt_sphere &s = out[i];
    const uint32_t gx = (i - 1) % side, gz = (i - 1) / side;
    double r = (double)draw() * 0.3 + 0.2;
    if (cell < 1.0 && r > 0.45 * cell) r = 0.45 * cell;
    const double x = -50.0 + (gx + 0.5 + ((double)draw() - 0.5) * 0.6) * cell;
    const double z = -110.0 + (gz + 0.5 + ((double)draw() - 0.5) * 0.6) * cell;
    const double dist = 1000.0 + r;
    const double under = dist * dist - x * x - (z + 60.0) * (z + 60.0);
    const double y = -1002.0 + std::sqrt(under > 0.0 ? under : 0.0);
    s.position[0] = (float)x; s.position[1] = (float)y; s.position[2] = (float)z;
    s.radius = (float)r;
    const float kind = draw();
    const float cr = draw(), cg = draw(), cb = draw();
    s.material.colour[0] = cr; s.material.colour[1] = cg; s.material.colour[2] = cb;
    if (kind < 0.60f) {
      s.material.colour[3] = 0.0f;
      s.material.specular_colour[0] = s.material.specular_colour[1] = 1.0f;
      s.material.specular_colour[2] = 1.0f;
      s.material.specular_colour[3] = 0.0f;
    } else if (kind < 0.85f) {
      s.material.colour[3] = 0.5f + 0.5f * draw();
      s.material.specular_colour[0] = 0.5f + 0.5f * cr;
      s.material.specular_colour[1] = 0.5f + 0.5f * cg;
      s.material.specular_colour[2] = 0.5f + 0.5f * cb;
      s.material.specular_colour[3] = 1.0f;
    } else {
      s.material.colour[3] = 1.0f;
      s.material.specular_colour[0] = s.material.specular_colour[1] = 1.0f;
      s.material.specular_colour[2] = 1.0f;
      s.material.specular_colour[3] = 0.15f + 0.35f * draw();
      if (ior_out) {
        ior_out[i] = 1.5f;
        s.material.colour[0] = 0.75f + 0.25f * cr;
        s.material.colour[1] = 0.75f + 0.25f * cg;
        s.material.colour[2] = 0.75f + 0.25f * cb;
      }
    }
    if (i % 50 == 0) {
      s.material.emission_colour[0] = cr; s.material.emission_colour[1] = cg;
      s.material.emission_colour[2] = cb; s.material.emission_colour[3] = 4.0f;
    }
  }
  return n;
}

int rt_write_ppm(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height,
                 int flip_y) {
  if (!path || !rgba8 || !width || !height) {
    errno = EINVAL;
    return -1;
  }
  FILE *f = std::fopen(path, "wb");
  if (!f) return -1;
  std::fprintf(f, "P6\n%u %u\n255\n", width, height);
  std::vector<uint8_t> row((size_t)width * 3);
  for (uint32_t y = 0; y < height; ++y) {
    const uint32_t sy = flip_y ? height - 1 - y : y;
    const uint8_t *src = rgba8 + (size_t)sy * width * 4;
    for (uint32_t x = 0; x < width; ++x) {
      row[3 * (size_t)x + 0] = src[4 * (size_t)x + 0];
      row[3 * (size_t)x + 1] = src[4 * (size_t)x + 1];
      row[3 * (size_t)x + 2] = src[4 * (size_t)x + 2];
    }
    if (std::fwrite(row.data(), 1, row.size(), f) != row.size()) {
      std::fclose(f);
      return -1;
    }
  }
  return std::fclose(f) == 0 ? 0 : -1;
}

}  // extern "C"

namespace {

uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
  return crc;
}

void put_be32(std::vector<uint8_t> &v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16));
  v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

bool write_chunk(FILE *f, const char tag[4], const std::vector<uint8_t> &data) {
  std::vector<uint8_t> head;
  put_be32(head, (uint32_t)data.size());
  uint32_t crc = crc32_update(0xffffffffu, (const uint8_t *)tag, 4);
  crc = crc32_update(crc, data.data(), data.size()) ^ 0xffffffffu;
  std::vector<uint8_t> tail;
  put_be32(tail, crc);
  return std::fwrite(head.data(), 1, 4, f) == 4 && std::fwrite(tag, 1, 4, f) == 4 &&
         (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size()) &&
         std::fwrite(tail.data(), 1, 4, f) == 4;
}

}  // namespace

extern "C" int rt_write_png(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height,
                            int flip_y) {
  if (!path || !rgba8 || !width || !height) {
    errno = EINVAL;
    return -1;
  }
  // raw scanlines: filter byte 0 + RGB
  const size_t stride = 1 + (size_t)width * 3;
  std::vector<uint8_t> raw(stride * height);
  for (uint32_t y = 0; y < height; ++y) {
    const uint32_t sy = flip_y ? height - 1 - y : y;
    const uint8_t *src = rgba8 + (size_t)sy * width * 4;
    uint8_t *dst = raw.data() + stride * y;
    dst[0] = 0;
    for (uint32_t x = 0; x < width; ++x) {
      dst[1 + 3 * (size_t)x + 0] = src[4 * (size_t)x + 0];
      dst[1 + 3 * (size_t)x + 1] = src[4 * (size_t)x + 1];
      dst[1 + 3 * (size_t)x + 2] = src[4 * (size_t)x + 2];
    }
  }
  // zlib stream of stored blocks (at most 65535 bytes each) + Adler-32
  std::vector<uint8_t> z;
  z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
  z.push_back(0x78); z.push_back(0x01);
  uint32_t a = 1, b = 0;
  for (size_t pos = 0; pos < raw.size();) {
    const size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n == raw.size() ? 1 : 0);
    z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
    z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    for (size_t i = 0; i < n; ++i) {
      a = (a + raw[pos + i]) % 65521u;
      b = (b + a) % 65521u;
    }
    pos += n;
  }
  put_be32(z, (b << 16) | a);
  FILE *f = std::fopen(path, "wb");
  if (!f) return -1;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, width);
  put_be32(ihdr, height);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  bool ok = std::fwrite(sig, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr) && write_chunk(f, "IDAT", z) &&
            write_chunk(f, "IEND", std::vector<uint8_t>());
  ok = (std::fclose(f) == 0) && ok;
  return ok ? 0 : -1;
}
