// Data formats on the way into the path tracer: Wavefront OBJ meshes and baseline JPEG
// images.  The reference links assimp and stb_image for these (CMakeLists.txt:12,
// main.cpp:29-33) but never calls either on the compute path; neither library exists in this
// image, so both readers are written out here.  Host code only.
#include "../../include/rt_host.h"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/stat.h>

namespace {

thread_local std::string g_ingest_error;

int ingest_fail(const char *msg) {
  g_ingest_error = msg;
  return -1;
}

// regular files of at most 4 GiB (a directory opens with fopen and reports a size of LONG_MAX)
bool read_file(const char *path, std::vector<uint8_t> &out) {
  struct stat st;
  if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return false;
  FILE *f = std::fopen(path, "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (n < 0 || (unsigned long long)n > (4ull << 30)) {
    std::fclose(f);
    return false;
  }
  try {
    out.resize((size_t)n);
  } catch (...) {
    std::fclose(f);
    return false;
  }
  size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
  std::fclose(f);
  return got == (size_t)n;
}

// ---------------------------------------------------------------------------- OBJ
// one face corner "v", "v/vt", "v//vn" or "v/vt/vn": only the position index is used
bool corner_index(const char *&p, long n_vertices, uint32_t &out) {
  char *end = nullptr;
  long v = std::strtol(p, &end, 10);
  if (end == p) return false;
  p = end;
  while (*p && *p != ' ' && *p != '\t' && *p != '\r') ++p;  // skip /vt/vn
  if (v < 0) v = n_vertices + v;  // relative to the vertices read so far
  else v = v - 1;
  if (v < 0 || v >= n_vertices) return false;
  out = (uint32_t)v;
  return true;
}

// ---------------------------------------------------------------------------- JPEG
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
  bool present = false;
  uint8_t values[256];
  int min_code[17], max_code[18], first_value[17];
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int pred = 0;
  int plane_w = 0, plane_h = 0;
  std::vector<uint8_t> plane;
};

struct Decoder {
  const uint8_t *data;
  size_t size, pos = 0;
  uint16_t quant[4][64];
  bool have_quant[4] = {false, false, false, false};
  Huffman dc[4], ac[4];
  int width = 0, height = 0, n_comp = 0;
  Component comp[3];
  int restart_interval = 0;
  // entropy-coded segment reader
  uint32_t bit_buf = 0;
  int bit_count = 0;
  bool hit_marker = false;

  // header reads stop at `limit` (the end of the current segment, or of the file): a read past
  // it returns 0 and sets `overrun`, which the segment loop turns into an error
  size_t limit = 0;
  bool overrun = false;

  bool need(size_t n) const { return pos + n <= size; }
  int u8() {
    if (pos + 1 > limit) {
      overrun = true;
      return 0;
    }
    return data[pos++];
  }
  int u16() {
    if (pos + 2 > limit) {
      overrun = true;
      return 0;
    }
    int v = (data[pos] << 8) | data[pos + 1];
    pos += 2;
    return v;
  }

  int next_bit() {
    if (bit_count == 0) {
      int byte = 0;
      if (!hit_marker && pos < size) {
        byte = data[pos++];
        if (byte == 0xFF) {
          int next = pos < size ? data[pos] : 0xD9;
          if (next == 0x00) {
            ++pos;  // stuffed zero
          } else {
            hit_marker = true;  // feed zeros until the caller deals with the marker
            --pos;
            byte = 0;
          }
        }
      }
      bit_buf = (uint32_t)byte;
      bit_count = 8;
    }
    --bit_count;
    return (int)((bit_buf >> bit_count) & 1u);
  }

  int receive(int n) {
    int v = 0;
    for (int i = 0; i < n; ++i) v = (v << 1) | next_bit();
    return v;
  }

  static int extend(int v, int n) { return n && v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

  int decode_symbol(const Huffman &h) {
    int code = next_bit();
    for (int len = 1; len <= 16; ++len) {
      if (h.max_code[len] >= 0 && code <= h.max_code[len] && code >= h.min_code[len])
        return h.values[h.first_value[len] + code - h.min_code[len]];
      code = (code << 1) | next_bit();
    }
    return -1;
  }
};

void build_huffman(Huffman &h, const uint8_t counts[16], const uint8_t *values, int n_values) {
  std::memcpy(h.values, values, (size_t)n_values);
  int code = 0, k = 0;
  for (int len = 1; len <= 16; ++len) {
    h.first_value[len] = k;
    h.min_code[len] = code;
    code += counts[len - 1];
    k += counts[len - 1];
    h.max_code[len] = counts[len - 1] ? code - 1 : -1;
    code <<= 1;
  }
  h.max_code[17] = 0x7fffffff;
  h.present = true;
}

// separable 8x8 inverse DCT in double precision, level shift and clamp
void idct_block(const int coef[64], uint8_t *dst, int stride) {
  static double basis[8][8];
  static bool ready = false;
  if (!ready) {
    for (int x = 0; x < 8; ++x)
      for (int u = 0; u < 8; ++u)
        basis[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * M_PI / 16.0);
    ready = true;
  }
  double tmp[64];
  for (int v = 0; v < 8; ++v)
    for (int x = 0; x < 8; ++x) {
      double s = 0.0;
      for (int u = 0; u < 8; ++u) s += basis[x][u] * coef[8 * v + u];
      tmp[8 * v + x] = s;
    }
  for (int x = 0; x < 8; ++x)
    for (int y = 0; y < 8; ++y) {
      double s = 0.0;
      for (int v = 0; v < 8; ++v) s += basis[y][v] * tmp[8 * v + x];
      long r = std::lround(s + 128.0);
      dst[y * stride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

int decode_scan(Decoder &D) {
  int hmax = 1, vmax = 1;
  for (int c = 0; c < D.n_comp; ++c) {
    hmax = D.comp[c].h > hmax ? D.comp[c].h : hmax;
    vmax = D.comp[c].v > vmax ? D.comp[c].v : vmax;
  }
  const int mcus_x = (D.width + 8 * hmax - 1) / (8 * hmax);
  const int mcus_y = (D.height + 8 * vmax - 1) / (8 * vmax);
  for (int c = 0; c < D.n_comp; ++c) {
    Component &C = D.comp[c];
    if (!D.have_quant[C.tq]) return ingest_fail("jpeg: missing quantisation table");
    if (!D.dc[C.td].present || !D.ac[C.ta].present) return ingest_fail("jpeg: missing Huffman table");
    C.plane_w = mcus_x * C.h * 8;
    C.plane_h = mcus_y * C.v * 8;
    C.plane.assign((size_t)C.plane_w * C.plane_h, 0);
    C.pred = 0;
  }
  int until_restart = D.restart_interval;
  int expected_rst = 0;
  for (int my = 0; my < mcus_y; ++my)
    for (int mx = 0; mx < mcus_x; ++mx) {
      if (D.restart_interval && until_restart == 0) {
        // byte-align, expect RSTn, reset the predictors
        D.bit_count = 0;
        D.hit_marker = false;
        while (D.pos + 1 < D.size && !(D.data[D.pos] == 0xFF && D.data[D.pos + 1] >= 0xD0 &&
                                       D.data[D.pos + 1] <= 0xD7))
          ++D.pos;
        if (D.pos + 1 >= D.size) return ingest_fail("jpeg: restart marker not found");
        if (D.data[D.pos + 1] != 0xD0 + expected_rst) return ingest_fail("jpeg: restart markers out of order");
        D.pos += 2;
        expected_rst = (expected_rst + 1) & 7;
        for (int c = 0; c < D.n_comp; ++c) D.comp[c].pred = 0;
        until_restart = D.restart_interval;
      }
      for (int c = 0; c < D.n_comp; ++c) {
        Component &C = D.comp[c];
        for (int by = 0; by < C.v; ++by)
          for (int bx = 0; bx < C.h; ++bx) {
            int coef[64] = {0};
            int t = D.decode_symbol(D.dc[C.td]);
            if (t < 0 || t > 11) return ingest_fail("jpeg: bad DC code");
            C.pred += Decoder::extend(D.receive(t), t);
            if (C.pred < -32768 || C.pred > 32767) return ingest_fail("jpeg: DC value out of range");
            coef[0] = C.pred * D.quant[C.tq][0];
            for (int k = 1; k < 64;) {
              int rs = D.decode_symbol(D.ac[C.ta]);
              if (rs < 0) return ingest_fail("jpeg: bad AC code");
              int run = rs >> 4, size = rs & 15;
              if (size == 0) {
                if (run != 15) break;  // end of block
                k += 16;
                continue;
              }
              k += run;
              if (k > 63) return ingest_fail("jpeg: coefficient index out of range");
              coef[kZigzag[k]] = Decoder::extend(D.receive(size), size) * D.quant[C.tq][k];
              ++k;
            }
            uint8_t *dst = C.plane.data() + (size_t)((my * C.v + by) * 8) * C.plane_w + (mx * C.h + bx) * 8;
            idct_block(coef, dst, C.plane_w);
          }
      }
      if (D.restart_interval) --until_restart;
    }
  return 0;
}

// sample of a subsampled plane at full-resolution pixel (x, y): chroma samples sit at the
// centre of the luma samples they cover; linear interpolation, edges clamped
double plane_sample(const Component &C, int x, int y, int hmax, int vmax) {
  if (C.h == hmax && C.v == vmax) return C.plane[(size_t)y * C.plane_w + x];
  const double fx = (x + 0.5) * C.h / hmax - 0.5, fy = (y + 0.5) * C.v / vmax - 0.5;
  int x0 = (int)std::floor(fx), y0 = (int)std::floor(fy);
  const double ax = fx - x0, ay = fy - y0;
  auto at = [&](int xx, int yy) {
    xx = xx < 0 ? 0 : (xx >= C.plane_w ? C.plane_w - 1 : xx);
    yy = yy < 0 ? 0 : (yy >= C.plane_h ? C.plane_h - 1 : yy);
    return (double)C.plane[(size_t)yy * C.plane_w + xx];
  };
  const double top = at(x0, y0) * (1 - ax) + at(x0 + 1, y0) * ax;
  const double bot = at(x0, y0 + 1) * (1 - ax) + at(x0 + 1, y0 + 1) * ax;
  return top * (1 - ay) + bot * ay;
}

uint8_t clamp8(double v) {
  long r = std::lround(v);
  return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

}  // namespace

extern "C" {

const char *rt_ingest_error(void) { return g_ingest_error.c_str(); }

void rt_free(void *p) { std::free(p); }

static int rt_load_obj_impl(const char *path, float **out_vertices, uint32_t *out_n_vertices, uint32_t **out_indices,
                uint32_t *out_n_triangles) {
  if (!path || !out_vertices || !out_n_vertices || !out_indices || !out_n_triangles)
    return ingest_fail("rt_load_obj: null argument");
  std::vector<uint8_t> bytes;
  if (!read_file(path, bytes)) return ingest_fail("rt_load_obj: cannot read file");
  bytes.push_back('\n');
  bytes.push_back(0);
  std::vector<float> v;
  std::vector<uint32_t> idx;
  char *p = (char *)bytes.data();
  long line_no = 0;
  while (*p) {
    ++line_no;
    const char *line = p;
    while (*p && *p != '\n') ++p;
    char *next = *p ? p + 1 : p;
    *p = 0;  // the number parsers below must not run on into the following line
    while (*line == ' ' || *line == '\t') ++line;
    if (line[0] == 'v' && (line[1] == ' ' || line[1] == '\t')) {
      const char *q = line + 1;
      for (int k = 0; k < 3; ++k) {
        char *end = nullptr;
        float f = std::strtof(q, &end);
        if (end == q) {
          static thread_local char msg[96];
          std::snprintf(msg, sizeof msg, "rt_load_obj: line %ld: vertex needs three numbers", line_no);
          return ingest_fail(msg);
        }
        v.push_back(f);
        q = end;
      }
    } else if (line[0] == 'f' && (line[1] == ' ' || line[1] == '\t')) {
      const char *q = line + 1;
      std::vector<uint32_t> corners;
      for (;;) {
        while (*q == ' ' || *q == '\t') ++q;
        if (*q == '\r' || *q == 0 || *q == '#') break;
        uint32_t c;
        if (!corner_index(q, (long)(v.size() / 3), c)) {
          static thread_local char msg[96];
          std::snprintf(msg, sizeof msg, "rt_load_obj: line %ld: bad vertex reference", line_no);
          return ingest_fail(msg);
        }
        corners.push_back(c);
      }
      if (corners.size() < 3) {
        static thread_local char msg[96];
        std::snprintf(msg, sizeof msg, "rt_load_obj: line %ld: face with fewer than three corners", line_no);
        return ingest_fail(msg);
      }
      for (size_t k = 1; k + 1 < corners.size(); ++k) {  // fan
        idx.push_back(corners[0]);
        idx.push_back(corners[k]);
        idx.push_back(corners[k + 1]);
      }
    }
    p = next;
  }
  float *vo = (float *)std::malloc(v.size() * sizeof(float) + 1);
  uint32_t *io = (uint32_t *)std::malloc(idx.size() * sizeof(uint32_t) + 1);
  if (!vo || !io) {
    std::free(vo);
    std::free(io);
    return ingest_fail("rt_load_obj: out of memory");
  }
  if (!v.empty()) std::memcpy(vo, v.data(), v.size() * sizeof(float));
  if (!idx.empty()) std::memcpy(io, idx.data(), idx.size() * sizeof(uint32_t));
  *out_vertices = vo;
  *out_n_vertices = (uint32_t)(v.size() / 3);
  *out_indices = io;
  *out_n_triangles = (uint32_t)(idx.size() / 3);
  return 0;
}

static int rt_decode_jpeg_impl(const void *bytes, size_t size, int flip_y, uint8_t **out_rgba8, uint32_t *out_width,
                   uint32_t *out_height) {
  if (!bytes || !out_rgba8 || !out_width || !out_height) return ingest_fail("rt_decode_jpeg: null argument");
  Decoder D;
  D.data = (const uint8_t *)bytes;
  D.size = size;
  if (size < 4 || D.data[0] != 0xFF || D.data[1] != 0xD8) return ingest_fail("jpeg: no SOI marker");
  D.pos = 2;
  bool decoded = false;
  while (!decoded) {
    D.limit = D.size;
    if (!D.need(4)) return ingest_fail("jpeg: truncated before the scan");
    if (D.data[D.pos] != 0xFF) return ingest_fail("jpeg: marker expected");
    while (D.pos < D.size && D.data[D.pos] == 0xFF) ++D.pos;  // fill bytes
    if (!D.need(3)) return ingest_fail("jpeg: truncated marker");
    const int marker = D.u8();
    if (marker == 0xD9) return ingest_fail("jpeg: EOI before any scan");
    const int len = D.u16();
    if (len < 2 || !D.need((size_t)len - 2)) return ingest_fail("jpeg: bad segment length");
    const size_t end = D.pos + (size_t)len - 2;
    D.limit = end;  // the handlers below read fixed-size fields: none may leave the segment
    switch (marker) {
      case 0xDB:  // DQT
        while (D.pos < end) {
          const int pq_tq = D.u8();
          const int precision = pq_tq >> 4, t = pq_tq & 15;
          if (t > 3) return ingest_fail("jpeg: bad quantisation table id");
          uint16_t table[64];
          for (int k = 0; k < 64; ++k) table[k] = (uint16_t)(precision ? D.u16() : D.u8());
          if (D.overrun) return ingest_fail("jpeg: quantisation table runs past its segment");
          std::memcpy(D.quant[t], table, sizeof table);
          D.have_quant[t] = true;
        }
        break;
      case 0xC0:
      case 0xC1: {  // baseline / extended sequential, Huffman
        if (D.u8() != 8) return ingest_fail("jpeg: only 8-bit samples are supported");
        D.height = D.u16();
        D.width = D.u16();
        D.n_comp = D.u8();
        if (D.overrun) {
          D.n_comp = 0;
          return ingest_fail("jpeg: frame header runs past its segment");
        }
        if (D.width <= 0 || D.height <= 0 || D.width > 32768 || D.height > 32768 ||
            (long long)D.width * D.height > (1ll << 26))
          return ingest_fail("jpeg: bad image size (at most 2^26 pixels)");
        if (D.n_comp != 1 && D.n_comp != 3) return ingest_fail("jpeg: 1 or 3 components are supported");
        for (int c = 0; c < D.n_comp; ++c) {
          D.comp[c].id = D.u8();
          const int hv = D.u8();
          D.comp[c].h = hv >> 4;
          D.comp[c].v = hv & 15;
          D.comp[c].tq = D.u8();
          if (D.comp[c].h < 1 || D.comp[c].h > 2 || D.comp[c].v < 1 || D.comp[c].v > 2 || D.comp[c].tq > 3)
            return ingest_fail("jpeg: unsupported sampling factors");
        }
        if (D.overrun) return ingest_fail("jpeg: frame header runs past its segment");
        break;
      }
      case 0xC2:
        return ingest_fail("jpeg: progressive files are not supported");
      case 0xC4:  // DHT
        while (D.pos < end) {
          const int tc_th = D.u8();
          const int cls = tc_th >> 4, t = tc_th & 15;
          if (cls > 1 || t > 3) return ingest_fail("jpeg: bad Huffman table id");
          uint8_t counts[16];
          int total = 0;
          for (int k = 0; k < 16; ++k) {
            counts[k] = (uint8_t)D.u8();
            total += counts[k];
          }
          if (D.overrun || total > 256 || D.pos + (size_t)total > end) return ingest_fail("jpeg: bad Huffman table");
          build_huffman(cls ? D.ac[t] : D.dc[t], counts, D.data + D.pos, total);
          D.pos += (size_t)total;
        }
        break;
      case 0xDD:  // DRI
        D.restart_interval = D.u16();
        if (D.overrun) return ingest_fail("jpeg: restart interval runs past its segment");
        break;
      case 0xDA: {  // SOS
        if (!D.n_comp) return ingest_fail("jpeg: scan before frame header");
        const int ns = D.u8();
        if (ns != D.n_comp) return ingest_fail("jpeg: non-interleaved scans are not supported");
        for (int k = 0; k < ns; ++k) {
          const int id = D.u8(), tables = D.u8();
          int c = 0;
          while (c < D.n_comp && D.comp[c].id != id) ++c;
          if (c == D.n_comp || c != k) return ingest_fail("jpeg: unexpected component order in scan");
          D.comp[c].td = tables >> 4;
          D.comp[c].ta = tables & 15;
          if (D.comp[c].td > 3 || D.comp[c].ta > 3) return ingest_fail("jpeg: bad table selector");
        }
        if (D.overrun) return ingest_fail("jpeg: scan header runs past its segment");
        D.pos = end;
        if (decode_scan(D) != 0) return -1;
        decoded = true;
        break;
      }
      default:
        break;  // APPn, COM and anything else: skipped
    }
    if (!decoded) D.pos = end;
  }

  int hmax = 1, vmax = 1;
  for (int c = 0; c < D.n_comp; ++c) {
    hmax = D.comp[c].h > hmax ? D.comp[c].h : hmax;
    vmax = D.comp[c].v > vmax ? D.comp[c].v : vmax;
  }
  uint8_t *out = (uint8_t *)std::malloc((size_t)D.width * D.height * 4);
  if (!out) return ingest_fail("rt_decode_jpeg: out of memory");
  for (int y = 0; y < D.height; ++y) {
    uint8_t *row = out + (size_t)(flip_y ? D.height - 1 - y : y) * D.width * 4;
    for (int x = 0; x < D.width; ++x) {
      const double Y = plane_sample(D.comp[0], x, y, hmax, vmax);
      if (D.n_comp == 1) {
        row[4 * x + 0] = row[4 * x + 1] = row[4 * x + 2] = clamp8(Y);
      } else {
        const double cb = plane_sample(D.comp[1], x, y, hmax, vmax) - 128.0;
        const double cr = plane_sample(D.comp[2], x, y, hmax, vmax) - 128.0;
        row[4 * x + 0] = clamp8(Y + 1.402 * cr);
        row[4 * x + 1] = clamp8(Y - 0.344136 * cb - 0.714136 * cr);
        row[4 * x + 2] = clamp8(Y + 1.772 * cb);
      }
      row[4 * x + 3] = 255;
    }
  }
  *out_rgba8 = out;
  *out_width = (uint32_t)D.width;
  *out_height = (uint32_t)D.height;
  return 0;
}

static int rt_load_jpeg_impl(const char *path, int flip_y, uint8_t **out_rgba8, uint32_t *out_width, uint32_t *out_height) {
  if (!path) return ingest_fail("rt_load_jpeg: null path");
  std::vector<uint8_t> bytes;
  if (!read_file(path, bytes)) return ingest_fail("rt_load_jpeg: cannot read file");
  return rt_decode_jpeg_impl(bytes.data(), bytes.size(), flip_y, out_rgba8, out_width, out_height);
}

int rt_load_obj(const char *path, float **out_vertices, uint32_t *out_n_vertices, uint32_t **out_indices,
                uint32_t *out_n_triangles) {
  try {
    return rt_load_obj_impl(path, out_vertices, out_n_vertices, out_indices, out_n_triangles);
  } catch (...) {
    return ingest_fail("rt_load_obj: out of memory");
  }
}

int rt_decode_jpeg(const void *bytes, size_t size, int flip_y, uint8_t **out_rgba8, uint32_t *out_width,
                   uint32_t *out_height) {
  try {
    return rt_decode_jpeg_impl(bytes, size, flip_y, out_rgba8, out_width, out_height);
  } catch (...) {
    return ingest_fail("rt_decode_jpeg: out of memory");
  }
}

int rt_load_jpeg(const char *path, int flip_y, uint8_t **out_rgba8, uint32_t *out_width, uint32_t *out_height) {
  try {
    return rt_load_jpeg_impl(path, flip_y, out_rgba8, out_width, out_height);
  } catch (...) {
    return ingest_fail("rt_load_jpeg: out of memory");
  }
}

}  // extern "C"
