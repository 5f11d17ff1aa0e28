// Robustness run for the file readers of include/rt_host.h, meant to be built with
// -fsanitize=address,undefined (tests/test_ingest.py does): the JPEG fixtures must decode,
// and a few thousand truncated or corrupted variants, as well as random OBJ text, must be
// either decoded or rejected -- never read or write out of bounds.
//
//   ingest_fuzz <dir with picture_*.jpg> <scratch file> <iterations>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_host.h"

static uint32_t g_state = 12345u;
static uint32_t rnd() {  // the shader's PCG hash, as everywhere else in this repository
  g_state = g_state * 747796405u + 2891336453u;
  uint32_t r = ((g_state >> ((g_state >> 28) + 4u)) ^ g_state) * 277803737u;
  return (r >> 22) ^ r;
}

static std::vector<uint8_t> slurp(const std::string &path) {
  std::vector<uint8_t> out;
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return out;
  int c;
  while ((c = std::fgetc(f)) != EOF) out.push_back((uint8_t)c);
  std::fclose(f);
  return out;
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const std::string dir = argv[1], scratch = argv[2];
  const int iterations = std::atoi(argv[3]);
  const char *names[3] = {"picture_420.jpg", "picture_444_restart.jpg", "picture_grey.jpg"};
  long decoded = 0, rejected = 0;
  for (const char *name : names) {
    const std::vector<uint8_t> good = slurp(dir + "/" + name);
    if (good.empty()) {
      std::fprintf(stderr, "missing fixture %s\n", name);
      return 1;
    }
    uint8_t *px = nullptr;
    uint32_t w = 0, h = 0;
    if (rt_decode_jpeg(good.data(), good.size(), 0, &px, &w, &h) != 0 || w != 93 || h != 61) {
      std::fprintf(stderr, "%s: %s\n", name, rt_ingest_error());
      return 1;
    }
    rt_free(px);
    for (int it = 0; it < iterations; ++it) {
      std::vector<uint8_t> bad = good;
      const uint32_t kind = rnd() % 4u;
      if (kind == 0) {
        bad.resize(rnd() % bad.size());                       // truncated
      } else {
        const uint32_t flips = 1u + rnd() % (kind == 1 ? 2u : 24u);
        for (uint32_t k = 0; k < flips; ++k) {
          // kind 3 aims at the headers, where lengths, table ids and sizes live
          const size_t at = kind == 3 ? rnd() % (bad.size() < 700 ? bad.size() : 700) : rnd() % bad.size();
          bad[at] = (uint8_t)rnd();
        }
      }
      px = nullptr;
      if (rt_decode_jpeg(bad.data(), bad.size(), (int)(rnd() & 1u), &px, &w, &h) == 0) {
        ++decoded;
        volatile uint8_t sink = px[(size_t)w * h * 4 - 1];   // the whole buffer must be there
        (void)sink;
        rt_free(px);
      } else {
        ++rejected;
      }
    }
  }

  // Short final segments: the declared length fits the file, the fields the segment must hold
  // do not (a quantisation table, a frame header, a scan header, a Huffman table, a restart
  // interval).  Each buffer is allocated at its exact size so that a read past its end is one
  // past the allocation.  All must be rejected.
  {
    const std::vector<std::vector<uint8_t>> cut = {
        {0xFF, 0xD8, 0xFF, 0xDB, 0x00, 0x03, 0x00},
        {0xFF, 0xD8, 0xFF, 0xC0, 0x00, 0x02},
        {0xFF, 0xD8, 0xFF, 0xDA, 0x00, 0x02},
        {0xFF, 0xD8, 0xFF, 0xC4, 0x00, 0x03, 0x00},
        {0xFF, 0xD8, 0xFF, 0xDD, 0x00, 0x02},
        {0xFF, 0xD8, 0xFF, 0xC0, 0x00, 0x08, 0x08, 0x00, 0x10, 0x00, 0x10, 0x03},
        {0xFF, 0xD8, 0xFF, 0xDB, 0x00, 0x04, 0x10, 0x01},
    };
    for (const std::vector<uint8_t> &c : cut) {
      uint8_t *exact = (uint8_t *)std::malloc(c.size());
      std::memcpy(exact, c.data(), c.size());
      uint8_t *px = nullptr;
      uint32_t w = 0, h = 0;
      const int rc = rt_decode_jpeg(exact, c.size(), 0, &px, &w, &h);
      std::free(exact);
      if (rc == 0) {
        std::fprintf(stderr, "a %zu-byte file with a short final segment was accepted\n", c.size());
        return 1;
      }
      ++rejected;
    }
  }

  const char *words[] = {"v", "f", "vn", "vt", "#", "o", "1", "2", "3", "-1", "-2", "0", "1/1", "2//3",
                         "3/2/1", "1e9", "-0.5", "nan", "x", "/", "99999999999999999999", "\r", "\t", " ",
                         "\n", "\n", "\n", "4", "1.5", "f 1 2 3", "v 0 0 0"};
  const int n_words = (int)(sizeof words / sizeof words[0]);
  long meshes = 0;
  for (int it = 0; it < iterations; ++it) {
    std::string text;
    const uint32_t n = rnd() % 60u;
    for (uint32_t k = 0; k < n; ++k) {
      text += words[rnd() % n_words];
      if (rnd() % 3u) text += ' ';
    }
    if (rnd() % 4u == 0) text.push_back('\0');
    FILE *f = std::fopen(scratch.c_str(), "wb");
    if (!f) return 1;
    std::fwrite(text.data(), 1, text.size(), f);
    std::fclose(f);
    float *v = nullptr;
    uint32_t *i = nullptr, nv = 0, nt = 0;
    if (rt_load_obj(scratch.c_str(), &v, &nv, &i, &nt) == 0) {
      ++meshes;
      for (uint32_t k = 0; k < 3 * nt; ++k)
        if (i[k] >= nv) {
          std::fprintf(stderr, "index %u of %u vertices accepted\n", i[k], nv);
          return 1;
        }
      rt_free(v);
      rt_free(i);
    }
  }
  std::printf("jpeg variants decoded %ld rejected %ld; obj texts accepted %ld of %d\n", decoded, rejected,
              meshes, iterations);
  return 0;
}
