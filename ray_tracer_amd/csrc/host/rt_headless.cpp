// rt_headless -- the reference's main() without the window: same camera, same uniform
// block, same frame loop (src/main.cpp:623-636 set-up, :697-703 per-frame UBO fill,
// :1028-1031 frame counter), with every ImGui field of :712-778 as a command-line flag,
// the compute dispatch replaced by the C ABI of librt_hip.so, and the picture written
// to a file instead of a swapchain.
//
// --gpus N renders on N devices of one node: one host thread and one context per device,
// device r traces progressive frames r, r+N, ... (the shader seeds every frame on its own,
// ray_tracing.comp:80-82, and averages frames with equal weight, :106-107), the float
// accumulation images are summed onto device 0 with ONE ncclReduce (RCCL over xGMI), and
// device 0 resolves the sum to the 8-bit picture.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../../include/rt_hip.h"
#include "../../../include/rt_host.h"

namespace {

void usage() {
  std::puts(
      "rt_headless [options]\n"
      "  --width W --height H        image size (default 800x608, the reference's window)\n"
      "  --viewport reference|image  camera viewport: 800x600 as the reference leaves it, or the image size (default image)\n"
      "  --samples N --bounces N     render settings (default 50, 25)\n"
      "  --frames N                  progressive frames (default 1)\n"
      "  --ground r,g,b --horizon r,g,b --zenith r,g,b --sun x,y,z\n"
      "  --sun-focus F --sun-intensity F --defocus F --diverge F\n"
      "  --pitch RAD --yaw RAD       camera orientation;  --rotate dx,dy  applies cameraRotate\n"
      "  --scene default|grid:N|glass:N   the reference's 3 spheres, or the synthetic N-sphere scene (glass: its\n"
      "                              third material class is dielectric, index of refraction 1.5)\n"
      "  --mesh FILE.obj             triangle mesh traced with the spheres (matte grey material)\n"
      "  --texture bricks|FILE.jpg   albedo texture;  --texture-prims FIRST,COUNT (default: the last sphere)\n"
      "  --texture-scale S           repeats per world unit (default 0.25);  --texture-map planar|octahedral\n"
      "  --mode megakernel|wavefront|nested   --strict   --linear   --brute   --bvh\n"
      "  --device D                  HIP device ordinal (with --gpus: the first of N consecutive ones)\n"
      "  --gpus N                    split the frames over N devices and sum the images with one RCCL reduce\n"
      "  --reduce                    take the multi-device path (threads, RCCL communicator, reduce, resolve) even for N = 1\n"
      "  --fail-rank R               fault injection: device R of the multi-device path reports a failed render\n"
      "  --out FILE.ppm|FILE.png     shown as the reference's window shows it (bottom row first in memory)\n");
}

bool parse3(const char *s, float out[3]) { return std::sscanf(s, "%f,%f,%f", &out[0], &out[1], &out[2]) == 3; }

int die(const char *what) {
  std::fprintf(stderr, "rt_headless: %s: %s\n", what, rt_last_error());
  return 1;
}

}  // namespace

int main(int argc, char **argv) {
  uint32_t width = 800, height = 608, frames = 1, mode = RT_MODE_MEGAKERNEL, flags = 0;
  int device = 0, gpus = 1, fail_rank = -1;
  bool viewport_image = true, always_reduce = false;
  std::string scene = "default", out_path, mesh_path, texture_name;
  long tex_first = -1, tex_count = 1;
  float tex_scale = 0.25f;
  uint32_t tex_map = RT_TEX_MAP_PLANAR_XZ;
  rt_camera camera;
  // main.cpp:624 passes window_width / window_height, an unsigned division, as the aspect
  rt_camera_create(90.0f, (float)(800u / 608u), 0.01f, 10000.0f, &camera);
  rt_ubo ubo;
  rt_default_ubo(&camera, &ubo);
  float rot[2] = {0.f, 0.f};
  bool have_rot = false;

  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> const char * {
      if (i + 1 >= argc) { usage(); std::exit(2); }
      return argv[++i];
    };
    if (a == "--width") width = (uint32_t)std::atoi(next());
    else if (a == "--height") height = (uint32_t)std::atoi(next());
    else if (a == "--viewport") viewport_image = std::string(next()) != "reference";
    else if (a == "--samples") ubo.render_settings[0] = (float)std::atof(next());
    else if (a == "--bounces") ubo.render_settings[1] = (float)std::atof(next());
    else if (a == "--frames") frames = (uint32_t)std::atoi(next());
    else if (a == "--ground") { if (!parse3(next(), ubo.ground_colour)) { usage(); return 2; } ubo.ground_colour[3] = 1.f; }
    else if (a == "--horizon") { if (!parse3(next(), ubo.sky_colour_horizon)) { usage(); return 2; } ubo.sky_colour_horizon[3] = 1.f; }
    else if (a == "--zenith") { if (!parse3(next(), ubo.sky_colour_zenith)) { usage(); return 2; } ubo.sky_colour_zenith[3] = 1.f; }
    else if (a == "--sun") { if (!parse3(next(), ubo.sun_position)) { usage(); return 2; } ubo.sun_position[3] = 1.f; }
    else if (a == "--sun-focus") ubo.sun_focus = (float)std::atof(next());
    else if (a == "--sun-intensity") ubo.sun_intensity = (float)std::atof(next());
    else if (a == "--defocus") ubo.defocus_strength = (float)std::atof(next());
    else if (a == "--diverge") ubo.diverge_strength = (float)std::atof(next());
    else if (a == "--pitch") camera.pitch = (float)std::atof(next());
    else if (a == "--yaw") camera.yaw = (float)std::atof(next());
    else if (a == "--rotate") { have_rot = std::sscanf(next(), "%f,%f", &rot[0], &rot[1]) == 2; }
    else if (a == "--scene") scene = next();
    else if (a == "--mesh") mesh_path = next();
    else if (a == "--texture") texture_name = next();
    else if (a == "--texture-prims") { if (std::sscanf(next(), "%ld,%ld", &tex_first, &tex_count) != 2) { usage(); return 2; } }
    else if (a == "--texture-scale") tex_scale = (float)std::atof(next());
    else if (a == "--texture-map") tex_map = std::string(next()) == "octahedral" ? RT_TEX_MAP_OCTAHEDRAL : RT_TEX_MAP_PLANAR_XZ;
    else if (a == "--mode") {
      const std::string m = next();
      mode = m == "wavefront" ? RT_MODE_WAVEFRONT : m == "nested" ? RT_MODE_MEGAKERNEL_NESTED : RT_MODE_MEGAKERNEL;
    } else if (a == "--strict") flags |= RT_FLAG_STRICT_MATH;
    else if (a == "--linear") flags |= RT_FLAG_ACCUM_LINEAR;
    else if (a == "--brute") flags |= RT_FLAG_FORCE_BRUTE;
    else if (a == "--bvh") flags |= RT_FLAG_FORCE_BVH;
    else if (a == "--device") device = std::atoi(next());
    else if (a == "--gpus") gpus = std::atoi(next());
    else if (a == "--reduce") always_reduce = true;
    else if (a == "--fail-rank") fail_rank = std::atoi(next());
    else if (a == "--out") out_path = next();
    else { usage(); return a == "--help" || a == "-h" ? 0 : 2; }
  }
  if (viewport_image) {
    camera.viewport_width = (float)width;
    camera.viewport_height = (float)height;
  }
  if (have_rot) rt_camera_rotate(&camera, rot[0], rot[1]);  // main.cpp:692-695

  if (gpus < 1 || gpus > 64) { usage(); return 2; }

  std::vector<rt_sphere> spheres;
  std::vector<float> ior;
  if (scene == "default") {
    spheres.resize(3);
    rt_default_scene(spheres.data());
  } else if (scene.rfind("grid:", 0) == 0 || scene.rfind("glass:", 0) == 0) {
    const bool glass = scene[1] == 'l';
    const uint32_t n = (uint32_t)std::atoi(scene.c_str() + (glass ? 6 : 5));
    if (n == 0) { usage(); return 2; }
    spheres.resize(n);
    if (glass) {
      ior.resize(n);
      rt_grid_scene_glass(n, 1337u, spheres.data(), ior.data());
    } else {
      rt_grid_scene(n, 1337u, spheres.data());
    }
  } else { usage(); return 2; }

  // files are read once; every device gets the same bytes
  float *vertices = nullptr;
  uint32_t *indices = nullptr, n_vertices = 0, n_triangles = 0;
  if (!mesh_path.empty() && rt_load_obj(mesh_path.c_str(), &vertices, &n_vertices, &indices, &n_triangles) != 0) {
    std::fprintf(stderr, "rt_headless: %s: %s\n", mesh_path.c_str(), rt_ingest_error());
    return 1;
  }
  uint8_t *texels = nullptr;
  uint32_t tw = 0, th = 0;
  std::vector<uint8_t> bricks;
  if (texture_name == "bricks") {
    tw = th = 1024;
    bricks.resize((size_t)tw * th * 4);
    rt_brick_texture(tw, th, bricks.data());
  } else if (!texture_name.empty() && rt_load_jpeg(texture_name.c_str(), 1, &texels, &tw, &th) != 0) {  // flipped, as loadTexture asks of stb
    std::fprintf(stderr, "rt_headless: %s: %s\n", texture_name.c_str(), rt_ingest_error());
    return 1;
  }

  // one context per device, the scene uploaded to each
  auto make_context = [&](int dev, void *stream, rt_ctx **out) -> const char * {
    rt_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.device = dev;
    cfg.width = width;
    cfg.height = height;
    cfg.mode = mode;
    cfg.flags = flags;
    cfg.stream = stream;
    if (rt_create(&cfg, out) != RT_OK) return "rt_create";
    rt_ctx *ctx = *out;
    if (rt_set_spheres(ctx, spheres.data(), (uint32_t)spheres.size()) != RT_OK) return "rt_set_spheres";
    if (!ior.empty() && rt_set_dielectric(ctx, ior.data(), (uint32_t)ior.size()) != RT_OK) return "rt_set_dielectric";
    if (n_triangles) {
      rt_material grey;
      std::memset(&grey, 0, sizeof grey);
      grey.colour[0] = grey.colour[1] = grey.colour[2] = 0.7f;
      grey.specular_colour[0] = grey.specular_colour[1] = grey.specular_colour[2] = 1.0f;
      if (rt_set_mesh(ctx, vertices, n_vertices, indices, n_triangles, &grey) != RT_OK) return "rt_set_mesh";
    }
    if (!texture_name.empty()) {
      rt_texture_desc td;
      std::memset(&td, 0, sizeof td);
      td.struct_size = sizeof td;
      td.width = tw;
      td.height = th;
      td.flags = RT_TEX_SRGB;
      td.mapping = tex_map;
      td.scale_u = td.scale_v = tex_scale;
      td.first_prim = tex_first >= 0 ? (uint32_t)tex_first : (uint32_t)(spheres.empty() ? 0 : spheres.size() - 1);
      td.n_prims = (uint32_t)tex_count;
      if (rt_set_texture(ctx, &td, texels ? texels : bricks.data()) != RT_OK) return "rt_set_texture";
    }
    return nullptr;
  };
  // the reference's frame loop for the frames first, first + stride, ...
  auto render_frames = [&](rt_ctx *ctx, uint32_t first, uint32_t stride) -> const char * {
    rt_camera cam = camera;
    rt_ubo u = ubo;
    for (uint32_t frame = first; frame < frames; frame += stride) {
      rt_ubo_set_camera(&cam, &u);              // main.cpp:697-701
      u.frame[0] = (float)frame;                // main.cpp:1028
      if (rt_set_params(ctx, &u) != RT_OK) return "rt_set_params";    // main.cpp:703
      if (rt_render(ctx, frame, 1, 1) != RT_OK) return "rt_render";   // main.cpp:819-868
    }
    return nullptr;
  };

  std::vector<rt_ctx *> ctx((size_t)gpus, nullptr);
  std::vector<rt_stats> st((size_t)gpus);
  std::vector<std::string> failure((size_t)gpus);
  double wall_ms = 0.0;
  size_t reduce_bytes = 0;   // bytes each non-root device contributes to the sum
  if (gpus == 1 && !always_reduce) {
    if (const char *what = make_context(device, nullptr, &ctx[0])) return die(what);
    const auto t0 = std::chrono::steady_clock::now();
    if (const char *what = render_frames(ctx[0], 0, 1)) return die(what);
    if (rt_sync(ctx[0]) != RT_OK) return die("rt_sync");                        // main.cpp:872-874
    wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rt_get_stats(ctx[0], &st[0]) != RT_OK) return die("rt_get_stats");
  } else {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device + gpus > n_dev) {
      std::fprintf(stderr, "rt_headless: --gpus %d from device %d, but %d devices are visible\n", gpus, device, n_dev);
      return 1;
    }
    std::vector<int> devs((size_t)gpus);
    for (int r = 0; r < gpus; ++r) devs[(size_t)r] = device + r;
    std::vector<ncclComm_t> comm((size_t)gpus);
    if (ncclCommInitAll(comm.data(), gpus, devs.data()) != ncclSuccess) {
      std::fprintf(stderr, "rt_headless: ncclCommInitAll failed\n");
      return 1;
    }
    std::vector<hipStream_t> stream((size_t)gpus, nullptr);
    // every way out of this branch that is not success goes through here: communicators that may have a
    // collective outstanding are aborted, not destroyed (destroy would wait for peers that never arrive)
    auto give_up = [&]() {
      for (int r = 0; r < gpus; ++r) ncclCommAbort(comm[(size_t)r]);
      for (int r = 0; r < gpus; ++r)
        if (ctx[(size_t)r]) rt_destroy(ctx[(size_t)r]);
      return 1;
    };
    for (int r = 0; r < gpus; ++r) {
      if (hipSetDevice(devs[(size_t)r]) != hipSuccess || hipStreamCreate(&stream[(size_t)r]) != hipSuccess) {
        std::fprintf(stderr, "rt_headless: cannot create a stream on device %d\n", devs[(size_t)r]);
        return give_up();
      }
      if (const char *what = make_context(devs[(size_t)r], stream[(size_t)r], &ctx[(size_t)r])) {
        die(what);
        return give_up();
      }
    }
    // Phase 1, one host thread per device: trace this device's share of the frames.  No collective is
    // enqueued here, so a rank that fails cannot leave its peers waiting inside one.  A context is only
    // ever touched by its own thread (rt_last_error is per thread: a rank keeps its own message).
    std::vector<void *> accum((size_t)gpus, nullptr);
    size_t bytes = 0;
    std::vector<size_t> rank_bytes((size_t)gpus, 0);
    auto rank_main = [&](int r) {
      const size_t k = (size_t)r;
      auto fail_with = [&](const char *what) { failure[k] = std::string(what) + ": " + rt_last_error(); };
      if (hipSetDevice(devs[k]) != hipSuccess) { failure[k] = "hipSetDevice failed"; return; }
      if (const char *what = render_frames(ctx[k], (uint32_t)r, (uint32_t)gpus)) { fail_with(what); return; }
      if (r == fail_rank) { failure[k] = "rt_render: injected failure (--fail-rank)"; return; }
      if (rt_accum_device_ptr(ctx[k], &accum[k], &rank_bytes[k]) != RT_OK) fail_with("rt_accum_device_ptr");
    };
    const auto t0 = std::chrono::steady_clock::now();
    {
      std::vector<std::thread> threads;
      for (int r = 0; r < gpus; ++r) threads.emplace_back(rank_main, r);
      for (auto &t : threads) t.join();
    }
    bool failed = false;
    for (int r = 0; r < gpus; ++r)
      if (!failure[(size_t)r].empty()) {
        std::fprintf(stderr, "rt_headless: device %d: %s\n", devs[(size_t)r], failure[(size_t)r].c_str());
        failed = true;
      }
    if (failed) return give_up();
    bytes = rank_bytes[0];
    // Phase 2, only when every rank has its image: ONE sum onto device 0, each rank's part enqueued on the
    // stream its render launches went to (so it runs behind them without a host wait), in place on the root.
    ncclResult_t nrc = ncclGroupStart();
    for (int r = 0; r < gpus && nrc == ncclSuccess; ++r)
      nrc = ncclReduce(accum[(size_t)r], r == 0 ? accum[0] : nullptr, bytes / sizeof(float), ncclFloat, ncclSum, 0,
                       comm[(size_t)r], stream[(size_t)r]);
    const ncclResult_t nend = ncclGroupEnd();
    if (nrc != ncclSuccess || nend != ncclSuccess) {
      std::fprintf(stderr, "rt_headless: ncclReduce failed: %s\n", ncclGetErrorString(nrc != ncclSuccess ? nrc : nend));
      return give_up();
    }
    if (rt_resolve(ctx[0]) != RT_OK) { die("rt_resolve"); return give_up(); }
    for (int r = 0; r < gpus; ++r) {
      if (rt_sync(ctx[(size_t)r]) != RT_OK) { die("rt_sync"); return give_up(); }
      if (rt_get_stats(ctx[(size_t)r], &st[(size_t)r]) != RT_OK) { die("rt_get_stats"); return give_up(); }
    }
    wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    reduce_bytes = gpus > 1 ? bytes : 0;
    for (int r = 0; r < gpus; ++r) ncclCommDestroy(comm[(size_t)r]);
  }
  rt_free(vertices);
  rt_free(indices);
  rt_free(texels);

  std::vector<uint8_t> pixels((size_t)width * height * 4);
  if (rt_read_rgba8(ctx[0], pixels.data(), pixels.size()) != RT_OK) return die("rt_read_rgba8");
  if (!out_path.empty()) {
    const bool png = out_path.size() > 4 && out_path.substr(out_path.size() - 4) == ".png";
    const int rc = png ? rt_write_png(out_path.c_str(), pixels.data(), width, height, 1)
                       : rt_write_ppm(out_path.c_str(), pixels.data(), width, height, 1);
    if (rc != 0) { std::perror(out_path.c_str()); return 1; }
  }
  uint64_t sum = 0, rays = 0;
  for (uint8_t v : pixels) sum += v;
  float device_ms = 0.f;   // the slowest device's render time
  for (int r = 0; r < gpus; ++r) {
    rays += st[(size_t)r].rays;
    device_ms = st[(size_t)r].total_render_ms > device_ms ? st[(size_t)r].total_render_ms : device_ms;
  }
  std::printf("{\"width\": %u, \"height\": %u, \"frames\": %u, \"gpus\": %d, \"spheres\": %zu, \"triangles\": %u, "
              "\"rays\": %llu, \"device_ms\": %.3f, \"wall_ms\": %.3f, \"mrays_per_s\": %.1f, \"bvh_nodes\": %u, "
              "\"bvh_build_ms\": %.3f, \"reduce_bytes_per_device\": %zu, \"image_byte_sum\": %llu}\n",
              width, height, frames, gpus, spheres.size(), n_triangles, (unsigned long long)rays, device_ms, wall_ms,
              device_ms > 0 ? rays / device_ms / 1e3 : 0.0, st[0].bvh_nodes, st[0].last_build_ms, reduce_bytes,
              (unsigned long long)sum);
  for (int r = 0; r < gpus; ++r) rt_destroy(ctx[(size_t)r]);
  return 0;
}
