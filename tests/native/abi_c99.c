/* The public headers must be plain C: this file is compiled (not run) by
 * tests/test_abi.py with gcc -std=c99 -pedantic -Wall -Werror.  It also spells out the
 * calls a C host makes: create, scene, parameters, render, read back, destroy. */
#include <stddef.h>
#include <string.h>

#include "../../include/rt_contract.h"
#include "../../include/rt_hip.h"
#include "../../include/rt_host.h"

int render_once(unsigned width, unsigned height, unsigned char *rgba8_out) {
  rt_config cfg;
  rt_ctx *ctx = NULL;
  rt_camera camera;
  rt_ubo ubo;
  rt_sphere spheres[3];
  rt_stats stats;
  int rc;

  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.width = width;
  cfg.height = height;
  cfg.mode = RT_MODE_MEGAKERNEL;
  rc = rt_create(&cfg, &ctx);
  if (rc != RT_OK) return rc;

  rt_camera_create(90.0f, 1.0f, 0.01f, 10000.0f, &camera);
  camera.viewport_width = (float)width;
  camera.viewport_height = (float)height;
  rt_default_ubo(&camera, &ubo);
  rt_default_scene(spheres);

  rc = rt_set_spheres(ctx, spheres, 3);
  if (rc == RT_OK) rc = rt_set_params(ctx, &ubo);
  if (rc == RT_OK) rc = rt_render(ctx, 0, 1, 1);
  if (rc == RT_OK) rc = rt_read_rgba8(ctx, rgba8_out, (size_t)width * height * 4);
  if (rc == RT_OK) rc = rt_get_stats(ctx, &stats);
  rt_destroy(ctx);
  return rc;
}

/* layout facts a C caller relies on */
typedef char ubo_is_272_bytes[sizeof(rt_ubo) == 272 ? 1 : -1];
typedef char sphere_is_64_bytes[sizeof(rt_sphere) == 64 ? 1 : -1];
typedef char material_is_48_bytes[sizeof(rt_material) == 48 ? 1 : -1];
