#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the CPU oracle.

The reference ships no golden vectors and cannot be run in this image (its hot path is a
GLSL compute shader; no GLSL compiler or Vulkan loader is installed), so these vectors
come from oracle/rt_oracle.c -- the CPU restatement of the shader -- and from closed
forms that follow from the shader text.  They pin the oracle (and through it the HIP
kernels) against regressions; they do not pin the oracle against the reference.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as oracle  # noqa: E402


def rle(a):
    a = np.asarray(a).ravel()
    change = np.flatnonzero(np.diff(a)) + 1
    starts = np.concatenate(([0], change))
    lengths = np.diff(np.concatenate((starts, [a.size])))
    return np.stack([a[starts], lengths], axis=1).astype(np.int32)


def main():
    # 1. PCG streams: state, raw output, float value for 8 draws of 6 seeds
    seeds = [0, 1, 639807, 719393, 0xFFFFFFFF, 123456789]
    pcg = {str(s): [[int(a), int(b), float(c)] for a, b, c in oracle.pcg_stream(s, 8)] for s in seeds}
    json.dump(pcg, open(os.path.join(HERE, "pcg_streams.json"), "w"), indent=1)

    # 2. primary-hit id map of the reference's own window: 800x608 image, 800x600 viewport,
    #    default scene, jitter off (run-length encoded, row-major)
    ubo = oracle.default_ubo()
    ubo[40], ubo[41], ubo[66], ubo[67] = 1, 1, 0, 0
    res = oracle.render(ubo, oracle.default_spheres(), 800, 608)
    np.save(os.path.join(HERE, "primary_ids_800x608_rle.npy"), rle(res["ids"]))

    # 3. a small float image: 64x48, 4 spp, 25 bounces, 2 frames, viewport = image, default
    #    jitter; gamma-space accumulation, the 8-bit progressive image and the ray count
    cam = oracle.camera_create(90.0, 1.0, 0.01, 10000.0)
    cam.viewport_width, cam.viewport_height = 64.0, 48.0
    ubo = oracle.default_ubo(cam)
    ubo[40], ubo[41] = 4, 25
    res = oracle.render(ubo, oracle.default_spheres(), 64, 48, 0, 2, 1)
    np.savez_compressed(os.path.join(HERE, "image_64x48_4spp_2frames.npz"), ubo=ubo,
                        accum=res["accum"], rgba8=res["rgba8"], ids=res["ids"],
                        rays=np.int64(res["rays"]))

    # 4. camera matrices at a rotated pose
    cam = oracle.camera_create(90.0, 1.0, 0.01, 10000.0)
    cam.pitch, cam.yaw = 0.3, -1.2
    np.savez(os.path.join(HERE, "camera_pose.npz"), view=oracle.camera_view(cam),
             projection=oracle.camera_projection(cam), pitch=np.float32(0.3), yaw=np.float32(-1.2))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
