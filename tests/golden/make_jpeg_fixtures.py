#!/usr/bin/env python3
"""Writes the JPEG fixtures of tests/test_ingest.py with Pillow (libjpeg): a synthetic
picture (the library's brick pattern plus two colour ramps) encoded as baseline JPEG in
three samplings, and Pillow's own decode of each file as the expected pixels.

    python tests/golden/make_jpeg_fixtures.py
"""
import io
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ray_tracer_amd as rt  # noqa: E402

bricks = rt.brick_texture(93, 61)[..., :3].astype(int)   # odd sizes: partial MCUs at both edges
yy, xx = np.mgrid[0:61, 0:93]
bricks[..., 2] = np.clip(bricks[..., 2] + xx, 0, 255)
bricks[..., 1] = np.clip(bricks[..., 1] + 2 * yy, 0, 255)
picture = bricks.astype(np.uint8)

expected = {}
for name, mode, kw in (("420", "RGB", dict(subsampling=2, quality=90)),
                       ("444_restart", "RGB", dict(subsampling=0, quality=80, restart_marker_blocks=5)),
                       ("grey", "L", dict(quality=85))):
    b = io.BytesIO()
    Image.fromarray(picture).convert(mode).save(b, "JPEG", **kw)
    data = b.getvalue()
    with open(os.path.join(HERE, f"picture_{name}.jpg"), "wb") as f:
        f.write(data)
    expected[name] = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    print(name, len(data), "bytes", "DRI" if b"\xff\xdd" in data else "")
np.savez_compressed(os.path.join(HERE, "picture_decoded.npz"), **expected)
