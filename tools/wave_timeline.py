#!/usr/bin/env python3
"""Per-wave begin/end times of one megakernel launch (needs a library built with -DRT_MEASURE_TIMELINE, named by RT_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
scene = sys.argv[1] if len(sys.argv) > 1 else "default"   # default | terrain | grid | glass4k
w, h = (3840, 2160) if scene == "glass4k" else (1920, 1080)
frames = 8 if scene == "glass4k" else 1
ubo = rt.make_ubo(w, h, 64, 25) if scene == "default" else rt.make_ubo(w, h, 32, 25) if scene == "glass4k" else \
    rt.make_ubo(w, h, 8, 8 if scene == "terrain" else 25)
with rt.Renderer(w, h) as r:
    r.set_params(ubo)
    if scene == "glass4k":
        sph, ior = rt.grid_spheres_scene_glass(10000)
        r.set_spheres(sph); r.set_dielectric(ior)
    elif scene == "terrain":
        r.set_spheres(rt.default_scene()[:2])
        r.set_mesh(*rt.terrain_mesh(709), rt.matte(0.55, 0.5, 0.4, smoothness=0.7, specular_probability=0.15))
    elif scene == "grid":
        r.set_spheres(rt.grid_spheres_scene(10000))
    else:
        r.set_spheres(rt.default_scene())
    r.render(0, frames, 1); r.sync()
    r.render(frames, frames, 1); r.sync()
    if len(sys.argv) > 2 and scene == "terrain":   # a new pose of the surface before the recorded launch, as in bench.py
        r.set_mesh(*rt.terrain_mesh(709, phase=float(sys.argv[2])), rt.matte(0.55, 0.5, 0.4, smoothness=0.7, specular_probability=0.15))
    r.render(2 * frames, frames, 1); r.sync()
    ms = r.stats().last_render_ms
    ids = r.read_primary_ids().reshape(-1)
n_waves = ((w + 15) // 16) * ((h + 15) // 16) * 4
rec = ids[:6 * n_waves].reshape(n_waves, 6).astype(np.int64) & 0xffffffff
t0 = rec[:, 0] | (rec[:, 1] << 32)
t1 = rec[:, 2] | (rec[:, 3] << 32)
base = t0.min()
b = (t0 - base) / 100.0   # microseconds
e = (t1 - base) / 100.0
dur = e - b
print(f"kernel {ms:.3f} ms; waves {n_waves}; span {e.max():.0f} us")
print("duration us: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print("sum of wave durations / (span * 6144 slots) = %.2f" % (dur.sum() / (e.max() * 6144)))
for q in (0.5, 0.75, 0.9, 0.95, 0.99, 1.0):
    print(f"  {q:.2f} of waves ended by {np.quantile(e, q):.0f} us")
act = np.zeros(int(e.max()) + 3)
np.add.at(act, b.astype(int), 1)
np.add.at(act, e.astype(int) + 1, -1)
act = np.cumsum(act)
step = max(1, len(act) // 24)
print("resident waves over time:", " ".join(str(int(act[i:i + step].mean())) for i in range(0, len(act), step)))
late = np.argsort(-e)[:8]
for k in late:
    blk = k // 4
    print(f"  wave {k} tile ({blk % ((w + 15) // 16)},{blk // ((w + 15) // 16)}) begin {b[k]:.0f} end {e[k]:.0f} dur {dur[k]:.0f} rays {rec[k,4]} slowest lane {rec[k,5]}")
np.save("gpurun_out/wave_timeline.npy", np.stack([b, e, rec[:, 4], rec[:, 5]], 1))
