import sys
sys.path.insert(0, '.'); sys.path.insert(0, '..')
import ray_tracer_amd as rt
w,h,spp=1920,1080,64
ubo = rt.make_ubo(w,h,spp,25); sph = rt.default_scene()
with rt.Renderer(w,h,mode=rt.MODE_WAVEFRONT,flags=0) as r:
    r.set_params(ubo); r.set_spheres(sph)
    r.render(0,1,1); r.sync()
    print(r.stats().last_render_ms, r.stats().last_launches)
