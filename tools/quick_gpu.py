import time, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
# BASELINE.json configs[2]: 10k spheres, 3840x2160, 256 spp as 8 frames x 32 spp
w,h,spp,frames=3840,2160,32,8
ubo = rt.make_ubo(w,h,spp,25); sph = rt.grid_spheres_scene(10000)
for mode,name in ((rt.MODE_MEGAKERNEL,'mega'),(rt.MODE_WAVEFRONT,'wavefront')):
    with rt.Renderer(w,h,mode=mode,flags=0) as r:
        r.set_params(ubo); r.set_spheres(sph)
        st=r.stats(); print('build ms', st.last_build_ms, 'height', st.bvh_height, 'nodes', st.bvh_nodes, flush=True)
        r.render(0,1,1); r.sync(); r.clear()
        t0=time.perf_counter(); r.render(0,frames,1); r.sync(); t1=time.perf_counter()
        st1 = r.stats()
        print(name, 'bvh', 'ms', st1.last_render_ms, 'wall', (t1-t0)*1e3, 'rays', st1.rays, 'Grays/s', st1.rays/st1.last_render_ms/1e6, 'launches', st1.last_launches, flush=True)
        if mode == rt.MODE_MEGAKERNEL:
            img = r.read_rgba8()
            rt.write_ppm('gpurun_out/cfg3_4k.ppm', img[::4, ::4].copy())
