"""rt_render_static, the entry point that allocates nothing, against rt_render.

Both are called through the C ABI on the same inputs; the bar is the one of the strict and
fast parity tests for two launches of the same kernels on the same data: every output bit for
bit (accumulation image, 8-bit image, primary ids, ray count).  Two calls in a row share the
library's static scratch array; the second one is smaller (image and scene), so stale
contents of the first would show.
"""
import numpy as np
import pytest

import ray_tracer_amd as rt

pytestmark = pytest.mark.gpu


def _reference(ubo, sph, w, h, flags, first, n, stride):
    with rt.Renderer(w, h, flags=flags | rt.FLAG_FORCE_BRUTE) as r:
        r.set_params(ubo)
        r.set_spheres(sph)
        r.render(first, n, stride)
        r.sync()
        return dict(accum=r.read_accum(), rgba8=r.read_rgba8(), ids=r.read_primary_ids(), rays=int(r.stats().rays))


def _static(ubo, sph, w, h, flags, first, n, stride, stream=None):
    import torch
    dev = torch.device("cuda", 0)
    d_sph = torch.from_numpy(np.ascontiguousarray(sph, dtype=np.float32)).to(dev)
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
    rgba8 = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    ids = torch.full((h, w), -1, dtype=torch.int32, device=dev)
    rays = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    free_before = torch.cuda.mem_get_info(dev)[0]
    rt.render_static(w, h, ubo, d_sph.data_ptr(), sph.shape[0], accum.data_ptr(), rgba8.data_ptr(),
                     ids.data_ptr(), rays.data_ptr(), first, n, stride, flags=flags,
                     stream=stream.cuda_stream if stream is not None else 0)
    torch.cuda.synchronize(dev)
    free_after = torch.cuda.mem_get_info(dev)[0]
    return dict(accum=accum.cpu().numpy(), rgba8=rgba8.cpu().numpy(), ids=ids.cpu().numpy(),
                rays=int(rays.item()), freed=free_after - free_before)


def _same(got, want):
    assert got["rays"] == want["rays"]
    assert np.array_equal(got["ids"], want["ids"])
    assert np.array_equal(got["accum"].view(np.uint32), want["accum"].view(np.uint32))
    assert np.array_equal(got["rgba8"], want["rgba8"])


@pytest.mark.parametrize("flags", [0, rt.FLAG_STRICT_MATH])
def test_static_entry_point_equals_rt_render_twice_in_a_row(flags):
    # first call: 40 spheres, 232x136 (ragged tiles), 2 frames x 20 spp (the cost probe runs)
    w, h = 232, 136
    ubo = rt.make_ubo(w, h, 20, 12, sun_intensity=2.0, defocus=1.0)
    sph = rt.grid_spheres_scene(40)
    sph[0, :4] = (0, -101, -5, 100)
    sph[1:, 2] += 55.0                              # bring the grid in front of the camera
    want = _reference(ubo, sph, w, h, flags, 1, 2, 3)
    # set-up happens once per device, in the first call: the runtime loads the code object that
    # holds the static array.  Free device memory is compared around the calls after that one.
    _static(rt.make_ubo(16, 16, 1, 1), rt.default_scene(), 16, 16, flags, 0, 1, 1)
    got = _static(ubo, sph, w, h, flags, 1, 2, 3)
    assert len(np.unique(want["ids"])) > 5
    _same(got, want)
    assert got["freed"] == 0                        # nothing allocated, nothing released
    # second call, smaller on every axis: the reference's 3 spheres, 96x64, 1 frame x 3 spp
    w2, h2 = 96, 64
    ubo2 = rt.make_ubo(w2, h2, 3, 7)
    sph2 = rt.default_scene()
    want2 = _reference(ubo2, sph2, w2, h2, flags, 0, 1, 1)
    got2 = _static(ubo2, sph2, w2, h2, flags, 0, 1, 1)
    _same(got2, want2)
    assert got2["freed"] == 0
    # and the first one again, now after a smaller one
    _same(_static(ubo, sph, w, h, flags, 1, 2, 3), want)


def test_static_entry_point_on_a_side_stream_and_degenerate_counts():
    import torch
    w, h = 80, 48
    sph = rt.default_scene()
    side = torch.cuda.Stream(device=torch.device("cuda", 0))
    ubo = rt.make_ubo(w, h, 4, 9)
    _same(_static(ubo, sph, w, h, 0, 2, 3, 1, stream=side), _reference(ubo, sph, w, h, 0, 2, 3, 1))
    for spp, bounces in ((2, 0), (2.5, 3)):         # routed to the nested kernel, as rt_render does
        ubo = rt.make_ubo(w, h, 1, 1)
        ubo[40], ubo[41] = spp, bounces
        _same(_static(ubo, sph, w, h, rt.FLAG_STRICT_MATH, 0, 1, 1), _reference(ubo, sph, w, h, rt.FLAG_STRICT_MATH, 0, 1, 1))
    empty = np.zeros((0, 16), np.float32)
    ubo = rt.make_ubo(w, h, 2, 4)
    _same(_static(ubo, empty, w, h, 0, 0, 1, 1), _reference(ubo, empty, w, h, 0, 0, 1, 1))


def test_static_entry_point_rejects_sizes_beyond_its_scratch():
    import torch
    dev = torch.device("cuda", 0)
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    p = buf.data_ptr()
    ubo = rt.make_ubo(64, 64, 1, 1)
    with pytest.raises(rt.RtError, match="spheres"):
        rt.render_static(64, 64, ubo, p, rt.STATIC_MAX_SPHERES + 1, p, p, p)
    with pytest.raises(rt.RtError, match="tiles"):
        rt.render_static(16 * 400, 16 * 400, ubo, p, 3, p, p, p)      # 160,000 tiles
    with pytest.raises(rt.RtError, match="flags"):
        rt.render_static(64, 64, ubo, p, 3, p, p, p, flags=rt.FLAG_FORCE_BVH)
    with pytest.raises(rt.RtError):
        rt.render_static(64, 64, ubo, p, 3, 0, p, p)
    with pytest.raises(rt.RtError):
        rt.render_static(64, 64, ubo, p, 3, p, p, p, stride=0)
    with pytest.raises(rt.RtError, match="reserved"):
        rt.render_static(64, 64, ubo, p, 3, p, p, p, reserved=7)


def test_static_entry_point_captured_into_a_graph_and_replayed():
    """What the entry point is for: after one ordinary call on the device (which looks the scratch
    array's address up), a call under stream capture records its kernels and its one memset into a
    HIP graph; every replay renders the frames again into the same buffers."""
    import torch
    dev = torch.device("cuda", 0)
    w, h = 120, 72
    ubo = rt.make_ubo(w, h, 18, 7, sun_intensity=1.0)
    sph = rt.grid_spheres_scene(24)
    sph[1:, 2] += 55.0
    want = _reference(ubo, sph, w, h, 0, 3, 2, 1)
    _static(rt.make_ubo(16, 16, 1, 1), rt.default_scene(), 16, 16, 0, 0, 1, 1)     # warm-up, outside capture
    d_sph = torch.from_numpy(np.ascontiguousarray(sph, dtype=np.float32)).to(dev)
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
    rgba8 = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    ids = torch.full((h, w), -1, dtype=torch.int32, device=dev)
    rays = torch.zeros(1, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rt.render_static(w, h, ubo, d_sph.data_ptr(), sph.shape[0], accum.data_ptr(), rgba8.data_ptr(),
                         ids.data_ptr(), rays.data_ptr(), 3, 2, 1, stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    assert int(rays.item()) == 0 and not accum.any().item()      # capture records, it does not run
    graph.replay()
    torch.cuda.synchronize(dev)
    got = dict(accum=accum.cpu().numpy(), rgba8=rgba8.cpu().numpy(), ids=ids.cpu().numpy(), rays=int(rays.item()))
    _same(got, want)
    graph.replay()                                                # the same two frames once more
    torch.cuda.synchronize(dev)
    assert int(rays.item()) == 2 * want["rays"]
    twice = accum.cpu().numpy()
    assert (twice[..., 3] == 4).all()
    # ((a + b) + a) + b against 2 (a + b): the same four addends, another order
    assert np.max(np.abs(twice - 2.0 * want["accum"])) <= 4 * 2.0 ** -23 * np.max(want["accum"])
