"""The entry point that allocates nothing (mirt_render_tiles_noalloc) against mirt_render_tiles:
same frame (to the order of the fp64 pixel sums: <= 1e-12), same ray tree, same statistics -- on two
calls in a row on the GPU's shared static array, the second with a smaller frame -- and a clear
error for a frame that does not fit."""
import numpy as np
import pytest

from raytracer_amd import MirtError, default_config

pytestmark = pytest.mark.gpu

COUNTERS = ("primary_rays", "secondary_rays", "shadow_segments", "node_tests", "prim_tests", "shadow_node_tests",
            "shadow_prim_tests")


def test_noalloc_entry_point_matches_render_tiles(host_scenes, device_scenes):
    import torch
    from raytracer_amd import _abi as A
    assert A.engine_lib().mirt_noalloc_scratch_bytes() == 64 << 20
    hs, ds = host_scenes("Test5"), device_scenes("Test5")
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=5, glossy_samples=4)  # glass, mirrors, glossy lobes, moving primitives
    dev = torch.device("cuda", 0)
    # first a frame large enough that the static queues force several primary launches, then a smaller one
    for (w, h, spp) in ((640, 360, 4), (200, 120, 2)):
        want = torch.empty(w * h * 3, dtype=torch.float64, device=dev)
        got = torch.full((w * h * 3,), -7.0, dtype=torch.float64, device=dev)
        st_want = ds.render_tiles(cam, cfg, w, h, spp, w, h, 0, 1, want.data_ptr())
        ds.render_tiles_noalloc(cam, cfg, w, h, spp, w, h, 0, 1, got.data_ptr())  # (the first call uploads the light table)
        torch.cuda.synchronize(dev)
        free_before = torch.cuda.mem_get_info(dev)[0]
        st_got = ds.render_tiles_noalloc(cam, cfg, w, h, spp, w, h, 0, 1, got.data_ptr())
        assert torch.cuda.mem_get_info(dev)[0] == free_before  # the call took no device memory
        err = float((got - want).abs().max())
        print(f"{w}x{h}x{spp}: max abs diff {err:.3e}, launches {st_want['launches']} vs {st_got['launches']}")
        assert err <= 1e-12
        for k in COUNTERS:
            assert st_got[k] == st_want[k], k
        assert st_got["frame_attempts"] == 1
    # a tile split through the same entry point, other scene handle on the same GPU (same static array)
    hs2, ds2 = host_scenes("Test4"), device_scenes("Test4")
    cam2, _ = hs2.camera(0)
    cfg2 = default_config(max_ray_depth=4)
    from raytracer_amd import tile_buffer_pixels
    w, h, spp, tile, world = 100, 70, 2, 16, 3
    for r in range(world):
        n = tile_buffer_pixels(w, h, tile, tile, r, world) * 3
        want = torch.empty(n, dtype=torch.float64, device=dev)
        got = torch.full((n,), -7.0, dtype=torch.float64, device=dev)
        ds2.render_tiles(cam2, cfg2, w, h, spp, tile, tile, r, world, want.data_ptr())
        ds2.render_tiles_noalloc(cam2, cfg2, w, h, spp, tile, tile, r, world, got.data_ptr())
        assert float((got - want).abs().max()) <= 1e-12
    # beyond the static array: refused, with the limit named, and the handle stays usable
    big = torch.empty(16, dtype=torch.float64, device=dev)  # never written: the call fails before any launch
    with pytest.raises(MirtError, match="static working memory"):
        ds.render_tiles_noalloc(cam, cfg, 3840, 2160, 16, 3840, 2160, 0, 1, big.data_ptr())
    got = torch.empty(64 * 36 * 3, dtype=torch.float64, device=dev)
    want = torch.empty(64 * 36 * 3, dtype=torch.float64, device=dev)
    ds.render_tiles_noalloc(cam, cfg, 64, 36, 1, 64, 36, 0, 1, got.data_ptr())
    ds.render_tiles(cam, cfg, 64, 36, 1, 64, 36, 0, 1, want.data_ptr())
    assert float((got - want).abs().max()) <= 1e-12
