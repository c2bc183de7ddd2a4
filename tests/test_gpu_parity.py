"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle and the committed goldens."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN, SCENES
from raytracer_amd import default_config

pytestmark = pytest.mark.gpu

W, H = 96, 54  # size of the committed golden maps


@pytest.mark.parametrize("name", SCENES)
def test_closest_hit_vs_golden_and_oracle(name, host_scenes, device_scenes):
    """Primitive id, distance, normal and uv of pixel-centre rays (mirt_trace_ids)."""
    hs, ds = host_scenes(name), device_scenes(name)
    cam, _ = hs.camera(0)
    got = ds.trace_ids(cam, W, H)
    gold = np.load(os.path.join(GOLDEN, f"hits_{name}.npz"))
    ref = O.trace_ids(hs.desc, cam, W, H)
    assert np.array_equal(ref["prim_id"], gold["prim_id"])  # the oracle is pinned to the reference
    agree = float((got["prim_id"] == gold["prim_id"]).mean())
    print(f"{name}: prim-id agreement {agree:.6f}")
    assert agree >= 0.9995  # required: >= 0.999
    same = got["prim_id"] == gold["prim_id"]
    hit = same & (gold["prim_id"] >= 0)
    assert np.allclose(got["t"][hit], gold["t"][hit], rtol=1e-9, atol=1e-9)
    assert np.allclose(got["normal"][hit], gold["normal"][hit], atol=1e-7)
    assert np.allclose(got["uv"][hit], gold["uv"][hit], atol=1e-7)


@pytest.mark.parametrize("name", SCENES)
def test_deterministic_render_vs_golden(name, host_scenes, device_scenes):
    """Pixel-centre, time 0.5, depth 5 trace() colours against the reference's own values."""
    hs, ds = host_scenes(name), device_scenes(name)
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=5, enable_motion_blur=0, sampling=1, enable_gamma_correction=0)
    gold = np.load(os.path.join(GOLDEN, f"hits_{name}.npz"))["trace_rgb_depth5"].reshape(H, W, 3)
    if name in ("Test6", "Test7"):
        # area lights draw random strata: the golden used the reference's sequential draws, the
        # engine uses addressed draws, so compare with the oracle in addressed mode instead
        gold, _ = O.render(hs.desc, cam, cfg, W, H, 1, O.RNG_KEYED)
    got, st = ds.render(cam, cfg, W, H, 1)
    p = O.psnr(got, gold)
    err = np.abs(got - gold).max()
    print(f"{name}: PSNR {p:.1f} dB, max abs err {err:.3e}, stats {st}")
    assert p >= 80.0  # required: >= 50 dB
