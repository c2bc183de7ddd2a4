"""The CPU oracle against the golden vectors produced by the reference itself (oracle/_ref/ref_harness,
see oracle/make_goldens.py).  Everything here is bit-exact: the oracle keeps the reference's order of
arithmetic and both are built without FMA contraction."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN, SCENES, SYNTHETIC, TIES
from golden_cases import REAL_CASES, RENDER_CASES
from raytracer_amd import default_config


@pytest.mark.parametrize("name", SCENES + TIES + sorted(SYNTHETIC))
def test_closest_hit_bit_exact(name, host_scenes):
    hs = host_scenes(name)
    gold = np.load(os.path.join(GOLDEN, f"hits_{name}.npz"))
    w, h = int(gold["width"]), int(gold["height"])
    cam, _ = hs.camera(0)
    got = O.trace_ids(hs.desc, cam, w, h)
    assert np.array_equal(got["prim_id"], gold["prim_id"])
    assert np.array_equal(got["front"], gold["front"])
    for k in ("t", "normal", "point", "uv"):
        assert np.array_equal(got[k], gold[k]), k


@pytest.mark.parametrize("name", SCENES + TIES + sorted(SYNTHETIC))
def test_trace_rgb_bit_exact(name, host_scenes):
    """Raytracer::trace of pixel-centre rays at depth 5, motion blur off (sequential draws feed the
    area-light strata of Test6/Test7)."""
    hs = host_scenes(name)
    gold = np.load(os.path.join(GOLDEN, f"hits_{name}.npz"))
    w, h = int(gold["width"]), int(gold["height"])
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=5, enable_motion_blur=0, sampling=1, enable_gamma_correction=0)
    rgb, _ = O.render(hs.desc, cam, cfg, w, h, 1, O.RNG_SEQUENTIAL)
    assert np.array_equal(rgb.reshape(-1, 3), gold["trace_rgb_depth5"])


@pytest.mark.parametrize("case", sorted(RENDER_CASES))
def test_stochastic_render_bit_exact(case, host_scenes):
    """The whole render loop -- jitter, motion-blur time, lens, soft shadows, glossy lobes, glass --
    with the reference's generators fed the same sequential stream."""
    scene, dof, over = RENDER_CASES[case]
    hs = host_scenes(scene)
    gold = np.load(os.path.join(GOLDEN, f"render_{case}.npz"))
    w, h, spp = int(gold["width"]), int(gold["height"]), int(gold["spp"])
    cam, _ = hs.camera(0, dof=dof)
    cfg = default_config(**over)
    rgb, st = O.render(hs.desc, cam, cfg, w, h, spp, O.RNG_SEQUENTIAL)
    assert np.array_equal(rgb.reshape(-1, 3), gold["rgb"])
    assert st["prim_tests"] == int(gold["prim_tests"])  # g_intersection_tests of the reference


def test_stochastic_render_of_the_million_primitive_cloud_bit_exact(host_scenes):
    """The large-scene bench configuration's flags (jitter, motion blur on, depth 5) on its own cloud of 10^6
    primitives, the reference's generators fed the same sequential stream (oracle/make_golden_million.py)."""
    hs = host_scenes("synthetic1m")
    gold = np.load(os.path.join(GOLDEN, "render_s1m_cfg4.npz"))
    w, h, spp = int(gold["width"]), int(gold["height"]), int(gold["spp"])
    cam, _ = hs.camera(0)
    rgb, st = O.render(hs.desc, cam, default_config(max_ray_depth=5), w, h, spp, O.RNG_SEQUENTIAL)
    assert np.array_equal(rgb.reshape(-1, 3), gold["rgb"])
    assert st["prim_tests"] == int(gold["prim_tests"])  # g_intersection_tests of the reference


@pytest.mark.parametrize("case", sorted(REAL_CASES))
def test_frames_of_the_reference_render_function_bit_exact(case, host_scenes):
    """Raytracer::render ITSELF (Raytracer.cpp:475-607: sample loop, mean, Reinhard / exposure tone
    mapping, gamma) run on one thread with its two generators fed one stream through the whole frame,
    against the oracle consuming the same stream in the same order."""
    scene, dof, over = REAL_CASES[case]
    hs = host_scenes(scene)
    gold = np.load(os.path.join(GOLDEN, f"{case}.npz"))
    w, h, spp = int(gold["width"]), int(gold["height"]), int(gold["spp"])
    cam, _ = hs.camera(0, dof=dof)
    cfg = default_config(**over)
    rgb, st = O.render(hs.desc, cam, cfg, w, h, spp, O.RNG_FRAME)
    assert np.array_equal(rgb.reshape(-1, 3), gold["rgb"])
    assert st["prim_tests"] == int(gold["prim_tests"])


def test_a_window_is_a_crop_of_the_frame(host_scenes):
    hs = host_scenes("Test5")
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=5, glossy_samples=2)
    for mode in (O.RNG_KEYED, O.RNG_SEQUENTIAL):
        full, _ = O.render(hs.desc, cam, cfg, 64, 36, 2, mode)
        win, st = O.render_window(hs.desc, cam, cfg, 64, 36, 2, 20, 9, 17, 11, mode)
        assert np.array_equal(win, full[9:20, 20:37])
        assert st["primary_rays"] == 17 * 11 * 2


def test_keyed_and_sequential_agree_without_randomness(host_scenes):
    """With nothing random in the frame the two addressings of the random function must coincide."""
    hs = host_scenes("Test5")
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=4, enable_motion_blur=0, sampling=1)
    a, _ = O.render(hs.desc, cam, cfg, 48, 27, 1, O.RNG_SEQUENTIAL)
    b, _ = O.render(hs.desc, cam, cfg, 48, 27, 1, O.RNG_KEYED)
    assert np.array_equal(a, b)


def test_addressed_draws_give_the_reference_programs_image(host_scenes, tmp_path):
    """The oracle with ADDRESSED draws (the mode the GPU is compared with) against a frame of the
    reference program itself at the same sample count: two independent Monte-Carlo estimates, so the
    bound is statistical (mean absolute error of the 8-bit image <= 0.01 of full scale at 64 spp)."""
    import subprocess

    from conftest import FIXTURES, ROOT
    from raytracer_amd import read_ppm
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "raytracer_ref")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/raytracer_ref not built")
    w, h, spp = 160, 90, 64
    os.symlink(os.path.join(FIXTURES, "ASCII"), tmp_path / "ASCII")
    os.symlink(os.path.join(FIXTURES, "Textures"), tmp_path / "Textures")
    out = tmp_path / "ref.ppm"
    subprocess.run([ref_bin, "--scene", "ASCII/Test5.txt", "--resolution", str(w), str(h), "--samples", str(spp),
                    "--max-depth", "5", "--output", str(out)], cwd=tmp_path, check=True, capture_output=True, timeout=300)
    ref, _ = read_ppm(str(out))
    hs = host_scenes("Test5")
    cam, _ = hs.camera(0)
    got, _ = O.render(hs.desc, cam, default_config(max_ray_depth=5), w, h, spp, O.RNG_KEYED)
    got8 = np.floor(np.clip(got, 0.0, 1.0) * 255.0 + 0.5) / 255.0
    mae = float(np.abs(got8 - ref.reshape(h, w, 3)).mean())
    print(f"oracle (addressed draws) vs reference program: MAE {mae:.5f}")
    assert mae <= 0.01
