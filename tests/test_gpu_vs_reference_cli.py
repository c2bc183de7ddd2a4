"""GPU frames against frames of the reference program itself (oracle/_ref/raytracer_ref, the reference's
own main() built in place by oracle/Makefile and shipped to the GPU box with the tree), at equal sample
counts.  The reference seeds its generators from std::random_device, so the two frames are independent
Monte-Carlo estimates of the same image: the comparison is statistical, with the bound stated per case.
The deterministic, bit-level comparisons with the reference go through the golden vectors
(test_gpu_parity.py, test_oracle_golden.py)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import FIXTURES, ROOT
from raytracer_amd import default_config, read_ppm

pytestmark = pytest.mark.gpu

REF = os.path.join(ROOT, "oracle", "_ref", "raytracer_ref")

# name -> (scene, reference CLI flags, config overrides, dof, spp, bound on the mean absolute error of the
# 8-bit image, as a fraction of full scale).  Two independent 64-sample estimates differ by about
# sigma * sqrt(2 / 64) per pixel; the bounds leave a factor of ~3 over what was observed.
CASES = {
    "test5_motion_blur": ("Test5", ["--max-depth", "5"], dict(max_ray_depth=5), None, 64, 0.010),
    "test5_glossy": ("Test5", ["--max-depth", "6", "--glossy-reflection", "4"],
                     dict(max_ray_depth=6, glossy_samples=4), None, 64, 0.012),
    "test7_dof_soft_shadows": ("Test7", ["--max-depth", "5", "--soft-shadows", "16", "--depth-of-field", "0.95", "16.12"],
                               dict(max_ray_depth=5, shadow_samples=16), (0.95, 16.12), 64, 0.015),
    "test2_textures": ("Test2", ["--max-depth", "4"], dict(max_ray_depth=4), None, 32, 0.010),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_frame_matches_reference_program(case, host_scenes, device_scenes, tmp_path):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/raytracer_ref not built (needs the reference sources at build time)")
    scene, flags, over, dof, spp, bound = CASES[case]
    w, h = 160, 90
    # the reference resolves Textures/ and writes its row-time CSV relative to the working directory
    os.symlink(os.path.join(FIXTURES, "ASCII"), tmp_path / "ASCII")
    os.symlink(os.path.join(FIXTURES, "Textures"), tmp_path / "Textures")
    out = tmp_path / "ref.ppm"
    subprocess.run([REF, "--scene", f"ASCII/{scene}.txt", "--resolution", str(w), str(h), "--samples", str(spp),
                    "--output", str(out), *flags], cwd=tmp_path, check=True, capture_output=True, timeout=300)
    ref, _ = read_ppm(str(out))
    hs, ds = host_scenes(scene), device_scenes(scene)
    cam, _ = hs.camera(0, dof=dof)
    got, _ = ds.render(cam, default_config(**over), w, h, spp)
    got8 = np.floor(np.clip(got, 0.0, 1.0) * 255.0 + 0.5) / 255.0  # what the PPM writer keeps (Image.cpp:123-128)
    mae = float(np.abs(got8 - ref.reshape(h, w, 3)).mean())
    bias = float((got8 - ref.reshape(h, w, 3)).mean())
    print(f"{case}: MAE {mae:.5f} (bound {bound}), mean difference {bias:+.5f}")
    assert mae <= bound
    assert abs(bias) <= bound / 4  # no systematic brightness difference
