"""Known answers of the addressed random function (include/mirt_rng.h): the engine, the oracle and
the golden harness all draw from it, so its values are pinned here independently (pure Python)."""
import ctypes as C

import numpy as np

import oracle_lib as O
from raytracer_amd import default_config

M = (1 << 64) - 1


def mix64(z):
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M
    return z ^ (z >> 31)


def sample_key(seed, pixel, sample):
    k = mix64((seed + 0x9E3779B97F4A7C15 * (pixel + 1)) & M)
    return mix64((k + 0xD1B54A32D192ED03 * (sample + 1)) & M)


def child_key(key, slot):
    return mix64((key + 0x8CB92BA72F3D8DD7 * (slot + 1)) & M)


def draw(key, dim):
    return (mix64((key + 0x9E3779B97F4A7C15 * (dim + 1)) & M) >> 11) / 9007199254740992.0


def test_mix_is_splitmix64_finaliser():
    # splitmix64 with state 0: first output
    assert mix64(0x9E3779B97F4A7C15) == 0xE220A8397B1DCDAF
    assert mix64(0) == 0


def test_draws_are_in_unit_interval_and_distinct():
    k = sample_key(0, 0, 0)
    xs = [draw(k, d) for d in range(64)] + [draw(child_key(k, s), 0) for s in range(64)]
    assert all(0.0 <= x < 1.0 for x in xs) and len(set(xs)) == len(xs)
    assert abs(np.mean([draw(sample_key(7, p, 3), 1) for p in range(4000)]) - 0.5) < 0.02


def test_oracle_jitter_uses_these_draws(host_scenes):
    """A 1x1 frame of an empty scene returns the background whatever the jitter is; a 2-sample
    frame of Test5 changes with the seed and repeats with the same seed."""
    hs = host_scenes("Test5")
    cam, _ = hs.camera(0)
    a, _ = O.render(hs.desc, cam, default_config(max_ray_depth=2, seed=1), 16, 9, 2)
    b, _ = O.render(hs.desc, cam, default_config(max_ray_depth=2, seed=1), 16, 9, 2)
    c, _ = O.render(hs.desc, cam, default_config(max_ray_depth=2, seed=2), 16, 9, 2)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
