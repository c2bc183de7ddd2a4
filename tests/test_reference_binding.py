"""The reference-side binding (integration/reference_binding.cpp, raytracer_amd/host/flatten_impl.h).

The flattening templates are instantiated for the REFERENCE's own classes by oracle/Makefile (which
compiles against the reference's headers where they lie and links the reference's own load_scene):
  oracle/_ref/flatten_dump       reference load_scene -> templates -> every array, dumped
  oracle/_ref/raytracer_ref_gpu  the reference's main() with Render/Raytracer.cpp replaced by the binding
Here: the dump equals, byte for byte, what the drop-in host library makes of the same scene file (CPU);
and the reference's command line, bound to the engine, writes the image the drop-in command line writes (GPU).
Both binaries travel with the tree; where they were never built the tests skip."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import FIXTURES, ROOT, SCENES, scene_path
from raytracer_amd import HostScene, read_ppm
from raytracer_amd import _abi as A

DUMP = os.path.join(ROOT, "oracle", "_ref", "flatten_dump")
REF_GPU = os.path.join(ROOT, "oracle", "_ref", "raytracer_ref_gpu")


def read_dump(path):
    out, data = [], open(path, "rb").read()
    at = 0
    while at < len(data):
        nl = data.index(b"\n", at)
        name, size = data[at:nl].split()
        size = int(size)
        out.append((name.decode(), data[nl + 1: nl + 1 + size]))
        at = nl + 1 + size + 1
    return out


def raw(ptr, count, ctype):
    return C.string_at(ptr, count * C.sizeof(ctype)) if count else b""


@pytest.mark.skipif(not os.path.exists(DUMP), reason="oracle/_ref/flatten_dump not built (needs /root/reference)")
@pytest.mark.parametrize("name", SCENES)
def test_templates_on_the_reference_classes_equal_the_drop_in(name, tmp_path):
    out = tmp_path / "dump.bin"
    r = subprocess.run([DUMP, scene_path(name), FIXTURES, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = read_dump(out)
    hs = HostScene(scene_path(name), FIXTURES)
    d = hs.desc
    want = [("nodes", raw(d.nodes, d.n_nodes, A.Node)), ("leaf_prims", raw(d.leaf_prims, d.n_leaf_prims, C.c_int32)),
            ("prims", raw(d.prims, d.n_prims, A.Prim)), ("plane_points", raw(d.plane_points, 3 * d.n_plane_points, C.c_double)),
            ("materials", raw(d.materials, d.n_materials, A.Material)), ("lights", raw(d.lights, d.n_lights, A.Light))]
    for i in range(d.n_textures):
        t = d.textures[i]
        want.append(("texture_dims", np.array([t.width, t.height, t.max_value], np.int32).tobytes()))
        want.append(("texels", raw(t.rgb, t.width * t.height * 3, C.c_uint16)))
    want.append(("background", np.array([*d.background_color, d.background_strength]).tobytes()))
    cam, _ = hs.camera(0)
    want.append(("camera", bytes(cam)))
    cfg = A.Config()
    A.host_lib().mirt_host_config_defaults(C.byref(cfg))
    want.append(("config", bytes(cfg)))
    assert [n for n, _ in got] == [n for n, _ in want]
    for (n, a), (_, b) in zip(got, want):
        assert a == b, f"{name}: array {n} differs ({len(a)} vs {len(b)} bytes)"
    assert d.n_prims > 0 and d.n_nodes > 0


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_GPU), reason="oracle/_ref/raytracer_ref_gpu not built (needs /root/reference)")
@pytest.mark.parametrize("name,flags", [("Test1", ["--samples", "1", "--max-depth", "3"]),
                                        ("Test2", ["--samples", "2", "--max-depth", "3"]),
                                        ("Test5", ["--samples", "2", "--max-depth", "4", "--glossy-reflection", "2"])])
def test_reference_command_line_on_the_engine(name, flags, tmp_path):
    """The reference's own main(), flag parser, load_scene, classes and PPM writer, with the engine behind
    render_scene_bvh_antialiased: the image is the one the drop-in command line writes for the same flags."""
    cli = os.path.join(ROOT, "raytracer_amd", "bin", "raytracer")
    if not os.path.exists(cli):
        pytest.skip("drop-in CLI not built")
    args = ["--scene", f"ASCII/{name}.txt", "--resolution", "320", "240", *flags]
    a, b = str(tmp_path / "ref.ppm"), str(tmp_path / "dropin.ppm")
    r = subprocess.run([REF_GPU, *args, "--output", a], cwd=FIXTURES, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([cli, *args, "--output", b], cwd=FIXTURES, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    with open(a, "rb") as f:
        assert f.read(2) == b"P3"  # the reference's own writer (Image.cpp:184-204)
    ia, _ = read_ppm(a)
    ib, _ = read_ppm(b)
    assert ia.shape == ib.shape == (240, 320, 3)
    # same rays, same seed: only the order of the fp64 sums differs, which can move a value across a rounding boundary
    assert (np.abs(ia - ib) > 1e-9).mean() < 1e-4
    assert 0.01 < ia.mean() < 0.99
