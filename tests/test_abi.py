"""The C ABI: the shared library loads, exports every function include/mirt.h declares, and the
ctypes mirror of the structs has the compiler's layout.  No compute calls (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess

from conftest import ROOT
from raytracer_amd import _abi as A

HEADER = os.path.join(ROOT, "include", "mirt.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mirt_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = A.engine_lib()
    names = declared_functions()
    assert set(names) >= set(A.ENGINE_SYMBOLS) and len(names) >= 11
    for n in names:
        assert hasattr(lib, n), f"libmirt.so does not export {n}"


def test_only_the_c_abi_is_exported():
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(A.LIB_DIR, "libmirt.so")],
                         capture_output=True, text=True, check=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    ours = [s for s in syms if not s.startswith(("_init", "_fini", "__hip", "_Z"))]
    assert sorted(ours) == declared_functions()
    assert not [s for s in syms if s.startswith("_Z") and "Mat4" in s]  # host math stays private


def test_struct_layout_matches_the_compiler(tmp_path):
    structs = {"mirt_node": A.Node, "mirt_prim": A.Prim, "mirt_material": A.Material, "mirt_light": A.Light,
               "mirt_texture": A.Texture, "mirt_scene_desc": A.SceneDesc, "mirt_camera": A.Camera,
               "mirt_config": A.Config, "mirt_stats": A.Stats}
    src = tmp_path / "sz.c"
    body = "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in structs)
    body += 'printf("off_seed %zu\\n", offsetof(mirt_config, seed));'
    body += 'printf("off_param %zu\\n", offsetof(mirt_prim, param));'
    body += 'printf("off_textures %zu\\n", offsetof(mirt_scene_desc, textures));'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mirt.h"\nint main(void){' + body + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for n, t in structs.items():
        assert int(got[n]) == C.sizeof(t), n
    assert int(got["mirt_node"]) == 64
    assert int(got["off_seed"]) == A.Config.seed.offset
    assert int(got["off_param"]) == A.Prim.param.offset
    assert int(got["off_textures"]) == A.SceneDesc.textures.offset


def test_abi_version_and_defaults():
    lib = A.engine_lib()
    assert lib.mirt_abi_version() == 1
    cfg = A.Config()
    lib.mirt_config_defaults(C.byref(cfg))
    # RenderConfig defaults (Config.h)
    assert (cfg.max_ray_depth, cfg.shadow_samples, cfg.glossy_samples) == (12, 16, 0)
    assert (cfg.enable_motion_blur, cfg.enable_textures, cfg.use_adaptive_epsilon) == (1, 1, 1)
    assert (cfg.light_intensity_factor, cfg.ambient_factor, cfg.gamma) == (0.2, 1.0, 2.2)
    assert (cfg.ray_offset_epsilon, cfg.adaptive_epsilon_scale, cfg.pure_glass_threshold) == (0.001, 0.0001, 0.99)
    host = A.Config()
    A.host_lib().mirt_host_config_defaults(C.byref(host))
    assert bytes(host) == bytes(cfg)  # the host's RenderConfig() flattens to the same snapshot


def test_tile_buffer_arithmetic():
    f = A.engine_lib().mirt_tile_buffer_pixels
    assert f(1920, 1080, 1920, 1080, 0, 1) == 1920 * 1080
    tiles = 60 * 34  # 32x32 tiles, the last row padded
    assert sum(f(1920, 1080, 32, 32, r, 8) for r in range(8)) == tiles * 32 * 32
    assert f(100, 100, 32, 32, 7, 8) == 2 * 32 * 32  # 16 tiles: ranks 0..7 own two each
    assert f(100, 100, 64, 64, 5, 8) == 0            # 4 tiles only
    assert f(0, 10, 8, 8, 0, 1) == -1 and f(10, 10, 8, 8, 2, 2) == -1


def test_invalid_arguments_fail_without_a_gpu():
    lib = A.engine_lib()
    h = C.c_void_p()
    assert lib.mirt_scene_create(None, 0, C.byref(h)) == -1
    assert b"NULL" in lib.mirt_last_error()
    d = A.SceneDesc()  # no nodes
    assert lib.mirt_scene_create(C.byref(d), 0, C.byref(h)) == -1
    assert b"BVH node" in lib.mirt_last_error()
    nodes = (A.Node * 1)()
    nodes[0].left = 5  # inner node whose children are not in depth-first order
    d.n_nodes, d.nodes = 1, nodes
    assert lib.mirt_scene_create(C.byref(d), 0, C.byref(h)) == -1
    assert b"depth-first" in lib.mirt_last_error()
    lib.mirt_scene_destroy(None)  # harmless
