#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Generates tests/golden/*.npz by running oracle/_ref/ref_harness, i.e. the
reference renderer itself (built in place by oracle/Makefile), on the bundled scenes.

Run in the build container (needs /root/reference to have built oracle/_ref):
    make -C oracle ref && python oracle/make_goldens.py
The outputs are plain numeric arrays (numpy.load(..., allow_pickle=False) reads them).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
FIX = os.path.join(ROOT, "fixtures")
OUT = os.path.join(ROOT, "tests", "golden")
# the reference's eight scenes, Test7 with a glass sphere added (the 64-spp depth-of-field configuration) and a
# hand-written scene whose sphere, cube and plane carry normal maps (no bundled scene has one)
SCENES = ["Test1", "Test2", "Test3", "Test4", "Test5", "Test6", "Test7", "def_scene", "Test7_glass", "NormalMap"]
# Test4 with 600 small spheres added (oracle/make_ties_scene.py): coincident surfaces in a scene large enough
# for the engine's large-scene walk; hits and colours only
TIES = ["Test4_ties"]
W, H = 96, 54  # deterministic (pixel-centre) maps
# generated scenes (raytracer_amd/host/synthetic.h): a deep BVH; the cloud of 10^6 has a script of its own
# (oracle/make_golden_million.py: minutes and gigabytes)
SYNTHETIC = [("synthetic100k", 100000, 0x5EED)]

# stochastic renders: (name, scene, w, h, spp, harness flags); tests/golden_cases.py maps the flags
# to mirt_config fields
RENDERS = [
    ("t5_default", "Test5", 64, 36, 4, ["--max-depth", "5"]),
    ("t5_glossy", "Test5", 64, 36, 2, ["--max-depth", "8", "--glossy", "8", "--shadow-samples", "16"]),
    ("t7_dof_soft", "Test7", 48, 27, 2, ["--max-depth", "8", "--glossy", "4", "--dof", "0.95", "16.12"]),
    ("t6_soft9", "Test6", 48, 27, 2, ["--max-depth", "4", "--shadow-samples", "9"]),
    ("t2_texture", "Test2", 64, 36, 2, ["--max-depth", "4"]),
    ("t3_bump", "Test3", 64, 36, 2, ["--max-depth", "4"]),
    ("t4_glossy3", "Test4", 64, 36, 2, ["--max-depth", "6", "--glossy", "3"]),
    ("t1_cfg0", "Test1", 80, 60, 1, ["--max-depth", "3"]),
    ("t5_nomb_seed7", "Test5", 64, 36, 2, ["--max-depth", "5", "--motion-blur", "0", "--seed", "7"]),
    ("t7g_cfg5", "Test7_glass", 48, 27, 2, ["--max-depth", "8", "--glossy", "4", "--shadow-samples", "16",
                                           "--dof", "0.95", "16.12"]),
    ("nm_default", "NormalMap", 64, 36, 2, ["--max-depth", "4"]),
    ("nm_notex", "NormalMap", 64, 36, 1, ["--max-depth", "3", "--textures", "0"]),
]

# frames of Raytracer::render ITSELF (harness `render-real`: one thread, one random stream through the
# whole frame): (name, scene, w, h, spp, flags); they pin the render loop, tone mapping and gamma
REAL_RENDERS = [
    ("real_t5_none", "Test5", 48, 27, 2, ["--max-depth", "5"]),
    ("real_t5_reinhard", "Test5", 48, 27, 2, ["--max-depth", "5", "--tone-mapping", "reinhard"]),
    ("real_t5_exposure", "Test5", 48, 27, 2, ["--max-depth", "5", "--tone-mapping", "exposure", "--exposure", "1.7"]),
    ("real_t4_exposure_nogamma", "Test4", 48, 27, 1, ["--max-depth", "4", "--tone-mapping", "exposure", "--exposure",
                                                      "2.5", "--gamma-correction", "0"]),
    ("real_t7_dof", "Test7", 32, 18, 1, ["--max-depth", "4", "--shadow-samples", "4", "--dof", "0.95", "16.12"]),
    ("real_nm", "NormalMap", 48, 27, 1, ["--max-depth", "3"]),
]


def harness(*args, root=FIX):
    r = subprocess.run([HARNESS, "--root", root, *args], check=True, capture_output=True)
    return r.stdout, r.stderr.decode()


def main():
    if not os.path.exists(HARNESS):
        sys.exit("oracle/_ref/ref_harness missing: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(OUT, exist_ok=True)
    n = W * H
    dumps = {}
    with tempfile.TemporaryDirectory() as tmp:
        # the generated scenes reach the reference through its own parser, as ASCII files
        sys.path.insert(0, ROOT)
        from raytracer_amd import HostScene
        os.makedirs(os.path.join(tmp, "ASCII"))
        for name, count, seed in SYNTHETIC:
            HostScene.synthetic(count, seed).write_ascii(os.path.join(tmp, "ASCII", f"{name}.txt"))
        for s in SCENES + TIES + [name for name, _, _ in SYNTHETIC]:
            scene = f"ASCII/{s}.txt"
            root = FIX if s in SCENES + TIES else tmp
            if s in SCENES:
                dumps[f"{s}_scene"] = np.frombuffer(harness("dump-scene", scene)[0], dtype=np.uint8)
                dumps[f"{s}_bvh"] = np.frombuffer(harness("dump-bvh", scene)[0], dtype=np.uint8)
            out = os.path.join(tmp, "h.bin")
            harness("hits", scene, str(W), str(H), out, root=root)
            raw = np.fromfile(out, dtype=np.uint8)
            ids = raw[: 4 * n].view(np.int32)
            front = raw[4 * n: 8 * n].view(np.int32)
            rest = raw[8 * n:].view(np.float64)
            harness("trace", scene, str(W), str(H), out, "--max-depth", "5", "--motion-blur", "0", root=root)
            rgb = np.fromfile(out, dtype=np.float64)
            np.savez_compressed(
                os.path.join(OUT, f"hits_{s}.npz"), width=W, height=H, prim_id=ids, front=front, t=rest[:n],
                normal=rest[n: 4 * n].reshape(n, 3), point=rest[4 * n: 7 * n].reshape(n, 3),
                uv=rest[7 * n: 9 * n].reshape(n, 2), trace_rgb_depth5=rgb.reshape(n, 3))
            print(s, "hit fraction", float((ids >= 0).mean()))
        jobs = [("render", "render_", r) for r in RENDERS] + [("render-real", "", r) for r in REAL_RENDERS]
        for cmd, prefix, (name, s, w, h, spp, flags) in jobs:
            out = os.path.join(tmp, "r.bin")
            _, err = harness(cmd, f"ASCII/{s}.txt", str(w), str(h), str(spp), out, *flags)
            prim_tests = int(err.split()[1])
            rgb = np.fromfile(out, dtype=np.float64).reshape(h * w, 3)
            np.savez_compressed(os.path.join(OUT, f"{prefix}{name}.npz"), width=w, height=h, spp=spp, rgb=rgb,
                                prim_tests=prim_tests)
            print(name, "mean", float(rgb.mean()), "prim_tests", prim_tests)
    np.savez_compressed(os.path.join(OUT, "dumps.npz"), **dumps)


if __name__ == "__main__":
    main()
