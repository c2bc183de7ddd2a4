"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle and the committed goldens."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import FIXTURES, GOLDEN, SCENES, SYNTHETIC
from raytracer_amd import default_config

pytestmark = pytest.mark.gpu

W, H = 96, 54  # size of the committed golden maps


@pytest.mark.parametrize("traversal", ["packet", "lane", "ordered", "wide"])
@pytest.mark.parametrize("name", SCENES + sorted(SYNTHETIC))
def test_closest_hit_vs_golden_and_oracle(name, traversal, host_scenes, device_scenes):
    """Primitive id, distance, normal and uv of pixel-centre rays (mirt_trace_ids), with either BVH walk."""
    hs, ds = host_scenes(name), device_scenes(name, traversal)
    cam, _ = hs.camera(0)
    got = ds.trace_ids(cam, W, H)
    gold = np.load(os.path.join(GOLDEN, f"hits_{name}.npz"))
    ref = O.trace_ids(hs.desc, cam, W, H)
    assert np.array_equal(ref["prim_id"], gold["prim_id"])  # the oracle is pinned to the reference
    same = got["prim_id"] == gold["prim_id"]
    hit = same & (gold["prim_id"] >= 0)
    # Everything but the torus is arithmetic-for-arithmetic the reference's and must agree to rounding.
    # The torus quartic runs in x87 long double in the reference and in double here, and the root is
    # then polished by a Newton iteration that stops at a step of 1e-6 (Torus.cpp:220-254).  For a
    # torus a few hundredths across seen from hundreds of units away (the generated cloud) even long
    # double loses the roots, and the reference reports hit points units away from the torus; those
    # pixels count as disagreements here.
    kinds = np.array([hs.desc.prims[i].type for i in range(hs.desc.n_prims)] + [-1])
    torus = kinds[gold["prim_id"]] == 3
    dt = np.abs(got["t"] - gold["t"])
    close = np.where(torus, dt <= 2e-5 * np.maximum(1.0, gold["t"]), dt <= 1e-9 * np.maximum(1.0, gold["t"]))
    assert close[hit & ~torus].all()
    agree = float((same & (close | ~hit)).mean())
    print(f"{name}: prim-id agreement {float(same.mean()):.6f}, with distance {agree:.6f}")
    assert float(same.mean()) >= 0.999  # the required bar
    assert agree >= 0.998
    if name not in SYNTHETIC:
        assert float(same.mean()) >= 0.9995 and close[hit].all()
    ok = hit & close
    for sel, tol in ((ok & ~torus, 1e-7), (ok & torus, 2e-3)):
        assert np.allclose(got["normal"][sel], gold["normal"][sel], atol=tol)
        assert np.allclose(got["uv"][sel], gold["uv"][sel], atol=tol)


@pytest.mark.parametrize("traversal", ["packet", "lane", "ordered", "wide"])
@pytest.mark.parametrize("name", SCENES + sorted(SYNTHETIC))
def test_deterministic_render_vs_golden(name, traversal, host_scenes, device_scenes):
    """Pixel-centre, time 0.5, depth 5 trace() colours against the reference's own values."""
    if traversal == "ordered" and name in ("Test4", "Test6"):
        # these two scenes have coincident surfaces (a glass cylinder standing on the ground plane, ...):
        # which of two hits at the same distance wins depends on the order of the tests, and the binary
        # walk that visits the nearer child first (opt-in only) does not keep the reference's order; the
        # wide walk does not either, but walks such rays again in that order
        pytest.skip("ties are order dependent")
    hs, ds = host_scenes(name), device_scenes(name, traversal)
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=5, enable_motion_blur=0, sampling=1, enable_gamma_correction=0)
    gold = np.load(os.path.join(GOLDEN, f"hits_{name}.npz"))["trace_rgb_depth5"].reshape(H, W, 3)
    if any(hs.desc.lights[i].type == 1 for i in range(hs.desc.n_lights)):
        # area lights draw random strata: the golden used the reference's sequential draws, the
        # engine uses addressed draws, so compare with the oracle in addressed mode instead
        gold, _ = O.render(hs.desc, cam, cfg, W, H, 1, O.RNG_KEYED)
    got, st = ds.render(cam, cfg, W, H, 1)
    p = O.psnr(got, gold)
    err = np.abs(got - gold).max()
    print(f"{name}: PSNR {p:.1f} dB, max abs err {err:.3e}, stats {st}")
    assert p >= 80.0  # required: >= 50 dB


from golden_cases import REAL_CASES, RENDER_CASES  # noqa: E402
from raytracer_amd import DeviceScene, HostScene, scatter_tiles, tile_buffer_pixels  # noqa: E402
from conftest import scene_path, FIXTURES  # noqa: E402


@pytest.mark.parametrize("case", sorted(RENDER_CASES))
def test_stochastic_render_vs_oracle(case, host_scenes, device_scenes):
    """Jitter, motion-blur time, lens, soft shadows, glossy lobes, glass: same addressed draws on
    both sides, so the frames agree to rounding.  Stated bound: PSNR >= 60 dB and mean absolute
    error <= 1e-6 at equal sample counts (the north star asks for >= 50 dB)."""
    scene, dof, over = RENDER_CASES[case]
    hs, ds = host_scenes(scene), device_scenes(scene)
    gold = np.load(os.path.join(GOLDEN, f"render_{case}.npz"))
    w, h, spp = int(gold["width"]), int(gold["height"]), int(gold["spp"])
    cam, _ = hs.camera(0, dof=dof)
    cfg = default_config(**over)
    ref, ost = O.render(hs.desc, cam, cfg, w, h, spp, O.RNG_KEYED)
    got, st = ds.render(cam, cfg, w, h, spp)
    p, mae = O.psnr(got, ref), float(np.abs(got - ref).mean())
    print(f"{case}: PSNR {p:.1f} dB, MAE {mae:.3e}, max {np.abs(got - ref).max():.3e}")
    assert st["primary_rays"] == ost["primary_rays"] == w * h * spp
    # the ray tree is the oracle's, up to torus silhouettes (the device solves the quartic in double,
    # the reference in x87 long double): a grazing secondary ray may hit on one side and miss on the other
    assert abs(st["secondary_rays"] - ost["secondary_rays"]) <= max(2, 1e-3 * ost["secondary_rays"])
    assert st["shadow_segments"] <= ost["shadow_segments"] * 1.001  # lights behind the surface are not traced
    assert p >= 60.0 and mae <= 1e-6
    # against the reference's own frame (sequential draws): same estimator, different samples
    stat_mae = float(np.abs(np.clip(got, 0, 1).reshape(-1, 3) - np.clip(gold["rgb"], 0, 1)).mean())
    print(f"{case}: MAE vs the reference's own draws {stat_mae:.4f}")
    assert stat_mae <= 0.08


@pytest.mark.parametrize("case", sorted(REAL_CASES))
def test_tone_mapping_and_render_loop_vs_oracle(case, host_scenes, device_scenes):
    """The configurations whose frames of Raytracer::render itself pin the oracle (Reinhard and exposure
    tone mapping, gamma off, depth of field, normal maps), GPU against oracle with addressed draws."""
    scene, dof, over = REAL_CASES[case]
    hs, ds = host_scenes(scene), device_scenes(scene)
    cam, _ = hs.camera(0, dof=dof)
    cfg = default_config(**over)
    ref, _ = O.render(hs.desc, cam, cfg, 64, 36, 2, O.RNG_KEYED)
    got, _ = ds.render(cam, cfg, 64, 36, 2)
    p, mae = O.psnr(got, ref), float(np.abs(got - ref).mean())
    print(f"{case}: PSNR {p:.1f} dB, MAE {mae:.3e}")
    assert p >= 60.0 and mae <= 1e-6
    if over.get("tone_mapping_mode") == 2:
        assert got.max() <= 1.0 + 1e-12  # min(1, exposure * c) before gamma (Raytracer.cpp:552-559)


@pytest.mark.parametrize("name", SCENES)
def test_random_rays_vs_oracle(name, host_scenes, device_scenes):
    """Rays from anywhere in any direction at any shutter time (inside objects, grazing, from below)."""
    hs, ds = host_scenes(name), device_scenes(name)
    rng = np.random.default_rng(11)
    n = 100000
    root = hs.desc.nodes[0]
    lo, hi = np.array(root.bmin[:]), np.array(root.bmax[:])
    o = lo - 0.1 * (hi - lo) + 1.2 * (hi - lo) * rng.random((n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tm = rng.random(n)
    ref, got = O.trace_rays(hs.desc, o, d, tm), ds.trace_rays(o, d, tm)
    # rays that start inside overlapping cubes report the range start for each of them
    # (Cube.cpp:31-59): an exact tie whose winner is decided by the last bit
    tie = (ref["t"] < 2e-5) & (ref["prim_id"] >= 0)
    same = (ref["prim_id"] == got["prim_id"]) | tie
    print(f"{name}: agreement {same.mean():.6f} over {n} rays, {int((ref['prim_id'] >= 0).sum())} hits")
    assert same.mean() >= 0.9999
    ok = (ref["prim_id"] == got["prim_id"]) & (ref["prim_id"] >= 0)
    assert np.allclose(got["t"][ok], ref["t"][ok], rtol=1e-8, atol=1e-9)
    assert np.allclose(got["normal"][ok], ref["normal"][ok], atol=1e-6)


@pytest.mark.parametrize("case", ["t5_glossy", "t7_dof_soft", "t2_texture", "t1_cfg0", "t4_glossy3"])
def test_lane_walk_renders_what_packet_walk_renders(case, host_scenes, device_scenes):
    """Same rays, same tests per ray, same addressed draws: the two BVH walks differ only in the
    order in which the fp64 atomics reach a pixel.  (Test1 and Test5 have point lights only and no attenuating
    glass: their packet-walk shadow items run as one any-hit segment each, the lane walk's as chained closest
    hits; Test4's glass attenuates, Test7's lights are areas.)"""
    scene, dof, over = RENDER_CASES[case]
    hs = host_scenes(scene)
    cam, _ = hs.camera(0, dof=dof)
    cfg = default_config(**over)
    a, sa = device_scenes(scene, "packet").render(cam, cfg, 64, 36, 2)
    b, sb = device_scenes(scene, "lane").render(cam, cfg, 64, 36, 2)
    assert np.abs(a - b).max() <= 1e-12
    for k in ("primary_rays", "secondary_rays", "node_tests", "prim_tests"):
        assert sa[k] == sb[k], k
    # shadow segments: where every primitive is opaque or perfectly clear the packet walk stops a lane at its
    # first occluder and never tests (or chains through) a clear primitive; the lane walk chains closest hits
    any_hit = device_scenes(scene, "packet").info()["shadow_any_hit"]
    for k in ("shadow_segments", "shadow_node_tests", "shadow_prim_tests"):
        assert (sa[k] <= sb[k]) if any_hit else (sa[k] == sb[k]), k


def test_small_queues_split_the_frame_without_changing_it(host_scenes, monkeypatch):
    """With queues far too small for the frame the engine cuts it into many primary launches, renders
    it again with finer launches when a queue shard overflows anyway, and returns the same frame."""
    hs = host_scenes("Test4")  # mirrors and glass: with 8 glossy samples some hits spawn 10 rays
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=6, glossy_samples=8)
    w, h, spp = 320, 180, 4
    want, st0 = DeviceScene(hs, 0).render(cam, cfg, w, h, spp)
    monkeypatch.setenv("MIRT_RAY_QUEUE", "65536")
    monkeypatch.setenv("MIRT_SHADOW_QUEUE", "262144")
    small = DeviceScene(hs, 0)
    got, st1 = small.render(cam, cfg, w, h, spp)
    again, st2 = small.render(cam, cfg, w, h, spp)  # the split found for this frame shape is remembered
    print("launches", st0["launches"], st1["launches"], st2["launches"], "attempts", st1["frame_attempts"],
          st2["frame_attempts"], "rays per sample", (st0["primary_rays"] + st0["secondary_rays"]) / st0["primary_rays"])
    assert st0["frame_attempts"] == 1 and st1["frame_attempts"] > 1 and st2["frame_attempts"] == 1
    assert st1["launches"] > 4 * st0["launches"]
    for got_i, st in ((got, st1), (again, st2)):
        assert np.abs(got_i - want).max() <= 1e-12
        for k in ("primary_rays", "secondary_rays", "shadow_segments", "node_tests", "prim_tests"):
            assert st[k] == st0[k], k


def test_two_frames_in_flight_equal_two_frames_in_turn(host_scenes, device_scenes):
    """mirt_render_tiles_begin / _end with both working-memory sets busy on two streams."""
    import torch
    hs, ds = host_scenes("Test5"), device_scenes("Test5")
    cam, _ = hs.camera(0)
    w, h, spp = 160, 90, 2
    cfgs = [default_config(max_ray_depth=5, seed=11), default_config(max_ray_depth=5, seed=12, glossy_samples=4)]
    want = [ds.render(cam, c, w, h, spp) for c in cfgs]
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    bufs = [torch.empty(w * h * 3, dtype=torch.float64, device=dev) for _ in range(2)]
    for rounds in range(3):  # the sets are reused frame after frame
        for k in range(2):
            ds.render_tiles_begin(cam, cfgs[k], w, h, spp, w, h, 0, 1, bufs[k].data_ptr(), streams[k].cuda_stream, k)
        for k in (1, 0):
            st = ds.render_tiles_end(k)
            got = bufs[k].cpu().numpy().reshape(h, w, 3)
            assert np.abs(got - want[k][0]).max() <= 1e-12
            for key in ("primary_rays", "secondary_rays", "shadow_segments", "node_tests", "prim_tests"):
                assert st[key] == want[k][1][key], key
    with pytest.raises(RuntimeError):
        ds.render_tiles_end(0)  # nothing in flight


def test_static_geometry_is_bit_exact(host_scenes, device_scenes):
    """Distances and normals of every primitive kind but the torus equal the reference's bit for
    bit (device arithmetic is not contracted into FMAs and keeps the reference's order)."""
    for name in ("Test1", "Test4", "def_scene"):
        hs, ds = host_scenes(name), device_scenes(name)
        cam, _ = hs.camera(0)
        gold = np.load(os.path.join(GOLDEN, f"hits_{name}.npz"))
        got = ds.trace_ids(cam, W, H)
        types = np.array([hs.desc.prims[i].type for i in range(hs.desc.n_prims)])
        sel = (gold["prim_id"] >= 0) & (types[np.maximum(gold["prim_id"], 0)] != 3)
        assert np.array_equal(got["t"][sel], gold["t"][sel])
        assert np.array_equal(got["normal"][sel], gold["normal"][sel])


def test_tile_split_equals_whole_frame(host_scenes, device_scenes):
    hs, ds = host_scenes("Test5"), device_scenes("Test5")
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=4)
    w, h, spp, tile, world = 100, 70, 2, 16, 3
    whole, st = ds.render(cam, cfg, w, h, spp)
    import torch
    frame = np.zeros((h, w, 3))
    rays = 0
    for r in range(world):
        npx = tile_buffer_pixels(w, h, tile, tile, r, world)
        buf = torch.full((npx * 3,), -1.0, dtype=torch.float64, device="cuda:0")
        s = ds.render_tiles(cam, cfg, w, h, spp, tile, tile, r, world, buf.data_ptr())
        rays += s["primary_rays"] + s["secondary_rays"]
        scatter_tiles(frame, buf.cpu().numpy(), w, h, tile, tile, r, world)
    assert rays == st["primary_rays"] + st["secondary_rays"]
    assert np.allclose(frame, whole, rtol=0, atol=1e-12)  # same rays; only the order of the pixel sums varies


def test_edge_cases(host_scenes, device_scenes, tmp_path):
    hs, ds = host_scenes("Test5"), device_scenes("Test5")
    cam, _ = hs.camera(0)
    # depth 0: trace() returns black before touching the scene (Raytracer.cpp:297)
    img, st = ds.render(cam, default_config(max_ray_depth=0), 16, 9, 1)
    assert np.array_equal(img, np.zeros((9, 16, 3))) and st["primary_rays"] == 0
    # one pixel, one sample
    img, _ = ds.render(cam, default_config(max_ray_depth=3), 1, 1, 1)
    ref, _ = O.render(hs.desc, cam, default_config(max_ray_depth=3), 1, 1, 1)
    assert np.allclose(img, ref, atol=1e-9)
    # a scene without objects or lights renders its background
    empty = tmp_path / "empty.txt"
    empty.write_text("SCENE_SETTINGS\nbackground_color 0.25 0.5 1.0\nbackground_strength 0.5\n\nCAMERAS 1\n"
                     "name c\nlocation 0 0 5\ngaze 0 0 -1\nup 0 1 0\nfocal 35\nsensor 36 24\nresolution 8 8\n")
    he = HostScene(str(empty), FIXTURES)
    de = DeviceScene(he, 0)
    ce, _ = he.camera(0)
    img, _ = de.render(ce, default_config(enable_gamma_correction=0), 8, 8, 2)
    assert np.allclose(img, np.broadcast_to([0.125, 0.25, 0.5], (8, 8, 3)), atol=1e-15)
    # ... also with every ray-spawning feature switched on: lanes without a hit read no material
    img, st = de.render(ce, default_config(enable_gamma_correction=0, glossy_samples=4, max_ray_depth=3), 8, 8, 2)
    assert np.allclose(img, np.broadcast_to([0.125, 0.25, 0.5], (8, 8, 3)), atol=1e-15)
    assert st["secondary_rays"] == 0 and st["shadow_segments"] == 0
    # invisible objects stay in the BVH but are never hit (Bvh.cpp:255-278)
    hv = HostScene(scene_path("Test5"), FIXTURES)
    for i in range(hv.desc.n_prims):
        hv.desc.prims[i].visible = 0
    dv = DeviceScene(hv, 0)
    assert (dv.trace_ids(cam, 32, 18)["prim_id"] == -1).all()
    # bad arguments are errors, not crashes
    from raytracer_amd import MirtError
    with pytest.raises(MirtError):
        ds.render(cam, default_config(), 0, 10, 1)
    with pytest.raises(MirtError):
        ds.render(cam, default_config(), 10, 10, 0)


def test_mixed_point_and_area_lights(tmp_path):
    """Shadow items of one wave with different numbers of strata (a point light: 1, area lights with 16,
    9 and 4 samples): the wave numbers them with a prefix sum and finds a stratum's item by search, a path
    no bundled scene takes (their lights are all of one kind).  Test6 with its lights edited."""
    from raytracer_amd import DeviceScene, HostScene
    text = open(os.path.join(FIXTURES, "ASCII", "Test6.txt")).read().split("\n")
    kinds = [i for i, l in enumerate(text) if l.startswith("light_type AREA")]
    samples = [i for i, l in enumerate(text) if l.startswith("samples ")]
    assert len(kinds) == 5 and len(samples) == 5
    text[kinds[1]] = "light_type POINT"
    for i, n in zip(samples, (16, 1, 9, 4, 16)):
        text[i] = f"samples {n}"
    (tmp_path / "ASCII").mkdir()
    path = tmp_path / "ASCII" / "Mixed.txt"
    path.write_text("\n".join(text))
    hs = HostScene(str(path), os.path.join(FIXTURES, ""))
    ds = DeviceScene(hs, 0)
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=4)  # shadow_samples left at 0: every light keeps its own count
    for traversal in ("packet", "lane"):
        ds.set_traversal(traversal)
        ref, ost = O.render(hs.desc, cam, cfg, 160, 90, 4, O.RNG_KEYED)
        got, st = ds.render(cam, cfg, 160, 90, 4)
        p, mae = O.psnr(got, ref), float(np.abs(got - ref).mean())
        print(f"mixed lights, {traversal}: PSNR {p:.1f} dB, MAE {mae:.3e}, segments {st['shadow_segments']} / {ost['shadow_segments']}")
        # (fewer segments than the oracle's: lights behind the surface are not traced, and the packet walk of a
        # scene without partly transparent primitives never chains through the clear ones)
        least = 0.6 if traversal == "packet" and ds.info()["shadow_any_hit"] else 0.9
        assert st["shadow_segments"] <= ost["shadow_segments"] and st["shadow_segments"] >= least * ost["shadow_segments"]
        assert p >= 60.0 and mae <= 1e-6


@pytest.mark.parametrize("name", ["Test1", "Test5"])
def test_caller_supplied_tree_with_large_leaves(name, host_scenes):
    """The C ABI takes any depth-first tree, not only the two-per-leaf one the host builder makes: the
    same scene with a three-node tree whose two leaves hold half of the primitives each (the wide walk
    keeps such leaves as index ranges, the others loop over them) must give the hits of the full tree
    -- the closest hit does not depend on the tree, ties aside -- and the oracle's on the same tree."""
    import ctypes as C

    from raytracer_amd import DeviceScene
    from raytracer_amd import _abi as A
    hs = host_scenes(name)
    d = hs.desc
    root = d.nodes[0]
    assert root.left == 1 and root.right > 1
    left, right = d.nodes[root.left], d.nodes[root.right]

    def first_leaf(i):
        while d.nodes[i].left >= 0:
            i = d.nodes[i].left
        return d.nodes[i].first

    k = first_leaf(root.right)  # leaf_prims[0:k] is the left subtree's, the rest the right one's
    nodes = (A.Node * 3)()
    for j, (src, lo, cnt) in enumerate(((root, 0, 0), (left, 0, k), (right, k, d.n_leaf_prims - k))):
        nodes[j].bmin, nodes[j].bmax = src.bmin, src.bmax
        nodes[j].left, nodes[j].right, nodes[j].first, nodes[j].count = (1, 2, 0, 0) if j == 0 else (-1, -1, lo, cnt)
    flat = type(d).from_buffer_copy(d)  # same arrays, another tree
    flat.nodes = C.cast(nodes, C.POINTER(A.Node))
    flat.n_nodes = 3
    cam, _ = hs.camera(0)
    ref = O.trace_ids(flat, cam, W, H)
    full = O.trace_ids(d, cam, W, H)
    assert float((ref["prim_id"] == full["prim_id"]).mean()) >= 0.999
    ds = DeviceScene(flat, 0)
    for traversal in ("packet", "lane", "ordered", "wide"):
        ds.set_traversal(traversal)
        got = ds.trace_ids(cam, W, H)
        same = got["prim_id"] == ref["prim_id"]
        print(f"{name}, two big leaves ({k} + {d.n_leaf_prims - k}), {traversal}: agreement {float(same.mean()):.6f}")
        assert float(same.mean()) >= 0.9995
        hit = same & (ref["prim_id"] >= 0)
        assert np.allclose(got["t"][hit], ref["t"][hit], rtol=2e-5, atol=1e-9)
    frame, st = ds.render(cam, default_config(max_ray_depth=3), W, H, 2)
    want, _ = O.render(flat, cam, default_config(max_ray_depth=3), W, H, 2, O.RNG_KEYED)
    assert O.psnr(frame, want) >= 60.0
    ds.close()


def test_more_lights_than_the_light_table_holds(tmp_path):
    """The shadow kernels keep the first 32 lights' positions and strata in LDS and read the others from
    memory.  Test1 with 40 point lights (and, second, with every fourth one an area light of 4 samples):
    lights on both sides of the limit, in both forms of the shadow kernel."""
    import random
    from raytracer_amd import DeviceScene, HostScene
    text = open(os.path.join(FIXTURES, "ASCII", "Test1.txt")).read().split("\n")
    start = text.index("LIGHTS 2")
    end = next(i for i in range(start + 1, len(text)) if text[i].startswith("SPHERES"))
    rnd = random.Random(7)
    (tmp_path / "ASCII").mkdir()
    for variant in ("point", "mixed"):
        lights = ["LIGHTS 40"]
        for i in range(40):
            area = variant == "mixed" and i % 4 == 3
            lights += [f"name L{i}", f"location {rnd.uniform(-1, 2):.6f} {rnd.uniform(-1, 2):.6f} {rnd.uniform(1.2, 2.5):.6f}",
                       "intensity 0.6", f"color {rnd.uniform(0.3, 1):.4f} {rnd.uniform(0.3, 1):.4f} {rnd.uniform(0.3, 1):.4f}"]
            if area:
                lights += ["light_type AREA", "area_shape RECTANGLE", "area_size 0.4 0.3", "normal 0.0 0.0 -1.0", "samples 4",
                           "cast_shadows 1", "shadow_soft_size 0.0"]
            else:
                lights += ["light_type POINT", "cast_shadows 1", "shadow_soft_size 0.0"]
        path = tmp_path / "ASCII" / f"Lights40_{variant}.txt"
        path.write_text("\n".join(text[:start] + lights + text[end:]))
        hs = HostScene(str(path), os.path.join(FIXTURES, ""))
        assert hs.desc.n_lights == 40
        ds = DeviceScene(hs, 0)
        cam, _ = hs.camera(0)
        cfg = default_config(max_ray_depth=3)
        ref, ost = O.render(hs.desc, cam, cfg, 128, 96, 2, O.RNG_KEYED)
        got, st = ds.render(cam, cfg, 128, 96, 2)
        p, mae = O.psnr(got, ref), float(np.abs(got - ref).mean())
        print(f"40 lights ({variant}): PSNR {p:.1f} dB, MAE {mae:.3e}, segments {st['shadow_segments']} / {ost['shadow_segments']}")
        assert st["shadow_segments"] > 0 and st["shadow_segments"] <= ost["shadow_segments"]
        assert p >= 60.0 and mae <= 1e-6


def crop_vs_oracle(ds, hs, cam, cfg, width, height, spp, windows, frame=None, min_psnr=60.0, max_mae=1e-6):
    """Renders the full frame on the GPU and compares windows of it, pixel by pixel, with the oracle's
    render of the same windows (same pixel keys, so a window is a crop of the oracle's full frame)."""
    if frame is None:
        frame, st = ds.render(cam, cfg, width, height, spp)
    for (x0, y0, w, h) in windows:
        ref, _ = O.render_window(hs.desc, cam, cfg, width, height, spp, x0, y0, w, h)
        got = frame[y0:y0 + h, x0:x0 + w]
        p, mae = O.psnr(got, ref), float(np.abs(got - ref).mean())
        print(f"{width}x{height}x{spp} window {x0},{y0} {w}x{h}: PSNR {p:.1f} dB, MAE {mae:.3e}")
        assert p >= min_psnr and mae <= max_mae
    return frame


def test_full_size_properties(host_scenes, device_scenes):
    """BASELINE configs 1 and 2 (Test5 1920x1080 16 spp, depth 5; depth 8 with glossy 8 and soft shadows 16)
    at full size: ray accounting, repeatability, and per-pixel agreement of cropped windows with the oracle."""
    hs, ds = host_scenes("Test5"), device_scenes("Test5")
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=5)
    a, sa = ds.render(cam, cfg, 1920, 1080, 16)
    b, sb = ds.render(cam, cfg, 1920, 1080, 16)
    assert sa["primary_rays"] == 1920 * 1080 * 16
    for k in ("secondary_rays", "shadow_segments", "node_tests", "prim_tests"):
        assert sa[k] == sb[k]  # the ray tree is deterministic; only float summation order varies
    assert np.allclose(a, b, rtol=0, atol=1e-12)
    assert np.isfinite(a).all() and a.min() >= 0.0
    per_sample = (sa["primary_rays"] + sa["secondary_rays"]) / sa["primary_rays"]
    assert 1.10 <= per_sample <= 1.18  # the reference measures 1.139 rays per sample on this config
    # windows: the glass cylinder and the torus (centre), the moving sphere's blur, a frame corner
    crop_vs_oracle(ds, hs, cam, cfg, 1920, 1080, 16, [(928, 520, 64, 36), (600, 300, 64, 36), (1856, 1044, 64, 36)], frame=a)
    cfg3 = default_config(max_ray_depth=8, glossy_samples=8, shadow_samples=16)
    crop_vs_oracle(ds, hs, cam, cfg3, 1920, 1080, 16, [(928, 520, 64, 36), (1100, 640, 48, 27)])


def test_cfg5_scene_at_full_size(host_scenes, device_scenes):
    """The 64-spp configuration (Test7 + glass sphere, 1920x1080, depth of field 0.95 / 16.12, soft
    shadows 16, glossy 4, max-depth 8, motion blur on) at full size: windows of the frame against the oracle."""
    hs, ds = host_scenes("Test7_glass"), device_scenes("Test7_glass")
    cam, _ = hs.camera(0, dof=(0.95, 16.12))
    cfg = default_config(max_ray_depth=8, glossy_samples=4, shadow_samples=16)
    # the glass sphere, and a patch of floor under the area lights
    crop_vs_oracle(ds, hs, cam, cfg, 1920, 1080, 64, [(948, 560, 24, 14), (400, 800, 16, 9)])


MILLION = ("synthetic1m", 1000000, 0x5EED)


@pytest.fixture(scope="module")
def million(host_scenes):
    """The generated cloud of 10^6 primitives (BASELINE's deep-BVH configuration) on the host and on the GPU."""
    hs = HostScene.synthetic(MILLION[1], MILLION[2])
    return hs, DeviceScene(hs, 0)


def test_million_primitive_cloud_vs_oracle(million):
    """Closest hits and a stochastic frame of the 10^6-primitive cloud at a resolution the oracle
    finishes in seconds (the oracle reproduces the reference's own hits and colours of this very cloud bit for
    bit: tests/golden/hits_synthetic1m.npz, tests/test_oracle_golden.py)."""
    hs, ds = million
    cam, _ = hs.camera(0)
    w, h = 192, 108
    ref = O.trace_ids(hs.desc, cam, w, h)
    for walk in ("lane", "ordered", "wide", "auto"):
        ds.set_traversal(walk)
        got = ds.trace_ids(cam, w, h)
        same = got["prim_id"] == ref["prim_id"]
        hit = same & (ref["prim_id"] >= 0)
        print(f"cloud 1e6 {walk}: prim-id agreement {same.mean():.6f}, hit fraction {(ref['prim_id'] >= 0).mean():.3f}")
        assert same.mean() >= 0.999
        assert np.allclose(got["t"][hit], ref["t"][hit], rtol=2e-5, atol=0)
    cfg = default_config(max_ray_depth=5)
    want, ost = O.render(hs.desc, cam, cfg, 96, 54, 2)
    got, st = ds.render(cam, cfg, 96, 54, 2)
    p, mae = O.psnr(got, want), float(np.abs(got - want).mean())
    print(f"cloud 1e6 render: PSNR {p:.1f} dB, MAE {mae:.3e}, rays/sample {(st['primary_rays'] + st['secondary_rays']) / st['primary_rays']:.3f}")
    assert st["primary_rays"] == ost["primary_rays"]
    assert abs(st["secondary_rays"] - ost["secondary_rays"]) <= max(2, 2e-3 * ost["secondary_rays"])
    # Tori a few hundredths across seen from hundreds of units away: the reference solves the quartic
    # in x87 long double, the device re-centres it and solves in double, and both stop the Newton polish
    # at a step of 1e-6 (Torus.cpp:220-254), so hit points on such tori differ by ~1e-6 to 1e-3 along
    # the ray (measured: 36 of 2652 torus pixels shade visibly differently, by 1e-9 to 2e-4; a grazing
    # shadow ray can flip).  Stated bound for this scene: >= 60 dB, mean absolute error <= 2e-5, at most
    # 1 % of the pixels off by more than 1e-6, and the others equal to rounding.
    off = np.abs(got - want).max(axis=2) > 1e-6
    print(f"cloud 1e6 render: {int(off.sum())} of {off.size} pixels differ, MAE of the others "
          f"{float(np.abs(got - want)[~off].mean()):.3e}")
    assert p >= 60.0 and mae <= 2e-5 and off.mean() <= 1e-2
    assert float(np.abs(got - want)[~off].mean()) <= 1e-8


def test_million_primitive_cloud_at_4k(million):
    """BASELINE's deep-BVH configuration at its full size (3840x2160, 16 spp, max-depth 5): windows
    against the oracle, and the reference-order walk against the nearer-child-first walk."""
    hs, ds = million
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=5)
    ds.set_traversal("auto")
    a = crop_vs_oracle(ds, hs, cam, cfg, 3840, 2160, 16, [(1904, 1072, 32, 18), (1000, 600, 32, 18)], max_mae=5e-5)  # far tori: see the test above
    ds.set_traversal("ordered")
    b, _ = ds.render(cam, cfg, 3840, 2160, 16)
    ds.set_traversal("auto")
    # exact ties aside, the closest hit does not depend on the visiting order
    differ = float((np.abs(a - b).max(axis=2) > 1e-9).mean())
    print(f"4K: pixels that differ between the two walks: {differ:.2e}")
    assert differ <= 1e-4


def test_exact_ties_under_the_large_scene_walk(host_scenes, device_scenes):
    """Above 512 primitives the engine picks the wide walk by itself, which visits nearer children first.  Where
    two surfaces lie at one distance -- the glass cylinder standing on the ground plane of Test4 -- the reference
    gives the hit to whichever it tests later and accepts (Cylinder.cpp:40, Plane.cpp:32, Bvh.cpp:312-332), so a
    walk in another order may give it to the other: 37 of 5184 pixels of this scene (Test4 with 600 spheres added,
    two thirds of them resting on the plane), 29.8 dB, before the wide walk learnt to notice such rays and walk
    them again in the reference's order.  Now: the reference's own hits and colours, from either walk; the
    nearer-first binary walk, which has no such guard, shows what the guard is worth."""
    name = "Test4_ties"
    hs = host_scenes(name)
    assert hs.desc.n_prims > 512
    cam, _ = hs.camera(0)
    gold = np.load(os.path.join(GOLDEN, f"hits_{name}.npz"))
    cfg = default_config(max_ray_depth=5, enable_motion_blur=0, sampling=1, enable_gamma_correction=0)
    want = gold["trace_rgb_depth5"].reshape(H, W, 3)
    result = {}
    for walk in ("lane", "auto", "ordered"):
        ds = device_scenes(name, walk)
        got = ds.trace_ids(cam, W, H)
        frame, _ = ds.render(cam, cfg, W, H, 1)
        ids_off = int((got["prim_id"] != gold["prim_id"]).sum())
        px_off = int((np.abs(frame - want).max(axis=2) > 1e-9).sum())
        result[walk] = (ids_off, px_off, O.psnr(frame, want), frame)
        print(f"{name} {walk} (walk {ds.info()['traversal']}): {ids_off} of {W * H} primary hits and {px_off} pixels of the "
              f"depth-5 frame differ from the reference's, PSNR {result[walk][2]:.1f} dB")
    assert device_scenes(name, "auto").info()["traversal"] == 4  # the wide walk
    for walk in ("lane", "auto"):  # (the 35 pixels: the last digits of sums taken in another order)
        assert result[walk][0] == 0 and result[walk][2] >= 150.0
    assert np.abs(result["auto"][3] - result["lane"][3]).max() <= 1e-12
    assert result["ordered"][1] > result["auto"][1]  # the scene does hold ties that an unguarded walk gets wrong


def test_cli_renders_the_plumbing_configuration(host_scenes, device_scenes, tmp_path):
    """The drop-in CLI end to end on the reference's own smallest configuration (Test1, 640x480, 1 spp,
    max-depth 3): same flags, P3 output, and the same image as the engine called directly; --binary-ppm
    writes the same pixels as P6."""
    import subprocess
    from conftest import FIXTURES, ROOT
    from raytracer_amd import read_ppm
    cli = os.path.join(ROOT, "raytracer_amd", "bin", "raytracer")
    if not os.path.exists(cli):
        pytest.skip("CLI not built")
    args = ["--scene", "ASCII/Test1.txt", "--resolution", "640", "480", "--samples", "1", "--max-depth", "3"]
    p3, p6 = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    r = subprocess.run([cli, *args, "--output", p3], cwd=FIXTURES, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Render time" in r.stdout + r.stderr
    r = subprocess.run([cli, *args, "--output", p6, "--binary-ppm"], cwd=FIXTURES, capture_output=True, timeout=120)
    assert r.returncode == 0
    with open(p3, "rb") as f:
        assert f.read(2) == b"P3"
    with open(p6, "rb") as f:
        assert f.read(2) == b"P6"
    a, _ = read_ppm(p3)
    b, _ = read_ppm(p6)
    assert np.array_equal(a, b)
    hs, ds = host_scenes("Test1"), device_scenes("Test1")
    cam, _ = hs.camera(0)
    want, _ = ds.render(cam, default_config(max_ray_depth=3), 640, 480, 1)
    want8 = np.floor(np.clip(want, 0.0, 1.0) * 255.0 + 0.5) / 255.0
    # fp64 atomics reach a pixel in a different order from run to run: a value on a rounding boundary may flip
    assert (np.abs(a - want8) > 1e-9).mean() < 1e-4
