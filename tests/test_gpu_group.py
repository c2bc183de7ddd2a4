"""Several GPUs behind the C ABI (mirt_group_*): the frame a group renders is the frame one GPU renders.
On a box with one GPU the group's members all sit on device 0: that exercises everything but RCCL's
transfers between devices (RCCL wants one rank per device; the group then moves the buffers with
hipMemcpyPeerAsync and says so), and a one-member group goes through RCCL's calls with itself as peer."""
import subprocess

import numpy as np
import pytest

from conftest import FIXTURES, ROOT
from raytracer_amd import DeviceGroup, default_config, read_ppm

pytestmark = pytest.mark.gpu

COUNTERS = ("primary_rays", "secondary_rays", "shadow_segments", "node_tests", "prim_tests", "shadow_node_tests",
            "shadow_prim_tests")


@pytest.mark.parametrize("members", [1, 3, 8])
def test_group_frame_equals_single_gpu_frame(members, host_scenes, device_scenes):
    hs, ds = host_scenes("Test5"), device_scenes("Test5")
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=5, glossy_samples=2)
    w, h, spp = 200, 117, 2  # not a multiple of the tile size in either direction
    want, st = ds.render(cam, cfg, w, h, spp)
    grp = DeviceGroup(hs, [0] * members)
    print(f"{members} member(s): {grp.transport}")
    assert grp.transport.startswith("RCCL") == (members == 1)
    for _ in range(2):  # buffers and communicators are reused frame after frame
        got, total, per = grp.render(cam, cfg, w, h, spp)
        assert np.abs(got - want).max() <= 1e-12
        for k in COUNTERS:
            assert total[k] == st[k], k
        assert sum(p["primary_rays"] for p in per) == w * h * spp and len(per) == members
    if members > 1:
        assert min(p["primary_rays"] for p in per) > 0  # everybody got tiles
    grp.close()


def test_group_over_two_devices_uses_rccl(host_scenes, device_scenes):
    """The gather between distinct devices (ncclSend / ncclRecv over xGMI); needs a second GPU."""
    from raytracer_amd import _abi as A
    if A.engine_lib().mirt_device_count() < 2:
        pytest.skip("one GPU: the RCCL gather between devices cannot run here (it has never run: parity unpinned on more than one device)")
    hs, ds = host_scenes("Test5"), device_scenes("Test5")
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=5, glossy_samples=2)
    want, st = ds.render(cam, cfg, 200, 117, 2)
    grp = DeviceGroup(hs, [0, 1])
    assert grp.transport.startswith("RCCL")
    for _ in range(2):
        got, total, per = grp.render(cam, cfg, 200, 117, 2)
        assert np.abs(got - want).max() <= 1e-12
        for k in COUNTERS:
            assert total[k] == st[k], k
    grp.close()


def test_group_on_a_large_scene_and_bad_arguments(host_scenes):
    from raytracer_amd import HostScene, DeviceScene, MirtError
    hs = HostScene.synthetic(20000)
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=3)
    want, _ = DeviceScene(hs, 0).render(cam, cfg, 96, 54, 1)
    grp = DeviceGroup(hs, [0, 0])
    got, _, _ = grp.render(cam, cfg, 96, 54, 1)
    assert np.abs(got - want).max() <= 1e-12
    with pytest.raises(MirtError):
        grp.render(cam, cfg, 0, 54, 1)
    with pytest.raises(MirtError):
        DeviceGroup(hs, [])
    with pytest.raises(MirtError):
        DeviceGroup(hs, [99])


def test_group_whose_members_build_their_own_trees():
    """mirt_group_create_with(MIRT_BUILDER_DEVICE): every member builds its BVH on its GPU from a description
    without a tree; the frame is the one a single scene built the same way renders."""
    from raytracer_amd import HostScene, DeviceScene, MirtError
    hs = HostScene.synthetic(20000, build_tree=False)
    assert hs.desc.n_nodes == 0
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=3)
    one = DeviceScene(hs, 0, builder="device")
    want, wst = one.render(cam, cfg, 192, 108, 2)
    grp = DeviceGroup(hs, [0, 0, 0], builder="device")
    got, st, members = grp.render(cam, cfg, 192, 108, 2)
    assert len(members) == 3 and st["primary_rays"] == wst["primary_rays"] == 192 * 108 * 2
    assert np.abs(got - want).max() <= 1e-12
    grp.close()
    with pytest.raises(MirtError):
        DeviceGroup(hs, [0, 0])  # the reference-order builder needs the description's tree


def test_group_frames_in_flight_and_device_output(host_scenes, device_scenes):
    """mirt_group_render_begin / _end: two frames in flight in the two slots, frame i read back while frame
    i + 1 renders, and the frame left on the first device (mirt_group_frame)."""
    import torch
    hs, ds = host_scenes("Test5"), device_scenes("Test5")
    cam, _ = hs.camera(0)
    w, h, spp = 256, 144, 2  # 16 tiles a row: a multiple of 2, 4 and 8 members, so the tile rows are staggered
    cfgs = [default_config(max_ray_depth=5, seed=s) for s in (1, 2, 3)]
    want = [ds.render(cam, c, w, h, spp)[0] for c in cfgs]
    for members in (2, 4, 8):
        grp = DeviceGroup(hs, [0] * members)
        grp.render_begin(cam, cfgs[0], w, h, spp, slot=0)
        grp.render_begin(cam, cfgs[1], w, h, spp, slot=1)
        with pytest.raises(Exception):
            grp.render_begin(cam, cfgs[2], w, h, spp, slot=1)  # the slot is taken
        a, st, per = grp.render_end(0)
        grp.render_begin(cam, cfgs[2], w, h, spp, slot=0)   # frame 2 starts while frame 1 is still running
        b, _, _ = grp.render_end(1)
        none, _, _ = grp.render_end(0, to_host=False)
        assert none is None
        with pytest.raises(Exception):
            grp.render_end(0)  # nothing in flight
        assert np.abs(a - want[0]).max() <= 1e-12 and np.abs(b - want[1]).max() <= 1e-12
        assert st["primary_rays"] == w * h * spp and len(per) == members
        # equal shares: every member owns the same number of tiles here
        assert len({p["primary_rays"] for p in per}) == 1
        # the third frame stayed on the device
        ptr = grp.frame_pointer(0)
        assert ptr
        buf = torch.empty(h * w * 3, dtype=torch.float64, device="cuda:0")
        from raytracer_amd import _abi as A
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(ptr), C.c_size_t(h * w * 3 * 8), 3) == 0  # device to device
        assert np.abs(buf.cpu().numpy().reshape(h, w, 3) - want[2]).max() <= 1e-12
        grp.close()


def test_group_recovers_after_a_member_fails(host_scenes, device_scenes, monkeypatch):
    """A member whose begin fails must not leave the members begun before it holding a frame: the next
    frame of the same group renders."""
    hs, ds = host_scenes("Test5"), device_scenes("Test5")
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=4)
    want, _ = ds.render(cam, cfg, 120, 80, 2)
    grp = DeviceGroup(hs, [0, 0, 0])
    from raytracer_amd import MirtError
    for member in (2, 1, 0):
        monkeypatch.setenv("MIRT_GROUP_TEST_FAIL_MEMBER", str(member))
        with pytest.raises(MirtError, match="injected"):
            grp.render(cam, cfg, 120, 80, 2)
        monkeypatch.delenv("MIRT_GROUP_TEST_FAIL_MEMBER")
        got, _, _ = grp.render(cam, cfg, 120, 80, 2)
        assert np.abs(got - want).max() <= 1e-12
    with pytest.raises(MirtError):
        grp.render(cam, cfg, 100000, 100000, 16)  # refused by the first member already
    got, _, _ = grp.render(cam, cfg, 120, 80, 2)
    assert np.abs(got - want).max() <= 1e-12
    grp.close()


def test_eight_members_at_the_scaling_configuration(host_scenes, device_scenes):
    """BASELINE's tile-split configuration (Test5 1920x1080 16 spp, soft shadows 16, glossy 8, max-depth 8)
    through an 8-member group: windows of the frame equal the single-GPU frame's, and the members' shares of
    the work are balanced (the staggered tile rows: 120 tiles a row is a multiple of 8)."""
    hs, ds = host_scenes("Test5"), device_scenes("Test5")
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=8, glossy_samples=8, shadow_samples=16)
    w, h, spp = 1920, 1080, 16
    want, st = ds.render(cam, cfg, w, h, spp)
    grp = DeviceGroup(hs, [0] * 8)
    got, total, per = grp.render(cam, cfg, w, h, spp)
    for (x0, y0, cw, ch) in [(928, 520, 64, 36), (1100, 640, 48, 27), (0, 0, 64, 36), (1856, 1044, 64, 36)]:
        assert np.abs(got[y0:y0 + ch, x0:x0 + cw] - want[y0:y0 + ch, x0:x0 + cw]).max() <= 1e-12
    assert np.abs(got - want).max() <= 1e-12
    for k in COUNTERS:
        assert total[k] == st[k], k
    work = np.array([p["primary_rays"] + p["secondary_rays"] + p["shadow_segments"] for p in per], dtype=float)
    print(f"8 members, cfg3: rays + segments per member {work.astype(int).tolist()}, max / mean {work.max() / work.mean():.4f}, "
          f"device ms per member {[round(p['device_ms'], 2) for p in per]}")
    # 16x16 tiles against a glossy torus and a glass cylinder a few tiles across: measured 1.049 with the
    # lattice rule (1.066 with plain diagonals); whole columns of tiles per member would be the alternative
    assert work.max() / work.mean() <= 1.08
    grp.close()


def test_cli_devices_flag(tmp_path):
    """`raytracer --devices 2` (both on GPU 0 here: MIRT_DEVICES_SAME=1) writes the image `--devices 1` writes."""
    import os
    cli = os.path.join(ROOT, "raytracer_amd", "bin", "raytracer")
    if not os.path.exists(cli):
        pytest.skip("CLI not built")
    args = ["--scene", "ASCII/Test4.txt", "--resolution", "160", "90", "--samples", "2", "--max-depth", "4"]
    one, two = str(tmp_path / "one.ppm"), str(tmp_path / "two.ppm")
    r = subprocess.run([cli, *args, "--output", one], cwd=FIXTURES, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, MIRT_DEVICES_SAME="1")
    r = subprocess.run([cli, *args, "--devices", "2", "--output", two], cwd=FIXTURES, capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    assert "2 GPUs" in r.stdout + r.stderr
    a, _ = read_ppm(one)
    b, _ = read_ppm(two)
    assert (np.abs(a - b) > 1e-9).mean() < 1e-4  # a value on a rounding boundary may flip with the order of the sums


def test_cli_gpu_bvh_flag(tmp_path):
    """`raytracer --gpu-bvh` renders a generated cloud (written out as an ASCII scene file) with the tree built on
    the GPU -- alone and with --devices 2 -- and gets the image of the reference's own tree up to the handful of
    pixels whose far tori the two trees test differently; a scene with planes is refused with a message."""
    import os
    from raytracer_amd import HostScene
    cli = os.path.join(ROOT, "raytracer_amd", "bin", "raytracer")
    if not os.path.exists(cli):
        pytest.skip("CLI not built")
    (tmp_path / "ASCII").mkdir()
    HostScene.synthetic(3000, seed=5).write_ascii(str(tmp_path / "ASCII" / "cloud.txt"))
    args = ["--scene", "ASCII/cloud.txt", "--resolution", "160", "90", "--samples", "2", "--max-depth", "3", "--binary-ppm"]
    out = {k: str(tmp_path / f"{k}.ppm") for k in ("host", "gpu", "gpu2")}
    env = dict(os.environ, MIRT_DEVICES_SAME="1")
    for k, extra in (("host", []), ("gpu", ["--gpu-bvh"]), ("gpu2", ["--gpu-bvh", "--devices", "2"])):
        r = subprocess.run([cli, *args, *extra, "--output", out[k]], cwd=tmp_path, capture_output=True, text=True,
                           timeout=120, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
    a, _ = read_ppm(out["host"])
    b, _ = read_ppm(out["gpu"])
    c, _ = read_ppm(out["gpu2"])
    assert (np.abs(a - b) > 1e-9).mean() < 5e-3
    assert (np.abs(b - c) > 1e-9).mean() < 1e-4
    r = subprocess.run([cli, "--scene", "ASCII/Test5.txt", "--resolution", "64", "36", "--gpu-bvh", "--output", str(tmp_path / "x.ppm")],
                       cwd=FIXTURES, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "planes" in r.stdout + r.stderr


def test_progress_feed_and_render_recipes(host_scenes, tmp_path):
    """mirt_scene_record_progress / mirt_scene_progress, the row-times CSV of the drop-in renderer
    (written, like the reference's, when Output/ exists in the working directory), and the reference
    Makefile's `test-3` settings through `make test-3`."""
    import os
    import shutil
    from raytracer_amd import DeviceScene
    hs = host_scenes("Test5")
    ds = DeviceScene(hs, 0)
    cam, _ = hs.camera(0)
    cfg = default_config(max_ray_depth=4)
    want, _ = ds.render(cam, cfg, 320, 200, 4)
    assert ds.progress().size == 0
    ds.record_progress(True)
    got, _ = ds.render(cam, cfg, 320, 200, 4)
    assert np.abs(got - want).max() <= 1e-12  # recording changes nothing in the frame
    ms = ds.progress()
    assert ms.size == 13 and (ms > 0).all() and ms.max() < 1000.0  # 200 rows = 13 bands of 16
    ds.record_progress(False)
    assert ds.progress().size == 0
    # the CSV, through the command line, in a scratch copy of the fixture tree
    cli = os.path.join(ROOT, "raytracer_amd", "bin", "raytracer")
    if not os.path.exists(cli):
        pytest.skip("CLI not built")
    work = tmp_path / "work"
    os.makedirs(work / "Output")
    os.symlink(os.path.join(FIXTURES, "ASCII"), work / "ASCII")
    os.symlink(os.path.join(FIXTURES, "Textures"), work / "Textures")
    r = subprocess.run([cli, "--scene", "ASCII/Test1.txt", "--resolution", "160", "100", "--samples", "2", "--output",
                        "Output/t1.ppm"], cwd=work, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = open(work / "Output" / "render_row_times.csv").read().split("\n")
    assert lines[0] == "Row,Time(ms)" and len(lines) == 102 and lines[1].startswith("0,") and lines[100].startswith("99,")
    assert all(float(l.split(",")[1]) > 0 for l in lines[1:101])
    # one of the reference's render recipes
    if shutil.which("make"):
        out = tmp_path / "recipes"
        r = subprocess.run(["make", "-s", "test-3", f"OUT_DIR={out}"], cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        img, _ = read_ppm(str(out / "rendered_bvh_Test3.ppm"))
        assert img.shape == (400, 720, 3) and 0.01 < img.mean() < 0.99
