"""Python face of the drop-in path: parse a scene with the host library, hand it to the HIP engine
through the C ABI (include/mirt.h), get frames back as numpy arrays.

There is no CPU rendering path here: every render call goes to libmirt.so and fails loudly when
the library or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as A


class MirtError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        msg = A.engine_lib().mirt_last_error()
        raise MirtError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def default_config(**overrides):
    """RenderConfig defaults of the reference (Code/Core/Config.h), as a mirt_config."""
    cfg = A.Config()
    A.engine_lib().mirt_config_defaults(C.byref(cfg))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(f"mirt_config has no field {k!r}")
        setattr(cfg, k, v)
    return cfg


class HostScene:
    """A parsed + flattened scene (load_scene + BVH build + texture preload, all on the host)."""

    def __init__(self, path, texture_root=None, build_tree=True):
        lib = A.host_lib()
        if texture_root is None:
            # scenes live in <root>/ASCII/, textures in <root>/Textures/ (the reference's layout)
            texture_root = os.path.dirname(os.path.dirname(os.path.abspath(path)))
        self._lib = lib
        self._h = lib.mirt_host_scene_load_opt(os.fsencode(path), os.fsencode(texture_root), 1 if build_tree else 0)
        if not self._h:
            raise MirtError(f"mirt_host_scene_load({path}): {lib.mirt_host_last_error().decode()}")
        self.path = path
        self.desc = A.SceneDesc()
        lib.mirt_host_scene_desc(self._h, C.byref(self.desc))

    @classmethod
    def synthetic(cls, n_prims, seed=0x5EED, build_tree=True):
        """The synthetic cloud of `n_prims` small spheres, tori, cylinders and cones
        (raytracer_amd/host/synthetic.h), flattened like a parsed scene.  build_tree=False skips the
        host BVH build (for DeviceScene(..., builder="device")); flatten_ms is the wall clock of the
        flattening, tree included when it was built."""
        self = cls.__new__(cls)
        self._lib = A.host_lib()
        ms = C.c_double()
        self._h = self._lib.mirt_host_scene_synthetic_opt(int(n_prims), int(seed), 1 if build_tree else 0, C.byref(ms))
        self.flatten_ms = ms.value
        if not self._h:
            raise MirtError(f"mirt_host_scene_synthetic: {self._lib.mirt_host_last_error().decode()}")
        self.path = f"synthetic:{n_prims}:{seed}"
        self.desc = A.SceneDesc()
        self._lib.mirt_host_scene_desc(self._h, C.byref(self.desc))
        return self

    def write_ascii(self, path):
        """Writes the scene in the ASCII grammar the loader reads (sphere/torus/cylinder/cone scenes)."""
        if self._lib.mirt_host_scene_write_ascii(self._h, os.fsencode(path)) != 0:
            raise OSError(f"cannot write {path}")

    def close(self):
        if self._h:
            self._lib.mirt_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def camera_count(self):
        return self._lib.mirt_host_scene_camera_count(self._h)

    def camera(self, index=0, dof=None):
        """Camera `index`; dof=None keeps the file's setting, dof=(fstop, focus) is --depth-of-field,
        dof=False is --disable-dof (Main.cpp:160-176).  Returns (mirt_camera, (res_x, res_y))."""
        cam, rx, ry = A.Camera(), C.c_int(), C.c_int()
        if dof is None:
            mode, f, d = 0, 0.0, 0.0
        elif dof is False:
            mode, f, d = 2, 0.0, 0.0
        else:
            mode, (f, d) = 1, dof
        if self._lib.mirt_host_scene_camera(self._h, index, mode, f, d, C.byref(cam), C.byref(rx), C.byref(ry)) != 0:
            raise IndexError(f"scene has no camera {index}")
        return cam, (rx.value, ry.value)

    def bvh_stats(self):
        n, l, d = C.c_int(), C.c_int(), C.c_int()
        self._lib.mirt_host_bvh_stats(self._h, C.byref(n), C.byref(l), C.byref(d))
        return {"nodes": n.value, "leaves": l.value, "depth": d.value}

    def dump_scene(self, path):
        return self._lib.mirt_host_scene_dump(self._h, os.fsencode(path))

    def dump_bvh(self, path):
        return self._lib.mirt_host_bvh_dump(self._h, os.fsencode(path))


class DeviceScene:
    """A scene resident on one GPU (mirt_scene_create)."""

    def __init__(self, scene, device=0, builder="host", leaf_size=0):
        """builder: "host" (the tree in the description, the reference's) or "device" (built on the GPU,
        plane-free scenes, wide walk only; the description needs no tree)."""
        self._lib = A.engine_lib()
        self._h = C.c_void_p()
        desc = scene.desc if isinstance(scene, HostScene) else scene
        self._keep = scene  # the description's arrays must outlive the call only, but keep it simple
        opts = A.BuildOptions()
        opts.builder = {"host": A.MIRT_BUILDER_HOST, "device": A.MIRT_BUILDER_DEVICE}[builder]
        opts.leaf_size = leaf_size
        _check(self._lib.mirt_scene_create_with(C.byref(desc), device, C.byref(opts), C.byref(self._h)),
               "mirt_scene_create_with")
        self.device = device

    def info(self):
        """mirt_scene_get_info as a dict (builder, traversal in use, tree sizes, creation times)."""
        i = A.SceneInfo()
        _check(self._lib.mirt_scene_get_info(self._h, C.byref(i)), "mirt_scene_get_info")
        return i.as_dict()

    def update_prims(self, first, prims):
        """New records (a ctypes array of mirt_prim) for primitives first .. of a device-built scene."""
        _check(self._lib.mirt_scene_update_prims(self._h, first, len(prims), prims), "mirt_scene_update_prims")

    def refit(self):
        _check(self._lib.mirt_scene_refit(self._h), "mirt_scene_refit")

    def set_traversal(self, mode):
        """mode: "auto" (packet up to 512 primitives, wide above), "packet" (wave-coherent BVH walk),
        "lane" (one walk per ray, the reference's order), "ordered" (one walk per ray, nearer child
        first) or "wide" (4-wide nodes, nearer child first, primitive tests batched by kind).  Only
        packet and lane resolve hits at exactly equal distances in the reference's order."""
        code = {"auto": 0, "packet": 1, "lane": 2, "ordered": 3, "wide": 4}[mode]
        _check(self._lib.mirt_scene_set_traversal(self._h, code), "mirt_scene_set_traversal")

    def close(self):
        if self._h:
            self._lib.mirt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, camera, config, width, height, spp):
        """One frame: (height, width, 3) float64 array of post-gamma un-clamped values, and stats."""
        out = np.empty((height, width, 3), dtype=np.float64)
        st = A.Stats()
        _check(self._lib.mirt_render(self._h, C.byref(camera), C.byref(config), width, height, spp,
                                     out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st)), "mirt_render")
        return out, st.as_dict()

    def record_progress(self, enable=True):
        _check(self._lib.mirt_scene_record_progress(self._h, 1 if enable else 0), "mirt_scene_record_progress")

    def progress(self):
        """Milliseconds (since the last frame's first block started) at which each band of 16 pixel
        rows had its primary samples shaded; empty when recording is off."""
        n = self._lib.mirt_scene_progress(self._h, None, 0)
        out = np.zeros(max(n, 0), dtype=np.float64)
        if n > 0:
            self._lib.mirt_scene_progress(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), n)
        return out

    def render_tiles(self, camera, config, width, height, spp, tile_w, tile_h, rank, world, out_ptr, stream=None):
        """This rank's tiles into the device buffer at `out_ptr` (see mirt_render_tiles)."""
        st = A.Stats()
        _check(self._lib.mirt_render_tiles(self._h, C.byref(camera), C.byref(config), width, height, spp, tile_w,
                                           tile_h, rank, world, C.c_void_p(out_ptr), C.c_void_p(stream or 0),
                                           C.byref(st)), "mirt_render_tiles")
        return st.as_dict()

    def render_tiles_noalloc(self, camera, config, width, height, spp, tile_w, tile_h, rank, world, out_ptr, stream=None):
        """mirt_render_tiles_noalloc: the same frame out of the per-GPU static working memory."""
        st = A.Stats()
        _check(self._lib.mirt_render_tiles_noalloc(self._h, C.byref(camera), C.byref(config), width, height, spp,
                                                   tile_w, tile_h, rank, world, C.c_void_p(out_ptr),
                                                   C.c_void_p(stream or 0), C.byref(st)), "mirt_render_tiles_noalloc")
        return st.as_dict()

    def render_tiles_begin(self, camera, config, width, height, spp, tile_w, tile_h, rank, world, out_ptr, stream,
                           slot):
        """Enqueues this rank's tiles on `stream` with working-memory set `slot` (0 .. 3) and returns
        without waiting (mirt_render_tiles_begin)."""
        _check(self._lib.mirt_render_tiles_begin(self._h, C.byref(camera), C.byref(config), width, height, spp,
                                                 tile_w, tile_h, rank, world, C.c_void_p(out_ptr),
                                                 C.c_void_p(stream or 0), slot), "mirt_render_tiles_begin")

    def render_tiles_end(self, slot):
        """Waits for the frame of `slot`; returns its stats (frame_attempts > 1: it was rendered again)."""
        st = A.Stats()
        _check(self._lib.mirt_render_tiles_end(self._h, slot, C.byref(st)), "mirt_render_tiles_end")
        return st.as_dict()

    def trace_ids(self, camera, width, height):
        n = width * height
        ids = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float64)
        nrm = np.empty((n, 3), dtype=np.float64)
        uv = np.empty((n, 2), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _check(self._lib.mirt_trace_ids(self._h, C.byref(camera), width, height,
                                        ids.ctypes.data_as(C.POINTER(C.c_int32)), t.ctypes.data_as(dp),
                                        nrm.ctypes.data_as(dp), uv.ctypes.data_as(dp)), "mirt_trace_ids")
        return {"prim_id": ids, "t": t, "normal": nrm, "uv": uv}

    def trace_rays(self, origin, direction, time=None, t_min=1e-5, t_max=None):
        """Closest hit of arbitrary rays (mirt_trace_rays)."""
        rc, out = _trace_rays(self._lib.mirt_trace_rays, self._h, origin, direction, time, t_min, t_max)
        _check(rc, "mirt_trace_rays")
        return out


class DeviceGroup:
    """A scene replicated on several GPUs of this process (mirt_group_create); frames are split into
    16x16 tiles dealt round-robin and gathered on the first device."""

    def __init__(self, scene, devices, builder="host", leaf_size=0):
        """builder / leaf_size: as for DeviceScene, for every member's replica (mirt_group_create_with)."""
        self._lib = A.engine_lib()
        self._h = C.c_void_p()
        desc = scene.desc if isinstance(scene, HostScene) else scene
        self._keep = scene
        devs = (C.c_int * len(devices))(*devices)
        opts = A.BuildOptions()
        opts.builder = {"host": A.MIRT_BUILDER_HOST, "device": A.MIRT_BUILDER_DEVICE}[builder]
        opts.leaf_size = leaf_size
        _check(self._lib.mirt_group_create_with(C.byref(desc), devs, len(devices), C.byref(opts), C.byref(self._h)),
               "mirt_group_create_with")
        self.devices = list(devices)

    @property
    def transport(self):
        return self._lib.mirt_group_transport(self._h).decode()

    def set_traversal(self, mode):
        code = {"auto": 0, "packet": 1, "lane": 2, "ordered": 3, "wide": 4}[mode]
        _check(self._lib.mirt_group_set_traversal(self._h, code), "mirt_group_set_traversal")

    def close(self):
        if self._h:
            self._lib.mirt_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, camera, config, width, height, spp):
        """One frame: (height, width, 3) float64 array, summed stats, per-member stats."""
        out = np.empty((height, width, 3), dtype=np.float64)
        st = A.Stats()
        per = (A.Stats * len(self.devices))()
        _check(self._lib.mirt_group_render(self._h, C.byref(camera), C.byref(config), width, height, spp,
                                           out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st), per),
               "mirt_group_render")
        return out, st.as_dict(), [p.as_dict() for p in per]


    # the two halves of render(): several frames in flight (mirt_group_render_begin / _end)
    def render_begin(self, camera, config, width, height, spp, slot=0):
        _check(self._lib.mirt_group_render_begin(self._h, C.byref(camera), C.byref(config), width, height, spp, slot),
               "mirt_group_render_begin")
        self._shape = getattr(self, "_shape", {})
        self._shape[slot] = (height, width)

    def render_end(self, slot=0, to_host=True):
        """Waits for the slot's frame; returns (frame or None, summed stats, per-member stats).  With
        to_host=False the frame stays on the first device: frame_pointer(slot)."""
        h, w = self._shape[slot]
        out = np.empty((h, w, 3), dtype=np.float64) if to_host else None
        st = A.Stats()
        per = (A.Stats * len(self.devices))()
        _check(self._lib.mirt_group_render_end(self._h, slot, _dptr(out), C.byref(st), per), "mirt_group_render_end")
        return out, st.as_dict(), [p.as_dict() for p in per]

    def frame_pointer(self, slot=0):
        return self._lib.mirt_group_frame(self._h, slot)


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _trace_rays(fn, handle, origin, direction, time, t_min, t_max):
    origin = np.ascontiguousarray(origin, dtype=np.float64).reshape(-1, 3)
    direction = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
    n = origin.shape[0]
    time = None if time is None else np.ascontiguousarray(time, dtype=np.float64)
    t_max = None if t_max is None else np.ascontiguousarray(t_max, dtype=np.float64)
    ids = np.empty(n, dtype=np.int32)
    t = np.empty(n, dtype=np.float64)
    nrm = np.empty((n, 3), dtype=np.float64)
    rc = fn(handle, n, _dptr(origin), _dptr(direction), _dptr(time), float(t_min), _dptr(t_max),
            ids.ctypes.data_as(C.POINTER(C.c_int32)), _dptr(t), _dptr(nrm))
    return rc, {"prim_id": ids, "t": t, "normal": nrm}


def tile_buffer_pixels(width, height, tile_w, tile_h, rank, world):
    return int(A.engine_lib().mirt_tile_buffer_pixels(width, height, tile_w, tile_h, rank, world))


def tile_owner(tx, ty, tiles_x, world):
    """The rank that renders tile (tx, ty): (ty * tiles_x + tx) % world, or (tx + k * ty) % world with
    k = round(sqrt(world)) when tiles_x is a multiple of world (mirt_tile_owner's rule, restated so that it
    also runs without the library)."""
    if world > 1 and tiles_x % world == 0:
        return (tx + tile_lattice_step(world) * ty) % world
    return (ty * tiles_x + tx) % world


def tile_lattice_step(world):
    k = 1
    while (2 * k + 1) ** 2 <= 4 * world:
        k += 1
    return k


def owned_tiles(width, height, tile_w, tile_h, rank, world):
    """Tile indices (row-major) rendered by `rank`, in buffer order."""
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    return [t for t in range(tiles_x * tiles_y) if tile_owner(t % tiles_x, t // tiles_x, tiles_x, world) == rank]


def scatter_tiles(frame, buf, width, height, tile_w, tile_h, rank, world):
    """Copies rank's compact tile buffer `buf` ((n_tiles*tile_h*tile_w, 3)) into the (H, W, 3) frame."""
    tiles_x = (width + tile_w - 1) // tile_w
    buf = buf.reshape(-1, tile_h, tile_w, 3)
    for slot, t in enumerate(owned_tiles(width, height, tile_w, tile_h, rank, world)):
        tx, ty = t % tiles_x, t // tiles_x
        x0, y0 = tx * tile_w, ty * tile_h
        w, h = min(tile_w, width - x0), min(tile_h, height - y0)
        frame[y0:y0 + h, x0:x0 + w] = buf[slot, :h, :w]
    return frame


def write_ppm(path, rgb, binary=False):
    """Writes an (H, W, 3) float64 frame as the reference writes its output (ASCII P3 by default)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w, _ = rgb.shape
    A.host_lib().mirt_host_write_ppm(os.fsencode(path), rgb.ctypes.data_as(C.POINTER(C.c_double)), w, h,
                                     1 if binary else 0)


def read_ppm(path):
    w, h, m = C.c_int(), C.c_int(), C.c_int()
    lib = A.host_lib()
    if lib.mirt_host_read_ppm(os.fsencode(path), C.byref(w), C.byref(h), C.byref(m), None) != 0:
        raise IOError(f"cannot read PPM {path}")
    out = np.empty((h.value, w.value, 3), dtype=np.float64)
    lib.mirt_host_read_ppm(os.fsencode(path), C.byref(w), C.byref(h), C.byref(m),
                           out.ctypes.data_as(C.POINTER(C.c_double)))
    return out, m.value
