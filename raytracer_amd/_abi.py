"""ctypes mirror of include/mirt.h (the engine's C ABI) and of the host library's C entry points.

Field order and types follow the header one for one; `tests/test_abi.py` checks the sizes against
the compiled library.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIRT_LIB_DIR selects another build of the two libraries (A/B experiments); default: in-tree lib/
LIB_DIR = os.environ.get("MIRT_LIB_DIR") or os.path.join(_HERE, "lib")

MIRT_SPHERE, MIRT_CUBE, MIRT_PLANE, MIRT_TORUS, MIRT_CYLINDER, MIRT_CONE = range(6)
MIRT_SAMPLING_JITTER, MIRT_SAMPLING_CENTRE = 0, 1
MIRT_TONEMAP_NONE, MIRT_TONEMAP_REINHARD, MIRT_TONEMAP_EXPOSURE = 0, 1, 2

d3 = C.c_double * 3
d16 = C.c_double * 16


class Node(C.Structure):
    _fields_ = [("bmin", d3), ("bmax", d3), ("left", C.c_int32), ("right", C.c_int32),
                ("first", C.c_int32), ("count", C.c_int32)]


class Prim(C.Structure):
    _fields_ = [("type", C.c_int32), ("visible", C.c_int32), ("has_motion", C.c_int32), ("material", C.c_int32),
                ("o2w", d16), ("w2o", d16), ("m0", d16), ("m1", d16), ("param", C.c_double * 4),
                ("first_point", C.c_int32), ("n_points", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("diffuse", d3), ("specular", d3), ("ambient", d3), ("emission", d3),
                ("shininess", C.c_double), ("glossiness", C.c_double), ("reflectivity", C.c_double),
                ("transparency", C.c_double), ("refractive_index", C.c_double), ("emission_strength", C.c_double),
                ("bump_strength", C.c_double), ("has_texture", C.c_int32), ("texture", C.c_int32),
                ("normal_map", C.c_int32), ("bump_map", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("location", d3), ("color", d3), ("normal", d3), ("intensity", C.c_double),
                ("area_size", C.c_double * 2), ("type", C.c_int32), ("area_shape", C.c_int32),
                ("samples", C.c_int32), ("pad_", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_value", C.c_int32), ("pad_", C.c_int32),
                ("rgb", C.POINTER(C.c_uint16))]


class SceneDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_leaf_prims", C.c_int32), ("n_prims", C.c_int32),
                ("n_plane_points", C.c_int32), ("n_materials", C.c_int32), ("n_lights", C.c_int32),
                ("n_textures", C.c_int32), ("pad_", C.c_int32),
                ("nodes", C.POINTER(Node)), ("leaf_prims", C.POINTER(C.c_int32)), ("prims", C.POINTER(Prim)),
                ("plane_points", C.POINTER(C.c_double)), ("materials", C.POINTER(Material)),
                ("lights", C.POINTER(Light)), ("textures", C.POINTER(Texture)),
                ("background_color", d3), ("background_strength", C.c_double)]


class Camera(C.Structure):
    _fields_ = [("location", d3), ("gaze", d3), ("up", d3), ("focal_length", C.c_double),
                ("sensor_height", C.c_double), ("focus_distance", C.c_double), ("aperture_fstop", C.c_double),
                ("dof_enabled", C.c_int32), ("pad_", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("max_ray_depth", C.c_int32), ("shadow_samples", C.c_int32), ("glossy_samples", C.c_int32),
                ("enable_motion_blur", C.c_int32), ("enable_textures", C.c_int32),
                ("use_adaptive_epsilon", C.c_int32), ("tone_mapping_mode", C.c_int32),
                ("enable_gamma_correction", C.c_int32), ("light_intensity_factor", C.c_double),
                ("ambient_factor", C.c_double), ("ray_offset_epsilon", C.c_double),
                ("adaptive_epsilon_scale", C.c_double), ("pure_glass_threshold", C.c_double),
                ("exposure", C.c_double), ("gamma", C.c_double), ("seed", C.c_uint64),
                ("sampling", C.c_int32), ("kernel_timing", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("secondary_rays", C.c_uint64), ("shadow_segments", C.c_uint64),
                ("node_tests", C.c_uint64), ("prim_tests", C.c_uint64), ("shadow_node_tests", C.c_uint64),
                ("shadow_prim_tests", C.c_uint64), ("launches", C.c_uint64),
                ("device_ms", C.c_double), ("trace_kernel_ms", C.c_double), ("shadow_kernel_ms", C.c_double),
                ("trace_kernel_launches", C.c_uint64), ("shadow_kernel_launches", C.c_uint64),
                ("frame_attempts", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BuildOptions(C.Structure):
    _fields_ = [("builder", C.c_int32), ("leaf_size", C.c_int32), ("reserved", C.c_int32 * 6)]


class SceneInfo(C.Structure):
    _fields_ = [("builder", C.c_int32), ("traversal", C.c_int32), ("n_prims", C.c_int32),
                ("n_wide_nodes", C.c_int32), ("n_leaves", C.c_int32), ("wide_depth", C.c_int32),
                ("bvh_depth", C.c_int32), ("shadow_any_hit", C.c_int32), ("create_ms", C.c_double),
                ("tree_ms", C.c_double), ("refit_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


MIRT_BUILDER_HOST, MIRT_BUILDER_DEVICE = 0, 1

ENGINE_SYMBOLS = [
    "mirt_tile_owner", "mirt_group_render_begin", "mirt_group_render_end", "mirt_group_frame", "mirt_scene_create_with", "mirt_scene_get_info", "mirt_scene_update_prims", "mirt_scene_refit",
    "mirt_device_count", "mirt_config_defaults", "mirt_scene_create", "mirt_scene_destroy", "mirt_render",
    "mirt_render_tiles", "mirt_tile_buffer_pixels", "mirt_trace_ids", "mirt_trace_rays", "mirt_scene_set_traversal", "mirt_render_tiles_begin", "mirt_render_tiles_end",
    "mirt_last_error", "mirt_abi_version", "mirt_render_tiles_noalloc", "mirt_noalloc_scratch_bytes",
    "mirt_group_create", "mirt_group_create_with", "mirt_group_destroy", "mirt_group_size", "mirt_group_transport", "mirt_group_set_traversal",
    "mirt_group_render", "mirt_scene_record_progress", "mirt_scene_progress",
]


def _declare_engine(lib):
    dp = C.POINTER(C.c_double)
    lib.mirt_device_count.restype = C.c_int
    lib.mirt_config_defaults.argtypes = [C.POINTER(Config)]
    lib.mirt_config_defaults.restype = None
    lib.mirt_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
    lib.mirt_scene_create_with.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(BuildOptions), C.POINTER(C.c_void_p)]
    lib.mirt_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
    lib.mirt_scene_update_prims.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Prim)]
    lib.mirt_scene_refit.argtypes = [C.c_void_p]
    lib.mirt_scene_destroy.argtypes = [C.c_void_p]
    lib.mirt_scene_destroy.restype = None
    lib.mirt_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Config), C.c_int, C.c_int, C.c_int,
                                dp, C.POINTER(Stats)]
    lib.mirt_render_tiles.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Config), C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    lib.mirt_render_tiles_begin.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Config), C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.mirt_render_tiles_end.argtypes = [C.c_void_p, C.c_int, C.POINTER(Stats)]
    lib.mirt_render_tiles_noalloc.argtypes = lib.mirt_render_tiles.argtypes
    lib.mirt_noalloc_scratch_bytes.restype = C.c_uint64
    lib.mirt_scene_record_progress.argtypes = [C.c_void_p, C.c_int]
    lib.mirt_scene_progress.argtypes = [C.c_void_p, dp, C.c_int]
    lib.mirt_group_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    lib.mirt_group_create_with.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(BuildOptions),
                                           C.POINTER(C.c_void_p)]
    lib.mirt_group_destroy.argtypes = [C.c_void_p]
    lib.mirt_group_destroy.restype = None
    lib.mirt_group_size.argtypes = [C.c_void_p]
    lib.mirt_group_transport.argtypes = [C.c_void_p]
    lib.mirt_group_transport.restype = C.c_char_p
    lib.mirt_group_set_traversal.argtypes = [C.c_void_p, C.c_int]
    lib.mirt_group_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Config), C.c_int, C.c_int, C.c_int, dp,
                                      C.POINTER(Stats), C.POINTER(Stats)]
    lib.mirt_group_render_begin.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Config), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.mirt_group_render_end.argtypes = [C.c_void_p, C.c_int, dp, C.POINTER(Stats), C.POINTER(Stats)]
    lib.mirt_group_frame.argtypes = [C.c_void_p, C.c_int]
    lib.mirt_group_frame.restype = C.c_void_p
    lib.mirt_tile_buffer_pixels.argtypes = [C.c_int] * 6
    lib.mirt_tile_buffer_pixels.restype = C.c_int64
    lib.mirt_tile_owner.argtypes = [C.c_int] * 4
    lib.mirt_trace_ids.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int, C.c_int, C.POINTER(C.c_int32), dp, dp, dp]
    lib.mirt_trace_rays.argtypes = [C.c_void_p, C.c_int64, dp, dp, dp, C.c_double, dp, C.POINTER(C.c_int32), dp, dp]
    lib.mirt_scene_set_traversal.argtypes = [C.c_void_p, C.c_int]
    lib.mirt_last_error.restype = C.c_char_p
    lib.mirt_abi_version.restype = C.c_int
    return lib


def _declare_host(lib):
    lib.mirt_host_scene_synthetic.restype = C.c_void_p
    lib.mirt_host_scene_synthetic.argtypes = [C.c_int, C.c_uint64]
    lib.mirt_host_scene_synthetic_opt.restype = C.c_void_p
    lib.mirt_host_scene_synthetic_opt.argtypes = [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    lib.mirt_host_scene_load_opt.restype = C.c_void_p
    lib.mirt_host_scene_load_opt.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    lib.mirt_host_scene_write_ascii.restype = C.c_int
    lib.mirt_host_scene_write_ascii.argtypes = [C.c_void_p, C.c_char_p]
    lib.mirt_host_scene_load.restype = C.c_void_p
    lib.mirt_host_scene_load.argtypes = [C.c_char_p, C.c_char_p]
    lib.mirt_host_last_error.restype = C.c_char_p
    lib.mirt_host_last_error.argtypes = []
    lib.mirt_host_scene_free.argtypes = [C.c_void_p]
    lib.mirt_host_scene_free.restype = None
    lib.mirt_host_scene_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
    lib.mirt_host_scene_desc.restype = None
    lib.mirt_host_scene_camera_count.argtypes = [C.c_void_p]
    lib.mirt_host_scene_camera.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(Camera),
                                           C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mirt_host_bvh_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mirt_host_bvh_stats.restype = None
    lib.mirt_host_scene_dump.argtypes = [C.c_void_p, C.c_char_p]
    lib.mirt_host_bvh_dump.argtypes = [C.c_void_p, C.c_char_p]
    lib.mirt_host_write_ppm.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int]
    lib.mirt_host_read_ppm.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_double)]
    lib.mirt_host_config_defaults.argtypes = [C.POINTER(Config)]
    lib.mirt_host_config_defaults.restype = None
    return lib


_engine = None
_host = None


def engine_lib():
    """libmirt.so (the HIP engine).  Raises if it has not been built: there is no fallback."""
    global _engine
    if _engine is None:
        path = os.path.join(LIB_DIR, "libmirt.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make` (or __graft_entry__.build()); "
                               "the engine has no CPU fallback")
        _engine = _declare_engine(C.CDLL(path, mode=C.RTLD_GLOBAL))
    return _engine


def host_lib():
    """libmirt_host.so (scene loader, BVH builder, PPM I/O)."""
    global _host
    if _host is None:
        engine_lib()  # libmirt_host.so links against it
        path = os.path.join(LIB_DIR, "libmirt_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make`")
        _host = _declare_host(C.CDLL(path))
    return _host
