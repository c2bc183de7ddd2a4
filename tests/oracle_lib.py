"""ctypes binding of oracle/liboracle.so -- the CPU checker.  Test infrastructure: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import ctypes as C
import os

import numpy as np

from raytracer_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RNG_SEQUENTIAL, RNG_KEYED, RNG_FRAME = 0, 1, 2


class OracleStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("primary_rays", "secondary_rays", "shadow_segments", "node_tests",
                                          "prim_tests")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle oracle`")
        _lib = C.CDLL(path)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        _lib.oracle_trace_ids.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(A.Camera), C.c_int, C.c_int, ip, dp, dp,
                                          dp, dp, ip]
        _lib.oracle_render.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(A.Camera), C.POINTER(A.Config), C.c_int,
                                       C.c_int, C.c_int, C.c_int, dp, C.POINTER(OracleStats)]
        _lib.oracle_render_window.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(A.Camera), C.POINTER(A.Config), C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp,
                                              C.POINTER(OracleStats)]
        _lib.oracle_trace_rays.argtypes = [C.POINTER(A.SceneDesc), C.c_int64, dp, dp, dp, C.c_double, dp, ip, dp, dp]
        _lib.oracle_max_threads.restype = C.c_int
    return _lib


def trace_ids(desc, camera, width, height):
    n = width * height
    out = {"prim_id": np.empty(n, np.int32), "front": np.empty(n, np.int32), "t": np.empty(n),
           "normal": np.empty((n, 3)), "uv": np.empty((n, 2)), "point": np.empty((n, 3))}
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    rc = lib().oracle_trace_ids(C.byref(desc), C.byref(camera), width, height, out["prim_id"].ctypes.data_as(ip),
                                out["t"].ctypes.data_as(dp), out["normal"].ctypes.data_as(dp),
                                out["uv"].ctypes.data_as(dp), out["point"].ctypes.data_as(dp),
                                out["front"].ctypes.data_as(ip))
    assert rc == 0
    return out


def trace_rays(desc, origin, direction, time=None, t_min=1e-5, t_max=None):
    from raytracer_amd.api import _trace_rays
    rc, out = _trace_rays(lib().oracle_trace_rays, C.byref(desc), origin, direction, time, t_min, t_max)
    assert rc == 0
    return out


def render(desc, camera, config, width, height, spp, rng_mode=RNG_KEYED):
    rgb = np.empty((height, width, 3))
    st = OracleStats()
    rc = lib().oracle_render(C.byref(desc), C.byref(camera), C.byref(config), width, height, spp, rng_mode,
                             rgb.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st))
    assert rc == 0
    return rgb, st.as_dict()


def render_window(desc, camera, config, width, height, spp, x0, y0, w, h, rng_mode=RNG_KEYED):
    """The window [x0, x0+w) x [y0, y0+h) of the width x height frame: a crop of render()'s frame."""
    rgb = np.empty((h, w, 3))
    st = OracleStats()
    rc = lib().oracle_render_window(C.byref(desc), C.byref(camera), C.byref(config), width, height, spp, rng_mode,
                                    x0, y0, w, h, rgb.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st))
    assert rc == 0
    return rgb, st.as_dict()


def psnr(a, b, peak=1.0):
    """PSNR of two frames after the writer's clamp to [0, 1] (Image.cpp:123-128)."""
    a = np.clip(a, 0.0, 1.0)
    b = np.clip(b, 0.0, 1.0)
    mse = float(np.mean((a - b) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)
