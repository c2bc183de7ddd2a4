"""raytracer_amd -- MI355X-native render hot path of the lovesahaj/raytracer renderer.

`api` wraps the two in-tree shared libraries: libmirt.so (HIP engine behind the C ABI of
include/mirt.h) and libmirt_host.so (the C++ drop-in surface: scene loader, BVH builder, PPM I/O).
"""
from .api import (DeviceGroup, DeviceScene, HostScene, MirtError, default_config, owned_tiles, read_ppm, scatter_tiles,  # noqa: F401
                  tile_buffer_pixels, write_ppm)
