import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


FIXTURES = os.path.join(ROOT, "fixtures")
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the reference's eight scenes, Test7 with a glass sphere added, and a hand-written scene with normal maps
SCENES = ["Test1", "Test2", "Test3", "Test4", "Test5", "Test6", "Test7", "def_scene", "Test7_glass", "NormalMap"]
# Test4 plus 600 small spheres (oracle/make_ties_scene.py): coincident surfaces in a scene above the 512
# primitives at which the engine chooses its large-scene walk; hits and depth-5 colours from the reference
TIES = ["Test4_ties"]
# generated scenes (raytracer_amd/host/synthetic.h): name -> (primitives, seed); oracle/make_goldens.py
# (and, for the cloud of a million, oracle/make_golden_million.py) feed the same scenes to the reference as ASCII files
SYNTHETIC = {"synthetic100k": (100000, 0x5EED), "synthetic1m": (1000000, 0x5EED)}


def scene_path(name):
    return os.path.join(FIXTURES, "ASCII", f"{name}.txt")


@pytest.fixture(scope="session")
def host_scenes():
    """name -> HostScene, loaded lazily and kept for the session."""
    from raytracer_amd import HostScene
    cache = {}

    def get(name):
        if name not in cache:
            if name in SYNTHETIC:
                cache[name] = HostScene.synthetic(*SYNTHETIC[name])
            else:
                cache[name] = HostScene(scene_path(name), FIXTURES)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def device_scenes(host_scenes):
    """(name, traversal="auto") -> DeviceScene on GPU 0, kept for the session."""
    from raytracer_amd import DeviceScene
    cache = {}

    def get(name, traversal="auto"):
        if (name, traversal) not in cache:
            ds = DeviceScene(host_scenes(name), 0)
            ds.set_traversal(traversal)
            cache[(name, traversal)] = ds
        return cache[(name, traversal)]
    return get
