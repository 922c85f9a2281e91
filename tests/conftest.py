import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLDEN, "scenes")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """The product package (ctypes bindings over the HIP and host libraries)."""
    return importlib.import_module("ray-tracing-in-one-weekend_amd")


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def recipe(name):
    return os.path.join(SCENES, name + ".recipe")


TRACE_SCENES = ["book1", "book2", "stress4096", "two_perlin", "mixed", "ties"]
# ... and one whose bvh groups hold object_lists (no topology fixture: the reference's tree has list objects as leaves)
HIT_SCENES = TRACE_SCENES + ["lists"]
