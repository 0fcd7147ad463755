import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402

load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tmp_ckpt_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("ckpt")


_CKPT_CACHE = {}


def checkpoint(tmp_dir, geom_name: str, seed: int = 1234):
    """Write (once per session) the synthetic GGUF for a geometry and return (path, geometry, tensors)."""
    from zerovox_cpp_amd import gguf, synth
    key = (geom_name, seed)
    if key not in _CKPT_CACHE:
        g = synth.GEOMETRIES[geom_name]
        path = os.path.join(str(tmp_dir), f"{geom_name}_{seed}.gguf")
        synth.write_checkpoint(path, g, seed)
        _, tensors = gguf.read_gguf(path)
        _CKPT_CACHE[key] = (path, g, tensors)
    return _CKPT_CACHE[key]


@pytest.fixture(scope="session")
def ckpt(tmp_ckpt_dir):
    return lambda name, seed=1234: checkpoint(tmp_ckpt_dir, name, seed)
