"""pytest configuration: the `gpu` marker, package loading by path (the package directory is `vits.cpp_amd`, with a
dot, so it cannot be imported by name), and shared fixtures (synthetic model bytes, golden taps)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


def load_package():
    name = "vits_cpp_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "vits.cpp_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def full_bytes(pkg):
    return pkg.synth_model_bytes(0x5EED, pkg.SYNTH_FULL)


@pytest.fixture(scope="session")
def tiny_bytes(pkg):
    return pkg.synth_model_bytes(0x5EED, pkg.SYNTH_TINY)


@pytest.fixture(scope="session")
def tiny_hf_bytes():
    with open(os.path.join(GOLDEN, "tiny_hf_export.ggml"), "rb") as f:
        return f.read()


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def rel_err(a, b):
    """max |a-b| relative to the RMS of the expected array b"""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    assert a.size == b.size, (a.size, b.size)
    rms = np.sqrt((b ** 2).mean()) + 1e-30
    return float(np.abs(a - b).max() / rms)
