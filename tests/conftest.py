import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts: compile the HIP library, the oracle and the C++ test program once
    (hipcc cross-compiles gfx950 without a GPU; about a minute).  With them present this is a no-op `make`."""
    lib = os.path.join(ROOT, "halo2-pse_amd", "libhalo2hip.so")
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    if not (os.path.exists(lib) and os.path.exists(exe)):
        import __graft_entry__
        __graft_entry__.build()


def load_pkg():
    """The product package lives in `halo2-pse_amd/` (not an importable identifier): load it
    under the module name halo2_pse_amd."""
    if "halo2_pse_amd" in sys.modules:
        return sys.modules["halo2_pse_amd"]
    path = os.path.join(ROOT, "halo2-pse_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(
        "halo2_pse_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["halo2_pse_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session")
def h2():
    return load_pkg()
