import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """Import conjugate-gradient_amd/ (the hyphen keeps it from a plain import)."""
    name = "conjugate_gradient_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "conjugate-gradient_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    return orc.Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle import oracle as orc
    return orc.Reference()


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def lib(pkg):
    return pkg.load_library()


@pytest.fixture(params=["auto", "rows", "planes"])
def engine_env(request, monkeypatch):
    """Run a test once per lexicographic engine: the automatic choice (which sends the small grids
    the oracle can check to the scalar engine) and the rows / planes engines forced wherever the
    grid allows them (CG_ENGINE is ignored where it does not)."""
    if request.param == "auto":
        monkeypatch.delenv("CG_ENGINE", raising=False)
    else:
        monkeypatch.setenv("CG_ENGINE", {"rows": "1", "planes": "3"}[request.param])
    return request.param
