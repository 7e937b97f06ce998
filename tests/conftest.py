"""pytest configuration: marker registration, module loading helpers shared by all tests.

`-m "not gpu"` tests run anywhere (oracle vs golden vectors / real reference, host front, ABI symbol export);
`-m gpu` tests are the parity tests proper and call the HIP path through the C ABI."""
import importlib.util
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(REPO, "assets")
GOLDEN = os.path.join(REPO, "tests", "golden")
os.environ["CRT_ENABLE_DEBUG_HOOKS"] = "1"      # the tests steer the library through its diagnostic environment switches (dead in a process that does not opt in)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_crt():
    name = "cpu_ray_tracer_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def crt():
    mod = load_crt()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def ref(orc):
    """The real reference compiled in place (oracle/_ref).  Only exists where /root/reference is mounted."""
    if not os.path.exists(orc.REF_LIB_PATH):
        pytest.skip("oracle/_ref not built (reference tree absent): real-reference pinning tests run in the authoring container only")
    return orc.Ref()


def scene_path(name):
    return os.path.join(ASSETS, "scenes", name)
