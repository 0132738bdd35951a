import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "gpu_scalar: run with torch-GPU scalar arithmetic (the package default) instead of torch-CPU's")
    config.addinivalue_line("markers", "both_scalar_forms: run once per scalar arithmetic (torch-CPU's, then the package default torch-GPU's); "
                                       "for tests whose checker is the oracle (which follows the switch), not a CPU-made golden")


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("both_scalar_forms") is not None:
        metafunc.parametrize("scalar_form", [False, True], ids=["cpu_scalar", "gpu_scalar"])


@pytest.fixture
def scalar_form():
    """None: decided by the `gpu_scalar` marker; tests marked `both_scalar_forms` get False / True here (pytest_generate_tests)."""
    return None


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _cpu_scalar_semantics_for_cpu_made_goldens(request, scalar_form):
    """The package default is torch-GPU scalar arithmetic (sampling.GPU_SCALAR_SEMANTICS = True).  Most fixtures and the oracle's
    default follow the reference run on the build container's CPU, so tests run the kernels in the torch-CPU form unless they are
    marked `gpu_scalar` (those check the default / the second golden set) or `both_scalar_forms` (engine / processor tests whose
    checker is the oracle loop: they run in both forms, the package default included)."""
    import llava_align_amd.sampling as S
    from oracle import vdd_oracle as O
    want = (request.node.get_closest_marker("gpu_scalar") is not None) if scalar_form is None else bool(scalar_form)
    old = (S.GPU_SCALAR_SEMANTICS, O.GPU_SCALAR)
    S.GPU_SCALAR_SEMANTICS = O.GPU_SCALAR = want
    yield
    S.GPU_SCALAR_SEMANTICS, O.GPU_SCALAR = old
