"""Laboratory tests (marker `probe`): not collected by `pytest tests/`.  On a GPU box:
    python tools/probes/lost_kernels/lost_ops.py && python -m pytest tools/probes/lost_kernels -q"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X")
    config.addinivalue_line("markers", "probe: laboratory kernels that are not part of libvdd_hip.so")
    import lost_ops
    lost_ops.build_lost_lib()
