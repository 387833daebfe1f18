import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "8")   # the oracle's OpenMP pool (a 256-thread default only oversubscribes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest


@pytest.fixture(autouse=True)
def _collect_device_objects_between_gpu_tests(request):
    """Runners hold HIP graphs, side streams and events in reference cycles; left to Python's cyclic collector they are finalised at an
    arbitrary later moment — e.g. in the middle of a LATER test's graph replay, which the HIP runtime answers with a segmentation fault
    (seen with tests/test_gpu_update_golden.py run on its own).  Collect them at the test boundary, with the device idle."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
