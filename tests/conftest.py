import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "8")   # the oracle's OpenMP pool (a 256-thread default only oversubscribes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
