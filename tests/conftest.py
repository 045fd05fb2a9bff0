import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Foldseek's mat3di.out cannot be shipped: tests and benchmarks opt into the seeded stand-in matrix
# explicitly; without this the engine refuses to run (uc_options.cpp:finalize_params)
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# torch bundles its own HIP runtime: load it BEFORE libunicore_cluster.so so the process has one runtime (a second
# one initialised later reports "No HIP GPUs are available"); bench.py imports torch first too
try:
    import torch  # noqa: F401
except Exception:   # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")


@pytest.fixture(scope="session")
def root():
    return ROOT
