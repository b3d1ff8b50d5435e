import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="module")
def ctx():
    """The default device context of the product library (lsq_amd.default_context); GPU tests only."""
    import lsq_amd
    return lsq_amd.default_context()


def pytest_collection_modifyitems(config, items):
    """File order IS the contract order (test_a_gpu_contract < test_b_gpu_kernels < ... < test_zz_gpu_stress); keep it even
    if someone passes the files in another order, and keep every stress test behind everything else."""
    def tier(item):
        name = os.path.basename(str(item.fspath))
        return (2 if name.startswith("test_zz_") else 0 if name.startswith("test_a_") else 1, )
    items.sort(key=tier)          # (stable: the order inside a tier is unchanged)
