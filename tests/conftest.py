import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import load_oracle

    return load_oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled reference, where oracle/_ref has been built; else skip."""
    from oracle import load_reference

    ref = load_reference()
    if ref is None:
        pytest.skip("oracle/_ref/libbwtref.so not built")
    return ref


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the HIP path has no CPU fallback)")
    from bwt_amd import gpu as g

    g.init(0)
    return g
