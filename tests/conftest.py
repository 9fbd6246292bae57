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
    from oracle import oracle as O
    return O.Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference C itself, where oracle/_ref was built (this container; it also travels to the
    GPU box as a built .so).  Tests that need it skip when it is absent."""
    from oracle import oracle as O
    if not O.have_reference():
        pytest.skip("oracle/_ref/libetsi_ref.so not built (no /root/reference here)")
    return O.Reference()
