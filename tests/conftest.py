import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# libkmcpgpu.so honours its test hooks (KMCPG_TEST_MAX_BASES: fake ENOMEM above a batch size) only when this is set, and reads
# it once per process, at the first GPU-half call
os.environ.setdefault("KMCPG_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as O
    O.build()
    return O
