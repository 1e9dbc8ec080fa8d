import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (os.path.join(ROOT, "gimp-fix-ca_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# the suite drives the kernels' tuning knobs (FIXCA_TH, FIXCA_NO_ROLL, ...); the library honours them
# only under FIXCA_DEBUG (tests/test_host_logic.py::test_tuning_knobs_need_debug checks the gate)
os.environ.setdefault("FIXCA_DEBUG", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size CPU oracle runs (several seconds each)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
