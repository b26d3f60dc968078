import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    import oraclelib
    oraclelib.build_oracle()
    yield


@pytest.fixture(scope="session", autouse=True)
def _slot_option_of_the_run():
    """TGPU_TEST_SLOT=0|1|2 (read HERE, by the test harness -- the library itself reads no environment variable): the whole GPU suite
    under another setting of TGPU_OPT_SLOT than the default (0: the lane-per-block trellis kernels of rounds 1-5, 2: front end and
    trellises of device-walk batches in one launch where a channel has a code to decode on).  Tests that set the option themselves
    (the A/B tests) put it back to this value."""
    v = os.environ.get("TGPU_TEST_SLOT")
    if v is not None:
        import torch
        if torch.cuda.is_available():
            import osmo_tetra_amd as T
            T.set_option(T.OPT_SLOT, int(v))
    yield
