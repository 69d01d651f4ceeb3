import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from speech_to_image_translation_without_text_amd import _lib
    _lib.load()
    _lib.require_device()
    return torch.device("cuda:0")


# The CPU oracle's convolutions (MKLDNN) sum in an order that depends on the intra-op thread count, and some tests compare
# quantities that sit on LeakyReLU decisions (a 1e-7 change of a pre-activation can flip one).  One fixed thread count for
# the whole session -- re-applied before every test, because the multi-process tests change it -- makes the oracle's
# results independent of test order and of the box's core count.
ORACLE_THREADS = 8


@pytest.fixture(autouse=True)
def _fixed_oracle_threads():
    import torch
    if torch.get_num_threads() != ORACLE_THREADS:
        torch.set_num_threads(ORACLE_THREADS)
    yield
