import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "extras: an operator outside the hot path (bfloat16 / float32 tensors, fp8-E5M2 pages, "
                            "block-sparse attention, reshape_and_cache_flash, convert_fp8): runs on "
                            "libvmi_paged_attention_extras.so — every other test runs on the product library")


@pytest.fixture(autouse=True)
def _library_for_the_test(request):
    """Tests marked `extras` run inside _lib.use_extras(); all others on the product library (the default)."""
    if request.node.get_closest_marker("extras") is None:
        yield
        return
    from vllmini_amd import _lib

    with _lib.use_extras():
        yield


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
