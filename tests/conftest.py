import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "subprocess_only: the test only starts other processes (no allocator poisoning in this one)")


@pytest.fixture(autouse=True)
def _restore_default_dtype():
    # the reference's tests leak torch.set_default_dtype(float64); ours never do
    import torch
    old = torch.get_default_dtype()
    yield
    torch.set_default_dtype(old)


@pytest.fixture(autouse=True)
def _poison_free_device_memory(request):
    """GPU tests: what the caching allocator hands out next is NaN / Inf, not the zeros of a fresh process or the leftovers of the
    previous test -- a kernel that reads scratch it never wrote fails here instead of passing by luck."""
    if request.node.get_closest_marker("gpu") is None or request.node.get_closest_marker("subprocess_only") is not None:
        yield
        return
    import torch
    if torch.cuda.is_available():
        # ASG_TEST_POISON_MIB: size of the poisoned block (default 256 MiB, what tools/fuzz_routes.py uses; the block stays in the
        # caching allocator, so every later allocation of the test that fits is carved from poisoned memory)
        mib = int(os.environ.get("ASG_TEST_POISON_MIB", "256"))
        if mib > 0:
            junk = torch.full((mib * 1024 * 1024 // 4,), float("nan"), device="cuda:0")
            junk[::2] = float("inf")
            del junk
    yield


@pytest.fixture(autouse=True)
def _reload_library_switches(request):
    """The library caches its ASG_* developer switches; tests that change them (util.setenv) must not leak the change: read the
    environment again once the test -- and monkeypatch's restoration, which runs before this finaliser -- is over."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        try:
            from torch_asg_amd import _lib
            _lib.lib().asg_reload_env()
        except Exception:
            pass
