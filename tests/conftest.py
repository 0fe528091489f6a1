import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--slow", action="store_true", default=False,
                     help="also run the tests marked `slow` (long duplicates of cases the default GPU suite already covers)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` via gpurun)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")
    config.addinivalue_line("markers", "slow: a long GPU case whose code path a faster case of the default suite also takes; "
                                       "deselected unless --slow is given (`pytest -m gpu --slow` runs everything)")


def pytest_collection_modifyitems(config, items):
    import torch
    has_gpu = torch.cuda.is_available()
    has_ref = os.path.isdir("/root/reference/nanovllm")
    if not config.getoption("--slow"):
        slow = [it for it in items if "slow" in it.keywords]
        if slow:
            config.hook.pytest_deselected(items=slow)
            items[:] = [it for it in items if "slow" not in it.keywords]
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU"))
        if "reference" in item.keywords and not has_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))
