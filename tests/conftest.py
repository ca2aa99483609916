import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running oracle check, skipped unless SDP_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("SDP_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="set SDP_SLOW=1 to run")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import sdpref
    sdpref.build()
    return sdpref


@pytest.fixture(scope="session")
def sia():
    import stochastic_inventory_amd
    return stochastic_inventory_amd
