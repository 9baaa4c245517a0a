import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a GPU test that hangs (a lost event, a wedged queue) must fail by itself instead of eating the whole run's time limit
    # (the marker belongs to the pytest-timeout plugin: without it the guard is simply absent, not an unknown-marker error)
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900, method="thread"))


def gjson(name):
    return json.loads((GOLDEN / f"{name}.json").read_text(encoding="utf-8"))


def gnpz(name):
    return np.load(GOLDEN / f"{name}.npz")


@pytest.fixture(scope="session")
def pack():
    """The product package (hyphenated directory, loaded the way ComfyUI loads a custom node)."""
    from packload import load_pack
    return load_pack()
