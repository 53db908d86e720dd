import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def rules_golden():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "rules.npz")))


@pytest.fixture(scope="session")
def mcts_golden():
    import json
    return json.load(open(os.path.join(GOLDEN, "mcts.json")))


@pytest.fixture(scope="session")
def mcts_deep_golden():
    """Reference searches at the metric's depth (playout 1600) and with selected paths of 40-60 levels (gen_mcts_deep)."""
    import json
    return json.load(open(os.path.join(GOLDEN, "mcts_deep.json")))


@pytest.fixture(scope="session")
def tables_golden():
    import json
    return json.load(open(os.path.join(GOLDEN, "tables.json")))
