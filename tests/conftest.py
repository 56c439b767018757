"""pytest wiring: `gpu` marker, import paths, shared fixture loaders."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "llm-mixed-q_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_quantizers():
    meta = json.loads((GOLDEN / "quantizers.json").read_text())
    data = np.load(GOLDEN / "quantizers.npz")
    return meta, data


@pytest.fixture(scope="session")
def golden_modules():
    meta = json.loads((GOLDEN / "modules.json").read_text())
    data = np.load(GOLDEN / "modules.npz")
    return meta, data


@pytest.fixture(scope="session")
def golden_config_profile():
    return json.loads((GOLDEN / "config_profile.json").read_text())
