import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)          # tests/golden/pn_inputs.py (seeded inputs shared with the fixture generators)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


AGREEMENT_DIR = os.path.join(ROOT, "gpurun_out", "parity")


def record_agreement(name, payload):
    """Write one measured-agreement record (robust / fragile counts, flips with their margins, max |dR| in 5-decimal
    units ...) to gpurun_out/parity/<name>.json on the GPU box; tools/collect_parity.py merges the records into the
    committed tests/golden/agreement_r02.json."""
    import json
    os.makedirs(AGREEMENT_DIR, exist_ok=True)
    clean = {k: v for k, v in payload.items() if k != "same_mask"}
    with open(os.path.join(AGREEMENT_DIR, name.replace("/", "_") + ".json"), "w") as f:
        json.dump(clean, f, indent=1)
