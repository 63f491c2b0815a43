import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_cases():
    with open(os.path.join(GOLD, "search_cases.json")) as f:
        return {c["id"]: c for c in json.load(f)["cases"]}


@pytest.fixture(scope="session")
def golden_rows():
    return dict(np.load(os.path.join(GOLD, "rows.npz")))


@pytest.fixture(scope="session")
def golden_chips():
    with open(os.path.join(GOLD, "chips_sha256.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_nco():
    with open(os.path.join(GOLD, "nco_indices.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def icd_kat():
    with open(os.path.join(GOLD, "icd_kat.json")) as f:
        return json.load(f)


def case_iq(case):
    """Regenerate a golden case's samples from its stored spec and check them against the stored hash."""
    import hashlib
    from gnss_dsp_tools_amd import signals, synth
    sig = signals.get(case["script"])
    blocks = max(sig.blocks(case["ms"]), 0)
    x = synth.make_iq(sig, blocks, case["seed"], [tuple(s) for s in case["sats"]], nsamp=case["nsamp"])
    assert hashlib.sha256(x.tobytes()).hexdigest() == case["x_sha256"], "synthetic IQ generator drifted"
    return x


@pytest.fixture(scope="session")
def engine():
    """One Engine on cuda:0 for the whole GPU session (fails loudly without a GPU / without libgacq.so)."""
    from gnss_dsp_tools_amd import acquire
    eng = acquire.Engine(0)
    yield eng
    eng.close()
