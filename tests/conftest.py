import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "emul")):
    if p not in sys.path:
        sys.path.insert(0, p)


SUMMARY_LINES = []  # one-line findings a test wants in the terminal summary even under -q (e.g. "VIENNA: absent")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    for line in SUMMARY_LINES:
        terminalreporter.write_line(line)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_python_half.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def default_params():
    from scanfold_amd import params
    return params.default_params()


@pytest.fixture()
def oracle(default_params):
    """The CPU oracle with the default parameter set loaded (test infrastructure only)."""
    from oracle import oracle as orc
    orc.build()
    orc.set_params(default_params)
    return orc


@pytest.fixture(scope="session")
def gpu_engine():
    """The product engine on cuda:0 through the C ABI; fails (not skips) if the HIP library is missing."""
    from scanfold_amd import _lib
    return _lib.get_engine(0)


def random_seqs(rng, n, W, alphabet=b"ACGU"):
    import numpy as np
    return np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), (n, W))]
