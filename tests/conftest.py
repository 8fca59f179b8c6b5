import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CORPUS = os.path.join(ROOT, "data", "canterbury_corpus")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # PyTorch ships its own HIP runtime; it finds no device when the system runtime that
    # libbwtgpu.so links has been initialised first in the same process.  Let torch look first,
    # whatever subset of the tests is run.
    try:
        import torch

        torch.cuda.is_available()
    except Exception:
        pass


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def known_answers():
    return load_golden("known_answers.json")


@pytest.fixture(scope="session")
def corpus_golden():
    return load_golden("corpus_sha256.json")


@pytest.fixture(scope="session")
def synthetic_golden():
    return load_golden("synthetic.json")


@pytest.fixture(scope="session")
def mixed_golden():
    return load_golden("mixed.json")
