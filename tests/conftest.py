"""Shared pytest configuration: the ``gpu`` marker and fixture loaders."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def g1():
    return load_json("g1_abridged.json")


@pytest.fixture(scope="session")
def g2_small():
    return load_npz("g2_loop_T512.npz")


@pytest.fixture(scope="session")
def g2_full():
    return load_json("g2_loop_T10000.json")


@pytest.fixture(scope="session")
def g3():
    return load_json("g3_stop_rule.json")


@pytest.fixture(scope="session")
def g4():
    return load_npz("g4_init.npz"), load_json("g4_init.json")


@pytest.fixture(scope="session")
def g5():
    return load_npz("g5_transform_reg.npz")
