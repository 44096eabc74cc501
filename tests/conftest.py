"""pytest configuration: registers the `gpu` marker and shared fixtures/paths."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_bin():
    """Path to the CPU oracle CLI (built on demand; test infrastructure only)."""
    path = os.path.join(ROOT, "oracle", "depth_oracle")
    lib = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(path) and os.path.exists(lib)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return path


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
