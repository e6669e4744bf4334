import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def harness():
    """oracle/harness.py with the checker libraries built (the reference build only where /root/reference exists)."""
    from oracle import harness as H
    targets = ["oracle", "synth"]
    if os.path.isdir("/root/reference/source"):
        targets.append("ref")
    H.build(targets)
    return H


@pytest.fixture(scope="session")
def oracle(harness):
    b = harness.oracle_backend()
    yield b
    b.close()


@pytest.fixture(scope="session")
def gpu(harness):
    """The HIP path seen through the same entry-point set as reference and oracle."""
    import jpegsnoop_amd
    lib = jpegsnoop_amd.load()          # raises loudly when the extension or the device is missing
    b = harness.Backend(lib, "jsnoop_", "hip")
    yield b
    b.close()
