import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def switches(monkeypatch):
    """``switches(IVG_X="0", IVG_Y=None)``: set / delete IVG_* variables and publish them to the loaded library (the switch table
    of csrc/switches.h is read at load, at ivg_create and on ivg_reload_switches -- an op-level test that flips a switch between two
    launches has to say so).  The environment and the library's table are restored after the test.  The A/B switches are development
    tools the library honours only under IVG_DEV=1 (csrc/switches.h): the fixture sets it."""
    from ivideogpt_amd import _lib

    def apply(**kv):
        monkeypatch.setenv("IVG_DEV", "1")
        for k, v in kv.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, str(v))
        _lib.reload_switches()
    yield apply
    monkeypatch.undo()
    _lib.reload_switches()
