import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest  # noqa: E402


class _SwitchAwareMonkeypatch:
    """pytest's `monkeypatch` whose setenv / delenv of an SDF_* name also refresh the framework's switch tables: the library and
    sdformerflow_amd.hip read the diagnostic SDF_* switches ONCE (csrc/switches.hip, hip.sw), not per call, so a test that scopes one
    says so through hip.reload_switches() - here, instead of at every call site."""

    def __init__(self, mp):
        self._mp = mp

    def __getattr__(self, name):
        return getattr(self._mp, name)

    @staticmethod
    def _reload():
        from sdformerflow_amd import hip
        hip.reload_switches()

    def setenv(self, name, value, prepend=None):
        self._mp.setenv(name, value, prepend)
        if name.startswith("SDF_"):
            self._reload()

    def delenv(self, name, raising=True):
        self._mp.delenv(name, raising)
        if name.startswith("SDF_"):
            self._reload()


@pytest.fixture
def monkeypatch(monkeypatch):
    mp = _SwitchAwareMonkeypatch(monkeypatch)
    yield mp
    monkeypatch.undo()              # (restore the environment first, then the tables)
    mp._reload()
