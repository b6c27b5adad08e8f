import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _dgs_tuning_follows_env(monkeypatch):
    """The library snapshots its DGS_* tuning overrides once per process (no getenv() on the launch path); tests flip them
    through monkeypatch, so every DGS_* change made that way - and the restore at teardown - is followed by a reload."""
    try:
        from dgsparse import _capi
    except Exception:  # CPU-only collection of tests that never import the package
        yield
        return
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def _setenv(name, value, *a, **k):
        setenv(name, value, *a, **k)
        if name.startswith('DGS_'):
            _capi.reload_tuning()

    def _delenv(name, *a, **k):
        delenv(name, *a, **k)
        if name.startswith('DGS_'):
            _capi.reload_tuning()

    monkeypatch.setenv, monkeypatch.delenv = _setenv, _delenv
    yield
    monkeypatch.undo()
    _capi.reload_tuning()
