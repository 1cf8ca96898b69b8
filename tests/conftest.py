import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, 'tests') not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, 'tests'))          # tests/fixtures.py (seeded input builders)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='module')
def mrdis():
    """the product package on a GPU box; loading fails loudly when libmrdis_hip.so is missing."""
    import torch
    import mrdis as m
    assert torch.cuda.is_available()
    m.hip.load()
    return m


@pytest.fixture(autouse=True)
def _restore_library_options():
    """tests switch kernel-selection options (hip.set_option) in-process: EVERY switch of the library is snapshotted before a test and put back
    after it, so no test runs under the policy another one forced (bconv4, split6, debug_mode ...) and results do not depend on test order or -k."""
    m = sys.modules.get('mrdis')
    loaded = m is not None and getattr(m.hip, '_lib', None) is not None
    snap = m.hip.options_snapshot() if loaded else None
    yield
    m = sys.modules.get('mrdis')
    if m is not None and getattr(m.hip, '_lib', None) is not None:
        if snap is None:        # the library was first loaded inside this test: its load-time values (environment / defaults) are what to go back to
            snap = _LOAD_DEFAULTS.get('snap')
        if snap is not None:
            m.hip.options_restore(snap)


_LOAD_DEFAULTS = {}


@pytest.fixture(autouse=True, scope='session')
def _record_load_defaults():
    """the option table as the library built it from the environment, taken before any test can change it"""
    try:
        import mrdis as m
        m.hip.load()
        _LOAD_DEFAULTS['snap'] = m.hip.options_snapshot()
    except Exception:       # no library (a CPU box before build()): the tests that need it fail loudly on their own
        pass
    yield
