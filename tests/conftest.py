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
    """tests switch kernel-selection options (hip.set_option) in-process: put the defaults back after each test."""
    yield
    m = sys.modules.get('mrdis')
    if m is not None and getattr(m.hip, '_lib', None) is not None:
        m.hip.set_option('wino', int(os.environ.get('MRDIS_WINO', '1')))
        m.hip.set_option('nt_mb', int(os.environ.get('MRDIS_NT_MB', '128')))
        m.hip.set_option('wino_pipe', int(os.environ.get('MRDIS_WINO_PIPE', '1')))
        m.hip.set_option('wino_u', int(os.environ.get('MRDIS_WINO_U', '1')))
        m.hip.set_option('wino4', int(os.environ.get('MRDIS_WINO4', '1')))
        m.hip.set_option('wino4r', int(os.environ.get('MRDIS_WINO4R', '1')))
        m.hip.set_option('debug_now16', 1 if 'MRDIS_DEBUG_NOW16' in os.environ else 0)
        m.hip.set_option('debug_nopack', 1 if 'MRDIS_DEBUG_NOPACK' in os.environ else 0)
