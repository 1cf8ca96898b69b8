"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and
exports exactly the entry points include/mrdis.h declares (no compute calls here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def mrdis():
    import mrdis as m
    if not os.path.exists(m.LIB_PATH):
        subprocess.check_call(['make', '-C', os.path.dirname(m.LIB_PATH), 'libmrdis_hip.so'])
    return m


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'mrdis.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(mrdis_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol(mrdis):
    lib = mrdis.hip.load()
    declared = header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/mrdis.h but not exported'
    assert sorted(mrdis.hip.EXPORTED_SYMBOLS) == declared, 'python binding table drifted from the header'


def test_version_and_strerror(mrdis):
    lib = mrdis.hip.load()
    assert lib.mrdis_version() >= 100
    assert lib.mrdis_strerror(0) == b'ok'
    assert lib.mrdis_strerror(-3) == b'workspace too small'


def test_workspace_queries_are_host_only(mrdis):
    lib = mrdis.hip.load()
    assert lib.mrdis_conv2d_bwd_weight_workspace(32, 256, 256, 4, 32, 3, 3, 1, 1) > 0
    assert lib.mrdis_conv2d_bwd_weight_workspace(2, 8, 8, 4, 4, 5, 5, 1, 2) == 0      # 25 taps: unsupported
    assert lib.mrdis_norm_workspace(1, 1 << 20, 64) >= 2 * 64 * 4
    assert lib.mrdis_sumsq_workspace() > 0


def test_missing_library_fails_loudly(mrdis, tmp_path):
    import importlib
    hip = mrdis.hip
    saved = hip._lib
    hip._lib = None
    try:
        with pytest.raises(hip.MrdisLibraryError):
            hip.load(str(tmp_path / 'nope.so'))
    finally:
        hip._lib = saved


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'representation-disentanglement_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert 'oracle' not in src.replace('# oracle', ''), fn


def test_mix_job_struct_matches_library():
    """the ctypes mirror of csrc `MixJob` (hip.MixJob) has the library's size; job block counts are positive (host-only calls)"""
    import ctypes
    import mrdis
    lib = mrdis.hip.load()
    assert ctypes.sizeof(mrdis.hip.MixJob) == lib.mrdis_mix_job_bytes() == 368
    assert 1 <= lib.mrdis_mix_job_blocks(32, 16, 9) <= 128


def test_option_and_counter_tables_match_the_library(mrdis):
    """hip.OPTION_NAMES / hip.KERNEL_FAMILIES (what tests/conftest.py snapshots and the parity tests assert on) are names the library knows;
    the dynamic-LDS table hands out whole lines only (host-only calls)."""
    hip = mrdis.hip
    lib = hip.load()
    snap = hip.options_snapshot()
    assert len(snap) == len(hip.OPTION_NAMES) == len(set(hip.OPTION_NAMES))
    for name, v in snap.items():
        hip.set_option(name, v)                    # raises on a name the library does not know
    assert lib.mrdis_set_option(b'no_such_option', 1) == -1
    for fam in hip.KERNEL_FAMILIES:
        assert lib.mrdis_launch_count(fam.encode()) >= 0, fam
    assert lib.mrdis_launch_count(b'no_such_family') == -1
    assert isinstance(hip.dynamic_lds(), dict)
    import ctypes
    tiny = ctypes.create_string_buffer(4)
    assert lib.mrdis_dynamic_lds_table(tiny, 4) >= 0 and b'=' not in tiny.value.replace(b'\n', b'')[:0]


def test_library_issues_no_memset_or_memcpy_calls():
    """the training step is recorded into HIP graphs (trainer.GraphedTrainStep): a hipMemsetAsync node did not run again on later replays (round 6, max_pool
    backward), so device buffers are cleared / copied by kernels only"""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'representation-disentanglement_amd', 'csrc')
    bad = []
    for f in sorted(glob.glob(os.path.join(root, '*.hip')) + glob.glob(os.path.join(root, '*.h'))):
        for n, line in enumerate(open(f), 1):
            code = line.split('//')[0]
            if re.search(r'hipMem(set|cpy)\w*\s*\(', code):
                bad.append(f'{os.path.basename(f)}:{n}')
    assert not bad, bad
