"""Instruction census of the barrier-to-barrier segments of every MFMA kernel in a .hip file (cross-compiled for gfx950, no GPU needed):
MFMA / VALU / LDS / VMEM / SALU / waitcnt counts per segment, the most frequent VALU opcodes (v_pk_* = SLP-packed fp32, an anti-lever beside
MFMAs), which LDS opcodes (ds_read2_* / ds_write2* = hipcc paired neighbouring accesses: half rate, 32-bank rule), scratch traffic.

    python tools/isa_census.py representation-disentanglement_amd/csrc/mrdis_wino4.hip [-fno-slp-vectorize ...] [--min-mfma 16]
"""
import collections
import os
import re
import subprocess
import sys


def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'lds_rd'
    if op.startswith('ds_write') or op.startswith('ds_store'): return 'lds_wr'
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    if op.startswith('buffer_') or op.startswith('global_') or op.startswith('flat_'): return 'vmem'
    return 'other'


def main():
    src = sys.argv[1]
    flags = [a for a in sys.argv[2:] if not a.startswith('--min-mfma')]
    mm = 16
    for i, a in enumerate(sys.argv):
        if a == '--min-mfma':
            mm = int(sys.argv[i + 1]); flags = [f for f in flags if f != sys.argv[i + 1]]
    out = '/tmp/isa_census.s'
    subprocess.check_call(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '--cuda-device-only', '-S', '-o', out, os.path.abspath(src)] + flags,
                          stderr=subprocess.DEVNULL, cwd=os.path.dirname(os.path.abspath(src)))
    kern, body = None, collections.defaultdict(list)
    for l in open(out):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            kern = m.group(1)
        t = l.strip()
        if kern and t and not t.startswith(';') and not t.startswith('.') and not t.endswith(':'):
            body[kern].append(t)
    for k, b in body.items():
        c = collections.Counter(cls(x.split()[0]) for x in b)
        if c['mfma'] < mm:
            continue
        name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()[:100]
        print(f'{name}\n   whole kernel: {dict(c)}')
        seg, cur = [], []
        for x in b:
            cur.append(x)
            if x.startswith('s_barrier'):
                seg.append(cur); cur = []
        seen = set()
        for sg in seg:
            cc = collections.Counter(cls(x.split()[0]) for x in sg)
            if cc['mfma'] < mm:
                continue
            vops = collections.Counter(x.split()[0] for x in sg if x.startswith('v_') and not x.startswith('v_mfma'))
            dsops = collections.Counter(x.split()[0] for x in sg if x.startswith('ds_'))
            key = (tuple(sorted(cc.items())), tuple(sorted(dsops.items())))
            if key in seen:
                continue
            seen.add(key)
            non = sum(v for kk, v in cc.items() if kk not in ('mfma', 'barrier'))
            print(f'   segment: {dict(cc)}  = {non / cc["mfma"]:.2f} other instructions per MFMA (VALU {cc["valu"] / cc["mfma"]:.2f})')
            print(f'        VALU: {vops.most_common(7)}')
            print(f'        LDS: {dict(dsops)}')


if __name__ == '__main__':
    main()
