"""Round-6 probe of the full-resolution F(4x4) kernels (sp6 dgrad = wino4r_kernel, sp6 fwd_spade = wino4_kernel<0, true>): time and PMC
HBM traffic under the library's existing switches -- which form re-reads less, and is the re-read what the time goes to?

    python tools/w4_probe.py            # variants: wino4r 1 / 3 (64-tile / channel-split form), nt_mb 128 / 100000 (non-temporal / plain output stores)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mrdis  # noqa: E402


def one(entry, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        for k, v in env.items():
            mrdis.hip.set_option(k[len('MRDIS_'):].lower(), int(v))
        r = bench.roofline_step(mrdis, torch.device('cuda:0'), 32, 256, 256, 'f32', iters=10, only=('sp6.gamma+beta', entry))
        us = r['layers'][0][entry]['us']
        tr, src, kn = bench.measure_traffic_inrun(script=('step_kernel.py', 'sp6.gamma+beta', entry, '32', '256', '256'), kernel_substr='wino', n_last=6,
                                                  what='tools/step_kernel.py')
        ab = r['layers'][0][entry]['algorithmic_bytes']
        print(json.dumps({'entry': entry, 'options': env, 'us': us, 'kernel': kn, 'traffic_MB': None if tr is None else round(tr / 1e6, 1),
                          'traffic_over_algorithmic': None if tr is None else round(tr / ab, 3), 'src': (src or '')[-120:]}), flush=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


if __name__ == '__main__':
    mrdis.hip.load()
    snap = mrdis.hip.options_snapshot()
    for entry, env in (('dgrad', {}), ('dgrad', {'MRDIS_WINO4R': 3}), ('dgrad', {'MRDIS_NT_MB': 100000}), ('dgrad', {'MRDIS_WINO4R': 3, 'MRDIS_NT_MB': 100000}),
                       ('fwd_spade', {}), ('fwd_spade', {'MRDIS_NT_MB': 100000}), ('fwd', {}), ('wgrad', {})):
        one(entry, env)
        mrdis.hip.options_restore(snap)
