"""debug: entry point eager vs graph, per-iteration losses"""
import os, sys, tempfile
import numpy as np, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis as m
dev = torch.device('cuda:0')
res = {}
for graph in (False, True):
    root = tempfile.mkdtemp()
    base = dict(contrast_list=['T1', 'T2', 'T2_FLAIR'], input_height=64, input_width=96, batch_size=8, epochs=2, gpu='0', graph=graph,
                data_source='synthetic', ckpt_root=os.path.join(root, 'ckpt'), ckpt_timelabel='t0', lambda_adv_s=1.0, shuffle=True)
    open(os.path.join(root, 'config.yaml'), 'w').write(yaml.dump(base))
    cfg = m.train.setup_config(os.path.join(root, 'config.yaml'), device=dev)
    run = m.train.Run(cfg, log=lambda *a: None)
    inner = run.step
    log = []
    class W:
        def __init__(s): s.optimizer, s.optimizer_d_s = inner.optimizer, inner.optimizer_d_s
        def __getattr__(s, n): return getattr(inner, n)
        def __call__(s, *a, **k):
            out = inner(*a, **k)
            w = inner.optimizer.flat_p
            log.append((k.get('it'), float(out[0]), float(w.double().abs().sum()), float(a[0].double().sum()), a[3].sum().item() if a[3] is not None else None,
                        torch.get_rng_state().sum().item(), int(np.random.get_state()[2])))
            return out
    run.step = W()
    run.train(max_iters_per_epoch=8)
    res[graph] = log
for a, b in zip(res[False], res[True]):
    print('same' if a == b else 'DIFF', a, b)
