"""The runnable entry (mrdis/train.py = the reference's src/main_missing.py as functions).
CPU part: config / checkpoint-directory plumbing, stat.csv rows, checkpoint layout vs the reference-generated layout
fixture (tests/golden/ckpt_layout_m2.json).  GPU part: BASELINE configs[0] -- 10 training steps at M = 2, B = 4 from a
yaml, checkpoints written, resumed (continue_train), next step bit-identical."""
import json
import os

import numpy as np
import pytest
import torch
import yaml


@pytest.fixture(scope='module')
def m():
    import mrdis
    return mrdis


def test_config_dir_and_saved_yaml_merge(m, tmp_path):
    """main_missing.py:25-58: <root>/<dataset>/<model>/<label>; an existing directory's config.yaml wins over the current
    file except for `phase` and `continue_train`."""
    cfgfile = tmp_path / 'config.yaml'
    base = dict(contrast_list=['T1', 'T2'], input_height=64, input_width=64, batch_size=4, ckpt_root=str(tmp_path / 'ckpt'),
                ckpt_timelabel='run_a', continue_train=True, lr=0.001)
    cfgfile.write_text(yaml.dump(base))
    cpu = torch.device('cpu')
    c1 = m.train.setup_config(str(cfgfile), device=cpu)
    assert c1['ckpt_path'] == os.path.join(str(tmp_path / 'ckpt'), 'BraTS', 'MultimodalModel', 'run_a')
    assert c1['in_num_ch'] == 14 and c1['is_discrim_s'] is False and c1['input_output_act'] == 'no'
    saved = yaml.safe_load(open(os.path.join(c1['ckpt_path'], 'config.yaml')))
    assert saved['lr'] == 0.001 and 'device' not in saved                      # only plain values are written (util.py:913-925)
    cfgfile.write_text(yaml.dump(dict(base, lr=0.5, phase='test', lambda_adv_s=1.0)))
    c2 = m.train.setup_config(str(cfgfile), device=cpu)
    assert c2['lr'] == 0.001 and c2['phase'] == 'test'                         # saved yaml wins, except phase / continue_train
    assert c2['lambda_adv_s'] == 0.0 and c2['is_discrim_s'] is False
    c3 = m.train.setup_config(str(cfgfile), overrides={'continue_train': False, 'ckpt_timelabel': None}, device=cpu)
    assert c3['ckpt_path'] != c1['ckpt_path']                                  # fresh time label


def test_stat_csv_rows(m, tmp_path):
    """util.py:854-866: header = ['info'] + sorted keys on first use, one appended row per call (pandas index column)."""
    cfg = {'ckpt_path': str(tmp_path)}
    m.train.save_result_stat({k: float(i) for i, k in enumerate(m.LOSS_KEYS)}, cfg, info='epoch[ 0]')
    m.train.save_result_stat({k: 2.0 * i for i, k in enumerate(m.LOSS_KEYS)}, cfg, info='val')
    lines = open(tmp_path / 'stat.csv').read().strip().split('\n')
    assert lines[0] == ',info,' + ','.join(sorted(m.LOSS_KEYS))
    assert lines[1].startswith('0,epoch[ 0],') and lines[2].startswith('0,val,') and len(lines) == 3
    assert len(lines[1].split(',')) == 2 + len(m.LOSS_KEYS)


def test_checkpoint_layout_matches_reference(m, golden_dir):
    """a state_dict written by the reference's own classes (layout fixture: every key, shape, dtype and sum of the seeded
    init) and this model's state_dict are the same set of tensors: a reference checkpoint loads with zero skipped keys."""
    lay = json.load(open(os.path.join(golden_dir, 'ckpt_layout_m2.json')))
    cfg = dict(m.DEFAULT_CONFIG); cfg.update(contrast_list=['a', 'b'], lambda_adv_s=1.0, lambda_recon_y=1.0)
    cfg = m.derive_config(cfg, torch.device('cpu'))
    torch.manual_seed(10); np.random.seed(10)
    model = m.build_model(cfg)
    sd = model.state_dict()
    assert set(sd) == set(lay['model'])                                        # incl. output_decoder.*, discrim_s.*, dead convs, buffers
    fake = {}
    for k, rec in lay['model'].items():
        assert list(sd[k].shape) == rec['shape'] and str(sd[k].dtype).replace('torch.', '') == rec['dtype'], k
        if not k.startswith('discrim_s.') and sd[k].dtype.is_floating_point:   # the reference's discriminator init is torch-default, ours too,
            got = float(sd[k].double().sum())                                  # but its RNG position differs (constructed after output_decoder)
            assert abs(got - rec['sum']) <= 1e-6 * max(1.0, abs(rec['sum'])), k
        fake[k] = torch.full(rec['shape'], 0.25, dtype=sd[k].dtype)
    assert m.load_checkpoint_model(model, fake) == []                          # nothing skipped (util.py:895-903)
    assert all(bool((v == 0.25).all()) for k, v in model.state_dict().items() if v.dtype.is_floating_point)
    assert lay['ckpt_keys'][:6] == ['epoch', 'monitor_metric', 'stat', 'optimizer', 'scheduler', 'model']


@pytest.mark.gpu
def test_ten_steps_from_yaml_then_resume(m, tmp_path):
    """BASELINE configs[0] (plumbing): 2-modality synthetic 128x128 slices, batch 4, 10 steps through the entry point;
    epochNNN / model_best / stat.csv / config files appear; continue_train reloads model + optimizer + scheduler and the
    next training step is bit-identical to the one the original run takes."""
    dev = torch.device('cuda:0')
    base = dict(contrast_list=['T1', 'T2'], input_height=128, input_width=128, batch_size=4, epochs=2, gpu='0',
                data_source='synthetic', ckpt_root=str(tmp_path / 'ckpt'), ckpt_timelabel='t0', lambda_adv_s=1.0, shuffle=True)
    (tmp_path / 'config.yaml').write_text(yaml.dump(base))
    cfg = m.train.setup_config(str(tmp_path / 'config.yaml'), device=dev)
    logs = []
    run = m.train.Run(cfg, log=logs.append)
    assert run.step.accum == 4
    run.train(max_iters_per_epoch=5)                                           # 2 epochs x 5 iterations = 10 steps
    d = cfg['ckpt_path']
    for f in ('epoch000.pth.tar', 'epoch001.pth.tar', 'model_best.pth.tar', 'stat.csv', 'config.yaml', 'config.txt'):
        assert os.path.exists(os.path.join(d, f)), f
    rows = open(os.path.join(d, 'stat.csv')).read().strip().split('\n')
    assert len(rows) == 5 and [r.split(',')[1] for r in rows[1:]] == ['epoch[ 0]', 'val', 'epoch[ 1]', 'val']
    ck = torch.load(os.path.join(d, 'epoch001.pth.tar'), weights_only=False)
    assert set(ck) == {'epoch', 'monitor_metric', 'stat', 'optimizer', 'scheduler', 'model', 'optimizer_d_s'} and ck['epoch'] == 1
    assert np.isfinite(ck['monitor_metric']) and {'rmse', 'psnr', 'ssim', 'recon_x_mix', 'all'} <= set(ck['stat'])
    assert float(run.optimizer.step_state[0]) == 2                             # iterations 4 of each epoch: (it + 1) % 4 == 0
    # the next step of the ORIGINAL run (the accumulator of the pending iteration is not part of a checkpoint -- it is not
    # in the reference's either -- so it is cleared on both sides for the comparison)
    x, mask, mask_img = m.synthetic_batch(4, 2, 128, 128, seed=5)
    args = (x.to(dev).contiguous(memory_format=torch.channels_last), mask.to(dev), mask_img.to(dev), mask)

    def four_more(r):
        if r.step.acc is not None:
            r.step.acc.zero_()
        torch.manual_seed(77); np.random.seed(77)
        for it in range(4):
            r.step(*args, it=it)
        return r.optimizer.flat_p.clone(), r.optimizer_d_s.m.clone()
    want_p, want_md = four_more(run)
    cfg2 = m.train.setup_config(str(tmp_path / 'config.yaml'), overrides={'continue_train': True, 'ckpt_name': 'epoch001.pth.tar', 'ckpt_timelabel': os.path.basename(d)}, device=dev)   # :30-31: resume = name the run's time label
    assert cfg2['ckpt_path'] == d
    run2 = m.train.Run(cfg2, loaders=run.loaders, log=logs.append)
    assert run2.start_epoch == 1 and any('optimizer' in str(l) and 'model' in str(l) for l in logs)
    assert run2.scheduler.state_dict()['last_epoch'] == run.scheduler.state_dict()['last_epoch'] == 2
    got_p, got_md = four_more(run2)
    assert torch.equal(want_p, got_p) and torch.equal(want_md, got_md)


@pytest.mark.gpu
def test_entry_point_with_graph_replay_trains_the_same_weights(m, tmp_path):
    """`graph: true` in config.yaml (or MRDIS_GRAPH=1): the entry point's steady-state iterations are HIP-graph replays (trainer.GraphedTrainStep) -- epochs,
    validation passes (model.eval() / train() around them), the accumulation schedule (accum = 4) and the checkpoints in between; weights, Adam moments and
    the per-epoch statistics must equal the eager run's bit for bit."""
    dev = torch.device('cuda:0')
    res = {}
    for graph in (False, True):
        root = tmp_path / ('g' if graph else 'e')
        root.mkdir()
        base = dict(contrast_list=['T1', 'T2', 'T2_FLAIR'], input_height=64, input_width=96, batch_size=8, epochs=2, gpu='0', graph=graph,
                    data_source='synthetic', ckpt_root=str(root / 'ckpt'), ckpt_timelabel='t0', lambda_adv_s=1.0, shuffle=True)
        (root / 'config.yaml').write_text(yaml.dump(base))
        cfg = m.train.setup_config(str(root / 'config.yaml'), device=dev)
        run = m.train.Run(cfg, log=lambda *a: None)
        assert isinstance(run.step, m.GraphedTrainStep) == graph and run.step.accum == 2
        run.train(max_iters_per_epoch=8)
        torch.cuda.synchronize()
        rows = open(os.path.join(cfg['ckpt_path'], 'stat.csv')).read().strip().split('\n')
        res[graph] = (run.optimizer.flat_p.clone(), run.optimizer.m.clone(), run.optimizer_d_s.v.clone(), rows,
                      torch.cat([b.detach().float().reshape(-1) for b in run.model.buffers()]).clone())
        if graph:
            st = run.step.stats
            assert st['replays'] >= 4 and st['eager'] == 4 and st['captures'] == 2 * 6, st          # two configurations (accumulate / step) x 6 ordered adv_s pairs (M = 3)
    for a, b in zip(res[False][:3], res[True][:3]):
        assert torch.equal(a, b)
    assert res[False][3] == res[True][3]
    assert torch.equal(res[False][4], res[True][4])
