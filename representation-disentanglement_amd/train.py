"""Runnable entry of the hot path: the reference's `src/main_missing.py` (a module-level script) restated as
functions around `TrainStep` / `EvalStep`.

    python main_missing.py [config.yaml] [key=value ...]          # repo-root launcher, same name as the reference's script

What is kept from the reference, statement by statement:
  * seeds (main_missing.py:17-22), config.yaml keys (src/config.yaml), derived keys (:26-28, :75-86)
  * checkpoint directory `<ckpt_root>/<dataset_name>/<model_name>/<time label>` and the saved-yaml merge rule
    (:30-58; keys `phase` / `continue_train` always come from the current file)
  * Adam(lr, wd 1e-5, amsgrad) + ReduceLROnPlateau(min, 0.1, patience 5, min_lr 1e-5) (:118-119), optimizer_d_s (:121-122)
  * continue_train / test: `load_checkpoint_by_key` for optimizer, scheduler, model[, optimizer_d_s] (:125-135,
    util.py:870-892), name + shape filtered model load (util.py:895-903)
  * epoch loop (:141-335): per-epoch loss means -> stat.csv row 'epoch[ k]', validation pass -> row 'val',
    `scheduler.step(monitor)` with monitor = val recon_x_mix (or recon_y_fused), `epochNNN.pth.tar` every epoch and
    `model_best.pth.tar` when the monitor improves (util.py:148-153, 854-866)
  * evaluate() (:337-609): eval-mode pass over a loader, loss means + mean MSE / PSNR / SSIM of the mix
    reconstructions, capped at 502 batches (:562-563).  Result dumping to h5 (:565-606) is out of scope.
What differs, deliberately: no per-iteration `.item()` syncs (device accumulators, one D2H copy per epoch);
`data_source: synthetic` (added key) builds BraTS-shaped volumes in HBM instead of opening the private h5 file;
`16 // batch_size` is guarded for batch_size > 16 (:282 divides by zero there).
"""
import copy
import os
import sys
import time

import numpy as np
import torch
import yaml

from .data import BatchLoader, SliceDataset, VolumeStore, load_idx_list
from .trainer import (DEFAULT_CONFIG, LOSS_KEYS, EvalStep, TrainStep, build_model, derive_config, load_checkpoint_model,
                      load_config_yaml, save_checkpoint)

SEED = 10                                                                   # main_missing.py:18


# --------------------------------------------------------------------------- config plumbing (main_missing.py:25-58)
def save_config_yaml(ckpt_path, config):
    """util.py:913-925: only int / float / str / list / dict values are written."""
    keep = {k: v for k, v in config.items() if isinstance(v, (int, float, str, list, dict))}
    with open(os.path.join(ckpt_path, 'config.yaml'), 'w') as f:
        yaml.dump(copy.deepcopy(keep), f)


def save_config_file(config):
    """util.py:846-851."""
    with open(os.path.join(config['ckpt_path'], 'config.txt'), 'w') as f:
        for k, v in config.items():
            f.write(f'{k}: {v}\n')


def time_label_now():
    t = time.localtime(time.time())                                          # :33-34
    return f'{t.tm_year}_{t.tm_mon}_{t.tm_mday}_{t.tm_hour}_{t.tm_min}'


def setup_config(config_path='config.yaml', overrides=None, ckpt_root='../ckpt/', device=None):
    """main_missing.py:25-58: load, derive, name the checkpoint directory, merge a saved yaml."""
    found, config = load_config_yaml(config_path)
    if not found and config_path != 'config.yaml':
        raise FileNotFoundError(config_path)
    config.update(overrides or {})
    if device is None:
        device = torch.device('cuda:' + str(config['gpu']))
    config = derive_config(config, device)
    if config['ckpt_timelabel'] and (config['phase'] == 'test' or config['continue_train'] is True):
        label = config['ckpt_timelabel']
    else:
        label = time_label_now()
    config['ckpt_path'] = os.path.join(config.get('ckpt_root', ckpt_root), config['dataset_name'], config['model_name'], label)
    if not os.path.exists(config['ckpt_path']):
        os.makedirs(config['ckpt_path'])
        save_config_yaml(config['ckpt_path'], config)
    elif config['load_yaml']:
        flag, saved = load_config_yaml(os.path.join(config['ckpt_path'], 'config.yaml'))
        if flag:
            for k, v in saved.items():                                      # :45-51
                if k in ('phase', 'continue_train') or k not in config:
                    continue
                config[k] = v
            config = derive_config(config, device)
        else:
            save_config_yaml(config['ckpt_path'], config)
    return config


# --------------------------------------------------------------------------- stat.csv (util.py:854-866)
def save_result_stat(stat, config, info='Default'):
    import pandas as pd
    stat = dict(stat)
    path = os.path.join(config['ckpt_path'], 'stat.csv')
    columns = ['info'] + sorted(stat.keys())
    if not os.path.exists(path):
        pd.DataFrame(columns=columns).to_csv(path, mode='a', header=True)
    stat['info'] = info
    df = pd.DataFrame.from_dict({k: [v] for k, v in stat.items()})[columns]
    df.to_csv(path, mode='a', header=False)


def load_checkpoint_by_key(values, checkpoint_dir, keys, device, ckpt_name='model_best.pth.tar'):
    """util.py:870-892.  `values[i].load_state_dict(checkpoint[key])`, the model through the name + shape filter; a key
    that fails to load is reported and skipped, as in the reference."""
    filename = os.path.join(checkpoint_dir, ckpt_name)
    if not os.path.isfile(filename):
        raise ValueError('No correct checkpoint')
    checkpoint = torch.load(filename, map_location=device, weights_only=False)
    loaded = []
    for i, key in enumerate(keys):
        try:
            if key == 'model':
                load_checkpoint_model(values[i], checkpoint[key])
            else:
                values[i].load_state_dict(checkpoint[key])
            loaded.append(key)
        except Exception as e:                                              # noqa: BLE001  (reference: bare except + print)
            print(f'loading {key} failed! ({type(e).__name__}: {e})')
    return values, checkpoint['epoch'], loaded


# --------------------------------------------------------------------------- data (main_missing.py:62-68)
FOLD_FILES = {                                                              # util.py:643-694
    'BraTS': ('BraTS_All.h5', 'BraTS_All_zscore_10.h5', 'fold_BraTS_{f}_{s}_noval.txt'),
    'NCANDA': ('NCANDA_All.h5', 'NCANDA_All_zscore_10.h5', 'fold_NCANDA_{f}_{s}.txt'),
    'Tau': (None, 'Tau_All_zscore.h5', 'fold_Tau_{f}_{s}.txt'),
}


def synthetic_store(config, device, n_subj=4, depth=24, seed=21):
    """BraTS-shaped volumes (z-scored inside an ellipsoid, background -10, a label volume) when no h5 file is around
    (`data_source: synthetic`): every slice list entry is valid for block_size 3."""
    H, W = config['input_height'], config['input_width']
    g = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    inside = (((yy - H / 2 + 0.5) / (0.40 * H)) ** 2 + ((xx - W / 2 + 0.5) / (0.42 * W)) ** 2) <= 1
    arrays, subj, idx = {}, [], []
    for s in range(n_subj):
        name = f'Synth_{s:03d}'
        for c in config['contrast_list']:
            v = g.randn(H, W, depth).astype(np.float32)
            arrays[f'{name}/{c}'] = np.where(inside[:, :, None], v, np.float32(-10.0))
        arrays[f'{name}/seg'] = (g.randint(0, 5, (H, W, depth)) * inside[:, :, None]).astype(np.float32)
        for k in range(config['block_size'], depth - config['block_size'] - 1):
            subj.append(name); idx.append(k)
    return VolumeStore.from_arrays(arrays, device), np.array(subj), np.array(idx)


def make_loaders(config):
    """trainLoader / valLoader / testLoader of ZeroDoseDataAll (util.py:635-708) over volumes resident in HBM."""
    dev, name = config['device'], config['dataset_name']
    kw = dict(block_size=config['block_size'], contrast_list=config['contrast_list'])
    if config.get('data_source', 'h5') == 'synthetic':
        store, subj, idx = synthetic_store(config, dev)
        n = len(subj); a, b = int(0.6 * n), int(0.8 * n)
        split = {'train': (subj[:a], idx[:a]), 'val': (subj[a:b], idx[a:b]), 'test': (subj[b:], idx[b:])}
    else:
        if name not in FOLD_FILES:
            raise NotImplementedError(f'dataset {name}: only BraTS / NCANDA / Tau file naming is restated (util.py:643-694)')
        mean_h5, z_h5, pat = FOLD_FILES[name]
        h5 = mean_h5 if config['norm_type'] == 'mean' else z_h5
        if h5 is None:
            raise ValueError('Need preprocessing data!')                   # util.py:686
        store = VolumeStore.from_h5(os.path.join(config['data_path'], h5), dev)
        split = {s: load_idx_list(os.path.join(config['data_path'], pat.format(f=config['fold'], s=s))) for s in ('train', 'val', 'test')}
    ds = {s: SliceDataset(name, store, *split[s], dropoff=(config['dropoff'] and s != 'test'), **kw) for s in split}
    return {'train': BatchLoader(ds['train'], config['batch_size'], shuffle=config['shuffle']),
            'val': BatchLoader(ds['val'], config['batch_size'], shuffle=False),
            'test': BatchLoader(ds['test'], config['batch_size'], shuffle=False)}


# --------------------------------------------------------------------------- the run
class Run:
    """model + optimizers + scheduler + loaders, with train() / evaluate() of main_missing.py."""

    def __init__(self, config, loaders=None, log=print):
        self.config, self.log = config, log
        torch.manual_seed(SEED); np.random.seed(SEED)                       # :18-21 (the CPU generator seeds the init)
        if config['device'].type == 'cuda':
            torch.cuda.manual_seed(SEED)
        if config['model_name'] != 'MultimodalModel':
            raise ValueError('not supporting other models yet!')            # :99-100
        self.model = build_model(config)
        if config['fix_pretrain'] and config['continue_train']:             # :104-116
            for part in (self.model.anatomy_encoder_enc_list, self.model.anatomy_encoder_dec, self.model.modality_encoder_list,
                         self.model.input_decoder_list):
                for p in part.parameters():
                    p.requires_grad = False
        self.step = TrainStep(self.model, config)
        self.eval_step = EvalStep(self.model, config)
        self.optimizer, self.optimizer_d_s = self.step.optimizer, self.step.optimizer_d_s
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode='min', factor=0.1, patience=5, min_lr=1e-5)   # :119
        self.start_epoch = -1
        if config['continue_train'] or config['phase'] == 'test':           # :125-135
            _, self.start_epoch, loaded = load_checkpoint_by_key([self.optimizer, self.scheduler, self.model], config['ckpt_path'],
                                                                 ['optimizer', 'scheduler', 'model'], config['device'], config['ckpt_name'])
            log(f'loaded {loaded} from {config["ckpt_name"]} (epoch {self.start_epoch})')
            if self.optimizer_d_s is not None:
                try:
                    load_checkpoint_by_key([self.optimizer_d_s], config['ckpt_path'], ['optimizer_d_s'], config['device'], config['ckpt_name'])
                except Exception:                                           # noqa: BLE001
                    log('Pretrained model does not have discriminator')
        if config['phase'] == 'train':
            save_config_file(config)                                        # :137-138
        self.loaders = loaders if loaders is not None else make_loaders(config)
        self.global_iter = 0
        self.monitor_metric_best = 100                                      # :143

    # ---- main_missing.py:141-335
    def train(self, max_iters_per_epoch=None):
        cfg = self.config
        for epoch in range(self.start_epoch + 1, cfg['epochs']):
            self.model.train()
            acc, n_iter = None, 0
            for it, sample in enumerate(self.loaders['train']):
                self.global_iter += 1
                targets = sample['targets'] if cfg['lambda_recon_y'] > 0 else None
                loss, parts, _ = self.step(sample['inputs'], sample['mask'], sample['mask_img'], sample.get('mask_host'), targets=targets, it=it)
                vec = torch.stack([parts[k].float().reshape(()) for k in LOSS_KEYS])
                acc = vec if acc is None else acc + vec                     # :253-263 without the 11 .item() syncs
                n_iter += 1
                if self.global_iter % 10 == 0:                              # :291-293 (one D2H copy per 10 iterations)
                    v = vec.cpu()
                    self.log('Epoch[%3d], iter[%3d]: ' % (epoch, it) + ', '.join(f'{k}=[{float(v[i]):.4f}]' for i, k in enumerate(LOSS_KEYS)))
                if max_iters_per_epoch is not None and n_iter >= max_iters_per_epoch:
                    break
            if n_iter == 0:
                raise RuntimeError(f'epoch {epoch}: the train loader yielded no batch (dataset smaller than one batch, or an empty subject list)')
            mean = (acc / n_iter).cpu()
            loss_all = {k: float(mean[i]) for i, k in enumerate(LOSS_KEYS)}
            if not np.isfinite(loss_all['all']):
                raise FloatingPointError(f'epoch {epoch}: loss is {loss_all["all"]} (the reference stops in pdb, :265-266)')
            skipped = self.optimizer.skipped_steps()
            if skipped:
                self.log(f'{skipped} optimizer step(s) skipped so far: non-finite gradients (:273-278)')
            save_result_stat(loss_all, cfg, info='epoch[%2d]' % epoch)      # :310-313
            stat = self.evaluate(phase='val', set_='val')                   # :317
            if cfg['lambda_recon_y'] == 0 or cfg['lambda_recon_y_fused'] == 0:
                monitor = stat['recon_x_mix']                               # :318-321
            else:
                monitor = stat['recon_y_fused']
            self.scheduler.step(monitor)                                    # :322
            save_result_stat(stat, cfg, info='val')
            is_best = monitor <= self.monitor_metric_best                   # :326-329
            if is_best:
                self.monitor_metric_best = monitor
            state = {'epoch': epoch, 'monitor_metric': monitor, 'stat': stat, 'optimizer': self.optimizer.state_dict(),
                     'scheduler': self.scheduler.state_dict(), 'model': self.model.state_dict()}        # :330-332
            if cfg['is_discrim_s']:
                state['optimizer_d_s'] = self.optimizer_d_s.state_dict()
            save_checkpoint(state, is_best, cfg['ckpt_path'])
            self.log(f'epoch {epoch}: train {loss_all["all"]:.4f}, val monitor {monitor:.4f}, lr {self.optimizer.lr:g}, best {is_best}')
        return self

    # ---- main_missing.py:337-609 (losses + reconstruction metrics; no result dump)
    def evaluate(self, phase='val', set_='val', max_batches=502):
        loader = self.loaders['val'] if phase == 'val' else self.loaders[set_]
        cfg = self.config
        acc, n_iter, met = None, 0, {'rmse': [], 'psnr': [], 'ssim': []}
        for it, sample in enumerate(loader):
            targets = sample['targets'] if cfg['lambda_recon_y'] > 0 else None
            loss, parts, metrics, _ = self.eval_step(sample['inputs'], sample['mask'], sample['mask_img'], sample.get('mask_host'), targets=targets)
            vec = torch.stack([parts[k].float().reshape(()) for k in LOSS_KEYS])
            acc = vec if acc is None else acc + vec
            for k in met:
                met[k].append(metrics[k])
            n_iter += 1
            if it > 500:                                                    # :562-563
                break
            if n_iter >= max_batches:
                break
        if n_iter == 0:
            raise RuntimeError(f'evaluate({phase!r}, {set_!r}): the loader yielded no batch')
        mean = (acc / n_iter).cpu()
        stat = {k: float(mean[i]) for i, k in enumerate(LOSS_KEYS)}
        for k, v in met.items():                                            # :568-569
            stat[k] = float(torch.cat(v).double().mean()) if v else float('nan')
        return stat


def parse_overrides(args):
    out = {}
    for a in args:
        k, _, v = a.partition('=')
        out[k.lstrip('-')] = yaml.safe_load(v)
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    path = 'config.yaml'
    if argv and '=' not in argv[0]:
        path = argv.pop(0)
    config = setup_config(path, parse_overrides(argv))
    run = Run(config)
    if config['phase'] == 'train':                                          # :611-614
        run.train()
    else:
        print(run.evaluate(phase='test', set_='test'))
    return run
