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
Data parallel (north_star; BASELINE configs[2] / [3]: the entry point as the 8-GPU job).  Launched as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 main_missing.py ...` (or with the
reserved config key `ddp: true` under such a launcher) every rank runs this file on `cuda:LOCAL_RANK`: one RCCL process group
(`init_distributed`), identical seeded initialisation, the train loader sharded by rank with an identical permutation
(`BatchLoader(rank, world, equal_steps)`), the gradient mean inside `TrainStep` (`GradAllReduce`), the per-epoch loss means and
the validation statistics summed over the ranks BEFORE `scheduler.step` (every rank holds the same monitor, hence the same
lr and the same `is_best`), and `config.*`, `stat.csv`, `epochNNN.pth.tar`, `model_best.pth.tar` written by rank 0 only, with a
barrier behind the writes.  BatchNorm statistics stay per replica (the reference has no SyncBN); rank 0's are checkpointed.
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
import torch.distributed as dist
import yaml

from .data import BatchLoader, SliceDataset, VolumeStore, load_idx_list
from .trainer import (DEFAULT_CONFIG, LOSS_KEYS, EvalStep, TrainStep, make_train_step, build_model, derive_config, load_checkpoint_model,
                      load_config_yaml, save_checkpoint)

SEED = 10                                                                   # main_missing.py:18


# --------------------------------------------------------------------------- data-parallel launch (replaces main_missing.py:28)
def dist_info():
    """(rank, world) of the default process group; (0, 1) when there is none."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_distributed(config=None, backend=None):
    """Join the launcher's process group when there is one to join: WORLD_SIZE > 1 in the environment (torch.distributed.run)
    or `ddp: true` in the config (the key the reference's yaml reserves).  One process per GPU: the device is cuda:LOCAL_RANK
    (returned as the `gpu` override), the group is RCCL with `device_id` set (eager communicator on that device, no
    guessing from the first collective); each rank takes its share of the host cores -- every process defaulting to all
    cores makes N ranks fight while each enqueues ~3,000 launches per step.  Returns {} when single-process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    want = world > 1 or bool((config or {}).get('ddp', False))
    if not want or 'RANK' not in os.environ:
        return {}
    if dist.is_initialized():
        return {'gpu': str(int(os.environ.get('LOCAL_RANK', '0')))}
    rank, local = int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')                # dmabuf IPC only on this host driver (RCCL needs it)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    torch.set_num_threads(max(1, min(8, ncpu // max(1, local_world))))
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        dev = torch.device('cuda', local)
        torch.cuda.set_device(dev)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return {'gpu': str(local)}


def _bcast_object(obj, src=0):
    if dist_info()[1] == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def _barrier():
    if dist_info()[1] > 1:
        dist.barrier()


def _sum_over_ranks(vec, device):
    """float64 sum over the ranks of a small host vector (loss sums, batch counts, metric sums)."""
    rank, world = dist_info()
    t = torch.as_tensor(vec, dtype=torch.float64)
    if world == 1:
        return t
    # RCCL reduces device memory, gloo host memory
    on = device if (dist.get_backend() == 'nccl') else torch.device('cpu')
    t = t.to(on)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu()


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
    rank, world = dist_info()
    if config['ckpt_timelabel'] and (config['phase'] == 'test' or config['continue_train'] is True):
        label = config['ckpt_timelabel']
    else:
        label = _bcast_object(time_label_now())                             # one directory per job: rank 0's clock names it
    config['ckpt_path'] = os.path.join(config.get('ckpt_root', ckpt_root), config['dataset_name'], config['model_name'], label)
    if world > 1:
        # rank 0 creates / merges first; the others then find the directory and take the saved yaml like a resumed run would
        # (writes by rank 0 only)
        if rank == 0 and not os.path.exists(config['ckpt_path']):
            os.makedirs(config['ckpt_path'])
            save_config_yaml(config['ckpt_path'], config)
            fresh = True
        else:
            fresh = False
        fresh = _bcast_object(fresh)
        _barrier()
        if fresh:
            return config
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
        elif rank == 0:
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
    """trainLoader / valLoader / testLoader of ZeroDoseDataAll (util.py:635-708) over volumes resident in HBM.  Under data
    parallelism every rank holds the volumes (a BraTS fold is 10 % of one GPU's HBM) and serves its share of the batches."""
    dev, name = config['device'], config['dataset_name']
    rank, world = dist_info()
    gen = None
    if world > 1:                        # permutation seeds from a generator every rank seeds alike (the default one is per rank, Run.__init__)
        gen = torch.Generator(); gen.manual_seed(SEED)
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
    kw = dict(rank=rank, world=world, generator=gen)
    return {'train': BatchLoader(ds['train'], config['batch_size'], shuffle=config['shuffle'], equal_steps=True, **kw),
            'val': BatchLoader(ds['val'], config['batch_size'], shuffle=False, **kw),
            'test': BatchLoader(ds['test'], config['batch_size'], shuffle=False, **kw)}


# --------------------------------------------------------------------------- the run
class Run:
    """model + optimizers + scheduler + loaders, with train() / evaluate() of main_missing.py."""

    def __init__(self, config, loaders=None, log=print, force_exchange=False, model=None):
        self.config = config
        self.rank, self.world = dist_info()
        self.log = log if self.rank == 0 else (lambda *a, **k: None)
        torch.manual_seed(SEED); np.random.seed(SEED)                       # :18-21 (the CPU generator seeds the init)
        if config['device'].type == 'cuda':
            torch.cuda.manual_seed(SEED)
        if config['model_name'] != 'MultimodalModel' and model is None:
            raise ValueError('not supporting other models yet!')            # :99-100
        self.model = build_model(config) if model is None else model        # `model`: an already built network (tests, fine-tuning scripts)
        if config['fix_pretrain'] and config['continue_train']:             # :104-116
            for part in (self.model.anatomy_encoder_enc_list, self.model.anatomy_encoder_dec, self.model.modality_encoder_list,
                         self.model.input_decoder_list):
                for p in part.parameters():
                    p.requires_grad = False
        self.step = make_train_step(self.model, config, force_exchange=force_exchange)    # joins the default process group when there is one; `graph: true` / MRDIS_GRAPH=1: HIP-graph replays
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
        if config['phase'] == 'train' and self.rank == 0:
            save_config_file(config)                                        # :137-138
        self.loaders = loaders if loaders is not None else make_loaders(config)
        if self.world > 1:
            # identical weights, permutations and (np.random) pair / drop-off draws on every rank; the noise of `sample` per rank
            torch.manual_seed(SEED + 90 + self.rank)
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
                raise RuntimeError(f'epoch {epoch}: the train loader yielded no batch (dataset smaller than one batch'
                                   f'{" per rank" if self.world > 1 else ""}, or an empty subject list)')
            # per-epoch means over ALL ranks' iterations: sums and counts are added over the ranks first
            tot = _sum_over_ranks(torch.cat([acc.double().cpu(), torch.tensor([float(n_iter)], dtype=torch.float64)]), cfg['device'])
            mean = tot[:-1] / tot[-1]
            loss_all = {k: float(mean[i]) for i, k in enumerate(LOSS_KEYS)}
            if not np.isfinite(loss_all['all']):
                raise FloatingPointError(f'epoch {epoch}: loss is {loss_all["all"]} (the reference stops in pdb, :265-266)')
            skipped = self.optimizer.skipped_steps()
            if skipped:
                self.log(f'{skipped} optimizer step(s) skipped so far: non-finite gradients (:273-278)')
            if self.rank == 0:
                save_result_stat(loss_all, cfg, info='epoch[%2d]' % epoch)  # :310-313
            stat = self.evaluate(phase='val', set_='val')                   # :317 (already summed over the ranks: identical everywhere)
            if cfg['lambda_recon_y'] == 0 or cfg['lambda_recon_y_fused'] == 0:
                monitor = stat['recon_x_mix']                               # :318-321
            else:
                monitor = stat['recon_y_fused']
            self.scheduler.step(monitor)                                    # :322 (same monitor on every rank -> same lr)
            if self.rank == 0:
                save_result_stat(stat, cfg, info='val')
            is_best = monitor <= self.monitor_metric_best                   # :326-329
            if is_best:
                self.monitor_metric_best = monitor
            if self.rank == 0:                                              # weights + optimizer state are identical on every rank
                state = {'epoch': epoch, 'monitor_metric': monitor, 'stat': stat, 'optimizer': self.optimizer.state_dict(),
                         'scheduler': self.scheduler.state_dict(), 'model': self.model.state_dict()}    # :330-332
                if cfg['is_discrim_s']:
                    state['optimizer_d_s'] = self.optimizer_d_s.state_dict()
                save_checkpoint(state, is_best, cfg['ckpt_path'])
            _barrier()                                                      # nobody runs ahead of (or exits under) rank 0's writes
            self.log(f'epoch {epoch}: train {loss_all["all"]:.4f}, val monitor {monitor:.4f}, lr {self.optimizer.lr:g}, best {is_best}')
        return self

    # ---- main_missing.py:337-609 (losses + reconstruction metrics; no result dump)
    def evaluate(self, phase='val', set_='val', max_batches=502):
        loader = self.loaders['val'] if phase == 'val' else self.loaders[set_]
        cfg = self.config
        acc, n_iter, met = None, 0, {'rmse': [], 'psnr': [], 'ssim': []}
        # the reference stops after global batch 501 (:562-563).  The cap is applied by the loader, on the GLOBAL batch index and before any meta of a later
        # batch is drawn, so every rank walks the same `max_batches` batches and leaves the host generators in the same state (a cap tested per rank after
        # the step let world = 8 evaluate up to batch 508 and end the ranks' np.random streams apart)
        batches = loader.batches(limit=max_batches) if hasattr(loader, 'batches') else loader
        for it, sample in enumerate(batches):
            if sample.get('batch_index', it) >= max_batches:
                break
            targets = sample['targets'] if cfg['lambda_recon_y'] > 0 else None
            loss, parts, metrics, _ = self.eval_step(sample['inputs'], sample['mask'], sample['mask_img'], sample.get('mask_host'), targets=targets)
            vec = torch.stack([parts[k].float().reshape(()) for k in LOSS_KEYS])
            acc = vec if acc is None else acc + vec
            for k in met:
                met[k].append(metrics[k])
            n_iter += 1
        # sums and counts over the ranks (a rank may hold one batch fewer than its neighbour, or none): the result is the mean over
        # every batch / every image of the loader, identical on all ranks
        nk = len(LOSS_KEYS)
        vec = torch.zeros(nk + 1 + 2 * len(met), dtype=torch.float64)
        if n_iter:
            vec[:nk] = acc.double().cpu(); vec[nk] = n_iter
            for j, k in enumerate(met):
                allv = torch.cat(met[k]).double()
                vec[nk + 1 + 2 * j], vec[nk + 2 + 2 * j] = float(allv.sum()), allv.numel()
        vec = _sum_over_ranks(vec, cfg['device'])
        if vec[nk] == 0:
            raise RuntimeError(f'evaluate({phase!r}, {set_!r}): the loader yielded no batch')
        stat = {k: float(vec[i] / vec[nk]) for i, k in enumerate(LOSS_KEYS)}
        for j, k in enumerate(met):                                         # :568-569
            stat[k] = float(vec[nk + 1 + 2 * j] / vec[nk + 2 + 2 * j]) if vec[nk + 2 + 2 * j] > 0 else float('nan')
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
    overrides = parse_overrides(argv)
    found, filecfg = load_config_yaml(path)
    overrides.update(init_distributed({**filecfg, **overrides}))            # data-parallel launch: cuda:LOCAL_RANK, one RCCL group
    config = setup_config(path, overrides)
    run = Run(config)
    try:
        if config['phase'] == 'train':                                      # :611-614
            run.train()
        else:
            stat = run.evaluate(phase='test', set_='test')
            if run.rank == 0:
                print(stat)
    finally:
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    return run
