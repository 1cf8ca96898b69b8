"""Training step of the reference (`src/main_missing.py:141-335`) restated for the
HIP path: same call order, same loss weights / config keys, same checkpoint
dict layout -- minus the ~355 host syncs per iteration.

  * parameters that receive gradients live in ONE flat fp32 arena (weights,
    grads, Adam m / v / vmax), so clip + finite check + Adam(amsgrad, L2 wd) is
    two kernel launches and the data-parallel exchange is an all-reduce over
    contiguous memory (no flatten/unflatten copies);
  * the 11 loss scalars leave the device in one D2H copy, when asked for.
"""
import os
import shutil

import numpy as np
import torch
import torch.distributed as dist
import yaml

from . import hip, ops
from .model import MultimodalModel, flush_batch_counters, discard_batch_counters

# config.yaml of the reference, verbatim keys and defaults (src/config.yaml:1-91)
DEFAULT_CONFIG = {
    'phase': 'train', 'load_yaml': True, 'epochs': 50, 'gpu': '0', 'dataset_name': 'BraTS',
    'contrast_list': ['T1', 'T1c', 'T2', 'T2_FLAIR'], 'norm_type': 'z-score', 'block_size': 3,
    'data_path': '../data/', 'batch_size': 8, 'num_fold': 5, 'fold': 0, 'shuffle': True, 'lr': 0.0002,
    'model_name': 'MultimodalModel', 'p': 1, 's_num_ch': 4, 'z_size': 16,
    'lambda_recon_y': 0., 'lambda_recon_y_fused': 0., 'lambda_recon_x': 1.0, 'lambda_recon_x_mix': 2.0,
    'lambda_sim_s': 10.0, 'lambda_sim_z': 2.0, 's_compact_method': 'max', 's_sim_method': 'cosine',
    'z_sim_method': 'cosine', 'lambda_kl': 0., 'lambda_latent_z': 0.1, 'lambda_adv_s': 0.,
    'is_cond': True, 'is_distri_z': False, 'shared_ana_enc': True, 'shared_mod_enc': True, 'shared_inp_dec': False,
    'others': {'mod_enc_s': False, 'ana_dec_act': 'softmax', 'old': False, 'softmax_remove_mask': True},
    'out_num_ch': 1, 'input_height': 160, 'input_width': 192, 'dropoff': False, 'skull_strip': False,
    'fuse_method': 'mean', 'target_model_name': 'U+SA', 'continue_train': False, 'fix_pretrain': False,
    'ckpt_name': 'model_best.pth.tar', 'ckpt_timelabel': None,
    # keys added by this implementation (defaults reproduce the reference)
    'backend': 'hip', 'is_patch_gan': False,
    'graph': None,                     # true: steady-state iterations as HIP-graph replays (trainer.GraphedTrainStep); None: MRDIS_GRAPH decides (default off)
    'compute_dtype': 'f32',            # BASELINE configs[2]: 'bf16' = bf16 activations + bf16 MFMA operands + fp32 accumulate; 'bf16m' = bf16 MFMA operands only
}


def load_config_yaml(path):
    """util.py:905-915: (found, dict)."""
    if os.path.exists(path):
        with open(path) as f:
            cfg = dict(DEFAULT_CONFIG)
            cfg.update(yaml.safe_load(f) or {})
            return True, cfg
    return False, dict(DEFAULT_CONFIG)


def derive_config(config, device):
    """main_missing.py:26-28, 75-86."""
    config = dict(config)
    config['is_discrim_s'] = config['lambda_adv_s'] > 0
    config['in_num_ch'] = len(config['contrast_list']) * (2 * config['block_size'] + 1)
    config['device'] = device
    config['target_output_act'] = 'no' if (config['dataset_name'] == 'BraTS' or config['norm_type'] == 'z-score') else 'softplus'
    config['input_output_act'] = 'softplus' if config['norm_type'] == 'mean' else 'no'
    return config


def build_model(config):
    """main_missing.py:87-95."""
    ops.set_compute_dtype(config.get('compute_dtype', 'f32'))
    return MultimodalModel(
        input_size=(config['input_height'], config['input_width']), modality_num=len(config['contrast_list']),
        in_num_ch=2 * config['block_size'] + 1, out_num_ch=config['out_num_ch'], s_num_ch=config['s_num_ch'],
        z_size=config['z_size'], is_cond=config['is_cond'], is_discrim_s=config['is_discrim_s'],
        is_distri_z=config['is_distri_z'], s_compact_method=config['s_compact_method'],
        s_sim_method=config['s_sim_method'], z_sim_method=config['z_sim_method'],
        shared_ana_enc=config['shared_ana_enc'], shared_mod_enc=config['shared_mod_enc'],
        shared_inp_dec=config['shared_inp_dec'], device=config['device'],
        input_output_act=config['input_output_act'], target_output_act=config['target_output_act'],
        target_model_name=config['target_model_name'], fuse_method=config['fuse_method'], others=config['others'],
        is_patch_gan=config.get('is_patch_gan', False),
        build_output_decoder=config['lambda_recon_y'] > 0 or config['lambda_recon_y_fused'] > 0)


# --------------------------------------------------------------------------- synthetic BraTS-shaped data
def synthetic_batch(B, M, H, W, seed, drop=False, block=7):
    """Loader contract of util.py:508-566 on synthetic slices: z-scored
    intensities inside a centred ellipse, background -10 (data_preprocessing_BraTS.py:85-95),
    `block` = 2*block_size+1 neighbouring slices per modality; optional modality drop-off
    (util.py:538-542): one random modality per sample zeroed and masked out.
    Returns CPU tensors inputs (B, block*M, H, W), mask (B, M), mask_img (B, H, W)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, block * M, H, W, generator=g)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    inside = (((yy - H / 2 + 0.5) / (0.40 * H)) ** 2 + ((xx - W / 2 + 0.5) / (0.42 * W)) ** 2) <= 1
    x = torch.where(inside[None, None], x, torch.full_like(x, -10.0))
    mask = torch.ones(B, M)
    if drop:
        for b in range(B):
            d = int(torch.randint(0, M, (1,), generator=g))
            mask[b, d] = 0
            x[b, block * d:block * (d + 1)] = 0
    mask_img = (x[:, 0] == 0).float()
    return x, mask, mask_img


def fit_to_model(x, size, fill=-10.0):
    """240x240 BraTS slices do not fit five stride-2 stages (model.py:2192).  Either centre-crop
    to the reference's 160x192 (data_preprocessing_BraTS.py:85) or pad with background to the
    next multiple of 32 (256x256); `size` selects which."""
    H, W = x.shape[-2:]
    th, tw = size
    if th <= H and tw <= W:
        t, l = (H - th) // 2, (W - tw) // 2
        return x[..., t:t + th, l:l + tw].contiguous()
    out = torch.full(x.shape[:-2] + (th, tw), fill, dtype=x.dtype)
    t, l = (th - H) // 2, (tw - W) // 2
    out[..., t:t + H, l:l + W] = x
    return out


# --------------------------------------------------------------------------- flat parameter arena + Adam
class ArenaAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, weight_decay=wd, amsgrad=True) (main_missing.py:118) with
    clip_grad_norm_(1.0) (:272) and the finite check (:273-278) folded into the step.

    The parameters that can receive a gradient live in ONE flat fp32 arena (weights, grads, Adam m / v / vmax);
    `p.data` / `p.grad` are views.  Membership is STATIC when `used` is given (TrainStep passes the model's
    `trainable_parameters()`: 43 % of the reference's parameters never get a gradient; torch's Adam skips
    those, so do we) and identical on every data-parallel rank; without `used` it is taken from the first backward
    (generic use).  A torch.optim.Optimizer subclass, so ReduceLROnPlateau (main_missing.py:119) attaches to it and
    `state_dict()` / `load_state_dict()` speak torch.optim.Adam's checkpoint format (:126, :330-335).

    * `step_state` (device float[2]): optimizer steps applied / skipped as non-finite.  The bias correction reads the
      device counter, so a skipped step does not advance it; `skipped_steps()` exposes the count (one D2H copy).
    * gates: groups of parameters that receive no gradient when their modality is absent from the whole batch
      (`set_gates`); the per-group activity flags ride at the tail of the gradient buffer (so a data-parallel sum
      all-reduce ORs them across ranks) and the Adam kernel leaves a gated-off range untouched, as torch's Adam does
      for `grad is None`.  torch keeps `step` per parameter and does not advance it for a skipped parameter: each gate
      group has its own device step counter (`gate_steps`), used for the bias correction of its ranges and exported /
      imported as the per-parameter `step` of the checkpoint.
    """

    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, max_norm=1.0, used=None,
                 share_weights_of=None, order=None):
        """order: parameter lists in the order their gradients complete during backward (MultimodalModel.completion_groups);
        the arena is laid out in that order and `group_edges` holds the boundaries (GradAllReduce cuts its buckets there)."""
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=True, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False)
        super().__init__(list(params), defaults)
        self.params = [p for p in self.param_groups[0]['params']]
        self.max_norm = max_norm
        self.step_count = 0                     # host-side count of step() calls (the device counter is authoritative)
        self.flat_p = self.flat_g = self.m = self.v = self.vmax = None
        self.used = None
        self.norm_finite = self.step_state = None
        self.gate_ranges, self.gate_flag_index, self.gate_flags, self.n_flags = [], [], None, 0
        self.gate_steps, self.gate_of = None, {}
        self._share = share_weights_of
        self._order = order
        self.group_edges = None
        if share_weights_of is not None:
            self._build(share_weights_of.used)
        elif used is not None:
            self._build(list(used))

    # convenience views of the single param group (the scheduler writes group['lr'])
    @property
    def lr(self):
        return self.param_groups[0]['lr']

    @property
    def betas(self):
        return self.param_groups[0]['betas']

    @property
    def eps(self):
        return self.param_groups[0]['eps']

    @property
    def wd(self):
        return self.param_groups[0]['weight_decay']

    TAIL = 32                                   # floats reserved behind the gradients for the gate flags

    def _build(self, used=None):
        if used is None:
            used = [p for p in self.params if p.requires_grad and p.grad is not None]
            if not used:
                raise RuntimeError('ArenaAdam.step() before any backward()')
        used = [p for p in used if p.requires_grad]
        rank_of = {}
        if self._order is not None:
            # completion order of the backward pass: group k's gradients are complete before group k + 1's, so the arena is
            # laid out group by group and a bucket of the data-parallel exchange never waits for a later group
            for k, grp in enumerate(self._order):
                for p in grp:
                    rank_of.setdefault(id(p), k)
        # within a group (or without an order): parameters whose gradient the backward kernels add to in place (ops._grad_sink)
        # fire no post-accumulate hook; keep them behind the hook-driven ones (stable sort: constructor order otherwise)
        ngrp = len(self._order) if self._order is not None else 0
        share = self._share
        if share is not None:
            used = list(share.used)             # the layout IS the other optimizer's: same tensors at the same offsets
        else:
            used.sort(key=lambda p: (rank_of.get(id(p), ngrp), 1 if getattr(p, '_mrdis_sink', False) else 0))
        dev = used[0].device
        if share is not None:                   # second optimizer over the same weights (optimizer_d_s, main_missing.py:121-122)
            self.offsets, self.numel, self.flat_p = share.offsets, share.numel, share.flat_p
            self.group_edges = share.group_edges
            n = self.numel
        else:
            # 16-byte align every tensor inside the arena (vectorised kernels read params in place)
            offs, n = [], 0
            for p in used:
                offs.append(n); n += (p.numel() + 3) // 4 * 4
            self.offsets, self.numel = offs, n
            if self._order is not None:
                edges, prev = [0], rank_of.get(id(used[0]), ngrp)
                for p, o in zip(used, offs):
                    r = rank_of.get(id(p), ngrp)
                    if r != prev:
                        edges.append(o); prev = r
                edges.append(n)
                self.group_edges = edges
            self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self._g_full = torch.zeros(n + self.TAIL, dtype=torch.float32, device=dev)
        self.flat_g = self._g_full[:n]
        self.grad_views = []
        with torch.no_grad():
            for p, o in zip(used, self.offsets):
                k = p.numel()
                gv = self.flat_g[o:o + k].view(p.shape)
                if share is None:
                    self.flat_p[o:o + k].copy_(p.data.reshape(-1))
                    if p.grad is not None:
                        gv.copy_(p.grad)
                    p.data = self.flat_p[o:o + k].view(p.shape)
                    p.grad = gv
                self.grad_views.append(gv)
        self.m = torch.zeros_like(self.flat_p); self.v = torch.zeros_like(self.flat_p); self.vmax = torch.zeros_like(self.flat_p)
        self.used = used
        self.used_ids = {id(p) for p in used}
        self.norm_finite = torch.zeros(2, dtype=torch.float32, device=dev)
        self.step_state = torch.zeros(2, dtype=torch.float32, device=dev)

    def attach_grads(self):
        """make `p.grad` of every arena parameter a view of THIS optimizer's gradient buffer (two optimizers over one
        set of weights take turns: generator loss, then discriminator loss)."""
        for p, gv in zip(self.used, self.grad_views):
            p.grad = gv

    def set_gates(self, groups):
        """groups: list of parameter lists; group k's parameters are stepped only while flag k is non-zero."""
        off = {id(p): (o, p.numel()) for p, o in zip(self.used, self.offsets)}
        ranges, fidx = [], []
        for k, group in enumerate(groups):
            spans = sorted(off[id(p)] for p in group if id(p) in off)
            cur = None
            for o, n in spans:                  # merge neighbours (alignment padding between them belongs to nobody)
                if cur is not None and o <= cur[1] + 3:
                    cur[1] = o + n
                else:
                    if cur is not None:
                        ranges.append(tuple(cur)); fidx.append(k)
                    cur = [o, o + n]
            if cur is not None:
                ranges.append(tuple(cur)); fidx.append(k)
        if len(ranges) > 32 or len(groups) > self.TAIL:
            raise NotImplementedError('more gated segments than the Adam kernel takes (32)')
        self.gate_ranges, self.gate_flag_index, self.n_flags = ranges, fidx, len(groups)
        self.gate_flags = self._g_full[self.numel:self.numel + self.n_flags]
        # [k] = steps applied to group k; the rest is the kernel's scratch for the bias-correction pairs (include/mrdis.h)
        self.gate_steps = torch.zeros(3 * self.n_flags, dtype=torch.float32, device=self._g_full.device) if ranges else None
        self.gate_of = {id(p): k for k, group in enumerate(groups) for p in group if id(p) in off}

    def mark_active(self, flags):
        """flags: (n_groups,) device tensor, > 0 where the group receives a gradient from this (micro-)batch."""
        if self.n_flags:
            self.gate_flags.add_(flags.to(self.gate_flags.dtype))

    def _gates(self):
        return (self.gate_ranges, self.gate_flag_index, self.gate_flags) if self.gate_ranges else None

    def grad_norm_sq(self, g=None):
        """device tensor [sum g^2, #non-finite] over the arena."""
        self.norm_finite.zero_()
        hip.sumsq_finite(self.flat_g if g is None else g, self.norm_finite)
        return self.norm_finite

    def clip_in_place(self, g=None):
        """clip_grad_norm_ semantics on the accumulated gradient (used between micro-batches when
        accumulating, main_missing.py:272 runs every iteration).  Leaves [sum g^2, #non-finite] of the UNCLIPPED
        gradient in `norm_finite`, which then gates the Adam step (non-finite => skipped)."""
        g = self.flat_g if g is None else g
        nf = self.grad_norm_sq(g)
        coef = torch.clamp(self.max_norm / (torch.sqrt(nf[0]) + 1e-6), max=1.0)
        g.mul_(coef)
        return nf

    def check_new_grads(self):
        for p in self.params:
            if p.grad is not None and id(p) not in self.used_ids:
                raise RuntimeError('a parameter outside the arena received a gradient: the static trainable-parameter list '
                                   'of the model is out of date (MultimodalModel.trainable_parameters)')

    def step(self, fused_clip=True, grad_scale=1.0, g=None, gate_only=False, use_gates=False):
        """fused_clip: norm + clip + finite gate inside the step (accum == 1).  gate_only: `norm_finite` already holds the
        pair of the (clipped-in-place) gradient: only the non-finite gate is applied.  use_gates: honour the activity flags
        of `set_gates` (the caller marked this step's active groups with `mark_active`)."""
        if self.used is None:
            self._build()
        self.step_count += 1
        g = self.flat_g if g is None else g
        nf = None
        if fused_clip:
            nf = self.grad_norm_sq(g)
        elif gate_only:
            nf = self.norm_finite
        hip.adam_amsgrad_step(self.flat_p, g, self.m, self.v, self.vmax, self.lr, self.betas[0], self.betas[1],
                              self.eps, self.wd, self.step_count, nf, self.max_norm if fused_clip else 0.0, grad_scale,
                              step_state=self.step_state, gates=self._gates() if use_gates else None,
                              gate_steps=self.gate_steps if use_gates else None)

    def zero_grad(self, set_to_none=True):
        if self.used is None:
            for p in self.params:
                p.grad = None
        else:
            self._g_full.zero_()

    def skipped_steps(self):
        """number of optimizer steps the device skipped because the gradient was non-finite (one D2H copy)."""
        return 0 if self.step_state is None else int(self.step_state[1].item())

    # torch.optim.Adam-compatible checkpoint payload (main_missing.py:330-335 stores optimizer.state_dict())
    def state_dict(self):
        group = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        group['params'] = list(range(len(self.params)))
        state = {}
        if self.used is not None:
            applied = float(self.step_state[0].item())
            gsteps = self.gate_steps[:self.n_flags].tolist() if self.gate_steps is not None else []
            if applied > 0:
                idx = {id(p): i for i, p in enumerate(self.params)}
                for p, o in zip(self.used, self.offsets):
                    k = p.numel()
                    gk = self.gate_of.get(id(p))
                    steps = applied if gk is None else gsteps[gk]
                    if steps <= 0:              # a gated parameter that never received a gradient: torch has no state entry for it
                        continue
                    state[idx[id(p)]] = {'step': torch.tensor(steps, dtype=torch.float32),
                                         'exp_avg': self.m[o:o + k].view(p.shape).clone(),
                                         'exp_avg_sq': self.v[o:o + k].view(p.shape).clone(),
                                         'max_exp_avg_sq': self.vmax[o:o + k].view(p.shape).clone()}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, state_dict):
        """accepts what torch.optim.Adam(model.parameters(), amsgrad=True).state_dict() / our own state_dict() wrote.
        The payload is validated in full BEFORE the arena is touched: a bad checkpoint raises and leaves moments, step
        counters and hyper-parameters as they were."""
        groups = state_dict['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self.params):
            raise ValueError('optimizer state_dict does not match: expected one param group over model.parameters()')
        if self.used is None:
            if state_dict['state']:
                raise RuntimeError('load_state_dict() needs a built arena (construct with used=...)')
            for k, v in groups[0].items():
                if k != 'params':
                    self.param_groups[0][k] = v
            return
        off = {id(p): (o, p.numel()) for p, o in zip(self.used, self.offsets)}
        plan = []                                # (offset, numel, entry, gate group or None)
        applied, gsteps = 0.0, [0.0] * self.n_flags
        for i, st in state_dict['state'].items():
            if not 0 <= int(i) < len(self.params):
                raise ValueError(f'optimizer state for parameter #{i}: index out of range')
            p = self.params[int(i)]
            if id(p) not in off:
                raise ValueError(f'optimizer state for parameter #{i}, which is outside the arena')
            o, k = off[id(p)]
            for key in ('step', 'exp_avg', 'exp_avg_sq'):
                if key not in st:
                    raise ValueError(f'optimizer state for parameter #{i} lacks {key!r}')
            for key in ('exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'):
                if key in st and st[key].numel() != k:
                    raise ValueError(f'optimizer state for parameter #{i}: {key} has {st[key].numel()} elements, the parameter {k}')
            gk = self.gate_of.get(id(p))
            s = float(st['step'])
            if gk is None:
                applied = max(applied, s)
            else:
                gsteps[gk] = max(gsteps[gk], s)
            plan.append((o, k, st))
        # commit
        for k, v in groups[0].items():
            if k != 'params':
                self.param_groups[0][k] = v
        self.m.zero_(); self.v.zero_(); self.vmax.zero_()
        with torch.no_grad():
            for o, k, st in plan:
                self.m[o:o + k].copy_(st['exp_avg'].reshape(-1)); self.v[o:o + k].copy_(st['exp_avg_sq'].reshape(-1))
                if 'max_exp_avg_sq' in st:
                    self.vmax[o:o + k].copy_(st['max_exp_avg_sq'].reshape(-1))
        applied = max([applied] + gsteps)        # the arena-wide counter is at least every group's (a group steps only when the arena does)
        self.step_state.zero_(); self.step_state[0] = applied
        if self.gate_steps is not None:
            self.gate_steps.zero_()
            self.gate_steps[:self.n_flags] = torch.tensor(gsteps, dtype=torch.float32)
        self.step_count = int(applied)


# --------------------------------------------------------------------------- data-parallel exchange
class GradAllReduce:
    """Sum-all-reduce of a gradient arena over the data-parallel group (RCCL over xGMI on the GPU box, gloo in the CPU
    tests); the 1/world scale is folded into the optimizer step.  The arena is cut into contiguous buckets -- at the
    boundaries of the optimizer's completion groups when it has them (ArenaAdam(order=...): modality decoders, shared
    decoder, encoders), else into `buckets` equal slices -- and a bucket is reduced asynchronously as soon as every gradient
    in it is complete: autograd-accumulated gradients report through post-accumulate hooks, in-kernel gradient sinks
    (which fire no hook) through mark_ready(), called from the backward node of their mixing group (ops.set_group_ready_hook).
    What is still pending after backward (the group of the first encoder pass, the gate flags) goes from finish().
    BatchNorm statistics stay per replica (the reference has no SyncBN).  One reducer serves every optimizer that shares
    the arena layout: begin(optimizer) names the buffer to reduce.

    Diagnostics (`timing = True`): per finish(), a pair of events on the compute stream around the waits -- the time the
    compute stream idles for the exchange (`exposed_ms`) -- the number of buckets that had left before finish()
    (`early_buckets`) and the bytes reduced."""

    def __init__(self, optim, group=None, buckets=6, force_exchange=False):
        """force_exchange: issue the collectives even in a group of one rank (the sum over one rank is the identity, so the
        weights must not change by a bit): the way to run the RCCL code path -- async all-reduce of arena slices from the
        autograd thread, the waits on the compute stream, the gate-flag tail -- on a box with a single GPU."""
        self.optim, self.group, self.nbuckets = optim, group, buckets
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.exchanging = self.world > 1 or (force_exchange and dist.is_initialized())
        self.target = optim
        self.handles = []
        self.hooks = []
        self.armed = False
        self.timing = False
        self.events, self.early_buckets, self.bytes_reduced, self.calls = [], 0, 0, 0

    def _setup(self):
        o = self.optim
        n = o.numel
        if getattr(o, 'group_edges', None):
            edges = list(o.group_edges)
            big = max(n // 3, 1)                 # a group beyond a third of the arena is cut in two (the encoders' group)
            out = [edges[0]]
            for a, b in zip(edges[:-1], edges[1:]):
                if b - a > big:
                    out.append((a + (b - a) // 2) // 4 * 4)
                out.append(b)
            edges = sorted(set(out))
            self.nbuckets = len(edges) - 1
        else:
            edges = [int(round(n * k / self.nbuckets / 4)) * 4 for k in range(self.nbuckets + 1)]
            edges[-1] = n
        self.edges = edges
        self.bucket_of, self.pending0 = {}, [0] * self.nbuckets
        for p, off in zip(o.used, o.offsets):
            b = min(max(np.searchsorted(edges, off, side='right') - 1, 0), self.nbuckets - 1)
            last = min(max(np.searchsorted(edges, off + p.numel() - 1, side='right') - 1, 0), self.nbuckets - 1)
            # a tensor that straddles an edge gates every bucket it touches
            for bb in range(b, last + 1):
                self.pending0[bb] += 1
            self.bucket_of[id(p)] = (b, last)
            self.hooks.append(p.register_post_accumulate_grad_hook(self._hook))
        self.pending = list(self.pending0)
        self.done = set()

    def _hook(self, p):
        if not self.armed:
            return
        key = id(p)
        if key in self.done:                     # reported twice (hook and mark_ready): counted once
            return
        self.done.add(key)
        b, last = self.bucket_of[key]
        for bb in range(b, last + 1):
            self.pending[bb] -= 1
            if self.pending[bb] == 0:
                self._launch(bb)

    def mark_ready(self, params):
        """the gradients of `params` are complete in stream order (in-kernel sinks: no hook will fire for them)."""
        if not self.armed or not self.hooks:
            return
        for p in params:
            if id(p) in self.bucket_of:
                self._hook(p)

    def _launch(self, b):
        """bucket b is complete.  Collectives must be issued in the SAME order on every rank (RCCL pairs them by order), and which
        gradients a rank's backward produces first may depend on its batch (a modality missing from a rank's whole batch prunes
        that decoder's loss terms): buckets therefore leave strictly in index order -- the arena is laid out in completion order,
        so in the common case that IS the order in which they complete."""
        if not self.exchanging or self.ready[b]:
            return
        self.ready[b] = True
        t = self.target
        while self.next_bucket < self.nbuckets and self.ready[self.next_bucket]:
            k = self.next_bucket
            self.next_bucket += 1
            hi = self.edges[k + 1] if k + 1 < self.nbuckets else t._g_full.numel()      # last bucket: + the gate flags
            if self.armed:
                self.early_buckets += 1
            self.bytes_reduced += 4 * (hi - self.edges[k])
            self.handles.append(dist.all_reduce(t._g_full[self.edges[k]:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def begin(self, target=None):
        """call before backward(); `target`: the optimizer whose gradient buffer receives this backward pass."""
        self.target = target or self.optim
        if self.optim.used is not None and not self.hooks:
            self._setup()
        if self.hooks:
            self.pending = list(self.pending0)
            self.ready = [False] * self.nbuckets
            self.next_bucket = 0
            self.done = set()
        self.handles = []
        self.armed = True
        ops.set_group_ready_hook(self.mark_ready, key=id(self))

    def abort(self):
        """backward() raised: disarm (a later backward on this rank must not issue collectives the other ranks never
        issue) and forget the outstanding handles -- the step is lost, the caller re-raises."""
        self.armed = False
        ops.set_group_ready_hook(None, key=id(self))
        self.handles = []

    def finish(self):
        """call after backward(); returns the scale (1/world) the optimizer must apply."""
        self.armed = False
        ops.set_group_ready_hook(None, key=id(self))
        if not self.exchanging:
            return 1.0
        self.calls += 1
        if not self.hooks:                       # arena built lazily, first step: reduce per tensor
            for p in self.optim.params:
                if p.grad is not None:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
        else:
            for b in range(self.nbuckets):       # buckets whose gradients were not all reported (first-pass group, unused grads this step)
                self._launch(b)
            ev = None
            if self.timing and self.optim.flat_p.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            for h in self.handles:
                h.wait()
            if ev is not None:
                ev[1].record()
                self.events.append(ev)
        return 1.0 / self.world

    def exposed_ms(self):
        """total time (ms) the compute stream waited for the exchange over the finish() calls since the last reset
        (synchronises; call outside the timed region), then resets the diagnostics."""
        if self.events:
            torch.cuda.synchronize()
        tot = sum(a.elapsed_time(b) for a, b in self.events)
        out = dict(exposed_ms=tot, finish_calls=self.calls, early_buckets=self.early_buckets, bytes_reduced=self.bytes_reduced,
                   buckets=self.nbuckets)
        self.events, self.early_buckets, self.bytes_reduced, self.calls = [], 0, 0, 0
        return out


# --------------------------------------------------------------------------- one training iteration
_PLANAR_INPUTS = os.environ.get('MRDIS_PLANAR_INPUTS', '1') != '0'
_SKIP_DEAD_MAPS = os.environ.get('MRDIS_SKIP_DEAD_MAPS', '1') != '0'      # 0: the second encoder pass also computes its (unread) anatomy maps, as the reference does

LOSS_KEYS = ('recon_y', 'recon_y_fused', 'recon_x', 'recon_x_mix', 'kl', 'latent_z', 'sim_s', 'sim_z',
             'adv_s', 'adv_s_d', 'all')


def forward_losses(model, config, inputs, mask, mask_img, mask_host, phase='train', targets=None):
    """main_missing.py:165-251 (phase='train') / :389-505 (phase='test') for the loss set with non-zero
    weight in config.yaml."""
    M = len(config['contrast_list'])
    c = 2 * config['block_size'] + 1
    inputs_list = [inputs[:, i * c:(i + 1) * c] for i in range(M)]                               # :166-168 (views)
    if _PLANAR_INPUTS and M > 1 and inputs.is_cuda and inputs.is_contiguous(memory_format=torch.channels_last):
        # one pass that turns the (B, H, W, M c) batch into M dense (B, H, W, c) blocks: a c-channel slice of the interleaved tensor
        # touches every cache line of it, so each of the ~60 reads of a modality per step (first layers, their weight gradients,
        # the reconstruction losses) would fetch all M c channels from HBM
        B, _, H, W = inputs.shape
        planar = inputs.permute(0, 2, 3, 1).reshape(B, H, W, M, c).permute(3, 0, 1, 2, 4).contiguous()
        inputs_list = [planar[i].permute(0, 3, 1, 2) for i in range(M)]
    p = config['p']
    dev = inputs.device
    zero = torch.zeros((), device=dev)
    # the encoders' experts mixed for all modality labels by one launch; the decoder groups follow right before their first use
    # (MultimodalModel.premix: one backward node per group, so a group's gradients complete -- and its all-reduce starts -- mid-backward)
    if hasattr(model, 'premix'):
        model.premix('enc')
    else:
        ops.premix_all(model, model._type_table)
    si_list = model.compute_anatomy_encoding(inputs_list, mask_img)                              # :175
    zi_list, mu_list, lv_list = model.compute_modality_encoding(inputs_list, si_list, phase=phase)     # :176 / :400
    xi_fake_list = model.reconstruct_input_si_zi(si_list, zi_list)                               # :177
    xi_fake_mix_list = model.reconstruct_input_si_zj(si_list, zi_list)                           # :178
    parts = {k: zero for k in LOSS_KEYS}
    loss = zero
    if config['lambda_kl'] > 0:
        raise NotImplementedError('the kl loss is outside the hot path (lambda_kl = 0 in config.yaml)')
    if config['lambda_recon_y_fused'] > 0:
        # main_missing.py:201-208: reconstruct_output_si_fused returns sum(mask) rows (boolean-index quirk), so the
        # loss against B targets raises in the reference for every M > 1
        raise NotImplementedError('lambda_recon_y_fused: the reference path raises a shape error for M > 1')
    y_list = None
    if config['lambda_recon_y'] > 0:                                                             # :187-198
        if targets is None:
            raise ValueError('lambda_recon_y > 0 needs targets')
        y_list = model.reconstruct_output_si(si_list)
        if config['dataset_name'] == 'BraTS':
            parts['recon_y'] = model.compute_segmentation_loss_y_list(targets, y_list, mask, mask_host)
        else:
            parts['recon_y'] = model.compute_recon_loss_y_list(targets, y_list, mask, p, mask_host)
        loss = loss + config['lambda_recon_y'] * parts['recon_y']
    if config['lambda_recon_x'] > 0:
        parts['recon_x'] = model.compute_recon_loss_x_list(inputs_list, xi_fake_list, mask, p, mask_host)
        loss = loss + config['lambda_recon_x'] * parts['recon_x']
    if config['lambda_recon_x_mix'] > 0:
        parts['recon_x_mix'] = model.compute_recon_loss_x_mix_list(inputs_list, xi_fake_mix_list, mask, p, mask_host)
        loss = loss + config['lambda_recon_x_mix'] * parts['recon_x_mix']
    if config['lambda_latent_z'] > 0:                                                            # :228-233
        # (the maps of this second pass are read by the modality encoder only under others.mod_enc_s: otherwise the pass runs for its BatchNorm state alone)
        dead_maps = _SKIP_DEAD_MAPS and hasattr(model, 'modality_encoder_reads_s') and not model.modality_encoder_reads_s()
        si_new = model.compute_anatomy_encoding(xi_fake_list, mask_img, need_maps=False) if dead_maps else model.compute_anatomy_encoding(xi_fake_list, mask_img)
        _, mu_new, _ = model.compute_modality_encoding(xi_fake_list, si_new, phase=phase)
        parts['latent_z'] = model.compute_latent_z_loss(mu_list, mu_new, mask, mask_host)
        loss = loss + config['lambda_latent_z'] * parts['latent_z']
    if config['lambda_sim_s'] > 0:
        parts['sim_s'] = model.compute_similarity_s_loss(si_list, mask, mask_host=mask_host)
        loss = loss + config['lambda_sim_s'] * parts['sim_s']
    if config['lambda_sim_z'] > 0:
        parts['sim_z'] = model.compute_similarity_z_loss(zi_list, mask, mask_host=mask_host)
        loss = loss + config['lambda_sim_z'] * parts['sim_z']
    if config['lambda_adv_s'] > 0:
        parts['adv_s_d'], parts['adv_s'] = model.compute_adversarial_loss(si_list, mask, mask_host)
        loss = loss + config['lambda_adv_s'] * parts['adv_s']
    parts['all'] = loss
    aux = dict(si_list=si_list, zi_list=zi_list, mu_list=mu_list, lv_list=lv_list, xi_fake_list=xi_fake_list,
               xi_fake_mix_list=xi_fake_mix_list, y_list=y_list)
    return loss, parts, aux


class TrainStep:
    """Owns the optimizers and runs main_missing.py:165-289 for one batch.

    Accumulation (`accum = 16 // batch_size`, :282; the shipped config.yaml has batch_size 8 -> 2): every iteration
    adds its gradient to the accumulated one and clips the ACCUMULATED gradient in place (:272 runs each iteration);
    every `accum`-th iteration steps and zeroes.  The micro-batch gradient lands in its own buffer first, so under
    data parallelism each iteration's gradient is summed over the ranks exactly once before it joins the accumulator.

    Adversarial d-step: the reference calls `loss_adv_s_d.backward()` AFTER
    `optimizer.step()` on a retained graph (:283-289); under torch >= 1.5 that raises
    (weights were modified in place) -- see tests/golden/step_b2m2_adv.json.  The
    executable order used here is: both backward passes on the un-stepped graph
    (generator gradients first; then the discriminator-loss gradients into the second optimizer's own
    gradient buffer -- `p.grad` is re-pointed, nothing is cloned), then both
    Adam steps; `optimizer_d_s` spans ALL parameters as in the reference (:122).  Like the reference (:286-289) the
    discriminator step uses the d-loss of the iteration on which the optimizers step.
    """

    def __init__(self, model, config, ddp_group=None, ddp_buckets=6, force_exchange=False):
        self.model, self.config = model, config
        ops.set_compute_dtype(config.get('compute_dtype', 'f32'))
        self.accum = max(1, 16 // config['batch_size'])                                          # :282 (guarded for B > 16)
        used = model.trainable_parameters() if hasattr(model, 'trainable_parameters') else None
        order = model.completion_groups() if hasattr(model, 'completion_groups') else None
        self.optimizer = ArenaAdam(model.parameters(), lr=config['lr'], weight_decay=1e-5, used=used, order=order)   # :118
        if used is not None and hasattr(model, 'gated_parameter_groups'):
            self.optimizer.set_gates(model.gated_parameter_groups())
        self.optimizer_d_s = None
        if config['lambda_adv_s'] > 0:                                                            # :121-122
            if used is None:
                raise RuntimeError('the adversarial step needs a model with a static trainable-parameter list')
            self.optimizer_d_s = ArenaAdam(model.parameters(), lr=config['lr'], weight_decay=0.0, share_weights_of=self.optimizer)
        self.reducer = GradAllReduce(self.optimizer, ddp_group, ddp_buckets, force_exchange=force_exchange) \
            if (dist.is_available() and dist.is_initialized()) else None
        self.acc = None                          # accumulated (already reduced, clipped) gradient when accum > 1
        self.iter = 0
        self.last_grad_norm_sq = None            # device [sum g^2, #non-finite] of the gradient the last clip saw

    def __call__(self, inputs, mask, mask_img, mask_host=None, targets=None, it=None):
        """`it`: the loader index of this batch inside its epoch -- the reference steps on (it + 1) % accum == 0 with `it`
        restarting every epoch (:155, :282), a pending accumulation carries over; None = count iterations here."""
        if mask_host is None:
            mask_host = mask.cpu()
        do_step = self._advance(it)
        loss, parts, aux, scale = self._forward_backward(inputs, mask, mask_img, mask_host, targets, do_step)
        self._apply(scale, do_step)
        return loss.detach(), {k: v.detach() for k, v in parts.items()}, aux

    def _advance(self, it):
        self.iter = self.iter + 1 if it is None else it + 1
        return (self.iter % self.accum) == 0                                                     # :282

    def _forward_backward(self, inputs, mask, mask_img, mask_host, targets, do_step, exchange=True):
        """forward, losses, backward pass(es) into the optimizers' gradient buffers (:165-271).  exchange = False: no data-parallel
        exchange here (GraphedTrainStep reduces the arenas itself between its two graphs); returns the scale the optimizers apply."""
        cfg, model, opt = self.config, self.model, self.optimizer
        adv = cfg['lambda_adv_s'] > 0
        with ops.mix_cache():
            loss, parts, aux = forward_losses(model, cfg, inputs, mask, mask_img, mask_host, targets=targets)
            if opt.used is not None and opt.n_flags:
                def active():
                    cur = ops.step_mask_host()
                    return torch.from_numpy(model.active_decoders(mask_host if cur is None else cur))
                opt.mark_active(ops.host_value(active, mask.device))
            scale = self._backward(loss, opt, retain_graph=adv and do_step, exchange=exchange)    # :268-271
            if adv and do_step:
                # discriminator-loss gradients on the same (un-stepped) graph, into optimizer_d_s' own buffer
                od = self.optimizer_d_s
                od.attach_grads()
                try:
                    self._backward(parts['adv_s_d'], od, exchange=exchange)
                finally:
                    opt.attach_grads()
        if opt.used is None:
            opt._build()
        opt.check_new_grads()
        return loss, parts, aux, scale

    def _apply(self, scale, do_step):
        """clip, Adam step(s), zero the gradient buffers (:272-289)."""
        cfg, opt = self.config, self.optimizer
        adv = cfg['lambda_adv_s'] > 0
        if self.accum == 1:
            opt.step(fused_clip=True, grad_scale=scale, use_gates=True)                          # :272 + :283
            self.last_grad_norm_sq = opt.norm_finite
            opt.zero_grad()                                                                      # :284
        else:
            if self.acc is None:
                self.acc = torch.zeros_like(opt._g_full)
            self.acc.add_(opt._g_full, alpha=scale)                                              # += this iteration's (mean) gradient
            opt.zero_grad()
            n = opt.numel
            self.last_grad_norm_sq = opt.clip_in_place(self.acc[:n]).clone()                     # :272, every iteration
            if do_step:
                if opt.n_flags:
                    opt.gate_flags.copy_(self.acc[n:n + opt.n_flags])
                opt.step(fused_clip=False, gate_only=True, g=self.acc[:n], use_gates=True)       # :283; non-finite => skipped
                opt.zero_grad(); self.acc.zero_()                                                # :284
        if adv and do_step:
            od = self.optimizer_d_s
            od.step(fused_clip=False, grad_scale=scale)                                          # :287-289 (no clip on the d-step)
            od.zero_grad()
        flush_batch_counters()                                                                   # BatchNorm2d.num_batches_tracked of this step's calls, one launch

    def _backward(self, loss, target, retain_graph=False, exchange=True):
        """backward() with the gradient exchange armed around it; returns the scale the optimizer applies (1 / world).  A
        backward that raises leaves the reducer disarmed and its mid-backward hook removed."""
        red = self.reducer
        if red is None or not exchange:
            loss.backward(retain_graph=retain_graph)
            return 1.0
        red.begin(target)
        try:
            loss.backward(retain_graph=retain_graph)
        except BaseException:
            red.abort()
            raise
        return red.finish()

    def losses_to_host(self, parts):
        """the 11 scalars of main_missing.py:253-263 in one D2H copy."""
        vec = torch.stack([parts[k].float().reshape(()) for k in LOSS_KEYS]).cpu()
        return {k: float(vec[i]) for i, k in enumerate(LOSS_KEYS)}


def regular_mask(mask_host):
    """True when no loss term of main_missing.py:165-251 is pruned for this batch mask whichever (i, j) pair sim_s / adv_s draw: every
    modality present, every pair of modalities shares a sample, every (i, j) has a sample b with mask[b, i] mask[b, j] mask[b + 1, i] -- then
    the kernel sequence of the step does not depend on the mask's VALUES and a recorded step can be replayed for it."""
    mh = np.asarray(mask_host.numpy() if isinstance(mask_host, torch.Tensor) else mask_host, dtype=np.float32)
    M = mh.shape[1]
    if (mh.sum(0) == 0).any():
        return False
    for i in range(M):
        roll = np.concatenate([mh[1:, i], mh[0:1, i]], 0)
        for j in range(M):
            if i != j and ((mh[:, i] * mh[:, j]).sum() == 0 or (mh[:, i] * mh[:, j] * roll).sum() == 0):
                return False
    return True


class GraphedTrainStep:
    """TrainStep whose steady-state iterations are HIP-graph replays (config key `graph: true`, MRDIS_GRAPH=1, bench.py --graph).

    The library neither allocates nor synchronises and every host-produced value of a step (eps, the sim_s / adv_s pairs, the loss
    weights of the batch's mask, the decoder gate flags) reaches the device through ops.host_value closures, so one iteration --
    forward, losses, backward pass(es), clip, Adam, counters (main_missing.py:165-289) -- is recorded ONCE per configuration
    (step / accumulate phase, learning rates, shapes) after `WARM` eager iterations on the capture stream, and replayed thereafter:
    per step the host re-draws the closures in the recorded order (the global torch / numpy generators advance exactly as in the
    eager step, host draws model.py:3159-3162, 3485), ships them with one copy kernel, copies the batch into the static input
    buffers and launches the graph -- ~1.5 ms of host time instead of 45-70.  The adv_s pair picks two anatomy maps, i.e. which
    kernels run: it is drawn here, before the launch, and there is one recording per ordered pair (12 for M = 4, all made at the
    first recording, in one memory pool); the sim_s pair is data of the graph (every map is pooled, rows i, j picked by one-hot weights).
    A batch whose mask would prune a loss term (regular_mask() false: a modality absent from the whole batch ...) runs as an
    eager step, as does everything before the recording.  Results are bit-identical to the eager TrainStep
    (tests/test_gpu_graph.py); the third return value (aux: the step's activations) is None on replayed steps.

    Data parallel (world > 1): two graphs -- forward + backward(s) | clip + Adam -- with the gradient arenas all-reduced eagerly in
    between (one collective per arena; the bucketed overlap of the eager path is given up for a host-free step: the exchange is
    ~1 % of the step).  Returned tensors (loss, parts, aux) are static buffers of the graph: valid until the next call."""
    WARM = 2

    def __init__(self, step, warm=None):
        self.step = step
        self.warm = self.WARM if warm is None else int(warm)
        self.entries, self.seen = {}, {}
        self.stream = None
        self.static = None
        self.stats = {'replays': 0, 'captures': 0, 'eager': 0, 'eager_irregular_mask': 0}

    def __getattr__(self, name):                 # optimizer, reducer, losses_to_host, last_grad_norm_sq, accum ... are the wrapped step's
        return getattr(self.__dict__['step'], name)

    def _static_inputs(self, inputs, mask, mask_img, targets):
        st = self.static
        if st is None or st[0].shape != inputs.shape or st[0].dtype != inputs.dtype or (targets is None) != (st[3] is None):
            st = self.static = [torch.empty_like(inputs), torch.empty_like(mask), torch.empty_like(mask_img),
                                None if targets is None else torch.empty_like(targets)]
            self.entries.clear(); self.seen.clear()
        for dst, src in zip(st, (inputs, mask, mask_img, targets)):
            if src is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        return st

    def _predraw(self):
        """the step's two np.random draws (sim_s pair, then adv_s pair: model.py:3485, 3563), made here in the model's own order: the adv_s pair
        selects WHICH recorded graph replays, so it must be known before the launch.  The model then uses these instead of drawing."""
        cfg = self.step.config
        M = len(cfg['contrast_list'])
        pairs = {}
        if M > 2:
            if cfg['lambda_sim_s'] > 0:
                sel = np.random.choice(M, 2, replace=False); pairs['sim_s'] = (int(sel[0]), int(sel[1]))
            if cfg['lambda_adv_s'] > 0:
                sel = np.random.choice(M, 2, replace=False); pairs['adv_s'] = (int(sel[0]), int(sel[1]))
        return pairs

    def __call__(self, inputs, mask, mask_img, mask_host=None, targets=None, it=None):
        ts = self.step
        if mask_host is None:
            mask_host = mask.cpu()
        if not inputs.is_cuda:
            return ts(inputs, mask, mask_img, mask_host, targets, it)
        pairs = self._predraw()
        ops.set_forced_pairs(pairs)
        try:
            return self._call(inputs, mask, mask_img, mask_host, targets, it, pairs)
        finally:
            ops.set_forced_pairs(None)

    def _call(self, inputs, mask, mask_img, mask_host, targets, it, pairs):
        ts = self.step
        do_step = ts._advance(it)
        if not regular_mask(mask_host):
            self.stats['eager'] += 1; self.stats['eager_irregular_mask'] += 1
            return self._eager(inputs, mask, mask_img, mask_host, targets, do_step)
        opt, od = ts.optimizer, ts.optimizer_d_s
        key = (bool(do_step), tuple(inputs.shape), inputs.dtype, targets is not None, float(opt.lr), None if od is None else float(od.lr), ops.compute_dtype())
        x, m, mi, tg = self._static_inputs(inputs, mask, mask_img, targets)
        group = self.entries.get(key)
        if group is None:
            for k in [k for k in self.entries if k[:4] == key[:4] and k != key]:      # a scheduler moved the learning rate: the old recordings are dead
                del self.entries[k]
            n = self.seen.get(key, 0)
            self.seen[key] = n + 1
            if self.stream is None:
                self.stream = torch.cuda.Stream()
            if n < self.warm:                    # eager, on the capture stream: workspaces, plans and caches reach their final size there
                self.stats['eager'] += 1
                return self._eager(x, m, mi, mask_host, tg, do_step, side=self.stream)
            try:
                group = self._record_all(x, m, mi, mask_host, tg, do_step, pairs)
            except Exception as ex:                  # noqa: BLE001 -- a step that cannot be recorded (host-value block full, an op that refuses capture) trains eagerly
                import warnings
                warnings.warn(f'GraphedTrainStep: recording failed ({type(ex).__name__}: {ex}); this configuration runs eagerly from here on')
                torch.cuda.synchronize()
                discard_batch_counters()
                group = 'eager'
            self.entries[key] = group
        if group == 'eager':
            self.stats['eager'] += 1
            return self._eager(x, m, mi, mask_host, tg, do_step)
        ent = group[pairs.get('adv_s')]
        ops.set_step_mask_host(mask_host)
        ent['hv'].refill()                       # eps and the mask weights of THIS step (the closures draw from the global torch generator in the recorded order)
        self.stats['replays'] += 1
        self._launch(ent, do_step)
        return ent['out']

    def _eager(self, x, m, mi, mask_host, tg, do_step, side=None):
        """an un-recorded iteration (warm-up, or a batch whose mask prunes loss terms).  Under data parallelism it issues the SAME collectives as a replayed one
        -- one all-reduce per gradient arena between backward and the optimizer -- because whether a rank replays or not depends on ITS batch's mask: a rank
        on the bucketed exchange of TrainStep would never pair with a rank that replays.  side: run the kernels on the capture stream (warm-up); the
        collectives stay on the caller's stream, as in a replayed step."""
        ts = self.step
        cur = torch.cuda.current_stream()

        def on(fn):
            if side is None:
                return fn()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                out = fn()
            cur.wait_stream(side)
            return out
        if self._exchanging():
            loss, parts, aux, _ = on(lambda: ts._forward_backward(x, m, mi, mask_host, tg, do_step, exchange=False))
            self._all_reduce(do_step)
            on(lambda: ts._apply(1.0 / ts.reducer.world, do_step))
        else:
            def both():
                loss, parts, aux, scale = ts._forward_backward(x, m, mi, mask_host, tg, do_step)
                ts._apply(scale, do_step)
                return loss, parts, aux
            loss, parts, aux = on(both)
        return loss.detach(), {k: v.detach() for k, v in parts.items()}, aux

    def _all_reduce(self, do_step):
        ts = self.step
        red = ts.reducer
        dist.all_reduce(ts.optimizer._g_full, op=dist.ReduceOp.SUM, group=red.group)
        if ts.optimizer_d_s is not None and do_step:
            dist.all_reduce(ts.optimizer_d_s._g_full, op=dist.ReduceOp.SUM, group=red.group)

    def _exchanging(self):
        red = self.step.reducer
        return red is not None and red.exchanging

    def _record_all(self, x, m, mi, mask_host, tg, do_step, pairs):
        """one recording per adv_s pair (M (M - 1) ordered pairs; one when the model does not draw), all in ONE memory pool: replays never overlap and
        a recording keeps nothing but its loss scalars alive, so the pool holds one step's activations.  Recording executes no kernel and, with the
        generators restored afterwards, consumes no draw: the step itself then runs as the first replay."""
        cfg = self.step.config
        M = len(cfg['contrast_list'])
        variants = [None]
        if 'adv_s' in pairs:
            variants = [(i, j) for i in range(M) for j in range(M) if i != j]
        pool = torch.cuda.graph_pool_handle()
        rng = (torch.get_rng_state(), np.random.get_state())
        ts = self.step
        counts = (ts.optimizer.step_count, None if ts.optimizer_d_s is None else ts.optimizer_d_s.step_count)      # host-side call counters: recording is not stepping
        group = {}
        try:
            for v in variants:
                ops.set_forced_pairs(dict(pairs, adv_s=v) if v is not None else pairs)
                group[v] = self._record(x, m, mi, mask_host, tg, do_step, pool)
                self.stats['captures'] += 1
        finally:
            ops.set_forced_pairs(pairs)
            torch.set_rng_state(rng[0]); np.random.set_state(rng[1])
            ts.optimizer.step_count = counts[0]
            if ts.optimizer_d_s is not None:
                ts.optimizer_d_s.step_count = counts[1]
        return group

    def _record(self, x, m, mi, mask_host, tg, do_step, pool):
        ts = self.step
        dev = x.device
        split = self._exchanging()
        hv = ops.HostValues(dev)
        prev = ops.set_host_values(hv)
        ops.set_step_mask_host(mask_host)
        torch.cuda.synchronize()
        g1, g2 = torch.cuda.CUDAGraph(), None
        try:
            hv.start_recording()
            with torch.cuda.graph(g1, pool=pool, stream=self.stream, capture_error_mode='thread_local'):
                loss, parts, aux, _ = ts._forward_backward(x, m, mi, mask_host, tg, do_step, exchange=False)
                if not split:
                    ts._apply(1.0, do_step)
                out = (loss.detach(), {k: v.detach() for k, v in parts.items()}, None)      # (aux would pin a step's activations per recording)
                del loss, parts, aux
            hv.stop_recording()
            if split:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=pool, stream=self.stream, capture_error_mode='thread_local'):
                    ts._apply(1.0 / ts.reducer.world, do_step)
        finally:
            hv.stop_recording()
            ops.set_host_values(prev)
        return {'g1': g1, 'g2': g2, 'hv': hv, 'out': out}

    def _launch(self, ent, do_step):
        ts = self.step
        ent['hv'].ship()
        ent['g1'].replay()
        if ent['g2'] is not None:
            self._all_reduce(do_step)
            ent['g2'].replay()
        if ts.accum == 1 or do_step:
            ts.optimizer.step_count += 1
            if ts.optimizer_d_s is not None and do_step:
                ts.optimizer_d_s.step_count += 1


def make_train_step(model, config, **kw):
    """TrainStep, wrapped for graph replay when the configuration (`graph: true`) or the environment (MRDIS_GRAPH=1) asks for it."""
    step = TrainStep(model, config, **kw)
    want = config.get('graph', None)
    if want is None:
        want = os.environ.get('MRDIS_GRAPH', '0') not in ('', '0')
    return GraphedTrainStep(step) if want else step


class EvalStep:
    """evaluate() of the reference for one batch (main_missing.py:337-517): model.eval() (BatchNorm on
    running statistics), no_grad, z = mu, the same loss set, plus the reconstruction metrics of the
    cross-modality (mix) reconstructions (:520-528) computed on the device instead of shipping both
    stacks to skimage on the host (util.py:935-978): per image (channel 0 of each sample) min-shifted,
    data range = max of the shifted target; keys as in the reference ('rmse' holds the MSE, as there).
    Returns (loss, parts, metrics, aux); metrics values are (M(M-1)B,) device tensors in the reference order."""

    def __init__(self, model, config):
        self.model, self.config = model, config

    @torch.no_grad()
    def __call__(self, inputs, mask, mask_img, mask_host=None, targets=None):
        model, cfg = self.model, self.config
        if mask_host is None:
            mask_host = mask.cpu()
        was = model.training
        flush_batch_counters()
        model.eval()
        try:
            with ops.mix_cache():
                loss, parts, aux = forward_losses(model, cfg, inputs, mask, mask_img, mask_host, phase='test', targets=targets)
                M = len(cfg['contrast_list'])
                c = 2 * cfg['block_size'] + 1
                rows, k = [], 0
                for i in range(M):                                                   # main_missing.py:520-528
                    for j in range(M):
                        if i == j:
                            continue
                        rows.append(hip.recon_metrics(inputs[:, j * c:(j + 1) * c], aux['xi_fake_mix_list'][k]))
                        k += 1
                rows = torch.cat(rows, 0)                                            # (M(M-1)B, 3), reference order
                metrics = {'rmse': rows[:, 0], 'psnr': rows[:, 1], 'ssim': rows[:, 2]}
        finally:
            model.train(was)
        return loss, parts, metrics, aux


# --------------------------------------------------------------------------- checkpoint layout (util.py:148-153)
def save_checkpoint(state, is_best, checkpoint_dir):
    """epochNNN.pth.tar = {'epoch','monitor_metric','stat','optimizer','scheduler','model'[, 'optimizer_d_s']}
    (main_missing.py:330-335); best copied to model_best.pth.tar."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    fn = os.path.join(checkpoint_dir, 'epoch' + str(state['epoch']).zfill(3) + '.pth.tar')
    torch.save(state, fn)
    if is_best:
        shutil.copyfile(fn, os.path.join(checkpoint_dir, 'model_best.pth.tar'))
    return fn


def load_checkpoint_model(model, state_dict):
    """util.py:895-903: name + shape filtered load (missing / mismatching keys are skipped)."""
    cur = model.state_dict()
    ok = {k: v for k, v in state_dict.items() if k in cur and tuple(v.shape) == tuple(cur[k].shape)}
    cur.update(ok)
    model.load_state_dict(cur)
    return sorted(set(state_dict) - set(ok))
