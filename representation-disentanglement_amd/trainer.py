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
from .model import MultimodalModel

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
class ArenaAdam:
    """torch.optim.Adam(lr, weight_decay=wd, amsgrad=True) (main_missing.py:118) with
    clip_grad_norm_(1.0) (:272) and the finite check (:273-278) folded into the step.

    Built lazily at the first step from the parameters that actually carry a gradient (43 % of
    the reference's parameters never do; torch's Adam skips those, so do we).
    Afterwards `p.data` / `p.grad` are views into the arena.
    """

    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, max_norm=1.0):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.step_count = 0
        self.flat_p = self.flat_g = self.m = self.v = self.vmax = None
        self.used = None
        self.norm_finite = None

    def _build(self):
        used = [p for p in self.params if p.grad is not None]
        if not used:
            raise RuntimeError('ArenaAdam.step() before any backward()')
        # parameters whose gradient the backward kernels add to in place (ops._grad_sink) never pass through autograd's
        # accumulation, so their post-accumulate hooks do not fire: keep them together at the end of the arena, i.e. in
        # the last all-reduce bucket (GradAllReduce.finish() reduces it after backward; the others keep overlapping)
        used.sort(key=lambda p: 1 if getattr(p, '_mrdis_sink', False) else 0)
        dev = used[0].device
        # 16-byte align every tensor inside the arena (vectorised kernels read params in place)
        offs, n = [], 0
        for p in used:
            offs.append(n); n += (p.numel() + 3) // 4 * 4
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(used, offs):
                k = p.numel()
                self.flat_p[o:o + k].copy_(p.data.reshape(-1))
                self.flat_g[o:o + k].copy_(p.grad.reshape(-1))
                p.data = self.flat_p[o:o + k].view(p.shape)
                p.grad = self.flat_g[o:o + k].view(p.shape)
        self.m = torch.zeros_like(self.flat_p); self.v = torch.zeros_like(self.flat_p); self.vmax = torch.zeros_like(self.flat_p)
        self.used, self.offsets, self.numel = used, offs, n
        self.used_ids = {id(p) for p in used}
        self.norm_finite = torch.zeros(2, dtype=torch.float32, device=dev)

    def grad_norm_sq(self):
        """device tensor [sum g^2, #non-finite] over the arena."""
        self.norm_finite.zero_()
        hip.sumsq_finite(self.flat_g, self.norm_finite)
        return self.norm_finite

    def clip_in_place(self):
        """clip_grad_norm_ semantics on the accumulated gradient (used between micro-batches when
        accumulating, main_missing.py:272 runs every iteration)."""
        nf = self.grad_norm_sq()
        coef = torch.clamp(self.max_norm / (torch.sqrt(nf[0]) + 1e-6), max=1.0)
        self.flat_g.mul_(coef)

    def check_new_grads(self):
        for p in self.params:
            if p.grad is not None and id(p) not in self.used_ids:
                raise RuntimeError('a parameter outside the arena received a gradient; rebuild the optimizer')

    def step(self, fused_clip=True, grad_scale=1.0):
        if self.used is None:
            self._build()
        self.step_count += 1
        nf = None
        if fused_clip:
            nf = self.grad_norm_sq()
        hip.adam_amsgrad_step(self.flat_p, self.flat_g, self.m, self.v, self.vmax, self.lr, self.betas[0], self.betas[1],
                              self.eps, self.wd, self.step_count, nf, self.max_norm if fused_clip else 0.0, grad_scale)

    def zero_grad(self):
        if self.used is None:
            for p in self.params:
                p.grad = None
        else:
            self.flat_g.zero_()

    # torch.optim-compatible checkpoint payload (main_missing.py:330-335 stores optimizer.state_dict())
    def state_dict(self):
        if self.used is None:
            return {'state': {}, 'param_groups': [{'lr': self.lr, 'betas': self.betas, 'eps': self.eps,
                                                   'weight_decay': self.wd, 'amsgrad': True}]}
        idx = {id(p): i for i, p in enumerate(self.params)}
        state = {}
        for p, o in zip(self.used, self.offsets):
            k = p.numel()
            state[idx[id(p)]] = {'step': torch.tensor(float(self.step_count)),
                                 'exp_avg': self.m[o:o + k].view(p.shape).clone(),
                                 'exp_avg_sq': self.v[o:o + k].view(p.shape).clone(),
                                 'max_exp_avg_sq': self.vmax[o:o + k].view(p.shape).clone()}
        return {'state': state, 'param_groups': [{'lr': self.lr, 'betas': self.betas, 'eps': self.eps,
                                                  'weight_decay': self.wd, 'amsgrad': True,
                                                  'params': list(range(len(self.params)))}]}


# --------------------------------------------------------------------------- data-parallel exchange
class GradAllReduce:
    """Mean-all-reduce of the gradient arena over the data-parallel group (RCCL over xGMI on
    the GPU box, gloo in the CPU tests).  The arena is cut into `buckets` contiguous slices in
    reverse-execution order; each slice is reduced asynchronously as soon as autograd has
    produced every gradient in it (post-accumulate hooks), so the exchange overlaps the rest of
    the backward pass.  BatchNorm statistics stay per replica (the reference has no SyncBN)."""

    def __init__(self, optim, group=None, buckets=6):
        self.optim, self.group, self.nbuckets = optim, group, buckets
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.ready = None
        self.handles = []
        self.hooks = []

    def _setup(self):
        o = self.optim
        n = o.numel
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

    def _hook(self, p):
        b, last = self.bucket_of[id(p)]
        for bb in range(b, last + 1):
            self.pending[bb] -= 1
            if self.pending[bb] == 0:
                self._launch(bb)

    def _launch(self, b):
        if self.world == 1:
            return
        sl = self.optim.flat_g[self.edges[b]:self.edges[b + 1]]
        self.handles.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def begin(self):
        """call before backward()."""
        if self.optim.used is not None and not self.hooks:
            self._setup()
        if self.hooks:
            self.pending = list(self.pending0)
        self.handles = []

    def finish(self):
        """call after backward(); returns the scale (1/world) the optimizer must apply."""
        if self.world == 1:
            return 1.0
        if not self.hooks:                       # first step: arena not built yet -> reduce per tensor
            for p in self.optim.params:
                if p.grad is not None:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
        else:
            for b in range(self.nbuckets):       # buckets whose hooks never completed (unused grads this step)
                if self.pending[b] != 0:
                    self._launch(b)
            for h in self.handles:
                h.wait()
        return 1.0 / self.world


# --------------------------------------------------------------------------- one training iteration
LOSS_KEYS = ('recon_y', 'recon_y_fused', 'recon_x', 'recon_x_mix', 'kl', 'latent_z', 'sim_s', 'sim_z',
             'adv_s', 'adv_s_d', 'all')


def forward_losses(model, config, inputs, mask, mask_img, mask_host, phase='train', targets=None):
    """main_missing.py:165-251 (phase='train') / :389-505 (phase='test') for the loss set with non-zero
    weight in config.yaml."""
    M = len(config['contrast_list'])
    c = 2 * config['block_size'] + 1
    inputs_list = [inputs[:, i * c:(i + 1) * c] for i in range(M)]                               # :166-168 (views)
    p = config['p']
    dev = inputs.device
    zero = torch.zeros((), device=dev)
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
        si_new = model.compute_anatomy_encoding(xi_fake_list, mask_img)
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

    Adversarial d-step: the reference calls `loss_adv_s_d.backward()` AFTER
    `optimizer.step()` on a retained graph (:283-289); under torch >= 1.5 that raises
    (weights were modified in place) -- see tests/golden/step_b2m2_adv.json.  The
    executable order used here is: both backward passes on the un-stepped graph
    (generator gradients first, stashed; then discriminator-loss gradients), then both
    Adam steps; `optimizer_d_s` spans ALL parameters as in the reference (:122).
    """

    def __init__(self, model, config, ddp_group=None, ddp_buckets=6):
        self.model, self.config = model, config
        self.accum = max(1, 16 // config['batch_size'])                                          # :282 (guarded for B > 16)
        self.optimizer = ArenaAdam(model.parameters(), lr=config['lr'], weight_decay=1e-5)       # :118
        self.optimizer_d_s = ArenaAdam(model.parameters(), lr=config['lr'], weight_decay=0.0) \
            if config['lambda_adv_s'] > 0 else None                                              # :121-122
        self.reducer = GradAllReduce(self.optimizer, ddp_group, ddp_buckets) \
            if (dist.is_available() and dist.is_initialized()) else None
        self.iter = 0
        self._stash = None

    def __call__(self, inputs, mask, mask_img, mask_host=None, targets=None):
        cfg, model = self.config, self.model
        adv = cfg['lambda_adv_s'] > 0
        if mask_host is None:
            mask_host = mask.cpu()
        with ops.mix_cache():
            loss, parts, aux = forward_losses(model, cfg, inputs, mask, mask_img, mask_host, targets=targets)
            if self.reducer:
                self.reducer.begin()
            loss.backward(retain_graph=adv)                                                      # :268-271
            scale = self.reducer.finish() if self.reducer else 1.0
            self.iter += 1
            do_step = (self.iter % self.accum) == 0                                              # :282
            if adv:
                # stash generator grads, get discriminator-loss grads on the same (un-stepped) graph
                if self.optimizer.used is None:
                    self.optimizer._build()
                g_main = self.optimizer.flat_g.clone()
                self.optimizer.flat_g.zero_()
                parts['adv_s_d'].backward()
                if self.reducer and self.reducer.world > 1:
                    dist.all_reduce(self.optimizer.flat_g, group=self.reducer.group)
                g_d = self.optimizer.flat_g.clone()
                self.optimizer.flat_g.copy_(g_main)
        if self.accum == 1:
            self.optimizer.step(fused_clip=True, grad_scale=scale)                               # :272 + :283
            self.optimizer.zero_grad()                                                           # :284
        else:
            if scale != 1.0:
                self.optimizer.flat_g.mul_(scale) if self.optimizer.used is not None else None
            if self.optimizer.used is None:
                self.optimizer._build()
            self.optimizer.clip_in_place()
            if do_step:
                self.optimizer.step(fused_clip=False)
                self.optimizer.zero_grad()
        if adv and do_step:
            od = self.optimizer_d_s
            if od.used is None:      # share the weight arena; own gradient / moment buffers
                od.used, od.offsets, od.numel = self.optimizer.used, self.optimizer.offsets, self.optimizer.numel
                od.used_ids = self.optimizer.used_ids
                od.flat_p = self.optimizer.flat_p
                od.flat_g = torch.zeros_like(self.optimizer.flat_g)
                od.m = torch.zeros_like(od.flat_p); od.v = torch.zeros_like(od.flat_p); od.vmax = torch.zeros_like(od.flat_p)
                od.norm_finite = torch.zeros(2, dtype=torch.float32, device=od.flat_p.device)
            od.flat_g.copy_(g_d)
            od.step_count += 1
            hip.adam_amsgrad_step(od.flat_p, od.flat_g, od.m, od.v, od.vmax, od.lr, od.betas[0], od.betas[1], od.eps,
                                  od.wd, od.step_count, None, 0.0, scale)                        # :287-289 (no clip on the d-step)
        return loss.detach(), {k: v.detach() for k, v in parts.items()}, aux

    def losses_to_host(self, parts):
        """the 11 scalars of main_missing.py:253-263 in one D2H copy."""
        vec = torch.stack([parts[k].float().reshape(()) for k in LOSS_KEYS]).cpu()
        return {k: float(vec[i]) for i, k in enumerate(LOSS_KEYS)}


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
