"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden-vector generator.

Runs ONLY in the build container (it needs /root/reference).  It imports the
real reference `src/model.py` on CPU -- with empty stand-ins for the
non-arithmetic modules that are absent from this image (torchvision.models,
scipy.misc, skimage, nibabel, h5py, nonechucks) -- calls
its classes / methods in the order of `src/main_missing.py:175-284`, and
writes small fixtures (inputs are re-generated from seeds, only outputs are
stored) to tests/golden/.  No reference source text is written anywhere.

    python oracle/gen_golden.py            # all fixtures (~1-2 min CPU)
"""
import contextlib
import io
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden')
REF_SRC = '/root/reference/src'
if os.path.join(ROOT, 'tests') not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fixtures import (DATA_CFG, DATA_SLICES, data_lists, make_inputs, make_seg_targets, reinit_discriminator,      # noqa: E402
                      seeded)


def import_reference():
    import scipy
    sys.dont_write_bytecode = True

    def stub(n, **a):
        m = types.ModuleType(n); m.__dict__.update(a); sys.modules[n] = m; return m
    tv = stub('torchvision'); tv.models = stub('torchvision.models')
    scipy.misc = stub('scipy.misc')
    sk = stub('skimage')
    for s in ('io', 'transform', 'color', 'metrics'):
        setattr(sk, s, stub('skimage.' + s))
    sk.measure = stub('skimage.measure', compare_nrmse=None, compare_psnr=None, compare_ssim=None)
    for n in ('nibabel', 'h5py', 'nonechucks'):
        stub(n)
    sys.path.insert(0, REF_SRC)
    import model as ref     # noqa
    return ref


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def pool8(x):
    return F.avg_pool2d(x, 8).numpy()


# ---------------------------------------------------------------- unit blocks
def gen_units(ref):
    out = {}
    # --- CondConv2d, per-sample (non-uniform) types: 3 kernel geometries
    for name, (ci, co, k, s, p, hw) in dict(
            c3s1=(5, 6, 3, 1, 1, (9, 11)), c4s2=(7, 8, 4, 2, 1, (12, 10)),
            c3s2=(4, 6, 3, 2, 1, (11, 13)), c1s1=(6, 3, 1, 1, 0, (7, 5))).items():
        torch.manual_seed(100)
        m = ref.CondConv2d(ci, co, k, s, padding=p)
        with torch.no_grad():
            m.bias.copy_(seeded((co,), 7, 0.1))
        x = seeded((3, ci) + hw, 1).requires_grad_(True)
        t = torch.tensor([[1.], [2.], [4.]])
        y = m(x, t)
        gy = seeded(tuple(y.shape), 2)
        y.backward(gy)
        out[f'cond_{name}_y'] = y.detach().numpy()
        out[f'cond_{name}_dx'] = x.grad.numpy()
        out[f'cond_{name}_dw'] = m.weight.grad.numpy()
        out[f'cond_{name}_db'] = m.bias.grad.numpy()
        out[f'cond_{name}_dfcw'] = m._routing_fn.fc.weight.grad.numpy()
        out[f'cond_{name}_dfcb'] = m._routing_fn.fc.bias.grad.numpy()

    # --- Conv_BN_Act_New (identity act quirk) and Act_Deconv_BN_Concat_New
    torch.manual_seed(101)
    m = quiet(ref.Conv_BN_Act_New, 6, 8, is_cond=True)
    m.train()
    x = seeded((3, 6, 12, 16), 3).requires_grad_(True)
    t = 2 * torch.ones(3, 1)
    y = m(x, t); y.backward(seeded(tuple(y.shape), 4))
    out['cba_y'] = y.detach().numpy(); out['cba_dx'] = x.grad.numpy()
    out['cba_dw'] = m.conv.weight.grad.numpy()
    out['cba_dbn_w'] = m.bn.weight.grad.numpy(); out['cba_dbn_b'] = m.bn.bias.grad.numpy()
    out['cba_run_mean'] = m.bn.running_mean.numpy().copy()
    out['cba_run_var'] = m.bn.running_var.numpy().copy()

    torch.manual_seed(102)
    m = quiet(ref.Act_Deconv_BN_Concat_New, 6, 5, is_cond=True)
    m.train()
    xu = seeded((2, 6, 5, 6), 5).requires_grad_(True)
    xd = seeded((2, 4, 10, 12), 6)
    t = 3 * torch.ones(2, 1)
    y = m(xd, xu, t); y.backward(seeded(tuple(y.shape), 7))
    out['adb_y'] = y.detach().numpy(); out['adb_dx'] = xu.grad.numpy()
    out['adb_dw'] = m.conv.weight.grad.numpy()

    # --- SPADEBlockNew
    torch.manual_seed(103)
    m = quiet(ref.SPADEBlockNew, (10, 12), in_num_ch=8, out_num_ch=6, s_num_ch=4, is_cond=True)
    s = torch.softmax(seeded((2, 4, 40, 48), 8), 1).requires_grad_(True)
    z = seeded((2, 8, 10, 12), 9).requires_grad_(True)
    t = 1 * torch.ones(2, 1)
    y = m(s, z, t); y.backward(seeded(tuple(y.shape), 10))
    out['spade_y'] = y.detach().numpy(); out['spade_ds'] = s.grad.numpy()
    out['spade_dz'] = z.grad.numpy(); out['spade_dw_gamma'] = m.gamma.weight.grad.numpy()

    # --- ModalityEncoderNew (fixed 160x192 by construction, model.py:2396)
    torch.manual_seed(104)
    m = quiet(ref.ModalityEncoderNew, img_num_ch=7, s_num_ch=0, first_num_ch=16, z_size=16, is_cond=True)
    x = seeded((2, 7, 160, 192), 11)
    mu, lv = m(x, None, 2 * torch.ones(2, 1))
    (mu.sum() + 2 * lv.sum()).backward()
    out['modenc_mu'] = mu.detach().numpy(); out['modenc_lv'] = lv.detach().numpy()
    out['modenc_dw1'] = m.conv1.weight.grad.numpy()

    # --- Discriminator (dense head and PatchGAN head)
    for pg in (False, True):
        torch.manual_seed(105)
        m = ref.Discriminator(in_num_ch=4, inter_num_ch=16, is_patch_gan=pg)
        m.train()
        x = torch.softmax(seeded((2, 4, 160, 192), 12), 1)
        y = m(x)
        out[f'disc_{"patch" if pg else "dense"}_y'] = y.detach().numpy()

    # --- bilinear flavours used on the path (model.py:2175, 2432, 2501)
    x = seeded((2, 3, 5, 6), 13)
    out['bil_ac_true_x2'] = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True).numpy()
    out['bil_ac_false_x2'] = F.interpolate(x, scale_factor=(2, 2), mode='bilinear').numpy()
    xs = seeded((2, 4, 32, 64), 14)
    out['bil_down_to_5x6'] = F.interpolate(xs, size=(5, 6), mode='bilinear').numpy()
    np.savez_compressed(os.path.join(OUT, 'units.npz'), **out)
    print('units.npz', len(out), 'arrays')


# ------------------------------------------------------------ full train step


def build_ref_model(ref, M, adv=False, out_num_ch=1):
    return quiet(
        ref.MultimodalModel, input_size=(160, 192), modality_num=M, in_num_ch=7, out_num_ch=out_num_ch,
        s_num_ch=4, z_size=16, is_cond=True, is_discrim_s=adv, is_distri_z=False,
        s_compact_method='max', s_sim_method='cosine', z_sim_method='cosine', shared_ana_enc=True,
        shared_mod_enc=True, shared_inp_dec=False, device=torch.device('cpu'),
        input_output_act='no', target_output_act='no', target_model_name='U+SA', fuse_method='mean',
        others={'mod_enc_s': False, 'ana_dec_act': 'softmax', 'old': False, 'softmax_remove_mask': True})


HOT_PREFIXES = ('anatomy_encoder_enc_list.', 'anatomy_encoder_dec.', 'modality_encoder_list.',
                'input_decoder_list.', 'discrim_s.')


def gen_step(ref, tag, B, M, drop=False, adv=False, recon_y=False):
    lam = dict(recon_x=1.0, recon_x_mix=2.0, latent_z=0.1, sim_s=10.0, sim_z=2.0,
               adv_s=(1.0 if adv else 0.0), recon_y=(1.0 if recon_y else 0.0))
    torch.manual_seed(10); np.random.seed(10)                       # main_missing.py:18-21
    model = build_ref_model(ref, M, adv, out_num_ch=4 if recon_y else 1)
    prefixes = HOT_PREFIXES + (('output_decoder.',) if recon_y else ())
    if adv:
        reinit_discriminator(model.discrim_s)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, weight_decay=1e-5, amsgrad=True)
    opt_d = torch.optim.Adam(model.parameters(), lr=2e-4, amsgrad=True) if adv else None
    inputs, mask, mask_img = make_inputs(B, M, 160, 192, seed=10, drop=drop)
    # the reference's output_decoder / discriminator constructors consumed global RNG draws the
    # restatement does not replay; re-seed so the eps draws of `sample` (model.py:3160) and the
    # sim_s pair choice (model.py:3485) are reproducible on both sides.
    torch.manual_seed(11); np.random.seed(11)
    w0 = {k: float(v.double().sum()) for k, v in model.state_dict().items()
          if k.startswith(prefixes) and v.dtype.is_floating_point}

    x_list = [inputs[:, i * 7:(i + 1) * 7] for i in range(M)]
    # ---- main_missing.py:175-251 call order
    s_list = model.compute_anatomy_encoding(x_list, mask_img)
    z_list, mu_list, lv_list = model.compute_modality_encoding(x_list, s_list, phase='train')
    xf = model.reconstruct_input_si_zi(s_list, z_list)
    xmix = model.reconstruct_input_si_zj(s_list, z_list)
    parts = {}
    loss = 0
    y_list = None
    if recon_y:                                                       # main_missing.py:187-198, BraTS -> segmentation loss
        targets = make_seg_targets(B, 160, 192, seed=13)
        y_list = model.reconstruct_output_si(s_list)
        parts['recon_y'] = model.compute_segmentation_loss_y_list(targets, y_list, mask)
        loss = loss + lam['recon_y'] * parts['recon_y']
    parts['recon_x'] = model.compute_recon_loss_x_list(x_list, xf, mask, p=1)
    parts['recon_x_mix'] = model.compute_recon_loss_x_mix_list(x_list, xmix, mask, p=1)
    loss = loss + lam['recon_x'] * parts['recon_x'] + lam['recon_x_mix'] * parts['recon_x_mix']
    s_new = model.compute_anatomy_encoding(xf, mask_img)
    _, mu_new, _ = model.compute_modality_encoding(xf, s_new, phase='train')
    parts['latent_z'] = model.compute_latent_z_loss(mu_list, mu_new, mask)
    loss = loss + lam['latent_z'] * parts['latent_z']
    parts['sim_s'] = model.compute_similarity_s_loss(s_list, mask)
    loss = loss + lam['sim_s'] * parts['sim_s']
    parts['sim_z'] = model.compute_similarity_z_loss(z_list, mask)
    loss = loss + lam['sim_z'] * parts['sim_z']
    if adv:
        parts['adv_s_d'], parts['adv_s'] = model.compute_adversarial_loss(s_list, mask)
        loss = loss + lam['adv_s'] * parts['adv_s']
    loss.backward(retain_graph=adv)
    grad_norms = {n: float(p.grad.double().norm()) for n, p in model.named_parameters()
                  if p.grad is not None}
    gnorm = float(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0))
    opt.step(); opt.zero_grad()
    w1 = {k: float(v.double().sum()) for k, v in model.state_dict().items()
          if k.startswith(prefixes) and v.dtype.is_floating_point}
    d_step_error = None
    if adv:
        # main_missing.py:286-289 as written.  Under torch >= 1.5 the in-place
        # optimizer.step() above invalidates the retained graph and this raises;
        # the fixture records that fact (DESIGN.md "adversarial d-step").
        try:
            opt_d.zero_grad(); parts['adv_s_d'].backward(); opt_d.step()
        except RuntimeError as e:
            d_step_error = str(e).split('\n')[0][:160]

    meta = dict(B=B, M=M, H=160, W=192, drop=drop, adv=adv, lambdas=lam,
                loss=float(loss), parts={k: float(v) for k, v in parts.items()},
                grad_norm=gnorm, grad_norms=grad_norms, wsum_before=w0, wsum_after=w1,
                n_params_with_grad=len(grad_norms), torch=torch.__version__,
                d_step_reference_error=d_step_error)
    with open(os.path.join(OUT, f'step_{tag}.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    arrs = dict(mu=torch.stack(mu_list).detach().numpy(), z=torch.stack(z_list).detach().numpy(),
                lv=torch.stack(lv_list).detach().numpy(),
                s0_pool8=pool8(s_list[0].detach()), xf0_pool8=pool8(xf[0].detach()),
                xmix0_pool8=pool8(xmix[0].detach()), mask=mask.numpy())
    if y_list is not None:
        arrs['y0_pool8'] = pool8(y_list[0].detach()); arrs['y1_pool8'] = pool8(y_list[-1].detach())
    np.savez_compressed(os.path.join(OUT, f'step_{tag}.npz'), **arrs)
    print(f'step_{tag}: loss={float(loss):.7f} gnorm={gnorm:.4f}',
          {k: round(float(v), 7) for k, v in parts.items()})


def ref_loss(model, inputs, mask, mask_img, lam, M):
    """main_missing.py:166-251 for the shipped loss set (no adversarial / target terms)."""
    x_list = [inputs[:, i * 7:(i + 1) * 7] for i in range(M)]
    s_list = model.compute_anatomy_encoding(x_list, mask_img)
    z_list, mu_list, lv_list = model.compute_modality_encoding(x_list, s_list, phase='train')
    xf = model.reconstruct_input_si_zi(s_list, z_list)
    xmix = model.reconstruct_input_si_zj(s_list, z_list)
    parts = {'recon_x': model.compute_recon_loss_x_list(x_list, xf, mask, p=1),
             'recon_x_mix': model.compute_recon_loss_x_mix_list(x_list, xmix, mask, p=1)}
    s_new = model.compute_anatomy_encoding(xf, mask_img)
    _, mu_new, _ = model.compute_modality_encoding(xf, s_new, phase='train')
    parts['latent_z'] = model.compute_latent_z_loss(mu_list, mu_new, mask)
    parts['sim_s'] = model.compute_similarity_s_loss(s_list, mask)
    parts['sim_z'] = model.compute_similarity_z_loss(z_list, mask)
    loss = sum(lam[k] * parts[k] for k in parts)
    return loss, parts


def gen_accum(ref, tag='accum_b2m2', B=2, M=2, iters=4, batch_size=8):
    """The reference's DEFAULT optimizer schedule (config.yaml:17 `batch_size: 8` -> 16 // 8 = 2 micro-batches):
    main_missing.py:268-284 as written -- backward() accumulates, clip_grad_norm_ runs on the ACCUMULATING gradient every
    iteration (:272), optimizer.step() + zero_grad() on every second iteration (:282-284).  `iters` iterations = two
    optimizer steps on different micro-batches (seeds 10 + it), drop-off masks on the odd ones."""
    lam = dict(recon_x=1.0, recon_x_mix=2.0, latent_z=0.1, sim_s=10.0, sim_z=2.0)
    accum = 16 // batch_size
    torch.manual_seed(10); np.random.seed(10)
    model = build_ref_model(ref, M, False)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, weight_decay=1e-5, amsgrad=True)
    torch.manual_seed(11); np.random.seed(11)
    w0 = {k: float(v.double().sum()) for k, v in model.state_dict().items()
          if k.startswith(HOT_PREFIXES) and v.dtype.is_floating_point}
    rec = []
    for it in range(iters):
        inputs, mask, mask_img = make_inputs(B, M, 160, 192, seed=10 + it, drop=bool(it % 2))
        loss, parts = ref_loss(model, inputs, mask, mask_img, lam, M)
        loss.backward()                                                              # :268-271
        pre_clip = float(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0))    # :272 (norm BEFORE scaling)
        post_clip = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
        stepped = (it + 1) % accum == 0                                              # :282
        if stepped:
            opt.step(); opt.zero_grad()                                              # :283-284
        rec.append(dict(loss=float(loss), parts={k: float(v) for k, v in parts.items()}, grad_norm_before_clip=pre_clip,
                        grad_norm_after_clip=post_clip, stepped=stepped,
                        wsum={k: float(v.double().sum()) for k, v in model.state_dict().items()
                              if k.startswith(HOT_PREFIXES) and v.dtype.is_floating_point} if stepped else None))
    meta = dict(B=B, M=M, H=160, W=192, batch_size=batch_size, accum=accum, lambdas=lam, iters=rec, wsum_before=w0,
                torch=torch.__version__)
    with open(os.path.join(OUT, f'{tag}.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(tag, [(round(r['loss'], 6), round(r['grad_norm_before_clip'], 4), r['stepped']) for r in rec])


def gen_ckpt_layout(ref, tag='ckpt_layout_m2'):
    """Checkpoint dict of the reference after one epoch-end (main_missing.py:330-335) as a LAYOUT fixture: every
    state_dict key with shape / dtype / sum, the torch optimizer and ReduceLROnPlateau state_dict structure, and the
    stat.csv row format of util.py:854-866 -- data only, no tensors of 140 MB and no source text."""
    M = 2
    torch.manual_seed(10); np.random.seed(10)
    model = build_ref_model(ref, M, True)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, weight_decay=1e-5, amsgrad=True)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode='min', factor=0.1, patience=5, min_lr=1e-5)   # main_missing.py:119
    sd = model.state_dict()
    layout = {k: dict(shape=list(v.shape), dtype=str(v.dtype).replace('torch.', ''),
                      sum=float(v.double().sum()) if v.dtype.is_floating_point else int(v.sum())) for k, v in sd.items()}
    inputs, mask, mask_img = make_inputs(2, M, 160, 192, seed=10)
    torch.manual_seed(11); np.random.seed(11)
    loss, _ = ref_loss(model, inputs, mask, mask_img, dict(recon_x=1.0, recon_x_mix=2.0, latent_z=0.1, sim_s=10.0, sim_z=2.0), M)
    loss.backward(); torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); opt.step()
    monitor = [0.50, 0.40, 0.45, 0.41, 0.42, 0.43, 0.44, 0.46, 0.47, 0.39]
    lrs = []
    for v in monitor:                                                                # scheduler.step(monitor_metric), :322
        sched.step(v); lrs.append(opt.param_groups[0]['lr'])
    osd, ssd = opt.state_dict(), sched.state_dict()
    params = list(model.parameters())
    names = {id(p): n for n, p in model.named_parameters()}
    opt_layout = dict(param_group_keys=sorted(k for k in osd['param_groups'][0] if k != 'params'),
                      n_params=len(osd['param_groups'][0]['params']),
                      state_index_to_name={str(i): names[id(params[i])] for i in osd['state']},
                      state_entry_keys=sorted(next(iter(osd['state'].values())).keys()),
                      step_dtype=str(next(iter(osd['state'].values()))['step'].dtype).replace('torch.', ''),
                      step_shape=list(next(iter(osd['state'].values()))['step'].shape))
    sched_layout = {k: (v if isinstance(v, (int, float, str, bool, type(None))) else repr(type(v).__name__)) for k, v in ssd.items()}
    meta = dict(M=M, model=layout, optimizer=opt_layout, scheduler=sched_layout, monitor=monitor, lr_trajectory=lrs,
                ckpt_keys=['epoch', 'monitor_metric', 'stat', 'optimizer', 'scheduler', 'model', 'optimizer_d_s'],
                torch=torch.__version__)
    with open(os.path.join(OUT, f'{tag}.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(tag, len(layout), 'state_dict entries;', len(osd['state']), 'optimizer state entries; lr', lrs)


def gen_eval(ref, tag, B, M):
    """evaluate() of the reference for one batch (main_missing.py:337-517): model.eval(), z = mu."""
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    from oracle.ref_model import perturb_bn_running_stats
    torch.manual_seed(10); np.random.seed(10)
    model = build_ref_model(ref, M, False)
    perturb_bn_running_stats(model)
    model.eval()
    inputs, mask, mask_img = make_inputs(B, M, 160, 192, seed=12)
    torch.manual_seed(11); np.random.seed(11)
    x_list = [inputs[:, i * 7:(i + 1) * 7] for i in range(M)]
    with torch.no_grad():
        s_list = model.compute_anatomy_encoding(x_list, mask_img)
        z_list, mu_list, lv_list = model.compute_modality_encoding(x_list, s_list, phase='test')
        xf = model.reconstruct_input_si_zi(s_list, z_list)
        xmix = model.reconstruct_input_si_zj(s_list, z_list)
        parts = {'recon_x': model.compute_recon_loss_x_list(x_list, xf, mask, p=1),
                 'recon_x_mix': model.compute_recon_loss_x_mix_list(x_list, xmix, mask, p=1)}
        s_new = model.compute_anatomy_encoding(xf, mask_img)
        _, mu_new, _ = model.compute_modality_encoding(xf, s_new, phase='test')
        parts['latent_z'] = model.compute_latent_z_loss(mu_list, mu_new, mask)
        parts['sim_s'] = model.compute_similarity_s_loss(s_list, mask)
        parts['sim_z'] = model.compute_similarity_z_loss(z_list, mask)
        loss = 1.0 * parts['recon_x'] + 2.0 * parts['recon_x_mix'] + 0.1 * parts['latent_z'] + 10.0 * parts['sim_s'] + 2.0 * parts['sim_z']
    meta = dict(B=B, M=M, loss=float(loss), parts={k: float(v) for k, v in parts.items()}, torch=torch.__version__)
    with open(os.path.join(OUT, f'eval_{tag}.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, f'eval_{tag}.npz'), mu=torch.stack(mu_list).numpy(),
                        s0_pool8=pool8(s_list[0]), xf0_pool8=pool8(xf[0]), xmix0_pool8=pool8(xmix[0]))
    print(f'eval_{tag}: loss={float(loss):.7f}', {k: round(float(v), 7) for k, v in parts.items()})


# ---------------------------------------------------------------- batch assembly (util.py:444-566, 706)


def gen_data():
    """batches of the reference's own ZeroDoseDataset + DataLoader (shuffle, drop-off) over synthetic volumes."""
    import_reference()
    import util as ref_util          # noqa  (needs the same stubs as model.py)
    from torch.utils.data import DataLoader
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle.ref_data import synthetic_volumes
    c = DATA_CFG
    data = synthetic_volumes(c['n_subj'], c['contrasts'], c['H'], c['W'], c['D'], c['seed'], c['missing_every'])
    subj, idx = data_lists()
    ds = ref_util.ZeroDoseDataset('BraTS', data, np.array(subj), np.array(idx), None, block_size=3,
                                  contrast_list=c['contrasts'], aug=False, dropoff=True, skull_strip=False)
    ds.image_size = [c['H'], c['W']]                    # the class hard-codes 160x192 for its zero fill (util.py:462)
    np.random.seed(5); torch.manual_seed(7)
    out = {}
    for bi, batch in enumerate(DataLoader(ds, batch_size=4, shuffle=True, num_workers=0)):
        out[f'subj_{bi}'] = np.array(batch['subj_id'])
        out[f'slice_{bi}'] = batch['slice_idx'].numpy()
        out[f'mask_{bi}'] = batch['mask'].numpy()
        x = batch['inputs'].numpy().astype(np.float32)
        out[f'insum_{bi}'] = np.array([x.astype(np.float64).sum(), np.abs(x).astype(np.float64).sum()])
        out[f'mimg_{bi}'] = batch['mask_img'].numpy().astype(np.float32).sum((1, 2))
        out[f'tsum_{bi}'] = batch['targets'].numpy().astype(np.float64).sum((1, 2, 3))
        if bi < 2:
            out[f'inputs_{bi}'] = x
            out[f'targets_{bi}'] = batch['targets'].numpy().astype(np.float32)
    out['n_batches'] = np.array(bi + 1)
    np.savez_compressed(os.path.join(OUT, 'data_b4.npz'), **out)
    print('data_b4:', bi + 1, 'batches; dropped contrasts per batch', [int((out[f'mask_{k}'] == 0).sum()) for k in range(bi + 1)])


NV3D_CASES = {'a': dict(B=2, shape=(32, 16, 48), c=8), 'b': dict(B=2, shape=(16, 16, 16), c=16)}


def gen_nv3d(ref, tag):
    """NVNet3D (reference model.py:2050-2060) forward + gradients of oracle.ref_model3d.nvnet_loss; p = 0 (dropout's
    random mask has no cross-device counterpart), eps from the CPU generator seeded with 11."""
    sys.path.insert(0, ROOT)
    from oracle.ref_model3d import nvnet_loss, make_inputs3d
    cfg = NV3D_CASES[tag]
    torch.manual_seed(10); np.random.seed(10)
    model = ref.NVNet3D(cfg['shape'], in_channels=4, out_channels=3, init_channels=cfg['c'], p=0.0).train()
    w0 = {k: float(v.double().sum()) for k, v in model.state_dict().items()}
    x, t = make_inputs3d(cfg['B'], 4, cfg['shape'], seed=10)
    torch.manual_seed(11); np.random.seed(11)
    uout, vout, mu, logvar = model(x)
    loss, parts = nvnet_loss(uout, vout, mu, logvar, x, t)
    loss.backward()
    gn = {n: float(p.grad.double().norm()) for n, p in model.named_parameters() if p.grad is not None}
    gnorm = float(np.sqrt(sum(v * v for v in gn.values())))
    meta = dict(B=cfg['B'], shape=list(cfg['shape']), init_channels=cfg['c'], loss=float(loss),
                parts={k: float(v) for k, v in parts.items()}, grad_norm=gnorm, grad_norms=gn, wsum_before=w0,
                torch=torch.__version__)
    with open(os.path.join(OUT, f'nvnet3d_{tag}.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, f'nvnet3d_{tag}.npz'), mu=mu.detach().numpy(), logvar=logvar.detach().numpy(),
                        uout_pool4=F.avg_pool3d(uout.detach(), 4).numpy(), vout_pool4=F.avg_pool3d(vout.detach(), 4).numpy(),
                        uout_corner=uout.detach()[:, :, :4, :4, :4].numpy(),
                        g_conv1a=model.unet.conv1a.weight.grad.numpy(), g_ds2=model.unet.ds2.weight.grad[:8, :8].numpy())
    print(f'nvnet3d_{tag}: loss={float(loss):.7f} gnorm={gnorm:.5f}', {k: round(float(v), 7) for k, v in parts.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref = import_reference()
    only = sys.argv[1:]
    if not only or 'units' in only:
        gen_units(ref)
    if not only or 'steps' in only:
        gen_step(ref, 'b2m4', 2, 4)
        gen_step(ref, 'b4m2', 4, 2)
        gen_step(ref, 'b2m4_drop', 2, 4, drop=True)
    if not only or 'adv' in only:
        gen_step(ref, 'b2m2_adv', 2, 2, adv=True)
    if not only or 'recon_y' in only:
        gen_step(ref, 'b2m2_y', 2, 2, recon_y=True)
    if not only or 'accum' in only:
        gen_accum(ref)
    if not only or 'ckpt' in only:
        gen_ckpt_layout(ref)
    if not only or 'eval' in only:
        gen_eval(ref, 'b2m4', 2, 4)
    if not only or 'data' in only:
        gen_data()
    if not only or 'nv3d' in only:
        gen_nv3d(ref, 'a')
        gen_nv3d(ref, 'b')


if __name__ == '__main__':
    main()
