"""Pins the oracle (oracle/ref_model.py) against vectors produced by the real
reference (oracle/gen_golden.py, run in the build container)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_model as R
from fixtures import make_inputs, make_seg_targets, reinit_discriminator, seeded

TOL = dict(rtol=2e-5, atol=2e-6)


@pytest.fixture(scope='module')
def units(golden_dir):
    return np.load(os.path.join(golden_dir, 'units.npz'))


@pytest.mark.parametrize('name,ci,co,k,s,p,hw', [
    ('c3s1', 5, 6, 3, 1, 1, (9, 11)), ('c4s2', 7, 8, 4, 2, 1, (12, 10)),
    ('c3s2', 4, 6, 3, 2, 1, (11, 13)), ('c1s1', 6, 3, 1, 1, 0, (7, 5))])
def test_condconv_per_sample(units, name, ci, co, k, s, p, hw):
    torch.manual_seed(100)
    m = R.RefCondConv2d(ci, co, k, s, p)
    with torch.no_grad():
        m.bias.copy_(seeded((co,), 7, 0.1))
    x = seeded((3, ci) + hw, 1).requires_grad_(True)
    t = torch.tensor([[1.], [2.], [4.]])
    y = m(x, t)
    y.backward(seeded(tuple(y.shape), 2))
    np.testing.assert_allclose(y.detach().numpy(), units[f'cond_{name}_y'], **TOL)
    np.testing.assert_allclose(x.grad.numpy(), units[f'cond_{name}_dx'], **TOL)
    np.testing.assert_allclose(m.weight.grad.numpy(), units[f'cond_{name}_dw'], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(m.bias.grad.numpy(), units[f'cond_{name}_db'], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(m._routing_fn.fc.weight.grad.numpy(), units[f'cond_{name}_dfcw'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m._routing_fn.fc.bias.grad.numpy(), units[f'cond_{name}_dfcb'], rtol=1e-4, atol=1e-5)


def test_conv_bn_act(units):
    torch.manual_seed(101)
    m = R.RefConvBNAct(6, 8).train()
    x = seeded((3, 6, 12, 16), 3).requires_grad_(True)
    y = m(x, 2 * torch.ones(3, 1)); y.backward(seeded(tuple(y.shape), 4))
    np.testing.assert_allclose(y.detach().numpy(), units['cba_y'], **TOL)
    np.testing.assert_allclose(x.grad.numpy(), units['cba_dx'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.conv.weight.grad.numpy(), units['cba_dw'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.bn.weight.grad.numpy(), units['cba_dbn_w'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.bn.running_mean.numpy(), units['cba_run_mean'], **TOL)
    np.testing.assert_allclose(m.bn.running_var.numpy(), units['cba_run_var'], **TOL)


def test_up_conv_bn_cat(units):
    torch.manual_seed(102)
    m = R.RefUpConvBNCat(6, 5).train()
    xu = seeded((2, 6, 5, 6), 5).requires_grad_(True)
    xd = seeded((2, 4, 10, 12), 6)
    y = m(xd, xu, 3 * torch.ones(2, 1)); y.backward(seeded(tuple(y.shape), 7))
    np.testing.assert_allclose(y.detach().numpy(), units['adb_y'], **TOL)
    np.testing.assert_allclose(xu.grad.numpy(), units['adb_dx'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.conv.weight.grad.numpy(), units['adb_dw'], rtol=1e-4, atol=1e-5)


def test_spade_block(units):
    torch.manual_seed(103)
    m = R.RefSPADEBlock((10, 12), 8, 6, 4)
    s = torch.softmax(seeded((2, 4, 40, 48), 8), 1).requires_grad_(True)
    z = seeded((2, 8, 10, 12), 9).requires_grad_(True)
    y = m(s, z, torch.ones(2, 1)); y.backward(seeded(tuple(y.shape), 10))
    np.testing.assert_allclose(y.detach().numpy(), units['spade_y'], **TOL)
    np.testing.assert_allclose(s.grad.numpy(), units['spade_ds'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(z.grad.numpy(), units['spade_dz'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(m.gamma.weight.grad.numpy(), units['spade_dw_gamma'], rtol=1e-4, atol=1e-5)


def test_modality_encoder(units):
    torch.manual_seed(104)
    m = R.RefModalityEnc(7, 16, 16, 30)
    x = seeded((2, 7, 160, 192), 11)
    mu, lv = m(x, 2 * torch.ones(2, 1))
    (mu.sum() + 2 * lv.sum()).backward()
    np.testing.assert_allclose(mu.detach().numpy(), units['modenc_mu'], **TOL)
    np.testing.assert_allclose(lv.detach().numpy(), units['modenc_lv'], **TOL)
    np.testing.assert_allclose(m.conv1.weight.grad.numpy(), units['modenc_dw1'], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('pg', [False, True])
def test_discriminator(units, pg):
    torch.manual_seed(105)
    m = R.RefDiscriminator(4, 16, (160, 192), pg).train()
    x = torch.softmax(seeded((2, 4, 160, 192), 12), 1)
    np.testing.assert_allclose(m(x).detach().numpy(), units[f'disc_{"patch" if pg else "dense"}_y'], **TOL)


def _run_step(golden_dir, tag):
    meta = json.load(open(os.path.join(golden_dir, f'step_{tag}.json')))
    arrs = np.load(os.path.join(golden_dir, f'step_{tag}.npz'))
    B, M, adv = meta['B'], meta['M'], meta['adv']
    ry = meta['lambdas'].get('recon_y', 0.0) > 0
    torch.manual_seed(10); np.random.seed(10)
    model = R.RefMultimodalModel((160, 192), M, is_discrim_s=adv, out_num_ch=4 if ry else 0).train()
    if adv:
        reinit_discriminator(model.discrim_s)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, weight_decay=1e-5, amsgrad=True)
    inputs, mask, mask_img = make_inputs(B, M, 160, 192, seed=10, drop=meta['drop'])
    torch.manual_seed(11); np.random.seed(11)
    w0 = {k: float(v.double().sum()) for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    for k, v in meta['wsum_before'].items():       # identical init under the seed
        assert abs(w0[k] - v) <= 1e-9 * max(1, abs(v)), k
    targets = make_seg_targets(B, 160, 192, seed=13) if ry else None
    loss, parts, aux = R.ref_forward_losses(model, inputs, mask, mask_img, meta['lambdas'], targets=targets)
    loss.backward(retain_graph=adv)
    gn = {n: float(p.grad.double().norm()) for n, p in model.named_parameters() if p.grad is not None}
    gnorm = float(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0))
    opt.step()
    return meta, arrs, model, loss, parts, aux, gn, gnorm


@pytest.mark.parametrize('tag', ['b2m4', 'b4m2', 'b2m4_drop', 'b2m2_adv', 'b2m2_y'])
def test_full_step(golden_dir, tag):
    meta, arrs, model, loss, parts, aux, gn, gnorm = _run_step(golden_dir, tag)
    assert abs(float(loss) - meta['loss']) <= 2e-5 * abs(meta['loss'])
    for k, v in meta['parts'].items():
        assert abs(float(parts[k]) - v) <= 2e-5 * abs(v) + 1e-7, k
    np.testing.assert_allclose(torch.stack(aux['mu_list']).detach().numpy(), arrs['mu'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(torch.stack(aux['z_list']).detach().numpy(), arrs['z'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(F.avg_pool2d(aux['s_list'][0].detach(), 8).numpy(), arrs['s0_pool8'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(F.avg_pool2d(aux['xf'][0].detach(), 8).numpy(), arrs['xf0_pool8'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(F.avg_pool2d(aux['xmix'][0].detach(), 8).numpy(), arrs['xmix0_pool8'], rtol=1e-4, atol=1e-5)
    assert abs(gnorm - meta['grad_norm']) <= 1e-3 * meta['grad_norm']
    # same set of parameters receives a gradient, same per-tensor norms
    with_y = meta['lambdas'].get('recon_y', 0.0) > 0                  # output decoder ('U+SA') on the path: its tensors are checked too
    if with_y:
        np.testing.assert_allclose(F.avg_pool2d(aux['y_list'][0].detach(), 8).numpy(), arrs['y0_pool8'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(F.avg_pool2d(aux['y_list'][-1].detach(), 8).numpy(), arrs['y1_pool8'], rtol=1e-4, atol=1e-5)
    hot = {k: v for k, v in meta['grad_norms'].items() if with_y or not k.startswith('output_decoder')}
    assert set(hot) == set(gn)
    for k, v in hot.items():
        assert abs(gn[k] - v) <= 2e-3 * v + 1e-6 * meta['grad_norm'], (k, gn[k], v)
    w1 = {k: float(v.double().sum()) for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    for k, v in meta['wsum_after'].items():
        # conv biases that feed a BatchNorm / InstanceNorm have an analytically zero gradient:
        # what arrives is rounding noise, whose sign Adam's first step turns into +-lr.
        if meta['grad_norms'].get(k, 1.0) < 1e-5 * meta['grad_norm']:
            continue
        assert abs(w1[k] - v) <= 5e-5 * max(1.0, abs(v)), (k, w1[k], v)


def test_accumulation_schedule(golden_dir):
    """the reference's default optimizer schedule (config.yaml batch_size 8 -> two micro-batches per optimizer step, clip
    on the accumulating gradient every iteration, main_missing.py:268-284): oracle vs tests/golden/accum_b2m2.json."""
    meta = json.load(open(os.path.join(golden_dir, 'accum_b2m2.json')))
    B, M = meta['B'], meta['M']
    torch.manual_seed(10); np.random.seed(10)
    model = R.RefMultimodalModel((160, 192), M).train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, weight_decay=1e-5, amsgrad=True)
    torch.manual_seed(11); np.random.seed(11)
    batches = [make_inputs(B, M, 160, 192, seed=10 + it, drop=bool(it % 2)) for it in range(len(meta['iters']))]
    wsums = {}

    def snap(it, rec):
        if rec['stepped']:
            wsums[it] = {k: float(v.double().sum()) for k, v in model.state_dict().items() if k in meta['wsum_before']}
    got = R.ref_train_iterations(model, opt, batches, meta['batch_size'], meta['lambdas'], on_iter=snap)
    for it, (g, want) in enumerate(zip(got, meta['iters'])):
        assert g['stepped'] == want['stepped']
        assert abs(g['loss'] - want['loss']) <= 5e-5 * abs(want['loss']), it
        assert abs(g['grad_norm_before_clip'] - want['grad_norm_before_clip']) <= 1e-3 * want['grad_norm_before_clip'], it
        if want['stepped']:
            bad = [k for k, v in want['wsum'].items() if abs(wsums[it][k] - v) > 1e-4 * max(1.0, abs(v))]
            assert len(bad) <= 0.02 * len(want['wsum']), (it, bad[:5])    # noise-level gradients may flip Adam's first +-lr move


def test_evaluate_batch(golden_dir):
    """evaluate() of the reference (model.eval(), z = mu) for one batch."""
    meta = json.load(open(os.path.join(golden_dir, 'eval_b2m4.json')))
    arrs = np.load(os.path.join(golden_dir, 'eval_b2m4.npz'))
    torch.manual_seed(10); np.random.seed(10)
    model = R.RefMultimodalModel((160, 192), meta['M'])
    R.perturb_bn_running_stats(model)
    inputs, mask, mask_img = make_inputs(meta['B'], meta['M'], 160, 192, seed=12)
    torch.manual_seed(11); np.random.seed(11)
    loss, parts, aux = R.ref_evaluate_batch(model, inputs, mask, mask_img)
    assert abs(float(loss) - meta['loss']) <= 2e-5 * abs(meta['loss'])
    for k, v in meta['parts'].items():
        assert abs(float(parts[k]) - v) <= 2e-5 * abs(v) + 1e-7, k
    np.testing.assert_allclose(torch.stack(aux['mu_list']).numpy(), arrs['mu'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(F.avg_pool2d(aux['xmix'][0], 8).numpy(), arrs['xmix0_pool8'], rtol=1e-4, atol=1e-5)
    assert model.training            # restored


def test_reconstruction_metrics_definitions():
    """util.py:935-978 restated (skimage absent): check against brute-force definitions."""
    rng = np.random.RandomState(3)
    t = rng.randn(2, 3, 20, 24).astype(np.float32); t[:, :, :4] = -10
    p = (t + 0.3 * rng.randn(*t.shape)).astype(np.float32)
    m = R.ref_reconstruction_metrics(t, p)
    for i in range(2):
        a = t[i, 0].astype(np.float64) - t[i, 0].min(); b = p[i, 0].astype(np.float64) - p[i, 0].min()
        Rg = a.max(); C1, C2 = (0.01 * Rg) ** 2, (0.03 * Rg) ** 2
        acc = []
        for y in range(3, 17):
            for x in range(3, 21):
                wa, wb = a[y - 3:y + 4, x - 3:x + 4].ravel(), b[y - 3:y + 4, x - 3:x + 4].ravel()
                ua, ub = wa.mean(), wb.mean()
                va, vb = wa.var(ddof=1), wb.var(ddof=1)
                vab = ((wa - ua) * (wb - ub)).sum() / 48
                acc.append((2 * ua * ub + C1) * (2 * vab + C2) / ((ua * ua + ub * ub + C1) * (va + vb + C2)))
        assert abs(m['ssim'][i] - np.mean(acc)) < 1e-9
        assert abs(m['rmse'][i] - ((a - b) ** 2).mean()) < 1e-12
        assert abs(m['psnr'][i] - 10 * np.log10(Rg ** 2 / ((a - b) ** 2).mean())) < 1e-9
    same = R.ref_reconstruction_metrics(t, t + 5.0)          # min-shift invariance: identical after the shift
    assert all(abs(v - 1) < 1e-6 for v in same['ssim'])


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_nvnet3d_oracle_vs_reference_golden(golden_dir, tag):
    """oracle/ref_model3d.py against vectors of the real reference NVNet3D (model.py:2050-2060), SURVEY 8(f).2."""
    from oracle import ref_model3d as R3
    meta = json.load(open(os.path.join(golden_dir, f'nvnet3d_{tag}.json')))
    arrs = np.load(os.path.join(golden_dir, f'nvnet3d_{tag}.npz'))
    shape = tuple(meta['shape'])
    torch.manual_seed(10); np.random.seed(10)
    model = R3.RefNVNet3D(shape, 4, 3, meta['init_channels'], p=0.0).train()
    assert set(model.state_dict()) == set(meta['wsum_before'])
    for k, v in meta['wsum_before'].items():
        assert abs(float(model.state_dict()[k].double().sum()) - v) <= 1e-9 * max(1.0, abs(v)), k
    x, t = R3.make_inputs3d(meta['B'], 4, shape, seed=10)
    torch.manual_seed(11); np.random.seed(11)
    uout, vout, mu, logvar = model(x)
    loss, parts = R3.nvnet_loss(uout, vout, mu, logvar, x, t)
    loss.backward()
    assert abs(float(loss) - meta['loss']) <= 2e-5 * abs(meta['loss'])
    np.testing.assert_allclose(mu.detach().numpy(), arrs['mu'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(F.avg_pool3d(uout.detach(), 4).numpy(), arrs['uout_pool4'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(F.avg_pool3d(vout.detach(), 4).numpy(), arrs['vout_pool4'], rtol=1e-4, atol=1e-6)
    gn = {n: float(p.grad.double().norm()) for n, p in model.named_parameters() if p.grad is not None}
    assert set(gn) == set(meta['grad_norms'])
    for k, v in meta['grad_norms'].items():
        assert abs(gn[k] - v) <= 1e-3 * v + 1e-6 * meta['grad_norm'], k
