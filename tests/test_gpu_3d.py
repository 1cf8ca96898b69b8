"""GPU parity of the 3-D path: the Conv3d / GroupNorm+ReLU / nearest-upsample kernels against
torch fp32 on the CPU, and NVNet3D (a) against vectors captured from the real reference and (b) against the CPU oracle
at another size.  Tolerance: 1e-3 relative (fp32, BASELINE.json north_star)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_model3d as R3

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def seeded(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def cl3(x):
    return x.to(DEV).contiguous(memory_format=torch.channels_last_3d)


def close(got, want, rtol=1e-3, what=''):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().float().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = float(np.abs(want).max()) + 1e-12
    err = float(np.abs(got - want).max())
    assert err <= rtol * scale, f'{what}: max abs err {err:.3e} vs scale {scale:.3e}'


@pytest.mark.parametrize('N,Ci,Co,D,H,W,stride', [
    (2, 4, 16, 8, 16, 16, 1),        # conv1a
    (1, 16, 16, 16, 16, 16, 1),      # BasicBlock at the first level (CW = 16, two taps per sub-tile)
    (2, 8, 8, 6, 10, 12, 1),         # init_channels = 8, ragged boxes
    (1, 16, 32, 16, 16, 16, 2),      # ds1
    (2, 32, 64, 8, 8, 8, 2),         # ds2
    (1, 64, 128, 4, 8, 8, 2),        # ds3
    (2, 128, 128, 2, 2, 2, 1),       # bottom blocks
    (1, 128, 64, 4, 4, 4, 1),        # vconv3 / hidden_conv
    (1, 32, 32, 5, 7, 9, 1),         # odd extents
    (1, 16, 32, 7, 9, 11, 2),        # odd extents, stride 2
    (2, 32, 32, 1, 1, 1, 1),         # 1^3 volumes (input 16^3 at the bottom)
    (1, 32, 16, 6, 8, 10, 1),        # vconv1: narrow kernels with two 16-channel input slices
    (1, 16, 16, 9, 9, 9, 1),         # narrow kernels, ragged boxes, several boxes per workgroup
])
def test_conv3d_fwd_bwd(mrdis, N, Ci, Co, D, H, W, stride):
    x = seeded((N, Ci, D, H, W), 1)
    w = seeded((Co, Ci, 3, 3, 3), 2, 0.1)
    b = seeded((Co,), 3)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, br, stride=stride, padding=1)
    res = seeded(tuple(y_ref.shape), 4) if stride == 1 and Ci == Co else None
    if res is not None:
        y_ref = y_ref + res
    dy = seeded(tuple(y_ref.shape), 5)
    y_ref.backward(dy)

    conv = mrdis.HipConv3d(Ci, Co, (3, 3, 3), stride=(stride,) * 3, padding=(1, 1, 1)).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w); conv.bias.copy_(b)
    xg = cl3(x).requires_grad_(True)
    y = conv(xg, residual=None if res is None else cl3(res))
    close(y, y_ref, 1e-3, 'fwd')
    y.backward(cl3(dy))
    close(xg.grad, xr.grad, 1e-3, 'dgrad')
    close(conv.weight.grad, wr.grad, 1e-3, 'wgrad')
    close(conv.bias.grad, br.grad, 1e-3, 'bgrad')


@pytest.mark.parametrize('N,Ci,Co,D,H,W', [(1, 32, 32, 6, 20, 24), (2, 64, 64, 3, 9, 11), (1, 16, 48, 5, 16, 16), (1, 128, 32, 2, 8, 8)])
def test_conv3d_winograd_hybrid_forced(mrdis, N, Ci, Co, D, H, W):
    """3x3x3 stride-1 layers through the hybrid kernel (Winograd F(2x2,3x3) in (h, w), direct in depth; mrdis_wino.hip D3)
    forced on for shapes the size policy would leave to the direct kernel: forward with bias + fused residual, data
    gradient (reversed 27-tap filter), weight gradient (one 2-D Winograd launch per depth tap where Ci / Co are 32 / 64
    multiples), planes at the volume boundary, odd extents, cout / channel tails."""
    x = seeded((N, Ci, D, H, W), 1); w = seeded((Co, Ci, 3, 3, 3), 2, 0.1); b = seeded((Co,), 3)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, br, padding=1)
    res = seeded(tuple(y_ref.shape), 4)
    y_ref = y_ref + res
    dy = seeded(tuple(y_ref.shape), 5)
    y_ref.backward(dy)
    conv = mrdis.HipConv3d(Ci, Co, (3, 3, 3), padding=(1, 1, 1)).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w); conv.bias.copy_(b)
    outs = {}
    for mode in ('0', '2'):
        mrdis.hip.set_option('wino', int(mode))
        xg = cl3(x).requires_grad_(True)
        y = conv(xg, residual=cl3(res))
        y.backward(cl3(dy))
        outs[mode] = (y.detach(), xg.grad.detach())
        close(y, y_ref, 1e-3, f'fwd mode {mode}'); close(xg.grad, xr.grad, 1e-3, f'dgrad mode {mode}')
        close(conv.weight.grad, wr.grad, 1e-3, f'wgrad mode {mode}'); close(conv.bias.grad, br.grad, 1e-3, f'bgrad mode {mode}')
        conv.zero_grad()
    assert not torch.equal(outs['0'][0], outs['2'][0])          # really another kernel
    close(outs['2'][0], outs['0'][0], 1e-4, 'hybrid vs direct fwd'); close(outs['2'][1], outs['0'][1], 1e-4, 'hybrid vs direct dgrad')


def test_conv3d_pointwise(mrdis):
    x = seeded((2, 32, 4, 6, 8), 1); w = seeded((16, 32, 1, 1, 1), 2, 0.2); b = seeded((16,), 3)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, b)
    dy = seeded(tuple(y_ref.shape), 4)
    y_ref.backward(dy)
    conv = mrdis.HipConv3d(32, 16, (1, 1, 1)).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w); conv.bias.copy_(b)
    xg = cl3(x).requires_grad_(True)
    y = conv(xg)
    close(y, y_ref, 1e-3, 'fwd')
    y.backward(cl3(dy))
    close(xg.grad, xr.grad, 1e-3, 'dgrad'); close(conv.weight.grad, wr.grad, 1e-3, 'wgrad')


@pytest.mark.parametrize('N,C,D,H,W,relu', [(2, 16, 8, 8, 8, True), (1, 8, 5, 7, 9, True), (2, 128, 2, 2, 2, True),
                                             (1, 64, 33, 17, 9, False), (2, 32, 1, 1, 1, True)])
def test_groupnorm_relu(mrdis, N, C, D, H, W, relu):
    x = seeded((N, C, D, H, W), 1) * 2 + 0.7
    g = seeded((C,), 2) * 0.3 + 1; b = seeded((C,), 3) * 0.2
    xr = x.clone().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    y_ref = F.group_norm(xr, 8, gr, br, 1e-5)
    if relu:
        y_ref = F.relu(y_ref)
    dy = seeded(tuple(x.shape), 4)
    y_ref.backward(dy)
    gn = torch.nn.GroupNorm(8, C).to(DEV)
    with torch.no_grad():
        gn.weight.copy_(g); gn.bias.copy_(b)
    xg = cl3(x).requires_grad_(True)
    y = mrdis.model3d.groupnorm_relu(xg, gn, relu)
    close(y, y_ref, 1e-4, 'fwd')
    y.backward(cl3(dy))
    close(xg.grad, xr.grad, 1e-3, 'dx'); close(gn.weight.grad, gr.grad, 1e-3, 'dgamma'); close(gn.bias.grad, br.grad, 1e-3, 'dbeta')


@pytest.mark.parametrize('skip', [False, True])
def test_upsample2x(mrdis, skip):
    x = seeded((2, 16, 3, 5, 4), 1)
    s = seeded((2, 16, 6, 10, 8), 2) if skip else None
    xr = x.clone().requires_grad_(True)
    y_ref = F.interpolate(xr, scale_factor=2) + (s if skip else 0)
    dy = seeded(tuple(y_ref.shape), 3)
    y_ref.backward(dy)
    xg = cl3(x).requires_grad_(True)
    sg = cl3(s).requires_grad_(True) if skip else None
    y = mrdis.model3d.upsample2x(xg, sg)
    assert torch.equal(y.cpu(), y_ref.detach())
    y.backward(cl3(dy))
    close(xg.grad, xr.grad, 1e-5, 'dx')
    if skip:
        assert torch.equal(sg.grad.cpu(), dy)


def _grads(model):
    return {n: p.grad for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_nvnet3d_golden(mrdis, golden_dir, tag):
    """forward + gradients vs vectors of the real reference (oracle/gen_golden.py nv3d)."""
    meta = json.load(open(os.path.join(golden_dir, f'nvnet3d_{tag}.json')))
    arrs = np.load(os.path.join(golden_dir, f'nvnet3d_{tag}.npz'))
    shape = tuple(meta['shape'])
    torch.manual_seed(10); np.random.seed(10)
    model = mrdis.NVNet3D(shape, 4, 3, meta['init_channels'], p=0.0).train()
    assert set(model.state_dict()) == set(meta['wsum_before'])
    for k, v in meta['wsum_before'].items():
        assert abs(float(model.state_dict()[k].double().sum()) - v) <= 1e-9 * max(1.0, abs(v)), k
    model = model.to(DEV)
    x, t = R3.make_inputs3d(meta['B'], 4, shape, seed=10)
    torch.manual_seed(11); np.random.seed(11)
    xg = cl3(x)
    uout, vout, mu, logvar = model(xg)
    loss, parts = mrdis.nvnet_loss(uout, vout, mu, logvar, xg, cl3(t))
    loss.backward()
    assert abs(float(loss) - meta['loss']) <= 1e-3 * abs(meta['loss'])
    for k, v in meta['parts'].items():
        assert abs(float(parts[k]) - v) <= 1e-3 * abs(v) + 1e-6, k
    close(mu, arrs['mu'], 1e-3, 'mu'); close(logvar, arrs['logvar'], 1e-3, 'logvar')
    close(F.avg_pool3d(uout, 4), arrs['uout_pool4'], 1e-3, 'uout'); close(F.avg_pool3d(vout, 4), arrs['vout_pool4'], 1e-3, 'vout')
    close(uout[:, :, :4, :4, :4], arrs['uout_corner'], 1e-3, 'uout corner')
    close(model.unet.conv1a.weight.grad, arrs['g_conv1a'], 2e-3, 'g conv1a')
    close(model.unet.ds2.weight.grad[:8, :8], arrs['g_ds2'], 2e-3, 'g ds2')
    gn = {n: float(g.double().norm()) for n, g in _grads(model).items()}
    assert set(gn) == set(meta['grad_norms'])
    total = float(np.sqrt(sum(v * v for v in gn.values())))
    assert abs(total - meta['grad_norm']) <= 1e-3 * meta['grad_norm']
    for k, v in meta['grad_norms'].items():
        assert abs(gn[k] - v) <= 5e-3 * v + 2e-5 * meta['grad_norm'], (k, gn[k], v)


def test_nvnet3d_vs_oracle_other_size(mrdis):
    shape, c, B = (16, 32, 32), 16, 1
    torch.manual_seed(3)
    ref = R3.RefNVNet3D(shape, 4, 3, c, p=0.0).train()
    model = mrdis.NVNet3D(shape, 4, 3, c, p=0.0).train()
    model.load_state_dict(ref.state_dict())
    model = model.to(DEV)
    x, t = R3.make_inputs3d(B, 4, shape, seed=5)
    torch.manual_seed(7)
    out_r = ref(x)
    loss_r, _ = R3.nvnet_loss(*out_r, x, t)
    loss_r.backward()
    torch.manual_seed(7)
    xg = cl3(x)
    out = model(xg)
    loss, _ = mrdis.nvnet_loss(*out, xg, cl3(t))
    loss.backward()
    assert abs(float(loss) - float(loss_r)) <= 1e-3 * abs(float(loss_r))
    for a, b, n in zip(out, out_r, ('uout', 'vout', 'mu', 'logvar')):
        close(a, b, 1e-3, n)
    gr = _grads(ref)
    for n, g in _grads(model).items():
        close(g, gr[n], 3e-3, n)
