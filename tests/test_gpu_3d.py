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


# ---------------------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] at its own size: 4 x 4 x 128^3 volumes, init_channels 16 (tools/bench3d.py times exactly this).  The 2-D path needed tests
# at the benchmarked scale to catch scale-only defects (grid policies, 32-bit offsets, persistent-workgroup walks); so does the 3-D one.
def _geoms128(c=16, S=128):
    g = [('conv1a', 4, c, S, 1)]
    for lvl, mult in enumerate((1, 2, 4, 8)):
        s = S >> lvl
        g.append((f'block{lvl + 1}', c * mult, c * mult, s, 1))
        if lvl < 3:
            g.append((f'ds{lvl + 1}', c * mult, c * mult * 2, s, 2))
    for lvl, mult in enumerate((8, 4, 2)):
        g.append((f'vconv{3 - lvl}', c * mult, c * mult // 2, S >> (3 - lvl), 1))
    g.append(('hidden_conv', c * 8, c * 4, S >> 3, 1))
    return g


@pytest.mark.parametrize('name,ci,co,s,st', _geoms128(), ids=[g[0] for g in _geoms128()])
def test_conv3d_geometries_at_config4_size(mrdis, name, ci, co, s, st):
    """every distinct Conv3d geometry of NVNet3D (model.py:1856-2060) at B = 4, 128^3 input, as the step issues it (default kernel policy): forward,
    data gradient and weight gradient on the FULL tensors against (a) torch fp32 on the CPU on sampled sub-volumes -- two boxes, one at the
    volume's corner (zero padding on three faces) and one in the last image's far corner (the largest offsets) -- and (b) the direct kernels
    (wino = 0) on the full tensors."""
    hip = mrdis.hip
    B = 4
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(B, s, s, s, ci, device=DEV, generator=gen).permute(0, 4, 1, 2, 3)
    w = torch.randn(co, ci, 3, 3, 3, device=DEV, generator=gen) * 0.1
    bias = torch.randn(co, device=DEV, generator=gen)
    w_tck, w_tkc = hip.mix_experts_fwd(w.reshape(1, co, ci, 27, 1), torch.ones(1, device=DEV))
    so = (s - 1) // st + 1
    dy = torch.randn(B, so, so, so, co, device=DEV, generator=gen).permute(0, 4, 1, 2, 3)
    y = hip.conv3d_fwd(x, w_tck, bias, 3, st, 1)
    dx = hip.conv3d_bwd_data(dy, w_tkc, tuple(x.shape), 3, st, 1)
    # weight gradient through a cotangent that is zero outside two sampled boxes (every workgroup still walks the whole volume)
    e = min(so, 6)                                                    # box edge in output positions
    boxes = [(0, 0), (B - 1, so - e)]
    dyz = torch.zeros_like(dy)
    for n, o in boxes:
        dyz[n, :, o:o + e, o:o + e, o:o + e] = dy[n, :, o:o + e, o:o + e, o:o + e]
    dw, db = hip.conv3d_bwd_weight(x, dyz, 3, st, 1, True)
    wc, bc = w.cpu(), bias.cpu()
    dw_ref = torch.zeros_like(wc); db_ref = torch.zeros_like(bc)
    for n, o in boxes:
        # input box that feeds output positions [o, o + e): rows st * o - 1 .. st * (o + e - 1) + 1, clipped to the volume, explicit zero padding
        lo, hi = st * o - 1, st * (o + e - 1) + 1
        a, b_ = max(lo, 0), min(hi, s - 1)
        xb = x[n:n + 1, :, a:b_ + 1, a:b_ + 1, a:b_ + 1].cpu()
        pad = (a - lo, hi - b_) * 3
        xb = F.pad(xb, pad).requires_grad_(True)
        wr = wc.clone().requires_grad_(True); br = bc.clone().requires_grad_(True)
        yb = F.conv3d(xb, wr, br, stride=st)
        assert yb.shape[2] == e
        close(y[n:n + 1, :, o:o + e, o:o + e, o:o + e], yb, 1e-3, f'{name} fwd box {n}')
        dyb = dy[n:n + 1, :, o:o + e, o:o + e, o:o + e].cpu()
        yb.backward(dyb)
        dw_ref += wr.grad; db_ref += br.grad
        # data gradient: positions of x whose every reader lies inside the box (the box's interior) see only the box's cotangent
        if e >= 3:
            full = F.conv_transpose3d(dy[n:n + 1, :, max(o - 2, 0):o + e + 2, max(o - 2, 0):o + e + 2, max(o - 2, 0):o + e + 2].cpu(), wc, None, stride=st, padding=0)
            # `full` is indexed from input row st * max(o - 2, 0) - 1; compare a 2^3 block of x rows well inside
            base = st * max(o - 2, 0) - 1
            r0 = st * (o + 1)
            blk = full[:, :, r0 - base:r0 - base + 2, r0 - base:r0 - base + 2, r0 - base:r0 - base + 2]
            close(dx[n:n + 1, :, r0:r0 + 2, r0:r0 + 2, r0:r0 + 2], blk, 1e-3, f'{name} dgrad box {n}')
    close(dw.cpu().reshape(27, ci, co), dw_ref.permute(2, 3, 4, 1, 0).reshape(27, ci, co), 2e-3, f'{name} wgrad')
    close(db, db_ref, 2e-3, f'{name} bgrad')
    hip.set_option('wino', 0)
    y0 = hip.conv3d_fwd(x, w_tck, bias, 3, st, 1)
    dx0 = hip.conv3d_bwd_data(dy, w_tkc, tuple(x.shape), 3, st, 1)
    dw0, db0 = hip.conv3d_bwd_weight(x, dy, 3, st, 1, True)
    hip.set_option('wino', 1)
    dw1, db1 = hip.conv3d_bwd_weight(x, dy, 3, st, 1, True)
    close(y, y0, 1e-4, f'{name} fwd vs direct'); close(dx, dx0, 1e-4, f'{name} dgrad vs direct')
    close(dw1, dw0, 5e-4, f'{name} wgrad vs direct'); close(db1, db0, 5e-4, f'{name} bgrad vs direct')


def test_nvnet3d_step_at_config4_size(mrdis):
    """one NVNet3D training step (forward, nvnet_loss, backward; p = 0) on 4 x 4 x 128^3 volumes under the default policy against the direct kernels
    only (wino = 0): loss, loss parts, the four outputs and every parameter gradient."""
    S, B, c = 128, 4, 16
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, S, S, S, 4, generator=g).to(DEV).permute(0, 4, 1, 2, 3)
    t = (torch.rand(B, S, S, S, 3, generator=g) > 0.7).float().to(DEV).permute(0, 4, 1, 2, 3)
    res = {}
    for mode in (1, 0):
        mrdis.hip.set_option('wino', mode)
        torch.manual_seed(10)
        model = mrdis.NVNet3D((S, S, S), 4, 3, c, p=0.0).to(DEV).train()
        torch.manual_seed(11)
        out = model(x)
        loss, parts = mrdis.nvnet_loss(*out, x, t)
        loss.backward()
        res[mode] = (float(loss), {k: float(v) for k, v in parts.items()}, [F.avg_pool3d(o, 8).detach() if o.dim() == 5 else o.detach() for o in out],
                     {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
        del model, out, loss
        torch.cuda.empty_cache()
    mrdis.hip.set_option('wino', 1)
    (l1, p1, o1, g1), (l0, p0, o0, g0) = res[1], res[0]
    assert np.isfinite(l1) and abs(l1 - l0) <= 1e-4 * abs(l0)
    for k in p0:
        assert abs(p1[k] - p0[k]) <= 2e-4 * abs(p0[k]) + 1e-7, k
    for a, b in zip(o1, o0):
        close(a, b, 1e-4, 'outputs')
    assert set(g0) == set(g1) and len(g0) > 100
    assert any(not torch.equal(g0[n], g1[n]) for n in g0), 'the default policy did not change any kernel'
    tot = float(torch.sqrt(sum((v.double() ** 2).sum() for v in g0.values())))
    for n in g0:
        err = float((g1[n] - g0[n]).double().norm())
        assert err <= 2e-3 * float(g0[n].double().norm()) + 2e-5 * tot, (n, err)


@pytest.mark.parametrize('N,D,H,W', [(2, 64, 64, 64), (4, 50, 44, 70), (1, 36, 128, 112)])
def test_conv3d_16_to_16_six_product_kernel(mrdis, N, D, H, W):
    """conv3d16_s6_kernel / wgrad3d16_s6_kernel (mrdis_conv3d_s6.hip, option split6): the BasicBlock convolutions (16 -> 16, 3x3x3, stride 1: model.py:1861-1864)
    with both fp32 operands as three bf16 terms and the six products of order <= 2 on v_mfma_f32_16x16x32_bf16 -- forward (+ bias, + fused residual), data
    gradient (flipped taps), weight + bias gradient (positions as the k axis, transposing LDS reads) against the fp32 MFMA kernels (split6 = 0) and a float64
    reference: within 2e-6 of the fp32 kernels, at most 2x their error against float64; exact boxes (64^3), ragged boxes in every direction, a volume that is
    one box deep in places.  The launch counters prove which kernels ran."""
    hip = mrdis.hip
    x = seeded((N, 16, D, H, W), 1); w = seeded((16, 16, 3, 3, 3), 2, 0.1); b = seeded((16,), 3)
    res = seeded((N, 16, D, H, W), 4); dy = seeded((N, 16, D, H, W), 5)
    conv = mrdis.HipConv3d(16, 16, (3, 3, 3), padding=(1, 1, 1)).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w); conv.bias.copy_(b)
    out = {}
    for s6 in (0, 1):
        with hip.option('split6', s6):
            hip.launch_counts(reset=True)
            xg = cl3(x).requires_grad_(True)
            y = conv(xg, residual=cl3(res))
            y.backward(cl3(dy))
            c = hip.launch_counts()
            assert (c['split6_c3d'] == 2) == (s6 == 1) and (c['split6_w3d'] == 1) == (s6 == 1), (s6, c)      # forward + data gradient, weight gradient
            out[s6] = (y.detach().cpu().double(), xg.grad.detach().cpu().double(), conv.weight.grad.detach().cpu().double(), conv.bias.grad.detach().cpu().double())
            conv.zero_grad()
    y64 = F.conv3d(x.double(), w.double(), b.double(), padding=1) + res.double()
    g64 = torch.nn.grad.conv3d_input(x.shape, w.double(), dy.double(), padding=1)
    w64 = torch.nn.grad.conv3d_weight(x.double(), w.shape, dy.double(), padding=1)
    b64 = dy.double().sum((0, 2, 3, 4))
    for k, (name, ref) in enumerate((('fwd', y64), ('dgrad', g64), ('wgrad', w64), ('bgrad', b64))):
        scale = float(ref.abs().max())
        e0, e1 = float((out[0][k] - ref).abs().max()) / scale, float((out[1][k] - ref).abs().max()) / scale
        d01 = float((out[0][k] - out[1][k]).abs().max()) / scale
        # (the weight gradient sums ~1e5 .. 1e6 products per element: both kernels sit at their fp32 summation error there, compared against float64)
        assert d01 <= (2e-6 if k < 2 else 2e-5), (name, 'six-product vs fp32 kernel', d01)
        assert e1 <= max(2.0 * e0, 5e-7), (name, 'vs float64', e1, e0)


@pytest.mark.parametrize('N,Ci,Co,D,H,W', [(1, 32, 32, 64, 64, 64), (1, 32, 16, 64, 64, 64), (3, 16, 48, 40, 52, 36)])
def test_conv3d_weight_gradient_six_product_channel_slices(mrdis, N, Ci, Co, D, H, W):
    """wgrad3d16_s6_kernel on layers wider than 16 channels: one workgroup column per (16-channel slice of x, 16-cout slice of dy) pair (block2 32 -> 32 at 64^3,
    vconv1 32 -> 16 of the VAE branch: model.py:1861-1864, 1969-1984), slabs summed by the fp32 kernel's ordered reduction: weight + bias gradient against the
    fp32 kernel (split6 = 0) and float64."""
    hip = mrdis.hip
    x = seeded((N, Ci, D, H, W), 1); dy = seeded((N, Co, D, H, W), 5)
    xd, dyd = cl3(x), cl3(dy)
    out = {}
    for s6 in (0, 1):
        with hip.option('split6', s6):
            hip.launch_counts(reset=True)
            dw, db = hip.conv3d_bwd_weight(xd, dyd, 3, 1, 1, True)
            assert (hip.launch_counts()['split6_w3d'] == 1) == (s6 == 1)
            out[s6] = (dw.detach().cpu().double(), db.detach().cpu().double())
    w64 = torch.nn.grad.conv3d_weight(x.double(), (Co, Ci, 3, 3, 3), dy.double(), padding=1).permute(2, 3, 4, 1, 0).reshape(27, Ci, Co)      # [tap][ci][co]
    b64 = dy.double().sum((0, 2, 3, 4))
    for k, (name, ref) in enumerate((('wgrad', w64), ('bgrad', b64))):
        scale = float(ref.abs().max())
        e0, e1 = float((out[0][k] - ref).abs().max()) / scale, float((out[1][k] - ref).abs().max()) / scale
        assert float((out[0][k] - out[1][k]).abs().max()) / scale <= 2e-5, name
        assert e1 <= max(2.0 * e0, 5e-7), (name, e1, e0)


def test_basic_block_residual_gradient_inside_the_groupnorm_backward(mrdis):
    """BasicBlock (model.py:1856-1876): x feeds gn1 and the residual addition.  With MRDIS_GN_TAP (default) both gradients of x are summed inside the GroupNorm
    backward pass (mrdis_groupnorm_relu_bwd_add) instead of by autograd's extra add: every gradient must be BIT-IDENTICAL to the two-step form."""
    m3 = mrdis.model3d
    torch.manual_seed(3)
    blk = mrdis.BasicBlock(16, 16).to(DEV)
    x = cl3(seeded((2, 16, 12, 20, 24), 1)); dy = cl3(seeded((2, 16, 12, 20, 24), 2))
    res = {}
    for tap in (False, True):
        m3._GN_TAP = tap
        try:
            xg = x.clone().requires_grad_(True)
            pre = xg * 1.0                                   # (a non-leaf input, as inside the network)
            y = blk(pre)
            y.backward(dy)
            res[tap] = (y.detach().clone(), xg.grad.clone(), [p.grad.clone() for p in blk.parameters()])
            blk.zero_grad()
        finally:
            m3._GN_TAP = True
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
    for a, b in zip(res[False][2], res[True][2]):
        assert torch.equal(a, b)
