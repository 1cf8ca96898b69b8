"""Winograd F(4x4,3x3) (csrc/mrdis_wino4.hip, option wino4): the forward and the data gradient of the wide 3x3 / stride 1 / pad 1 layers
(reference: F.conv2d inside CondConv2d, model.py:2104-2117; SPADE block convolutions :2440-2446, U-Net decoder :2227-2245).
  * the 36-point filter image (mrdis_wino_u_jobs, format 4) against G g G^T in float64;
  * the convolution against torch fp32 and against the direct kernel (wino = 0) on small ragged shapes and on every layer of the
    benchmarked step (B = 32, 256x256 input) that the policy hands to this kernel.
Tolerance: north_star's bar is 1e-3 relative; the round-3 verdict asked for <= 1e-4 of the maximum on every layer at bench scale.  Measured:
3e-6 .. 1.6e-5 (the direct kernel: ~1e-6)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def rnd(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def cl(x):
    return x.to(DEV).contiguous(memory_format=torch.channels_last)


def build_images(hip, wt, wk):
    """(forward, data-gradient) images of a filter pair, in the format the library's policy gives each role (None: no image for that role)"""
    ci, co = wt.shape[1], wt.shape[2]
    jobs, imgs, blocks = [], [], 0
    for src, R, S, flip in ((wt, ci, co, 0), (wk, co, ci, 1)):
        if S <= 32 and hip.wino_u_format(R, S) != 5:         # (<= 32 couts: only the narrow F(4x4) form reads an image)
            imgs.append(None); continue
        img = torch.full((hip.wino_u_image_floats(R, S),), float('nan'), device=DEV)
        j = hip.WinoUJob(); j.w, j.img, j.R, j.S, j.flip, j.spadeC = src.data_ptr(), img.data_ptr(), R, S, flip, 0
        j.block0, j.nblk = blocks, hip.wino_u_job_blocks(R, S); blocks += j.nblk
        jobs.append(j); imgs.append(img)
    if jobs:
        hip.wino_u_jobs(hip.wino_u_table(jobs, DEV), len(jobs), blocks)
    return imgs


G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)


@pytest.mark.parametrize('R,S,flip', [(64, 64, 0), (72, 100, 1), (128, 256, 0), (16, 72, 0)])
def test_f44_filter_image(mrdis, R, S, flip):
    """format 4: [cout tile][chunk of 4][18 point pairs][4 channels kq][128 slots], slot of (cout m, point parity) = (2 m + parity + 32 kq) & 127,
    zero where the chunk / the tile runs past R / S (csrc/mrdis_wino4.h)."""
    hip = mrdis.hip
    hip.set_option('wino4', 2)                                      # the format for every filter the kernel can take (R = 16 included)
    assert hip.wino_u_format(R, S) == 4
    w = rnd((9, R, S), 5).to(DEV)
    tiles, nch = (S + 63) // 64, (R + 3) // 4
    n4 = tiles * nch * 18 * 4 * 128
    n2 = tiles * ((R + 7) // 8) * 8 * 64 * 16            # the 16-point image of the same filter follows (fallback for calls the F(4x4) kernel declines)
    assert hip.wino_u_image_floats(R, S) == n4 + n2
    img = torch.full((n4 + n2,), float('nan'), device=DEV)
    j = hip.WinoUJob(); j.w, j.img, j.R, j.S, j.flip, j.spadeC, j.block0, j.nblk = w.data_ptr(), img.data_ptr(), R, S, flip, 0, 0, hip.wino_u_job_blocks(R, S)
    hip.wino_u_jobs(hip.wino_u_table([j], DEV), 1, j.nblk)
    g = w.double().cpu().reshape(3, 3, R, S)
    if flip:
        g = g.flip(0, 1)
    U = torch.einsum('ai,ijrs,bj->abrs', G4, g, G4).reshape(36, R, S)
    Up = torch.zeros(36, nch * 4, tiles * 64, dtype=torch.float64); Up[:, :R, :S] = U
    Up = Up.reshape(18, 2, nch, 4, tiles, 64)                        # [pp][parity][chunk][kq][tile][m]
    want = torch.zeros(tiles, nch, 18, 4, 128, dtype=torch.float64)
    m = torch.arange(64)
    for kq in range(4):
        for par in range(2):
            slot = (2 * m + par + 32 * kq) & 127
            want[:, :, :, kq, slot] = Up[:, par, :, kq].permute(2, 1, 0, 3)        # [tile][chunk][pp][m]
    got = img[:n4].cpu().double().reshape(want.shape)
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
    hip.set_option('wino4', 0)                                      # the tail IS the format-2 image (tests/test_gpu_ops.py::test_winograd_filter_image)
    assert hip.wino_u_format(R, S) == 2 and hip.wino_u_image_floats(R, S) == n2
    img2 = torch.full((n2,), float('nan'), device=DEV)
    j.img, j.nblk, j.fmt = img2.data_ptr(), hip.wino_u_job_blocks(R, S), 2     # (a job keeps the format it was first tabled with unless told otherwise)
    hip.wino_u_jobs(hip.wino_u_table([j], DEV), 1, j.nblk)
    assert torch.equal(img[n4:], img2)


@pytest.mark.parametrize('B,ci,co,H,W', [(2, 64, 64, 20, 37), (3, 72, 100, 50, 70), (1, 64, 128, 64, 64), (2, 16, 72, 33, 31), (1, 128, 256, 16, 32)])
def test_f44_conv_small_shapes(mrdis, B, ci, co, H, W):
    """forward (bias + LeakyReLU) and data gradient vs torch fp32 and vs the direct kernel; ragged edges, couts / channels that do not fill a tile."""
    hip = mrdis.hip
    hip.set_option('wino', 2); hip.set_option('wino4', 2)
    x = rnd((B, ci, H, W), 1); w = rnd((co, ci, 3, 3), 2, 0.05); b = rnd((co,), 3, 0.1); dy = rnd((B, co, H, W), 4)
    wt = w.permute(2, 3, 1, 0).reshape(9, ci, co).contiguous().to(DEV)
    wk = wt.permute(0, 2, 1).contiguous()
    im_f, im_b = build_images(hip, wt, wk)
    assert hip.wino_u_format(ci, co) == 4                            # (the data gradient takes whichever format its own (R, S) = (co, ci) gets)
    y = hip.conv2d_fwd(cl(x), wt, b.to(DEV), 3, 3, 1, 1, lrelu=True, w_wino=im_f)
    g = hip.conv2d_bwd_data(cl(dy), wk, (H, W), 3, 3, 1, 1, w_wino=im_b)
    want_y = F.leaky_relu(F.conv2d(x, w, b, 1, 1), 0.2)
    want_g = F.conv_transpose2d(dy, w, None, 1, 1)
    assert float((y.cpu() - want_y).abs().max()) <= 5e-5 * float(want_y.abs().max())
    assert float((g.cpu() - want_g).abs().max()) <= 5e-5 * float(want_g.abs().max())
    hip.set_option('wino', 0)
    yd = hip.conv2d_fwd(cl(x), wt, b.to(DEV), 3, 3, 1, 1, lrelu=True)
    assert not torch.equal(yd, y), 'the F(4x4) kernel did not run'
    assert float((y - yd).abs().max()) <= 5e-5 * float(yd.abs().max())


# every 3x3 s1 layer of the benchmarked step (B = 32, 256x256; 8 calls of the anatomy decoder run at B = 32 too) whose forward and / or data
# gradient the default policy gives to the F(4x4) kernel: (name, Ci, Co, map)
BENCH_LAYERS = [('sp5.gamma|beta (two-step path)', 64, 128, 128), ('sp4.gamma|beta', 128, 256, 64), ('sp3.gamma|beta', 128, 256, 32), ('sp4.out', 128, 64, 64),
                ('ana.up_2', 256, 64, 64), ('ana.up_3', 512, 128, 32), ('sp5.gamma|beta at B = 8', 64, 128, 128)]


@pytest.mark.parametrize('name,ci,co,hw', BENCH_LAYERS)
def test_f44_layers_at_bench_scale(mrdis, name, ci, co, hw):
    """default policy (wino4 = 1) on the full B = 32 tensors against the direct kernels: <= 1e-4 of the maximum (the round-3 verdict's bar; measured
    <= 1.6e-5), and the kernel really is the F(4x4) one wherever format and grid say so (bit-different from the F(2x2) result)."""
    hip = mrdis.hip
    B = 32
    x = cl(rnd((B, ci, hw, hw), 1)); dy = cl(rnd((B, co, hw, hw), 2))
    wt = (rnd((9, ci, co), 3, 0.05)).to(DEV); wk = wt.permute(0, 2, 1).contiguous(); b = rnd((co,), 4, 0.1).to(DEV)
    hip.set_option('wino4', 1)
    fmt = (hip.wino_u_format(ci, co), hip.wino_u_format(co, ci))
    im_f, im_b = build_images(hip, wt, wk)
    y = hip.conv2d_fwd(x, wt, b, 3, 3, 1, 1, w_wino=im_f); g = hip.conv2d_bwd_data(dy, wk, (hw, hw), 3, 3, 1, 1, w_wino=im_b)
    hip.set_option('wino4', 0)
    im2_f, im2_b = build_images(hip, wt, wk)
    y2 = hip.conv2d_fwd(x, wt, b, 3, 3, 1, 1, w_wino=im2_f); g2 = hip.conv2d_bwd_data(dy, wk, (hw, hw), 3, 3, 1, 1, w_wino=im2_b)
    hip.set_option('wino', 0)
    yd = hip.conv2d_fwd(x, wt, b, 3, 3, 1, 1); gd = hip.conv2d_bwd_data(dy, wk, (hw, hw), 3, 3, 1, 1)
    ey, eg = float((y - yd).abs().max() / yd.abs().max()), float((g - gd).abs().max() / gd.abs().max())
    assert ey <= 1e-4 and eg <= 1e-4, (name, ey, eg)
    assert 4 in fmt, (name, fmt)
    ran_f = not torch.equal(y, y2); ran_b = not torch.equal(g, g2)
    assert ran_f or ran_b, f'{name}: neither direction ran on the F(4x4) kernel'
    if fmt[0] != 4:
        assert not ran_f
    if fmt[1] != 4:
        assert not ran_b


@pytest.mark.parametrize('case', [(3, 64, 64, 50, 72), (5, 128, 128, 33, 47), (2, 16, 40, 64, 80), (32, 128, 128, 64, 64), (8, 64, 64, 128, 128)], ids=str)
def test_f44_spade_epilogue(mrdis, case):
    """mrdis_conv2d_fwd_spade on the F(4x4) kernel (fused gamma | beta filter image in the SPADE cout order; gamma and beta of a channel meet in one
    lane through v_permlane32_swap; epilogue = InstanceNorm modulation, model.py:2440-2446) against the two-step form on the direct kernel and
    against torch: mix and gamma, ragged blocks, a channel count that is not a multiple of 32, and the two gamma | beta layers of the benchmarked
    step that the policy gives to this kernel (sp4 at B = 32, sp5 at B = 8)."""
    N, Ci, C, H, W = case
    hip = mrdis.hip
    x = cl(rnd((N, Ci, H, W), 1)); z = cl(rnd((N, C, H, W), 2))
    w = rnd((2 * C, Ci, 3, 3), 3, 0.1); b = rnd((2 * C,), 4, 0.1).to(DEV)
    wt = w.permute(2, 3, 1, 0).reshape(9, Ci, 2 * C).contiguous().to(DEV)
    hip.set_option('wino', 0)
    gb = hip.conv2d_fwd(x, wt, b, 3, 3, 1, 1)
    mix_ref, mean_ref, rstd_ref = hip.instnorm_spade_fwd(z, gb[:, :C], gb[:, C:], 1e-5)
    hip.set_option('wino', 2); hip.set_option('wino4', 2)
    assert hip.wino_u_format(Ci, 2 * C, C) == 4
    img = torch.full((hip.wino_u_image_floats(Ci, 2 * C, C),), float('nan'), device=DEV)
    j = hip.WinoUJob(); j.w, j.img, j.R, j.S, j.flip, j.spadeC, j.block0, j.nblk = wt.data_ptr(), img.data_ptr(), Ci, 2 * C, 0, C, 0, hip.wino_u_job_blocks(Ci, 2 * C, C)
    hip.wino_u_jobs(hip.wino_u_table([j], DEV), 1, j.nblk)
    assert torch.isfinite(img).all()
    res = hip.gb_spade_fwd(x, wt, b, z, 1e-5, w_wino=img)
    assert res is not None
    mix, gamma, mean, rstd = res
    hip.set_option('wino4', 0)
    res2 = hip.gb_spade_fwd(x, wt, b, z, 1e-5)
    assert res2 is None or not torch.equal(res2[0], mix), 'the F(4x4) SPADE kernel did not run'      # (None: a channel count the F(2x2) form declines)
    tol = 5e-5
    assert float((gamma - gb[:, :C]).abs().max()) <= tol * float(gb.abs().max())
    assert float((mix - mix_ref).abs().max()) <= tol * float(mix_ref.abs().max())
    assert torch.equal(mean, mean_ref) and torch.equal(rstd, rstd_ref)
    if N * H * W <= 200000:
        gr = F.conv2d(x.cpu(), w, b.cpu(), 1, 1)
        ref = F.instance_norm(z.cpu(), eps=1e-5) * (1 + gr[:, :C]) + gr[:, C:]
        assert float((mix.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize('B,ci,co,H,W', [(2, 64, 32, 40, 37), (3, 72, 20, 50, 70), (1, 16, 8, 64, 64), (2, 128, 32, 33, 65)])
def test_f44_narrow_form_small_shapes(mrdis, B, ci, co, H, W):
    """wino4n_kernel (<= 32 couts: 8 x 8 tiles x 32 couts per workgroup, image format 5): forward (bias + LeakyReLU) vs torch and the direct kernel;
    ragged 32 x 32 blocks, couts that do not fill the tile."""
    hip = mrdis.hip
    hip.set_option('wino', 2); hip.set_option('wino4', 2)
    assert hip.wino_u_format(ci, co) == 5
    x = rnd((B, ci, H, W), 1); w = rnd((co, ci, 3, 3), 2, 0.05); b = rnd((co,), 3, 0.1)
    wt = w.permute(2, 3, 1, 0).reshape(9, ci, co).contiguous().to(DEV)
    wk = wt.permute(0, 2, 1).contiguous()
    im_f, _ = build_images(hip, wt, wk)
    assert im_f is not None and torch.isfinite(im_f).all()
    y = hip.conv2d_fwd(cl(x), wt, b.to(DEV), 3, 3, 1, 1, lrelu=True, w_wino=im_f)
    want = F.leaky_relu(F.conv2d(x, w, b, 1, 1), 0.2)
    assert float((y.cpu() - want).abs().max()) <= 5e-5 * float(want.abs().max())
    y2 = hip.conv2d_fwd(cl(x), wt, b.to(DEV), 3, 3, 1, 1, lrelu=True)        # no image: the F(2x2) kernel for 32 couts
    assert not torch.equal(y, y2), 'the narrow F(4x4) kernel did not run'


@pytest.mark.parametrize('mode', [2, 3])
@pytest.mark.parametrize('B,ci,co,H,W', [(2, 64, 32, 40, 37), (3, 72, 20, 50, 70), (1, 16, 8, 64, 64), (2, 128, 32, 33, 65), (1, 24, 32, 17, 96)])
def test_f44_register_fed_form_small_shapes(mrdis, B, ci, co, H, W, mode):
    """wino4r_kernel (mrdis_wino4r.hip: every wave transforms its own 16 tiles x 4 channels and feeds the MFMAs from registers), option wino4r = 2: 64-tile
    workgroups, 3: channel-split wave pairs that add their partial outputs through LDS; ragged blocks, couts that do not fill the tile, odd stage counts."""
    hip = mrdis.hip
    hip.set_option('wino', 2); hip.set_option('wino4', 2)
    assert hip.wino_u_format(ci, co) == 5
    x = rnd((B, ci, H, W), 1); w = rnd((co, ci, 3, 3), 2, 0.05); b = rnd((co,), 3, 0.1)
    wt = w.permute(2, 3, 1, 0).reshape(9, ci, co).contiguous().to(DEV)
    im_f, _ = build_images(hip, wt, wt.permute(0, 2, 1).contiguous())
    hip.set_option('wino4r', 0)
    y0 = hip.conv2d_fwd(cl(x), wt, b.to(DEV), 3, 3, 1, 1, lrelu=True, w_wino=im_f)           # the shared-transform form
    hip.set_option('wino4r', mode)
    y = hip.conv2d_fwd(cl(x), wt, b.to(DEV), 3, 3, 1, 1, lrelu=True, w_wino=im_f)
    hip.set_option('wino4r', 1)
    want = F.leaky_relu(F.conv2d(x, w, b, 1, 1), 0.2)
    assert float((y.cpu() - want).abs().max()) <= 5e-5 * float(want.abs().max())
    if mode == 2:
        assert torch.equal(y, y0)            # the same arithmetic in the same order as the shared-transform form: the same bits
    else:
        assert not torch.equal(y, y0), 'the channel-split form did not run'


def test_f44_register_fed_form_at_bench_scale(mrdis):
    """the data gradient of the full-resolution gamma | beta convolution (64 -> 32 at 256x256, B = 32: a 537 MB input): the default policy gives it to the
    64-tile form of wino4r_kernel; against the direct kernel: <= 1e-4 of the maximum."""
    hip = mrdis.hip
    R, S, hw = 64, 32, 256
    x = cl(rnd((32, R, hw, hw), 1))
    wt = (rnd((9, R, S), 3, 0.05)).to(DEV)
    hip.set_option('wino4', 1); hip.set_option('wino4r', 1)
    assert hip.wino_u_format(R, S) == 5
    im_f, _ = build_images(hip, wt, wt.permute(0, 2, 1).contiguous())
    y = hip.conv2d_fwd(x, wt, None, 3, 3, 1, 1, w_wino=im_f)
    hip.set_option('wino4r', 0)
    y2 = hip.conv2d_fwd(x, wt, None, 3, 3, 1, 1, w_wino=im_f)     # (declined by the shared-transform form at this size: the F(2x2) kernel)
    hip.set_option('wino4r', 1); hip.set_option('wino', 0)
    yd = hip.conv2d_fwd(x, wt, None, 3, 3, 1, 1)
    assert not torch.equal(y, y2), 'the register-fed F(4x4) kernel did not run'
    e = float((y - yd).abs().max() / yd.abs().max())
    assert e <= 1e-4, e


@pytest.mark.parametrize('name,R,S,hw', [('sp5.out forward (register-fed form)', 64, 32, 128), ('ana.up_1 forward', 128, 32, 128)])
def test_f44_narrow_layers_at_bench_scale(mrdis, name, R, S, hw):
    """the 32-cout layers of the benchmarked step (B = 32) that the default policy gives to the two 32-cout forms (the 64 -> 32 data gradient at 256x256:
    test_f44_register_fed_form_at_bench_scale) against the direct kernel: <= 1e-4 of the maximum."""
    hip = mrdis.hip
    x = cl(rnd((32, R, hw, hw), 1))
    wt = (rnd((9, R, S), 3, 0.05)).to(DEV); b = rnd((S,), 4, 0.1).to(DEV)
    hip.set_option('wino4', 1)
    assert hip.wino_u_format(R, S) == 5
    im_f, _ = build_images(hip, wt, wt.permute(0, 2, 1).contiguous())
    y = hip.conv2d_fwd(x, wt, b, 3, 3, 1, 1, w_wino=im_f)
    y2 = hip.conv2d_fwd(x, wt, b, 3, 3, 1, 1)
    hip.set_option('wino', 0)
    yd = hip.conv2d_fwd(x, wt, b, 3, 3, 1, 1)
    assert not torch.equal(y, y2), f'{name}: the narrow F(4x4) kernel did not run'
    e = float((y - yd).abs().max() / yd.abs().max())
    assert e <= 1e-4, (name, e)
