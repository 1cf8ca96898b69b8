"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement, in plain fp32 torch ops, of the reference's multi-modal MR
disentanglement model (ouyangjiahong/representation-disentanglement,
`src/model.py`).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this file.

The arithmetic of the reference lives in a third-party dependency that is not
under /root/reference: PyTorch (un-pinned upstream; 2.10.0+rocm7.0 in this
image).  This restatement therefore calls the same published torch
primitives (F.conv2d, F.batch_norm, F.instance_norm, F.interpolate,
F.softmax ...) in the order the reference's call sites do.

Parity status: PINNED.  `oracle/gen_golden.py` imports the real reference in
the build container and writes `tests/golden/*.npz|json`;
`tests/test_oracle_golden.py` checks this file against those vectors.

Construction order (and therefore the torch RNG stream consumed by parameter
initialisation) follows the reference exactly, so `torch.manual_seed(s)`
followed by `RefMultimodalModel(...)` reproduces the reference's initial
weights bit for bit, and `state_dict()` has the reference's key names/shapes
(SURVEY.md Appendix D).

Each definition cites the reference lines it follows (paths relative to
/root/reference/src).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# operator seam: model.py:2065-2120
# --------------------------------------------------------------------------
class _Routing(nn.Module):
    """model.py:2065-2073 -- sigmoid(Linear(emb -> E)(type))."""

    def __init__(self, emb, experts):
        super().__init__()
        self.fc = nn.Linear(emb, experts)

    def forward(self, t):
        return torch.sigmoid(self.fc(t))


class RefCondConv2d(nn.Module):
    """model.py:2075-2117.  Expert-mixed conv; per-sample batch-1 conv loop.

    Parameters: weight (E,Co,Ci,kh,kw), bias (Co) [zeros], _routing_fn.fc.*.
    RNG: the reference first runs `_ConvNd.__init__` (kaiming-uniform weight +
    uniform bias draws), then builds the routing Linear, then xavier-normal on
    the 5-D weight (model.py:2083-2097); the throw-away nn.Conv2d below burns
    the same draws.
    """

    def __init__(self, cin, cout, k, stride=1, padding=0, embeddings=1, experts=3):
        super().__init__()
        k = (k, k) if isinstance(k, int) else tuple(k)
        self.stride, self.padding, self.k = stride, padding, k
        burn = nn.Conv2d(cin, cout, k, stride, padding)      # _ConvNd.reset_parameters draws
        self._routing_fn = _Routing(embeddings, experts)
        self.weight = nn.Parameter(torch.empty(experts, cout, cin, *k))
        self.bias = nn.Parameter(torch.empty(cout))
        del burn
        nn.init.xavier_normal_(self.weight)                   # model.py:2096
        nn.init.constant_(self.bias, 0)                       # model.py:2097

    def forward(self, x, t):
        r = self._routing_fn(t)                                              # (B,E)  :2111
        kern = torch.sum(r[:, :, None, None, None, None] * self.weight, 1)   # :2113
        outs = [F.conv2d(x[i:i + 1], kern[i], self.bias, self.stride, self.padding)
                for i in range(x.shape[0])]                                  # :2114-2116
        return torch.cat(outs, 0)


# --------------------------------------------------------------------------
# anatomy U-Net: model.py:2122-2195, 2218-2245, 2271-2296
# --------------------------------------------------------------------------
class RefConvBNAct(nn.Module):
    """model.py:2122-2153.  conv4x4 s2 -> BN(train) -> *identity*.

    The if/if/if-else chain (:2134-2141) leaves `act` an empty Sequential for
    'lrelu' and 'relu' (SURVEY 0-4); reproduced by having no activation.
    """

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = RefCondConv2d(cin, cout, 4, 2, 1)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x, t):
        return self.bn(self.conv(x, t))


class RefUpConvBNCat(nn.Module):
    """model.py:2155-2195.  identity -> bilinear x2 (align_corners=True) ->
    conv3x3 -> [BN -> cat(skip)] unless is_last.  `bn` exists even when
    is_last (:2179) and then never receives a gradient."""

    def __init__(self, cin, cout, is_last=False):
        super().__init__()
        self.is_last = is_last
        self.conv = RefCondConv2d(cin, cout, 3, 1, 1)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, skip, x, t):
        x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)   # :2175
        x = self.conv(x, t)
        if self.is_last:
            return x
        return torch.cat([skip, self.bn(x)], 1)                                      # :2191-2192


class RefAnatomyEnc(nn.Module):
    """model.py:2218-2245."""

    def __init__(self, cin=7, c=32):
        super().__init__()
        self.down_1 = RefCondConv2d(cin, c, 4, 2, 1)
        self.down_2 = RefConvBNAct(c, 2 * c)
        self.down_3 = RefConvBNAct(2 * c, 4 * c)
        self.down_4 = RefConvBNAct(4 * c, 8 * c)
        self.down_5 = RefConvBNAct(8 * c, 8 * c)

    def forward(self, x, t):
        d1 = F.leaky_relu(self.down_1(x, t), 0.2)           # act_1, :2227/:2240
        d2 = self.down_2(d1, t)
        d3 = self.down_3(d2, t)
        d4 = self.down_4(d3, t)
        d5 = self.down_5(d4, t)
        return [d1, d2, d3, d4, d5]


class RefAnatomyDec(nn.Module):
    """model.py:2271-2296."""

    def __init__(self, c=32, cout=4):
        super().__init__()
        self.up_4 = RefUpConvBNCat(8 * c, 8 * c)
        self.up_3 = RefUpConvBNCat(16 * c, 4 * c)
        self.up_2 = RefUpConvBNCat(8 * c, 2 * c)
        self.up_1 = RefUpConvBNCat(4 * c, c)
        self.output = RefUpConvBNCat(2 * c, cout, is_last=True)

    def forward(self, d, t):
        u4 = self.up_4(d[3], d[4], t)
        u3 = self.up_3(d[2], u4, t)
        u2 = self.up_2(d[1], u3, t)
        u1 = self.up_1(d[0], u2, t)
        return self.output(None, u1, t)


# --------------------------------------------------------------------------
# modality encoder: model.py:2332-2400
# --------------------------------------------------------------------------
class RefModalityEnc(nn.Module):
    """model.py:2332-2400.  `convs` (:2346-2357) is dead code that still owns
    parameters / RNG draws / state_dict keys.  The reference hard-codes the
    5*6 feature grid (:2360, :2396); `feat_hw` generalises it to
    (H/32)*(W/32) -- identical at 160x192."""

    def __init__(self, cin=7, c=16, z=16, feat_hw=30):
        super().__init__()
        chans = [cin, c, 2 * c, 4 * c, 8 * c, 8 * c]
        for i in range(5):
            setattr(self, f'conv{i + 1}', RefCondConv2d(chans[i], chans[i + 1], 3, 2, 1))
        dead = []
        for i in range(5):
            dead += [nn.Conv2d(chans[i], chans[i + 1], 3, 2, padding=1), nn.LeakyReLU(0.2)]
        self.convs = nn.Sequential(*dead)
        self.fcs = nn.Sequential(nn.Linear(feat_hw * 8 * c, 2 * z), nn.LeakyReLU(0.2))
        self.mean = nn.Linear(2 * c, z)
        self.log_var = nn.Linear(2 * c, z)

    def forward(self, x, t):
        for i in range(5):
            x = F.leaky_relu(getattr(self, f'conv{i + 1}')(x, t), 0.2)     # :2374-2383
        x = self.fcs(x.reshape(x.shape[0], -1))                              # :2396-2397
        return self.mean(x), self.log_var(x)


# --------------------------------------------------------------------------
# SPADE decoder: model.py:2424-2454, 2540-2632
# --------------------------------------------------------------------------
class RefSPADEBlock(nn.Module):
    """model.py:2424-2454.  IN(z)*(1+gamma(s))+beta(s) -> conv3x3; `s` is
    bilinearly resized (align_corners=False) to this block's grid."""

    def __init__(self, size, cin, cout, s_ch):
        super().__init__()
        self.size = tuple(size)
        self.si_layers = RefCondConv2d(s_ch, cin, 3, 1, 1)
        self.gamma = RefCondConv2d(cin, cin, 3, 1, 1)
        self.beta = RefCondConv2d(cin, cin, 3, 1, 1)
        self.out = RefCondConv2d(cin, cout, 3, 1, 1)

    def forward(self, s, z, t):
        zn = F.instance_norm(z, eps=1e-5)                                     # :2431/:2440
        s = F.interpolate(s, size=self.size, mode='bilinear', align_corners=False)
        so = self.si_layers(s, t)
        mix = zn * (1 + self.gamma(so, t)) + self.beta(so, t)                 # :2446
        return self.out(mix, t)


def _up2(x):
    """nn.Upsample(scale_factor=(2,2), mode='bilinear'), model.py:2551."""
    return F.interpolate(x, scale_factor=(2, 2), mode='bilinear', align_corners=False)


class RefSPADEShared(nn.Module):
    """model.py:2540-2582 (zi_scaler + sp1..sp3, three x2 upsamples)."""

    def __init__(self, size, z=16, zc=128, s_ch=4):
        super().__init__()
        H, W = size
        self.zc, self.grid = zc, (H // 32, W // 32)
        self.zi_scaler = nn.Linear(z, H * W * zc // 1024)
        self.sp1 = RefSPADEBlock((H // 32, W // 32), zc, zc, s_ch)
        self.sp2 = RefSPADEBlock((H // 16, W // 16), zc, zc, s_ch)
        self.sp3 = RefSPADEBlock((H // 8, W // 8), zc, zc, s_ch)

    def forward(self, s, z, t):
        x = self.zi_scaler(z).reshape(-1, self.zc, *self.grid)
        x = self.sp1(s, x, t)
        x = self.sp2(s, _up2(x), t)
        x = self.sp3(s, _up2(x), t)
        return _up2(x)


class RefSPADENotShared(nn.Module):
    """model.py:2584-2632 (sp4..sp6 + 1x1 out conv; out_act identity for
    z-score data, main_missing.py:83-86)."""

    def __init__(self, size, cin=7, zc=128, s_ch=4):
        super().__init__()
        H, W = size
        self.sp4 = RefSPADEBlock((H // 4, W // 4), zc, zc // 2, s_ch)
        self.sp5 = RefSPADEBlock((H // 2, W // 2), zc // 2, zc // 4, s_ch)
        self.sp6 = RefSPADEBlock((H, W), zc // 4, zc // 8, s_ch)
        self.out = RefCondConv2d(zc // 8, cin, 1, 1)

    def forward(self, s, x, t):
        x = self.sp4(s, x, t)
        x = self.sp5(s, _up2(x), t)
        x = self.sp6(s, _up2(x), t)
        return self.out(x, t)


# --------------------------------------------------------------------------
# discriminator: model.py:2769-2800
# --------------------------------------------------------------------------
class RefDiscriminator(nn.Module):
    def __init__(self, cin=4, c=16, input_shape=(160, 192), is_patch_gan=False):
        super().__init__()
        L = [nn.Conv2d(cin, c, 4, 2, padding=1), nn.LeakyReLU(0.2)]
        for a, b in ((c, 2 * c), (2 * c, 4 * c), (4 * c, 8 * c), (8 * c, 4 * c)):
            L += [nn.Conv2d(a, b, 4, 2, padding=1), nn.BatchNorm2d(b), nn.LeakyReLU(0.2)]
        self.discrim = nn.Sequential(*L)
        if is_patch_gan:
            self.fc = nn.Conv2d(4 * c, 1, 3, 1, padding=1)
        else:
            self.fc = nn.Sequential(
                nn.Flatten(),
                nn.Linear(int(input_shape[0] * input_shape[1] * 4 * c / (32 * 32)), c * 16),
                nn.LeakyReLU(0.2),
                nn.Linear(c * 16, 1))

    def forward(self, x):
        return self.fc(self.discrim(x))


# --------------------------------------------------------------------------
# orchestration + losses: model.py:2916-2970, 3086-3224, 3260-3587
# --------------------------------------------------------------------------
# ----------------------------------------------------------------------------- output decoder 'U+SA' (model.py:117-168, 341-390, 1303-1327)
class RefLegacyConvBNAct(nn.Module):
    """Conv_BN_Act (model.py:117-139).  QUIRK: the if/if/if-else chain leaves `act` = identity unless 'elu'."""

    def __init__(self, cin, cout, activation='lrelu'):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(cin, cout, 4, 2, padding=1), nn.BatchNorm2d(cout))
        self.act = nn.ELU(inplace=True) if activation == 'elu' else nn.Sequential()

    def forward(self, x):
        return self.act(self.conv(x))


class RefLegacyUpConcat(nn.Module):
    """Act_Deconv_BN_Concat (model.py:141-174), same identity-activation quirk; the BatchNorm exists (and draws no
    RNG) even when is_last."""

    def __init__(self, cin, cout, is_last=False):
        super().__init__()
        self.act = nn.Sequential()
        self.is_last = is_last
        self.up = nn.Sequential(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True),
                                nn.Conv2d(cin, cout, 3, 1, padding=1))
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x_down, x_up):
        x_up = self.up(self.act(x_up))
        if self.is_last:
            return x_up
        return torch.cat([x_down, self.bn(x_up)], 1)


class RefSpatialAttention(nn.Module):
    """SpatialAttentionLayer (model.py:1303-1327)."""

    def __init__(self, cin, cgate, cinter):
        super().__init__()
        self.W_x = nn.Conv2d(cin, cinter, (2, 2), (2, 2), bias=False)
        self.W_g = nn.Conv2d(cgate, cinter, 1, 1)
        self.W_psi = nn.Conv2d(cinter, 1, 1, 1)
        self.W_out = nn.Sequential(nn.Conv2d(cin, cin, 1, 1), nn.BatchNorm2d(cin))

    def forward(self, x, g):
        x_post = self.W_x(x)
        g_post = F.interpolate(self.W_g(g), size=x_post.shape[2:], mode='bilinear', align_corners=False)   # F.upsample default
        alpha = torch.sigmoid(self.W_psi(F.relu(x_post + g_post)))
        alpha_up = F.interpolate(alpha, size=x.shape[2:], mode='bilinear', align_corners=False)
        return self.W_out(alpha_up * x), alpha_up


class RefOutputDecoderUSA(nn.Module):
    """GANShortGeneratorWithSpatialAttention (model.py:341-390), output_activation 'no' (BraTS / z-score, main_missing.py:80)."""

    def __init__(self, in_num_ch, out_num_ch, c=64):
        super().__init__()
        self.down_1 = nn.Sequential(nn.Conv2d(in_num_ch, c, 4, 2, padding=1), nn.LeakyReLU(0.2, inplace=True))
        self.down_2 = RefLegacyConvBNAct(c, 2 * c)
        self.down_3 = RefLegacyConvBNAct(2 * c, 4 * c)
        self.down_4 = RefLegacyConvBNAct(4 * c, 8 * c)
        self.down_5 = RefLegacyConvBNAct(8 * c, 8 * c, activation='no')
        self.att_4 = RefSpatialAttention(8 * c, 8 * c, 8 * c)
        self.up_4 = RefLegacyUpConcat(8 * c, 8 * c)
        self.att_3 = RefSpatialAttention(4 * c, 16 * c, 4 * c)
        self.up_3 = RefLegacyUpConcat(16 * c, 4 * c)
        self.att_2 = RefSpatialAttention(2 * c, 8 * c, 2 * c)
        self.up_2 = RefLegacyUpConcat(8 * c, 2 * c)
        self.att_1 = RefSpatialAttention(c, 4 * c, c)
        self.up_1 = RefLegacyUpConcat(4 * c, c)
        self.output = RefLegacyUpConcat(2 * c, out_num_ch, is_last=True)

    def forward(self, x):
        d1 = self.down_1(x); d2 = self.down_2(d1); d3 = self.down_3(d2); d4 = self.down_4(d3); d5 = self.down_5(d4)
        c4, _ = self.att_4(d4, d5); u4 = self.up_4(c4, d5)
        c3, _ = self.att_3(d3, u4); u3 = self.up_3(c3, u4)
        c2, _ = self.att_2(d2, u3); u2 = self.up_2(c2, u3)
        c1, _ = self.att_1(d1, u2); u1 = self.up_1(c1, u2)
        return self.output(None, u1)


class RefMultimodalModel(nn.Module):
    """Default-config graph (config.yaml: is_cond, shared_ana_enc,
    shared_mod_enc, shared_inp_dec=False, mod_enc_s=False,
    softmax_remove_mask, s_compact 'max', cosine similarities).

    The output decoder (`output_decoder.*`, lambda_recon_y = 0) is not part of
    the hot path and is omitted; it is constructed *after* every hot-path
    module in the reference (model.py:2955-2962) so omitting it does not shift
    their RNG draws.  The discriminator (model.py:2967) is built after it: to
    keep its init identical we accept an `rng_skip` callable that the golden
    generator does not need (goldens with adv copy weights by state_dict).
    """

    def __init__(self, input_size=(160, 192), modality_num=4, in_num_ch=7, s_num_ch=4,
                 z_size=16, is_discrim_s=False, is_patch_gan=False, out_num_ch=0):
        super().__init__()
        H, W = input_size
        self.input_size, self.M = (H, W), modality_num
        self.anatomy_encoder_enc_list = nn.ModuleList([RefAnatomyEnc(in_num_ch, 32)])
        self.anatomy_encoder_dec = RefAnatomyDec(32, s_num_ch)
        self.modality_encoder_list = nn.ModuleList(
            [RefModalityEnc(in_num_ch, 16, z_size, (H // 32) * (W // 32))])
        dec = [RefSPADENotShared((H, W), in_num_ch, 128, s_num_ch) for _ in range(modality_num)]
        dec.append(RefSPADEShared((H, W), z_size, 128, s_num_ch))
        self.input_decoder_list = nn.ModuleList(dec)
        if out_num_ch > 0:                                                   # model.py:2957-2958 ('U+SA', fuse 'mean')
            self.output_decoder = RefOutputDecoderUSA(s_num_ch, out_num_ch)
        if is_discrim_s:
            self.discrim_s = RefDiscriminator(s_num_ch, 16, (H, W), is_patch_gan)

    @staticmethod
    def _type(i, B):
        return (1 + i) * torch.ones(B, 1)                                   # :3138

    # model.py:3135-3157
    def compute_anatomy_encoding(self, x_list, mask_img):
        out = []
        for i in range(self.M):
            t = self._type(i, x_list[0].shape[0])
            feats = self.anatomy_encoder_enc_list[0](x_list[i], t)
            s = self.anatomy_encoder_dec(feats, t)
            cat = torch.cat([100 * mask_img.unsqueeze(1), s], 1)           # :3150
            out.append(F.softmax(cat, dim=1)[:, 1:])                        # :3152-3153
        return out

    # model.py:3159-3185.  eps comes from the CPU generator (:3160).
    def compute_modality_encoding(self, x_list, phase='train'):
        zs, mus, lvs = [], [], []
        for i in range(self.M):
            t = self._type(i, x_list[0].shape[0])
            mu, lv = self.modality_encoder_list[0](x_list[i], t)
            if phase == 'train':
                eps = torch.normal(0, 1, size=(mu.shape[0], mu.shape[1]))
                z = mu + eps * torch.exp(0.5 * lv)
            else:
                z = mu
            zs.append(z); mus.append(mu); lvs.append(lv)
        return zs, mus, lvs

    # model.py:3230-3258.  QUIRK: `si_cat[mask == 1]` flattens (batch, modality) into one axis, so the "fused" map is
    # NOT a per-sample mean over modalities: every present (b, m) anatomy map becomes its own sample (the mean runs over
    # a singleton axis) and the output has sum(mask) rows, b-major.
    def reconstruct_output_si_fused(self, s_list, mask):
        s_cat = torch.stack(s_list, 1)
        sel = s_cat[mask == 1]
        if sel.dim() != s_cat.dim():
            sel = sel.unsqueeze(1)
        return self.output_decoder(torch.mean(sel, 1))

    def reconstruct_output_si(self, s_list):
        B = s_list[0].shape[0]
        return [self.reconstruct_output_si_fused([s_list[i]], torch.ones(B, 1)) for i in range(self.M)]

    def recon_y_list(self, gt, y_list, mask, p=2):                          # :3268-3278
        loss, idx = torch.tensor(0.), 0
        for i in range(len(y_list)):
            if mask[:, i].sum() == 0:
                continue
            idx += 1
            loss = loss + (mask[:, i] * self.recon(gt, y_list[i], p)).sum() / mask[:, i].sum()
        return loss if idx == 0 else loss / idx

    # model.py:3287-3313 (F.softmax without dim on a 4-D tensor = dim 1)
    def segmentation_loss_y(self, gt, y, weight=(1., 5., 5., 5.)):
        loss_seg = F.cross_entropy(y, gt.squeeze(1).long(), weight=torch.tensor(weight))
        y_act = F.softmax(y, dim=1)
        dice = 0
        for i in range(1, 4):
            gt_i = (gt[:, 0] == i).float()
            dice = dice + 1 - 2 * torch.sum(y_act[:, i] * gt_i) / (torch.sum(y_act[:, i] ** 2 + gt_i ** 2) + 1e-6)
        return loss_seg + dice / 3

    def segmentation_loss_y_list(self, gt, y_list, mask):
        loss, idx = torch.tensor(0.), 0
        for i in range(len(y_list)):
            if mask[:, i].sum() == 0:
                continue
            idx += 1
            loss = loss + self.segmentation_loss_y(gt, y_list[i])
        return loss if idx == 0 else loss / idx

    # model.py:3187-3203
    def reconstruct_input_si_zi(self, s_list, z_list):
        out = []
        for i in range(self.M):
            t = self._type(i, s_list[0].shape[0])
            mid = self.input_decoder_list[-1](s_list[i], z_list[i], t)
            out.append(self.input_decoder_list[i](s_list[i], mid, t))
        return out

    # model.py:3205-3224: decoder index i, type j, z_j
    def reconstruct_input_si_zj(self, s_list, z_list):
        out = []
        for i in range(self.M):
            for j in range(self.M):
                if i == j:
                    continue
                t = self._type(j, s_list[0].shape[0])
                mid = self.input_decoder_list[-1](s_list[i], z_list[j], t)
                out.append(self.input_decoder_list[i](s_list[i], mid, t))
        return out

    # ---------------- losses ----------------
    @staticmethod
    def recon(gt, out, p):                                                  # :3260-3266
        dims = list(range(1, gt.dim()))
        d = gt - out
        return d.abs().mean(dims) if p == 1 else d.pow(2).mean(dims)

    def recon_x_list(self, gt_list, x_list, mask, p):                       # :3315-3325
        loss, n = torch.tensor(0.), 0
        for i in range(len(x_list)):
            if mask[:, i].sum() == 0:
                continue
            n += 1
            loss = loss + (mask[:, i] * self.recon(gt_list[i], x_list[i], p)).sum() / mask[:, i].sum()
        return loss if n == 0 else loss / n

    def recon_x_mix_list(self, gt_list, x_list, mask, p):                   # :3327-3341
        loss, idx = torch.tensor(0.), 0
        M = mask.shape[1]
        for i in range(M):
            for j in range(M):
                if i == j:
                    continue
                mm = mask[:, i] * mask[:, j]
                if mm.sum() == 0:
                    continue
                # quirk kept: x_list index only advances on non-empty pairs (:3337-3338)
                loss = loss + (mm * self.recon(gt_list[j], x_list[idx], p)).sum() / mm.sum()
                idx += 1
        return loss if idx == 0 else loss / idx

    def latent_z(self, mu_list, mu_new, mask):                              # :3384-3394
        loss, n = torch.tensor(0.), 0
        for i in range(len(mu_list)):
            if mask[:, i].sum() == 0:
                continue
            n += 1
            loss = loss + (mask[:, i].unsqueeze(1) * (mu_list[i] - mu_new[i]).abs()).sum() / mask[:, i].sum()
        return loss if n == 0 else loss / n

    @staticmethod
    def cosine(x, y):                                                       # :3407-3415
        xn = torch.sqrt((x * x).sum(1) + 1e-8).clamp_min(1e-8)
        yn = torch.sqrt((y * y).sum(1) + 1e-8).clamp_min(1e-8)
        return (x * y).sum(1) / (xn * yn)

    @staticmethod
    def compact_s(x):                                                       # :3448-3451
        return F.max_pool2d(x, kernel_size=(16, 16)).reshape(x.shape[0], -1)

    def sim_s(self, s_list, mask, margin=0.1):                              # :3478-3513
        if len(s_list) == 1:
            return torch.tensor(0.)
        if len(s_list) == 2:
            i, j = 0, 1
        else:
            sel = np.random.choice(len(s_list), 2, replace=False)           # :3485
            i, j = int(sel[0]), int(sel[1])
        si, sj = s_list[i], s_list[j]
        si_perm = torch.cat([si[1:], si[0:1]], 0)
        mperm = torch.cat([mask[1:, i], mask[0:1, i]], 0)
        mm = mask[:, i] * mask[:, j] * mperm
        if mm.sum() > 0:
            a, b, c = self.compact_s(si), self.compact_s(sj), self.compact_s(si_perm)
            sim, sim_mix = self.cosine(a, b), self.cosine(c, a)
            return (mm * torch.clamp_min(margin - sim + sim_mix, 0)).sum() / mm.sum()
        return torch.tensor(0.)   # reference returns python 0 (:3512)

    def sim_z(self, z_list, mask, margin=0.1):                              # :3537-3557
        loss, n = torch.tensor(0.), 0
        if len(z_list) == 1:
            return loss
        for i in range(len(z_list) - 1):
            zi = z_list[i]
            zp = torch.cat([zi[1:], zi[0:1]], 0)
            mperm = torch.cat([mask[1:, i], mask[0:1, i]], 0)
            for j in range(i + 1, len(z_list)):
                mm = mask[:, i] * mask[:, j] * mperm
                if mm.sum() == 0:
                    continue
                n += 1
                c, cm = self.cosine(zi, z_list[j]), self.cosine(zi, zp)
                loss = loss + (mm * torch.clamp_min(margin - cm + c, 0)).sum() / mm.sum()
        return loss if n == 0 else loss / n

    def adversarial(self, s_list, mask):                                    # :3559-3587
        if len(s_list) == 2:
            i, j = 0, 1
        else:
            sel = np.random.choice(len(s_list), 2, replace=False)
            i, j = int(sel[0]), int(sel[1])
        d0 = self.discrim_s(s_list[i]).squeeze(1)
        d1 = self.discrim_s(s_list[j]).squeeze(1)
        bce = lambda d, tgt: F.binary_cross_entropy_with_logits(d, tgt, reduction='none')

        def wmean(m, v):
            # (B,) mask against (B,) dense head, or broadcast like the reference
            return (m * v).sum() / m.sum()
        z = torch.tensor(0.)
        if mask[:, i].sum() == 0:
            dl0, gl0 = z, z
        else:
            dl0 = wmean(mask[:, i], bce(d0, torch.zeros_like(d0)))
            gl0 = wmean(mask[:, i], bce(d0, torch.ones_like(d0)))
        if mask[:, j].sum() == 0:
            dl1, gl1 = z, z
        else:
            dl1 = wmean(mask[:, j], bce(d1, torch.ones_like(d1)))
            gl1 = wmean(mask[:, j], bce(d1, torch.ones_like(d1)))           # (sic) :3580
        return 0.5 * (dl0 + dl1), 0.5 * (gl0 + gl1)


DEFAULT_LAMBDAS = dict(recon_x=1.0, recon_x_mix=2.0, latent_z=0.1, sim_s=10.0, sim_z=2.0,
                       adv_s=0.0, recon_y=0.0)        # config.yaml:27-33, 54-56


def ref_forward_losses(model, inputs, mask, mask_img, lambdas=None, p=1, phase='train', targets=None, dataset_name='BraTS'):
    """main_missing.py:165-251 (phase='train') / :389-505 (phase='test', inside model.eval() + no_grad)
    for the default loss set.  Returns (loss, parts, aux)."""
    lam = dict(DEFAULT_LAMBDAS); lam.update(lambdas or {})
    M = model.M
    c = inputs.shape[1] // M
    x_list = [inputs[:, i * c:(i + 1) * c] for i in range(M)]               # :166-168
    s_list = model.compute_anatomy_encoding(x_list, mask_img)               # :175
    z_list, mu_list, lv_list = model.compute_modality_encoding(x_list, phase)   # :176 / :400
    xf = model.reconstruct_input_si_zi(s_list, z_list)                      # :177
    xmix = model.reconstruct_input_si_zj(s_list, z_list)                    # :178
    parts = {}
    loss = 0
    y_list = None
    if lam['recon_y'] > 0:                                                  # :187-198 (the *_fused variant cannot run for
        y_list = model.reconstruct_output_si(s_list)                        #  M > 1: its output has sum(mask) rows)
        if dataset_name == 'BraTS':
            parts['recon_y'] = model.segmentation_loss_y_list(targets, y_list, mask)
        else:
            parts['recon_y'] = model.recon_y_list(targets, y_list, mask, p)
        loss = loss + lam['recon_y'] * parts['recon_y']
    if lam['recon_x'] > 0:
        parts['recon_x'] = model.recon_x_list(x_list, xf, mask, p)
        loss = loss + lam['recon_x'] * parts['recon_x']
    if lam['recon_x_mix'] > 0:
        parts['recon_x_mix'] = model.recon_x_mix_list(x_list, xmix, mask, p)
        loss = loss + lam['recon_x_mix'] * parts['recon_x_mix']
    if lam['latent_z'] > 0:                                                 # :228-233
        s_new = model.compute_anatomy_encoding(xf, mask_img)
        _, mu_new, _ = model.compute_modality_encoding(xf, phase)
        parts['latent_z'] = model.latent_z(mu_list, mu_new, mask)
        loss = loss + lam['latent_z'] * parts['latent_z']
    if lam['sim_s'] > 0:
        parts['sim_s'] = model.sim_s(s_list, mask)
        loss = loss + lam['sim_s'] * parts['sim_s']
    if lam['sim_z'] > 0:
        parts['sim_z'] = model.sim_z(z_list, mask)
        loss = loss + lam['sim_z'] * parts['sim_z']
    if lam['adv_s'] > 0:
        parts['adv_s_d'], parts['adv_s'] = model.adversarial(s_list, mask)
        loss = loss + lam['adv_s'] * parts['adv_s']
    aux = dict(s_list=s_list, z_list=z_list, mu_list=mu_list, lv_list=lv_list, xf=xf, xmix=xmix, y_list=y_list)
    return loss, parts, aux


def ref_train_step(model, optimizer, inputs, mask, mask_img, lambdas=None, p=1,
                   optimizer_d=None, step_optimizer=True):
    """main_missing.py:165-289: forward, losses, backward, clip 1.0, Adam step."""
    lam = dict(DEFAULT_LAMBDAS); lam.update(lambdas or {})
    loss, parts, aux = ref_forward_losses(model, inputs, mask, mask_img, lam, p)
    adv = lam['adv_s'] > 0
    loss.backward(retain_graph=adv)                                          # :268-271
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)          # :272
    if step_optimizer:
        optimizer.step(); optimizer.zero_grad()                              # :283-284
        if adv:                                                              # :286-289
            optimizer_d.zero_grad()
            parts['adv_s_d'].backward()
            optimizer_d.step()
    return loss.detach(), {k: v.detach() for k, v in parts.items()}, gnorm, aux


def ref_train_iterations(model, optimizer, batches, batch_size, lambdas=None, p=1, on_iter=None):
    """The inner loop of train() (main_missing.py:155-305) over `batches` = [(inputs, mask, mask_img), ...] with the
    reference's accumulation rule: gradients accumulate in p.grad across iterations, clip_grad_norm_ acts on the
    accumulating gradient EVERY iteration (:272), `optimizer.step(); optimizer.zero_grad()` when
    (iter + 1) % (16 // batch_size) == 0 (:282-284).  Returns one record per iteration."""
    accum = 16 // batch_size
    out = []
    for it, (inputs, mask, mask_img) in enumerate(batches):
        stepped = (it + 1) % accum == 0
        loss, parts, gnorm, _ = ref_train_step(model, optimizer, inputs, mask, mask_img, lambdas, p, step_optimizer=stepped)
        out.append(dict(loss=float(loss), parts={k: float(v) for k, v in parts.items()}, grad_norm_before_clip=float(gnorm),
                        stepped=stepped))
        if on_iter is not None:
            on_iter(it, out[-1])
    return out


def perturb_bn_running_stats(model):
    """Deterministic non-trivial BatchNorm running statistics (a freshly built model has mean 0 / var 1,
    which would make inference-mode BN a no-op in the evaluate() fixtures)."""
    with torch.no_grad():
        for name, mod in sorted(model.named_modules()):
            if isinstance(mod, nn.BatchNorm2d):
                k = torch.arange(mod.num_features, dtype=torch.float32)
                mod.running_mean.copy_(0.05 * torch.sin(0.7 * k + len(name)))
                mod.running_var.copy_(1.0 + 0.4 * torch.cos(0.3 * k + len(name)))


def ref_evaluate_batch(model, inputs, mask, mask_img, lambdas=None, p=1):
    """main_missing.py:337-517 for one batch: model.eval(), no_grad, z = mu (phase='test')."""
    was = model.training
    model.eval()
    try:
        with torch.no_grad():
            return ref_forward_losses(model, inputs, mask, mask_img, lambdas, p, phase='test')
    finally:
        model.train(was)


def ref_reconstruction_metrics(target, pred):
    """compute_reconstruction_metrics (util.py:935-978) on numpy stacks (N, C, H, W): per sample, channel 0,
    both images shifted by their own min, data_range = max of the shifted target.  skimage is not in this
    image; its three functions are restated from their published definitions (skimage.metrics 0.16-0.19):
    mean_squared_error = mean((a-b)^2) in float64; peak_signal_noise_ratio = 10 log10(R^2 / mse);
    structural_similarity defaults = 7x7 uniform filter, sample covariance (NP/(NP-1)), K1 = 0.01,
    K2 = 0.03, mean of the map cropped by (win-1)//2.  PARITY UNPINNED for these three (no skimage here)."""
    from scipy.ndimage import uniform_filter
    out = {'ssim': [], 'psnr': [], 'rmse': []}
    for i in range(target.shape[0]):
        t = target[i, 0].astype(np.float64); p = pred[i, 0].astype(np.float64)
        t = t - t.min(); p = p - p.min()
        R = t.max()
        mse = np.mean((t - p) ** 2)
        out['rmse'].append(mse)                                             # util.py:963 (sic: MSE under 'rmse')
        out['psnr'].append(10 * np.log10(R * R / mse))
        win, NP = 7, 49
        cn = NP / (NP - 1.0)
        ux, uy = uniform_filter(t, win), uniform_filter(p, win)
        uxx, uyy, uxy = uniform_filter(t * t, win), uniform_filter(p * p, win), uniform_filter(t * p, win)
        vx, vy, vxy = cn * (uxx - ux * ux), cn * (uyy - uy * uy), cn * (uxy - ux * uy)
        C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        pad = (win - 1) // 2
        out['ssim'].append(S[pad:-pad, pad:-pad].mean())
    return out
