"""3-D networks of the reference on the HIP kernels: `BasicBlock`, `VAEBranch`, `UNet3D`,
`NVNet3D` (reference src/model.py:1856-2060) -- same class names, constructor arguments, submodule names (so
`state_dict()` keys and the seeded initialisation are the reference's), forward signatures and return values.

Activations are NDHWC (`torch.channels_last_3d`) fp32.  Every Conv3d (3x3x3 stride 1/2 and 1x1x1), every
GroupNorm+ReLU pair and every nearest x2 upsampling (+ skip) runs in csrc/mrdis_conv3d.hip / mrdis_elem3d.hip; the
VAE bottleneck (global average of an 8^3 map, three Linear layers on (B, 32) rows, reparameterisation) stays in torch.
There is no CPU fallback: without libmrdis_hip.so the first forward raises `MrdisLibraryError`.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import hip, ops

_GN_TAP = __import__('os').environ.get('MRDIS_GN_TAP', '1') != '0'      # BasicBlock: residual gradient summed inside the GroupNorm backward (0: autograd's add)


# --------------------------------------------------------------------------- autograd pairing
class _Conv3d(Function):
    """nn.Conv3d(k=3, padding=1, stride s) (+ fused residual addition)."""

    @staticmethod
    def forward(ctx, x, w_tck, w_tkc, bias, stride, residual):
        y = hip.conv3d_fwd(x, w_tck, bias, 3, stride, 1, residual)
        ctx.geom = (stride, tuple(x.shape), bias is not None, residual is not None)
        ctx.save_for_backward(x, w_tkc)
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, in_shape, has_bias, has_res = ctx.geom
        x, w_tkc = ctx.saved_tensors
        dx = hip.conv3d_bwd_data(dy, w_tkc, in_shape, 3, stride, 1) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[3]):
            dw, db = hip.conv3d_bwd_weight(x, dy, 3, stride, 1, has_bias)
        return dx, dw, None, db, None, (dy if has_res else None)


class _GroupNormReLU(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, relu):
        y, mean, rstd = hip.groupnorm_relu_fwd(x, gamma, beta, G, eps, relu)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (G, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        G, relu = ctx.cfg
        dx, dg, db = hip.groupnorm_relu_bwd(dy, x, gamma, beta, mean, rstd, G, relu)
        return dx, dg, db, None, None, None


class _GroupNormReLUTap(Function):
    """(relu(gn(x)), x): the second output is x itself for a consumer beside the normalised branch (BasicBlock's residual addition, model.py:1873).  Both
    gradients of x then arrive in ONE backward call and are summed inside the GroupNorm backward pass (mrdis_groupnorm_relu_bwd_add) instead of by an
    extra three-tensor add of autograd (537 MB per tensor at 4 x 16 x 128^3)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, relu):
        y, mean, rstd = hip.groupnorm_relu_fwd(x, gamma, beta, G, eps, relu)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (G, relu)
        return y, x.detach()

    @staticmethod
    def backward(ctx, dy, dx_tap):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        G, relu = ctx.cfg
        if dy is None:                      # (only the tap was used)
            return dx_tap, None, None, None, None, None
        dx, dg, db = hip.groupnorm_relu_bwd(dy, x, gamma, beta, mean, rstd, G, relu, add=dx_tap)
        return dx, dg, db, None, None, None


class _Upsample2xAdd(Function):
    @staticmethod
    def forward(ctx, x, skip):
        ctx.has_skip = skip is not None
        return hip.upsample2x_add_fwd(x, skip)

    @staticmethod
    def backward(ctx, dy):
        dx = hip.upsample2x_bwd(dy) if ctx.needs_input_grad[0] else None
        return dx, (dy if ctx.has_skip else None)


def groupnorm_relu(x, gn, relu=True):
    return _GroupNormReLU.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu)


def groupnorm_relu_tap(x, gn, relu=True):
    """(relu(gn(x)), x for the residual connection): see _GroupNormReLUTap"""
    return _GroupNormReLUTap.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu)


def upsample2x(x, skip=None):
    """nn.Upsample(scale_factor=2) (nearest) [+ skip]."""
    return _Upsample2xAdd.apply(x, skip)


class HipConv3d(nn.Conv3d):
    """nn.Conv3d whose forward/backward run in the HIP kernels; parameters, init and state_dict are nn.Conv3d's.
    Covers what the reference's 3-D nets use: 3x3x3 / padding 1 / stride 1 or 2, and 1x1x1."""

    def forward(self, x, residual=None):
        k = self.kernel_size
        if self.dilation != (1, 1, 1) or self.groups != 1 or self.padding_mode != 'zeros' or k[0] != k[1] or k[1] != k[2]:
            raise NotImplementedError
        one = torch.ones(1, dtype=torch.float32, device=x.device)
        Co, Ci = self.weight.shape[:2]
        if k[0] == 1:
            if self.stride != (1, 1, 1) or self.padding != (0, 0, 0) or residual is not None:
                raise NotImplementedError
            # 1x1x1: the 2-D kernels on the (N*D, H, W) view of the NDHWC tensor (no copy)
            x, _ = hip.ndhwc(x)
            N, _, D, H, W = x.shape
            x2 = x.permute(0, 2, 1, 3, 4).reshape(N * D, Ci, H, W)
            w_tck, w_tkc = ops.cached_mix((id(self), 0, 0), lambda: ops.mix_experts(self.weight.reshape(1, Co, Ci, 1, 1), one))
            y2 = ops.conv2d(x2, w_tck, w_tkc, self.bias, 1, 1, 1, 0, False)
            return y2.reshape(N, D, Co, H, W).permute(0, 2, 1, 3, 4)
        if k[0] != 3 or self.padding != (1, 1, 1) or self.stride[0] not in (1, 2) or len(set(self.stride)) != 1:
            raise NotImplementedError
        w_tck, w_tkc = ops.cached_mix((id(self), 0, 0), lambda: ops.mix_experts(self.weight.reshape(1, Co, Ci, 27, 1), one))
        return _Conv3d.apply(x, w_tck, w_tkc, self.bias, self.stride[0], residual)


# --------------------------------------------------------------------------- modules (model.py:1856-2060)
class BasicBlock(nn.Module):
    """model.py:1856-1876.  QUIRK kept: gn2 is built with `in_channels` (every call site has in == out)."""

    def __init__(self, in_channels, out_channels, n_groups=8):
        super().__init__()
        self.gn1 = nn.GroupNorm(n_groups, in_channels)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = HipConv3d(in_channels, out_channels, kernel_size=(3, 3, 3), padding=(1, 1, 1))
        self.gn2 = nn.GroupNorm(n_groups, in_channels)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = HipConv3d(out_channels, out_channels, kernel_size=(3, 3, 3), padding=(1, 1, 1))

    def forward(self, x):
        if _GN_TAP and x.requires_grad and torch.is_grad_enabled():
            g, x = groupnorm_relu_tap(x, self.gn1)                         # the residual path's gradient joins the branch's inside the GroupNorm backward
        else:
            g = groupnorm_relu(x, self.gn1)
        h = self.conv1(g)
        return self.conv2(groupnorm_relu(h, self.gn2), residual=x)        # `x + residul` in the conv epilogue


class _ConvUp(nn.Sequential):
    """nn.Sequential(Conv3d, Upsample[, BasicBlock]) with the reference's child indices (state_dict keys '0', '2')."""

    def forward(self, x):
        x = upsample2x(self[0](x))
        return self[2](x) if len(self) > 2 else x


class VAEBranch(nn.Module):
    """model.py:1879-1949."""

    def __init__(self, input_shape, init_channels, out_channels, squeeze_channels=None):
        super().__init__()
        self.input_shape = input_shape
        self.squeeze_channels = squeeze_channels if squeeze_channels else init_channels * 4
        c = init_channels
        self.hidden_conv = nn.Sequential(nn.GroupNorm(8, c * 8), nn.ReLU(inplace=True),
                                         HipConv3d(c * 8, self.squeeze_channels, (3, 3, 3), padding=(1, 1, 1)),
                                         nn.AdaptiveAvgPool3d(1))
        self.mu_fc = nn.Linear(self.squeeze_channels // 2, self.squeeze_channels // 2)
        self.logvar_fc = nn.Linear(self.squeeze_channels // 2, self.squeeze_channels // 2)
        recon_shape = int(np.prod(self.input_shape)) // (16 ** 3)
        self.reconstraction = nn.Sequential(nn.Linear(self.squeeze_channels // 2, c * 8 * recon_shape), nn.ReLU(inplace=True))
        self.vconv4 = _ConvUp(HipConv3d(c * 8, c * 8, (1, 1, 1)), nn.Upsample(scale_factor=2))
        self.vconv3 = _ConvUp(HipConv3d(c * 8, c * 4, (3, 3, 3), padding=(1, 1, 1)), nn.Upsample(scale_factor=2),
                              BasicBlock(c * 4, c * 4))
        self.vconv2 = _ConvUp(HipConv3d(c * 4, c * 2, (3, 3, 3), padding=(1, 1, 1)), nn.Upsample(scale_factor=2),
                              BasicBlock(c * 2, c * 2))
        self.vconv1 = _ConvUp(HipConv3d(c * 2, c, (3, 3, 3), padding=(1, 1, 1)), nn.Upsample(scale_factor=2),
                              BasicBlock(c, c))
        self.vconv0 = HipConv3d(c, out_channels, (1, 1, 1))

    def reparameterize(self, mu, logvar):
        std = torch.exp(0.5 * logvar)
        eps = ops.to_device(torch.randn(std.shape, dtype=std.dtype), std.device)   # CPU generator, as the reference on its CPU path
        return eps.mul(std).add_(mu)

    def forward(self, x):
        h = self.hidden_conv[2](groupnorm_relu(x, self.hidden_conv[0]))
        batch_size = h.size(0)
        h = h.mean(dim=(2, 3, 4)).view((batch_size, -1))                  # AdaptiveAvgPool3d(1)
        half = self.squeeze_channels // 2
        mu = self.mu_fc(h[:, :half])
        logvar = self.logvar_fc(h[:, half:])
        z = self.reparameterize(mu, logvar)
        re_x = self.reconstraction(z)
        re_x = re_x.view([batch_size, -1, self.input_shape[0] // 16, self.input_shape[1] // 16, self.input_shape[2] // 16])
        x = self.vconv4(re_x)
        x = self.vconv3(x)
        x = self.vconv2(x)
        x = self.vconv1(x)
        return self.vconv0(x), mu, logvar


class UNet3D(nn.Module):
    """model.py:1952-2048."""

    def __init__(self, input_shape, in_channels=4, out_channels=3, init_channels=32, p=0.2):
        super().__init__()
        self.input_shape = input_shape
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.init_channels = init_channels
        self.make_encoder()
        self.make_decoder()
        self.dropout = nn.Dropout(p=p)

    def make_encoder(self):
        c = self.init_channels
        self.conv1a = HipConv3d(self.in_channels, c, (3, 3, 3), padding=(1, 1, 1))
        self.conv1b = BasicBlock(c, c)
        self.ds1 = HipConv3d(c, c * 2, (3, 3, 3), stride=(2, 2, 2), padding=(1, 1, 1))
        self.conv2a = BasicBlock(c * 2, c * 2)
        self.conv2b = BasicBlock(c * 2, c * 2)
        self.ds2 = HipConv3d(c * 2, c * 4, (3, 3, 3), stride=(2, 2, 2), padding=(1, 1, 1))
        self.conv3a = BasicBlock(c * 4, c * 4)
        self.conv3b = BasicBlock(c * 4, c * 4)
        self.ds3 = HipConv3d(c * 4, c * 8, (3, 3, 3), stride=(2, 2, 2), padding=(1, 1, 1))
        self.conv4a = BasicBlock(c * 8, c * 8)
        self.conv4b = BasicBlock(c * 8, c * 8)
        self.conv4c = BasicBlock(c * 8, c * 8)
        self.conv4d = BasicBlock(c * 8, c * 8)

    def make_decoder(self):
        c = self.init_channels
        self.up4conva = HipConv3d(c * 8, c * 4, (1, 1, 1))
        self.up4 = nn.Upsample(scale_factor=2)
        self.up4convb = BasicBlock(c * 4, c * 4)
        self.up3conva = HipConv3d(c * 4, c * 2, (1, 1, 1))
        self.up3 = nn.Upsample(scale_factor=2)
        self.up3convb = BasicBlock(c * 2, c * 2)
        self.up2conva = HipConv3d(c * 2, c, (1, 1, 1))
        self.up2 = nn.Upsample(scale_factor=2)
        self.up2convb = BasicBlock(c, c)
        self.up1conv = HipConv3d(c, self.out_channels, (1, 1, 1))

    def forward(self, x):
        c1 = self.conv1b(self.conv1a(x))
        c2 = self.conv2b(self.conv2a(self.ds1(c1)))
        c3 = self.conv3b(self.conv3a(self.ds2(c2)))
        c4d = self.conv4d(self.conv4c(self.conv4b(self.conv4a(self.ds3(c3)))))
        c4d = self.dropout(c4d)
        u4 = self.up4convb(upsample2x(self.up4conva(c4d), c3))             # up + skip in one kernel
        u3 = self.up3convb(upsample2x(self.up3conva(u4), c2))
        u2 = self.up2convb(upsample2x(self.up2conva(u3), c1))
        return self.up1conv(u2), c4d


class NVNet3D(nn.Module):
    """model.py:2050-2060."""

    def __init__(self, input_shape, in_channels=4, out_channels=3, init_channels=16, p=0.2):
        super().__init__()
        self.unet = UNet3D(input_shape, in_channels, out_channels, init_channels, p)
        self.vae_branch = VAEBranch(input_shape, init_channels, out_channels=in_channels)

    def forward(self, x):
        uout, c4d = self.unet(x)
        vout, mu, logvar = self.vae_branch(c4d)
        return uout, vout, mu, logvar


def nvnet_loss(uout, vout, mu, logvar, x, target):
    """The reference ships no objective for NVNet3D; this is the one of the paper its docstring cites (Myronenko 2018:
    soft Dice of sigmoid(uout) + 0.1 * L2 of the VAE reconstruction + 0.1 * KL), used by bench3d.py and the parity tests
    to drive the backward pass."""
    p = torch.sigmoid(uout)
    dice = 1 - 2 * (p * target).sum() / ((p * p).sum() + (target * target).sum() + 1e-6)
    l2 = ((vout - x) ** 2).mean()
    kl = (mu ** 2 + logvar.exp() - logvar - 1).sum() / x[0].numel()
    return dice + 0.1 * l2 + 0.1 * kl, {'dice': dice, 'l2': l2, 'kl': kl}
