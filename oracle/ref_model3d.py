"""ORACLE -- TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product path).

CPU restatement, in plain torch fp32, of the reference's 3-D networks:
`BasicBlock` (src/model.py:1856-1876), `VAEBranch` (:1879-1949), `UNet3D` (:1952-2048), `NVNet3D` (:2050-2060).
Submodules are created in the reference's order with the reference's names, so a seeded construction gives the same
initial weights and `state_dict()` keys.  Pinned by tests/golden/nvnet3d_*.{json,npz}, which oracle/gen_golden.py
(selector `nv3d`) captured from the imported reference itself.

The reference never trains these classes (no loss / loop exists for them); gradients are pinned through
`nvnet_loss` below -- the objective of the paper the reference's docstring cites (Myronenko 2018: soft Dice +
0.1 * L2 of the VAE reconstruction + 0.1 * KL) -- applied identically to the reference's outputs when the goldens
were made.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class RefBasicBlock(nn.Module):                                            # model.py:1856-1876
    def __init__(self, in_channels, out_channels, n_groups=8):
        super().__init__()
        self.gn1 = nn.GroupNorm(n_groups, in_channels)
        self.relu1 = nn.ReLU()
        self.conv1 = nn.Conv3d(in_channels, out_channels, kernel_size=(3, 3, 3), padding=(1, 1, 1))
        self.gn2 = nn.GroupNorm(n_groups, in_channels)                     # QUIRK: in_channels (:1862)
        self.relu2 = nn.ReLU()
        self.conv2 = nn.Conv3d(out_channels, out_channels, kernel_size=(3, 3, 3), padding=(1, 1, 1))

    def forward(self, x):
        h = self.conv1(F.relu(self.gn1(x)))
        h = self.conv2(F.relu(self.gn2(h)))
        return h + x


class RefVAEBranch(nn.Module):                                             # model.py:1879-1949
    def __init__(self, input_shape, init_channels, out_channels, squeeze_channels=None):
        super().__init__()
        self.input_shape = input_shape
        self.squeeze_channels = squeeze_channels if squeeze_channels else init_channels * 4
        c = init_channels
        self.hidden_conv = nn.Sequential(nn.GroupNorm(8, c * 8), nn.ReLU(),
                                         nn.Conv3d(c * 8, self.squeeze_channels, (3, 3, 3), padding=(1, 1, 1)),
                                         nn.AdaptiveAvgPool3d(1))
        half = self.squeeze_channels // 2
        self.mu_fc = nn.Linear(half, half)
        self.logvar_fc = nn.Linear(half, half)
        recon_shape = int(np.prod(self.input_shape)) // (16 ** 3)
        self.reconstraction = nn.Sequential(nn.Linear(half, c * 8 * recon_shape), nn.ReLU())
        self.vconv4 = nn.Sequential(nn.Conv3d(c * 8, c * 8, (1, 1, 1)), nn.Upsample(scale_factor=2))
        self.vconv3 = nn.Sequential(nn.Conv3d(c * 8, c * 4, (3, 3, 3), padding=(1, 1, 1)), nn.Upsample(scale_factor=2),
                                    RefBasicBlock(c * 4, c * 4))
        self.vconv2 = nn.Sequential(nn.Conv3d(c * 4, c * 2, (3, 3, 3), padding=(1, 1, 1)), nn.Upsample(scale_factor=2),
                                    RefBasicBlock(c * 2, c * 2))
        self.vconv1 = nn.Sequential(nn.Conv3d(c * 2, c, (3, 3, 3), padding=(1, 1, 1)), nn.Upsample(scale_factor=2),
                                    RefBasicBlock(c, c))
        self.vconv0 = nn.Conv3d(c, out_channels, (1, 1, 1))

    def forward(self, x):
        h = self.hidden_conv(x)
        B = h.size(0)
        h = h.view((B, -1))
        half = self.squeeze_channels // 2
        mu = self.mu_fc(h[:, :half])
        logvar = self.logvar_fc(h[:, half:])
        std = torch.exp(0.5 * logvar)                                      # :1922-1926
        z = torch.randn_like(std).mul(std).add_(mu)
        re_x = self.reconstraction(z)
        re_x = re_x.view([B, -1, self.input_shape[0] // 16, self.input_shape[1] // 16, self.input_shape[2] // 16])
        x = self.vconv1(self.vconv2(self.vconv3(self.vconv4(re_x))))
        return self.vconv0(x), mu, logvar


class RefUNet3D(nn.Module):                                                # model.py:1952-2048
    def __init__(self, input_shape, in_channels=4, out_channels=3, init_channels=32, p=0.2):
        super().__init__()
        c = init_channels
        self.conv1a = nn.Conv3d(in_channels, c, (3, 3, 3), padding=(1, 1, 1))
        self.conv1b = RefBasicBlock(c, c)
        self.ds1 = nn.Conv3d(c, c * 2, (3, 3, 3), stride=(2, 2, 2), padding=(1, 1, 1))
        self.conv2a = RefBasicBlock(c * 2, c * 2)
        self.conv2b = RefBasicBlock(c * 2, c * 2)
        self.ds2 = nn.Conv3d(c * 2, c * 4, (3, 3, 3), stride=(2, 2, 2), padding=(1, 1, 1))
        self.conv3a = RefBasicBlock(c * 4, c * 4)
        self.conv3b = RefBasicBlock(c * 4, c * 4)
        self.ds3 = nn.Conv3d(c * 4, c * 8, (3, 3, 3), stride=(2, 2, 2), padding=(1, 1, 1))
        self.conv4a = RefBasicBlock(c * 8, c * 8)
        self.conv4b = RefBasicBlock(c * 8, c * 8)
        self.conv4c = RefBasicBlock(c * 8, c * 8)
        self.conv4d = RefBasicBlock(c * 8, c * 8)
        self.up4conva = nn.Conv3d(c * 8, c * 4, (1, 1, 1))
        self.up4 = nn.Upsample(scale_factor=2)
        self.up4convb = RefBasicBlock(c * 4, c * 4)
        self.up3conva = nn.Conv3d(c * 4, c * 2, (1, 1, 1))
        self.up3 = nn.Upsample(scale_factor=2)
        self.up3convb = RefBasicBlock(c * 2, c * 2)
        self.up2conva = nn.Conv3d(c * 2, c, (1, 1, 1))
        self.up2 = nn.Upsample(scale_factor=2)
        self.up2convb = RefBasicBlock(c, c)
        self.up1conv = nn.Conv3d(c, out_channels, (1, 1, 1))
        self.dropout = nn.Dropout(p=p)

    def forward(self, x):
        c1 = self.conv1b(self.conv1a(x))
        c2 = self.conv2b(self.conv2a(self.ds1(c1)))
        c3 = self.conv3b(self.conv3a(self.ds2(c2)))
        c4d = self.dropout(self.conv4d(self.conv4c(self.conv4b(self.conv4a(self.ds3(c3))))))
        u4 = self.up4convb(self.up4(self.up4conva(c4d)) + c3)
        u3 = self.up3convb(self.up3(self.up3conva(u4)) + c2)
        u2 = self.up2convb(self.up2(self.up2conva(u3)) + c1)
        return self.up1conv(u2), c4d


class RefNVNet3D(nn.Module):                                               # model.py:2050-2060
    def __init__(self, input_shape, in_channels=4, out_channels=3, init_channels=16, p=0.2):
        super().__init__()
        self.unet = RefUNet3D(input_shape, in_channels, out_channels, init_channels, p)
        self.vae_branch = RefVAEBranch(input_shape, init_channels, out_channels=in_channels)

    def forward(self, x):
        uout, c4d = self.unet(x)
        vout, mu, logvar = self.vae_branch(c4d)
        return uout, vout, mu, logvar


def nvnet_loss(uout, vout, mu, logvar, x, target):
    """Driver objective for the gradient parity (see module docstring): soft Dice + 0.1 L2 + 0.1 KL."""
    p = torch.sigmoid(uout)
    dice = 1 - 2 * (p * target).sum() / ((p * p).sum() + (target * target).sum() + 1e-6)
    l2 = ((vout - x) ** 2).mean()
    kl = (mu ** 2 + logvar.exp() - logvar - 1).sum() / x[0].numel()
    return dice + 0.1 * l2 + 0.1 * kl, {'dice': dice, 'l2': l2, 'kl': kl}


def make_inputs3d(B, C, shape, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, *shape, generator=g)
    t = (torch.rand(B, 3, *shape, generator=g) > 0.7).float()
    return x, t
