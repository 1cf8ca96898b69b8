"""Host-side mirror of the reference's operator / module interface for the hot
path (reference: src/model.py, cited per class).  Same class names, constructor
arguments, `forward` signatures and state_dict keys/shapes as the reference, so
`config.yaml`, a `main_missing.py`-style entry and checkpoints carry over; every
convolution / norm / resize / softmax / loss reduction below executes in the
hand-written gfx950 kernels of libmrdis_hip.so (see ops.py / hip.py).

Reference quirks that parity depends on are reproduced and marked `QUIRK`.
"""
import copy
import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# =============================================================================
# operator seam: model.py:2065-2120
# =============================================================================
class _routing(nn.Module):
    """model.py:2065-2073."""

    def __init__(self, in_channels, num_experts):
        super().__init__()
        self.fc = nn.Linear(in_channels, num_experts)

    def forward(self, inputs_type):
        return torch.sigmoid(self.fc(inputs_type))


class CondConv2d(nn.Module):
    """Drop-in for the reference CondConv2d (model.py:2075-2117).

    forward(inputs (B,Cin,H,W), inputs_type (B,embeddings)) -> (B,Cout,H',W').
    Parameters: weight (E,Cout,Cin,kh,kw), bias (Cout), _routing_fn.fc.{weight,bias}.

    When `inputs_type` is provably constant over the batch (a stride-0 expand of
    one row -- what every call site of the reference amounts to, model.py:3138,
    3169, 3190, 3211) the experts are mixed ONCE and one batched conv runs;
    otherwise the per-sample path of the reference is taken (one mix + one
    batch-1 conv per sample).  Both are the same arithmetic.
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 embeddings=1, bias=True, padding_mode='zeros', num_experts=3, dropout_rate=0):
        super().__init__()
        if _pair(dilation) != (1, 1) or groups != 1 or padding_mode != 'zeros':
            raise NotImplementedError('mrdis CondConv2d covers dilation=1, groups=1, zero padding (all the path uses)')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        if self.stride[0] != self.stride[1] or self.padding[0] != self.padding[1]:
            raise NotImplementedError('square stride / padding only')
        self.num_experts = num_experts
        # identical RNG consumption to the reference constructor (model.py:2083-2097)
        burn = nn.Conv2d(in_channels, out_channels, self.kernel_size, stride, padding)
        self._routing_fn = _routing(embeddings, num_experts)
        self.weight = nn.Parameter(torch.empty(num_experts, out_channels, in_channels, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        del burn
        nn.init.xavier_normal_(self.weight)
        if bias:
            nn.init.constant_(self.bias, 0)
            self.bias._mrdis_sink = True           # gradient accumulated in-kernel (ops._grad_sink)
        # expert weights and routing parameters: the multi-label mixing backward adds into their gradient buffers in-kernel
        self.weight._mrdis_sink = True
        self._routing_fn.fc.weight._mrdis_sink = True
        self._routing_fn.fc.bias._mrdis_sink = True

    def _mixed(self, t_row):
        """t_row: (1, embeddings) -> (w_tck, w_tkc) for that type."""
        return ops.mix_experts_routed(self.weight, self._routing_fn.fc.weight, self._routing_fn.fc.bias, t_row)

    def mixed_uniform(self, inputs_type):
        """(w_tck, w_tkc) for a batch-constant `inputs_type` (stride-0 expand of one row), memoised for the step."""
        hit = ops.lookup_type_row(inputs_type) if ops.mix_cache_active() else None
        if hit is not None:            # one of the model's modality labels: all labels are mixed in one launch per step
            table, row = hit
            key = (id(self), 'all', table.data_ptr(), table._version)
            allw = ops.cached_mix(key, lambda: ops.mix_experts_routed_all(
                self.weight, self._routing_fn.fc.weight, self._routing_fn.fc.bias, table))
            return allw[2 * row], allw[2 * row + 1]
        key = (id(self), inputs_type.data_ptr(), inputs_type._version)
        return ops.cached_mix(key, lambda: self._mixed(inputs_type[:1]))

    def forward(self, inputs, inputs_type, lrelu=False):
        kh, kw = self.kernel_size
        B = inputs.shape[0]
        if B == 1 or inputs_type.stride(0) == 0:
            if not ops.mix_cache_active() and not ops.storage_bf16():
                # the seam as ONE dispatcher-visible op (mix + conv, torch.ops.mrdis.cond_conv2d): plain module use
                return torch.ops.mrdis.cond_conv2d(inputs, inputs_type[:1], self.weight, self._routing_fn.fc.weight,
                                                   self._routing_fn.fc.bias, self.bias, self.stride[0], self.padding[0], lrelu,
                                                   ops.compute_dtype())        # 0 exact fp32 | 1 bf16 MFMA operands
            # inside a training step the mixed kernels of all modality labels are cached and shared by every call
            w_tck, w_tkc = self.mixed_uniform(inputs_type)
            return ops.conv2d(inputs, w_tck, w_tkc, self.bias, kh, kw, self.stride[0], self.padding[0], lrelu, co=self.out_channels)
        outs = []                                    # per-sample path, model.py:2114-2117
        for i in range(B):
            w_tck, w_tkc = self._mixed(inputs_type[i:i + 1])
            outs.append(ops.conv2d(inputs[i:i + 1], w_tck, w_tkc, self.bias, kh, kw, self.stride[0], self.padding[0], lrelu))
        return torch.cat(outs, 0)


    def forward_grouped(self, x, types, lrelu=False):
        """x: G sample blocks (block g belongs to the batch-constant type tensor types[g]) -> the G outputs, batch-concatenated:
        ONE op for the G calls of the reference's per-modality loop (ops.conv2d_grouped)."""
        kh, kw = self.kernel_size
        G = len(types)
        if ops.storage_bf16() and self.stride[0] != 1 and min(self.in_channels, self.out_channels) < 16:
            # the stride-2 first layers under bf16 storage keep ops.conv2d's policy (fp32 kernels between view casts), block by block
            B = x.shape[0] // G
            return torch.cat([self.forward(x[g * B:(g + 1) * B], types[g], lrelu) for g in range(G)], 0)
        return ops.conv2d_grouped(x, [self.mixed_uniform(t) for t in types], self.bias, kh, kw, self.padding[0], lrelu=lrelu,
                                  co=self.out_channels, stride=self.stride[0])


class HipConv2d(nn.Conv2d):
    """nn.Conv2d whose forward/backward run in the HIP kernels (discriminator,
    model.py:2773-2789).  Keeps nn.Conv2d's parameters, init and state_dict."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.bias is not None:
            self.bias._mrdis_sink = True           # gradient accumulated in-kernel (ops._grad_sink)

    def forward(self, x, lrelu=False):
        if self.dilation != (1, 1) or self.groups != 1 or self.padding_mode != 'zeros':
            raise NotImplementedError
        kh, kw = self.kernel_size
        one = torch.ones(1, dtype=torch.float32, device=x.device)
        w_tck, w_tkc = ops.cached_mix((id(self), 0, 0), lambda: ops.mix_experts(self.weight.unsqueeze(0), one))
        return ops.conv2d(x, w_tck, w_tkc, self.bias, kh, kw, self.stride[0], self.padding[0], lrelu)


def Conv2d(is_cond):
    """model.py:2119-2120 -- the plug point every block goes through."""
    return CondConv2d if is_cond else HipConv2d


_BN_PENDING = weakref.WeakSet()   # BatchNorm2d modules whose num_batches_tracked is behind by `_nbt_pending` calls (weak: a discarded model is not kept alive)


def flush_batch_counters():
    """bring every BatchNorm2d.num_batches_tracked up to date (one fused add over all pending layers).  Called at the end of a
    TrainStep, by EvalStep, and by every way of reading the buffer out of a module: state_dict(), train(False) / eval(),
    deepcopy (EMA copies) and pickling (torch.save(model))."""
    if not _BN_PENDING:
        return
    pend = list(_BN_PENDING)
    mods = [m for m in pend if getattr(m, '_nbt_pending', 0) and m.num_batches_tracked is not None]
    if mods:
        torch._foreach_add_([m.num_batches_tracked for m in mods], [int(m._nbt_pending) for m in mods])
    for m in pend:
        m._nbt_pending = 0
    _BN_PENDING.clear()


def discard_batch_counters():
    """forget the pending counter increments (a graph recording that failed half-way ran forward code whose kernels never executed)"""
    for m in list(_BN_PENDING):
        m._nbt_pending = 0
    _BN_PENDING.clear()


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d; training-mode forward/backward in HIP."""

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        flush_batch_counters()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._nbt_pending = 0
        _BN_PENDING.discard(self)
        super()._load_from_state_dict(*args, **kwargs)

    def train(self, mode=True):
        if not mode:
            flush_batch_counters()
        return super().train(mode)

    def __deepcopy__(self, memo):
        flush_batch_counters()                     # the copy starts with an exact counter and nothing pending
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        flush_batch_counters()
        return super().__getstate__() if hasattr(super(), '__getstate__') else self.__dict__

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.affine:
            self.weight._mrdis_sink = True         # gradients accumulated in-kernel (ops._grad_sink)
            self.bias._mrdis_sink = True

    def forward(self, x, into=None, groups=1):
        """into = (buf, c0): training mode writes the result into channels [c0, c0 + C) of `buf` and returns that view.
        groups = G: x holds G sample blocks that the reference sends through this layer in G calls (statistics per block)."""
        if self.training:
            if self.num_batches_tracked is not None:
                # the counter (state_dict key of the reference's layers; momentum is fixed, so nothing reads it during training) is advanced
                # on the host and written back by flush_batch_counters (end of a step, state_dict()): 72 one-element launches per step fewer
                self._nbt_pending = getattr(self, '_nbt_pending', 0) + groups
                _BN_PENDING.add(self)
            return ops.batch_norm_train(x, self.weight, self.bias, self.running_mean, self.running_var,
                                        self.eps, self.momentum, into, groups)
        assert groups == 1
        if not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)):
            return ops.hip.bn_eval_fwd(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps)   # evaluate()
        # eval-mode BatchNorm under autograd (fine-tuning with frozen statistics) is not on the path: plain torch
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, self.eps)


def expand_type(value, batch, device):
    """(B,1) modality label, constant over the batch, as a stride-0 expand so CondConv2d can
    prove it (the reference builds (1+i)*ones(B,1), model.py:3138)."""
    return torch.full((1, 1), float(value), dtype=torch.float32, device=device).expand(batch, 1)


# =============================================================================
# anatomy U-Net: model.py:2122-2195, 2218-2245, 2271-2296
# =============================================================================
class Conv_BN_Act_New(nn.Module):
    """model.py:2122-2153.  QUIRK: the if/if/if-else chain (:2134-2141) makes `act`
    the identity for 'lrelu'/'relu' (only 'elu' survives)."""

    def __init__(self, in_num_ch, out_num_ch, filter_size=4, stride=2, padding=1, activation='lrelu', is_bn=True, is_cond=False):
        super().__init__()
        self.is_bn, self.is_cond = is_bn, is_cond
        self.conv = Conv2d(is_cond)(in_num_ch, out_num_ch, filter_size, stride, padding=padding)
        if is_bn:
            self.bn = BatchNorm2d(out_num_ch)
        self.act = nn.ELU(inplace=True) if activation == 'elu' else nn.Sequential()

    def forward(self, x, inputs_type=None, skip=False):
        """skip: this output is also the skip half of the decoder's concatenation (model.py:2192) -- BatchNorm writes it straight into
        the first C channels of a 2C-channel buffer that the decoder level completes (ops.cat_join): no concatenation copy"""
        x = self.conv(x, inputs_type) if self.is_cond else self.conv(x)
        if self.is_bn:
            if (skip and self.training and x.is_cuda and ops.cat_elision() and isinstance(self.act, nn.Sequential) and len(self.act) == 0):
                N, C, H, W = x.shape
                buf = ops.hip.empty_nhwc(N, 2 * C, H, W, x.device, x.dtype)
                x = self.bn(x, into=(buf, 0))
                x._mrdis_catbuf = buf
                return x
            x = self.bn(x)
        return self.act(x)


    def forward_grouped(self, x, types, skip=False):
        """forward for G modality labels on the batch-concatenated input (block g: types[g]); BatchNorm statistics per block"""
        G = len(types)
        x = self.conv.forward_grouped(x, types)
        if self.is_bn:
            if (skip and self.training and x.is_cuda and ops.cat_elision() and isinstance(self.act, nn.Sequential) and len(self.act) == 0):
                N, C, H, W = x.shape
                buf = ops.hip.empty_nhwc(N, 2 * C, H, W, x.device, x.dtype)
                x = self.bn(x, into=(buf, 0), groups=G)
                x._mrdis_catbuf = buf
                return x
            x = self.bn(x, groups=G)
        return self.act(x)


class Act_Deconv_BN_Concat_New(nn.Module):
    """model.py:2155-2195.  identity act (same QUIRK) -> bilinear x2 align_corners=True ->
    conv3x3 -> [BN -> cat(skip)]; `bn` is created even when is_last (:2179)."""

    def __init__(self, in_num_ch, out_num_ch, filter_size=3, stride=1, padding=1, activation='relu', upsample=True,
                 is_last=False, is_bn=True, is_cond=False):
        super().__init__()
        if not upsample:
            raise NotImplementedError('ConvTranspose2d variant is not on the path (model.py:2178)')
        self.is_bn, self.is_cond, self.is_last = is_bn, is_cond, is_last
        self.act = nn.ELU(inplace=True) if activation == 'elu' else nn.Sequential()
        self.conv = Conv2d(is_cond)(in_num_ch, out_num_ch, filter_size, stride, padding=padding)
        self.bn = BatchNorm2d(out_num_ch)

    def forward(self, x_down, x_up, inputs_type=None):
        x_up = self.act(x_up)
        x_up = ops.bilinear(x_up, (2 * x_up.shape[2], 2 * x_up.shape[3]), True)
        x_up = self.conv(x_up, inputs_type) if self.is_cond else self.conv(x_up)
        if self.is_last:
            return x_up
        if self.is_bn and self.training and x_up.is_cuda and ops.cat_elision() and x_down.dtype == x_up.dtype and isinstance(self.act, nn.Sequential) \
                and x_down.shape[0] == x_up.shape[0] and x_down.shape[2:] == x_up.shape[2:]:
            Cd, Cu = x_down.shape[1], x_up.shape[1]
            buf = getattr(x_down, '_mrdis_catbuf', None)
            if buf is not None and buf.shape[1] == Cd + Cu and buf.dtype == x_up.dtype:
                x_down._mrdis_catbuf = None                   # one consumer only: a second decoder reading this skip tensor must not rewrite the upper half
                x_up = self.bn(x_up, into=(buf, Cd))
                return ops.cat_join(x_down, x_up, buf)
        if self.is_bn:
            x_up = self.bn(x_up)
        return torch.cat([x_down, x_up], 1)


    def forward_grouped(self, x_down, x_up, types):
        G = len(types)
        x_up = self.act(x_up)
        x_up = ops.bilinear(x_up, (2 * x_up.shape[2], 2 * x_up.shape[3]), True)
        x_up = self.conv.forward_grouped(x_up, types)
        if self.is_last:
            return x_up
        if self.is_bn and self.training and x_up.is_cuda and ops.cat_elision() and x_down.dtype == x_up.dtype and isinstance(self.act, nn.Sequential) \
                and x_down.shape[0] == x_up.shape[0] and x_down.shape[2:] == x_up.shape[2:]:
            Cd, Cu = x_down.shape[1], x_up.shape[1]
            buf = getattr(x_down, '_mrdis_catbuf', None)
            if buf is not None and buf.shape[1] == Cd + Cu and buf.dtype == x_up.dtype:
                x_down._mrdis_catbuf = None
                x_up = self.bn(x_up, into=(buf, Cd), groups=G)
                return ops.cat_join(x_down, x_up, buf)
        if self.is_bn:
            x_up = self.bn(x_up, groups=G)
        return torch.cat([x_down, x_up], 1)


class AnatomyEncoderEncNew(nn.Module):
    """model.py:2218-2245."""

    def __init__(self, in_num_ch=7, first_num_ch=32, is_cond=False):
        super().__init__()
        self.is_cond = is_cond
        c = first_num_ch
        self.down_1 = Conv2d(is_cond)(in_num_ch, c, 4, 2, padding=1)
        self.act_1 = nn.LeakyReLU(0.2, inplace=True)          # fused into down_1's epilogue
        self.down_2 = Conv_BN_Act_New(c, 2 * c, is_cond=is_cond)
        self.down_3 = Conv_BN_Act_New(2 * c, 4 * c, is_cond=is_cond)
        self.down_4 = Conv_BN_Act_New(4 * c, 8 * c, is_cond=is_cond)
        self.down_5 = Conv_BN_Act_New(8 * c, 8 * c, activation='no', is_cond=is_cond)

    def forward(self, x, inputs_type=None):
        d1 = self.down_1(x, inputs_type, lrelu=True) if self.is_cond else self.down_1(x, lrelu=True)
        d2 = self.down_2(d1, inputs_type, skip=True)
        d3 = self.down_3(d2, inputs_type, skip=True)
        d4 = self.down_4(d3, inputs_type, skip=True)
        d5 = self.down_5(d4, inputs_type)
        return [d1, d2, d3, d4, d5]

    def forward_grouped(self, x, types):
        d1 = self.down_1.forward_grouped(x, types, lrelu=True)
        d2 = self.down_2.forward_grouped(d1, types, skip=True)
        d3 = self.down_3.forward_grouped(d2, types, skip=True)
        d4 = self.down_4.forward_grouped(d3, types, skip=True)
        d5 = self.down_5.forward_grouped(d4, types)
        return [d1, d2, d3, d4, d5]


class AnatomyEncoderDecNew(nn.Module):
    """model.py:2271-2296."""

    def __init__(self, first_num_ch=32, out_num_ch=8, output_act='softmax', is_cond=False):
        super().__init__()
        c = first_num_ch
        self.up_4 = Act_Deconv_BN_Concat_New(8 * c, 8 * c, is_cond=is_cond)
        self.up_3 = Act_Deconv_BN_Concat_New(16 * c, 4 * c, is_cond=is_cond)
        self.up_2 = Act_Deconv_BN_Concat_New(8 * c, 2 * c, is_cond=is_cond)
        self.up_1 = Act_Deconv_BN_Concat_New(4 * c, c, is_cond=is_cond)
        self.output = Act_Deconv_BN_Concat_New(2 * c, out_num_ch, is_last=True, is_cond=is_cond)

    def forward(self, down_list, inputs_type=None, need_out=True):
        """need_out = False: stop behind up_1 -- the caller does not read the anatomy map (second encoder pass of main_missing.py:228-231 when the modality
        encoder takes no s: the map is dead there); every BatchNorm of the network has run by then, so the module's state is what the full pass leaves."""
        if inputs_type is None:
            inputs_type = expand_type(1.0, down_list[0].shape[0], down_list[0].device)   # :2286-2287
        u4 = self.up_4(down_list[3], down_list[4], inputs_type)
        u3 = self.up_3(down_list[2], u4, inputs_type)
        u2 = self.up_2(down_list[1], u3, inputs_type)
        u1 = self.up_1(down_list[0], u2, inputs_type)
        if not need_out:
            return None, None
        out = self.output(None, u1, inputs_type)
        return out, out

    def forward_grouped(self, down_list, types, need_out=True):
        u4 = self.up_4.forward_grouped(down_list[3], down_list[4], types)
        u3 = self.up_3.forward_grouped(down_list[2], u4, types)
        u2 = self.up_2.forward_grouped(down_list[1], u3, types)
        u1 = self.up_1.forward_grouped(down_list[0], u2, types)
        if not need_out:
            return None, None
        out = self.output.forward_grouped(None, u1, types)
        return out, out


# =============================================================================
# modality encoder: model.py:2332-2400
# =============================================================================
class ModalityEncoderNew(nn.Module):
    """model.py:2332-2400.  `convs` is dead code that still owns parameters and
    state_dict keys (:2346-2357).  `feat_hw` generalises the hard-coded 5*6
    (:2360, :2396) to (H/32)*(W/32); 30 reproduces the reference."""

    def __init__(self, img_num_ch=7, s_num_ch=8, first_num_ch=16, z_size=16, is_cond=False, feat_hw=30):
        super().__init__()
        self.s_num_ch, self.is_cond = s_num_ch, is_cond
        c = first_num_ch
        ch = [img_num_ch + s_num_ch, c, 2 * c, 4 * c, 8 * c, 8 * c]
        conv2d = Conv2d(is_cond)
        for i in range(5):
            setattr(self, f'conv{i + 1}', conv2d(ch[i], ch[i + 1], 3, 2, padding=1))
        dead = []
        for i in range(5):
            dead += [nn.Conv2d(ch[i], ch[i + 1], 3, 2, padding=1), nn.LeakyReLU(0.2, inplace=True)]
        self.convs = nn.Sequential(*dead)
        self.fcs = nn.Sequential(nn.Linear(feat_hw * 8 * c, 2 * z_size), nn.LeakyReLU(0.2, inplace=True))
        self.mean = nn.Linear(2 * c, z_size)
        self.log_var = nn.Linear(2 * c, z_size)

    def forward(self, xi, si, inputs_type=None):
        x = xi if self.s_num_ch == 0 else torch.cat([xi, si], 1)
        for i in range(5):
            conv = getattr(self, f'conv{i + 1}')
            x = conv(x, inputs_type, lrelu=True) if self.is_cond else conv(x, lrelu=True)   # :2374-2383
        # `view(-1, 5*6*128)` flattens NCHW order (:2396): make that order physical (tiny tensor)
        x = x.contiguous(memory_format=torch.contiguous_format).reshape(x.shape[0], -1).float()     # bf16 storage: the Linears stay fp32
        x = self.fcs(x)
        return self.mean(x), self.log_var(x)

    def forward_grouped(self, xi, types):
        """the G per-modality calls on the batch-concatenated input (the Linear layers are type-independent)"""
        assert self.s_num_ch == 0 and self.is_cond
        x = xi
        for i in range(5):
            x = getattr(self, f'conv{i + 1}').forward_grouped(x, types, lrelu=True)
        x = x.contiguous(memory_format=torch.contiguous_format).reshape(x.shape[0], -1).float()
        x = self.fcs(x)
        return self.mean(x), self.log_var(x)


# =============================================================================
# SPADE decoder: model.py:2424-2454, 2540-2632
# =============================================================================
class SPADEBlockNew(nn.Module):
    """model.py:2424-2454."""

    def __init__(self, input_size, in_num_ch=128, out_num_ch=128, s_num_ch=8, is_cond=False):
        super().__init__()
        self.is_cond, self.input_size = is_cond, tuple(input_size)
        conv2d = Conv2d(is_cond)
        self.zi_layers = nn.InstanceNorm2d(in_num_ch)     # parameter-free; fused with the modulation
        self.si_layers = conv2d(s_num_ch, in_num_ch, 3, 1, padding=1)
        self.gamma = conv2d(in_num_ch, in_num_ch, 3, 1, padding=1)
        self.beta = conv2d(in_num_ch, in_num_ch, 3, 1, padding=1)
        self.out = conv2d(in_num_ch, out_num_ch, 3, 1, padding=1)

    def forward(self, si, zi, inputs_type=None):
        # the resize depends on (s_i, scale) only, not on the modality type: one resize per step serves
        # every decoder call that uses this s_i (autograd sums the gradients of all uses)
        size = self.input_size
        src = si                                   # the cache entry keeps `src` alive, so id(src) cannot be recycled
        si = ops.step_cache(('bil', id(src), size), lambda: (src, ops.bilinear(src, size, False)))[1]     # :2432/:2441
        t = (inputs_type,) if self.is_cond else ()
        si_out = self.si_layers(si, *t)
        if self.is_cond and (si_out.shape[0] == 1 or inputs_type.stride(0) == 0):
            # gamma and beta read the same input with the same geometry: run them as ONE convolution with
            # concatenated (mixed) filters -- one pass over si_out forward, one K-doubled pass backward
            kh, kw = self.gamma.kernel_size
            key = (id(self), 'gb', inputs_type.data_ptr(), inputs_type._version)

            def fused():
                g_tck, g_tkc = self.gamma.mixed_uniform(inputs_type)
                b_tck, b_tkc = self.beta.mixed_uniform(inputs_type)
                return (torch.cat([g_tck, b_tck], 2), torch.cat([g_tkc, b_tkc], 1),
                        torch.cat([self.gamma.bias, self.beta.bias]))
            hit = ops.lookup_type_row(inputs_type) if ops.mix_cache_active() else None
            if hit is not None and self.gamma.weight.shape == self.beta.weight.shape:
                # one of the model's modality labels: gamma and beta are mixed for ALL labels straight into the fused filters
                table, row = hit
                allw = ops.cached_mix((id(self), 'gb_all', table.data_ptr(), table._version),
                                      lambda: ops.mix_pair_fused_all(self.gamma, self.beta, table))
                bias = ops.cached_mix((id(self), 'gb_bias'), lambda: torch.cat([self.gamma.bias, self.beta.bias]))
                w_tck, w_tkc = allw[2 * row], allw[2 * row + 1]
            else:
                w_tck, w_tkc, bias = ops.step_cache(key, fused)
            if ((kh, kw) == (3, 3) and self.gamma.padding[0] == 1 and type(si_out) is torch.Tensor and si_out.is_cuda
                    and ops.mix_cache_active() and torch.is_grad_enabled()):
                mix = ops.gb_spade(si_out, zi, [(w_tck, w_tkc)], bias, self.zi_layers.eps)            # :2440 + :2446, one node
            else:
                gb = ops.conv2d(si_out, w_tck, w_tkc, bias, kh, kw, 1, self.gamma.padding[0])
                mix = ops.instnorm_spade_gb(zi, gb, self.zi_layers.eps)       # :2440 + :2446
        else:
            gamma = self.gamma(si_out, *t)
            beta = self.beta(si_out, *t)
            mix = ops.instnorm_spade(zi, gamma, beta, self.zi_layers.eps)
        return self.out(mix, *t)

    def _fused_gb(self, inputs_type):
        """((w_tck, w_tkc), bias) of the fused gamma | beta convolution for one modality label (see forward)."""
        hit = ops.lookup_type_row(inputs_type) if ops.mix_cache_active() else None
        if hit is not None and self.gamma.weight.shape == self.beta.weight.shape:
            table, row = hit
            allw = ops.cached_mix((id(self), 'gb_all', table.data_ptr(), table._version),
                                  lambda: ops.mix_pair_fused_all(self.gamma, self.beta, table))
            bias = ops.cached_mix((id(self), 'gb_bias'), lambda: torch.cat([self.gamma.bias, self.beta.bias]))
            return (allw[2 * row], allw[2 * row + 1]), bias
        g_tck, g_tkc = self.gamma.mixed_uniform(inputs_type)
        b_tck, b_tkc = self.beta.mixed_uniform(inputs_type)
        return (torch.cat([g_tck, b_tck], 2), torch.cat([g_tkc, b_tkc], 1)), torch.cat([self.gamma.bias, self.beta.bias])

    def forward_grouped(self, si, z_cat, types):
        """The block for G modality labels at once: z_cat holds G sample blocks of B (block g belongs to types[g]), `si` the B anatomy
        maps every block reads.  One launch per normalisation kernel, one op per convolution layer (ops.conv2d_grouped)."""
        size = self.input_size
        src = si
        si = ops.step_cache(('bil', id(src), size), lambda: (src, ops.bilinear(src, size, False)))[1]
        kh, kw = self.gamma.kernel_size
        pad = self.gamma.padding[0]
        si_out = ops.conv2d_grouped(si, [self.si_layers.mixed_uniform(t) for t in types], self.si_layers.bias, kh, kw, pad, share_x=True,
                                    co=self.si_layers.out_channels)
        fused = [self._fused_gb(t) for t in types]
        mix = ops.gb_spade(si_out, z_cat, [f[0] for f in fused], fused[0][1], self.zi_layers.eps)     # :2440-2446, one node
        return ops.conv2d_grouped(mix, [self.out.mixed_uniform(t) for t in types], self.out.bias, kh, kw, pad, co=self.out.out_channels)


def _up2(x, norm=None):
    """nn.Upsample(scale_factor=(2,2), mode='bilinear')  (model.py:2551).  norm: the InstanceNorm2d of the SPADE block that reads the
    result -- its statistics are then taken while the result is written (ops.bilinear_up2) instead of in a pass of their own."""
    return ops.bilinear_up2(x, None if norm is None else norm.eps)


class SPADENewShared(nn.Module):
    """model.py:2540-2582.  QUIRK: `up2` is reused for the third upsample (:2573) -- no effect."""

    def __init__(self, image_size=(192, 160), in_num_ch=7, z_size=16, z_num_ch=128, s_num_ch=8, is_cond=False):
        super().__init__()
        self.z_num_ch, self.image_size, self.is_cond = z_num_ch, tuple(image_size), is_cond
        H, W = image_size
        self.zi_scaler = nn.Linear(z_size, H * W * z_num_ch // 1024)
        self.sp1 = SPADEBlockNew((H // 32, W // 32), z_num_ch, z_num_ch, s_num_ch, is_cond)
        self.sp2 = SPADEBlockNew((H // 16, W // 16), z_num_ch, z_num_ch, s_num_ch, is_cond)
        self.sp3 = SPADEBlockNew((H // 8, W // 8), z_num_ch, z_num_ch, s_num_ch, is_cond)

    def forward(self, si, zi, inputs_type=None, scatter=None):
        H, W = self.image_size
        x = ops.to_storage(self.zi_scaler(zi).reshape(-1, self.z_num_ch, H // 32, W // 32))     # opens the decoder's bf16 stretch
        x = self.sp1(si, x, inputs_type)
        x = self.sp2(si, _up2(x, self.sp2.zi_layers), inputs_type)
        x = self.sp3(si, _up2(x, self.sp3.zi_layers), inputs_type)
        if scatter is not None and ops.up2_scatter_applies(x):
            # scatter = (holder, j, M): the batch is M sample blocks for label j; block i goes straight to its place in decoder i's input
            return ops.bilinear_up2_scatter(x, scatter[0], scatter[1], scatter[2], self.sp3.zi_layers.eps)
        return _up2(x, self.sp3.zi_layers)             # read by sp4 of a SPADENewNotShared (same InstanceNorm2d defaults)


class SPADENewNotShared(nn.Module):
    """model.py:2584-2632."""

    def __init__(self, image_size=(192, 160), in_num_ch=7, z_size=16, z_num_ch=128, s_num_ch=8, is_cond=False,
                 output_activation='softplus'):
        super().__init__()
        self.is_cond = is_cond
        H, W = image_size
        self.sp4 = SPADEBlockNew((H // 4, W // 4), z_num_ch, z_num_ch // 2, s_num_ch, is_cond)
        self.sp5 = SPADEBlockNew((H // 2, W // 2), z_num_ch // 2, z_num_ch // 4, s_num_ch, is_cond)
        self.sp6 = SPADEBlockNew((H, W), z_num_ch // 4, z_num_ch // 8, s_num_ch, is_cond)
        self.out = Conv2d(is_cond)(z_num_ch // 8, in_num_ch, 1, 1)
        if output_activation == 'softplus':
            self.out_act = nn.Softplus()
        elif output_activation == 'no':
            self.out_act = nn.Sequential()
        else:
            raise ValueError('No activation in SPADENotShared')

    def forward(self, si, zi_sp4_input, inputs_type=None):
        x = self.sp4(si, zi_sp4_input, inputs_type)
        x = self.sp5(si, _up2(x, self.sp5.zi_layers), inputs_type)
        x = self.sp6(si, _up2(x, self.sp6.zi_layers), inputs_type)
        x = self.out(x, inputs_type) if self.is_cond else self.out(x)
        return self.out_act(x)

    def forward_grouped(self, si, z_cat, types):
        """forward for G modality labels on batch-concatenated inputs (see SPADEBlockNew.forward_grouped)."""
        x = self.sp4.forward_grouped(si, z_cat, types)
        x = self.sp5.forward_grouped(si, _up2(x, self.sp5.zi_layers), types)
        x = self.sp6.forward_grouped(si, _up2(x, self.sp6.zi_layers), types)
        kh, kw = self.out.kernel_size
        x = ops.conv2d_grouped(x, [self.out.mixed_uniform(t) for t in types], self.out.bias, kh, kw, self.out.padding[0], co=self.out.out_channels)
        return self.out_act(x)

    def grouped_ok(self):
        blocks = (self.sp4, self.sp5, self.sp6)
        return (self.is_cond and all(b.si_layers.stride == (1, 1) and b.gamma.kernel_size == b.out.kernel_size == b.si_layers.kernel_size
                                     and b.gamma.weight.shape == b.beta.weight.shape for b in blocks)
                and self.out.stride == (1, 1))


# =============================================================================
# discriminator: model.py:2769-2800
# =============================================================================
class Discriminator(nn.Module):
    """model.py:2769-2800: 5x (conv4x4 s2 [+BN] + LeakyReLU 0.2) + dense or PatchGAN head.
    state_dict keys match the reference's nn.Sequential indices."""

    def __init__(self, in_num_ch=8, inter_num_ch=16, input_shape=(160, 192), is_patch_gan=False):
        super().__init__()
        c = inter_num_ch
        L = [HipConv2d(in_num_ch, c, 4, 2, padding=1), nn.LeakyReLU(0.2)]
        for a, b in ((c, 2 * c), (2 * c, 4 * c), (4 * c, 8 * c), (8 * c, 4 * c)):
            L += [HipConv2d(a, b, 4, 2, padding=1), BatchNorm2d(b), nn.LeakyReLU(0.2)]
        self.discrim = nn.Sequential(*L)
        self.is_patch_gan = is_patch_gan
        if is_patch_gan:
            self.fc = HipConv2d(4 * c, 1, 3, 1, padding=1)
        else:
            self.fc = nn.Sequential(
                nn.Flatten(),
                nn.Linear(int(input_shape[0] * input_shape[1] * 4 * c / (32 * 32)), c * 16),
                nn.LeakyReLU(0.2),
                nn.Linear(c * 16, 1))

    def forward(self, x):
        mods = list(self.discrim)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, HipConv2d) and isinstance(mods[i + 1], nn.LeakyReLU):
                x = m(x, lrelu=True); i += 2                      # conv + fused LeakyReLU
            else:
                x = m(x); i += 1
        if self.is_patch_gan:
            return self.fc(x)
        # nn.Flatten flattens NCHW order: make it physical before the Linear (tiny tensor)
        return self.fc(x.contiguous(memory_format=torch.contiguous_format).float())


# =============================================================================
# orchestration + losses: model.py:2916-2970, 3086-3224, 3260-3587
# =============================================================================
# =============================================================================
# output decoder 'U+SA' (lambda_recon_y > 0): model.py:117-174, 341-390, 1303-1327
# =============================================================================
class LegacyConvBNAct(nn.Module):
    """Conv_BN_Act (model.py:117-139).  QUIRK: the if/if/if-else chain leaves `act` = identity unless 'elu'."""

    def __init__(self, in_num_ch, out_num_ch, activation='lrelu'):
        super().__init__()
        self.conv = nn.Sequential(HipConv2d(in_num_ch, out_num_ch, 4, 2, padding=1), BatchNorm2d(out_num_ch))
        self.act = nn.ELU(inplace=True) if activation == 'elu' else nn.Sequential()

    def forward(self, x):
        return self.act(self.conv(x))


class LegacyUpConcat(nn.Module):
    """Act_Deconv_BN_Concat (model.py:141-174): identity activation (same quirk), x2 bilinear (align_corners=True),
    3x3 conv, BatchNorm, concat with the gated skip; `bn` exists even when is_last (state_dict keys)."""

    def __init__(self, in_num_ch, out_num_ch, is_last=False):
        super().__init__()
        self.act = nn.Sequential()
        self.is_last = is_last
        self.up = nn.Sequential(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True),
                                HipConv2d(in_num_ch, out_num_ch, 3, 1, padding=1))
        self.bn = BatchNorm2d(out_num_ch)

    def forward(self, x_down, x_up):
        x_up = ops.bilinear(self.act(x_up), (2 * x_up.shape[2], 2 * x_up.shape[3]), True)
        x_up = self.up[1](x_up)
        if self.is_last:
            return x_up
        return torch.cat([x_down, self.bn(x_up)], 1)


class SpatialAttentionLayer(nn.Module):
    """model.py:1303-1327 (F.upsample bilinear = align_corners False)."""

    def __init__(self, in_num_ch, gate_num_ch, inter_num_ch, sample_factor=(2, 2)):
        super().__init__()
        self.W_x = HipConv2d(in_num_ch, inter_num_ch, sample_factor, sample_factor, bias=False)
        self.W_g = HipConv2d(gate_num_ch, inter_num_ch, 1, 1)
        self.W_psi = HipConv2d(inter_num_ch, 1, 1, 1)
        self.W_out = nn.Sequential(HipConv2d(in_num_ch, in_num_ch, 1, 1), BatchNorm2d(in_num_ch))

    def forward(self, x, g):
        x_post = self.W_x(x)
        g_post = ops.bilinear(self.W_g(g), tuple(x_post.shape[2:]), False)
        alpha = torch.sigmoid(self.W_psi(F.relu(x_post + g_post)))
        alpha_up = ops.bilinear(alpha, tuple(x.shape[2:]), False)
        return self.W_out(alpha_up * x), alpha_up


class GANShortGeneratorWithSpatialAttention(nn.Module):
    """model.py:341-390, `output_activation='no'` (BraTS / z-score data, main_missing.py:80) -> empty Sequential."""

    def __init__(self, in_num_ch, out_num_ch, first_num_ch=64, output_activation='no'):
        super().__init__()
        if output_activation != 'no':
            raise NotImplementedError("only target_output_act 'no' (main_missing.py:80 for BraTS / z-score) is on the path")
        c = first_num_ch
        self.down_1 = nn.Sequential(HipConv2d(in_num_ch, c, 4, 2, padding=1), nn.LeakyReLU(0.2, inplace=True))
        self.down_2 = LegacyConvBNAct(c, 2 * c)
        self.down_3 = LegacyConvBNAct(2 * c, 4 * c)
        self.down_4 = LegacyConvBNAct(4 * c, 8 * c)
        self.down_5 = LegacyConvBNAct(8 * c, 8 * c, activation='no')
        self.att_4 = SpatialAttentionLayer(8 * c, 8 * c, 8 * c)
        self.up_4 = LegacyUpConcat(8 * c, 8 * c)
        self.att_3 = SpatialAttentionLayer(4 * c, 16 * c, 4 * c)
        self.up_3 = LegacyUpConcat(16 * c, 4 * c)
        self.att_2 = SpatialAttentionLayer(2 * c, 8 * c, 2 * c)
        self.up_2 = LegacyUpConcat(8 * c, 2 * c)
        self.att_1 = SpatialAttentionLayer(c, 4 * c, c)
        self.up_1 = LegacyUpConcat(4 * c, c)
        self.output = LegacyUpConcat(2 * c, out_num_ch, is_last=True)
        self.output_act = nn.Sequential()

    def forward(self, x):
        d1 = self.down_1[0](x, lrelu=True)                 # conv + LeakyReLU(0.2) fused in the epilogue
        d2 = self.down_2(d1); d3 = self.down_3(d2); d4 = self.down_4(d3); d5 = self.down_5(d4)
        c4, a4 = self.att_4(d4, d5); u4 = self.up_4(c4, d5)
        c3, a3 = self.att_3(d3, u4); u3 = self.up_3(c3, u4)
        c2, a2 = self.att_2(d2, u3); u2 = self.up_2(c2, u3)
        c1, a1 = self.att_1(d1, u2); u1 = self.up_1(c1, u2)
        return self.output_act(self.output(None, u1)), {'alpha_4': a4, 'alpha_3': a3, 'alpha_2': a2, 'alpha_1': a1}


class MultimodalModel(nn.Module):
    """model.py:2916-3587 for the configuration the reference ships
    (config.yaml: is_cond, shared_ana_enc, shared_mod_enc, shared_inp_dec=False,
    mod_enc_s=False, softmax_remove_mask, 'max' compaction, cosine similarities).
    Method names and argument order follow the reference."""

    def __init__(self, input_size=(160, 192), modality_num=4, in_num_ch=7, out_num_ch=1, s_num_ch=8, z_size=16,
                 is_discrim_s=False, is_distri_z=False, shared_ana_enc=False, shared_mod_enc=True, shared_inp_dec=True,
                 s_compact_method='max', s_sim_method='cosine', z_sim_method='cosine', is_cond=True,
                 input_output_act='softplus', target_output_act='softplus', target_model_name='U', fuse_method='mean',
                 device=torch.device('cuda:0'), others=None, is_patch_gan=False, build_output_decoder=False):
        super().__init__()
        others = dict(others or {'mod_enc_s': True, 'ana_dec_act': 'softmax'})
        others.setdefault('old', False)
        if others['old'] or is_distri_z or shared_inp_dec or s_compact_method != 'max' or \
                s_sim_method != 'cosine' or z_sim_method != 'cosine' or others.get('mod_enc_s', True) or \
                others.get('ana_dec_act', 'softmax') != 'softmax' or not others.get('softmax_remove_mask', False):
            raise NotImplementedError('only the shipped config.yaml graph is built (SURVEY.md section 8a); '
                                      'got a variant that the hot path does not cover')
        H, W = input_size
        if H % 32 or W % 32:
            raise ValueError(f'input_size {input_size} must be a multiple of 32 (five stride-2 stages, model.py:2192)')
        self.input_size, self.modality_num, self.in_num_ch = tuple(input_size), modality_num, in_num_ch
        self.device, self.others, self.is_cond = device, others, is_cond
        self.shared_ana_enc, self.shared_mod_enc = shared_ana_enc, shared_mod_enc
        n_ana = 1 if shared_ana_enc else modality_num
        self.anatomy_encoder_enc_list = nn.ModuleList(
            [AnatomyEncoderEncNew(in_num_ch, 32, is_cond) for _ in range(n_ana)])                 # :3086-3100
        self.anatomy_encoder_dec = AnatomyEncoderDecNew(32, s_num_ch, is_cond=is_cond)
        n_mod = 1 if shared_mod_enc else modality_num
        self.modality_encoder_list = nn.ModuleList(
            [ModalityEncoderNew(in_num_ch, 0, 16, z_size, is_cond, (H // 32) * (W // 32)) for _ in range(n_mod)])  # :3102-3112
        dec = [SPADENewNotShared((H, W), in_num_ch, z_size, 128, s_num_ch, is_cond, input_output_act)
               for _ in range(modality_num)]
        dec.append(SPADENewShared((H, W), in_num_ch, z_size, 128, s_num_ch, is_cond))            # :3129-3131
        self.input_decoder_list = nn.ModuleList(dec)
        # output_decoder: only when a lambda_recon_y* weight is set (config.yaml:27-28 ships 0); built in the reference's
        # position (after the input decoders, before the discriminator, model.py:2955-2967) so seeds give the same init
        self.fuse_method = fuse_method
        if build_output_decoder:
            if ops.storage_bf16():
                raise NotImplementedError("compute_dtype 'bf16' (bf16 activations) covers the shipped loss set; the 'U+SA' output decoder runs in 'f32' / 'bf16m'")
            if target_model_name != 'U+SA' or fuse_method != 'mean':
                raise NotImplementedError("output decoder: only target_model_name 'U+SA' with fuse_method 'mean' (config.yaml:64, 66)")
            self.output_decoder = GANShortGeneratorWithSpatialAttention(s_num_ch, out_num_ch, 64, target_output_act)
        if is_discrim_s:
            self.discrim_s = Discriminator(s_num_ch, 16, (H, W), is_patch_gan)                    # :2966-2967
        self.to(device)
        self._types = {}
        # modality labels 1..M (model.py:3138) as one table, so CondConv2d can mix all of them at once
        self._type_table = ops.register_type_table(
            torch.arange(1, modality_num + 1, dtype=torch.float32, device=device).view(modality_num, 1))

    # ---- which parameters the step can reach
    _DEAD = ('.convs.',)                               # ModalityEncoderNew.convs: built, never called (model.py:2346-2357)
    _DEAD_LAST_BN = ('anatomy_encoder_dec.output.bn.', 'output_decoder.output.bn.')      # is_last: bn created, not applied (:2189)

    def trainable_parameters(self):
        """The static list of parameters that receive a gradient in a training step: everything that requires grad
        except the reference's dead modules (their state_dict keys and RNG draws exist, their gradients never do).
        Pinned against the reference by the step goldens (same set of tensors with a gradient)."""
        out = []
        for n, p in self.named_parameters():
            if not p.requires_grad:
                continue
            if n.startswith('modality_encoder_list.') and any(d in n for d in self._DEAD):
                continue
            if n.startswith(self._DEAD_LAST_BN):
                continue
            out.append(p)
        return out

    # ---- expert mixing in groups (ops.premix_all) and the order in which their gradients complete
    def mix_groups(self):
        """[(name, [root modules])]: the CondConv2d layers under each entry are mixed by one launch right before the entry's first
        use in a step and their gradients are taken apart by one backward node per entry."""
        M = self.modality_num
        g = [('enc', list(self.anatomy_encoder_enc_list) + [self.anatomy_encoder_dec] + list(self.modality_encoder_list)),
             ('dec_shared', [self.input_decoder_list[-1]])]
        g += [(f'dec{i}', [self.input_decoder_list[i]]) for i in range(M)]
        return g

    def premix(self, name):
        """mix the experts of group `name` for all modality labels (no-op outside a training step / when already done this step)."""
        roots = self.__dict__.get('_mix_roots')
        if roots is None:
            roots = self.__dict__['_mix_roots'] = dict(self.mix_groups())
        return ops.premix_all(self, self._type_table, name, roots[name])

    def completion_groups(self):
        """parameter lists in the order in which a step's backward pass completes their gradients: the modality decoders in reverse
        call order (each is done when its last SPADE block has back-propagated), then the shared decoder, then everything that the
        FIRST encoder pass uses (encoders, discriminator: complete only at the very end).  The optimizer lays the gradient arena out
        in this order and the data-parallel reducer cuts its buckets at the group boundaries, so a bucket can leave as soon as
        its group's backward node has run."""
        M = self.modality_num
        groups = [list(self.input_decoder_list[i].parameters()) for i in reversed(range(M))]
        groups.append(list(self.input_decoder_list[-1].parameters()))
        seen = {id(p) for g in groups for p in g}
        groups.append([p for p in self.parameters() if id(p) not in seen])
        return groups

    def gated_parameter_groups(self):
        """group i = the parameters of input decoder i (SPADENewNotShared): they get a gradient only from batches in which
        modality i is present (every loss term through decoder i is masked by mask[:, i], model.py:3319-3341, 3388);
        torch's Adam leaves such a parameter alone (grad None), and so does the arena step (ArenaAdam.set_gates)."""
        return [list(self.input_decoder_list[i].parameters()) for i in range(self.modality_num)]

    def active_decoders(self, mask_host):
        """(M,) 0/1 numpy vector: which input decoders a batch with this mask sends a gradient to -- the host-side twin of
        the masking in compute_recon_loss_x_list / _x_mix_list / compute_latent_z_loss (model.py:3315-3341, 3384-3394),
        including the QUIRK of the mix loss: its output index only advances on non-empty pairs (:3335-3338), so after a
        skipped pair a term reads the reconstruction of an EARLIER (decoder, type) pair and that decoder gets the gradient."""
        mh = _host_mask(None, mask_host)
        M = self.modality_num
        act = (mh.sum(0) > 0).astype(np.float32)                           # recon_x / latent_z terms of modality i
        pairs = [(i, j) for i in range(M) for j in range(M) if i != j]      # order of reconstruct_input_si_zj's outputs
        idx = 0
        for i, j in pairs:
            if (mh[:, i] * mh[:, j]).sum() == 0:
                continue
            act[pairs[idx][0]] = 1.0
            idx += 1
        return act

    # ---- helpers
    def _type(self, i, B):
        key = (i, B)
        t = self._types.get(key)
        if t is None:
            t = self._types[key] = self._type_table[i:i + 1].expand(B, 1)
        return t

    # ---- model.py:3135-3157
    def _encoders_grouped(self):
        """the per-modality encoder loops (model.py:3135-3157, 3164-3185) as ONE batch-concatenated pass with the kernels mixed per sample
        block: applies inside a training step when the modalities share the encoder modules (config.yaml: shared_ana_enc / shared_mod_enc)"""
        return ops.grouped_applies() and ops.grouped_encoders() and self.is_cond and self.shared_ana_enc and self.shared_mod_enc and self.modality_num > 1

    @staticmethod
    def _cat_inputs(inputs_list):
        """the M input blocks as one batch-concatenated tensor: free when they are the consecutive blocks of the modality-planar copy"""
        x0 = inputs_list[0]
        base = x0._base
        if base is not None and all(t._base is base for t in inputs_list) and base.dim() == 5 and base.is_contiguous() \
                and base.shape[0] == len(inputs_list) and all(t.data_ptr() == base[i].data_ptr() for i, t in enumerate(inputs_list)):
            M, B, H, W, c = base.shape
            return base.view(M * B, H, W, c).permute(0, 3, 1, 2)
        return torch.cat(list(inputs_list), 0)

    def modality_encoder_reads_s(self):
        """whether compute_modality_encoding consumes the anatomy maps (config others.mod_enc_s; model.py:2374: the modality encoder concatenates s only then)"""
        return any(getattr(e, 's_num_ch', 1) != 0 for e in self.modality_encoder_list)

    def compute_anatomy_encoding(self, inputs_list, mask_img, need_maps=True):
        """need_maps = False (the second encoder pass of main_missing.py:228-231 when the modality encoder takes no s): the maps are dead -- nothing reads them,
        no gradient reaches them -- so the pass stops behind the last BatchNorm of the anatomy network (the x2 resize to full resolution, the 64 -> 4
        convolution and the masked softmax of the last block are skipped: 1.5 ms per step at B = 32); the network's state (BatchNorm running statistics,
        counters) is exactly what the full pass leaves.  Returns [None] * M then."""
        if self._encoders_grouped():
            M, B = self.modality_num, inputs_list[0].shape[0]
            types = [self._type(i, B) for i in range(M)]
            x = self._cat_inputs(inputs_list)
            feats = self.anatomy_encoder_enc_list[0].forward_grouped(x, types)
            si, _ = self.anatomy_encoder_dec.forward_grouped(feats, types, need_out=need_maps)
            if not need_maps:
                return [None] * M
            m_all = mask_img if mask_img is None else mask_img.repeat(M, 1, 1)
            return list(ops.split_batch(ops.softmax_mask_drop(si, m_all, 100.0), M))
        si_list = []
        B = inputs_list[0].shape[0]
        for i in range(self.modality_num):
            t = self._type(i, B)
            enc = self.anatomy_encoder_enc_list[0 if self.shared_ana_enc else i]
            feats = enc(inputs_list[i], t)
            si, _ = self.anatomy_encoder_dec(feats, t, need_out=need_maps)
            si_list.append(None if not need_maps else ops.softmax_mask_drop(si, mask_img, 100.0))   # :3150-3153
        return si_list

    # ---- model.py:3159-3162: eps is drawn from the CPU generator, then moved
    def sample(self, z_mean, z_log_var):
        shape = (z_mean.shape[0], z_mean.shape[1])
        eps = ops.host_value(lambda: torch.normal(0, 1, size=shape), self.device)        # a graph replay draws again, in the same order
        return z_mean + eps * torch.exp(0.5 * z_log_var)

    # ---- model.py:3164-3185
    def compute_modality_encoding(self, inputs_list, si_list, phase='train'):
        zi_list, mu_list, lv_list = [], [], []
        B = inputs_list[0].shape[0]
        if self._encoders_grouped() and self.modality_encoder_list[0].s_num_ch == 0:
            M = self.modality_num
            mu, lv = self.modality_encoder_list[0].forward_grouped(self._cat_inputs(inputs_list), [self._type(i, B) for i in range(M)])
            for i in range(M):
                mu_i, lv_i = mu[i * B:(i + 1) * B], lv[i * B:(i + 1) * B]
                zi_list.append(self.sample(mu_i, lv_i) if phase == 'train' else mu_i)       # eps drawn per modality, in order (:3159-3162)
                mu_list.append(mu_i); lv_list.append(lv_i)
            return zi_list, mu_list, lv_list
        for i in range(self.modality_num):
            t = self._type(i, B)
            enc = self.modality_encoder_list[0 if self.shared_mod_enc else i]
            mu, lv = enc(inputs_list[i], si_list[i] if si_list is not None else None, t)
            zi_list.append(self.sample(mu, lv) if phase == 'train' else mu)
            mu_list.append(mu); lv_list.append(lv)
        return zi_list, mu_list, lv_list

    def _shared_mids(self, si_list, zi_list):
        """SPADENewShared (sp1-sp3, model.py:3200 / :3221) for all M x M (s_i, z_j) pairs.  The reference
        runs it once per pair (16 calls at batch B); the pairs that share the modality type j also share
        the mixed kernels, and the block has no cross-sample coupling (InstanceNorm is per sample), so
        they run here as ONE call per type on the batch-concatenated anatomy maps (M calls at batch M*B):
        same arithmetic per sample, 4x larger grids for the 8x8 ... 32x32 layers that cannot fill the chip."""
        key = ('mids', id(si_list[0]), id(zi_list[0]))

        def make():
            M, B = self.modality_num, si_list[0].shape[0]
            self.premix('dec_shared')
            s_cat = torch.cat(list(si_list), 0)
            mids = {}
            holder = {}
            for j in range(M):
                mid = self.input_decoder_list[-1](s_cat, zi_list[j].repeat(M, 1), self._type(j, M * B), scatter=(holder, j, M))
                if isinstance(mid, list):                                 # blocks written in place into the [decoder i][label j] buffer (ops.bilinear_up2_scatter)
                    for i, part in enumerate(mid):
                        mids[(i, j)] = part
                    continue
                st = getattr(mid, '_mrdis_in_stats', None)                # instance statistics taken by the last x2 resize (ops.bilinear_up2)
                for i, part in enumerate(ops.split_batch(mid, M)):        # one-pass adjoint instead of M zero-fills + adds
                    if st is not None:
                        n = st[0].numel() // M
                        part._mrdis_in_stats = (st[0][i * n:(i + 1) * n], st[1][i * n:(i + 1) * n], st[2])
                    mids[(i, j)] = part
            return (list(si_list), list(zi_list), mids)       # inputs kept alive with the cache entry
        return ops.step_cache(key, make)[2]

    def _notshared_all(self, si_list, zi_list):
        """Every (decoder i, label j) reconstruction of the step, or None when the grouped path does not apply.  Decoder i serves
        (s_i, z_j) for all M labels j (model.py:3200-3203, 3219-3224): run as ONE batch-concatenated call per decoder, sample block j
        with the kernels mixed for label j (SPADENewNotShared.forward_grouped) -- the normalisation / resize kernels and the Python
        op dispatches of the four calls collapse into one."""
        decs = [self.input_decoder_list[i] for i in range(self.modality_num)]
        if not (ops.grouped_applies() and all(hasattr(d, 'grouped_ok') and d.grouped_ok() for d in decs)):
            return None
        key = ('nsall', id(si_list[0]), id(zi_list[0]))

        def make():
            M, B = self.modality_num, si_list[0].shape[0]
            mids = self._shared_mids(si_list, zi_list)
            types = [self._type(j, B) for j in range(M)]
            outs = {}
            for i in range(M):
                self.premix(f'dec{i}')
                z_cat = ops.join_blocks([mids[(i, j)] for j in range(M)])
                sts = [getattr(mids[(i, j)], '_mrdis_in_stats', None) for j in range(M)]
                if all(s_ is not None and s_[2] == sts[0][2] for s_ in sts):
                    z_cat._mrdis_in_stats = (torch.cat([s_[0] for s_ in sts]), torch.cat([s_[1] for s_ in sts]), sts[0][2])
                y = self.input_decoder_list[i].forward_grouped(si_list[i], z_cat, types)
                for j, part in enumerate(ops.split_batch(y, M)):
                    outs[(i, j)] = part
            return (list(si_list), list(zi_list), outs)
        return ops.step_cache(key, make)[2]

    # ---- model.py:3187-3203
    def reconstruct_input_si_zi(self, si_list, zi_list):
        out = []
        B = si_list[0].shape[0]
        allo = self._notshared_all(si_list, zi_list)
        if allo is not None:
            return [allo[(i, i)] for i in range(self.modality_num)]
        mids = self._shared_mids(si_list, zi_list)
        for i in range(self.modality_num):
            self.premix(f'dec{i}')
            out.append(self.input_decoder_list[i](si_list[i], mids[(i, i)], self._type(i, B)))
        return out

    # ---- model.py:3205-3224: decoder index i, type j, z_j
    def reconstruct_input_si_zj(self, si_list, zi_list):
        out = []
        B = si_list[0].shape[0]
        allo = self._notshared_all(si_list, zi_list)
        if allo is not None:
            return [allo[(i, j)] for i in range(self.modality_num) for j in range(self.modality_num) if i != j]
        mids = self._shared_mids(si_list, zi_list)
        for i in range(self.modality_num):
            self.premix(f'dec{i}')
            for j in range(self.modality_num):
                if i == j:
                    continue
                out.append(self.input_decoder_list[i](si_list[i], mids[(i, j)], self._type(j, B)))
        return out

    # ---- model.py:3230-3258.  QUIRK: `si_cat[mask == 1]` flattens (batch, modality) into ONE axis, so "fusion" is a
    # mean over a singleton axis: every present (b, m) anatomy map becomes its own sample and the output has sum(mask)
    # rows (b-major).  Only the per-modality caller (mask = ones(B, 1)) gives B rows; main_missing.py:203 with M > 1
    # cannot run in the reference either.  The row selection uses the host mask (no device sync).
    def reconstruct_output_si_fused(self, si_list, mask, mask_host=None):
        mh = _host_mask(mask, mask_host)
        si_cat = torch.stack(si_list, 1)                                  # (B, K, C, H, W)
        rows = np.flatnonzero(mh.reshape(-1) == 1)
        flat = si_cat.reshape((-1,) + tuple(si_cat.shape[2:]))
        if len(rows) != flat.shape[0]:
            flat = flat.index_select(0, ops.to_device(torch.from_numpy(rows), flat.device))
        return self.output_decoder(flat.contiguous(memory_format=torch.channels_last))[0]

    def reconstruct_output_si(self, si_list):
        B = si_list[0].shape[0]
        ones = np.ones((B, 1), dtype=np.float32)
        return [self.reconstruct_output_si_fused([si_list[i]], None, ones) for i in range(self.modality_num)]

    def compute_recon_loss_y_list(self, gt, y_list, mask, p=2, mask_host=None):                 # :3268-3278
        mh = _host_mask(mask, mask_host)
        errs = []
        terms = [i for i in range(len(y_list)) if mh[:, i].sum() != 0]
        for i in terms:
            errs.append(self.compute_recon_loss(gt, y_list[i], p))
        if not errs:
            return torch.zeros((), device=self.device)
        return (torch.stack(errs) * self._weights(lambda m: [m[:, i] / float(m[:, i].sum()) for i in terms], mh)).sum() / len(errs)

    def compute_segmentation_loss_y(self, gt, y, weight=(1., 5., 5., 5.)):                      # :3287-3297
        w = torch.tensor(weight, dtype=torch.float32, device=y.device)
        loss_seg = F.cross_entropy(y, gt.squeeze(1).long(), weight=w)
        y_act = F.softmax(y, dim=1)                                       # F.softmax(y) on a 4-D tensor: dim 1
        dice = 0
        for i in range(1, 4):
            gt_i = (gt[:, 0] == i).float()
            dice = dice + 1 - 2 * torch.sum(y_act[:, i] * gt_i) / (torch.sum(y_act[:, i] ** 2 + gt_i ** 2) + 1e-6)
        return loss_seg + dice / 3

    def compute_segmentation_loss_y_list(self, gt, y_list, mask, mask_host=None):               # :3299-3313
        mh = _host_mask(mask, mask_host)
        loss, idx = torch.zeros((), device=self.device), 0
        for i in range(len(y_list)):
            if mh[:, i].sum() == 0:
                continue
            idx += 1
            loss = loss + self.compute_segmentation_loss_y(gt, y_list[i])
        return loss if idx == 0 else loss / idx

    # ---------------------------------------------------------------- losses
    # The reference branches on `mask[:, i].sum() == 0` on the device (a host sync per branch,
    # SURVEY 0-8).  `mask` arrives from the loader on the host anyway, so callers pass the host
    # copy (`mask_host`, a CPU tensor or ndarray) next to the device copy and the branches are
    # decided without touching the GPU.
    def compute_recon_loss(self, gt, output, p=2):                                               # :3260-3266
        return ops.recon_err(gt, output, p)

    # The loss heads below are the reference's loops with the per-term scalar arithmetic folded into ONE weighted
    # reduction: the weights (mask / count / number of terms) come from the host copy of the mask, so a term costs its
    # HIP reduction and nothing else (the loop form issued ~1000 one-workgroup torch kernels per step, forward and
    # backward).  Term order, the skip rules and the quirks are unchanged.
    def _weights(self, rows_of, mh):
        """rows_of(mh) -> list of numpy (B,) weight vectors -> (T, B) device tensor (one H2D copy).  Handed over as a function of the
        host mask: a step recorded for graph replay (trainer.GraphedTrainStep) evaluates it again on every replay's own mask
        (ops.step_mask_host()); the eager step evaluates it once, on `mh`."""
        def make():
            cur = ops.step_mask_host()                 # None outside a graph replay
            return torch.from_numpy(np.stack(rows_of(mh if cur is None else cur)).astype(np.float32))
        return ops.host_value(make, self.device)

    def compute_recon_loss_x_list(self, gt_list, x_list, mask, p=2, mask_host=None):             # :3315-3325
        mh = _host_mask(mask, mask_host)
        errs = []
        terms = [i for i in range(len(x_list)) if mh[:, i].sum() != 0]
        for i in terms:
            errs.append(self.compute_recon_loss(gt_list[i], x_list[i], p))
        if not errs:
            return torch.zeros((), device=self.device)
        return (torch.stack(errs) * self._weights(lambda m: [m[:, i] / float(m[:, i].sum()) for i in terms], mh)).sum() / len(errs)

    def compute_recon_loss_x_mix_list(self, gt_list, x_list, mask, p=2, mask_host=None):         # :3327-3341
        mh = _host_mask(mask, mask_host)
        errs, pairs = [], []
        M = mh.shape[1]
        idx = 0
        for i in range(M):
            for j in range(M):
                if i == j:
                    continue
                mm_h = mh[:, i] * mh[:, j]
                if mm_h.sum() == 0:
                    continue
                # QUIRK (:3337-3338): x_list is indexed by a counter that only advances on non-empty pairs
                errs.append(self.compute_recon_loss(gt_list[j], x_list[idx], p))
                pairs.append((i, j))
                idx += 1
        if not errs:
            return torch.zeros((), device=self.device)
        return (torch.stack(errs) * self._weights(lambda m: [m[:, i] * m[:, j] / float((m[:, i] * m[:, j]).sum()) for i, j in pairs], mh)).sum() / idx

    def compute_latent_z_loss(self, zi_mean_list, zi_mean_list_new, mask, mask_host=None):       # :3384-3394
        mh = _host_mask(mask, mask_host)
        terms = [i for i in range(len(zi_mean_list)) if mh[:, i].sum() != 0]
        if not terms:
            return torch.zeros((), device=self.device)
        z0 = torch.stack([zi_mean_list[i] for i in terms])                       # (T, B, Z)
        z1 = torch.stack([zi_mean_list_new[i] for i in terms])
        w = self._weights(lambda m: [m[:, i] / float(m[:, i].sum()) for i in terms], mh)      # (T, B)
        return (torch.abs(z0 - z1).sum(2) * w).sum() / len(terms)

    def compute_cosine(self, x, y):                                                              # :3407-3415
        xn = torch.sqrt(torch.sum(x * x, 1) + 1e-8).clamp_min(1e-8)
        yn = torch.sqrt(torch.sum(y * y, 1) + 1e-8).clamp_min(1e-8)
        return torch.sum(x * y, 1) / (xn * yn)

    def compute_compact_s(self, x):                                                              # :3448-3451
        # view(B,-1) of the NCHW-logical pooled map: (C, H/16, W/16) order
        p = ops.max_pool(x, 16)
        return p.contiguous(memory_format=torch.contiguous_format).reshape(x.shape[0], -1)

    def compute_similarity_s_loss(self, si_list, mask, margin=0.1, mask_host=None):              # :3478-3513
        mh = _host_mask(mask, mask_host)
        loss = torch.zeros((), device=self.device)
        if len(si_list) == 1:
            return loss
        M = len(si_list)
        st = {}

        def draw():                                    # host RNG, :3485 (a graph replay draws again, or is handed the caller's draw)
            fp = ops.forced_pair('sim_s')
            if M == 2:
                st['ij'] = (0, 1)
            elif fp is not None:
                st['ij'] = fp
            else:
                sel = np.random.choice(M, 2, replace=False)
                st['ij'] = (int(sel[0]), int(sel[1]))
            return torch.tensor(st['ij'], dtype=torch.long)

        def weights():                                 # mask_i * mask_j * roll(mask_i) / count of the drawn pair, from the host mask
            cur = ops.step_mask_host()
            m = mh if cur is None else cur
            i, j = st['ij']
            mm_h = m[:, i] * m[:, j] * np.concatenate([m[1:, i], m[0:1, i]], 0)
            return torch.from_numpy((mm_h / float(mm_h.sum())).astype(np.float32))
        if ops.recording_host_values():
            # the drawn pair is DATA of the recorded step: pool every map and pick rows i, j with one-hot weights from the host (trainer.GraphedTrainStep only
            # records steps whose mask leaves no (i, j) term empty, so the branch below is the same for every pair).  1 * x + 0 * y + 0 * z is x bit for bit
            # and the unselected maps receive exact zeros, as in the eager step
            def onehots():
                draw()
                oh = np.zeros((2, M), dtype=np.float32)
                oh[0, st['ij'][0]] = 1.0; oh[1, st['ij'][1]] = 1.0
                return torch.from_numpy(oh)
            oh = ops.host_value(onehots, self.device)                              # (2, M)
            C = torch.stack([self.compute_compact_s(s) for s in si_list])         # (M, B, D)
            si_c = (oh[0].reshape(M, 1, 1) * C).sum(0)
            sj_c = (oh[1].reshape(M, 1, 1) * C).sum(0)
        else:
            draw()
            i, j = st['ij']
            mperm_h = np.concatenate([mh[1:, i], mh[0:1, i]], 0)
            if (mh[:, i] * mh[:, j] * mperm_h).sum() <= 0:
                return loss                                                                      # reference returns int 0 (:3512)
            si_c = self.compute_compact_s(si_list[i])
            sj_c = self.compute_compact_s(si_list[j])
        # compact(roll(s_i)) == roll(compact(s_i)): pooling is per-sample
        si_perm_c = torch.cat([si_c[1:], si_c[0:1]], 0)
        sim = self.compute_cosine(si_c, sj_c)
        sim_mix = self.compute_cosine(si_perm_c, si_c)
        return (ops.host_value(weights, self.device) * torch.clamp_min(margin - sim + sim_mix, 0)).sum()

    def compute_similarity_z_loss(self, zi_list, mask, margin=0.1, mask_host=None):              # :3537-3557
        mh = _host_mask(mask, mask_host)
        if len(zi_list) == 1:
            return torch.zeros((), device=self.device)
        I, J = [], []
        for i in range(len(zi_list) - 1):
            mperm_h = np.concatenate([mh[1:, i], mh[0:1, i]], 0)
            for j in range(i + 1, len(zi_list)):
                mm_h = mh[:, i] * mh[:, j] * mperm_h
                if mm_h.sum() == 0:
                    continue
                I.append(i); J.append(j)

        def rows_of(m):
            out = []
            for i, j in zip(I, J):
                mm = m[:, i] * m[:, j] * np.concatenate([m[1:, i], m[0:1, i]], 0)
                out.append(mm / float(mm.sum()))
            return out
        if not I:
            return torch.zeros((), device=self.device)
        Z = torch.stack(list(zi_list))                                            # (M, B, Z)
        Zp = torch.cat([Z[:, 1:], Z[:, 0:1]], 1)                                  # roll by one sample, per modality
        idx = ops.to_device(torch.tensor([I, J], dtype=torch.long), self.device)   # list indexing would be a blocking H2D copy
        zi, zj, zp = Z.index_select(0, idx[0]), Z.index_select(0, idx[1]), Zp.index_select(0, idx[0])   # (P, B, Z)

        def cos(x, y):                                                            # compute_cosine on the last axis
            xn = torch.sqrt(torch.sum(x * x, 2) + 1e-8).clamp_min(1e-8)
            yn = torch.sqrt(torch.sum(y * y, 2) + 1e-8).clamp_min(1e-8)
            return torch.sum(x * y, 2) / (xn * yn)
        hinge = torch.clamp_min(margin - cos(zi, zp) + cos(zi, zj), 0)             # (P, B)
        return (hinge * self._weights(rows_of, mh)).sum() / len(I)

    def compute_adversarial_loss(self, si_list, mask, mask_host=None):                           # :3559-3587
        mh = _host_mask(mask, mask_host)
        M = len(si_list)
        st = {}

        def draw():
            fp = ops.forced_pair('adv_s')
            if M == 2:
                st['ij'] = (0, 1)
            elif fp is not None:
                st['ij'] = fp
            else:
                sel = np.random.choice(M, 2, replace=False)
                st['ij'] = (int(sel[0]), int(sel[1]))
            return torch.tensor(st['ij'], dtype=torch.long)

        def col_weights(k):                            # mask[:, i or j] / its count, from the host mask (zeros for an empty column: the term vanishes)
            def make():
                cur = ops.step_mask_host()
                m = mh if cur is None else cur
                c = m[:, st['ij'][k]]
                n = float(c.sum())
                return torch.from_numpy((c / n if n > 0 else 0.0 * c).astype(np.float32))
            return make
        # (a step recorded for graph replay is recorded PER adv_s pair -- trainer.GraphedTrainStep hands the pair in: selecting the two maps on the
        # device instead would send zero gradients through the anatomy network of the other modalities in the discriminator-loss backward, +5 % per step)
        draw()
        s_i, s_j = si_list[st['ij'][0]], si_list[st['ij'][1]]
        i, j = st['ij']
        d0 = self.discrim_s(s_i).squeeze(1)
        d1 = self.discrim_s(s_j).squeeze(1)
        bce = lambda d, tgt: F.binary_cross_entropy_with_logits(d, tgt, reduction='none')
        zero = torch.zeros((), device=self.device)

        def wsum(w, v):
            w = w.reshape((-1,) + (1,) * (v.dim() - 1)) if v.dim() > 1 else w
            return (w * v).sum()
        if mh[:, i].sum() == 0 and not ops.recording_host_values():
            dl0 = gl0 = zero
        else:
            w0 = ops.host_value(col_weights(0), self.device)
            dl0 = wsum(w0, bce(d0, torch.zeros_like(d0)))
            gl0 = wsum(w0, bce(d0, torch.ones_like(d0)))
        if mh[:, j].sum() == 0 and not ops.recording_host_values():
            dl1 = gl1 = zero
        else:
            w1 = ops.host_value(col_weights(1), self.device)
            dl1 = wsum(w1, bce(d1, torch.ones_like(d1)))
            gl1 = wsum(w1, bce(d1, torch.ones_like(d1)))                                          # QUIRK (sic) :3580
        return 0.5 * (dl0 + dl1), 0.5 * (gl0 + gl1)


def _host_mask(mask, mask_host):
    if mask_host is None:
        mask_host = mask.detach().cpu()        # one sync; pass mask_host to avoid it
    if isinstance(mask_host, torch.Tensor):
        mask_host = mask_host.numpy()
    return np.asarray(mask_host, dtype=np.float32)
