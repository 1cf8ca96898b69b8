"""Autograd surface of the HIP hot path, plus `torch.ops.mrdis.*` registrations.

Each Function is a thin pairing of a forward and a backward entry point of
libmrdis_hip.so (include/mrdis.h); there is no arithmetic here.  Tensors stay
logically NCHW (reference interface) and physically NHWC (channels_last).
"""
import contextlib
import numpy as _np
import os as _os

import torch
from torch.autograd import Function

from . import hip


# --------------------------------------------------------------------------- expert mixing
class _MixExperts(Function):
    """model.py:2111-2113: kernel = sum_e r[e] * W[e]; emits both conv layouts."""

    @staticmethod
    def forward(ctx, W, r):
        w_tck, w_tkc = hip.mix_experts_fwd(W, r)
        ctx.save_for_backward(W, r)
        ctx.mark_non_differentiable(w_tkc)
        return w_tck, w_tkc

    @staticmethod
    def backward(ctx, g_tck, _g_tkc):
        W, r = ctx.saved_tensors
        dW, dr = hip.mix_experts_bwd(g_tck, W, r)
        return dW, dr


def mix_experts(W, r):
    return _MixExperts.apply(W, r)


class _MixExpertsRouted(Function):
    """routing (model.py:2071-2073) + expert mixing (:2113) in one forward and one backward launch pair."""

    @staticmethod
    def forward(ctx, W, fcw, fcb, t_row):
        w_tck, w_tkc, r = hip.mix_experts_routed_fwd(W, fcw, fcb, t_row)
        ctx.save_for_backward(W, r, t_row)
        ctx.emb = fcw.shape[1]
        ctx.mark_non_differentiable(w_tkc)
        return w_tck, w_tkc

    @staticmethod
    def backward(ctx, g_tck, _g_tkc):
        W, r, t_row = ctx.saved_tensors
        dW, dfcw, dfcb = hip.mix_experts_routed_bwd(g_tck, W, r, t_row, ctx.emb)
        return dW, dfcw, dfcb, None


def mix_experts_routed(W, fcw, fcb, t_row):
    return _MixExpertsRouted.apply(W, fcw, fcb, t_row)


class _MixExpertsRoutedAll(Function):
    """_MixExpertsRouted for every modality type of the step in one launch pair: returns
    (w_tck_0, w_tkc_0, ..., w_tck_{M-1}, w_tkc_{M-1}); the backward sums dW / dfc over the types in-kernel."""

    @staticmethod
    def forward(ctx, W, fcw, fcb, types):
        if _COMPUTE_DTYPE != hip.DT_F32:
            # bf16 modes: the mixing launch also writes the bf16 MFMA operands (one launch instead of two casts per filter)
            tck, tkc, r, btck, btkc = hip.mix_experts_routed_multi_fwd(W, fcw, fcb, types, want_bf16=True)
            if _MIX_CACHE is not None:
                for a, ba, bb in zip(tck, btck, btkc):
                    _MIX_CACHE[('bf16w', id(a))] = (a, bb, ba)       # what bf16_filters would make: (w_tck, bf16(w_tkc), bf16(w_tck))
        else:
            tck, tkc, r = hip.mix_experts_routed_multi_fwd(W, fcw, fcb, types)
        ctx.save_for_backward(W, r, types)
        ctx.params = (W, fcw, fcb)                    # the Parameter objects: their .grad may be in-kernel gradient sinks
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*tkc)
        out = []
        for a, b in zip(tck, tkc):
            out += [a, b]
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        W, r, types = ctx.saved_tensors
        sinks = tuple(_grad_sink(q) for q in ctx.params)
        if all(g is not None for g in sinks) and sinks[0].shape == W.shape:
            # the three gradients are added to the optimizer's persistent buffers inside the two launches: no result tensors,
            # no AccumulateGrad add_ per parameter (3 x ~90 modules per step at the host-bound tail of the backward pass)
            hip.mix_experts_routed_multi_bwd(list(grads[0::2]), W, r, types, sinks=sinks)
            return None, None, None, None
        dW, dfcw, dfcb = hip.mix_experts_routed_multi_bwd(list(grads[0::2]), W, r, types)
        return dW, dfcw, dfcb, None


def mix_experts_routed_all(W, fcw, fcb, types):
    return _MixExpertsRoutedAll.apply(W, fcw, fcb, types)


class _MixPairFusedAll(Function):
    """The gamma and beta experts of a SPADE block (model.py:2443-2444), mixed for every modality label of the step straight into ONE
    fused filter per label -- [T][Ci][2C] and [T][2C][Ci], gamma in the first C couts, beta in the second -- by two launches
    (the op-by-op construction was two mixes + three concatenations per label forward and two strided copies per label backward).
    Returns (w_tck_0, w_tkc_0, ..., w_tck_{M-1}, w_tkc_{M-1}); the backward reads each half of the fused gradients in place."""

    @staticmethod
    def forward(ctx, Wg, fwg, fbg, Wb, fwb, fbb, types):
        E, C, Ci, kh, kw = Wg.shape
        T, M = kh * kw, types.shape[0]
        dev_ = Wg.device
        tck = [torch.empty((T, Ci, 2 * C), dtype=torch.float32, device=dev_) for _ in range(M)]
        tkc = [torch.empty((T, 2 * C, Ci), dtype=torch.float32, device=dev_) for _ in range(M)]
        btck = btkc = None
        if _COMPUTE_DTYPE != hip.DT_F32:
            btck = [torch.empty((T, Ci, 2 * C), dtype=torch.bfloat16, device=dev_) for _ in range(M)]
            btkc = [torch.empty((T, 2 * C, Ci), dtype=torch.bfloat16, device=dev_) for _ in range(M)]
        rg = hip.mix_experts_routed_multi_fwd(Wg, fwg, fbg, types, into=(tck, tkc, btck, btkc, 0, 2 * C))
        rb = hip.mix_experts_routed_multi_fwd(Wb, fwb, fbb, types, into=(tck, tkc, btck, btkc, C, 2 * C))
        if btck is not None and _MIX_CACHE is not None:
            for a, ba, bb in zip(tck, btck, btkc):
                _MIX_CACHE[('bf16w', id(a))] = (a, bb, ba)
        ctx.save_for_backward(Wg, rg, Wb, rb, types)
        ctx.params = (Wg, fwg, fbg, Wb, fwb, fbb)
        ctx.C = C
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*tkc)
        out = []
        for a, b in zip(tck, tkc):
            out += [a, b]
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        Wg, rg, Wb, rb, types = ctx.saved_tensors
        C = ctx.C
        g = list(grads[0::2])
        res = []
        for W, r, prm, col0 in ((Wg, rg, ctx.params[0:3], 0), (Wb, rb, ctx.params[3:6], C)):
            sinks = tuple(_grad_sink(q) for q in prm)
            if all(x is not None for x in sinks) and sinks[0].shape == W.shape:
                hip.mix_experts_routed_multi_bwd(g, W, r, types, sinks=sinks, col0=col0, ld=2 * C)
                res += [None, None, None]
            else:
                res += list(hip.mix_experts_routed_multi_bwd(g, W, r, types, col0=col0, ld=2 * C))
        return tuple(res) + (None,)


def mix_pair_fused_all(gamma, beta, types):
    """gamma / beta: CondConv2d modules with equal geometry -> the tuple of _MixPairFusedAll."""
    return _MixPairFusedAll.apply(gamma.weight, gamma._routing_fn.fc.weight, gamma._routing_fn.fc.bias,
                                  beta.weight, beta._routing_fn.fc.weight, beta._routing_fn.fc.bias, types)


# --------------------------------------------------------------------------- all layers of a model mixed in one launch
def native_head(kh, kw, Ci, Co):
    """the 1x1 decoder head (16 -> <= 8, model.py:2605): under bf16 storage it runs on the mixed-storage streaming kernels of
    csrc/mrdis_pointwise.hip (bf16 in, fp32 out: MRDIS_DT_XBF16_YF32) with its filter as it is -- no zero-padding to a 16 -> 16 layer"""
    return kh == 1 and kw == 1 and Ci == 16 and 1 <= Co <= 8


class MixPlan:
    """The static part of the all-layers mixing launches of one model (mrdis_mix_jobs_fwd / _bwd): which CondConv2d layers (`singles`)
    and fused gamma | beta pairs (`pairs`: (block, gamma, beta)) there are, persistent output buffers -- a step's mixed filters are
    dead once its backward has run, so every step writes the same storage and the job table is built once -- the routing values,
    the partial-sum workspace and the job table in device memory.  None if a parameter has no in-kernel gradient sink."""

    def __init__(self, singles, pairs, M, want_bf16, device, pad16=False, group=None):
        self.singles, self.pairs, self.M, self.want_bf16 = singles, pairs, M, want_bf16
        self.group = group
        self.params = []
        self.probe = singles[0].weight if singles else (pairs[0][1].weight if pairs else None)
        shapes = []                                    # per entry: (T, Ci, Cw) with Cw = Co or 2 C
        specs = []                                     # per job: (entry index, conv module, col0, ld)
        self.padded = {}                               # id(module) -> (Ci, Co) true channel counts of a layer whose filters are zero-padded
        for m in singles:
            E, Co, Ci, kh, kw = m.weight.shape
            Ci_p, Co_p = Ci, Co
            if pad16 and m.stride[0] == 1 and (Ci < 16 or Co < 16) and hip.bconv_eligible(max(Ci, 16), max(Co, 16)) and not native_head(kh, kw, Ci, Co):
                # bf16 storage: the narrow side of the 4 -> C si_layers and of the 64 -> 4 / 16 -> 7 heads is zero-padded to 16 so that the
                # bf16 MFMA kernels take the layer (ops.conv2d); the mixing launch writes the filter straight into the padded layout
                Ci_p, Co_p = max(Ci, 16), max(Co, 16)
                self.padded[id(m)] = (Ci, Co)
            shapes.append((kh * kw, Ci_p, Co_p)); specs.append((len(shapes) - 1, m, 0, Co_p))
        for blk, g, b in pairs:
            E, C, Ci, kh, kw = g.weight.shape
            shapes.append((kh * kw, Ci, 2 * C))
            specs.append((len(shapes) - 1, g, 0, 2 * C)); specs.append((len(shapes) - 1, b, C, 2 * C))
        self.shapes = shapes
        sizes = [t * ci * cw for (t, ci, cw) in shapes]
        self.sizes = sizes
        tot = sum(sizes)
        # [entry][label] blocks: tck then tkc
        self.flat = torch.zeros(2 * M * tot, dtype=torch.float32, device=device)         # zero once: the padding of padded filters is never written
        self.flat16 = torch.zeros(2 * M * tot, dtype=torch.bfloat16, device=device) if want_bf16 else None
        offs, o = [], 0
        for sz in sizes:
            offs.append(o); o += 2 * M * sz
        self.offs = offs
        jobs, blocks = [], 0
        self.ok = True
        rbuf = torch.zeros(len(specs) * M * 8, dtype=torch.float32, device=device)
        parts = []
        for k, (ei, m, col0, ld) in enumerate(specs):
            E, Co, Ci, kh, kw = m.weight.shape
            T = kh * kw
            W, fw, fb = m.weight, m._routing_fn.fc.weight, m._routing_fn.fc.bias
            sinks = [_grad_sink(q) for q in (W, fw, fb)]
            if any(q is None for q in sinks) or sinks[0].shape != W.shape or E > 8 or M > 8:
                self.ok = False
                return
            self.params += [W, fw, fb]
            nblk = hip.mix_job_blocks(Co, Ci, T)
            part = torch.empty(8 * nblk * M, dtype=torch.float32, device=device)
            parts.append(part)
            j = hip.MixJob()
            j.W, j.fcw, j.fcb = W.data_ptr(), fw.data_ptr(), fb.data_ptr()
            j.r = rbuf.data_ptr() + 4 * k * M * 8
            sz = sizes[ei]
            for mm in range(M):
                base = offs[ei] + 2 * mm * sz
                j.tck[mm] = self.flat.data_ptr() + 4 * (base + col0)
                j.tkc[mm] = self.flat.data_ptr() + 4 * (base + sz + col0 * Ci)
                if want_bf16:
                    j.btck[mm] = self.flat16.data_ptr() + 2 * (base + col0)
                    j.btkc[mm] = self.flat16.data_ptr() + 2 * (base + sz + col0 * Ci)
            j.dW, j.dfcw, j.dfcb = (q.data_ptr() for q in sinks)
            j.part = part.data_ptr()
            cip = self.shapes[ei][1]                   # channel pitch of this entry's tensors (Ci, or 16 for a padded narrow layer)
            j.tap_tkc = ld * cip
            j.E, j.Co, j.Ci, j.T, j.ld_tck, j.ld_dw = E, Co, Ci, T, ld, ld
            j.block0, j.nblk, j.accumulate, j.ci_pitch = blocks, nblk, 1, cip
            blocks += nblk
            jobs.append(j)
        self.specs, self.njobs, self.blocks = specs, len(jobs), blocks
        self.keep = (rbuf, parts)
        self.jobs, self.device = jobs, device
        self.table = hip.mix_job_table(jobs, device)
        self.tables = {self.probe.grad.data_ptr(): self.table}       # by gradient arena: the d-step re-points every .grad (trainer.py)
        # every parameter whose gradient is complete once this plan's backward has been enqueued: expert weights and routing
        # parameters (written by that launch) and the layers' biases (written by the weight-gradient launches that produced its inputs)
        mods = list(singles) + [m for _, g, b in pairs for m in (g, b)]
        self.ready_params = list(self.params) + [m.bias for m in mods if m.bias is not None]
        self._build_wino_images(device)

    def _build_wino_images(self, device):
        """fp32 path: the Winograd-domain images of the 3x3 stride-1 filters that the software-pipelined kernels read (include/mrdis.h,
        mrdis_wino_u_jobs): per entry and label up to two roles -- 'fwd' (or 'spade' for a fused gamma | beta pair) and 'dgrad' -- built by
        ONE launch per mixing group right behind the mixing launch, into persistent buffers (a job table like the mixing launch's own)."""
        self.u_njobs = self.u_blocks = 0
        self.u_opts = None
        self.u_imgs = {}                                # (entry, label) -> {role: image tensor}
        if self.want_bf16 or not _WINO_U:
            return
        M, flat = self.M, self.flat
        self.u_opts = (hip.get_option('wino4'),)       # what the formats below were chosen under: still_valid() rebuilds the plan when it changes
        want = []                                      # (entry, label, role, src element offset, R, S, flip, spadeC)
        nsing = len(self.singles)
        for ei, (T, Ci, Cw) in enumerate(self.shapes):
            if T != 9 or Ci % 4 != 0 or Cw % 4 != 0:
                continue
            if ei < nsing:
                m = self.singles[ei]
                if m.stride != (1, 1) or m.padding != (1, 1) or id(m) in self.padded:
                    continue
                C = 0
            else:
                C = Cw // 2
                if C < 16 or C % 16 != 0:
                    continue
            sz, o = self.sizes[ei], self.offs[ei]
            for mm in range(M):
                tck_off, tkc_off = o + 2 * mm * sz, o + 2 * mm * sz + sz
                # (<= 32 couts: only the narrow F(4x4) form -- format 5 -- reads an image; the F(2x2) kernel for 32 couts transforms the taps itself)
                if Cw > 32 or (not C and hip.wino_u_format(Ci, Cw) == 5):
                    want.append((ei, mm, 'spade' if C else 'fwd', tck_off, Ci, Cw, 0, C))
                if Ci > 32 or hip.wino_u_format(Cw, Ci) == 5:
                    want.append((ei, mm, 'dgrad', tkc_off, Cw, Ci, 1, 0))
        if not want:
            return
        fmts = [hip.wino_u_format(R, S, C) for (_, _, _, _, R, S, _, C) in want]
        sizes = [hip.wino_u_image_floats(R, S, C, f) for (_, _, _, _, R, S, _, C), f in zip(want, fmts)]
        self.u_flat = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
        jobs, blocks, pos = [], 0, 0
        for (ei, mm, role, src, R, S, flip, C), n, fmt in zip(want, sizes, fmts):
            img = self.u_flat[pos:pos + n]
            img.mrdis_fmt = fmt                        # the format travels with the image (hip.wino_image_fmt -> w_wino_fmt of the C ABI)
            self.u_imgs.setdefault((ei, mm), {})[role] = img
            j = hip.WinoUJob()
            j.w, j.img = flat.data_ptr() + 4 * src, img.data_ptr()
            j.R, j.S, j.flip, j.spadeC, j.fmt = R, S, flip, C, fmt
            j.block0, j.nblk = blocks, hip.wino_u_job_blocks(R, S, C)
            blocks += j.nblk; pos += n
            jobs.append(j)
        self.u_table = hip.wino_u_table(jobs, device)
        self.u_njobs, self.u_blocks = len(jobs), blocks

    def table_for_current_sinks(self):
        """the job table whose gradient sinks are the parameters' CURRENT .grad buffers (TrainStep attaches another optimizer's arena for
        the discriminator-loss backward); built once per arena"""
        g = self.probe.grad
        key = g.data_ptr() if g is not None else None
        tab = self.tables.get(key)
        if tab is None:
            for j, (ei, m, col0, ld) in zip(self.jobs, self.specs):
                sinks = [_grad_sink(q) for q in (m.weight, m._routing_fn.fc.weight, m._routing_fn.fc.bias)]
                if any(q is None for q in sinks):
                    raise RuntimeError('all-layers mixing backward: a parameter lost its gradient buffer between forward and backward')
                j.dW, j.dfcw, j.dfcb = (q.data_ptr() for q in sinks)
            tab = self.tables[key] = hip.mix_job_table(self.jobs, self.device)
        return tab

    def still_valid(self):
        """the expert weights are views of the optimizer's arena: re-check (cheaply) that they have not moved"""
        if getattr(self, 'u_opts', None) is not None and self.u_opts != (hip.get_option('wino4'),):
            return False                               # the image formats were chosen under another value of 'wino4': rebuild (images keep the format they were built in)
        return all(j.W == m.weight.data_ptr() for j, (ei, m, col0, ld) in zip(self.jobs[:4], self.specs[:4])) and self.probe.grad is not None

    def views(self):
        """fresh tensor objects over the persistent buffers: per entry a list [tck_0, tkc_0, ..., tck_{M-1}, tkc_{M-1}] (+ bf16 pairs)"""
        out, out16 = [], []
        M, flat, flat16 = self.M, self.flat, self.flat16
        for ei, (T, Ci, Cw) in enumerate(self.shapes):
            sz, o = self.sizes[ei], self.offs[ei]
            s_tck, s_tkc = (Ci * Cw, Cw, 1), (Cw * Ci, Ci, 1)
            ent = []
            for mm in range(M):
                ent.append(flat.as_strided((T, Ci, Cw), s_tck, o + 2 * mm * sz))
                ent.append(flat.as_strided((T, Cw, Ci), s_tkc, o + 2 * mm * sz + sz))
            out.append(ent)
            if flat16 is not None:
                out16.append([(flat16.as_strided((T, Ci, Cw), s_tck, o + 2 * mm * sz), flat16.as_strided((T, Cw, Ci), s_tkc, o + 2 * mm * sz + sz))
                              for mm in range(M)])
        return out, out16


class _MixAllLayers(Function):
    """Every CondConv2d layer of the model mixed for all modality labels of the step by ONE launch; the backward takes the gradients
    of all mixed filters apart again (dW, dfc.weight, dfc.bias added to the optimizer's buffers in-kernel) by one launch pair.
    Same arithmetic in the same order as _MixExpertsRoutedAll / _MixPairFusedAll per layer."""

    @staticmethod
    def forward(ctx, holder, types, *params):
        plan = holder[0]
        hip.mix_jobs_fwd(plan.table, plan.njobs, plan.blocks, types)
        if plan.u_njobs:
            hip.wino_u_jobs(plan.u_table, plan.u_njobs, plan.u_blocks)      # the Winograd-domain images of this group's 3x3 filters
        ents, ents16 = plan.views()
        holder.append((ents, ents16))
        ctx.plan = plan
        ctx.save_for_backward(types)
        ctx.set_materialize_grads(False)
        flat = []
        for ent in ents:
            flat += ent
        ctx.mark_non_differentiable(*flat[1::2])
        return tuple(flat)

    @staticmethod
    def backward(ctx, *grads):
        plan = ctx.plan
        types, = ctx.saved_tensors
        M = plan.M
        ptrs = torch.zeros(plan.njobs, 8, dtype=torch.int64)
        keep = []
        acc = ptrs.numpy()
        for k, (ei, m, col0, ld) in enumerate(plan.specs):
            for mm in range(M):
                g = grads[2 * (ei * M + mm)]
                if g is None:
                    continue
                if not g.is_contiguous():
                    g = g.contiguous(); keep.append(g)
                acc[k, mm] = g.data_ptr() + 4 * col0
        dev_tab = to_device(ptrs, types.device)
        hip.mix_jobs_bwd(plan.table_for_current_sinks(), plan.njobs, plan.blocks, dev_tab, types)
        for fn in tuple(_GROUP_READY.values()):
            fn(plan.ready_params)                      # in-kernel sinks fire no autograd hook: the exchange of this group's buckets starts here
        return (None, None) + (None,) * (len(plan.params))


_PREMIX = _os.environ.get('MRDIS_PREMIX', '1') != '0'
_WINO_U = _os.environ.get('MRDIS_WINO_IMAGES', '1') != '0'      # build the Winograd filter images with the mixing launch (fp32 path)


def wino_images(w_tck):
    """{role: image} of a mixed filter whose Winograd-domain images the step's mixing launch built ('fwd' | 'spade', 'dgrad'), else {}"""
    if _MIX_CACHE is None or _COMPUTE_DTYPE != hip.DT_F32:
        return {}
    hit = _MIX_CACHE.get(('winoU', id(w_tck)))
    return hit[1] if hit is not None else {}


_GROUP_READY = {}           # key -> fn: one entry per armed trainer.GradAllReduce (two reducers in a process do not overwrite each other)


def set_group_ready_hook(fn, key=None):
    """fn(list of parameters) is called from the backward of each all-layers mixing node, right after its launches: the gradients of
    those parameters (in-kernel sinks, which fire no post-accumulate hook) are complete in stream order.  trainer.GradAllReduce
    starts the all-reduce of the buckets they fill from here.  `key` names the subscriber; fn = None removes its hook."""
    if fn is None:
        _GROUP_READY.pop(key, None)
    else:
        _GROUP_READY[key] = fn


def premix_all(model, table, group=None, roots=None):
    """Fill the step's mixed-filter cache for every CondConv2d layer under `roots` (default: the whole model) with one launch
    (see MixPlan); no-op outside a training step, without gradient sinks, or when disabled (MRDIS_PREMIX=0).  The modules find
    their filters under the same cache keys their own lazy mixing would have used.
    `group` names the plan: the model mixes its layers in GROUPS (encoders | shared decoder | one per modality decoder,
    MultimodalModel.mix_groups), each right before its first use, so that each group's backward node sits in the autograd graph
    behind the group's own layers and runs as soon as their filter gradients exist -- not at the very end of the backward pass:
    the data-parallel exchange of a group's gradients then overlaps the backward of everything that was computed before it."""
    if not _PREMIX or _MIX_CACHE is None or not torch.is_grad_enabled() or not table.is_cuda:
        return False
    if _MIX_CACHE.get(('premixed', id(model), group)):
        return True
    want16 = _COMPUTE_DTYPE != hip.DT_F32
    key = (table.shape[0], _COMPUTE_DTYPE, table.device, group)
    plans = model.__dict__.setdefault('_mrdis_mix_plans', {})
    plan = plans.get(key)
    if plan is not None and not plan.ok and plan.probe is not None and _grad_sink(plan.probe) is not None:
        plan = None                                   # built before the optimizer gave the parameters their gradient buffers
    if plan is None or (plan.ok and not plan.still_valid()):
        singles, pairs, fused_ids, seen = [], [], set(), set()
        from . import model as _model
        roots_ = [model] if roots is None else list(roots)
        for root in roots_:
            for blk in root.modules():
                if isinstance(blk, _model.SPADEBlockNew) and blk.is_cond and blk.gamma.weight.shape == blk.beta.weight.shape and id(blk) not in seen:
                    seen.add(id(blk))
                    pairs.append((blk, blk.gamma, blk.beta)); fused_ids.update((id(blk.gamma), id(blk.beta)))
        for root in roots_:
            for m in root.modules():
                if isinstance(m, _model.CondConv2d) and id(m) not in fused_ids and id(m) not in seen:
                    seen.add(id(m))
                    singles.append(m)
        if not singles and not pairs:
            return False
        plan = plans[key] = MixPlan(singles, pairs, table.shape[0], want16, table.device, pad16=_COMPUTE_DTYPE == hip.DT_BF16, group=group)
    if not plan.ok:
        return False
    holder = [plan]
    outs = _MixAllLayers.apply(holder, table, *plan.params)
    ents, ents16 = holder[1]
    M = plan.M
    tkey = (table.data_ptr(), table._version)
    pos = 0
    for ei in range(len(plan.shapes)):
        allw = outs[pos:pos + 2 * M]; pos += 2 * M
        if ei < len(plan.singles):
            _MIX_CACHE[(id(plan.singles[ei]), 'all') + tkey] = allw
        else:
            _MIX_CACHE[(id(plan.pairs[ei - len(plan.singles)][0]), 'gb_all') + tkey] = allw
        if ents16:
            for mm in range(M):
                a = allw[2 * mm]
                btck, btkc = ents16[ei][mm]
                _MIX_CACHE[('bf16w', id(a))] = (a, btkc, btck)
        if plan.u_njobs:
            for mm in range(M):
                imgs = plan.u_imgs.get((ei, mm))
                if imgs:
                    _MIX_CACHE[('winoU', id(allw[2 * mm]))] = (allw[2 * mm], imgs)
    _MIX_CACHE[('premixed', id(model), group)] = True
    return True


# registry of the per-model type tables: storage pointer -> (M, emb) tensor whose rows are the modality labels
_TYPE_TABLES = {}


def register_type_table(table):
    _TYPE_TABLES[table.untyped_storage().data_ptr()] = table
    return table


def lookup_type_row(inputs_type):
    """(table, row index) if `inputs_type` is a stride-0 expand of one row of a registered table, else None."""
    table = _TYPE_TABLES.get(inputs_type.untyped_storage().data_ptr())
    if table is None or inputs_type.dim() != 2 or inputs_type.shape[1] != table.shape[1]:
        return None
    emb = table.shape[1]
    off = inputs_type.storage_offset()
    if off % emb != 0 or off // emb >= table.shape[0] or (emb > 1 and inputs_type.stride(1) != 1):
        return None
    return table, off // emb


# --------------------------------------------------------------------------- batch split with a one-pass adjoint
class _SplitBatch(Function):
    """x (parts*B, ...) -> parts views of B samples.  autograd's own slicing adjoint materialises one full-size zero
    tensor per slice, copies the slice gradient in and then adds the full-size tensors pairwise (for the 268 MB output
    of the shared SPADE blocks: 4 fills + 4 copies + 3 adds per modality type); here the adjoint is ONE concatenation."""

    @staticmethod
    def forward(ctx, x, parts):
        ctx.parts = parts
        ctx.shape = tuple(x.shape)
        B = x.shape[0] // parts
        return tuple(x[k * B:(k + 1) * B] for k in range(parts))

    @staticmethod
    def backward(ctx, *grads):
        B = ctx.shape[0] // ctx.parts
        ref = next(g for g in grads if g is not None)
        gs = [g if g is not None else torch.zeros_like(ref) for g in grads]
        return torch.cat(gs, 0), None


def split_batch(x, parts):
    if x.shape[0] % parts != 0:
        raise ValueError('batch not divisible')
    return _SplitBatch.apply(x, parts)


# --------------------------------------------------------------------------- parameter-gradient sinks
# A conv bias / BatchNorm affine parameter of a module that runs 8-16 times per step receives that many
# gradients; autograd adds them with one tiny kernel each (~1000 launches of ~5 us per step).  Once the
# optimizer owns a persistent gradient buffer for the parameter (`p.grad` is a view of the arena, zeroed
# per step), the backward kernels add their contribution to it directly and hand autograd `None`.
# Parameters opt in with `p._mrdis_sink = True` (set by the modules of model.py).
_GRAD_SINK = _os.environ.get('MRDIS_GRAD_SINK', '1') != '0'


def _grad_sink(p):
    if not _GRAD_SINK or p is None or not getattr(p, '_mrdis_sink', False):
        return None
    g = p.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32:
        return None
    return g


def set_grad_sink(enabled):
    global _GRAD_SINK
    _GRAD_SINK = bool(enabled)


# --------------------------------------------------------------------------- convolution
# Compute dtype (config key `compute_dtype`, BASELINE.json configs[2]):
#   'f32'   exact fp32 MFMA, fp32 activations -- the parity path (MRDIS_DT_F32);
#   'bf16m' bf16 MFMA operands + fp32 accumulate on fp32 activations (MRDIS_DT_F32_BF16M);
#   'bf16'  bf16 activations in HBM + bf16 MFMA operands + fp32 accumulate (MRDIS_DT_BF16): every tensor with >= 16 channels is
#           stored in bf16; master weights, biases, all statistics, the 4-channel anatomy maps, reconstructions, losses, the
#           optimizer and every parameter gradient stay fp32.
_COMPUTE_DTYPE = hip.DT_F32


def set_compute_dtype(name):
    global _COMPUTE_DTYPE
    table = {'f32': hip.DT_F32, 'fp32': hip.DT_F32, 'float32': hip.DT_F32, 'bf16m': hip.DT_F32_BF16M,
             'bf16': hip.DT_BF16, 'bfloat16': hip.DT_BF16}
    if name not in table:
        raise ValueError(f"compute_dtype must be 'f32', 'bf16m' or 'bf16', got {name!r}")
    _COMPUTE_DTYPE = table[name]


def compute_dtype():
    return _COMPUTE_DTYPE


def storage_bf16():
    return _COMPUTE_DTYPE == hip.DT_BF16


class _CastView(Function):
    """copy of an NHWC view in another storage type and / or channel count (mrdis_cast_view: first min(C, channels) channels, zero
    padding beyond); the adjoint casts / slices / pads the gradient back."""

    @staticmethod
    def forward(ctx, x, dtype, channels):
        ctx.src = (x.dtype, x.shape[1])
        return hip.cast_view(x, dtype, channels)

    @staticmethod
    def backward(ctx, g):
        return hip.cast_view(g, ctx.src[0], ctx.src[1]), None, None


def cast_view(x, dtype, channels=None):
    channels = x.shape[1] if channels is None else int(channels)
    return x if (x.dtype == dtype and channels == x.shape[1]) else _CastView.apply(x, dtype, channels)


def to_storage(x):
    """an activation tensor in the storage type of the current compute dtype (bf16 for 'bf16', unchanged otherwise)."""
    return cast_view(x, torch.bfloat16) if (storage_bf16() and x.dtype == torch.float32) else x


def bf16_filters(w_tck, w_tkc):
    """bf16 copies of a mixed filter with the reduction axis contiguous: (forward = cast(w_tkc), data gradient = cast(w_tck));
    memoised for the step next to the mixed kernels themselves (the multi-label mixing launch files its own bf16 outputs under
    the same key, so filters that come straight from it are never cast)."""
    def make():
        with torch.no_grad():
            return (w_tck, hip.cast_bf16(w_tkc.detach()), hip.cast_bf16(w_tck.detach()))
    hit = cached_mix(('bf16w', id(w_tck)), make)
    return hit[1], hit[2]


def _pad_bias16(bias, Co_p, detach=False):
    """bias zero-padded to Co_p entries, once per step and bias (autograd-connected unless `detach`)"""
    import torch.nn.functional as F
    return cached_mix(('padb', id(bias), Co_p, detach), lambda: (bias, F.pad(bias.detach() if detach else bias, (0, Co_p - bias.shape[0]))))[1]


def conv2d(x, w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu=False, co=None):
    """-> torch.ops.mrdis.conv2d (registered at the bottom of this file: CUDA kernel, fake kernel, autograd formula).
    With bf16 storage the bf16 kernels run bf16 -> bf16 where the geometry allows (reduction axis % 16, >= 16 outputs);
    other layers (Cin = 4 / 7 first layers, heads with < 16 outputs) run the fp32 kernels between explicit view casts:
    fp32 in -> bf16 out for the layers that open a bf16 stretch, bf16 in -> fp32 out for the heads."""
    if _COMPUTE_DTYPE == hip.DT_F32:
        im = wino_images(w_tck) if (kh == 3 and stride == 1) else {}
        if im:      # the auxiliary-filter slots of the op carry the Winograd images on the fp32 path (bf16 copies in the bf16 modes)
            return torch.ops.mrdis.conv2d(x, w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu, im.get('fwd'), im.get('dgrad'))
        return torch.ops.mrdis.conv2d(x, w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu)
    Ci, Co = w_tck.shape[1], w_tck.shape[2]
    wb_fwd, wb_bwd = bf16_filters(w_tck, w_tkc)
    if _COMPUTE_DTYPE == hip.DT_F32_BF16M:
        return torch.ops.mrdis.conv2d(x, w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu, wb_fwd, wb_bwd)
    if native_head(kh, kw, Ci, Co) and stride == 1 and pad == 0 and x.shape[1] == 16 and type(x) is torch.Tensor:
        return conv2d_grouped(cast_view(x, torch.bfloat16), [(w_tck, w_tkc)], bias, kh, kw, pad, lrelu)      # bf16 in, fp32 out
    if (kh, kw, stride, pad) == (3, 3, 1, 1) and Ci in (4, 16) and Co >= 16 and x.shape[1] == 4 and x.dtype == torch.float32 and not lrelu and type(x) is torch.Tensor:
        # the 4 -> C si_layers (filter as it is, or in the 16-row layout of the mixing launch): the route of the grouped decoders
        # (forward and weight gradient on the Cin = 4 kernels: fp32 map, bf16 output / output gradient; data gradient on the zero-padded bf16 MFMA kernel)
        return conv2d_grouped(x, [(w_tck, w_tkc)], bias, kh, kw, pad, co=co)
    if (kh, kw, stride, pad) == (3, 3, 1, 1) and (co or Co) == 4 and Ci % 16 == 0 and x.shape[1] == Ci and not lrelu and type(x) is torch.Tensor:
        # the C -> 4 layer (ana_dec.output; filter as it is or in the 16-column layout of the mixing launch): the grouped node pads it like the
        # branches below and sends the data gradient through the Cin = 4 kernel (fp32 dy, reversed taps, bf16 out: MRDIS_DT_XBF16_YF32)
        return conv2d_grouped(x, [(w_tck, w_tkc)], bias, kh, kw, pad, co=4)
    if hip.bconv_eligible(Ci, Co):
        # (a filter that the all-layers mixing launch wrote zero-padded to 16 channels -- MixPlan.padded -- meets a narrower input / bias /
        #  true output width `co` here: the view cast pads the input, the bias is padded once per step, the output is sliced back to fp32)
        if bias is not None and bias.shape[0] < Co:
            bias = _pad_bias16(bias, Co)
        y = torch.ops.mrdis.conv2d(cast_view(x, torch.bfloat16, Ci), w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu, wb_fwd, wb_bwd)
        return y if (co is None or co == Co) else cast_view(y, torch.float32, co)
    # Narrow side (the 4-channel anatomy maps into the SPADE `si_layers`, the 64 -> 4 and 16 -> 7 heads): pad the narrow channel
    # count to 16 with zeros -- the view cast writes the zero channels, the filter gets zero rows / columns -- and the bf16 MFMA
    # kernels take the layer in all three directions (the matrix pipe has 16x the fp32 rate: the padding is free, the layer is
    # HBM-bound); a padded output is sliced back to fp32.  Stride-1 layers only: the bf16 weight-gradient kernel has no stride 2.
    Ci_p, Co_p = max(Ci, 16), max(Co, 16)
    if stride == 1 and Ci_p % 16 == 0 and hip.bconv_eligible(Ci_p, Co_p):
        def padded():
            import torch.nn.functional as F
            return (w_tck, F.pad(w_tck, (0, Co_p - Co, 0, Ci_p - Ci)), F.pad(w_tkc.detach(), (0, Ci_p - Ci, 0, Co_p - Co)),
                    None if bias is None else F.pad(bias, (0, Co_p - Co)))
        _, wp_tck, wp_tkc, bias_p = cached_mix(('pad16', id(w_tck), None if bias is None else id(bias)), padded)
        wpb_fwd, wpb_bwd = bf16_filters(wp_tck, wp_tkc)
        y = torch.ops.mrdis.conv2d(cast_view(x, torch.bfloat16, Ci_p), wp_tck, wp_tkc, bias_p, kh, kw, stride, pad, lrelu, wpb_fwd, wpb_bwd)
        return y if Co_p == Co else cast_view(y, torch.float32, Co)
    y = torch.ops.mrdis.conv2d(cast_view(x, torch.float32), w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu)
    return cast_view(y, torch.bfloat16) if Co >= 16 else y


# --------------------------------------------------------------------------- grouped convolution (one autograd node, G filter sets)
# The M modality types of a not-shared decoder (model.py:3187-3224) share the layer geometry and differ in the mixed filters only.  Run
# batch-concatenated (G * B samples, sample block g uses filter set g) every normalisation / resize / activation kernel of the
# decoder is ONE launch instead of G, and the G convolutions of a layer are one Python op (G calls of the C entry on batch slices):
# a quarter of the op dispatches, autograd nodes and allocations of the per-type form.  Same kernels per sample block, same results.
_NO_CO4B_WGRAD = _os.environ.get('MRDIS_CO4B_WGRAD', '1') == '0'      # 0: the C -> 4 layer's weight gradient on the padded bf16 kernel (A/B of wgrad_c4b_kernel<.., SWAP>)
_GROUPED = _os.environ.get('MRDIS_GROUPED', '1') != '0'


def set_grouped(enabled):
    global _GROUPED
    _GROUPED = bool(enabled)


# Default OFF by measurement (round 3, same-box A/B at B = 32 / 256x256): with the encoders batch-concatenated over the four modality labels the
# step takes 167.3 ms instead of 162.3 (bf16: 97.6 vs 93.2) although it issues ~450 launches fewer -- a modality's activations (134 MB at the
# second level) stay in the 256 MiB Infinity Cache between a layer and its consumer, the 4x tensors of the batch-concatenated form do not; and
# the G convolutions of a layer are still G launches, so the small maps fill the chip no better.  Kept as an option (bit-identical losses, tested).
_GROUPED_ENC = _os.environ.get('MRDIS_GROUPED_ENC', '0') != '0'


def set_grouped_encoders(enabled):
    global _GROUPED_ENC
    _GROUPED_ENC = bool(enabled)


def grouped_encoders():
    """the anatomy / modality encoders run batch-concatenated over the modality labels (MultimodalModel._encoders_grouped)"""
    return _GROUPED_ENC


def grouped_applies():
    """inside a training step (mixed-kernel cache open, autograd on), every compute dtype."""
    return _GROUPED and _MIX_CACHE is not None and torch.is_grad_enabled()


def _pad16_filters(w_tck, w_tkc, bias, Ci_p, Co_p):
    """zero-padded copies of a narrow filter (+ bias) for the bf16 MFMA kernels and their bf16 layouts, memoised for the step:
    -> (wp_tck, wp_tkc, bias_p, bf16 forward operand, bf16 data-gradient operand).  Detached: the grouped Function slices the
    padded weight gradient back itself."""
    import torch.nn.functional as F
    Ci, Co = w_tck.shape[1], w_tck.shape[2]

    def make():
        with torch.no_grad():
            wp_tck = F.pad(w_tck.detach(), (0, Co_p - Co, 0, Ci_p - Ci))
            wp_tkc = F.pad(w_tkc.detach(), (0, Ci_p - Ci, 0, Co_p - Co))
            bp = None if bias is None else F.pad(bias.detach(), (0, Co_p - Co))
            return (w_tck, wp_tck, wp_tkc, bp, hip.cast_bf16(wp_tkc), hip.cast_bf16(wp_tck))
    return cached_mix(('pad16g', id(w_tck), None if bias is None else id(bias)), make)[1:]


class _GroupedConvFn(Function):
    """G convolutions with the same geometry on the G sample blocks of a batch-concatenated tensor (or on one shared input).  In the
    bf16 storage mode it applies the policy of ops.conv2d per group: bf16 kernels on bf16 views, a narrow channel side (the 4-channel
    anatomy maps in, the 7-channel reconstruction out) zero-padded to 16."""

    @staticmethod
    def forward(ctx, x, bias, G, share_x, kh, kw, pad, lrelu, co, stride, *filt):
        bm = _COMPUTE_DTYPE != hip.DT_F32
        st = _COMPUTE_DTYPE == hip.DT_BF16
        B = x.shape[0] if share_x else x.shape[0] // G
        H, W = x.shape[2], x.shape[3]
        Ho, Wo = hip.conv_out_hw(H, W, kh, kw, stride, pad)
        Cif, Cof = filt[0].shape[1], filt[0].shape[2]           # the filters' channel counts: already zero-padded to 16 when they come from
        Ci, Co = x.shape[1], (int(co) if co else Cof)           # the all-layers mixing launch (MixPlan.padded); Ci, Co: the layer's own
        head = st and native_head(kh, kw, Cif, Cof) and pad == 0 and Ci == 16        # bf16 in, fp32 out on the streaming 1x1 kernels
        Ci_p, Co_p = (max(Cif, 16), max(Cof, 16)) if (st and not head) else (Cif, Cof)
        padded = (Ci_p, Co_p) != (Cif, Cof)                     # the filters still need padding here
        # the 3x3 4 -> C si_layers with the filter in the 16-row layout of the mixing launch: the Cin = 4 kernel reads the fp32 anatomy
        # map and writes bf16, the weight gradient's reads the fp32 map and the bf16 gradient (MRDIS_DT_XF32_YBF16); maps narrower than 64 make the zero-padded bf16 copy
        si4 = st and x.dtype == torch.float32 and Ci == 4 and Ci_p == 16 and (kh, kw, pad) == (3, 3, 1) and Co_p == Cof and not lrelu
        xin = x
        if st and (x.dtype != torch.bfloat16 or Ci_p != Ci):
            # fp32 -> bf16 view cast, zero channels up to 16; the si_layers make it only when a kernel asks for it
            xin = None if (si4 and stride == 1) else hip.cast_view(x, torch.bfloat16, Ci_p)
        y = hip.empty_nhwc(G * B, Co_p, Ho, Wo, x.device, torch.float32 if head else (torch.bfloat16 if st else x.dtype))
        # the C -> 4 layer (ana_dec.output) under bf16 storage: the window-free 4-cout kernel reads the bf16 map and writes the fp32 result itself
        # (MRDIS_DT_XBF16_YF32) -- no 16-channel bf16 intermediate, no slicing cast
        y4 = None
        if st and Co == 4 and Co_p == 16 and not head and (kh, kw, pad, stride) == (3, 3, 1, 1) and not lrelu and xin is not None and xin.dtype == torch.bfloat16:
            y4 = hip.empty_nhwc(G * B, 4, Ho, Wo, x.device, torch.float32)
        use_tkc, wbs, filt_p = [], [], []
        for g in range(G):
            tck, tkc = filt[2 * g], filt[2 * g + 1]
            bg = bias
            if head:
                use_tkc.append(tkc); wbs.append(None)
                hip.conv2d_fwd(xin if share_x else xin[g * B:(g + 1) * B], tck, bg, kh, kw, stride, pad, lrelu, out=y[g * B:(g + 1) * B])
                continue
            if padded:
                tck, tkc, bg, wb_f, wb_b = _pad16_filters(tck, tkc, bias, Ci_p, Co_p)
            else:
                if bm:
                    wb_f, wb_b = bf16_filters(tck, tkc)
                else:
                    im = wino_images(tck) if kh == 3 else {}
                    wb_f, wb_b = im.get('fwd'), im.get('dgrad')          # fp32: the Winograd-domain images (hip.conv2d_* tell them by dtype)
                if bias is not None and bias.shape[0] < Co_p:
                    bg = _pad_bias16(bias, Co_p, detach=True)
            use_tkc.append(tkc); wbs.append(wb_b)
            filt_p.append((tck, bg, wb_f))
            if si4 and stride == 1 and hip.conv2d_fwd(x if share_x else x[g * B:(g + 1) * B], tck, bg, kh, kw, 1, pad, out=y[g * B:(g + 1) * B], may_decline=True) is not None:
                continue
            if xin is None:
                xin = hip.cast_view(x, torch.bfloat16, Ci_p)
            if y4 is not None:
                if hip.conv2d_fwd(xin if share_x else xin[g * B:(g + 1) * B], tck, bg, 3, 3, 1, 1, out=y4[g * B:(g + 1) * B], may_decline=True) is not None:
                    continue
                y4 = None                                     # outside the kernel's shapes: the padded bf16 kernel for every group
                for g2 in range(g):
                    hip.conv2d_fwd(xin if share_x else xin[g2 * B:(g2 + 1) * B], filt_p[g2][0], filt_p[g2][1], kh, kw, stride, pad, lrelu, out=y[g2 * B:(g2 + 1) * B], w_bf16=filt_p[g2][2])
            hip.conv2d_fwd(xin if share_x else xin[g * B:(g + 1) * B], tck, bg, kh, kw, stride, pad, lrelu, out=y[g * B:(g + 1) * B], w_bf16=wb_f)
        ctx.meta = (G, B, share_x, kh, kw, pad, lrelu, hip.DT_F32_BF16M if bm else hip.DT_F32, Ci, Co, Ci_p, Co_p, x.dtype, padded, Cif, Cof)
        ctx.stride = stride
        ctx.head = head
        ctx.dx_in_gb = bool(getattr(x, '_mrdis_want_dgb', False)) and _GB_INPLACE
        ctx.wbs = wbs
        ctx.bias_param = bias
        ctx.x32 = x if (si4 and stride == 1) else None          # the weight gradient's Cin = 4 kernel reads the fp32 map (MRDIS_DT_XF32_YBF16)
        ctx.xshape = (x.shape[2], x.shape[3])
        ctx.ydtype = y.dtype
        ctx.save_for_backward(xin, y if lrelu else None, *use_tkc)
        if y4 is not None:
            return y4
        return y if Co_p == Co else hip.cast_view(y, torch.float32, Co)       # a padded head leaves as fp32 (the losses read it)

    @staticmethod
    def backward(ctx, dy):
        G, B, share_x, kh, kw, pad, lrelu, dt, Ci, Co, Ci_p, Co_p, x_dtype, padded, Cif, Cof = ctx.meta
        xin, y = ctx.saved_tensors[0], ctx.saved_tensors[1]
        tkcs = ctx.saved_tensors[2:]
        bias = ctx.bias_param
        dy4 = None
        if ctx.head:
            dy = hip.cast_view(dy, torch.float32)                 # fp32 reconstruction gradient in, bf16 trunk gradient out
        elif Co_p != Co:
            if Co == 4 and dy.dtype == torch.float32 and (kh, kw, pad, ctx.stride) == (3, 3, 1, 1) and not lrelu and tkcs[0].shape[1] == 16:
                dy4 = dy                                          # the data and weight gradients' four-channel kernels read the fp32 gradient itself (below)
            # (the zero-padded bf16 copy is made only if one of them declines: dy_pad())
            dy = hip.cast_view(dy, torch.bfloat16, Co_p) if dy4 is None else None
        elif dy.dtype != ctx.ydtype:
            dy = hip.cast_view(dy, ctx.ydtype)
        if lrelu:
            dy = hip.lrelu_bwd(dy, y, 0.2)
        dy_dev = (dy if dy is not None else dy4).device
        pad_box = [dy]

        def dy_pad():
            if pad_box[0] is None:
                pad_box[0] = hip.cast_view(dy4, torch.bfloat16, Co_p)
            return pad_box[0]
        H, W = ctx.xshape
        x32 = ctx.x32
        xdt = torch.bfloat16 if _COMPUTE_DTYPE == hip.DT_BF16 else x_dtype          # the (padded) input view's storage type
        if xin is not None:
            xdt = xin.dtype
        need_x = ctx.needs_input_grad[0]
        dxb = None
        dx4 = None
        if need_x and x32 is not None and dy is not None and dy.dtype == torch.bfloat16 and tkcs[0].shape[2] == 16:
            # si_layers: dx of the 4-channel fp32 map straight from the window-free 4-cout kernel on the bf16 gradient (MRDIS_DT_XF32_YBF16)
            dx4 = hip.empty_nhwc(G * B, 4, H, W, dy_dev, torch.float32)
            for g in range(G):
                if hip.conv2d_bwd_data(dy[g * B:(g + 1) * B], tkcs[g], (H, W), 3, 3, 1, 1, out=dx4[g * B:(g + 1) * B], may_decline=True) is None:
                    dx4 = None
                    break
        if need_x and dx4 is None:
            if ctx.dx_in_gb and not share_x and Ci_p == Ci and xin is not None and xin.dtype == x_dtype:
                # the input is the modulated map of a fused SPADE node (ops._GbSpadeFn): its gradient is also the beta half of that node's
                # [dgamma | dbeta] buffer -- write it there, the node then fills in the other half (hip.gb_slot)
                gbuf = hip.empty_nhwc(G * B, 2 * Ci, H, W, xin.device, xin.dtype)
                gbuf._mrdis_gb_private = True                  # hip.gb_slot takes the in-place path for tagged buffers only
                dxb = gbuf[:, Ci:]
            else:
                dxb = hip.empty_nhwc(G * B, Ci_p, H, W, dy_dev, xdt)
        sink = _grad_sink(bias) if (bias is not None and Co_p == Co) else None
        dws, db_total = [], None
        for g in range(G):
            dyg = dy[g * B:(g + 1) * B] if dy is not None else None          # (None: a four-channel fp32 gradient whose padded bf16 copy nobody has asked for yet)
            if need_x and dx4 is None:
                # ana_dec.output (C -> 4): fp32 dy x reversed taps on the Cin = 4 kernel with a bf16 output (MRDIS_DT_XBF16_YF32), else the bf16 MFMA kernel
                if dy4 is None or dxb.dtype != torch.bfloat16 or hip.conv2d_bwd_data(dy4[g * B:(g + 1) * B], tkcs[g], (H, W), 3, 3, 1, 1, out=dxb[g * B:(g + 1) * B],
                                                                                      may_decline=True) is None:
                    if dyg is None:
                        dyg = dy_pad()[g * B:(g + 1) * B]
                    wb_g = ctx.wbs[g] if ctx.wbs[g] is not None else s6_dgrad_image(dyg, tkcs[g], (H, W), kh, kw, ctx.stride)
                    hip.conv2d_bwd_data(dyg, tkcs[g], (H, W), kh, kw, ctx.stride, pad, w_bf16=wb_g, out=dxb[g * B:(g + 1) * B])
            res = None
            if dy4 is not None and x32 is None and xin is not None and xin.dtype == torch.bfloat16 and not _NO_CO4B_WGRAD:
                # ... and its weight gradient: the bf16 trunk x the fp32 gradient on the bf16 matrix pipe (wgrad_c4b_kernel<.., SWAP>) -- (9, C, 4) + the 4 bias sums
                p16 = Cof == 16 and Cif == xin.shape[1]                    # the filter is stored column-padded (mixing layout): the reduce launch writes that shape
                r4 = hip.conv2d_bwd_weight(xin if share_x else xin[g * B:(g + 1) * B], dy4[g * B:(g + 1) * B], 3, 3, 1, 1, need_bias=bias is not None, may_decline=True,
                                           pad16=p16)
                if r4 is not None:
                    dw4, db4 = r4
                    if dw4.shape[1] != Cif or dw4.shape[2] != Cof:
                        dw4 = torch.nn.functional.pad(dw4[:, :Cif], (0, Cof - 4))      # the filter's stored shape
                    if db4 is not None and Co_p != Co:
                        db4 = torch.nn.functional.pad(db4, (0, Co_p - Co))           # (cut back to Co below, like the padded kernel's)
                    res = (dw4, db4)
            if x32 is not None:          # si_layers: fp32 map x bf16 gradient on the Cin = 4 kernel -- (9, 4, C), no padded copy of the map
                res = hip.conv2d_bwd_weight(x32 if share_x else x32[g * B:(g + 1) * B], dyg, kh, kw, 1, pad, need_bias=bias is not None, bias_sink=sink, may_decline=True,
                                            pad16=(Cif == 16 and Ci == 4 and dyg.dtype == torch.bfloat16))     # (9, 16, C) straight from the reduce launch
                if res is None:
                    # maps narrower than 64: the fp32 thin-layer kernel on the fp32 map and an fp32 copy of the (small) gradient -- the bf16 kernel
                    # would run a 16-row padded layer through 128 slabs (19 vs 45 us per call at 32x32)
                    res = hip.conv2d_bwd_weight(x32 if share_x else x32[g * B:(g + 1) * B], hip.cast_view(dyg, torch.float32), kh, kw, 1, pad,
                                                need_bias=bias is not None, bias_sink=sink)
                if Cif > res[0].shape[1]:
                    res = (torch.nn.functional.pad(res[0], (0, 0, 0, Cif - Ci)), res[1])      # rows of the 16-row mixing layout beyond the map's channels: zero
            if res is None:
                if xin is None:
                    xin = hip.cast_view(x32, torch.bfloat16, Ci_p)
                xg = xin if share_x else xin[g * B:(g + 1) * B]
                if dyg is None:
                    dyg = dy_pad()[g * B:(g + 1) * B]
                res = hip.conv2d_bwd_weight(xg, dyg, kh, kw, ctx.stride, pad, need_bias=bias is not None, bias_sink=sink, dtype=dt)
                if padded:
                    res = (res[0][:, :Cif, :Cof].contiguous(), res[1])
            dw, db = res
            dws += [dw, None]
            if db is not None:
                db_total = db if db_total is None else db_total + db
        if db_total is not None and Co_p != Co:
            db_total = db_total[:Co]
            s_ = _grad_sink(bias)
            if s_ is not None:
                s_.add_(db_total); db_total = None
        dx = None
        if need_x and dx4 is not None:
            dx = dx4.permute(0, 2, 3, 1).reshape(G, B, H, W, 4).sum(0).permute(0, 3, 1, 2) if (share_x and G > 1) else dx4
        elif need_x:
            # a shared input collects the gradients of all G uses: one reduction over the group axis
            dx = dxb.permute(0, 2, 3, 1).reshape(G, B, H, W, Ci_p).sum(0).permute(0, 3, 1, 2) if share_x else dxb
            if dx.dtype != x_dtype or Ci_p != Ci:
                dx = hip.cast_view(dx, x_dtype, Ci)
        return (dx, db_total, None, None, None, None, None, None, None, None) + tuple(dws)


class _GbSpadeFn(Function):
    """gamma | beta convolution + InstanceNorm modulation of a SPADE block (model.py:2440-2446) for G sample blocks with G fused filters,
    as one autograd node.  fp32: the convolution's epilogue applies the modulation (hip.gb_spade_fwd: the 2C-channel gamma | beta tensor
    is never written or re-read -- 0.8 GB less HBM traffic per full-resolution block call); elsewhere conv + instnorm_spade kernels.
    The backward is that of the two-step form (it needs gamma, which the fused kernel stores)."""

    @staticmethod
    def forward(ctx, si_out, z, bias, G, eps, smean, srstd, up2_src, *filt):
        # smean / srstd: the instance statistics of z when its producer already took them (ops.bilinear_up2), else None
        # up2_src: the map x with z = bilinear_up2(x), or None.  Given, z arrives DETACHED and the node's backward returns d x instead of d z: the resize's
        # adjoint runs inside the SPADE backward kernel (hip.instnorm_spade_bwd up2=True) and the full-resolution d z never exists.
        bm = _COMPUTE_DTYPE == hip.DT_F32_BF16M
        B = z.shape[0] // G
        C, H, W = z.shape[1], z.shape[2], z.shape[3]
        mix = gamma = mean = rstd = None
        wbs = [None] * G
        fusable = (_COMPUTE_DTYPE == hip.DT_F32 and si_out.dtype == torch.float32 and z.dtype == torch.float32) or \
                  (_COMPUTE_DTYPE == hip.DT_BF16 and si_out.dtype == torch.bfloat16 and z.dtype == torch.bfloat16)
        if _GB_SPADE and fusable:
            mix = hip.empty_nhwc(G * B, C, H, W, z.device, z.dtype); gamma = hip.empty_nhwc(G * B, C, H, W, z.device, z.dtype)
            if smean is not None:
                mean, rstd = smean, srstd
            else:
                mean = torch.empty(G * B * C, dtype=torch.float32, device=z.device); rstd = torch.empty_like(mean)
            for g in range(G):
                sl = slice(g * B, (g + 1) * B)
                wb = bf16_filters(filt[2 * g], filt[2 * g + 1]) if _COMPUTE_DTYPE == hip.DT_BF16 else (None, None)
                im = wino_images(filt[2 * g])
                wbs[g] = wb[1] if wb[1] is not None else im.get('dgrad')
                ok = hip.gb_spade_fwd(si_out[sl], filt[2 * g], bias, z[sl], eps, w_bf16=wb[0], stats_ready=smean is not None, w_wino=im.get('spade'),
                                      out=(mix[sl], gamma[sl], mean[g * B * C:(g + 1) * B * C], rstd[g * B * C:(g + 1) * B * C]))
                if ok is None:
                    assert g == 0                    # the decision depends on the geometry only
                    mix = None
                    break
        if mix is None:
            zs = z
            if _COMPUTE_DTYPE == hip.DT_BF16 and z.dtype == torch.float32:
                zs = hip.cast_view(z, torch.bfloat16)
            gb = hip.empty_nhwc(G * B, 2 * C, H, W, z.device, si_out.dtype)
            for g in range(G):
                wb = bf16_filters(filt[2 * g], filt[2 * g + 1]) if _COMPUTE_DTYPE != hip.DT_F32 else (None, wino_images(filt[2 * g]).get('dgrad'))
                wbs[g] = wb[1]
                hip.conv2d_fwd(si_out[g * B:(g + 1) * B], filt[2 * g], bias, 3, 3, 1, 1, out=gb[g * B:(g + 1) * B], w_bf16=wb[0])
            mix, mean, rstd = hip.instnorm_spade_fwd(zs, gb[:, :C], gb[:, C:], eps, stats=(smean, srstd) if (smean is not None and zs is z) else None)
            gamma = gb[:, :C]
            ctx.z_cast = zs is not z
            z = zs
        else:
            ctx.z_cast = False
        ctx.meta = (G, B, hip.DT_F32_BF16M if _COMPUTE_DTYPE != hip.DT_F32 else hip.DT_F32)
        ctx.wbs = wbs
        ctx.up2 = up2_src is not None
        # the backward of the resized form interpolates z from the resize's input: the up-sampled map is not kept (a quarter of the bytes saved and read)
        ctx.z_is_src = ctx.up2 and not ctx.z_cast and up2_src.dtype == z.dtype
        ctx.save_for_backward(si_out, up2_src if ctx.z_is_src else z, gamma, mean, rstd, *[filt[2 * g + 1] for g in range(G)])
        return mix

    @staticmethod
    def backward(ctx, dmix):
        G, B, dt = ctx.meta
        si_out, z, gamma, mean, rstd = ctx.saved_tensors[:5]
        tkcs = ctx.saved_tensors[5:]
        H, W = gamma.shape[2], gamma.shape[3]
        dxs = None
        if ctx.up2:
            if ctx.z_is_src:
                res = hip.instnorm_spade_bwd(dmix, None, gamma, mean, rstd, fused_gb=True, up2=True, xlo=z)
                if res is None:
                    z = hip.bilinear_fwd(z, (H, W), False)
            else:
                res = hip.instnorm_spade_bwd(dmix, z, gamma, mean, rstd, fused_gb=True, up2=True) if not ctx.z_cast else None
            if res is not None:
                dxs, dgb = res
                dz = None
        if dxs is None:
            dz, dgb = hip.instnorm_spade_bwd(dmix, z, gamma, mean, rstd, fused_gb=True)
            if ctx.z_cast:
                dz = hip.cast_view(dz, torch.float32)
            if ctx.up2:                              # the library declined the fused form: the resize's adjoint as a kernel of its own
                dxs = hip.bilinear_bwd(dz, (H // 2, W // 2), False)
                dz = None
        need_x = ctx.needs_input_grad[0]
        dx = hip.empty_nhwc(G * B, si_out.shape[1], H, W, gamma.device, si_out.dtype) if need_x else None
        dws, db_total = [], None
        for g in range(G):
            sl = slice(g * B, (g + 1) * B)
            if need_x:
                hip.conv2d_bwd_data(dgb[sl], tkcs[g], (H, W), 3, 3, 1, 1, w_bf16=ctx.wbs[g], out=dx[sl])
            dw, db = hip.conv2d_bwd_weight(si_out[sl], dgb[sl], 3, 3, 1, 1, need_bias=True, dtype=dt)
            dws += [dw, None]
            db_total = db if db_total is None else db_total + db
        return (dx, dz, db_total, None, None, None, None, dxs) + tuple(dws)


_GB_SPADE = _os.environ.get('MRDIS_GB_SPADE', '1') != '0'
_UP2_BWD_FUSED = _os.environ.get('MRDIS_UP2_BWD_FUSED', '1') != '0'
_UP2_BWD_FUSED_BF16 = _os.environ.get('MRDIS_UP2_BWD_FUSED_BF16', '1') != '0'      # the fused form on bf16 maps too (round 5, 512-thread kernel with all loads in flight: 1323 -> 971 us per full-resolution block of 128 images, level on the 128x128 level)      # SPADE backward + the adjoint of the x2 resize in front of it as one kernel


def set_up2_bwd_fused(enabled):
    global _UP2_BWD_FUSED
    _UP2_BWD_FUSED = bool(enabled)

_GB_INPLACE = _os.environ.get('MRDIS_GB_INPLACE', '1') != '0'      # d(mix) written straight into the beta half of [dgamma | dbeta]


def set_gb_spade(enabled):
    global _GB_SPADE
    _GB_SPADE = bool(enabled)


def gb_spade(si_out, z, filters, bias, eps):
    """si_out, z: G sample blocks; filters: G fused gamma | beta pairs (w_tck [9][C][2C], w_tkc); bias (2C).  -> mix (see _GbSpadeFn)."""
    flat = []
    for a, b in filters:
        flat += [a, b]
    smean, srstd = in_stats_of(z, eps)
    src = getattr(z, '_mrdis_up2_src', None) if _UP2_BWD_FUSED else None
    # (fp32 maps only by default: on bf16 maps -- half the bytes to save, the same LDS and ALU work -- both fused forms measured level with the kernels they replace: 80.6-80.8 vs 80.2-80.5 ms)
    if src is not None and src.requires_grad and torch.is_grad_enabled() and src.shape[0] == z.shape[0] and src.dtype == z.dtype and (z.dtype == torch.float32 or _UP2_BWD_FUSED_BF16):
        mix = _GbSpadeFn.apply(si_out, z.detach(), bias, len(filters), eps, smean, srstd, src, *flat)
    else:
        mix = _GbSpadeFn.apply(si_out, z, bias, len(filters), eps, smean, srstd, None, *flat)
    mix._mrdis_want_dgb = True                       # a grouped convolution that reads `mix` writes d(mix) into the node's [dgamma | dbeta] buffer
    return mix


def conv2d_grouped(x, filters, bias, kh, kw, pad, lrelu=False, share_x=False, co=None, stride=1):
    """x: (G * B, Ci, H, W) sample blocks (or (B, Ci, H, W) read by every group when share_x); filters: G pairs (w_tck, w_tkc).
    -> (G * B, Co, H', W'): block g = conv(x_g, filters[g]) + bias."""
    flat = []
    for a, b in filters:
        flat += [a, b]
    return _GroupedConvFn.apply(x, bias, len(filters), bool(share_x), kh, kw, pad, bool(lrelu), co, int(stride), *flat)


# --------------------------------------------------------------------------- norms
class _BatchNormTrain(Function):
    """nn.BatchNorm2d in training mode (model.py:2151, 2191, 2776-2785).  groups = G: the batch holds G sample blocks that the
    reference puts through this layer in G separate calls (the modalities of an encoder pass): statistics per block, running
    statistics updated block by block -- bit-identical to the G calls, in one launch per kernel."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, into=None, groups=1):
        # into = (buf, c0): the result is written into channels [c0, c0 + C) of the NHWC tensor `buf` (one half of a skip-connection
        # concatenation, model.py:2192) and returned as that view
        out = None
        if into is not None:
            out = into[0][:, into[1]:into[1] + x.shape[1]]
        y, mean, rstd = hip.bn_train_fwd(x, gamma, beta, running_mean, running_var, eps, momentum, out=out, groups=groups)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.params = (gamma, beta)
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        sg, sb = _grad_sink(ctx.params[0]), _grad_sink(ctx.params[1])
        if sg is not None and sb is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            dx, _, _ = hip.bn_train_bwd(dy, x, gamma, mean, rstd, sink=(sg, sb), groups=ctx.groups)
            return dx, None, None, None, None, None, None, None, None
        dx, dg, db = hip.bn_train_bwd(dy, x, gamma, mean, rstd, groups=ctx.groups)
        return dx, dg, db, None, None, None, None, None, None


def batch_norm_train(x, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1, into=None, groups=1):
    return _BatchNormTrain.apply(x, gamma, beta, running_mean, running_var, eps, momentum, into, groups)


class _CatJoin(Function):
    """torch.cat([a, b], 1) of a U-Net skip connection (model.py:2192) without the copy: both halves were WRITTEN into `buf` by their
    producers (BatchNorm with `into`); the result is `buf`, the adjoint hands each producer its channel slice of the gradient.
    A half that lives elsewhere (the first encoder level has no BatchNorm) is copied in."""

    @staticmethod
    def forward(ctx, a, b, buf):
        Ca = a.shape[1]
        ctx.Ca = Ca
        if a.data_ptr() != buf.data_ptr():
            buf[:, :Ca].copy_(a)
        if b.data_ptr() != buf[:, Ca:].data_ptr():
            buf[:, Ca:].copy_(b)
        return buf[:, :]

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.Ca], g[:, ctx.Ca:], None


def cat_join(a, b, buf):
    return _CatJoin.apply(a, b, buf)


_CAT_ELISION = _os.environ.get('MRDIS_CAT_ELISION', '1') != '0'


def cat_elision():
    return _CAT_ELISION and torch.is_grad_enabled()


class _InstNormSpade(Function):
    """InstanceNorm2d(z) * (1 + gamma) + beta  (model.py:2440, 2446)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, eps):
        out, mean, rstd = hip.instnorm_spade_fwd(z, gamma, beta, eps)
        ctx.save_for_backward(z, gamma, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, gamma, mean, rstd = ctx.saved_tensors
        dz, dg = hip.instnorm_spade_bwd(dout, z, gamma, mean, rstd)
        return dz, dg, dout, None


def instnorm_spade(z, gamma, beta, eps=1e-5):
    return _InstNormSpade.apply(z, gamma, beta, eps)


class _InstNormSpadeGB(Function):
    """Same op with gamma and beta delivered as the two channel halves of ONE tensor (the output of a
    fused gamma+beta convolution); the backward writes [dgamma | dbeta] straight into one buffer."""

    @staticmethod
    def forward(ctx, z, gb, eps):
        C = z.shape[1]
        gamma, beta = gb[:, :C], gb[:, C:]
        out, mean, rstd = hip.instnorm_spade_fwd(z, gamma, beta, eps)
        ctx.save_for_backward(z, gb, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, gb, mean, rstd = ctx.saved_tensors
        dz, dgb = hip.instnorm_spade_bwd(dout, z, gb[:, :z.shape[1]], mean, rstd, fused_gb=True)
        return dz, dgb, None


def instnorm_spade_gb(z, gb, eps=1e-5):
    return _InstNormSpadeGB.apply(z, gb, eps)


# --------------------------------------------------------------------------- resize / softmax / losses
class _Bilinear(Function):
    @staticmethod
    def forward(ctx, x, out_hw, align_corners):
        ctx.geom = (x.shape[2], x.shape[3], align_corners)
        return hip.bilinear_fwd(x, out_hw, align_corners)

    @staticmethod
    def backward(ctx, dy):
        Hi, Wi, ac = ctx.geom
        return hip.bilinear_bwd(dy, (Hi, Wi), ac), None, None


def bilinear(x, out_hw, align_corners):
    """nn.Upsample(mode='bilinear'): model.py:2175 (align_corners=True), 2432/2501 (False)."""
    return _Bilinear.apply(x, tuple(out_hw), bool(align_corners))


class _BilinearUp2Stats(Function):
    """nn.Upsample(scale_factor=(2,2), bilinear) in front of a SPADE block (model.py:2551-2573, 2622-2627) that also leaves the
    InstanceNorm statistics (:2440) of its result: mrdis_bilinear_up2_stats_fwd takes the sums from the values while it stores them,
    so the block's statistics pass over the 4x tensor disappears.  (mean, rstd) are auxiliary outputs: the SPADE node treats the
    dependence of the statistics on z analytically in its own backward, as before."""

    @staticmethod
    def forward(ctx, x, eps):
        ctx.geom = (x.shape[2], x.shape[3])
        res = hip.bilinear_up2_stats(x, eps)
        if res is None:
            y = hip.bilinear_fwd(x, (2 * x.shape[2], 2 * x.shape[3]), False)
            mean = rstd = torch.empty(0, dtype=torch.float32, device=x.device)
        else:
            y, mean, rstd = res
        ctx.mark_non_differentiable(mean, rstd)
        return y, mean, rstd

    @staticmethod
    def backward(ctx, dy, _gm, _gr):
        return hip.bilinear_bwd(dy, ctx.geom, False), None


class _Up2ScatterFn(Function):
    """_BilinearUp2Stats for the LAST resize of the shared SPADE decoder (model.py:2573), whose batch holds M sample blocks (one per anatomy
    source i) for ONE modality label j while the per-modality decoder i reads the M labels of ITS block (model.py:3200-3224): the kernel writes
    block i straight into zbuf[i][j] (mrdis_bilinear_up2_stats_fwd, out_block), so that zbuf[i] IS decoder i's batch-concatenated input -- the
    concatenation copy (268 MB per decoder call at B = 32) and the batch split disappear.  Outputs: the M block views + (mean, rstd)."""

    @staticmethod
    def forward(ctx, x, zbuf, j, eps):
        M = zbuf.shape[0]
        B = x.shape[0] // M
        ctx.geom = (x.shape[2], x.shape[3]); ctx.M = M; ctx.B = B
        blocks = zbuf[:, j * B:(j + 1) * B]                     # (M, B, C, 2H, 2W)
        res = hip.bilinear_up2_stats(x, eps, out_blocks=blocks)
        if res is None:
            raise hip.MrdisError('bilinear_up2 (scatter): unsupported geometry')
        _, mean, rstd = res
        ctx.mark_non_differentiable(mean, rstd)
        # (zbuf is written in place but not marked dirty: mark_dirty wants the tensor itself among the outputs, and the outputs are its block views; nothing
        #  saves zbuf for backward and it does not require grad, so autograd's version counters have nothing to protect before the j-loop ends)
        return tuple(blocks[i] for i in range(M)) + (mean, rstd)

    @staticmethod
    def backward(ctx, *grads):
        M, B = ctx.M, ctx.B
        gs = grads[:M]
        ref = next(g for g in gs if g is not None)
        dx = hip.empty_nhwc(M * B, ref.shape[1], ctx.geom[0], ctx.geom[1], ref.device, ref.dtype)
        for i in range(M):                                       # each block's gradient comes from another decoder's backward: no gather copy
            if gs[i] is None:
                dx[i * B:(i + 1) * B].zero_()
            else:
                hip.bilinear_bwd(gs[i], ctx.geom, False, out=dx[i * B:(i + 1) * B])
        return dx, None, None, None


class _JoinBlocksFn(Function):
    """the M blocks that _Up2ScatterFn calls wrote side by side ARE their concatenation: hand out the enclosing view, split the gradient into views"""

    @staticmethod
    def forward(ctx, whole, *parts):
        ctx.sizes = [p_.shape[0] for p_ in parts]
        return whole.view(whole.shape)

    @staticmethod
    def backward(ctx, g):
        out, o = [], 0
        for n in ctx.sizes:
            out.append(g[o:o + n]); o += n
        return (None,) + tuple(out)


_UP2_SCATTER = _os.environ.get('MRDIS_UP2_SCATTER', '1') != '0'


def set_up2_scatter(enabled):
    global _UP2_SCATTER
    _UP2_SCATTER = bool(enabled)


def up2_scatter_applies(x):
    """the block-scattered resize has no fallback inside its autograd node: ask the library beforehand whether its kernel takes this geometry (channel
    counts such as 24 or 48 do not divide its block size; the grid's y extent limits N) -- otherwise the caller runs the dense resize + concatenation"""
    return (_UP2_SCATTER and _UP2_STATS and x.is_cuda and type(x) is torch.Tensor and x.shape[1] % 4 == 0
            and hip.bilinear_up2_stats_applies(x.shape[0], x.shape[3], x.shape[1]))


def bilinear_up2_scatter(x, holder, j, M, stats_eps):
    """x: (M * B, C, H, W), block i = samples of anatomy source i, for label j of M.  holder: dict shared by the M calls of a step (the buffer
    (M, M * B, C, 2H, 2W) is made by the first).  -> list of M block tensors (B, C, 2H, 2W), each carrying its instance statistics and its place."""
    B = x.shape[0] // M
    C, H2, W2 = x.shape[1], 2 * x.shape[2], 2 * x.shape[3]
    zbuf = holder.get('buf')
    if zbuf is None:
        zbuf = torch.empty((M, M * B, H2, W2, C), dtype=x.dtype, device=x.device).permute(0, 1, 4, 2, 3)
        holder['buf'] = zbuf
    res = _Up2ScatterFn.apply(x, zbuf, int(j), float(stats_eps))
    parts, mean, rstd = res[:M], res[M], res[M + 1]
    n = B * C
    for i, part in enumerate(parts):
        part._mrdis_in_stats = (mean[i * n:(i + 1) * n], rstd[i * n:(i + 1) * n], float(stats_eps))
        part._mrdis_block = (zbuf, i, int(j))
    return list(parts)


def join_blocks(parts):
    """torch.cat(parts, 0) -- free when the parts are the label blocks 0 .. M-1 of one row of a bilinear_up2_scatter buffer"""
    tags = [getattr(p_, '_mrdis_block', None) for p_ in parts]
    if all(t is not None for t in tags) and all(t[0] is tags[0][0] and t[1] == tags[0][1] and t[2] == k for k, t in enumerate(tags)) \
            and len(parts) * parts[0].shape[0] == tags[0][0].shape[1]:
        return _JoinBlocksFn.apply(tags[0][0][tags[0][1]], *parts)
    return torch.cat(list(parts), 0)


_UP2_STATS = _os.environ.get('MRDIS_UP2_STATS', '1') != '0'


def set_up2_stats(enabled):
    global _UP2_STATS
    _UP2_STATS = bool(enabled)


def bilinear_up2(x, stats_eps=None):
    """x2 bilinear, align_corners=False.  stats_eps: the eps of the InstanceNorm that reads the result -- its statistics then ride on the
    result (`_mrdis_in_stats` = (mean, rstd, eps), picked up by ops.gb_spade)."""
    if stats_eps is None or not _UP2_STATS or not x.is_cuda or type(x) is not torch.Tensor:
        return bilinear(x, (2 * x.shape[2], 2 * x.shape[3]), False)
    y, mean, rstd = _BilinearUp2Stats.apply(x, float(stats_eps))
    if mean.numel():
        y._mrdis_in_stats = (mean, rstd, float(stats_eps))
    y._mrdis_up2_src = x                                 # ops.gb_spade: the SPADE node takes x's gradient itself (no full-resolution d z)
    return y


def in_stats_of(z, eps):
    """(mean, rstd) that ride on z for an InstanceNorm with this eps, or (None, None)"""
    st = getattr(z, '_mrdis_in_stats', None)
    if st is None or st[2] != float(eps) or st[0].numel() != z.shape[0] * z.shape[1]:
        return None, None
    return st[0], st[1]


class _SoftmaxMaskDrop(Function):
    @staticmethod
    def forward(ctx, s, mask_img, scale):
        out = hip.softmax_mask_drop_fwd(s, mask_img, scale)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        return hip.softmax_mask_drop_bwd(dout, out), None, None


def softmax_mask_drop(s, mask_img, scale=100.0):
    """softmax(cat([scale*mask_img, s], 1), 1)[:, 1:]  (model.py:3150-3153)."""
    return _SoftmaxMaskDrop.apply(s, mask_img, scale)


class _ReconErr(Function):
    @staticmethod
    def forward(ctx, gt, x, p):
        ctx.p = p
        ctx.save_for_backward(gt, x)
        return hip.recon_err_fwd(gt, x, p)

    @staticmethod
    def backward(ctx, w):
        gt, x = ctx.saved_tensors
        dx = hip.recon_err_bwd(gt, x, w, ctx.p)
        return (-dx if ctx.needs_input_grad[0] else None), dx, None


def recon_err(gt, x, p):
    """per-sample mean over (C,H,W) of |gt-x|^p, p in {1,2}  (model.py:3260-3266)."""
    return _ReconErr.apply(gt, x, int(p))


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x, k):
        y, arg = hip.maxpool_fwd(x, k)
        ctx.k, ctx.shape = k, tuple(x.shape)
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        return hip.maxpool_bwd(dy, arg, ctx.shape, ctx.k), None


def max_pool(x, k):
    """F.max_pool2d(x, kernel_size=(k,k))  (model.py:3449)."""
    return _MaxPool.apply(x, int(k))


# --------------------------------------------------------------------------- mixed-kernel cache
# The reference re-mixes the experts on every CondConv2d call.  Inside one training step the same
# (layer, modality type) pair recurs (encoder passes 1 and 2, SPADEShared for every s_i), so the
# trainer opens this scope and the mix node is shared -- autograd sums the kernel gradients of all
# uses and runs the mix backward once.  Outside the scope nothing is cached.
_MIX_CACHE = None


@contextlib.contextmanager
def mix_cache():
    global _MIX_CACHE
    prev, _MIX_CACHE = _MIX_CACHE, {}
    try:
        yield
    finally:
        _MIX_CACHE = prev


def step_cache(key, make):
    """generic per-step memo (same scope as the mixed-kernel cache): e.g. the bilinear resizes of an
    anatomy map s_i, which every SPADE block of every decoder call recomputes in the reference."""
    return cached_mix(key, make)


def mix_cache_active():
    return _MIX_CACHE is not None


def to_device(t_cpu, device):
    """Host tensor -> device without stalling the host OR the GPU.  A pageable source makes the copy wait for everything already
    queued on the stream (62 ms per step were spent in three such copies); a pinned source is stream-ordered, but the runtime still
    waits on the host for the preceding kernel before it programs the copy engine: ~0.5 ms of GPU idle time per transfer, ten
    transfers per step (the CPU-drawn eps of `sample`, loss weights, index lists).  Small tensors therefore go through a ring of
    pinned slots that a copy KERNEL reads (hip.to_device_small); anything else through the pinned-memory cache.
    While a step is being recorded for graph replay (HostValues.recording) the tensor is a CONSTANT of the recorded step: it gets a slot
    of the step's host-value block and is sent again, unchanged, with every replay."""
    if torch.device(device).type != 'cuda':
        return t_cpu.to(device)
    if _HOST_VALUES is not None and _HOST_VALUES.recording:
        return _HOST_VALUES.add(lambda: t_cpu, t_cpu)
    if _MAILBOX:
        out = hip.to_device_small(t_cpu, device)
        if out is not None:
            return out
    return t_cpu.pin_memory().to(device, non_blocking=True)


# --------------------------------------------------------------------------- host-drawn values of a step (graph replay)
# A training step consumes values that only the HOST can produce: eps of `sample` (the reference draws it from the CPU generator,
# model.py:3159-3162), the (i, j) pair of sim_s / adv_s (np.random.choice, :3485), the loss weights derived from the batch's mask.
# The model hands each of them over as a CLOSURE (host_value(fn, device)): an eager step calls it once and ships the result; a step
# recorded into a HIP graph (trainer.GraphedTrainStep) keeps the closure, and every replay calls the closures again IN THE RECORDED
# ORDER -- the global torch / numpy generators advance exactly as in the eager step -- packs the results into one pinned block and
# ships the block with ONE copy kernel in front of the graph launch; the recorded kernels read their values from fixed device slots.
_HOST_VALUES = None
_STEP_MASK_HOST = None        # the (B, M) numpy mask of the step being built / replayed (closures read it through step_mask_host())


def step_mask_host():
    """the host mask closures must use: the replayed step's own mask while a recorded step refills its values, else None (= the mask the
    eager caller passed)"""
    return _STEP_MASK_HOST if _REPLAYING else None


_REPLAYING = False


def set_step_mask_host(mh):
    global _STEP_MASK_HOST
    _STEP_MASK_HOST = None if mh is None else _np.asarray(mh.numpy() if isinstance(mh, torch.Tensor) else mh, dtype=_np.float32)


class HostValues:
    """The host-value block of ONE recorded step: a device buffer with a 16-byte aligned slot per value, the closures that
    produce the values, and a small ring of pinned staging blocks (a block is rewritten only after the event behind its last
    copy kernel has completed, so the host may run several replays ahead of the GPU)."""
    CAP = 1 << 18           # bytes
    NRING = 4

    def __init__(self, device):
        self.device = device
        self.dev = torch.zeros(self.CAP, dtype=torch.uint8, device=device)
        self.ring = [torch.zeros(self.CAP, dtype=torch.uint8).pin_memory() for _ in range(self.NRING)]
        self.events = [None] * self.NRING
        self.k = 0
        self.sources = []       # (fn, offset, nbytes, shape, dtype)
        self.used = 0
        self.recording = False
        self.stage = None       # the pinned block being filled for the next launch

    def _begin_fill(self):
        k = self.k
        self.k = (k + 1) % self.NRING
        ev = self.events[k]
        if ev is not None:
            ev.synchronize()
        self.stage = self.ring[k]
        self._stage_k = k

    def add(self, fn, value=None):
        """recording: call fn (unless its value is given), give it a slot, return the device view the recorded kernels will read"""
        t = fn() if value is None else value
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        off = self.used
        if nbytes == 0 or off + nbytes > self.CAP:
            raise RuntimeError(f'host-value block full ({off} + {nbytes} of {self.CAP} bytes)')
        self.used = (off + nbytes + 15) // 16 * 16
        self.sources.append((fn, off, nbytes, tuple(t.shape), t.dtype))
        self.stage[off:off + nbytes].copy_(t.view(-1).view(torch.uint8))
        return self.dev[off:off + nbytes].view(t.dtype).view(t.shape)

    def start_recording(self):
        self.sources, self.used = [], 0
        self._begin_fill()
        self.recording = True

    def stop_recording(self):
        self.recording = False

    def refill(self):
        """replay: call every closure again, in the recorded order, into a fresh pinned block"""
        global _REPLAYING
        self._begin_fill()
        st = self.stage
        _REPLAYING = True
        try:
            for fn, off, nbytes, shape, dtype in self.sources:
                t = fn().contiguous()
                if tuple(t.shape) != shape or t.dtype != dtype:
                    raise RuntimeError(f'host value changed its shape between the recorded step and a replay: {tuple(t.shape)} {t.dtype} vs {shape} {dtype}')
                st[off:off + nbytes].copy_(t.view(-1).view(torch.uint8))
        finally:
            _REPLAYING = False

    def ship(self):
        """one copy kernel: pinned block -> device slots, on the current stream (in front of the graph launch)"""
        n = max(16, self.used)
        hip.copy_bytes(self.stage, self.dev, n)
        k = self._stage_k
        if self.events[k] is None:
            self.events[k] = torch.cuda.Event()
        self.events[k].record()


def host_value(fn, device):
    """fn() -> CPU tensor produced on the host for THIS step (a draw from the global generators, a function of step_mask_host()).
    Eager: called once, shipped through the mailbox.  Recording for graph replay: kept, see HostValues."""
    if _HOST_VALUES is not None and _HOST_VALUES.recording and torch.device(device).type == 'cuda':
        return _HOST_VALUES.add(fn)
    return to_device(fn(), device)


_FORCED_PAIRS = None


def set_forced_pairs(pairs):
    """{'sim_s': (i, j), 'adv_s': (i, j)} drawn by the caller (trainer.GraphedTrainStep draws them from np.random in the model's own order
    BEFORE the step, because the adv_s pair selects which recorded graph replays) or None: the model draws for itself (model.py:3485)."""
    global _FORCED_PAIRS
    _FORCED_PAIRS = pairs


def forced_pair(kind):
    return None if _FORCED_PAIRS is None else _FORCED_PAIRS.get(kind)


def recording_host_values():
    """a step is being recorded for graph replay: host draws that select tensors (sim_s / adv_s pairs) must become data"""
    return _HOST_VALUES is not None and _HOST_VALUES.recording


def set_host_values(hv):
    global _HOST_VALUES
    prev, _HOST_VALUES = _HOST_VALUES, hv
    return prev


_MAILBOX = _os.environ.get('MRDIS_MAILBOX', '1') != '0'


def cached_mix(key, make):
    if _MIX_CACHE is None:
        return make()
    hit = _MIX_CACHE.get(key)
    if hit is None:
        hit = _MIX_CACHE[key] = make()
    return hit


# --------------------------------------------------------------------------- six-product filter images (mrdis_s6conv.hip)
# The data gradient of the 4x4 stride-2 encoder convolutions (model.py:2104 under :1935-1990) has no Winograd form; on maps of >= 16k positions the
# six-product kernel (fp32 operands as three bf16 terms on the bf16 matrix pipe, fp32-equivalent results) runs it in 69-80 us where the fp32 MFMA kernel
# takes 96-112 (tools/s6conv_check.py; the forward pass and the smaller maps gain nothing and stay where they were).  The filter's image is built once per
# mixed kernel and step (the mix cache scope) by one small launch.  MRDIS_S6_DGRAD=0: off.
_S6_DGRAD = _os.environ.get('MRDIS_S6_DGRAD', '1') != '0'


def s6_dgrad_image(dy, w_tkc, in_hw, kh, kw, stride):
    if not _S6_DGRAD or stride != 2 or kh != 4 or kw != 4 or type(dy) is not torch.Tensor or dy.dtype is not torch.float32 or not dy.is_cuda \
            or w_tkc.dtype is not torch.float32 or w_tkc.shape[1] % 8 != 0 or w_tkc.shape[2] % 4 != 0 or w_tkc.shape[2] < 16 \
            or dy.shape[0] * in_hw[0] * in_hw[1] < 32768 or _COMPUTE_DTYPE != hip.DT_F32:
        return None
    return cached_mix(('s6d', id(w_tkc)), lambda: (w_tkc, hip.s6_filter_image(w_tkc)))[1]      # (the tuple keeps w_tkc alive: its id stays unique inside the scope)


# --------------------------------------------------------------------------- torch.ops.mrdis.*
# Dispatcher-visible custom ops over the C ABI: schema, CUDA kernel (ctypes -> libmrdis_hip.so), fake
# (meta) kernel and autograd formula for each, so `torch.ops.mrdis.*` composes with torch's tooling (opcheck,
# FakeTensor shape propagation, AOT tracing).  The seam op is `mrdis::cond_conv2d` = CondConv2d.forward of the reference
# (model.py:2108-2117) for a batch-constant type; `mrdis::conv2d` is the same convolution on an already mixed kernel
# (what the model calls inside a training step, where one mix serves 8-16 calls).
_lib = torch.library.Library('mrdis', 'DEF')
_lib.define('conv2d(Tensor x, Tensor w_tck, Tensor w_tkc, Tensor? bias, int kh, int kw, int stride, int pad, bool lrelu, Tensor? wb_fwd=None, Tensor? wb_bwd=None) -> Tensor')
_lib.define('conv2d_bwd_data(Tensor dy, Tensor w_tkc, int H, int W, int kh, int kw, int stride, int pad, Tensor? w_bf16=None) -> Tensor')
_lib.define('conv2d_bwd_weight(Tensor x, Tensor dy, int kh, int kw, int stride, int pad, bool need_bias, int dtype=0) -> (Tensor, Tensor)')
_lib.define('conv2d_bwd_weight_sink(Tensor x, Tensor dy, int kh, int kw, int stride, int pad, Tensor(a!) bias_grad, int dtype=0) -> Tensor')
_lib.define('lrelu_bwd(Tensor dy, Tensor y, float slope) -> Tensor')
_lib.define('mix_experts_routed(Tensor W, Tensor fc_w, Tensor fc_b, Tensor type_row) -> (Tensor, Tensor, Tensor)')
_lib.define('mix_experts_routed_bwd(Tensor dw_tck, Tensor W, Tensor r, Tensor type_row) -> (Tensor, Tensor, Tensor)')
_lib.define('cond_conv2d(Tensor x, Tensor type_row, Tensor weight, Tensor fc_w, Tensor fc_b, Tensor? bias, int stride, int pad, bool lrelu, int dtype=0) -> Tensor')


def _nhwc_like(x, N, C, H, W):
    return torch.empty((N, C, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)


def _out_hw(x, kh, kw, stride, pad):
    return hip.conv_out_hw(x.shape[2], x.shape[3], kh, kw, stride, pad)


# ---- kernels (CUDA = the HIP device under PyTorch-ROCm)
def _k_conv2d(x, w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu, wb_fwd=None, wb_bwd=None):
    return hip.conv2d_fwd(x, w_tck, bias, kh, kw, stride, pad, lrelu, w_bf16=wb_fwd)


def _k_bwd_data(dy, w_tkc, H, W, kh, kw, stride, pad, w_bf16=None):
    return hip.conv2d_bwd_data(dy, w_tkc, (H, W), kh, kw, stride, pad, w_bf16=w_bf16)


def _k_bwd_weight(x, dy, kh, kw, stride, pad, need_bias, dtype=0):
    dw, db = hip.conv2d_bwd_weight(x, dy, kh, kw, stride, pad, need_bias=need_bias, dtype=dtype)
    return dw, (db if db is not None else dw.new_zeros(dy.shape[1]))


def _k_bwd_weight_sink(x, dy, kh, kw, stride, pad, bias_grad, dtype=0):
    return hip.conv2d_bwd_weight(x, dy, kh, kw, stride, pad, need_bias=True, bias_sink=bias_grad, dtype=dtype)[0]


def _k_mix(W, fc_w, fc_b, type_row):
    return hip.mix_experts_routed_fwd(W, fc_w, fc_b, type_row)


def _k_mix_bwd(dw_tck, W, r, type_row):
    return hip.mix_experts_routed_bwd(dw_tck, W, r, type_row, type_row.shape[-1])


def _k_cond_conv2d(x, type_row, weight, fc_w, fc_b, bias, stride, pad, lrelu, dtype=0):
    if dtype not in (hip.DT_F32, hip.DT_F32_BF16M) or x.dtype != torch.float32:
        raise hip.MrdisError('mrdis::cond_conv2d takes fp32 activations (dtype 0 or 1); bf16 storage goes through mrdis::conv2d')
    w_tck, w_tkc, _ = hip.mix_experts_routed_fwd(weight, fc_w, fc_b, type_row)
    wb = hip.cast_bf16(w_tkc) if dtype == hip.DT_F32_BF16M else None
    return hip.conv2d_fwd(x, w_tck, bias, weight.shape[3], weight.shape[4], stride, pad, lrelu, w_bf16=wb)


for _name, _fn in (('conv2d', _k_conv2d), ('conv2d_bwd_data', _k_bwd_data), ('conv2d_bwd_weight', _k_bwd_weight),
                   ('conv2d_bwd_weight_sink', _k_bwd_weight_sink), ('lrelu_bwd', lambda dy, y, slope: hip.lrelu_bwd(dy, y, slope)),
                   ('mix_experts_routed', _k_mix), ('mix_experts_routed_bwd', _k_mix_bwd), ('cond_conv2d', _k_cond_conv2d)):
    _lib.impl(_name, _fn, 'CUDA')


# ---- fake kernels: shapes / strides only
@torch.library.register_fake('mrdis::conv2d')
def _f_conv2d(x, w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu, wb_fwd=None, wb_bwd=None):
    Ho, Wo = _out_hw(x, kh, kw, stride, pad)
    return _nhwc_like(x, x.shape[0], w_tck.shape[2], Ho, Wo)


@torch.library.register_fake('mrdis::conv2d_bwd_data')
def _f_bwd_data(dy, w_tkc, H, W, kh, kw, stride, pad, w_bf16=None):
    return _nhwc_like(dy, dy.shape[0], w_tkc.shape[2], H, W)


@torch.library.register_fake('mrdis::conv2d_bwd_weight')
def _f_bwd_weight(x, dy, kh, kw, stride, pad, need_bias, dtype=0):
    return x.new_empty((kh * kw, x.shape[1], dy.shape[1])), x.new_empty((dy.shape[1],))


@torch.library.register_fake('mrdis::conv2d_bwd_weight_sink')
def _f_bwd_weight_sink(x, dy, kh, kw, stride, pad, bias_grad, dtype=0):
    return x.new_empty((kh * kw, x.shape[1], dy.shape[1]))


@torch.library.register_fake('mrdis::lrelu_bwd')
def _f_lrelu_bwd(dy, y, slope):
    return _nhwc_like(y, *y.shape)


@torch.library.register_fake('mrdis::mix_experts_routed')
def _f_mix(W, fc_w, fc_b, type_row):
    E, Co, Ci, kh, kw = W.shape
    return W.new_empty((kh * kw, Ci, Co)), W.new_empty((kh * kw, Co, Ci)), W.new_empty((E,))


@torch.library.register_fake('mrdis::mix_experts_routed_bwd')
def _f_mix_bwd(dw_tck, W, r, type_row):
    return torch.empty_like(W, memory_format=torch.contiguous_format), W.new_empty((W.shape[0], type_row.shape[-1])), W.new_empty((W.shape[0],))


@torch.library.register_fake('mrdis::cond_conv2d')
def _f_cond_conv2d(x, type_row, weight, fc_w, fc_b, bias, stride, pad, lrelu, dtype=0):
    Ho, Wo = _out_hw(x, weight.shape[3], weight.shape[4], stride, pad)
    return _nhwc_like(x, x.shape[0], weight.shape[1], Ho, Wo)


# ---- autograd formulas (backward = other mrdis ops, so it traces too)
class _Conv2dFn(Function):
    """autograd kernel of mrdis::conv2d, registered on the Autograd dispatch key.  (torch.library.register_autograd's generic
    wrapper costs ~45 us of Python per call -- fill_defaults, keyset bookkeeping -- and the step makes 326 such calls; this
    Function + one redispatch below the Autograd key is ~15 us.)"""

    @staticmethod
    def forward(ctx, x, w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu, wb_fwd, wb_bwd):
        with torch._C._AutoDispatchBelowAutograd():
            y = torch.ops.mrdis.conv2d(x, w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu, wb_fwd, wb_bwd)
        ctx.geom = (kh, kw, stride, pad, lrelu, x.shape[2], x.shape[3])
        ctx.bias_param = bias                         # the Parameter object: its .grad may be an in-kernel gradient sink
        aux = wb_fwd if wb_fwd is not None else wb_bwd
        ctx.dtype = hip.DT_F32_BF16M if (aux is not None and aux.dtype is torch.bfloat16) else hip.DT_F32
        ctx.save_for_backward(x, w_tkc, y if lrelu else None, wb_bwd)
        return y

    @staticmethod
    def backward(ctx, dy):
        kh, kw, stride, pad, lrelu, H, W = ctx.geom
        x, w_tkc, y, wb_bwd = ctx.saved_tensors
        has_bias = ctx.bias_param is not None
        want_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[3])
        dw = db = None
        if type(dy) is torch.Tensor and type(x) is torch.Tensor:
            # plain device tensors (the training step): straight to the C ABI -- three dispatcher round trips per convolution
            # (~10 us of Python each, ~1000 per step) buy nothing here
            if lrelu:
                dy = hip.lrelu_bwd(dy, y, 0.2)
            if wb_bwd is None and ctx.needs_input_grad[0]:
                wb_bwd = s6_dgrad_image(dy, w_tkc, (H, W), kh, kw, stride)
            dx = hip.conv2d_bwd_data(dy, w_tkc, (H, W), kh, kw, stride, pad, w_bf16=wb_bwd) if ctx.needs_input_grad[0] else None
            if want_w:
                sink = _grad_sink(ctx.bias_param) if has_bias else None
                dw, db = hip.conv2d_bwd_weight(x, dy, kh, kw, stride, pad, need_bias=has_bias, bias_sink=sink, dtype=ctx.dtype)
            return dx, dw, None, db, None, None, None, None, None, None, None
        # tensor subclasses (FakeTensor, functional tensors: opcheck, AOT tracing): the backward stays a composition of mrdis ops
        if lrelu:
            dy = torch.ops.mrdis.lrelu_bwd(dy, y, 0.2)
        dx = torch.ops.mrdis.conv2d_bwd_data(dy, w_tkc, H, W, kh, kw, stride, pad, wb_bwd) if ctx.needs_input_grad[0] else None
        if want_w:
            sink = _grad_sink(ctx.bias_param) if has_bias else None
            if sink is not None:
                dw = torch.ops.mrdis.conv2d_bwd_weight_sink(x, dy, kh, kw, stride, pad, sink, ctx.dtype)
            else:
                dw, db = torch.ops.mrdis.conv2d_bwd_weight(x, dy, kh, kw, stride, pad, has_bias, ctx.dtype)
                if not has_bias:
                    db = None
        return dx, dw, None, db, None, None, None, None, None, None, None


def _conv2d_autograd(x, w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu, wb_fwd=None, wb_bwd=None):
    return _Conv2dFn.apply(x, w_tck, w_tkc, bias, kh, kw, stride, pad, lrelu, wb_fwd, wb_bwd)


_lib.impl('conv2d', _conv2d_autograd, 'Autograd')


def _mix_setup(ctx, inputs, output):
    W, fc_w, fc_b, type_row = inputs
    ctx.save_for_backward(W, output[2], type_row)


def _mix_backward(ctx, g_tck, _g_tkc, _g_r):
    W, r, type_row = ctx.saved_tensors
    dW, dfcw, dfcb = torch.ops.mrdis.mix_experts_routed_bwd(g_tck.contiguous(), W, r, type_row)
    return dW, dfcw, dfcb, None


torch.library.register_autograd('mrdis::mix_experts_routed', _mix_backward, setup_context=_mix_setup)


def _cond_setup(ctx, inputs, output):
    x, type_row, weight, fc_w, fc_b, bias, stride, pad, lrelu, dtype = inputs
    ctx.geom = (stride, pad, lrelu)
    ctx.dtype = dtype
    ctx.has_bias = bias is not None
    ctx.save_for_backward(x, type_row, weight, fc_w, fc_b, output if lrelu else None)


def _cond_backward(ctx, dy):
    stride, pad, lrelu = ctx.geom
    x, type_row, weight, fc_w, fc_b, y = ctx.saved_tensors
    kh, kw = weight.shape[3], weight.shape[4]
    if lrelu:
        dy = torch.ops.mrdis.lrelu_bwd(dy, y, 0.2)
    # the kernel is re-mixed here: nothing but x is kept alive between forward and backward
    w_tck, w_tkc, r = torch.ops.mrdis.mix_experts_routed(weight, fc_w, fc_b, type_row)
    wb = hip.cast_bf16(w_tck) if (ctx.dtype == hip.DT_F32_BF16M and w_tck.is_cuda and not isinstance(w_tck, torch._subclasses.FakeTensor)) else None
    dx = torch.ops.mrdis.conv2d_bwd_data(dy, w_tkc, x.shape[2], x.shape[3], kh, kw, stride, pad, wb) if ctx.needs_input_grad[0] else None
    dw_tck, db = torch.ops.mrdis.conv2d_bwd_weight(x, dy, kh, kw, stride, pad, ctx.has_bias, ctx.dtype)
    dW, dfcw, dfcb = torch.ops.mrdis.mix_experts_routed_bwd(dw_tck, weight, r, type_row)
    return dx, None, dW, dfcw, dfcb, (db if ctx.has_bias else None), None, None, None, None


torch.library.register_autograd('mrdis::cond_conv2d', _cond_backward, setup_context=_cond_setup)
