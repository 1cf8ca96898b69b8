"""ctypes binding of libmrdis_hip.so (include/mrdis.h) for torch tensors.

Plumbing only: torch owns device memory and the HIP stream; every function
here turns tensors into (device pointer, leading dimension) views and
enqueues the hand-written gfx950 kernels on torch's current stream.  There is
NO fallback: if the shared library is missing, importing the product path on
a GPU box fails loudly (`MrdisLibraryError`).
"""
import collections
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libmrdis_hip.so')


class MrdisLibraryError(RuntimeError):
    pass


class MrdisError(RuntimeError):
    pass


_lib = None
_c = ctypes
_P, _I, _L, _F, _Z = _c.c_void_p, _c.c_int, _c.c_longlong, _c.c_float, _c.c_size_t

_SIGS = {
    'mrdis_strerror': (_c.c_char_p, [_I]),
    'mrdis_version': (_I, []),
    'mrdis_set_option': (_I, [_c.c_char_p, _L]),
    'mrdis_get_option': (_L, [_c.c_char_p]),
    'mrdis_launch_count': (_L, [_c.c_char_p]),
    'mrdis_launch_count_reset': (None, []),
    'mrdis_dynamic_lds_table': (_I, [_c.c_char_p, _I]),
    'mrdis_mix_experts_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'mrdis_mix_experts_bwd_workspace': (_Z, [_I, _I, _I, _I]),
    'mrdis_mix_experts_bwd': (_I, [_P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _P]),
    'mrdis_mix_experts_routed_fwd': (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    'mrdis_mix_experts_routed_bwd': (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _P]),
    'mrdis_copy_bytes': (_I, [_P, _P, _L, _P]),
    'mrdis_stream_fill': (_I, [_P, _L, _F, _P]),
    'mrdis_instnorm_stats': (_I, [_P, _I, _P, _P, _P, _Z, _I, _L, _I, _F, _I, _P]),
    'mrdis_conv2d_fwd_spade': (_I, [_P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    'mrdis_wino_u_job_bytes': (_Z, []),
    'mrdis_wino_u_format': (_I, [_I, _I, _I]),
    'mrdis_wino_u_image_floats': (_L, [_I, _I, _I]),
    'mrdis_s6_filter_image_bytes': (ctypes.c_size_t, [_I, _I, _I]),
    'mrdis_s6_filter_image': (_I, [_P, _I, _I, _I, _P, ctypes.c_size_t, _P]),
    'mrdis_wino_u_image_floats_fmt': (_L, [_I, _I, _I, _I]),
    'mrdis_wino_u_job_blocks': (_I, [_I, _I, _I]),
    'mrdis_wino_u_jobs': (_I, [_P, _I, _I, _P]),
    'mrdis_mix_experts_routed_multi_fwd': (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _P]),
    'mrdis_mix_experts_routed_multi_bwd_workspace': (_Z, [_I, _I, _I, _I, _I]),
    'mrdis_mix_experts_routed_multi_bwd': (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _P, _Z, _I, _I, _I, _I, _P]),
    'mrdis_mix_job_bytes': (_Z, []),
    'mrdis_mix_job_blocks': (_I, [_I, _I, _I]),
    'mrdis_mix_jobs_fwd': (_I, [_P, _I, _I, _P, _I, _I, _P]),
    'mrdis_mix_jobs_bwd': (_I, [_P, _I, _I, _P, _P, _I, _I, _P]),
    'mrdis_conv2d_fwd': (_I, [_P, _I, _P, _P, _P, _P, _I] + [_I] * 11 + [_P, _I, _P]),
    'mrdis_conv2d_bwd_data': (_I, [_P, _I, _P, _P, _P, _I] + [_I] * 10 + [_P, _I, _P]),
    'mrdis_cast_bf16': (_I, [_P, _P, _L, _P]),
    'mrdis_cast_view': (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _L, _P]),
    'mrdis_conv2d_bwd_weight_workspace': (_Z, [_I] * 9),
    'mrdis_conv2d_bwd_weight': (_I, [_P, _I, _P, _I, _P, _P, _P, _Z] + [_I] * 11 + [_P]),
    'mrdis_lrelu_bwd': (_I, [_P, _I, _P, _I, _P, _I, _L, _I, _F, _I, _P]),
    'mrdis_norm_workspace': (_Z, [_I, _L, _I]),
    'mrdis_bn_train_fwd': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _L, _I, _F, _F, _I, _I, _P]),
    'mrdis_bn_eval_fwd': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _L, _I, _F, _I, _P]),
    'mrdis_bn_train_bwd': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _L, _I, _I, _I, _P]),
    'mrdis_instnorm_spade_fwd': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _Z, _I, _L, _I, _F, _I, _P]),
    'mrdis_instnorm_spade_bwd_workspace': (_Z, [_I, _L, _I]),
    'mrdis_instnorm_spade_bwd': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _P, _I, _P, _Z, _I, _L, _I, _I, _P]),
    'mrdis_instnorm_spade_bwd_up2_workspace': (_Z, [_I, _I, _I, _I, _I]),
    'mrdis_instnorm_spade_bwd_up2': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _P, _I, _P, _Z, _I, _I, _I, _I, _P, _I, _I, _P]),
    'mrdis_bilinear_fwd': (_I, [_P, _I, _P, _I] + [_I] * 8 + [_P]),
    'mrdis_bilinear_bwd': (_I, [_P, _I, _P, _I] + [_I] * 8 + [_P]),
    'mrdis_bilinear_up2_stats_workspace': (_Z, [_I, _I, _I]),
    'mrdis_bilinear_up2_stats_applies': (_I, [_I, _I, _I]),
    'mrdis_bilinear_up2_stats_fwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _L, _P, _P, _F, _P, _Z, _I, _P]),
    'mrdis_softmax_mask_drop_fwd': (_I, [_P, _I, _P, _P, _I, _L, _I, _F, _P]),
    'mrdis_softmax_mask_drop_bwd': (_I, [_P, _I, _P, _I, _P, _I, _L, _I, _P]),
    'mrdis_recon_err_workspace': (_Z, [_I, _L, _I]),
    'mrdis_recon_err_fwd': (_I, [_P, _I, _P, _I, _P, _P, _Z, _I, _L, _I, _I, _P]),
    'mrdis_recon_err_bwd': (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _L, _I, _I, _P]),
    'mrdis_recon_metrics_workspace': (_Z, [_I, _I]),
    'mrdis_recon_metrics': (_I, [_P, _I, _P, _I, _P, _P, _Z, _I, _I, _I, _P]),
    'mrdis_slice_gather': (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'mrdis_maxpool_fwd': (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    'mrdis_maxpool_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'mrdis_sumsq_workspace': (_Z, []),
    'mrdis_sumsq_finite': (_I, [_P, _L, _P, _P, _Z, _P]),
    'mrdis_adam_amsgrad_step': (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P, _P, _F, _F, _P, _P, _I, _P, _P, _I, _P]),
    'mrdis_conv3d_fwd': (_I, [_P, _I, _P, _P, _P, _I, _P, _I] + [_I] * 9 + [_P]),
    'mrdis_conv3d_bwd_data': (_I, [_P, _I, _P, _P, _I] + [_I] * 9 + [_P]),
    'mrdis_conv3d_bwd_weight_workspace': (_Z, [_I] * 9),
    'mrdis_conv3d_bwd_weight': (_I, [_P, _I, _P, _I, _P, _P, _P, _Z] + [_I] * 9 + [_P]),
    'mrdis_groupnorm_workspace': (_Z, [_I, _L, _I, _I]),
    'mrdis_groupnorm_relu_fwd': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _Z, _I, _L, _I, _I, _F, _I, _P]),
    'mrdis_groupnorm_relu_bwd': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _Z, _I, _L, _I, _I, _I, _P]),
    'mrdis_groupnorm_relu_bwd_add': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _Z, _I, _L, _I, _I, _I, _P]),
    'mrdis_upsample2x_add_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'mrdis_upsample2x_bwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


def load(path=None):
    """Load (once) and type the C-ABI library.  Raises MrdisLibraryError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise MrdisLibraryError(
            f'{path} not found: build it with `python __graft_entry__.py` / `make -C '
            f'{os.path.dirname(path)}`.  There is no CPU or eager fallback for the hot path.')
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the ABI drifted from include/mrdis.h
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _chk(rc, what):
    if rc != 0:
        raise MrdisError(f'{what}: {load().mrdis_strerror(rc).decode()} ({rc})')


def set_option(name, value):
    """process-wide kernel-selection switch (include/mrdis.h: "wino", "nt_mb", "debug_*")."""
    _chk(load().mrdis_set_option(name.encode(), int(value)), f'set_option({name})')


def get_option(name):
    return int(load().mrdis_get_option(name.encode()))


# every switch of csrc/mrdis_elem.hip OPT_DEFS (tests/test_abi.py checks that the library knows each name)
OPTION_NAMES = ('wino', 'nt_mb', 'wino_pipe', 'wino_u', 'wino4', 'wino4r', 'bconv4', 'split6', 'debug_no16', 'debug_nothin', 'debug_noc4', 'debug_nodma',
                'debug_no16_3d', 'debug_bilgen', 'debug_now16', 'debug_nopack', 'debug_mode', 'debug_bn', 'debug_kc', 'debug_bm', 'debug_c4_tw',
                'debug_wgsplit', 'debug_bn3', 'debug_kc3', 'c4_grid', 'debug_c4_blocks')


def options_snapshot():
    """{name: value} of every process-wide switch; `options_restore(snap)` puts them back (tests/conftest.py does so around every test)."""
    return {n: get_option(n) for n in OPTION_NAMES}


def options_restore(snap):
    for n, v in snap.items():
        if get_option(n) != v:
            set_option(n, v)


KERNEL_FAMILIES = WINO_FAMILIES = ('wino', 'wino_spade', 'wino2', 'wino2_spade', 'wino4', 'wino4_spade', 'wino4n', 'wino4r', 'wino_wgrad', 'wino_wgrad2', 'wino4_wgrad', 'bconv3', 'bconv3_spade', 'bconv4', 'bconv4_spade',
                                   'split6_c4', 'split6_c16', 'split6_wgrad16', 'split6_co4', 'split6_c3d', 'split6_w3d', 'split6_tap',
                                   'all')        # 'all': every kernel launch of the library (bench.py: library_launches_per_step)


def stream_fill(t, value=0.0):
    """store-only probe over a dense fp32 tensor (include/mrdis.h mrdis_stream_fill)"""
    assert t.dtype == torch.float32 and t.numel() % 4 == 0
    _chk(load().mrdis_stream_fill(t.data_ptr(), t.numel(), float(value), _stream()), 'stream_fill')
    return t


def dynamic_lds():
    """{kernel family: largest dynamic LDS bytes it was launched with in this process} (include/mrdis.h mrdis_dynamic_lds_table; template arguments folded)"""
    import re
    buf = _c.create_string_buffer(16384)
    load().mrdis_dynamic_lds_table(buf, 16384)
    out = {}
    for line in buf.value.decode().splitlines():
        expr, _, b = line.rpartition('=')
        if not expr or not b.isdigit():        # (the library only hands out whole lines; never let a diagnostic take the caller down)
            continue
        name = re.sub(r'<.*', '', expr.strip().lstrip('(')).strip()
        out[name] = max(out.get(name, 0), int(b))
    return out


def launch_counts(reset=False):
    """{family: launches since load / the last reset} of the Winograd, bf16 LDS-DMA and six-product (split6) kernel families (include/mrdis.h mrdis_launch_count)"""
    lib = load()
    out = {f: int(lib.mrdis_launch_count(f.encode())) for f in WINO_FAMILIES}
    if reset:
        lib.mrdis_launch_count_reset()
    return out


class option:
    """`with hip.option('wino', 0): ...` -- set a switch for a scope (tests, A/B timing)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = get_option(self.name)
        set_option(self.name, self.value)

    def __exit__(self, *exc):
        set_option(self.name, self.prev)


_DEV_INDEX = None


def _stream():
    """raw hipStream_t of torch's CURRENT stream on this process's device (one process per GPU).  `torch.cuda.current_stream()`
    costs ~9 us of Python per call (2,000 calls per step); the raw query is a C call."""
    global _DEV_INDEX
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_DEV_INDEX)


def _ptr(t):
    return None if t is None else t.data_ptr()


_WS = {}          # (device, raw stream) -> grow-only scratch buffer


def _ws(nbytes, device):
    """Scratch memory of one call.  Every kernel that uses it runs on the current stream, so consecutive calls can share ONE buffer per
    (device, stream): stream order keeps them apart, and no caller keeps workspace contents beyond its own launches.  (A fresh
    torch.empty per call was ~1,000 allocator round trips per training step.)"""
    n = max(int(nbytes), 16)
    key = (device, _stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < n:
        size = max(n, 1 << 20) if buf is None else max(n, 2 * buf.numel())
        buf = _WS[key] = torch.empty(size, dtype=torch.uint8, device=device)
    return buf


_WS_BYTES = {}    # (entry point, geometry) -> workspace bytes: the queries are pure functions of their arguments


def _ws_bytes(fn, *args):
    key = (fn.__name__, args)
    v = _WS_BYTES.get(key)
    if v is None:
        v = _WS_BYTES[key] = fn(*args)
    return v


# ---------------------------------------------------------------- NHWC views
def nhwc(t):
    """(tensor, ld) for a logical (N,C,H,W) fp32 or bf16 tensor whose memory is NHWC with pixel
    stride ld in elements (channels_last tensors and channel slices of them qualify as they are)."""
    assert t.dim() == 4 and (t.dtype is torch.float32 or t.dtype is torch.bfloat16), (t.shape, t.dtype)
    if t.is_contiguous(memory_format=torch.channels_last) and t.shape[1] > 1 and t.shape[3] > 1:
        return t, t.shape[1]                         # the common case: a dense channels_last tensor
    N, C, H, W = t.shape
    s = t.stride()
    ld = s[3] if W > 1 else (s[2] if H > 1 else (s[0] if N > 1 else C))
    ok = (ld >= C and (C == 1 or s[1] == 1) and (W == 1 or s[3] == ld) and
          (H == 1 or s[2] == W * ld) and (N == 1 or s[0] == H * W * ld))
    if not ok:
        t = t.contiguous(memory_format=torch.channels_last)
        if t.stride()[1] != 1 and C > 1:      # degenerate shapes: force a real NHWC copy
            t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        ld = C
    return t, ld


def empty_nhwc(N, C, H, W, device, dtype=torch.float32):
    return torch.empty((N, C, H, W), dtype=dtype, device=device, memory_format=torch.channels_last)


def _dt(*tensors):
    """MRDIS_DT_* storage code of a set of activation views (all fp32 or all bf16)."""
    kind = None
    for t in tensors:
        if t is None:
            continue
        d = t.dtype
        if kind is None:
            kind = d
        elif d is not kind:
            raise MrdisError(f'activation views must be all fp32 or all bf16, got {kind} and {d}')
    if kind is torch.float32:
        return 0
    if kind is torch.bfloat16:
        return 2
    raise MrdisError(f'activation views must be fp32 or bf16, got {kind}')


class _Mailbox:
    """Small host -> device transfers by a KERNEL that reads pinned host memory (mrdis_copy_bytes) instead of hipMemcpyAsync:
    a ring of pinned slots; a slot is rewritten only after the event recorded behind its last copy kernel has completed."""
    SLOT = 1 << 16          # bytes per slot = the largest tensor that goes through the mailbox
    NSLOT = 64

    def __init__(self):
        self.buf = torch.empty(self.SLOT * self.NSLOT, dtype=torch.uint8).pin_memory()
        self.events = [None] * self.NSLOT
        self.next = 0

    def send(self, t_cpu, device):
        nbytes = t_cpu.numel() * t_cpu.element_size()
        k = self.next
        self.next = (k + 1) % self.NSLOT
        ev = self.events[k]
        if ev is not None:
            ev.synchronize()                           # 64 transfers ago: long done unless the host is a whole ring ahead of the GPU
        slot = self.buf[k * self.SLOT:k * self.SLOT + nbytes]
        slot.copy_(t_cpu.contiguous().view(-1).view(torch.uint8))
        out = torch.empty(t_cpu.shape, dtype=t_cpu.dtype, device=device)
        _chk(load().mrdis_copy_bytes(slot.data_ptr(), _ptr(out), nbytes, _stream()), 'copy_bytes')
        if ev is None:
            ev = self.events[k] = torch.cuda.Event()
        ev.record()
        return out


_mailbox = None


def copy_bytes(src, dst, nbytes):
    """word-wise copy by a kernel on the current stream (include/mrdis.h mrdis_copy_bytes); src may be a PINNED host tensor"""
    assert nbytes % 4 == 0 and nbytes <= src.numel() * src.element_size() and nbytes <= dst.numel() * dst.element_size()
    _chk(load().mrdis_copy_bytes(src.data_ptr(), _ptr(dst), int(nbytes), _stream()), 'copy_bytes')


def to_device_small(t_cpu, device):
    """CPU tensor (<= 64 KB, element size 4 or 8) -> new device tensor, stream-ordered, without the copy engine; None if it does not qualify."""
    global _mailbox
    nbytes = t_cpu.numel() * t_cpu.element_size()
    if nbytes == 0 or nbytes > _Mailbox.SLOT or nbytes % 4 != 0 or t_cpu.dtype in (torch.bool,):
        return None
    if _mailbox is None:
        _mailbox = _Mailbox()
    return _mailbox.send(t_cpu, device)


class MixJob(_c.Structure):
    """csrc/mrdis_conv.hip `MixJob`: one CondConv2d layer (or one half of a fused gamma | beta pair) of the all-layers mixing launches."""
    _fields_ = [('W', _c.c_void_p), ('fcw', _c.c_void_p), ('fcb', _c.c_void_p), ('r', _c.c_void_p),
                ('tck', _c.c_void_p * 8), ('tkc', _c.c_void_p * 8), ('btck', _c.c_void_p * 8), ('btkc', _c.c_void_p * 8),
                ('dW', _c.c_void_p), ('dfcw', _c.c_void_p), ('dfcb', _c.c_void_p), ('part', _c.c_void_p),
                ('tap_tkc', _c.c_longlong),
                ('E', _c.c_int), ('Co', _c.c_int), ('Ci', _c.c_int), ('T', _c.c_int), ('ld_tck', _c.c_int), ('ld_dw', _c.c_int),
                ('block0', _c.c_int), ('nblk', _c.c_int), ('accumulate', _c.c_int), ('ci_pitch', _c.c_int)]


class WinoUJob(_c.Structure):
    """csrc/mrdis_wino2.hip `WinoUJob`: one (filter, role) of the Winograd filter-image launch (include/mrdis.h)."""
    _fields_ = [('w', _c.c_void_p), ('img', _c.c_void_p), ('R', _c.c_int), ('S', _c.c_int), ('flip', _c.c_int), ('spadeC', _c.c_int),
                ('block0', _c.c_int), ('nblk', _c.c_int), ('fmt', _c.c_int), ('pad_', _c.c_int)]


def wino_u_format(R, S, spadeC=0):
    """2: the F(2x2,3x3) image (csrc/mrdis_wino2.hip), 4: the F(4x4,3x3) image (csrc/mrdis_wino4.hip) -- a function of the filter's shape and
    the option 'wino4' alone, so the image built once per step fits every call of the layer."""
    return int(load().mrdis_wino_u_format(R, S, spadeC))


def wino_u_table(jobs, device):
    lib = load()
    nb = _c.sizeof(WinoUJob)
    if nb != lib.mrdis_wino_u_job_bytes():
        raise MrdisError(f'WinoUJob layout mismatch: binding {nb} bytes, library {lib.mrdis_wino_u_job_bytes()}')
    for j in jobs:
        if not j.fmt:                              # a caller may pin the format (MixPlan records the one it sized the image for)
            j.fmt = wino_u_format(j.R, j.S, j.spadeC)
    arr = (WinoUJob * len(jobs))(*jobs)
    host = torch.frombuffer(bytearray(_c.string_at(_c.addressof(arr), nb * len(jobs))), dtype=torch.uint8)
    return host.to(device)


def wino_u_image_floats(R, S, spadeC=0, fmt=None):
    """floats of the filter image in format `fmt` (default: the format an image built NOW gets); -1 if that shape never has the format"""
    if fmt is None:
        return int(load().mrdis_wino_u_image_floats(R, S, spadeC))
    return int(load().mrdis_wino_u_image_floats_fmt(R, S, spadeC, fmt))


S6_IMAGE_FMT = 6


def s6_filter_image_bytes(taps, Cred, Cout):
    return int(load().mrdis_s6_filter_image_bytes(int(taps), int(Cred), int(Cout)))


def s6_filter_image(w):
    """w [taps][Cred][Cout] fp32 (w_tck for the forward pass, w_tkc for the data gradient) -> its six-product image (mrdis_s6conv.hip), a float32-typed
    tensor that travels in the auxiliary-filter slot of conv2d_fwd / conv2d_bwd_data (attribute mrdis_fmt = 6); None where the layer has no such image."""
    T, R, S = w.shape
    nb = s6_filter_image_bytes(T, R, S)
    if nb == 0 or nb >= 2 ** 31 or w.dtype is not torch.float32:
        return None
    w = w.contiguous()
    img = torch.empty(nb // 4, dtype=torch.float32, device=w.device)
    _chk(load().mrdis_s6_filter_image(_ptr(w), T, R, S, _ptr(img), nb, _stream()), 's6_filter_image')
    img.mrdis_fmt = S6_IMAGE_FMT
    return img


def wino_image_fmt(img, R, S, spadeC=0, taps=0):
    """The format a Winograd filter image was BUILT in: it travels with the image (attribute `mrdis_fmt`, set by whoever built it) and goes to
    the library beside the pointer -- never re-derived from the current value of the option 'wino4', which may have changed since.  Images
    without the attribute (tests, tools) are recognised by their size, which differs between the formats of one filter shape."""
    fmt = getattr(img, 'mrdis_fmt', None)
    n = img.numel()
    if fmt == S6_IMAGE_FMT or (fmt is None and taps and s6_filter_image_bytes(taps, R, S) == 4 * n):
        if s6_filter_image_bytes(taps, R, S) != 4 * n:
            raise MrdisError(f'six-product image of {4 * n} bytes, a [{taps}][{R}][{S}] filter needs {s6_filter_image_bytes(taps, R, S)}')
        return S6_IMAGE_FMT
    if fmt is None:
        hits = [f for f in (2, 4, 5) if wino_u_image_floats(R, S, spadeC, f) == n]
        if len(hits) != 1:
            raise MrdisError(f'Winograd image of {n} floats fits no single format of a [9][{R}][{S}] filter (spadeC {spadeC}): candidates {hits}')
        fmt = hits[0]
    elif wino_u_image_floats(R, S, spadeC, fmt) != n:
        raise MrdisError(f'Winograd image tagged format {fmt} has {n} floats, a [9][{R}][{S}] filter (spadeC {spadeC}) needs {wino_u_image_floats(R, S, spadeC, fmt)}')
    return fmt


def wino_u_job_blocks(R, S, spadeC=0):
    return int(load().mrdis_wino_u_job_blocks(R, S, spadeC))


def wino_u_jobs(table, njobs, total_blocks):
    _chk(load().mrdis_wino_u_jobs(_ptr(table), njobs, total_blocks, _stream()), 'wino_u_jobs')


def mix_job_table(jobs, device):
    """list of MixJob -> device tensor holding the table (checked against the library's struct size)."""
    lib = load()
    nb = _c.sizeof(MixJob)
    if nb != lib.mrdis_mix_job_bytes():
        raise MrdisError(f'MixJob layout mismatch: binding {nb} bytes, library {lib.mrdis_mix_job_bytes()}')
    arr = (MixJob * len(jobs))(*jobs)
    host = torch.frombuffer(bytearray(_c.string_at(_c.addressof(arr), nb * len(jobs))), dtype=torch.uint8)
    return host.to(device)


def mix_job_blocks(Co, Ci, T):
    return int(load().mrdis_mix_job_blocks(Co, Ci, T))


def mix_jobs_fwd(table, njobs, total_blocks, types):
    M, emb = types.shape
    _chk(load().mrdis_mix_jobs_fwd(_ptr(table), njobs, total_blocks, _ptr(types), emb, M, _stream()), 'mix_jobs_fwd')


def mix_jobs_bwd(table, njobs, total_blocks, dw_table, types):
    M, emb = types.shape
    _chk(load().mrdis_mix_jobs_bwd(_ptr(table), njobs, total_blocks, _ptr(dw_table), _ptr(types), emb, M, _stream()), 'mix_jobs_bwd')


def cast_view(x, dtype, channels=None):
    """NHWC view -> new NHWC tensor of storage type `dtype` (fp32 / bf16, round to nearest even) with `channels` channels
    (default: unchanged): the first min(C, channels) are copied, a wider result is zero-padded."""
    lib = load()
    x, ldx = nhwc(x)
    N, C, H, W = x.shape
    Cd = C if channels is None else int(channels)
    if x.dtype == dtype and Cd == C:
        return x
    y = empty_nhwc(N, Cd, H, W, x.device, dtype)
    _chk(lib.mrdis_cast_view(_ptr(x), ldx, _dt(x), C, _ptr(y), Cd, _dt(y), Cd, N * H * W, _stream()), 'cast_view')
    return y


def bconv_eligible(c_reduce, c_out):
    """geometry the bf16 MFMA convolution kernels cover (csrc/mrdis_bf16.hip): reduction axis % 16, >= 16 outputs, % 4."""
    return c_reduce % 16 == 0 and c_out % 4 == 0 and c_out >= 16


# ---------------------------------------------------------------- expert mixing
def mix_experts_fwd(W, r):
    """W (E,Co,Ci,kh,kw), r (E) -> (w_tck [T,Ci,Co], w_tkc [T,Co,Ci])."""
    lib = load()
    E, Co, Ci, kh, kw = W.shape
    T = kh * kw
    W = W.contiguous(); r = r.contiguous()
    w_tck = torch.empty((T, Ci, Co), dtype=torch.float32, device=W.device)
    w_tkc = torch.empty((T, Co, Ci), dtype=torch.float32, device=W.device)
    _chk(lib.mrdis_mix_experts_fwd(_ptr(W), _ptr(r), _ptr(w_tck), _ptr(w_tkc), E, Co, Ci, T, _stream()), 'mix_experts_fwd')
    return w_tck, w_tkc


def mix_experts_bwd(dw_tck, W, r):
    lib = load()
    E, Co, Ci, kh, kw = W.shape
    T = kh * kw
    W = W.contiguous(); r = r.contiguous(); dw_tck = dw_tck.contiguous()
    dW = torch.empty_like(W)
    dr = torch.zeros(E, dtype=torch.float32, device=W.device)
    nb = _ws_bytes(lib.mrdis_mix_experts_bwd_workspace, E, Co, Ci, T)
    ws = _ws(nb, W.device)
    _chk(lib.mrdis_mix_experts_bwd(_ptr(dw_tck), _ptr(W), _ptr(r), _ptr(dW), _ptr(dr), _ptr(ws), nb, E, Co, Ci, T, _stream()),
         'mix_experts_bwd')
    return dW, dr


def mix_experts_routed_fwd(W, fcw, fcb, t_row):
    """W (E,Co,Ci,kh,kw), routing Linear (fcw (E,emb), fcb (E)), t_row (1,emb) -> (w_tck, w_tkc, r)."""
    lib = load()
    E, Co, Ci, kh, kw = W.shape
    T = kh * kw
    W = W.contiguous(); fcw = fcw.contiguous(); fcb = fcb.contiguous(); t_row = t_row.contiguous()
    emb = fcw.shape[1]
    w_tck = torch.empty((T, Ci, Co), dtype=torch.float32, device=W.device)
    w_tkc = torch.empty((T, Co, Ci), dtype=torch.float32, device=W.device)
    r = torch.empty(E, dtype=torch.float32, device=W.device)
    _chk(lib.mrdis_mix_experts_routed_fwd(_ptr(W), _ptr(fcw), _ptr(fcb), _ptr(t_row), emb, _ptr(r), _ptr(w_tck), _ptr(w_tkc),
                                          E, Co, Ci, T, _stream()), 'mix_experts_routed_fwd')
    return w_tck, w_tkc, r


def mix_experts_routed_bwd(dw_tck, W, r, t_row, emb):
    lib = load()
    E, Co, Ci, kh, kw = W.shape
    T = kh * kw
    W = W.contiguous(); dw_tck = dw_tck.contiguous(); t_row = t_row.contiguous()
    dW = torch.empty_like(W)
    dfcw = torch.empty((E, emb), dtype=torch.float32, device=W.device)
    dfcb = torch.empty(E, dtype=torch.float32, device=W.device)
    nb = _ws_bytes(lib.mrdis_mix_experts_bwd_workspace, E, Co, Ci, T)
    ws = _ws(nb, W.device)
    _chk(lib.mrdis_mix_experts_routed_bwd(_ptr(dw_tck), _ptr(W), _ptr(r), _ptr(t_row), emb, _ptr(dW), _ptr(dfcw), _ptr(dfcb),
                                          _ptr(ws), nb, E, Co, Ci, T, _stream()), 'mix_experts_routed_bwd')
    return dW, dfcw, dfcb


def mix_experts_routed_multi_fwd(W, fcw, fcb, types, want_bf16=False, into=None):
    """all M type rows at once: -> ([w_tck_m], [w_tkc_m], r (M,E)); want_bf16: + ([bf16(w_tck_m)], [bf16(w_tkc_m)]) from the same launch.
    into = (tck, tkc, btck, btkc, col0, co_total): write into column block [col0, col0 + Co) of existing WIDER filters (lists of M tensors
    (T, Ci, co_total) / (T, co_total, Ci); btck / btkc None or the bf16 twins) -- the halves of a fused gamma | beta filter; -> r only."""
    lib = load()
    E, Co, Ci, kh, kw = W.shape
    T = kh * kw
    W = W.contiguous(); fcw = fcw.contiguous(); fcb = fcb.contiguous(); types = types.contiguous()
    M, emb = types.shape
    r = torch.empty((M, E), dtype=torch.float32, device=W.device)
    if into is not None:
        tck, tkc, btck, btkc, col0, cot = into
        a = (_c.c_void_p * M)(*[t.data_ptr() + 4 * col0 for t in tck]); b = (_c.c_void_p * M)(*[t.data_ptr() + 4 * col0 * Ci for t in tkc])
        ba = bb = None
        if btck is not None:
            ba = (_c.c_void_p * M)(*[t.data_ptr() + 2 * col0 for t in btck]); bb = (_c.c_void_p * M)(*[t.data_ptr() + 2 * col0 * Ci for t in btkc])
        _chk(lib.mrdis_mix_experts_routed_multi_fwd(_ptr(W), _ptr(fcw), _ptr(fcb), _ptr(types), emb, M, _ptr(r), a, b, ba, bb, cot, cot * Ci,
                                                    E, Co, Ci, T, _stream()), 'mix_experts_routed_multi_fwd')
        return r
    tck = [torch.empty((T, Ci, Co), dtype=torch.float32, device=W.device) for _ in range(M)]
    tkc = [torch.empty((T, Co, Ci), dtype=torch.float32, device=W.device) for _ in range(M)]
    a = (_c.c_void_p * M)(*[t.data_ptr() for t in tck]); b = (_c.c_void_p * M)(*[t.data_ptr() for t in tkc])
    ba = bb = None
    if want_bf16:
        btck = [torch.empty((T, Ci, Co), dtype=torch.bfloat16, device=W.device) for _ in range(M)]
        btkc = [torch.empty((T, Co, Ci), dtype=torch.bfloat16, device=W.device) for _ in range(M)]
        ba = (_c.c_void_p * M)(*[t.data_ptr() for t in btck]); bb = (_c.c_void_p * M)(*[t.data_ptr() for t in btkc])
    _chk(lib.mrdis_mix_experts_routed_multi_fwd(_ptr(W), _ptr(fcw), _ptr(fcb), _ptr(types), emb, M, _ptr(r), a, b, ba, bb, 0, 0, E, Co, Ci, T, _stream()),
         'mix_experts_routed_multi_fwd')
    if want_bf16:
        return tck, tkc, r, btck, btkc
    return tck, tkc, r


def mix_experts_routed_multi_bwd(dw_list, W, r, types, sinks=None, col0=0, ld=0):
    """dw_list: M tensors (T,Ci,Co) or None -> (dW, dfcw, dfcb) summed over the types.  sinks = (gW, gfcw, gfcb): contiguous fp32
    gradient buffers the three results are ADDED to in-kernel (returned as they are).  col0 / ld: the gradients are the column block
    [col0, col0 + Co) of contiguous (T, Ci, ld) tensors (one half of a fused gamma | beta filter gradient)."""
    lib = load()
    E, Co, Ci, kh, kw = W.shape
    T = kh * kw
    W = W.contiguous(); types = types.contiguous()
    M, emb = types.shape
    dw_list = [None if g is None else g.contiguous() for g in dw_list]
    a = (_c.c_void_p * M)(*[None if g is None else g.data_ptr() + 4 * col0 for g in dw_list])
    if sinks is not None:
        dW, dfcw, dfcb = sinks
    else:
        dW = torch.empty_like(W)
        dfcw = torch.empty((E, emb), dtype=torch.float32, device=W.device)
        dfcb = torch.empty(E, dtype=torch.float32, device=W.device)
    nb = _ws_bytes(lib.mrdis_mix_experts_routed_multi_bwd_workspace, M, E, Co, Ci, T)
    ws = _ws(nb, W.device)
    _chk(lib.mrdis_mix_experts_routed_multi_bwd(a, _ptr(W), _ptr(r), _ptr(types), emb, M, _ptr(dW), _ptr(dfcw), _ptr(dfcb),
                                                1 if sinks is not None else 0, ld, _ptr(ws), nb, E, Co, Ci, T, _stream()), 'mix_experts_routed_multi_bwd')
    return dW, dfcw, dfcb


# ---------------------------------------------------------------- convolution
def conv_out_hw(H, W, kh, kw, stride, pad):
    return (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1


DT_F32, DT_F32_BF16M, DT_BF16 = 0, 1, 2          # include/mrdis.h MRDIS_DT_*
DT_DW_PAD16 = 0x100                            # conv2d_bwd_weight(pad16=True): dw in the stored 16-row / 16-column shape of the filter (mrdis.h)
DT_XBF16_YF32, DT_XF32_YBF16 = 3, 4              # mixed storage at the ends of a bf16 stretch (layer input x / output y)


def _dt_xy(x, y):
    """storage code of a layer whose input-side view is x (or dx) and output-side view y (or dy)"""
    if x.dtype is y.dtype:
        return _dt(x)
    if x.dtype is torch.bfloat16 and y.dtype is torch.float32:
        return DT_XBF16_YF32
    if x.dtype is torch.float32 and y.dtype is torch.bfloat16:
        return DT_XF32_YBF16
    raise MrdisError(f'activation views must be fp32 or bf16, got {x.dtype} and {y.dtype}')


def cast_bf16(t):
    """fp32 tensor -> bf16 copy (round to nearest even) through the library's cast kernel."""
    lib = load()
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    _chk(lib.mrdis_cast_bf16(_ptr(t), _ptr(out), t.numel(), _stream()), 'cast_bf16')
    return out


def conv2d_fwd(x, w_tck, bias, kh, kw, stride, pad, lrelu=False, out=None, w_bf16=None, out_dtype=None, may_decline=False, w_wino=None):
    """w_bf16: bf16 [T][Co][Ci] copy of the filter (reduction axis contiguous) -> bf16 MFMA operands, fp32 accumulate
    (MRDIS_DT_F32_BF16M) where the geometry allows; None -> exact fp32.
    Mixed storage (out / out_dtype differ from x.dtype): the 1x1 head reads bf16 and writes fp32 (MRDIS_DT_XBF16_YF32); the 3x3
    4 -> C si_layers read the fp32 anatomy map and write bf16 (MRDIS_DT_XF32_YBF16, w_tck in the 16-row layout of the mixing launch).
    may_decline: return None instead of raising when the library has no kernel for a mixed-storage geometry."""
    lib = load()
    x, ldx = nhwc(x)
    N, Ci, H, W = x.shape
    T, Ci2, Co = w_tck.shape
    Ho, Wo = conv_out_hw(H, W, kh, kw, stride, pad)
    if out is None:
        out = empty_nhwc(N, Co, Ho, Wo, x.device, out_dtype or x.dtype)
    y, ldy = nhwc(out)
    assert y.data_ptr() == out.data_ptr(), 'conv2d_fwd: `out` must already be an NHWC view'
    mixed = _dt_xy(x, y)
    if mixed == DT_XBF16_YF32 and kh == 3 and Co == 16 and out.shape[1] == 4:
        Co = 4                       # the C -> 4 layer with its filter in the column-padded [9][Ci][16] layout (ana_dec.output under bf16 storage)
    assert T == kh * kw and (Ci2 == Ci or (mixed == DT_XF32_YBF16 and Ci2 == max(Ci, 16))), (w_tck.shape, x.shape, kh, kw)
    if mixed in (DT_XBF16_YF32, DT_XF32_YBF16):
        rc = lib.mrdis_conv2d_fwd(_ptr(x), ldx, _ptr(w_tck), None, _ptr(bias), _ptr(y), ldy, N, H, W, Ci, Co, kh, kw, stride, pad,
                                  1 if lrelu else 0, mixed, None, 0, _stream())
        if rc == -2 and may_decline:
            return None
        _chk(rc, 'conv2d_fwd (mixed storage)')
        return out
    if w_bf16 is not None and w_bf16.dtype is torch.float32:      # the auxiliary-filter slot carries the Winograd image on the fp32 path
        w_wino, w_bf16 = w_bf16, None
    if w_bf16 is not None:
        assert w_bf16.dtype == torch.bfloat16 and tuple(w_bf16.shape) == (T, Co, Ci) and w_bf16.is_contiguous()
    wfmt = 0
    if w_wino is not None:
        assert w_wino.dtype == torch.float32
        wfmt = wino_image_fmt(w_wino, Ci, Co, taps=T)
        assert wfmt == S6_IMAGE_FMT or (kh == 3 and kw == 3)
    dt = DT_BF16 if mixed == DT_BF16 else (DT_F32 if w_bf16 is None else DT_F32_BF16M)     # bf16 views: bf16 kernels only
    rc = lib.mrdis_conv2d_fwd(_ptr(x), ldx, _ptr(w_tck), _ptr(w_bf16), _ptr(bias), _ptr(y), ldy, N, H, W, Ci, Co, kh, kw, stride, pad,
                              1 if lrelu else 0, dt, _ptr(w_wino) if dt == DT_F32 else None, wfmt, _stream())
    if rc == -2 and dt == DT_BF16:
        # a tile geometry the bf16 kernel cannot stage (or a view it cannot address): the fp32 kernel between two view casts
        y32 = conv2d_fwd(cast_view(x, torch.float32), w_tck, bias, kh, kw, stride, pad, lrelu)
        lib.mrdis_cast_view(_ptr(y32), Co, DT_F32, Co, _ptr(y), ldy, DT_BF16, Co, N * Ho * Wo, _stream())
        return out
    _chk(rc, 'conv2d_fwd')
    return out


def conv2d_bwd_data(dy, w_tkc, in_hw, kh, kw, stride, pad, w_bf16=None, out=None, w_wino=None, may_decline=False):
    """w_bf16: bf16 [T][Ci][Co] copy (the data gradient reduces over Co).  out: a dense NHWC (N, Ci, H, W) view to write into.
    Mixed storage (fp32 dy, bf16 out: the 1x1 head; the 3x3 C <- 4 layer with the filter in its 16-row layout [9][16][Ci]; bf16 dy, fp32
    out: the si_layers' 4 <- C data gradient with the filter as [9][Co][16]) has only its dedicated kernels: may_decline returns None
    instead of raising outside them."""
    lib = load()
    dy, lddy = nhwc(dy)
    N, Co, Ho, Wo = dy.shape
    T, Co2, Ci = w_tkc.shape
    H, W = in_hw
    mixed = out is not None and out.dtype != dy.dtype
    if mixed and kh == 3 and Ci == 16 and out.shape[1] == 4:
        Ci = 4                       # the si_layers' data gradient with the filter in the column-padded [9][Co][16] layout (bf16 dy -> fp32 dx: MRDIS_DT_XF32_YBF16)
    assert (Co2 == Co or (mixed and Co2 == 16 and Co == 4 and kh == 3)) and conv_out_hw(H, W, kh, kw, stride, pad) == (Ho, Wo)
    if out is None:
        dx = empty_nhwc(N, Ci, H, W, dy.device, dy.dtype); ldo = Ci
    else:
        dx, ldo = nhwc(out)          # a channel slice of a wider NHWC buffer is fine (ldo > Ci)
        assert dx.data_ptr() == out.data_ptr() and ldo >= Ci and tuple(out.shape) == (N, Ci, H, W)
        if mixed:                    # bf16 storage: dy fp32 -> dx bf16 (MRDIS_DT_XBF16_YF32), or dy bf16 -> dx fp32 (MRDIS_DT_XF32_YBF16)
            rc = lib.mrdis_conv2d_bwd_data(_ptr(dy), lddy, _ptr(w_tkc), None, _ptr(dx), ldo, N, H, W, Ci, Co, kh, kw, stride, pad, _dt_xy(dx, dy), None, 0, _stream())
            if rc == -2 and may_decline:
                return None
            _chk(rc, 'conv2d_bwd_data (mixed storage)')
            return dx
    if w_bf16 is not None and w_bf16.dtype is torch.float32:
        w_wino, w_bf16 = w_bf16, None
    if w_bf16 is not None:
        assert w_bf16.dtype == torch.bfloat16 and tuple(w_bf16.shape) == (T, Ci, Co) and w_bf16.is_contiguous()
    wfmt = 0
    if w_wino is not None:
        assert w_wino.dtype == torch.float32
        wfmt = wino_image_fmt(w_wino, Co, Ci, taps=T)
        assert wfmt == S6_IMAGE_FMT or (kh == 3 and kw == 3)
    dt = DT_BF16 if _dt(dy) == DT_BF16 else (DT_F32 if w_bf16 is None else DT_F32_BF16M)
    rc = lib.mrdis_conv2d_bwd_data(_ptr(dy), lddy, _ptr(w_tkc), _ptr(w_bf16), _ptr(dx), ldo, N, H, W, Ci, Co, kh, kw, stride, pad, dt,
                                   _ptr(w_wino) if dt == DT_F32 else None, wfmt, _stream())
    if rc == -2 and dt == DT_BF16:
        # a geometry outside the bf16 kernels (e.g. a reduction axis that is not a multiple of 16): fp32 kernel between two view casts
        res = cast_view(conv2d_bwd_data(cast_view(dy, torch.float32), w_tkc, in_hw, kh, kw, stride, pad), torch.bfloat16)
        if out is None:
            return res
        out.copy_(res)
        return out
    _chk(rc, 'conv2d_bwd_data')
    return dx


fallbacks = collections.Counter()      # routes that left the bf16 kernels for fp32 kernels between view casts (tests read it)


def conv2d_bwd_weight(x, dy, kh, kw, stride, pad, need_bias=True, bias_sink=None, dtype=DT_F32, may_decline=False, pad16=False):
    """-> (dw_tck, dbias).  bias_sink: a (Co,) buffer the bias gradient is ADDED to in the reduce launch
    (then dbias is returned as None).  dtype DT_F32_BF16M: bf16 MFMA operands where the geometry allows.
    may_decline: mixed-storage views (x fp32 / dy bf16: the si_layers; x bf16 / dy fp32: the 1x1 head) only have their dedicated
    kernels -- return None instead of raising when the geometry is outside them (the caller then casts a view).
    pad16 (mixed-storage views with a four-channel side only): dw_tck comes back in the stored shape of a filter of the mixing launch,
    (T, 16, Co) for a 4 -> C layer / (T, Ci, 16) for a C -> 4 layer, zeros beyond the layer's own rows / columns (MRDIS_DT_DW_PAD16)."""
    lib = load()
    x, ldx = nhwc(x)
    dy, lddy = nhwc(dy)
    N, Ci, H, W = x.shape
    Co = dy.shape[1]
    if pad16:
        assert x.dtype is not dy.dtype and 4 in (Ci, Co), 'pad16: the four-channel mixed-storage kernels only'
    dw = torch.empty((kh * kw, 16 if (pad16 and Ci == 4) else Ci, 16 if (pad16 and Co == 4 and Ci != 4) else Co), dtype=torch.float32, device=x.device)
    db = torch.empty(Co, dtype=torch.float32, device=x.device) if (need_bias and bias_sink is None) else None
    nb = _ws_bytes(lib.mrdis_conv2d_bwd_weight_workspace, N, H, W, Ci, Co, kh, kw, stride, pad)
    if nb == 0:
        raise MrdisError('conv2d_bwd_weight: unsupported geometry')
    ws = _ws(nb, x.device)
    sink = bias_sink if (need_bias and bias_sink is not None) else None
    if x.dtype is not dy.dtype:      # bf16 storage: the 1x1 head (x bf16, dy fp32: MRDIS_DT_XBF16_YF32), the si_layers (x fp32, dy bf16: MRDIS_DT_XF32_YBF16)
        rc = lib.mrdis_conv2d_bwd_weight(_ptr(x), ldx, _ptr(dy), lddy, _ptr(dw), _ptr(sink if sink is not None else db), _ptr(ws), nb,
                                         N, H, W, Ci, Co, kh, kw, stride, pad, 1 if sink is not None else 0, _dt_xy(x, dy) | (DT_DW_PAD16 if pad16 else 0), _stream())
        if rc == -2 and may_decline:
            return None
        _chk(rc, 'conv2d_bwd_weight (mixed storage)')
        return dw, db
    if _dt(x, dy) == DT_BF16:
        rc = lib.mrdis_conv2d_bwd_weight(_ptr(x), ldx, _ptr(dy), lddy, _ptr(dw), _ptr(sink if sink is not None else db), _ptr(ws), nb,
                                         N, H, W, Ci, Co, kh, kw, stride, pad, 1 if sink is not None else 0, DT_BF16, _stream())
        if rc == -2:      # tiny maps / shapes outside the bf16 kernels' domain: the fp32 weight-gradient kernels on fp32 copies of the two views
            fallbacks['conv2d_bwd_weight_bf16'] += 1
            return conv2d_bwd_weight(cast_view(x, torch.float32), cast_view(dy, torch.float32), kh, kw, stride, pad, need_bias, bias_sink, DT_F32)
        _chk(rc, 'conv2d_bwd_weight')
        return dw, db
    _chk(lib.mrdis_conv2d_bwd_weight(_ptr(x), ldx, _ptr(dy), lddy, _ptr(dw), _ptr(sink if sink is not None else db), _ptr(ws), nb,
                                     N, H, W, Ci, Co, kh, kw, stride, pad, 1 if sink is not None else 0, DT_BF16 if _dt(x, dy) == DT_BF16 else int(dtype), _stream()), 'conv2d_bwd_weight')
    return dw, db


def lrelu_bwd(dy, y, slope=0.2):
    lib = load()
    dy, lddy = nhwc(dy); y, ldy = nhwc(y)
    N, C, H, W = y.shape
    dx = empty_nhwc(N, C, H, W, y.device, y.dtype)
    _chk(lib.mrdis_lrelu_bwd(_ptr(dy), lddy, _ptr(y), ldy, _ptr(dx), C, N * H * W, C, slope, _dt(dy, y), _stream()), 'lrelu_bwd')
    return dx


# ---------------------------------------------------------------- norms
def bn_train_fwd(x, gamma, beta, running_mean, running_var, eps, momentum, out=None, groups=1):
    """groups = G: the batch holds G equal sample blocks that the reference normalises in G separate calls of this layer
    (statistics per block: mean / rstd are (G * C); running statistics updated block by block)."""
    lib = load()
    x, ldx = nhwc(x)
    N, C, H, W = x.shape
    assert N % groups == 0
    P = (N // groups) * H * W
    if out is None:
        out = empty_nhwc(N, C, H, W, x.device, x.dtype)
    y, ldy = nhwc(out)
    assert y.data_ptr() == out.data_ptr()
    mean = torch.empty(groups * C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(groups * C, dtype=torch.float32, device=x.device)
    nb = groups * _ws_bytes(lib.mrdis_norm_workspace, 1, P, C)
    ws = _ws(nb, x.device)
    _chk(lib.mrdis_bn_train_fwd(_ptr(x), ldx, _ptr(y), ldy, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                _ptr(mean), _ptr(rstd), _ptr(ws), nb, P, C, eps, momentum, groups, _dt(x, y), _stream()), 'bn_train_fwd')
    return out, mean, rstd


def bn_eval_fwd(x, gamma, beta, running_mean, running_var, eps):
    lib = load()
    x, ldx = nhwc(x)
    N, C, H, W = x.shape
    y = empty_nhwc(N, C, H, W, x.device, x.dtype)
    _chk(lib.mrdis_bn_eval_fwd(_ptr(x), ldx, _ptr(y), C, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                               N * H * W, C, eps, _dt(x), _stream()), 'bn_eval_fwd')
    return y


def bn_train_bwd(dy, x, gamma, mean, rstd, sink=None, groups=1):
    """-> (dx, dgamma, dbeta).  sink = (acc_dgamma, acc_dbeta): buffers this call's parameter gradients are also added to.
    groups: as bn_train_fwd (dgamma / dbeta are then summed over the groups here)."""
    lib = load()
    dy, lddy = nhwc(dy); x, ldx = nhwc(x)
    N, C, H, W = x.shape
    P = (N // groups) * H * W
    dx = empty_nhwc(N, C, H, W, x.device, x.dtype)
    dg = torch.empty(groups * C, dtype=torch.float32, device=x.device)
    db = torch.empty(groups * C, dtype=torch.float32, device=x.device)
    nb = groups * _ws_bytes(lib.mrdis_norm_workspace, 1, P, C)
    ws = _ws(nb, x.device)
    ag, ab = sink if sink is not None else (None, None)
    _chk(lib.mrdis_bn_train_bwd(_ptr(dy), lddy, _ptr(x), ldx, _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dx), C, _ptr(dg), _ptr(db),
                                _ptr(ag), _ptr(ab), _ptr(ws), nb, P, C, groups, _dt(dy, x), _stream()), 'bn_train_bwd')
    if groups > 1 and sink is None:
        dg, db = dg.view(groups, C).sum(0), db.view(groups, C).sum(0)
    return dx, dg, db


def instnorm_spade_fwd(z, gamma, beta, eps=1e-5, stats=None):
    """stats = (mean, rstd) of z already computed (bilinear_up2_stats): the statistics pass is skipped"""
    lib = load()
    z, ldz = nhwc(z); gamma, ldg = nhwc(gamma); beta, ldb = nhwc(beta)
    N, C, H, W = z.shape
    out = empty_nhwc(N, C, H, W, z.device, z.dtype)
    if stats is not None:
        mean, rstd = stats
        assert mean.numel() == N * C and rstd.numel() == N * C and mean.is_contiguous() and rstd.is_contiguous()
        _chk(lib.mrdis_instnorm_spade_fwd(_ptr(z), ldz, _ptr(gamma), ldg, _ptr(beta), ldb, _ptr(out), C, _ptr(mean), _ptr(rstd),
                                          None, 0, N, H * W, C, eps, _dt(z, gamma, beta), _stream()), 'instnorm_spade_fwd')
        return out, mean, rstd
    mean = torch.empty(N * C, dtype=torch.float32, device=z.device)
    rstd = torch.empty(N * C, dtype=torch.float32, device=z.device)
    nb = _ws_bytes(lib.mrdis_norm_workspace, N, H * W, C)
    ws = _ws(nb, z.device)
    _chk(lib.mrdis_instnorm_spade_fwd(_ptr(z), ldz, _ptr(gamma), ldg, _ptr(beta), ldb, _ptr(out), C, _ptr(mean), _ptr(rstd),
                                      _ptr(ws), nb, N, H * W, C, eps, _dt(z, gamma, beta), _stream()), 'instnorm_spade_fwd')
    return out, mean, rstd


def gb_spade_fwd(si_out, w_tck, bias, z, eps=1e-5, out=None, w_bf16=None, stats_ready=False, w_wino=None):
    """fused gamma | beta convolution + InstanceNorm modulation (mrdis_conv2d_fwd_spade): -> (mix, gamma, mean, rstd), or None where the
    fused kernel does not apply (the caller then runs conv2d_fwd + instnorm_spade_fwd).  fp32 views, or bf16 views with w_bf16 = the bf16
    [9][2C][Ci] filter.  out = (mix, gamma, mean, rstd) dense views to fill."""
    lib = load()
    if si_out.dtype is not z.dtype or (z.dtype is torch.bfloat16 and w_bf16 is None):
        return None
    x, ldx = nhwc(si_out); z, ldz = nhwc(z)
    N, C, H, W = z.shape
    Ci = x.shape[1]
    if tuple(w_tck.shape) != (9, Ci, 2 * C) or x.shape[0] != N or x.shape[2] != H or x.shape[3] != W:
        return None
    dt = _dt(x, z)
    if out is None:
        mean = torch.empty(N * C, dtype=torch.float32, device=z.device)
        rstd = torch.empty(N * C, dtype=torch.float32, device=z.device)
        mix = empty_nhwc(N, C, H, W, z.device, z.dtype)
        gamma = empty_nhwc(N, C, H, W, z.device, z.dtype)
    else:
        mix, gamma, mean, rstd = out
    nb = _ws_bytes(lib.mrdis_norm_workspace, N, H * W, C)
    ws = _ws(nb, z.device)
    st = _stream()
    # the statistics first (stream order); if the fused kernel then declines, they are simply recomputed by the two-step path
    # (stats_ready: `out`'s mean / rstd already hold them -- the x2 resize that produced z took them on the way, bilinear_up2_stats)
    if not stats_ready:
        _chk(lib.mrdis_instnorm_stats(_ptr(z), ldz, _ptr(mean), _ptr(rstd), _ptr(ws), nb, N, H * W, C, eps, dt, st), 'instnorm_stats')
    wfmt = 0
    if w_wino is not None:
        assert w_wino.dtype == torch.float32
        wfmt = wino_image_fmt(w_wino, Ci, 2 * C, C)
    rc = lib.mrdis_conv2d_fwd_spade(_ptr(x), ldx, _ptr(w_tck), _ptr(w_bf16), _ptr(bias), _ptr(z), ldz, _ptr(mean), _ptr(rstd), _ptr(mix), C, _ptr(gamma), C,
                                    N, H, W, Ci, C, dt, _ptr(w_wino) if dt == DT_F32 else None, wfmt, st)
    if rc == -2:
        return None
    _chk(rc, 'conv2d_fwd_spade')
    return mix, gamma, mean, rstd


def gb_slot(t):
    """the (N, 2C, H, W) NHWC buffer whose channels [C, 2C) are exactly the NCHW-shaped tensor `t`, or None"""
    base = t._base
    if base is None or base.dim() != 4 or t.dim() != 4 or not getattr(base, '_mrdis_gb_private', False):
        return None      # only buffers a grouped convolution allocated for this purpose: any other upper-half channel slice (a torch.cat adjoint,
                         # a skip-connection half) shares its lower half with another consumer
    N, C, H, W = t.shape
    if tuple(base.shape) != (N, 2 * C, H, W) or base.dtype != t.dtype or not base.is_contiguous(memory_format=torch.channels_last):
        return None
    if t.stride() != (H * W * 2 * C, 1, W * 2 * C, 2 * C) or t.storage_offset() != base.storage_offset() + C:
        return None
    return base


def instnorm_spade_bwd(dout, z, gamma, mean, rstd, fused_gb=False, up2=False, xlo=None):
    """returns (dz, dgamma); dbeta == dout and is not materialised.
    fused_gb=True: returns (dz, dgb) with dgb (N,2C,H,W) = [dgamma | dout] in one buffer (the gradient of
    a fused gamma+beta convolution output).
    up2=True (with fused_gb): z is the x2 bilinear resize of a map x -- the first result is d x (N, C, H/2, W/2), the resize's adjoint applied inside
    the kernel (mrdis_instnorm_spade_bwd_up2); None when the library declines the geometry.  xlo = x: z may be None, the kernels interpolate it from x."""
    lib = load()
    dout, lddo = nhwc(dout); gamma, ldg = nhwc(gamma)
    N, C, H, W = gamma.shape
    if z is not None:
        z, ldz = nhwc(z)
    else:
        assert up2 and xlo is not None
        ldz = 0
    dt = _dt(dout, gamma) if z is None else _dt(dout, z, gamma)
    nb = _ws_bytes(lib.mrdis_instnorm_spade_bwd_up2_workspace, N, H // 2, W // 2, C, dt) if up2 else _ws_bytes(lib.mrdis_instnorm_spade_bwd_workspace, N, H * W, C)
    ws = _ws(nb, gamma.device)
    if up2:
        assert fused_gb and H % 2 == 0 and W % 2 == 0
        dx = empty_nhwc(N, C, H // 2, W // 2, gamma.device, gamma.dtype)
        xl, ldxl = (None, 0)
        if xlo is not None:
            xl, ldxl = nhwc(xlo)
            assert tuple(xl.shape) == (N, C, H // 2, W // 2) and xl.dtype == gamma.dtype
        zp = _ptr(z) if z is not None else None
        xp = _ptr(xl) if xl is not None else None
        base = gb_slot(dout)
        if base is not None:
            rc = lib.mrdis_instnorm_spade_bwd_up2(_ptr(dout), lddo, zp, ldz, _ptr(gamma), ldg, _ptr(mean), _ptr(rstd), _ptr(dx), C,
                                                  base.data_ptr(), 2 * C, None, 0, _ptr(ws), nb, N, H // 2, W // 2, C, xp, ldxl, dt, _stream())
            dgb = base
        else:
            dgb = empty_nhwc(N, 2 * C, H, W, gamma.device, gamma.dtype)
            rc = lib.mrdis_instnorm_spade_bwd_up2(_ptr(dout), lddo, zp, ldz, _ptr(gamma), ldg, _ptr(mean), _ptr(rstd), _ptr(dx), C,
                                                  dgb.data_ptr(), 2 * C, dgb.data_ptr() + dgb.element_size() * C, 2 * C, _ptr(ws), nb, N, H // 2, W // 2, C, xp, ldxl, dt, _stream())
        if rc == -2:
            return None
        _chk(rc, 'instnorm_spade_bwd_up2')
        return dx, dgb
    dz = empty_nhwc(N, C, H, W, z.device, z.dtype)
    if fused_gb:
        base = gb_slot(dout)
        if base is not None:
            # dout already IS channels [C, 2C) of a 2C-channel buffer (its producer wrote it there, ops._GroupedConvFn): only dgamma is
            # written, into the first half -- one pass over the tensor less
            _chk(lib.mrdis_instnorm_spade_bwd(_ptr(dout), lddo, _ptr(z), ldz, _ptr(gamma), ldg, _ptr(mean), _ptr(rstd), _ptr(dz), C,
                                              base.data_ptr(), 2 * C, None, 0, _ptr(ws), nb, N, H * W, C, dt, _stream()), 'instnorm_spade_bwd')
            return dz, base
        dgb = empty_nhwc(N, 2 * C, H, W, z.device, z.dtype)
        _chk(lib.mrdis_instnorm_spade_bwd(_ptr(dout), lddo, _ptr(z), ldz, _ptr(gamma), ldg, _ptr(mean), _ptr(rstd), _ptr(dz), C,
                                          dgb.data_ptr(), 2 * C, dgb.data_ptr() + dgb.element_size() * C, 2 * C, _ptr(ws), nb, N, H * W, C, dt, _stream()),
             'instnorm_spade_bwd')
        return dz, dgb
    dg = empty_nhwc(N, C, H, W, z.device, z.dtype)
    _chk(lib.mrdis_instnorm_spade_bwd(_ptr(dout), lddo, _ptr(z), ldz, _ptr(gamma), ldg, _ptr(mean), _ptr(rstd), _ptr(dz), C,
                                      _ptr(dg), C, None, 0, _ptr(ws), nb, N, H * W, C, dt, _stream()), 'instnorm_spade_bwd')
    return dz, dg


# ---------------------------------------------------------------- resize / softmax / losses
def bilinear_fwd(x, out_hw, align_corners):
    lib = load()
    x, ldx = nhwc(x)
    N, C, Hi, Wi = x.shape
    Ho, Wo = out_hw
    y = empty_nhwc(N, C, Ho, Wo, x.device, x.dtype)
    _chk(lib.mrdis_bilinear_fwd(_ptr(x), ldx, _ptr(y), C, N, Hi, Wi, Ho, Wo, C, 1 if align_corners else 0, _dt(x), _stream()), 'bilinear_fwd')
    return y


def bilinear_up2_stats_applies(N, Wi, C):
    return bool(load().mrdis_bilinear_up2_stats_applies(int(N), int(Wi), int(C)))


def bilinear_up2_stats(x, eps, out_blocks=None):
    """x2 bilinear (align_corners=False) + instance statistics of the result: -> (y, mean, rstd), or None where the fused kernel does not apply.
    out_blocks: a (G, Bb, C, 2 Hi, 2 Wi) view, G * Bb = N, each out_blocks[g] a dense NHWC block: image n is written to out_blocks[n // Bb][n % Bb]
    (then y is out_blocks itself)."""
    lib = load()
    x, ldx = nhwc(x)
    N, C, Hi, Wi = x.shape
    if C % 4 != 0:
        return None
    blk, bstride = 0, 0
    if out_blocks is None:
        y = empty_nhwc(N, C, 2 * Hi, 2 * Wi, x.device, x.dtype)
        yptr = _ptr(y)
    else:
        G, Bb = out_blocks.shape[0], out_blocks.shape[1]
        y0, ld0 = nhwc(out_blocks[0])
        assert G * Bb == N and tuple(out_blocks.shape[2:]) == (C, 2 * Hi, 2 * Wi) and out_blocks.dtype == x.dtype and ld0 == C
        assert y0.data_ptr() == out_blocks.data_ptr(), 'blocks must be dense NHWC'
        y, yptr, blk, bstride = out_blocks, _ptr(y0), Bb, out_blocks.stride(0)
    mean = torch.empty(N * C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(N * C, dtype=torch.float32, device=x.device)
    nb = _ws_bytes(lib.mrdis_bilinear_up2_stats_workspace, N, Hi, C)
    ws = _ws(nb, x.device)
    rc = lib.mrdis_bilinear_up2_stats_fwd(_ptr(x), ldx, yptr, C, N, Hi, Wi, C, blk, bstride, _ptr(mean), _ptr(rstd), eps, _ptr(ws), nb, _dt(x), _stream())
    if rc == -2:
        return None
    _chk(rc, 'bilinear_up2_stats_fwd')
    return y, mean, rstd


def bilinear_bwd(dy, in_hw, align_corners, out=None):
    lib = load()
    dy, lddy = nhwc(dy)
    N, C, Ho, Wo = dy.shape
    Hi, Wi = in_hw
    if out is None:
        dx = empty_nhwc(N, C, Hi, Wi, dy.device, dy.dtype)
    else:
        dx, ldo = nhwc(out)
        assert dx.data_ptr() == out.data_ptr() and ldo == C and tuple(out.shape) == (N, C, Hi, Wi) and out.dtype == dy.dtype
    _chk(lib.mrdis_bilinear_bwd(_ptr(dy), lddy, _ptr(dx), C, N, Hi, Wi, Ho, Wo, C, 1 if align_corners else 0, _dt(dy), _stream()), 'bilinear_bwd')
    return dx


def softmax_mask_drop_fwd(s, mask_img, scale=100.0):
    lib = load()
    s, lds = nhwc(s)
    N, C, H, W = s.shape
    out = empty_nhwc(N, C, H, W, s.device)
    m = None if mask_img is None else mask_img.contiguous()
    _chk(lib.mrdis_softmax_mask_drop_fwd(_ptr(s), lds, _ptr(m), _ptr(out), C, N * H * W, C, scale, _stream()), 'softmax_mask_drop_fwd')
    return out


def softmax_mask_drop_bwd(dout, out):
    lib = load()
    dout, lddo = nhwc(dout); out, ldo = nhwc(out)
    N, C, H, W = out.shape
    ds = empty_nhwc(N, C, H, W, out.device)
    _chk(lib.mrdis_softmax_mask_drop_bwd(_ptr(dout), lddo, _ptr(out), ldo, _ptr(ds), C, N * H * W, C, _stream()), 'softmax_mask_drop_bwd')
    return ds


def recon_err_fwd(gt, x, p):
    lib = load()
    gt, ldgt = nhwc(gt); x, ldx = nhwc(x)
    N, C, H, W = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    nb = _ws_bytes(lib.mrdis_recon_err_workspace, N, H * W, C)
    ws = _ws(nb, x.device)
    _chk(lib.mrdis_recon_err_fwd(_ptr(gt), ldgt, _ptr(x), ldx, _ptr(out), _ptr(ws), nb, N, H * W, C, p, _stream()), 'recon_err_fwd')
    return out


def recon_metrics(target, pred):
    """(N, 3) = MSE / PSNR / SSIM of channel 0 of each sample (util.py:935-978)."""
    lib = load()
    target, ldt = nhwc(target); pred, ldp = nhwc(pred)
    N, _, H, W = pred.shape
    out = torch.empty(N, 3, dtype=torch.float32, device=pred.device)
    nb = _ws_bytes(lib.mrdis_recon_metrics_workspace, N, H)
    ws = _ws(nb, pred.device)
    _chk(lib.mrdis_recon_metrics(_ptr(target), ldt, _ptr(pred), ldp, _ptr(out), _ptr(ws), nb, N, H, W, _stream()), 'recon_metrics')
    return out


def slice_gather(vol_ptrs, slice_idx, drop, H, W, D, block):
    """vol_ptrs (B,M) int64, slice_idx (B) int32, drop (B) int32 on the device -> inputs (B,(2b+1)M,H,W) NHWC, mask (B,M),
    mask_img (B,H,W)."""
    lib = load()
    B, M = vol_ptrs.shape
    C = M * (2 * block + 1)
    inputs = empty_nhwc(B, C, H, W, vol_ptrs.device)
    mask = torch.empty((B, M), dtype=torch.float32, device=vol_ptrs.device)
    mask_img = torch.empty((B, H, W), dtype=torch.float32, device=vol_ptrs.device)
    _chk(lib.mrdis_slice_gather(_ptr(vol_ptrs), _ptr(slice_idx), _ptr(drop), _ptr(inputs), C, _ptr(mask), _ptr(mask_img),
                                B, M, H, W, D, block, _stream()), 'slice_gather')
    return inputs, mask, mask_img


def recon_err_bwd(gt, x, w, p):
    lib = load()
    gt, ldgt = nhwc(gt); x, ldx = nhwc(x)
    N, C, H, W = x.shape
    dx = empty_nhwc(N, C, H, W, x.device)
    w = w.contiguous()
    _chk(lib.mrdis_recon_err_bwd(_ptr(gt), ldgt, _ptr(x), ldx, _ptr(w), _ptr(dx), C, N, H * W, C, p, _stream()), 'recon_err_bwd')
    return dx


def maxpool_fwd(x, k):
    lib = load()
    x, ldx = nhwc(x)
    N, C, H, W = x.shape
    y = empty_nhwc(N, C, H // k, W // k, x.device)
    arg = torch.empty((N, H // k, W // k, C), dtype=torch.int32, device=x.device)
    _chk(lib.mrdis_maxpool_fwd(_ptr(x), ldx, _ptr(y), _ptr(arg), N, H, W, C, k, _stream()), 'maxpool_fwd')
    return y, arg


def maxpool_bwd(dy, arg, in_shape, k):
    lib = load()
    N, C, H, W = in_shape
    dy, lddy = nhwc(dy)
    if lddy != C:
        dy = dy.contiguous(memory_format=torch.channels_last)
    dx = empty_nhwc(N, C, H, W, dy.device)
    _chk(lib.mrdis_maxpool_bwd(_ptr(dy), _ptr(arg), _ptr(dx), C, N, H, W, C, k, _stream()), 'maxpool_bwd')
    return dx


# ---------------------------------------------------------------- optimizer arena
def sumsq_finite(g, out):
    lib = load()
    nb = _ws_bytes(lib.mrdis_sumsq_workspace)
    ws = _ws(nb, g.device)
    _chk(lib.mrdis_sumsq_finite(_ptr(g), g.numel(), _ptr(out), _ptr(ws), nb, _stream()), 'sumsq_finite')


def adam_amsgrad_step(p, g, m, v, vmax, lr, beta1, beta2, eps, weight_decay, step, norm_finite, max_norm, grad_scale=1.0,
                      step_state=None, gates=None, gate_steps=None):
    """step: 1-based host count, ignored when `step_state` (device float[2]: applied, skipped) is given.
    gates: (ranges [(lo, hi), ...], flag_index [...], flags device tensor) or None.
    gate_steps: device float[3 * n_flags] per-flag step counters (+ scratch), see include/mrdis.h."""
    lib = load()
    if gates is not None and len(gates[0]):
        ranges, fidx, flags = gates
        n_g = len(ranges)
        ra = (_c.c_longlong * (2 * n_g))(*[int(x) for r in ranges for x in r])
        fa = (_c.c_int * n_g)(*[int(i) for i in fidx])
        fl = flags.data_ptr()
    else:
        n_g, ra, fa, fl = 0, None, None, None
    n_flags = 0
    if n_g and gate_steps is not None:
        n_flags = gate_steps.numel() // 3
    else:
        gate_steps = None
    _chk(lib.mrdis_adam_amsgrad_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(vmax), p.numel(), lr, beta1, beta2, eps, weight_decay,
                                     int(step), _ptr(step_state), _ptr(norm_finite), max_norm, grad_scale, ra, fa, n_g, fl, _ptr(gate_steps), n_flags, _stream()),
         'adam_amsgrad_step')


# ================================================================ 3-D path (NDHWC, torch.channels_last_3d)
def ndhwc(t):
    """(tensor, ld) of a logical (N,C,D,H,W) fp32 tensor stored NDHWC (channels_last_3d); copies if it is not."""
    assert t.dim() == 5 and t.dtype == torch.float32, (t.shape, t.dtype)
    N, C, D, H, W = t.shape
    want = (D * H * W * C, 1, H * W * C, W * C, C)
    if any(t.shape[i] > 1 and t.stride()[i] != want[i] for i in range(5)):
        t = t.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
    return t, C


def empty_ndhwc(N, C, D, H, W, device):
    return torch.empty((N, D, H, W, C), dtype=torch.float32, device=device).permute(0, 4, 1, 2, 3)


def _out3(D, H, W, k, stride, pad):
    return tuple((v + 2 * pad - k) // stride + 1 for v in (D, H, W))


def conv3d_fwd(x, w_tck, bias, k, stride, pad, residual=None):
    x, ldx = ndhwc(x)
    N, Ci, D, H, W = x.shape
    Co = w_tck.shape[2]
    Do, Ho, Wo = _out3(D, H, W, k, stride, pad)
    y = empty_ndhwc(N, Co, Do, Ho, Wo, x.device)
    ldr = 0
    if residual is not None:
        residual, ldr = ndhwc(residual)
        assert tuple(residual.shape) == tuple(y.shape)
    _chk(load().mrdis_conv3d_fwd(x.data_ptr(), ldx, w_tck.data_ptr(), _ptr(bias), _ptr(residual), ldr, y.data_ptr(), Co,
                                 N, D, H, W, Ci, Co, k, stride, pad, _stream()), 'conv3d_fwd')
    return y


def conv3d_bwd_data(dy, w_tkc, in_shape, k, stride, pad):
    dy, lddy = ndhwc(dy)
    N, Ci, D, H, W = in_shape
    Co = dy.shape[1]
    dx = empty_ndhwc(N, Ci, D, H, W, dy.device)
    _chk(load().mrdis_conv3d_bwd_data(dy.data_ptr(), lddy, w_tkc.data_ptr(), dx.data_ptr(), Ci,
                                      N, D, H, W, Ci, Co, k, stride, pad, _stream()), 'conv3d_bwd_data')
    return dx


def conv3d_bwd_weight(x, dy, k, stride, pad, want_bias):
    x, ldx = ndhwc(x)
    dy, lddy = ndhwc(dy)
    N, Ci, D, H, W = x.shape
    Co = dy.shape[1]
    lib = load()
    need = lib.mrdis_conv3d_bwd_weight_workspace(N, D, H, W, Ci, Co, k, stride, pad)
    if need == 0:
        raise MrdisError('conv3d_bwd_weight: unsupported geometry')
    ws = _ws(need, x.device)
    dw = torch.empty((k * k * k, Ci, Co), dtype=torch.float32, device=x.device)
    db = torch.empty((Co,), dtype=torch.float32, device=x.device) if want_bias else None
    _chk(lib.mrdis_conv3d_bwd_weight(x.data_ptr(), ldx, dy.data_ptr(), lddy, dw.data_ptr(), _ptr(db), ws.data_ptr(), ws.numel(),
                                     N, D, H, W, Ci, Co, k, stride, pad, _stream()), 'conv3d_bwd_weight')
    return dw, db


def groupnorm_relu_fwd(x, gamma, beta, G, eps, relu):
    x, ld = ndhwc(x)
    N, C, D, H, W = x.shape
    P = D * H * W
    y = empty_ndhwc(N, C, D, H, W, x.device)
    mean = torch.empty((N, G), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    lib = load()
    ws = _ws(lib.mrdis_groupnorm_workspace(N, P, C, G), x.device)
    _chk(lib.mrdis_groupnorm_relu_fwd(x.data_ptr(), ld, y.data_ptr(), C, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                      rstd.data_ptr(), ws.data_ptr(), ws.numel(), N, P, C, G, eps, int(relu), _stream()), 'groupnorm_relu_fwd')
    return y, mean, rstd


def groupnorm_relu_bwd(dy, x, gamma, beta, mean, rstd, G, relu, add=None):
    """add: a second gradient of x (same shape), summed into dx by the same pass (include/mrdis.h mrdis_groupnorm_relu_bwd_add)"""
    x, ld = ndhwc(x)
    dy, lddy = ndhwc(dy)
    ldadd = 0
    if add is not None:
        add, ldadd = ndhwc(add)
        assert tuple(add.shape) == tuple(x.shape)
    N, C, D, H, W = x.shape
    P = D * H * W
    dx = empty_ndhwc(N, C, D, H, W, x.device)
    dgamma = torch.empty((C,), dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    lib = load()
    ws = _ws(lib.mrdis_groupnorm_workspace(N, P, C, G), x.device)
    _chk(lib.mrdis_groupnorm_relu_bwd_add(dy.data_ptr(), lddy, x.data_ptr(), ld, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), dx.data_ptr(), C, dgamma.data_ptr(), dbeta.data_ptr(), _ptr(add), ldadd, ws.data_ptr(), ws.numel(),
                                          N, P, C, G, int(relu), _stream()), 'groupnorm_relu_bwd')
    return dx, dgamma, dbeta


def upsample2x_add_fwd(x, skip):
    x, _ = ndhwc(x)
    N, C, D, H, W = x.shape
    if skip is not None:
        skip, _ = ndhwc(skip)
        assert tuple(skip.shape) == (N, C, 2 * D, 2 * H, 2 * W)
    y = empty_ndhwc(N, C, 2 * D, 2 * H, 2 * W, x.device)
    _chk(load().mrdis_upsample2x_add_fwd(x.data_ptr(), _ptr(skip), y.data_ptr(), N, D, H, W, C, _stream()), 'upsample2x_add_fwd')
    return y


def upsample2x_bwd(dy):
    dy, _ = ndhwc(dy)
    N, C, D2, H2, W2 = dy.shape
    dx = empty_ndhwc(N, C, D2 // 2, H2 // 2, W2 // 2, dy.device)
    _chk(load().mrdis_upsample2x_bwd(dy.data_ptr(), dx.data_ptr(), N, D2 // 2, H2 // 2, W2 // 2, C, _stream()), 'upsample2x_bwd')
    return dx
