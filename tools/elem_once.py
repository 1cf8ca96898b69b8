"""Times the SPADE backward pair (statistics pass + apply pass) at one shape with its six tensors placed a chosen distance apart:
    python tools/elem_once.py [N C H W] [stagger bytes ...]
The tensors are carved out of one allocation, tensor k starting at k * (tensor bytes rounded up to 2 MiB + stagger): stagger 0 is what the
caching allocator gives equal-sized power-of-two tensors (every stream at the same offset of a different 2^28-byte-aligned region)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from mrdis import hip  # noqa: E402

args = [int(a) for a in sys.argv[1:]]
N, C, H, W = args[:4] if len(args) >= 4 else (32, 32, 256, 256)
staggers = args[4:] or [0, 4096, 65536, 1 << 20, 69632, 17 * 4096 + 256]
dev = torch.device('cuda:0')
hip.load()


def carve(flat, off_bytes, n, c):
    t = flat[off_bytes // 4: off_bytes // 4 + n * H * W * c].view(n, H, W, c).permute(0, 3, 1, 2)
    return t


def timed(fn, reps=6, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for st in staggers:
    one = N * H * W * C * 4
    pitch1 = ((one + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + st
    pitch2 = ((2 * one + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + st
    flat = torch.empty((3 * pitch1 + 2 * pitch2) // 4 + 1024, dtype=torch.float32, device=dev)
    flat.normal_()
    dout_src = carve(flat, 0, N, C)
    z = carve(flat, pitch1, N, C)
    gb = carve(flat, 2 * pitch1, N, 2 * C)                # gamma = channels [0, C) of the fused gamma+beta map
    gamma = gb[:, :C]
    mean = torch.zeros(N * C, device=dev); rstd = torch.ones(N * C, device=dev)
    # fused_gb=True without a private slot: dz, [dgamma | dbeta] freshly allocated by the caching allocator (as in the step)
    us_alloc = timed(lambda: hip.instnorm_spade_bwd(dout_src, z, gamma, mean, rstd, fused_gb=True))
    # the same call with every tensor at a staggered address: the library entry point directly
    lib = hip.load()
    dz = carve(flat, 2 * pitch1 + pitch2, N, C)
    dgb = carve(flat, 3 * pitch1 + pitch2, N, 2 * C)
    nb = hip._ws_bytes(lib.mrdis_instnorm_spade_bwd_workspace, N, H * W, C)
    ws = hip._ws(nb, dev)

    def direct():
        hip._chk(lib.mrdis_instnorm_spade_bwd(dout_src.data_ptr(), C, z.data_ptr(), C, gamma.data_ptr(), 2 * C, mean.data_ptr(), rstd.data_ptr(),
                                              dz.data_ptr(), C, dgb.data_ptr(), 2 * C, dgb.data_ptr() + 4 * C, 2 * C, ws.data_ptr(), nb, N, H * W, C,
                                              hip._dt(dout_src), hip._stream()), 'instnorm_spade_bwd')
    us_direct = timed(direct)
    gbytes = one * (3 + 3 + 3) / 1e9
    print(f'N{N} C{C} {H}x{W} stagger {st:8d}: allocator-placed outputs {us_alloc:8.1f} us, all staggered {us_direct:8.1f} us '
          f'({gbytes / us_direct * 1e6 / 1e3:.2f} TB/s over {gbytes:.2f} GB of reads + writes)', flush=True)
    del flat, dout_src, z, gb, gamma, dz, dgb
    torch.cuda.empty_cache()
