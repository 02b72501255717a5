"""Achieved HBM rate of the MRU pointwise / reduction kernels (mru_ops.hip) at the shapes of the batch-32 train step.
usage: mru_pw_bench.py [iters]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import ACT_MIU

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)


def timed(name, fn, nbytes):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print('%-34s %8.1f us  %7.1f MB  %6.2f TB/s' % (name, ms * 1e3, nbytes / 1e6, nbytes / ms / 1e9))


N = 32
for (H, ch, cs, d) in ((96, 128, 64, 64), (48, 256, 128, 128), (24, 512, 256, 256)):
    h = H // 2
    ht, z, skip = r(N, h, h, ch), r(N, H, H, 4), r(N, H, H, cs)
    ct = ch + 3 + cs
    ld = (ct + 3) // 4 * 4
    full = torch.zeros(N, H, H, ld, device='cuda')
    rg = r(N, H, H, ch)
    mm = torch.stack([rg.amin(dim=(1, 2)), rg.amax(dim=(1, 2))], 1).contiguous()
    P = N * H * H
    parts = [dict(x=ht, upsample=True), dict(x=z, C=3), dict(x=skip)]
    timed('concat up(%d)|3|%d @%d' % (ch, cs, H), lambda: hip.concat_parts(full, parts), P * 4 * (ct + cs + 3 + ch / 4))
    partsg = [dict(x=ht, upsample=True, gate=(rg, mm)), dict(x=z, C=3), dict(x=skip)]
    timed('concat gate*up(%d)|3|%d @%d' % (ch, cs, H), lambda: hip.concat_parts(full, partsg),
          P * 4 * (ct + cs + 3 + ch / 4 + ch))
    raw = r(N, H, H, d)
    abn = r(N, 2 * d)
    out = torch.empty_like(raw)
    timed('norm_activ (one part, %d) @%d' % (d, H), lambda: hip.concat_parts(out, [dict(x=raw, ab=abn, act=ACT_MIU)]),
          P * 4 * 2 * d)
    h2, zg = r(N, H, H, d), r(N, H, H, d)
    mmz = torch.stack([zg.amin(dim=(1, 2)), zg.amax(dim=(1, 2))], 1).contiguous()
    pj = r(N, h, h, d)
    timed('blend (%d) @%d' % (d, H), lambda: hip.call('ssc_mru_blend', pj, abn, 1, h2, abn, zg, mmz, out, N, H, H, d),
          P * 4 * (3 * d + d / 4))
    mnmx = torch.empty(N, 2, d, device='cuda')
    timed('minmax_hw (%d) @%d' % (d, H), lambda: hip.minmax_hw(zg, mnmx), P * 4 * d)
    pooled = torch.empty(N, h, h, d, device='cuda')
    timed('mean_pool2 (%d) @%d' % (d, H), lambda: hip.call('ssc_mean_pool2', raw, d, pooled, d, N, H, H, d), P * 4 * d * 1.25)
    dst = torch.empty_like(raw)
    timed('copy (torch) (%d) @%d' % (d, H), lambda: dst.copy_(raw), P * 4 * 2 * d)
