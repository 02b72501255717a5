"""Stand-alone timing of hip.concat_parts at the MRU decoder's shapes (batch 32, 192 x 192): GB/s of the bytes it must move.
usage: concat_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sketchyscenecolorization_amd import hip

N = 32
dev = 'cuda'


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (H, ch, cs, gate) in [(192, 128, 0, False), (192, 128, 0, True), (96, 128, 64, False), (96, 128, 64, True), (48, 256, 128, True),
                          (24, 384, 256, True)]:
    h = H // 2
    ht = torch.randn(N, h, h, ch, device=dev)
    z = torch.randn(N, H, H, 4, device=dev)
    skip = torch.randn(N, H, H, cs, device=dev) if cs else None
    ct = ch + 3 + cs
    ld = (ct + 3) // 4 * 4
    out = torch.zeros(N, H, H, ld, device=dev)
    rg = torch.randn(N, H, H, ch, device=dev)
    mm = torch.stack([rg.view(N, -1, ch).amin(1), rg.view(N, -1, ch).amax(1)], 1).contiguous()
    parts = [dict(x=ht, upsample=True, gate=(rg, mm) if gate else None), dict(x=z, C=3)] + ([dict(x=skip)] if cs else [])
    us = timeit(lambda: hip.concat_parts(out, parts))
    nbytes = 4.0 * (out.numel() + ht.numel() + z.numel() + (skip.numel() if cs else 0) + (rg.numel() if gate else 0))
    print('deconv concat H=%d ch=%d skip=%d gate=%d: %.1f us, %.2f GB moved, %.2f TB/s' % (H, ch, cs, gate, us, nbytes / 1e9, nbytes / us / 1e6))
# conv block: [miu(cbn(ht)) | xin]
for (H, ch) in [(96, 8), (48, 64), (24, 128)]:
    ht = torch.randn(N, H, H, ch, device=dev)
    xin = torch.randn(N, H, H, 4, device=dev)
    abn = torch.randn(N, 2 * ch, device=dev)
    out = torch.zeros(N, H, H, ch + 4, device=dev)
    us = timeit(lambda: hip.concat_parts(out, [dict(x=ht, ab=abn, act=hip.ACT_MIU), dict(x=xin, C=3)]))
    nbytes = 4.0 * (out.numel() + ht.numel() + xin.numel())
    print('conv-block concat H=%d ch=%d: %.1f us, %.2f TB/s' % (H, ch, us, nbytes / us / 1e6))
