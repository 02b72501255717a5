"""Is the bf16-split conv launch bit-reproducible -- alone, and with another conv running beside it on a second stream?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
N = 32
cases = {}
x3, w3 = r(N, 48, 48, 128), r(4, 4, 128, 256) * 0.02
ab3 = torch.cat([torch.ones(128, device='cuda'), torch.zeros(128, device='cuda')])
cases['enc3'] = (lambda out: hip.conv_forward(View(x3, None, ab3, 2), w3, 2, 1, out), (N, 24, 24, 256))
x2, w2 = r(N, 96, 96, 64), r(4, 4, 64, 128) * 0.02
ab2 = torch.cat([torch.ones(64, device='cuda'), torch.zeros(64, device='cuda')])
cases['enc2'] = (lambda out: hip.conv_forward(View(x2, None, ab2, 2), w2, 2, 1, out), (N, 48, 48, 128))
xd0, xd1, fd = r(N, 24, 24, 256), r(N, 24, 24, 256), r(4, 4, 128, 512) * 0.02
cases['dec3'] = (lambda out: hip.deconv_forward(View(xd0, xd1, None, 1, None), fd, out), (N, 48, 48, 128))
x4, w4 = r(N, 24, 24, 256), r(4, 4, 256, 512) * 0.02
other_out = torch.empty(N, 23, 23, 512, device='cuda')
ab4 = torch.cat([torch.ones(256, device='cuda'), torch.zeros(256, device='cuda')])
other = lambda: hip.conv_forward(View(x4, None, ab4, 2), w4, 1, 1, other_out)
side = torch.cuda.Stream()
for sk in (True, False):
    hip.SK_ENABLED = sk
    for name, (fn, shape) in cases.items():
        ref = torch.empty(shape, device='cuda')
        fn(ref)
        torch.cuda.synchronize()
        bad_alone = bad_co = 0
        for rep in range(30):
            out = torch.full(shape, float('nan'), device='cuda')
            fn(out)
            torch.cuda.synchronize()
            bad_alone += int(not torch.equal(out, ref))
        for rep in range(30):
            out = torch.full(shape, float('nan'), device='cuda')
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                other()
                other()
            fn(out)
            torch.cuda.synchronize()
            bad_co += int(not torch.equal(out, ref))
        print('streamk=%d %-5s  differs from the first run: alone %d / 30, beside another stream %d / 30' % (sk, name, bad_alone, bad_co))
