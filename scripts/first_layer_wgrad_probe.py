"""Filter gradients of the two first-layer convs (3(4) -> 64 and 8 -> 64 channels at 192x192, batch 32): tiny outputs, all
parallelism from the split over 294912 pixels.  Diagnostic: times ssc_conv_wgrad on those shapes."""
import sys

import torch

sys.path.insert(0, '.')
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, c, creal in (('G encoder_1 (4 ch, 3 real)', 4, 3), ('D layer_1 (8 ch)', 8, 8), ('64 ch for scale', 64, 64)):
    x = torch.randn(32, 192, 192, c, device='cuda')
    dy = torch.randn(32, 96, 96, 64, device='cuda')
    dw = torch.empty(4, 4, creal, 64, device='cuda')
    ms = timeit(lambda: hip.conv_wgrad(View(x), View(dy), dw, 2, 1))
    fl = 2.0 * 32 * 96 * 96 * 16 * creal * 64
    byts = (x.numel() + dy.numel()) * 4
    print('%-28s %7.3f ms  %6.1f TFLOP/s   (operands once from HBM at 4 TB/s: %.3f ms)' % (name, ms, fl / ms / 1e9, byts / 4e9))
