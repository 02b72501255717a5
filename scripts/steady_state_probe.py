"""Where does the implicit-GEMM kernel lose time?  Times ssc_conv_forward on (a) a large pure GEMM (1x1 conv),
(b) the same with the folded norm + relu applied on load, (c) 4x4/stride-2 gathers at 8x the batch of the training
step (steady state) and (d) at the training batch.  Diagnostic only."""
import sys

import torch

sys.path.insert(0, '.')
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import ACT_NONE, ACT_RELU, View


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


def case(name, N, H, C, CO, k, stride, pad, norm):
    x = torch.randn(N, H, H, C, device='cuda')
    w = torch.randn(k, k, C, CO, device='cuda') * 0.05
    ab = torch.randn(2 * C, device='cuda') if norm else None
    v = View(x, ab0=ab, act=ACT_RELU if norm else ACT_NONE)
    OH = (H + 2 * pad - k) // stride + 1
    out = torch.empty(N, OH, OH, CO, device='cuda')
    ms = timeit(lambda: hip.conv_forward(v, w, stride, pad, out))
    fl = 2.0 * N * OH * OH * k * k * C * CO
    print('%-34s M=%7d N=%5d K=%5d  %7.3f ms  %6.1f TFLOP/s' % (name, N * OH * OH, CO, k * k * C, ms, fl / ms / 1e9))


case('1x1 GEMM plain', 16, 64, 1024, 1024, 1, 1, 0, False)
case('1x1 GEMM norm+relu on load', 16, 64, 1024, 1024, 1, 1, 0, True)
case('1x1 GEMM plain 4096^3', 16, 16, 4096, 4096, 1, 1, 0, False)
case('4x4s2 enc3 shape, batch 256', 256, 48, 128, 256, 4, 2, 1, True)
case('4x4s2 enc3 shape, batch 32', 32, 48, 128, 256, 4, 2, 1, True)
case('4x4s2 enc2 shape, batch 256', 256, 96, 64, 128, 4, 2, 1, True)
case('4x4s2 enc2 shape, batch 32', 32, 96, 64, 128, 4, 2, 1, True)
case('4x4s2 enc4 shape, batch 32', 32, 24, 256, 512, 4, 2, 1, True)
case('4x4s2 enc5 shape, batch 32', 32, 12, 512, 512, 4, 2, 1, True)

# round quantisation: 64x128 tiles, 3 workgroups per CU -> 768 resident; M = 64 * tiles
if len(sys.argv) > 1 and sys.argv[1] == 'rounds':
    for n in (12, 18, 24, 36, 48):
        case('1x1 K=1024 N=128, %d tiles' % (n * 64 * 64 // 64), n, 64, 1024, 128, 1, 1, 0, True)
