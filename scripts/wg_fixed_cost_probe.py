"""Fixed cost of a conv workgroup: 1x1 convs with K = 32 .. 512 over many tiles; time per round of 768 tiles vs K-tiles
per tile -> intercept = what a workgroup costs beside its K loop (prologue, first loads, epilogue, dispatch)."""
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


for norm in (False, True):
    for C in (32, 64, 128, 256, 512):
        N, H, CO = 64, 96, 128            # M = 589824 -> 9216 tiles of 64x128 = 12 rounds of 768
        x = torch.randn(N, H, H, C, device='cuda')
        w = torch.randn(1, 1, C, CO, device='cuda') * 0.05
        ab = torch.randn(2 * C, device='cuda') if norm else None
        v = View(x, ab0=ab, act=ACT_RELU if norm else ACT_NONE)
        out = torch.empty(N, H, H, CO, device='cuda')
        ms = timeit(lambda: hip.conv_forward(v, w, 1, 0, out))
        tiles = N * H * H // 64
        rounds = tiles / 768.0
        print('norm %d  K=%4d (%2d K-tiles)  %7.3f ms  %6.1f us per round of 768 tiles  %6.1f TFLOP/s  out %.0f MB -> %.2f TB/s written'
              % (norm, C, C // 32, ms, ms * 1e3 / rounds, 2.0 * N * H * H * C * CO / ms / 1e9, out.numel() * 4 / 1e6, out.numel() * 4 / ms / 1e9))
