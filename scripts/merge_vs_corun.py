"""Is one launch over two batches better than two launches side by side on two streams?  (DESIGN 6.2 lead 1.)
Replays hipGraphs of (a) two batch-32 launches on two streams, (b) one batch-64 launch, (c) two batch-32 launches in line."""
import os, sys
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)


def bench(fn, iters=200):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode='thread_local'):
        for _ in range(10):
            fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters // 10):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, hin, ci, co in (('enc3', 48, 128, 256), ('enc4', 24, 256, 512), ('enc5', 12, 512, 512)):
    N = 32
    w = r(4, 4, ci, co) * 0.02
    ab = torch.cat([torch.ones(ci, device='cuda'), torch.zeros(ci, device='cuda')])
    xa, xb, x2 = r(N, hin, hin, ci), r(N, hin, hin, ci), r(2 * N, hin, hin, ci)
    oa, ob, o2 = (torch.empty(n, hin // 2, hin // 2, co, device='cuda') for n in (N, N, 2 * N))
    s1 = torch.cuda.Stream()
    hip.workspace(); hip.sk_flags()
    with torch.cuda.stream(s1):
        hip.workspace(); hip.sk_flags()
    torch.cuda.synchronize()

    def two_streams():
        main = torch.cuda.current_stream()
        s1.wait_stream(main)
        with torch.cuda.stream(s1):
            hip.conv_forward(View(xb, None, ab, 2), w, 2, 1, ob)
        hip.conv_forward(View(xa, None, ab, 2), w, 2, 1, oa)
        main.wait_stream(s1)

    def in_line():
        hip.conv_forward(View(xa, None, ab, 2), w, 2, 1, oa)
        hip.conv_forward(View(xb, None, ab, 2), w, 2, 1, ob)

    def merged():
        hip.conv_forward(View(x2, None, ab, 2), w, 2, 1, o2)
    fl = 2.0 * 2 * N * (hin // 2) ** 2 * co * 16 * ci
    for label, fn in (('two streams', two_streams), ('in line', in_line), ('one launch', merged)):
        us = bench(fn)
        print('%s  %-12s %7.1f us per pair  %6.1f TFLOP/s' % (name, label, us, fl / us / 1e6))
