"""Hosted norm-backward apply pass (hip.ApplyJob inside a filter-gradient launch) against the two launches one after the other.
usage: side_apply_probe.py [batch]   -- encoder_k filter gradient + the norm backward of encoder_{k-1}'s output, k = 2..5"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import ACT_LRELU, ACT_RELU, View

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for k, hin, ci, co in ((2, 96, 64, 128), (3, 48, 128, 256), (4, 24, 256, 512), (5, 12, 512, 512)):
    x = r(N, hin, hin, ci)                      # e[k-1], raw
    ab = torch.cat([torch.ones(ci, device='cuda'), torch.zeros(ci, device='cuda')])
    st = torch.cat([torch.zeros(ci, device='cuda'), torch.ones(ci, device='cuda')])
    dy = r(N, hin // 2, hin // 2, co)           # gradient w.r.t. e[k]
    dw = torch.empty(4, 4, ci, co, device='cuda')
    g1, g2, dx = r(N, hin, hin, ci), r(N, hin, hin, ci), torch.empty(N, hin, hin, ci, device='cuda')
    coef = torch.zeros(2 * ci, device='cuda')
    x2d, g1r, g2r, dxr = [t.view(-1, ci) for t in (x, g1, g2, dx)]
    xin = View(x, None, ab, ACT_LRELU)

    def job():
        return hip.bn_act_backward(x2d, ab, st, g1r, ACT_LRELU, dxr, g2=g2r, act2=ACT_RELU, defer=True, coef=coef)

    def sums_only():
        job().done = True

    def separate():
        j = job()
        hip.conv_wgrad(xin, View(dy), dw, 2, 1)
        hip.apply_now(j)

    def hosted():
        hip.conv_wgrad(xin, View(dy), dw, 2, 1, host=job())

    t_w = timeit(lambda: hip.conv_wgrad(xin, View(dy), dw, 2, 1))
    t_s = timeit(sums_only)
    t_sep = timeit(separate)
    t_h = timeit(hosted)
    mb = 4 * x.numel() * 4 / 1e6
    print('encoder_%d: wgrad %.1f us | sums %.1f | wgrad + sums + apply: separate %.1f, hosted %.1f  (apply moves %.0f MB)'
          % (k, t_w, t_s, t_sep, t_h, mb), flush=True)
