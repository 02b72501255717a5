"""Does the speed of the fp32-MFMA conv kernel depend on the operand VALUES?  Same kernel, same shape (1x1 conv =
GEMM 65536 x 1024 x 1024), inputs: dense randn, relu(randn), zeros, ones."""
import sys, torch
sys.path.insert(0, '.')
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
w = torch.randn(1, 1, 1024, 1024, device='cuda') * 0.05
out = torch.empty(16, 64, 64, 1024, device='cuda')
fl = 2.0 * 65536 * 1024 * 1024
for name, x in (('randn', torch.randn(16, 64, 64, 1024, device='cuda')),
                ('relu(randn): half zeros', torch.relu(torch.randn(16, 64, 64, 1024, device='cuda'))),
                ('zeros', torch.zeros(16, 64, 64, 1024, device='cuda')),
                ('ones', torch.ones(16, 64, 64, 1024, device='cuda'))):
    v = View(x)
    ms = timeit(lambda: hip.conv_forward(v, w, 1, 0, out))
    print('%-26s %.3f ms  %.1f TFLOP/s' % (name, ms, fl / ms / 1e9))
