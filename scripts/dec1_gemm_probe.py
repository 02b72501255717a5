"""Feasibility of decoder_1 (transposed conv 4x4 s2, 128 -> 3) as GEMM [pixels x 128] x [128 x 48] + a 4-tap col2im: time of the
GEMM part through the conv kernel (1x1 conv, two-source view with folded norm + relu as the layer reads it) against the layer's
present launch.  usage: dec1_gemm_probe.py [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import ACT_RELU, View

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator(device='cuda').manual_seed(0)
d2 = torch.randn(n, 96, 96, 64, device='cuda', generator=g)
e1 = torch.randn(n, 96, 96, 64, device='cuda', generator=g)
ab = torch.cat([torch.ones(64), torch.zeros(64)]).cuda()
w1 = torch.randn(1, 1, 128, 64, device='cuda', generator=g) * 0.05
hip.register_param_buffer(w1)
f = torch.randn(4, 4, 3, 128, device='cuda', generator=g) * 0.05
hip.register_param_buffer(f)
T = torch.empty(n, 96, 96, 64, device='cuda')
out = torch.empty(n, 192, 192, 4, device='cuda')
v = View(d2, e1, ab, ACT_RELU, None)


def timed(fn, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(20):
                fn()
        gr.replay()
        st.synchronize()
        e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1_.record()
        st.synchronize()
    print('%-60s %8.2f us' % (label, e0.elapsed_time(e1_) * 10))


timed(lambda: hip.deconv_forward(v, f, out, coff=0, nstore=4, epi=1), 'decoder_1 as it runs (narrow kernel, tanh), batch %d' % n)
timed(lambda: hip.conv_forward(v, w1, 1, 0, T), 'GEMM part: 1x1 conv 128 -> 64 over the 96^2 lattice')
x = torch.empty(n * 96 * 96 * 64, device='cuda')
timed(lambda: hip.call('ssc_axpy', x, T.view(-1), 1.0, x.numel()), '(an elementwise pass over the GEMM result: read 2, write 1)')
