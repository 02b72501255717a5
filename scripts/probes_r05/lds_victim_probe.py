"""Does a conv launch write outside its LDS allocation?  Victim workgroups (scripts/lds_victim.hip) guard an LDS pattern on one
stream while conv launches run on another."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View
V = ctypes.CDLL(os.path.join(os.path.dirname(hip.__file__), 'lib', 'libvictim.so'))
V.victim_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
N = 32
x3, w3 = r(N, 48, 48, 128), r(4, 4, 128, 256) * 0.02
ab3 = torch.cat([torch.ones(128, device='cuda'), torch.zeros(128, device='cuda')])
out3 = torch.empty(N, 24, 24, 256, device='cuda')
dy, w = r(N, 24, 24, 256), r(4, 4, 128, 256) * 0.02
dx = torch.empty(N, 48, 48, 128, device='cuda')
convs = {'enc3 fwd': lambda: hip.conv_forward(View(x3, None, ab3, 2), w3, 2, 1, out3),
         'dg3 dgrad': lambda: hip.conv_dgrad(View(dy), w, 2, 1, dx)}
side = torch.cuda.Stream()
bad = torch.zeros(2, dtype=torch.int32, device='cuda')
for lds_kb in (64, 80):
    for name, fn in convs.items():
        for mode in ('bf16x6', 'fp32'):
            hip.ARITH_BF16 = mode == 'bf16x6'
            fn(); torch.cuda.synchronize()
            bad.zero_()
            for rep in range(20):
                with torch.cuda.stream(side):
                    V.victim_run(256, lds_kb * 1024, 400, bad.data_ptr(), torch.cuda.current_stream().cuda_stream)
                for k in range(6):
                    fn()
                torch.cuda.synchronize()
            print('victim %d KB beside %-10s %-7s: corrupted words %d (highest first index %d)' % (lds_kb, name, mode, int(bad[0]), int(bad[1])))
hip.ARITH_BF16 = True
