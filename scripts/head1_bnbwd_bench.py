import sys, torch
sys.path.insert(0, '.')
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View
N=32
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
x, dy, w = r(N,23,23,512), r(N,22,22,4), r(4,4,512,1)*0.02
ab, st = torch.empty(1024, device='cuda'), torch.empty(1024, device='cuda')
hip.bn_stats(x.view(-1,512), torch.ones(512,device='cuda'), torch.zeros(512,device='cuda'), ab, st)
dx, ds, do = torch.empty_like(x), torch.empty(512,device='cuda'), torch.empty(512,device='cuda')
v = r(N,512)
g4 = torch.empty_like(x)
def fused(): hip.head1_dgrad_bn_backward(View(dy), w, 1, x, ab, st, 2, dx, dscale=ds, doffset=do, rowb=(v, 1/529.))
def sep():
    hip.conv_dgrad(View(dy), w, 1, 1, g4, k_real=1)
    hip.bn_act_backward(x.view(-1,512), ab, st, g4.view(-1,512), 2, dx.view(-1,512), dscale=ds, doffset=do, rowb=(v,1/529.,529))
for name, fn in (('fused', fused), ('separate', sep)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1)*10, 'us')
