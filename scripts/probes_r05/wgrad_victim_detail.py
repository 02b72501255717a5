"""Anatomy of the corruption of the wgrad128 result beside a bf16-MFMA stream: which tile elements, and is the error the
contribution of particular pixels (K steps)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.hip import View
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
N = 32
x3, dy3 = r(N, 48, 48, 128), r(N, 24, 24, 256)
w3 = r(4, 4, 128, 256) * 0.02
dyd3, gd3 = r(N, 24, 24, 256), torch.empty(N, 48, 48, 128, device='cuda')
fn = lambda out: hip.conv_wgrad(View(x3), View(dy3), out, 2, 1)
nb = lambda: hip.conv_dgrad(View(dyd3), w3, 2, 1, gd3)
side = torch.cuda.Stream()
ref = torch.empty(4, 4, 128, 256, device='cuda')
fn(ref); nb(); torch.cuda.synchronize()
out = torch.empty_like(ref)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for q in range(4): nb()
fn(out); torch.cuda.synchronize()
D = (out - ref).double().reshape(16 * 128, 256)
nz = (D != 0)
print('differing', int(nz.sum()))
rows = nz.any(1).nonzero().flatten().tolist()
cols = nz.any(0).nonzero().flatten().tolist()
print('rows', rows[:40], '...', len(rows))
print('cols', cols[:10], '...', len(cols), 'min', min(cols), 'max', max(cols))
# im2col of the gathered side: A[p][(tap, c)]
xp = F.pad(x3.double().permute(0, 3, 1, 2), (1, 1, 1, 1))
colsA = F.unfold(xp, 4, stride=2)                        # [N, c*16, P]  index c*16 + tap
colsA = colsA.reshape(N, 128, 16, -1).permute(0, 3, 2, 1).reshape(-1, 16 * 128)    # [pixels, tap*128 + c]
B = dy3.double().reshape(-1, 256)
r0 = rows[0] // 128 * 128
c0 = cols[0] // 128 * 128
Dt = D[r0:r0 + 128, c0:c0 + 128]
At, Bt = colsA[:, r0:r0 + 128], B[:, c0:c0 + 128]
# error as a combination of per-K-step (32 pixels) contributions restricted to the corrupted rows
rsel = torch.tensor([x - r0 for x in rows if r0 <= x < r0 + 128], device='cuda')
best = []
for kt in range(At.shape[0] // 32):
    G = At[kt * 32:(kt + 1) * 32, rsel].t() @ Bt[kt * 32:(kt + 1) * 32]
    num = float((Dt[rsel] * G).sum()); den = float((G * G).sum())
    best.append((abs(num) / (den ** 0.5 * float(Dt[rsel].norm()) + 1e-30), kt, num / (den + 1e-30)))
best.sort(reverse=True)
print('tile rows %d.. cols %d..: |D| %.3e; best-correlated K steps (corr, kt, coefficient):' % (r0, c0, float(Dt.norm())), [(round(a, 3), k, round(c, 3)) for a, k, c in best[:6]])
