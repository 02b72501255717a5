"""The MRU pointwise / per-sample reduction kernels (csrc/mru_ops.hip) one by one against plain torch restatements
(float64 on the CPU) of the expressions in Foreground_Instance_Colorization/obj_lib/mru.py:15-28, 405-422, 560-591 and
models_collection.py:56-65.  Every op runs at a channel count that takes the 16-byte forms (`*_v4`, C % 4 == 0) and at one
that takes the scalar forms (C = 6): the two code paths must agree with the same reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _hip():
    from sketchyscenecolorization_amd import hip
    hip.lib()
    return hip


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def close(a, b, tol=TOL):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
    assert err < tol, err


def miu(v):
    return 0.5 * (v + torch.sqrt(0.09 + v * v))


def up2(x):      # nearest 2x upsample of NHWC
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def minmax(x):   # [N,H,W,C] -> [N,2,C]
    return torch.stack([x.amin(dim=(1, 2)), x.amax(dim=(1, 2))], 1).contiguous()


def norm01(g, mm):
    mn, mx = mm[:, 0][:, None, None, :], mm[:, 1][:, None, None, :]
    return (g - mn) / (mx - mn)


CS = [8, 6, 36]      # 16-byte form with one group per row, scalar form, 16-byte form with 9 groups


@pytest.mark.parametrize('C', CS)
def test_pools_and_minmax(C):
    hip = _hip()
    N, H, W = 3, 10, 6
    x = rnd(N, H, W, C, seed=1)
    ref = 0.25 * (x[:, 0::2, 0::2] + x[:, 1::2, 0::2] + x[:, 0::2, 1::2] + x[:, 1::2, 1::2])
    xd = x.cuda()
    out = torch.full((N, H // 2, W // 2, C), float('nan'), device='cuda')
    hip.call('ssc_mean_pool2', xd, C, out, C, N, H, W, C)
    close(out, ref)
    # pool2: scale and accumulate, a channel slice of a wider source into a wider destination
    wide = rnd(N, H, W, C + 4, seed=2).cuda()
    dst = rnd(N, H // 2, W // 2, C + 8, seed=3).cuda()
    ref2 = dst.cpu().clone()
    ref2[..., :C] += 4.0 * 0.5 * (wide.cpu()[..., :C][:, 0::2, 0::2] + wide.cpu()[..., :C][:, 1::2, 0::2] +
                                  wide.cpu()[..., :C][:, 0::2, 1::2] + wide.cpu()[..., :C][:, 1::2, 1::2]) / 4.0
    hip.call('ssc_pool2', wide, C + 4, dst, C + 8, N, H, W, C, 0.5, 1)
    close(dst, ref2)
    mm = torch.empty(N, 2, C, device='cuda')
    hip.minmax_hw(xd, mm)
    assert torch.equal(mm.cpu(), minmax(x))


@pytest.mark.parametrize('C', CS)
def test_concat_parts_forms(C):
    """[gate * up(ht) | image (3 of 4) | act(a[n]*skip + b[n])]: upsample, min-max gate, the 3-channel part that shifts everything
    behind it off the 16-byte grid, per-sample norm tables + miu_relu; then the prelu and the plain single-part forms."""
    hip = _hip()
    N, H, W = 2, 8, 6
    ht, z, skip, rg = rnd(N, H // 2, W // 2, C, seed=4), rnd(N, H, W, 4, seed=5), rnd(N, H, W, C, seed=6), rnd(N, H, W, C, seed=7)
    abn = torch.cat([1.0 + 0.1 * rnd(N, C, seed=8), 0.2 * rnd(N, C, seed=9)], 1).contiguous()
    mm = minmax(rg)
    ct = 2 * C + 3
    ld = (ct + 3) // 4 * 4
    ref = torch.zeros(N, H, W, ld)
    ref[..., :C] = up2(ht) * norm01(rg, mm)
    ref[..., C:C + 3] = z[..., :3]
    a, b = abn[:, :C][:, None, None, :], abn[:, C:][:, None, None, :]
    ref[..., C + 3:ct] = miu(a * skip + b)
    out = torch.zeros(N, H, W, ld, device='cuda')
    hip.concat_parts(out, [dict(x=ht.cuda(), upsample=True, gate=(rg.cuda(), mm.cuda())), dict(x=z.cuda(), C=3),
                           dict(x=skip.cuda(), ab=abn.cuda(), act=hip.ACT_MIU)])
    close(out, ref)
    leak = torch.tensor([0.25])
    o2 = torch.empty(N, H, W, C, device='cuda')
    hip.concat_parts(o2, [dict(x=skip.cuda(), ab=leak.cuda(), act=hip.ACT_PRELU)])
    close(o2, torch.maximum(0.25 * skip, skip))


@pytest.mark.parametrize('C', CS)
@pytest.mark.parametrize('proj', [True, False])
def test_gate_merge_blend_and_their_backward(C, proj):
    hip = _hip()
    N, H, W = 2, 8, 6
    # ht_plus = ht + r * img (mru.py:422) and its backward
    ht, rg, img = (rnd(N, H, W, C, seed=s).double().requires_grad_(True) for s in (10, 11, 12))
    mm = minmax(rg.detach().float())
    r = norm01(rg, mm.double())
    htp = ht + r.detach() * img                  # the kernels take r as given: its gradient is the gate backward's business
    out = torch.empty(N, H, W, C, device='cuda')
    f = lambda t: t.detach().float().cuda()
    hip.call('ssc_mru_gate_merge', f(ht), f(rg), mm.cuda(), f(img), out, N, H * W, C)
    close(out, htp)
    g = rnd(N, H, W, C, seed=13)
    gr, gimg = torch.empty_like(out), torch.empty_like(out)
    hip.call('ssc_mru_gate_merge_backward', g.cuda(), f(rg), mm.cuda(), f(img), gr, gimg, N, H * W, C)
    close(gr, g.double() * img.detach())
    close(gimg, g.double() * r.detach())
    # the min-max gate's backward: r = (lrelu(pre) - min) / (max - min), reduce_min / reduce_max included
    pre = rnd(N, H, W, C, seed=14).double().requires_grad_(True)
    gv = torch.maximum(pre, 0.2 * pre)
    rr = (gv - gv.amin(dim=(1, 2), keepdim=True)) / (gv.amax(dim=(1, 2), keepdim=True) - gv.amin(dim=(1, 2), keepdim=True))
    (rr * g.double()).sum().backward()
    dpre = torch.empty_like(out)
    ws = hip.workspace()
    hip.call('ssc_minmax_gate_backward', f(gv), minmax(gv.detach().float()).cuda(), g.cuda(), N, H * W, C, dpre, ws,
             ws.numel() * 4)
    close(dpre, pre.grad, tol=2e-4)
    # blend (mru.py:583-589): out = hp * (1 - z) + h * z, ht at half resolution, with / without the projected + normed ht
    htl, h2, zg = rnd(N, H // 2, W // 2, C, seed=15), rnd(N, H, W, C, seed=16), rnd(N, H, W, C, seed=17)
    ab1 = torch.cat([1.0 + 0.1 * rnd(N, C, seed=18), 0.2 * rnd(N, C, seed=19)], 1).contiguous()
    ab2 = torch.cat([1.0 + 0.1 * rnd(N, C, seed=20), 0.2 * rnd(N, C, seed=21)], 1).contiguous()
    mz = minmax(zg)
    bc = lambda t, lo, hi: t[:, lo:hi][:, None, None, :]
    hp = up2(htl)
    if proj:
        hp = miu(bc(ab1, 0, C) * hp + bc(ab1, C, 2 * C))
    h = miu(bc(ab2, 0, C) * h2 + bc(ab2, C, 2 * C))
    zz = norm01(zg, mz)
    bo = torch.empty_like(out)
    hip.call('ssc_mru_blend', htl.cuda(), ab1.cuda() if proj else None, 1, h2.cuda(), ab2.cuda(), zg.cuda(), mz.cuda(), bo,
             N, H, W, C)
    close(bo, hp * (1 - zz) + h * zz)
    ghp, gh, gz = torch.empty_like(out), torch.empty_like(out), torch.empty_like(out)
    hip.call('ssc_mru_blend_backward', g.cuda(), htl.cuda(), ab1.cuda() if proj else None, 1, h2.cuda(), ab2.cuda(), zg.cuda(),
             mz.cuda(), ghp, gh, gz, N, H, W, C)
    close(ghp, g * (1 - zz))
    close(gh, g * zz)
    close(gz, g * (h - hp))
    # [r * up(ht) | ...] backward: gr = G * up(ht), G[:, :C] *= r in place
    ldG = C + 8
    G = rnd(N, H, W, ldG, seed=22)
    Gd = G.cuda()
    gr2 = torch.empty_like(out)
    hip.call('ssc_mru_in2_gate_backward', Gd, ldG, zg.cuda(), mz.cuda(), htl.cuda(), gr2, N, H, W, C)
    close(gr2, G[..., :C] * up2(htl))
    refG = G.clone()
    refG[..., :C] = G[..., :C] * zz
    close(Gd, refG)


@pytest.mark.parametrize('C', CS)
def test_prelu_backward_and_strided_copy(C):
    hip = _hip()
    M = 77
    x, gy = rnd(M, C + 2, seed=30), rnd(M, C + 4, seed=31)
    leak = torch.tensor([0.3])
    first = 0.3 * x[:, :C] >= x[:, :C]
    ref_dx = gy[:, :C] * torch.where(first, torch.tensor(0.3), torch.tensor(1.0))
    ref_dl = (gy[:, :C] * x[:, :C])[first].double().sum()
    dx = rnd(M, C, seed=32).cuda()
    base = dx.cpu().clone()
    dleak = torch.zeros(1, device='cuda')
    ws = hip.workspace()
    hip.call('ssc_prelu_backward', x.cuda(), C + 2, leak.cuda(), gy.cuda(), C + 4, M, C, dx, C, 1, dleak, 0, ws, ws.numel() * 4)
    close(dx, base + ref_dx)
    close(dleak, ref_dl.reshape(1), tol=2e-4)
    # a slice that starts at an odd column (a concat gradient behind the 3-channel image part), added to the destination
    src = rnd(M, 2 * C + 3, seed=33)
    dst = rnd(M, C, seed=34).cuda()
    ref = dst.cpu() + src[:, C + 3:]
    hip.call('ssc_strided_copy', src.cuda().view(-1)[C + 3:], 2 * C + 3, dst, C, M, C, 1)
    close(dst, ref)


@pytest.mark.parametrize('C', CS)
def test_cond_norm_backward(C):
    """Backward of y = miu_relu(cond_batchnorm(x)) (models_collection.py:22-35, 63-65): dx through the batch statistics and the
    per-class scale / offset table gradients, against autograd; gy is a channel slice of a wider gradient (row stride ldg)."""
    hip = _hip()
    N, P, L = 4, 35, 5
    x = rnd(N, P, C, seed=40).double().requires_grad_(True)
    scale_m = (1.0 + 0.1 * rnd(L, C, seed=41)).double().requires_grad_(True)
    offset_m = (0.2 * rnd(L, C, seed=42)).double().requires_grad_(True)
    labels = torch.tensor([1, 3, 1, 0], dtype=torch.int32)
    mean = x.mean(dim=(0, 1))
    var = ((x - mean) ** 2).mean(dim=(0, 1))
    rstd = torch.rsqrt(var + 1e-5)
    lab = labels.long()
    y = miu(scale_m[lab][:, None, :] * ((x - mean) * rstd) + offset_m[lab][:, None, :])
    gy = rnd(N, P, C + 3, seed=43)
    (y * gy[..., :C].double()).sum().backward()
    xd = x.detach().float().cuda()
    one, zero = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    ab0, st = torch.empty(2 * C, device='cuda'), torch.empty(2 * C, device='cuda')
    if C % 4 == 0:
        hip.bn_stats(xd.view(-1, C), one, zero, ab0, st)
    else:           # ssc_bn_stats takes multiples of 4 channels: the statistics from the reference
        st = torch.cat([mean.detach().float(), rstd.detach().float()]).cuda()
    abn = torch.empty(N, 2 * C, device='cuda')
    sm, om = scale_m.detach().float().cuda(), offset_m.detach().float().cuda()
    hip.call('ssc_cbn_fold', st, sm, om, labels.cuda(), N, C, abn)
    dx = torch.full((N, P, C), float('nan'), device='cuda')
    ds, do = torch.full((L, C), float('nan'), device='cuda'), torch.full((L, C), float('nan'), device='cuda')
    ws = hip.workspace()
    hip.call('ssc_cbn_act_backward', xd, abn, st, sm, labels.cuda(), L, gy.cuda(), C + 3, hip.ACT_MIU, N, P, C, dx, C, 0, ds, do,
             0, ws, ws.numel() * 4)
    close(dx, x.grad, tol=2e-4)
    close(ds, scale_m.grad, tol=2e-4)
    close(do, offset_m.grad, tol=2e-4)
