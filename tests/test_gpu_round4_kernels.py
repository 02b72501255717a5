"""Round-4 kernels and entry points against torch restatements (through the C ABI):
  ssc_conv_forward_bnbwd2   norm-backward sums of BOTH halves of a merged decoder data gradient (models_collection.py:512-531)
  ssc_bn_bwd_sums / _apply   the norm backward in two steps
  slab_reduce4_stats_kernel batch statistics taken by the split-K slab sum
  ssc_block_out_backward    the backward through a bottleneck's output (residual_util.py:103-109, 138-146, 165-167)
  ssc_conv_forward_minmax   the MRU gates' per-sample extrema out of the conv epilogue (mru.py:407-415)"""
import pytest
import torch

from oracle import tf_ops as T

pytestmark = pytest.mark.gpu


def _hip():
    from sketchyscenecolorization_amd import hip
    return hip


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def close(a, b, tol=2e-4):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max()) / scale
    assert err < tol, err


def rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * std


def test_merged_decoder_dgrad_delivers_both_sites_sums():
    """decoder_k reads relu(concat[norm(d), norm(e)]): one data-gradient launch over both column ranges, the sums of both norms'
    backward out of its epilogue -- against autograd on the oracle ops and against the separate passes."""
    hip = _hip()
    n, h, c0, c1, co = 16, 24, 128, 128, 64
    xd_ = (rnd(n, c0, h, h, seed=1) * 1.3 + 0.2).requires_grad_(True)
    xe_ = (rnd(n, c1, h, h, seed=2) * 0.7 - 0.1).requires_grad_(True)
    sd, od = (1.0 + 0.1 * rnd(c0, seed=3)).requires_grad_(True), (0.1 * rnd(c0, seed=4)).requires_grad_(True)
    se, oe = (1.0 + 0.1 * rnd(c1, seed=5)).requires_grad_(True), (0.1 * rnd(c1, seed=6)).requires_grad_(True)
    f = rnd(4, 4, co, c0 + c1, seed=7, std=0.05)
    y = torch.relu(torch.cat([T.batchnorm(xd_, sd, od), T.batchnorm(xe_, se, oe)], 1))
    o = T.conv2d_transpose_same_s2(y, f)
    dy = rnd(*o.shape, seed=8)
    (o * dy).sum().backward()
    xs = [nhwc(xd_.detach()).cuda(), nhwc(xe_.detach()).cuda()]
    tabs = []
    for x, sc, of in ((xs[0], sd, od), (xs[1], se, oe)):
        c = x.shape[-1]
        ab, st = torch.empty(2 * c, device='cuda'), torch.empty(2 * c, device='cuda')
        hip.bn_stats(x.view(-1, c), sc.detach().cuda(), of.detach().cuda(), ab, st)
        tabs.append((ab, st))
    sums = [hip.BnBwdSums(x.view(-1, x.shape[-1]), ab, st,
                          torch.zeros(hip.BnBwdSums.rows_needed(n * h * h), 2 * x.shape[-1], device='cuda'))
            for x, (ab, st) in zip(xs, tabs)]
    g01 = torch.full((n, h, h, c0 + c1), float('nan'), device='cuda')
    hip.deconv_dgrad(hip.View(nhwc(dy).cuda()), f.cuda(), g01, n_off=0, nn=c0 + c1, bnbwd=[sums[0].take(1), sums[1].take(1)])
    assert all(sm.sources == 1 and sm.missed == 0 and sm.rows > 0 for sm in sums)       # both came out of the epilogue
    r01 = g01.view(-1, c0 + c1)
    for k, (x, ref_x, ref_s, ref_o) in enumerate(((xs[0], xd_, sd, od), (xs[1], xe_, se, oe))):
        c = x.shape[-1]
        g = r01[:, :c0] if k == 0 else r01[:, c0:]
        outs = []
        for pre in (sums[k], None):
            dx = torch.full((n * h * h, c), float('nan'), device='cuda')
            ds, do = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
            hip.bn_act_backward(x.view(-1, c), tabs[k][0], tabs[k][1], g, 1, dx, dscale=ds, doffset=do, pre=pre)
            outs.append((dx, ds, do))
            close(nchw(dx.view(n, h, h, c)), ref_x.grad, tol=5e-4)
            close(ds, ref_s.grad, tol=5e-4)
            close(do, ref_o.grad, tol=5e-4)
        for a, b in zip(*outs):
            close(a, b, tol=1e-4)


@pytest.mark.parametrize('c,two', [(128, True), (64, False)])
def test_norm_backward_in_two_steps(c, two):
    """ssc_bn_bwd_sums + ssc_bn_bwd_apply == ssc_bn_act_backward (same bits)."""
    hip = _hip()
    n, h, co = 4, 48, 2 * c
    dev = 'cuda'
    x = (rnd(n, h, h, c, seed=11) * 1.2 + 0.1).to(dev)
    g1, g2 = rnd(n, h, h, c, seed=12).to(dev), rnd(n, h, h, c, seed=13).to(dev)
    scale, offset = (1.0 + 0.1 * rnd(c, seed=14)).to(dev), (0.1 * rnd(c, seed=15)).to(dev)
    ab, st = torch.empty(2 * c, device=dev), torch.empty(2 * c, device=dev)
    x2d = x.view(-1, c)
    hip.bn_stats(x2d, scale, offset, ab, st)
    g2r = g2.view(-1, c) if two else None
    ref_dx = torch.full_like(x2d, float('nan'))
    ds0, do0 = torch.empty(c, device=dev), torch.empty(c, device=dev)
    hip.bn_act_backward(x2d, ab, st, g1.view(-1, c), 2, ref_dx, g2=g2r, act2=1, dscale=ds0, doffset=do0)
    # torch restatement
    z = ab[:c] * x2d + ab[c:]
    dz = g1.view(-1, c) * torch.where(z > 0, 1.0, 0.2) + (g2r * (z > 0).float() if two else 0.0)
    xh = (x2d - st[:c]) * st[c:]
    want = ab[:c] * (dz - dz.mean(0) - xh * (dz * xh).mean(0))
    close(ref_dx, want, tol=2e-5)
    # two steps
    dx = torch.full_like(x2d, float('nan'))
    coef = torch.zeros(2 * c, device=dev)
    ds, do = torch.empty(c, device=dev), torch.empty(c, device=dev)
    job = hip.bn_act_backward(x2d, ab, st, g1.view(-1, c), 2, dx, g2=g2r, act2=1, dscale=ds, doffset=do, defer=True, coef=coef)
    hip.apply_now(job)
    assert job.done
    assert torch.equal(dx, ref_dx) and torch.equal(ds, ds0) and torch.equal(do, do0)


def test_batch_statistics_from_the_slab_sum():
    """A conv with few rows and a long K is cut into split-K slabs; its batch statistics then come out of the slab sum
    (slab_reduce4_stats_kernel) instead of a pass of their own: output bits unchanged, (a, b) and (mean, 1/std) as torch's."""
    hip = _hip()
    n, h, ci, co = 2, 12, 512, 512
    x = rnd(n, ci, h, h, seed=21)
    w = rnd(4, 4, ci, co, seed=22, std=0.02)
    scale, offset = 1.0 + 0.1 * rnd(co, seed=23), 0.1 * rnd(co, seed=24)
    xv = hip.View(nhwc(x).cuda(), None, None, 2)
    plain = torch.full((n, 6, 6, co), float('nan'), device='cuda')
    hip.conv_forward(xv, w.cuda(), 2, 1, plain)
    out = torch.full((n, 6, 6, co), float('nan'), device='cuda')
    ab, st = torch.empty(2 * co, device='cuda'), torch.empty(2 * co, device='cuda')
    hip.conv_forward(xv, w.cuda(), 2, 1, out, bn=(scale.cuda(), offset.cuda(), ab, st))
    assert torch.equal(out, plain)
    o2 = plain.view(-1, co).double()
    mean, var = o2.mean(0), o2.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    close(st[:co], mean, tol=1e-5)
    close(st[co:], rstd, tol=1e-5)
    close(ab[:co], rstd * scale.cuda().double(), tol=1e-5)
    close(ab[co:], offset.cuda().double() - mean * rstd * scale.cuda().double(), tol=1e-5)


@pytest.mark.parametrize('two,act', [(True, 2), (True, 1), (False, 1)])
def test_block_output_backward(two, act):
    """out = act(norm(r3) + shortcut): ssc_block_out_backward against autograd through the same expression."""
    hip = _hip()
    m, c = 3 * 24 * 24, 128
    dev = 'cuda'
    r3 = (rnd(m, c, seed=31) * 1.1 + 0.2).to(dev).requires_grad_(True)
    sc = (rnd(m, c, seed=32) * 0.9 - 0.3).to(dev).requires_grad_(True)
    s3, o3 = (1.0 + 0.1 * rnd(c, seed=33)).to(dev).requires_grad_(True), (0.1 * rnd(c, seed=34)).to(dev).requires_grad_(True)
    ss, os_ = (1.0 + 0.1 * rnd(c, seed=35)).to(dev).requires_grad_(True), (0.1 * rnd(c, seed=36)).to(dev).requires_grad_(True)

    def bn(x, s, o):
        mu, var = x.mean(0), x.var(0, unbiased=False)
        return (x - mu) / torch.sqrt(var + 1e-5) * s + o

    pre = bn(r3, s3, o3) + (bn(sc, ss, os_) if two else sc)
    out = torch.relu(pre) if act == 1 else torch.maximum(pre, 0.2 * pre)
    g = rnd(m, c, seed=37).to(dev)
    (out * g).sum().backward()
    tabs = []
    for x, s, o in ((r3, s3, o3), (sc, ss, os_)):
        ab, st = torch.empty(2 * c, device=dev), torch.empty(2 * c, device=dev)
        hip.bn_stats(x.detach(), s.detach(), o.detach(), ab, st)
        tabs.append((ab, st))
    dz = torch.full((m, c), float('nan'), device=dev)
    dxa = torch.full((m, c), float('nan'), device=dev)
    dxb = torch.full((m, c), float('nan'), device=dev) if two else None
    gs = [torch.empty(c, device=dev) for _ in range(4)]
    coef = torch.zeros(3 * c, device=dev)
    ws = hip.workspace()
    hip.call('ssc_block_out_backward', out.detach().contiguous(), g, m, c, act, r3.detach(), tabs[0][0], tabs[0][1],
             sc.detach() if two else None, tabs[1][0] if two else None, tabs[1][1] if two else None, dz, dxa, dxb,
             gs[0], gs[1], gs[2] if two else None, gs[3] if two else None, coef, ws, ws.numel() * 4)
    close(dxa, r3.grad, tol=5e-5)
    close(gs[0], s3.grad, tol=5e-5)
    close(gs[1], o3.grad, tol=5e-5)
    if two:
        close(dxb, sc.grad, tol=5e-5)
        close(gs[2], ss.grad, tol=5e-5)
        close(gs[3], os_.grad, tol=5e-5)
    else:
        close(dz, sc.grad, tol=5e-5)       # identity shortcut: its gradient is dz itself


@pytest.mark.parametrize('n,h,ci,co', [(3, 48, 128, 128), (2, 96, 64, 64), (2, 12, 64, 128)])
def test_gate_extrema_from_the_conv_epilogue(n, h, ci, co):
    """conv 3x3 SAME + bias + lrelu, then reduce_min / reduce_max over the positions of every sample and channel (mru.py:
    407-415): ssc_conv_forward_minmax (extrema out of the epilogue where a sample's positions are whole tiles; the 12 x 12 case
    takes the separate pass) against torch, and the conv output against the plain launch bit for bit."""
    hip = _hip()
    x = rnd(n, h, h, ci, seed=41).cuda()
    w = rnd(3, 3, ci, co, seed=42, std=0.05).cuda()
    b = (0.1 * rnd(co, seed=43)).cuda()
    plain = torch.full((n, h, h, co), float('nan'), device='cuda')
    hip.conv_forward(hip.View(x), w, 1, 0, plain, bias=b, epi=2, same=True)
    out = torch.full((n, h, h, co), float('nan'), device='cuda')
    mm = torch.full((n, 2, co), float('nan'), device='cuda')
    hip.conv_forward(hip.View(x), w, 1, 0, out, bias=b, epi=2, same=True, minmax=mm)
    assert torch.equal(out, plain)
    flat = plain.view(n, h * h, co)
    assert torch.equal(mm[:, 0], flat.amin(1)) and torch.equal(mm[:, 1], flat.amax(1))


@pytest.mark.parametrize('m_hw,k,act,with_bn', [((3, 40, 48), 32, 2, True), ((4, 24, 24), 128, 1, True), ((5, 37, 21), 16, 0, False),
                                                 ((2, 48, 48), 64, 2, True), ((6, 40, 40), 16, 1, True)])
def test_bottleneck_expansion_conv_streaming_kernel(m_hw, k, act, with_bn):
    """The 1x1 expansion conv of the bottleneck blocks (C/4 -> C, residual_util.py:97-101) on the streaming kernel of pw1x1.hip
    (filter in registers, persistent workgroups, norm + activation on load, batch statistics as per-lane sums): against
    torch in float64, a ragged last tile included."""
    hip = _hip()
    n, h, w_ = m_hw
    co = 4 * k
    dev = 'cuda'
    x = rnd(n, h, w_, k, seed=51).to(dev)
    wt = rnd(1, 1, k, co, seed=52, std=0.1).to(dev)
    ab = torch.cat([1.0 + 0.2 * rnd(k, seed=53), 0.3 * rnd(k, seed=54)]).to(dev)
    d = hip.ConvDesc()      # the launch must be the streaming kernel's
    xv = hip.View(x, None, ab, act)
    out = torch.full((n, h, w_, co), float('nan'), device=dev)
    scale, offset = (1.0 + 0.1 * rnd(co, seed=55)).to(dev), (0.1 * rnd(co, seed=56)).to(dev)
    a2, s2 = torch.empty(2 * co, device=dev), torch.empty(2 * co, device=dev)
    hip.conv_forward(xv, wt, 1, 0, out, bn=(scale, offset, a2, s2) if with_bn else None)
    z = (ab[:k] * x + ab[k:]).double()
    z = torch.relu(z) if act == 1 else (torch.maximum(z, 0.2 * z) if act == 2 else z)
    ref = z.view(-1, k) @ wt.view(k, co).double()
    close(out.view(-1, co), ref, tol=2e-5)
    if with_bn:
        o2 = out.view(-1, co).double()
        mean, var = o2.mean(0), o2.var(0, unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        close(s2[:co], mean, tol=1e-5)
        close(s2[co:], rstd, tol=1e-5)
        close(a2[:co], rstd * scale.double(), tol=1e-5)
        close(a2[co:], offset.double() - mean * rstd * scale.double(), tol=1e-5)
    # the dispatcher really took the streaming kernel for this shape
    import ctypes as C
    KH, KW, ci, coo = wt.shape
    d.x = xv.c()
    d.w, d.out = wt.data_ptr(), out.data_ptr()
    d.NB, d.PH, d.PW, d.TH, d.TW, d.in_stride, d.nphase = n, h, w_, 1, 1, 1, 1
    d.kstep, d.KH, d.KW, d.wC0, d.wC1, d.k_real = 1, 1, 1, k, co, k
    d.Nn, d.Nstore, d.OH, d.OW, d.ldc, d.out_stride = co, co, h, w_, co, 1
    assert hip.lib().ssc_conv_pw1x1_supported(C.byref(d)) == 1


@pytest.mark.parametrize('shape,c,act', [((2, 96, 96), 16, 1), ((3, 48, 64), 32, 2), ((1, 131, 137), 16, 0), ((5, 61, 70), 32, 1),
                                             ((8, 48, 48), 32, 1), ((2, 90, 125), 16, 2), ((2, 90, 125), 32, 2)])
def test_bottleneck_3x3_conv_streaming_kernel(shape, c, act):
    """The 3x3 conv of the bottleneck blocks at 16 / 32 channels (residual_util.py:92-96) on the streaming kernel of c3x3.hip:
    forward (norm + activation on load, zero padding of the ACTIVATED tensor, batch statistics of the output as per-lane sums)
    against torch in float64, and its data gradient (flipped NK filter) with the norm-backward sums out of the epilogue
    against the separate pass.  Both tile shapes (4 x 32; 8 x 16 where it wastes fewer pixels), ragged tiles included."""
    import ctypes as C
    import torch.nn.functional as F
    hip = _hip()
    n, h, w_ = shape
    dev = 'cuda'
    x = rnd(n, h, w_, c, seed=61).to(dev)
    wt = rnd(3, 3, c, c, seed=62, std=0.1).to(dev)
    ab = torch.cat([1.0 + 0.2 * rnd(c, seed=63), 0.3 * rnd(c, seed=64)]).to(dev)
    xv = hip.View(x, None, ab, act)
    out = torch.full((n, h, w_, c), float('nan'), device=dev)
    scale, offset = (1.0 + 0.1 * rnd(c, seed=65)).to(dev), (0.1 * rnd(c, seed=66)).to(dev)
    a2, s2 = torch.empty(2 * c, device=dev), torch.empty(2 * c, device=dev)
    hip.conv_forward(xv, wt, 1, 0, out, same=True, bn=(scale, offset, a2, s2))
    z = (ab[:c] * x + ab[c:]).double()
    z = torch.relu(z) if act == 1 else (torch.maximum(z, 0.2 * z) if act == 2 else z)
    ref = nhwc(F.conv2d(nchw(z), wt.double().permute(3, 2, 0, 1), padding=1))
    close(out, ref, tol=2e-5)
    plain = torch.full_like(out, float('nan'))
    hip.conv_forward(xv, wt, 1, 0, plain, same=True)
    assert torch.equal(plain, out)
    o2 = out.view(-1, c).double()
    mean, var = o2.mean(0), o2.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    close(s2[:c], mean, tol=1e-5)
    close(s2[c:], rstd, tol=1e-5)
    close(a2[:c], rstd * scale.double(), tol=1e-5)
    close(a2[c:], offset.double() - mean * rstd * scale.double(), tol=1e-5)
    # data gradient of the same conv: dx = conv(dy, flipped filter); the sums of the backward of the norm of x from the epilogue
    dy = rnd(n, h, w_, c, seed=67).to(dev)
    st = torch.empty(2 * c, device=dev)
    abx = torch.empty(2 * c, device=dev)
    x2d = x.view(-1, c)
    hip.bn_stats(x2d, scale, offset, abx, st)
    sums = hip.BnBwdSums(x2d, abx, st, torch.zeros(hip.BnBwdSums.rows_needed(n * h * w_), 2 * c, device=dev))
    g = torch.full((n, h, w_, c), float('nan'), device=dev)
    hip.conv_dgrad(hip.View(dy), wt, 1, 1, g, bnbwd=sums.take(max(act, 1)))
    want_g = nhwc(F.conv_transpose2d(nchw(dy).double(), wt.double().permute(3, 2, 0, 1), padding=1))
    close(g, want_g, tol=2e-5)
    if n * h * w_ >= 16384:
        assert sums.sources == 1 and sums.missed == 0 and 0 < sums.rows <= 3 * 256
    outs = []
    for pre in (sums, None):
        dx = torch.full((n * h * w_, c), float('nan'), device=dev)
        ds, do = torch.empty(c, device=dev), torch.empty(c, device=dev)
        hip.bn_act_backward(x2d, abx, st, g.view(-1, c), max(act, 1), dx, dscale=ds, doffset=do, pre=pre)
        outs.append((dx, ds, do))
    for a, b in zip(*outs):
        close(a, b, tol=1e-4)
    # the dispatcher really took the streaming kernel for the forward shape
    d = hip.ConvDesc()
    d.x = xv.c()
    d.w, d.out = wt.data_ptr(), out.data_ptr()
    d.NB, d.PH, d.PW, d.TH, d.TW, d.in_stride, d.nphase = n, h, w_, 3, 3, 1, 1
    d.ioff_y = d.ioff_x = -1
    d.kstep, d.KH, d.KW, d.wC0, d.wC1, d.k_real = 1, 3, 3, c, c, c
    d.Nn, d.Nstore, d.OH, d.OW, d.ldc, d.out_stride = c, c, h, w_, c, 1
    assert hip.lib().ssc_conv_c3x3_supported(C.byref(d)) == (1 if n * h * w_ >= 16384 else 0)


@pytest.mark.parametrize('shape,co,act,stride', [((8, 128, 128), 16, 2, 2), ((9, 126, 132), 16, 1, 2), ((8, 128, 128), 8, 0, 2),
                                                  ((4, 96, 96), 16, 1, 1), ((4, 101, 107), 16, 2, 1)])
def test_4x4_conv_to_16_channels_on_the_16_column_mfma(shape, co, act, stride):
    """The 4x4 conv 64 -> 16 (stride 1: block_1 of the plain bottlenecks at the highest resolution, conv_ex SAME,
    residual_util.py:147-151; stride 2) on v_mfma_f32_16x16x4_f32 with K split over the four wavefronts (s2n16.hip): norm +
    activation on load, zero padding of the activated tensor (1 before, 2 after at stride 1), batch statistics as per-thread
    sums, ragged tiles; against torch in float64 and bit for bit against the launch without statistics."""
    import ctypes as C
    import torch.nn.functional as F
    hip = _hip()
    n, h, w_ = shape
    dev = 'cuda'
    c = 64
    oh, ow = h // stride, w_ // stride
    x = rnd(n, h, w_, c, seed=81).to(dev)
    wt = rnd(4, 4, c, co, seed=82, std=0.05).to(dev)
    ab = torch.cat([1.0 + 0.2 * rnd(c, seed=83), 0.3 * rnd(c, seed=84)]).to(dev)
    xv = hip.View(x, None, ab, act)
    out = torch.full((n, oh, ow, co), float('nan'), device=dev)
    scale, offset = (1.0 + 0.1 * rnd(co, seed=85)).to(dev), (0.1 * rnd(co, seed=86)).to(dev)
    a2, s2 = torch.empty(2 * co, device=dev), torch.empty(2 * co, device=dev)
    hip.conv_forward(xv, wt, stride, 0, out, same=True, bn=(scale, offset, a2, s2))
    z = (ab[:c] * x + ab[c:]).double()
    z = torch.relu(z) if act == 1 else (torch.maximum(z, 0.2 * z) if act == 2 else z)
    zp = F.pad(nchw(z), (1, 2, 1, 2)) if stride == 1 else F.pad(nchw(z), (1, 1, 1, 1))
    ref = nhwc(F.conv2d(zp, wt.double().permute(3, 2, 0, 1), stride=stride))
    close(out, ref, tol=2e-5)
    plain = torch.full_like(out, float('nan'))
    hip.conv_forward(xv, wt, stride, 0, plain, same=True)
    assert torch.equal(plain, out)
    o2 = out.view(-1, co).double()
    mean, var = o2.mean(0), o2.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    close(s2[:co], mean, tol=1e-5)
    close(s2[co:], rstd, tol=1e-5)
    close(a2[:co], rstd * scale.double(), tol=1e-5)
    close(a2[co:], offset.double() - mean * rstd * scale.double(), tol=1e-5)
    d = hip.ConvDesc()
    d.x = xv.c()
    d.w, d.out = wt.data_ptr(), out.data_ptr()
    d.NB, d.PH, d.PW, d.TH, d.TW, d.in_stride, d.nphase = n, oh, ow, 4, 4, stride, 1
    d.ioff_y = d.ioff_x = -1
    d.kstep, d.KH, d.KW, d.wC0, d.wC1, d.k_real = 1, 4, 4, c, co, c
    d.Nn, d.Nstore, d.OH, d.OW, d.ldc, d.out_stride = co, co, oh, ow, co, 1
    assert hip.lib().ssc_conv_s2n16_supported(C.byref(d)) == 1


def test_transposed_conv_data_gradient_on_the_16_column_mfma():
    """The data gradient of the k = 4 stride-2 transposed conv 16 -> 64 (the last decoder bottleneck) has the geometry of the 4x4
    stride-2 conv 64 -> 16 and runs on s2n16.hip; the two sums of the backward of the norm its output is the gradient of come out
    of its epilogue (ssc_conv_forward_bnbwd): against torch and against the separate pass."""
    import torch.nn.functional as F
    hip = _hip()
    n, h, w_, ci, co = 8, 64, 72, 16, 64
    dev = 'cuda'
    dy = rnd(n, 2 * h, 2 * w_, co, seed=91).to(dev)
    f = rnd(4, 4, co, ci, seed=92, std=0.05).to(dev)
    x = rnd(n, h, w_, ci, seed=93).to(dev)
    scale, offset = (1.0 + 0.1 * rnd(ci, seed=94)).to(dev), (0.1 * rnd(ci, seed=95)).to(dev)
    abx, st = torch.empty(2 * ci, device=dev), torch.empty(2 * ci, device=dev)
    x2d = x.view(-1, ci)
    hip.bn_stats(x2d, scale, offset, abx, st)
    sums = hip.BnBwdSums(x2d, abx, st, torch.zeros(hip.BnBwdSums.rows_needed(n * h * w_), 2 * ci, device=dev))
    g = torch.full((n, h, w_, ci), float('nan'), device=dev)
    hip.deconv_dgrad(hip.View(dy), f, g, bnbwd=sums.take(1))
    want = nhwc(F.conv2d(nchw(dy).double(), f.double().permute(3, 2, 0, 1), stride=2, padding=1))
    close(g, want, tol=2e-5)
    assert sums.sources == 1 and sums.missed == 0 and 0 < sums.rows <= 512
    outs = []
    for pre in (sums, None):
        dx = torch.full((n * h * w_, ci), float('nan'), device=dev)
        ds, do = torch.empty(ci, device=dev), torch.empty(ci, device=dev)
        hip.bn_act_backward(x2d, abx, st, g.view(-1, ci), 1, dx, dscale=ds, doffset=do, pre=pre)
        outs.append((dx, ds, do))
    for a, b in zip(*outs):
        close(a, b, tol=1e-4)


@pytest.mark.parametrize('shape,act', [((2, 96, 96), 1), ((3, 37, 53), 0)])
def test_region_branch_transposed_conv_thread_per_pixel(shape, act):
    """The region branch's k = 4 stride-2 transposed conv 3 -> 3 over a 4-channel padded tensor (bg_colorization_main.py:392-397) on
    tr4tiny.hip: folded norm + activation on load, the padding column written as 0, batch statistics as per-thread sums; against
    torch in float64."""
    import ctypes as C
    import torch.nn.functional as F
    hip = _hip()
    n, h, w_ = shape
    dev = 'cuda'
    x = rnd(n, h, w_, 4, seed=101).to(dev)
    x[..., 3] = 7.0         # the padding channel of the input must not be read
    f = rnd(4, 4, 3, 3, seed=102, std=0.3).to(dev)
    ab = torch.cat([1.0 + 0.2 * rnd(4, seed=103), 0.3 * rnd(4, seed=104)]).to(dev)
    xv = hip.View(x, None, ab, act)
    out = torch.full((n, 2 * h, 2 * w_, 4), float('nan'), device=dev)
    scale, offset = torch.ones(4, device=dev), torch.zeros(4, device=dev)
    a2, s2 = torch.empty(8, device=dev), torch.empty(8, device=dev)
    hip.deconv_forward(xv, f, out, nstore=4, bn=(scale, offset, a2, s2))
    z = (ab[:4] * x + ab[4:]).double()[..., :3]
    z = torch.relu(z) if act == 1 else z
    ref = nhwc(F.conv_transpose2d(nchw(z), f.double().permute(3, 2, 0, 1), stride=2, padding=1))
    close(out[..., :3], ref, tol=2e-5)
    assert torch.all(out[..., 3] == 0)
    o2 = out.view(-1, 4).double()
    mean, var = o2.mean(0), o2.var(0, unbiased=False)
    close(s2[:4], mean, tol=1e-5)
    close(s2[4:], 1.0 / torch.sqrt(var + 1e-5), tol=1e-5)
    d = hip.deconv_forward(xv, f, out, nstore=4, _desc_only=True)
    assert hip.lib().ssc_conv_tr4_tiny_supported(C.byref(d)) == 1


@pytest.mark.parametrize('shape', [(2, 128, 192), (1, 384, 384)])
def test_first_7x7_conv_over_the_image_channels(shape):
    """encoder_1 of the Residual / Background generators: 7x7 stride-2 SAME conv over the 3 image channels (padded to 4) -> 64 + batch
    statistics (fewchan7.hip); against torch in float64 (SAME = 2 before, 3 after)."""
    import ctypes as C
    import torch.nn.functional as F
    hip = _hip()
    n, h, w_ = shape
    dev = 'cuda'
    x = rnd(n, h, w_, 4, seed=111).to(dev)
    x[..., 3] = 5.0         # the padding channel must not be read
    wt = rnd(7, 7, 3, 64, seed=112, std=0.1).to(dev)
    out = torch.full((n, h // 2, w_ // 2, 64), float('nan'), device=dev)
    scale, offset = (1.0 + 0.1 * rnd(64, seed=113)).to(dev), (0.1 * rnd(64, seed=114)).to(dev)
    a2, s2 = torch.empty(128, device=dev), torch.empty(128, device=dev)
    hip.conv_forward(hip.View(x), wt, 2, 0, out, same=True, bn=(scale, offset, a2, s2))
    z = nchw(x[..., :3].double())
    ref = nhwc(F.conv2d(F.pad(z, (2, 3, 2, 3)), wt.double().permute(3, 2, 0, 1), stride=2))
    close(out, ref, tol=2e-5)
    o2 = out.view(-1, 64).double()
    mean, var = o2.mean(0), o2.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    close(s2[:64], mean, tol=1e-5)
    close(s2[64:], rstd, tol=1e-5)
    close(a2[:64], rstd * scale.double(), tol=1e-5)
    d = hip.ConvDesc()
    d.x = hip.View(x).c()
    d.w, d.out = wt.data_ptr(), out.data_ptr()
    d.NB, d.PH, d.PW, d.TH, d.TW, d.in_stride, d.nphase = n, h // 2, w_ // 2, 7, 7, 2, 1
    d.ioff_y = d.ioff_x = -2
    d.kstep, d.KH, d.KW, d.wC0, d.wC1, d.k_real = 1, 7, 7, 3, 64, 3
    d.Nn, d.Nstore, d.OH, d.OW, d.ldc, d.out_stride = 64, 64, h // 2, w_ // 2, 64, 1
    assert hip.lib().ssc_conv_fewchan7_supported(C.byref(d)) == 1


@pytest.mark.parametrize('shape,c0,act', [((8, 48, 48), 128, 1), ((5, 50, 70), 256, 2), ((8, 48, 48), 64, 0)])
def test_transposed_conv_to_16_channels_on_the_16_column_mfma(shape, c0, act):
    """block_1 of the last decoder bottleneck: k = 4 stride-2 transposed conv over concat[128, 128] -> 16 + batch statistics on
    tr4n16.hip (16-column MFMA, one sub-pixel phase per workgroup, K split by tap over the waves): two sources with their own
    folded norms, ragged tiles; against torch in float64 and bit for bit against the launch without statistics."""
    import ctypes as C
    import torch.nn.functional as F
    hip = _hip()
    n, h, w_ = shape
    dev = 'cuda'
    c1 = 256 - c0
    x0 = rnd(n, h, w_, c0, seed=121).to(dev)
    x1 = rnd(n, h, w_, c1, seed=122).to(dev) if c1 else None
    ab0 = torch.cat([1.0 + 0.2 * rnd(c0, seed=123), 0.3 * rnd(c0, seed=124)]).to(dev)
    ab1 = torch.cat([1.0 + 0.2 * rnd(c1, seed=125), 0.3 * rnd(c1, seed=126)]).to(dev) if c1 else None
    f = rnd(4, 4, 16, 256, seed=127, std=0.05).to(dev)
    xv = hip.View(x0, x1, ab0, act, ab1) if c1 else hip.View(x0, None, ab0, act)
    out = torch.full((n, 2 * h, 2 * w_, 16), float('nan'), device=dev)
    scale, offset = (1.0 + 0.1 * rnd(16, seed=128)).to(dev), (0.1 * rnd(16, seed=129)).to(dev)
    a2, s2 = torch.empty(32, device=dev), torch.empty(32, device=dev)
    hip.deconv_forward(xv, f, out, bn=(scale, offset, a2, s2))
    zs = [(ab0[:c0] * x0 + ab0[c0:]).double()] + ([(ab1[:c1] * x1 + ab1[c1:]).double()] if c1 else [])
    z = torch.cat(zs, -1)
    z = torch.relu(z) if act == 1 else (torch.maximum(z, 0.2 * z) if act == 2 else z)
    ref = nhwc(F.conv_transpose2d(nchw(z), f.double().permute(3, 2, 0, 1), stride=2, padding=1))
    close(out, ref, tol=2e-5)
    plain = torch.full_like(out, float('nan'))
    hip.deconv_forward(xv, f, plain)
    assert torch.equal(plain, out)
    o2 = out.view(-1, 16).double()
    mean, var = o2.mean(0), o2.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    close(s2[:16], mean, tol=1e-5)
    close(s2[16:], rstd, tol=1e-5)
    close(a2[:16], rstd * scale.double(), tol=1e-5)
    d = hip.deconv_forward(xv, f, out, _desc_only=True)
    assert hip.lib().ssc_conv_tr4n16_supported(C.byref(d)) == 1


@pytest.mark.parametrize('k,ci,shape,act,acc', [(3, 16, (4, 96, 96), 1, False), (3, 16, (3, 101, 119), 2, True),
                                                (4, 64, (4, 96, 96), 1, False), (4, 64, (3, 101, 119), 0, True)])
def test_filter_gradients_with_16_output_channels(k, ci, shape, act, acc):
    """dW of the bottlenecks' 3x3 conv 16 -> 16 and 4x4 stride-1 conv 64 -> 16 (residual_util.py:92-96, 147-151) on the 16-column MFMA
    (wgn16.hip): folded norm + activation of the gathered tensor on load, SAME padding (1 before), ragged tiles, the accumulate of a
    second pass; against torch autograd in float64."""
    import ctypes as C
    import torch.nn.functional as F
    hip = _hip()
    n, h, w_ = shape
    dev = 'cuda'
    co = 16
    x = rnd(n, h, w_, ci, seed=131).to(dev)
    dy = rnd(n, h, w_, co, seed=132).to(dev)
    ab = torch.cat([1.0 + 0.2 * rnd(ci, seed=133), 0.3 * rnd(ci, seed=134)]).to(dev)
    xv = hip.View(x, None, ab, act)
    base = rnd(k, k, ci, co, seed=135).to(dev)
    dw = base.clone() if acc else torch.full((k, k, ci, co), float('nan'), device=dev)
    hip.conv_wgrad(xv, hip.View(dy), dw, 1, 1, accumulate=acc)
    z = (ab[:ci] * x + ab[ci:]).double()
    z = torch.relu(z) if act == 1 else (torch.maximum(z, 0.2 * z) if act == 2 else z)
    zp = F.pad(nchw(z), (1, k - 2, 1, k - 2))
    wt = torch.zeros(co, ci, k, k, dtype=torch.float64, device=dev, requires_grad=True)
    (F.conv2d(zp, wt) * nchw(dy).double()).sum().backward()
    want = wt.grad.permute(2, 3, 1, 0) + (base.double() if acc else 0.0)
    scale = float(want.abs().max())
    assert float((dw.double() - want).abs().max()) < 2e-5 * scale
    d = hip.WgradDesc()
    d.g, d.d = xv.c(), hip.View(dy).c()
    d.out = dw.data_ptr()
    d.NB, d.PH, d.PW = n, h, w_
    d.TH, d.TW, d.in_stride, d.ioff_y, d.ioff_x = k, k, 1, -1, -1
    d.Cg_real, d.Nn, d.ldc, d.accumulate = ci, co, co, int(acc)
    assert hip.lib().ssc_conv_wgn16_supported(C.byref(d)) == 1
