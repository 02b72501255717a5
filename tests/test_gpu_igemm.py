"""HIP implicit-GEMM kernels vs the CPU oracle (oracle/tf_ops.py) on seeded inputs.

Tolerances: fp32 MFMA == fp32 fmaf chain; the oracle sums in a different order,
so compare with 1e-4 relative to the output scale (north-star budget is 1e-3).
"""
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


def act_ref(x, act):
    if act == 1:
        return torch.relu(x)
    if act == 2:
        return T.lrelu(x, 0.2)
    return x


def rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * std


@pytest.mark.parametrize('n,h,ci,co,stride,act', [
    (2, 20, 8, 64, 2, 0), (3, 16, 64, 128, 2, 2), (2, 12, 32, 160, 1, 1), (1, 24, 128, 40, 2, 2),
])
def test_conv_forward(n, h, ci, co, stride, act):
    hip = _hip()
    x = rnd(n, ci, h, h, seed=1)
    w = rnd(4, 4, ci, co, seed=2, std=0.05)
    ab = torch.cat([1.0 + 0.1 * rnd(ci, seed=3), 0.2 * rnd(ci, seed=4)])
    ref = T.conv2d_valid_pad(act_ref(x * ab[:ci].view(1, -1, 1, 1) + ab[ci:].view(1, -1, 1, 1), act), w, stride, 1)
    oh = ref.shape[2]
    out = torch.full((n, oh, oh, co), float('nan'), device='cuda')
    hip.conv_forward(hip.View(nhwc(x).cuda(), None, ab.cuda(), act), w.cuda(), stride, 1, out)
    close(nchw(out), ref)


def test_conv_forward_padded_channels_and_cout1():
    hip = _hip()
    n, h = 2, 24
    x3 = rnd(n, 3, h, h, seed=5)
    x4 = torch.zeros(n, h, h, 4)
    x4[..., :3] = nhwc(x3)
    w = rnd(4, 4, 3, 64, seed=6, std=0.05)
    ref = T.conv2d_valid_pad(x3, w, 2, 1)
    out = torch.full((n, 12, 12, 64), float('nan'), device='cuda')
    hip.conv_forward(hip.View(x4.cuda()), w.cuda(), 2, 1, out)
    close(nchw(out), ref)
    # Cout = 1, stride 1, stored into a 4-channel padded tensor
    x = rnd(n, 64, 13, 13, seed=7)
    w1 = rnd(4, 4, 64, 1, seed=8, std=0.05)
    ref1 = T.conv2d_valid_pad(x, w1, 1, 1)
    out1 = torch.full((n, 12, 12, 4), float('nan'), device='cuda')
    hip.conv_forward(hip.View(nhwc(x).cuda()), w1.cuda(), 1, 1, out1, nstore=4)
    close(out1[..., 0], ref1[:, 0])
    assert float(out1[..., 1:].abs().max()) == 0.0
    # the PatchGAN logit conv: few lattice tiles, 512 channels -> channel-split narrow kernel + slab reduce, with bias
    xl = rnd(n, 512, 13, 13, seed=9)
    wl = rnd(4, 4, 512, 1, seed=10, std=0.02)
    bias = torch.tensor([0.25])
    refl = T.conv2d_valid_pad(T.lrelu(xl, 0.2), wl, 1, 1) + 0.25
    outl = torch.full((n, 12, 12, 4), float('nan'), device='cuda')
    hip.conv_forward(hip.View(nhwc(xl).cuda(), None, None, 2), wl.cuda(), 1, 1, outl, nstore=4, bias=bias.cuda())
    close(outl[..., 0], refl[:, 0])
    assert float(outl[..., 1:].abs().max()) == 0.0


@pytest.mark.parametrize('c0,c1,real1,co,k,same,act', [
    (576, 4, 3, 384, 3, True, 1),      # the MRU concat [h, sketch]: 580 channels, 579 real (mru.py:452-470)
    (160, 8, 8, 128, 4, False, 2),     # two sources, neither a multiple of 32
    (200, 0, 0, 64, 3, True, 0),       # one source, last chunk 8 of 32 channels
])
def test_conv_forward_channel_chunks(c0, c1, real1, co, k, same, act):
    """Channel counts that are not multiples of 32 run the chunked uniform-tap kernel (conv_ut_kernel<KMASK>): the K loop
    walks (tap, source, 32-channel chunk), the partly empty last chunk of a source must contribute nothing -- whatever
    the padding channels hold (finite garbage here) and although the filter has no rows for them."""
    hip = _hip()
    n, h = 2, 14
    xa = rnd(n, c0, h, h, seed=21)
    parts = [xa]
    if c1:
        xb = rnd(n, c1, h, h, seed=22)
        parts.append(xb[:, :real1])
    x = torch.cat(parts, 1)
    ci = x.shape[1]
    w = rnd(k, k, ci, co, seed=23, std=0.03)
    ab = torch.cat([1.0 + 0.1 * rnd(c0, seed=24), 0.2 * rnd(c0, seed=25)])
    xin = torch.cat([act_ref(xa * ab[:c0].view(1, -1, 1, 1) + ab[c0:].view(1, -1, 1, 1), act)] + parts[1:], 1)
    if same:
        ref = T.conv2d_same(xin, w, 1)
    else:
        ref = T.conv2d_valid_pad(xin, w, 2, 1)
    oh = ref.shape[2]
    s1 = None
    if c1:
        s1 = nhwc(xb).clone()
        s1[..., real1:] = 1.0e3                 # padding channels: the kernel may not rely on zeros there
        s1 = s1.cuda()
    out = torch.full((n, oh, oh, co), float('nan'), device='cuda')
    v = hip.View(nhwc(xa).cuda(), s1, ab.cuda(), act, None, 0)
    if same:
        hip.conv_forward(v, w.cuda(), 1, 0, out, same=True)
    else:
        hip.conv_forward(v, w.cuda(), 2, 1, out)
    close(nchw(out), ref)


@pytest.mark.parametrize('stride,h', [(2, 26), (1, 17)])
def test_conv_forward_row_tap(stride, h):
    """4x4 conv on a plain 8-channel tensor (the discriminator's first layer, models_collection.py:805-811): TW * C == 32, so
    conv_ut_kernel<KM = 2> runs one filter row per K-tile; the x bound of the zero padding differs per thread."""
    hip = _hip()
    n, ci, co = 3, 8, 64
    x = rnd(n, ci, h, h, seed=51)
    w = rnd(4, 4, ci, co, seed=52, std=0.05)
    ref = T.conv2d_valid_pad(x, w, stride, 1)
    oh = ref.shape[2]
    out = torch.full((n, oh, oh, co), float('nan'), device='cuda')
    hip.conv_forward(hip.View(nhwc(x).cuda()), w.cuda(), stride, 1, out)
    close(nchw(out), ref)


def test_large_launches_tail_split():
    """Launches with more tiles than the chip holds at once (1152-1536 tiles of 64x128 on 768 resident workgroups): the
    partly filled last round is cut into K slices that are combined inside the launch (igemm.hip, tail split), for the plain
    grid, the 4 sub-pixel phases of the transposed conv, and the accumulating data-gradient epilogue.  Reference: the CPU
    oracle (oracle/tf_ops.py) and its autograd."""
    hip = _hip()
    # conv 3x3 SAME, M = 8*96*96 = 73728 -> 1152 row tiles
    x = rnd(8, 32, 96, 96, seed=41)
    w = rnd(3, 3, 32, 128, seed=42, std=0.05)
    ref = T.conv2d_same(torch.relu(x), w, 1)
    out = torch.full((8, 96, 96, 128), float('nan'), device='cuda')
    hip.conv_forward(hip.View(nhwc(x).cuda(), None, None, 1), w.cuda(), 1, 0, out, same=True)
    close(nchw(out), ref)
    # transposed conv k=4 s=2: 4 phases x (8*48*48 / 64) = 1152 tiles
    xd = rnd(8, 64, 48, 48, seed=43)
    f = rnd(4, 4, 128, 64, seed=44, std=0.05)
    refd = T.conv2d_transpose_same_s2(xd, f)
    outd = torch.full((8, 96, 96, 128), float('nan'), device='cuda')
    hip.deconv_forward(hip.View(nhwc(xd).cuda()), f.cuda(), outd)
    close(nchw(outd), refd)
    # data gradient of a stride-1 3x3 SAME conv (128 -> 64 channels), accumulated onto an existing tensor
    w2 = rnd(3, 3, 128, 64, seed=47, std=0.05)
    xin = rnd(8, 128, 96, 96, seed=48).requires_grad_(True)
    y = T.conv2d_same(xin, w2, 1)
    dy = rnd(*y.shape, seed=45)
    y.backward(dy)
    base = rnd(8, 96, 96, 128, seed=46).cuda()
    dx = base.clone()
    hip.conv_dgrad(hip.View(nhwc(dy).cuda()), w2.cuda(), 1, 1, dx, accumulate=True)
    close(nchw(dx - base), xin.grad)
    assert hip.sk_timeouts() == 0


def test_conv_forward_splitk_small_m():
    hip = _hip()
    n, h, ci, co = 2, 12, 512, 512
    x = rnd(n, ci, h, h, seed=9)
    w = rnd(4, 4, ci, co, seed=10, std=0.02)
    ref = T.conv2d_valid_pad(T.lrelu(x, 0.2), w, 2, 1)
    out = torch.full((n, 6, 6, co), float('nan'), device='cuda')
    hip.conv_forward(hip.View(nhwc(x).cuda(), None, None, 2), w.cuda(), 2, 1, out)
    close(nchw(out), ref)


@pytest.mark.parametrize('n,h,c0,c1,co', [(2, 6, 512, 64, 512), (2, 12, 64, 64, 3), (1, 10, 128, 0, 64)])
def test_deconv_forward(n, h, c0, c1, co):
    hip = _hip()
    xa = rnd(n, c0, h, h, seed=11)
    xb = rnd(n, c1, h, h, seed=12) if c1 else None
    x = torch.cat([xa, xb], 1) if c1 else xa
    ci = c0 + c1
    f = rnd(4, 4, co, ci, seed=13, std=0.05)
    ab = torch.cat([1.0 + 0.1 * rnd(ci, seed=14), 0.2 * rnd(ci, seed=15)])
    ref = T.conv2d_transpose_same_s2(torch.relu(x * ab[:ci].view(1, -1, 1, 1) + ab[ci:].view(1, -1, 1, 1)), f)
    ldc = 8 if co == 3 else co
    coff = 3 if co == 3 else 0
    out = torch.zeros((n, 2 * h, 2 * h, ldc), device='cuda')
    ab0 = torch.cat([ab[:c0], ab[ci:ci + c0]]).cuda()
    ab1 = torch.cat([ab[c0:ci], ab[ci + c0:]]).cuda() if c1 else None
    v = hip.View(nhwc(xa).cuda(), nhwc(xb).cuda() if c1 else None, ab0, 1, ab1)
    hip.deconv_forward(v, f.cuda(), out, coff=coff, epi=(1 if co == 3 else 0))
    got = nchw(out[..., coff:coff + co])
    close(got, torch.tanh(ref) if co == 3 else ref)
    if co == 3:
        assert float(out[..., :3].abs().max()) == 0.0 and float(out[..., 6:].abs().max()) == 0.0


@pytest.mark.parametrize('stride,h', [(2, 16), (1, 13)])
def test_conv_dgrad_and_wgrad(stride, h):
    hip = _hip()
    n, ci, co = 2, 64, 96
    x = rnd(n, ci, h, h, seed=21).requires_grad_(True)
    w = rnd(4, 4, ci, co, seed=22, std=0.05).requires_grad_(True)
    ab = torch.cat([1.0 + 0.1 * rnd(ci, seed=23), 0.2 * rnd(ci, seed=24)])
    xa = T.lrelu(x * ab[:ci].view(1, -1, 1, 1) + ab[ci:].view(1, -1, 1, 1), 0.2)
    xa.retain_grad()
    y = T.conv2d_valid_pad(xa, w, stride, 1)
    dy = rnd(*y.shape, seed=25)
    y.backward(dy)
    dx = torch.full((n, h, h, ci), float('nan'), device='cuda')
    hip.conv_dgrad(hip.View(nhwc(dy).cuda()), w.detach().cuda(), stride, 1, dx)
    close(nchw(dx), xa.grad)
    # split: channels [32, 64) only
    dxs = torch.full((n, h, h, 32), float('nan'), device='cuda')
    hip.conv_dgrad(hip.View(nhwc(dy).cuda()), w.detach().cuda(), stride, 1, dxs, n_off=32, nn=32)
    close(nchw(dxs), xa.grad[:, 32:])
    dw = torch.full((4, 4, ci, co), float('nan'), device='cuda')
    hip.conv_wgrad(hip.View(nhwc(x.detach()).cuda(), None, ab.cuda(), 2), hip.View(nhwc(dy).cuda()), dw, stride, 1)
    close(dw, w.grad)


def test_conv_wgrad_padded_small():
    hip = _hip()
    n, h = 2, 32
    x3 = rnd(n, 3, h, h, seed=31)
    w = rnd(4, 4, 3, 64, seed=32, std=0.05).requires_grad_(True)
    y = T.conv2d_valid_pad(x3, w, 2, 1)
    dy = rnd(*y.shape, seed=33)
    y.backward(dy)
    x4 = torch.zeros(n, h, h, 4)
    x4[..., :3] = nhwc(x3)
    dw = torch.full((4, 4, 3, 64), float('nan'), device='cuda')
    hip.conv_wgrad(hip.View(x4.cuda()), hip.View(nhwc(dy).cuda()), dw, 2, 1)
    close(dw, w.grad)
    # Cout = 1 (dy padded to 4 channels)
    x = rnd(n, 64, 13, 13, seed=34)
    w1 = rnd(4, 4, 64, 1, seed=35, std=0.05).requires_grad_(True)
    y1 = T.conv2d_valid_pad(x, w1, 1, 1)
    dy1 = rnd(*y1.shape, seed=36)
    y1.backward(dy1)
    dy4 = torch.zeros(n, 12, 12, 4)
    dy4[..., :1] = nhwc(dy1)
    dw1 = torch.full((4, 4, 64, 1), float('nan'), device='cuda')
    hip.conv_wgrad(hip.View(nhwc(x).cuda()), hip.View(dy4.cuda()), dw1, 1, 1)
    close(dw1, w1.grad)
    # dgrad of the Cout=1 conv
    xg = x.clone().requires_grad_(True)
    T.conv2d_valid_pad(xg, w1.detach(), 1, 1).backward(dy1)
    dx = torch.full((n, 13, 13, 64), float('nan'), device='cuda')
    hip.conv_dgrad(hip.View(dy4.cuda()), w1.detach().cuda(), 1, 1, dx, k_real=1)
    close(nchw(dx), xg.grad)


def test_deconv_dgrad_and_wgrad():
    hip = _hip()
    n, h, c0, c1, co = 2, 8, 64, 32, 48
    ci = c0 + c1
    x = rnd(n, ci, h, h, seed=41).requires_grad_(True)
    f = rnd(4, 4, co, ci, seed=42, std=0.05).requires_grad_(True)
    ab = torch.cat([1.0 + 0.1 * rnd(ci, seed=43), 0.2 * rnd(ci, seed=44)])
    xa = torch.relu(x * ab[:ci].view(1, -1, 1, 1) + ab[ci:].view(1, -1, 1, 1))
    xa.retain_grad()
    y = T.conv2d_transpose_same_s2(xa, f)
    dy = rnd(*y.shape, seed=45)
    y.backward(dy)
    dyv = hip.View(nhwc(dy).cuda())
    d0 = torch.full((n, h, h, c0), float('nan'), device='cuda')
    d1 = torch.full((n, h, h, c1), float('nan'), device='cuda')
    hip.deconv_dgrad(dyv, f.detach().cuda(), d0, n_off=0, nn=c0)
    hip.deconv_dgrad(dyv, f.detach().cuda(), d1, n_off=c0, nn=c1)
    close(nchw(d0), xa.grad[:, :c0])
    close(nchw(d1), xa.grad[:, c0:])
    xd = x.detach()
    ab0 = torch.cat([ab[:c0], ab[ci:ci + c0]]).cuda()
    ab1 = torch.cat([ab[c0:ci], ab[ci + c0:]]).cuda()
    v = hip.View(nhwc(xd[:, :c0]).cuda(), nhwc(xd[:, c0:]).cuda(), ab0, 1, ab1)
    df = torch.full((4, 4, co, ci), float('nan'), device='cuda')
    hip.deconv_wgrad(v, dyv, df)
    close(df, f.grad)


def test_matmuls():
    hip = _hip()
    a = rnd(300, 512, seed=51)
    b = rnd(512, 260, seed=52)
    bias = rnd(260, seed=53)
    out = torch.full((300, 260), float('nan'), device='cuda')
    hip.matmul(a.cuda(), b.cuda(), out, bias=bias.cuda())
    close(out, a @ b + bias, tol=1e-4)
    bt = rnd(260, 512, seed=54)
    out2 = torch.full((300, 260), float('nan'), device='cuda')
    hip.matmul_nt(a.cuda(), bt.cuda(), out2)
    close(out2, a @ bt.t(), tol=1e-4)
    c = rnd(300, 132, seed=55)
    out3 = torch.full((512, 132), float('nan'), device='cuda')
    hip.matmul_tn(a.cuda(), c.cuda(), out3)
    close(out3, a.t() @ c, tol=1e-4)
    # accumulate
    hip.matmul_tn(a.cuda(), c.cuda(), out3, accumulate=True)
    close(out3, 2 * (a.t() @ c), tol=1e-4)


@pytest.mark.parametrize('m,c', [(5000, 64), (333, 512), (70000, 128)])
def test_bn_stats_and_backward(m, c):
    hip = _hip()
    x = (rnd(m, c, seed=61) * 2.0 + 0.5).requires_grad_(True)
    scale = (1.0 + 0.1 * rnd(c, seed=62)).requires_grad_(True)
    offset = (0.1 * rnd(c, seed=63)).requires_grad_(True)
    x4 = x.t().reshape(1, c, m, 1)
    y = T.batchnorm(x4, scale, offset)
    z1 = T.lrelu(y, 0.2)
    z2 = torch.relu(y)
    g1 = rnd(m, c, seed=64)
    g2 = rnd(m, c, seed=65)
    loss = (z1.reshape(c, m).t() * g1).sum() + (z2.reshape(c, m).t() * g2).sum()
    loss.backward()
    ab = torch.empty(2 * c, device='cuda')
    st = torch.empty(2 * c, device='cuda')
    xd = x.detach().cuda()
    hip.bn_stats(xd, scale.detach().cuda(), offset.detach().cuda(), ab, st)
    yk = xd * ab[:c] + ab[c:]
    close(yk, y.reshape(c, m).t())
    dx = torch.full((m, c), float('nan'), device='cuda')
    ds = torch.empty(c, device='cuda')
    do = torch.empty(c, device='cuda')
    hip.bn_act_backward(xd, ab, st, g1.cuda(), 2, dx, g2=g2.cuda(), act2=1, dscale=ds, doffset=do)
    close(dx, x.grad, tol=5e-4)
    close(ds, scale.grad, tol=5e-4)
    close(do, offset.grad, tol=5e-4)
    # no-norm form: dx = g1*lrelu'(x) + g2*relu'(x)
    dx2 = torch.full((m, c), float('nan'), device='cuda')
    hip.bn_act_backward(xd, None, None, g1.cuda(), 2, dx2, g2=g2.cuda(), act2=1)
    xr = x.detach()
    ref = g1 * torch.where(xr > 0, torch.ones_like(xr), torch.full_like(xr, 0.2)) + g2 * (xr > 0).float()
    close(dx2, ref)


def test_layout_roundtrip():
    hip = _hip()
    x = rnd(3, 3, 10, 12, seed=71)
    d = torch.zeros(3, 10, 12, 8, device='cuda')
    hip.nchw_to_nhwc(x.cuda(), d, coff=3)
    close(d[..., 3:6], nhwc(x))
    back = torch.empty(3, 3, 10, 12, device='cuda')
    hip.nhwc_to_nchw(d, back, coff=3)
    close(back, x)


@pytest.mark.parametrize('kind', ['rmsprop', 'adagrad', 'adadelta'])
def test_other_optimizers_match_tf_formulas(kind):
    """ssc_optimizer_step vs the TF dense-apply formulas (oracle/tf_ops.py) over three steps."""
    import torch
    from oracle import tf_ops as T
    from sketchyscenecolorization_amd import hip
    g = torch.Generator().manual_seed(4)
    n = 1000
    w = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * 0.3 for _ in range(3)]
    wd = w.cuda().clone()
    lr = torch.tensor([1e-3], device='cuda')
    if kind == 'rmsprop':
        s1, s2 = torch.ones(n), torch.zeros(n)
        d1, d2 = s1.cuda().clone(), s2.cuda().clone()
        for gr in grads:
            T.tf_rmsprop_update(w, gr, s1, s2, 1e-3)
            hip.call('ssc_optimizer_step', 1, wd, gr.cuda(), d1, d2, n, lr, 0.9, 0.0, 1e-10, 1.0)
    elif kind == 'adagrad':
        s1 = torch.full((n,), 0.1)
        d1 = s1.cuda().clone()
        for gr in grads:
            T.tf_adagrad_update(w, gr, s1, 1e-3)
            hip.call('ssc_optimizer_step', 2, wd, gr.cuda(), d1, None, n, lr, 0.0, 0.0, 0.0, 1.0)
    else:
        s1, s2 = torch.zeros(n), torch.zeros(n)
        d1, d2 = s1.cuda().clone(), s2.cuda().clone()
        for gr in grads:
            T.tf_adadelta_update(w, gr, s1, s2, 1e-3)
            hip.call('ssc_optimizer_step', 3, wd, gr.cuda(), d1, d2, n, lr, 0.95, 0.0, 1e-8, 1.0)
    assert float((wd.cpu() - w).abs().max()) < 1e-6
    assert float((d1.cpu() - s1).abs().max()) < 1e-5 * max(1.0, float(s1.abs().max()))


@pytest.mark.parametrize('case', [
    # N, H, C0, C1, CO, k, stride, pad, norm          (uniform-tap layers of the training step + odd ones)
    (32, 48, 128, 0, 256, 4, 2, 1, True),       # encoder_3 at batch 32: 576 tiles, fewer than the chip holds
    (32, 96, 64, 0, 128, 4, 2, 1, True),        # encoder_2: 1152 tiles, 1.5 rounds
    (32, 12, 512, 0, 512, 4, 2, 1, True),       # encoder_5: 72 tiles x 256 K-tiles, every tile cut in parts
    (3, 24, 256, 256, 512, 4, 1, 1, False),     # two sources, stride 1, ragged M
    (2, 24, 64, 0, 579, 3, 1, 1, False),        # N not a multiple of the tile width
    (5, 20, 320, 260, 64, 3, 1, 1, True),       # channel-chunk K loop (C1 % 32 != 0), 128x64 tile
])
def test_in_launch_tail_split_equals_slab_launch(case):
    """Tail split with the K slices of a tile combined inside the launch (partial tiles handed to the tile's last slice
    through agent-scope flags) computes what the launch without it (split-K slabs + reduce kernel, or whole tiles only)
    computes: same products, another summation order.  Also: flags are left zero, no hand-off timed out, run-to-run
    bitwise."""
    from sketchyscenecolorization_amd import hip
    from sketchyscenecolorization_amd.hip import ACT_LRELU, ACT_NONE, View
    N, H, C0, C1, CO, k, stride, pad, norm = case
    g = torch.Generator(device='cuda').manual_seed(3)
    x0 = torch.randn(N, H, H, C0, device='cuda', generator=g)
    x1 = torch.randn(N, H, H, C1, device='cuda', generator=g) if C1 else None
    C1p = 0 if x1 is None else C1
    w = torch.randn(k, k, C0 + C1p, CO, device='cuda', generator=g) * 0.05
    ab0 = torch.randn(2 * C0, device='cuda', generator=g) if norm else None
    v = View(x0, x1, ab0, ACT_LRELU if norm else ACT_NONE)
    OH = (H + 2 * pad - k) // stride + 1
    ldc = (CO + 3) // 4 * 4
    outs = []
    for sk in (False, True, True):
        hip.SK_ENABLED = sk
        out = torch.zeros(N, OH, OH, ldc, device='cuda')
        hip.conv_forward(v, w, stride, pad, out, nstore=ldc if ldc != CO else None)
        torch.cuda.synchronize()
        outs.append(out)
    hip.SK_ENABLED = True
    ref, a, b = outs
    assert torch.equal(a, b)
    scale = float(ref.abs().max())
    assert float((a - ref).abs().max()) <= 2e-5 * scale, (float((a - ref).abs().max()), scale)
    assert hip.sk_timeouts() == 0
    assert int(hip.sk_flags().abs().sum()) == 0


@pytest.mark.parametrize('n,h,c,expect_fused', [(16, 48, 128, True), (2, 16, 64, False)])
def test_norm_backward_sums_from_dgrad_epilogues(n, h, c, expect_fused):
    """batchnorm -> (lrelu -> stride-2 conv) + (relu -> transposed conv), models_collection.py:36-46, 434-439, 512-531: the two
    per-channel sums of the norm's backward taken by the epilogues of the two data-gradient launches
    (ssc_conv_forward_bnbwd + ssc_bn_act_backward_pre) against the oracle's autograd, and against the separate pass."""
    hip = _hip()
    co1, co2 = 2 * c, c // 2
    x = (rnd(n, c, h, h, seed=71) * 1.5 + 0.3).requires_grad_(True)
    scale = (1.0 + 0.1 * rnd(c, seed=72)).requires_grad_(True)
    offset = (0.1 * rnd(c, seed=73)).requires_grad_(True)
    w1 = rnd(4, 4, c, co1, seed=74, std=0.05)
    f2 = rnd(4, 4, co2, c, seed=75, std=0.05)
    y = T.batchnorm(x, scale, offset)
    o1 = T.conv2d_valid_pad(T.lrelu(y, 0.2), w1, 2, 1)
    o2 = T.conv2d_transpose_same_s2(torch.relu(y), f2)
    dy1, dy2 = rnd(*o1.shape, seed=76), rnd(*o2.shape, seed=77)
    ((o1 * dy1).sum() + (o2 * dy2).sum()).backward()

    xd = nhwc(x.detach()).cuda()
    x2d = xd.view(-1, c)
    ab, st = torch.empty(2 * c, device='cuda'), torch.empty(2 * c, device='cuda')
    hip.bn_stats(x2d, scale.detach().cuda(), offset.detach().cuda(), ab, st)
    res = {}
    for fused in (True, False):
        sums = hip.BnBwdSums(x2d, ab, st, torch.zeros(hip.BnBwdSums.rows_needed(x2d.shape[0], 2), 2 * c, device='cuda')) \
            if fused else None
        g1 = torch.full((n, h, h, c), float('nan'), device='cuda')
        g2 = torch.full((n, h, h, c), float('nan'), device='cuda')
        hip.conv_dgrad(hip.View(nhwc(dy1).cuda()), w1.cuda(), 2, 1, g1, bnbwd=(sums.take(2) if fused else None))
        hip.deconv_dgrad(hip.View(nhwc(dy2).cuda()), f2.cuda(), g2, bnbwd=(sums.take(1) if fused else None))
        dx = torch.full((n * h * h, c), float('nan'), device='cuda')
        ds, do = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
        hip.bn_act_backward(x2d, ab, st, g1.view(-1, c), 2, dx, g2=g2.view(-1, c), act2=1, dscale=ds, doffset=do, pre=sums)
        if fused:
            assert sums.sources == 2
            if expect_fused:        # both launches finish their tiles in one workgroup each: the rows come from the epilogues
                assert sums.missed == 0 and sums.rows > 0
        res[fused] = (dx, ds, do)
        close(nchw(dx.view(n, h, h, c)), x.grad, tol=5e-4)
        close(ds, scale.grad, tol=5e-4)
        close(do, offset.grad, tol=5e-4)
    for a, b in zip(res[True], res[False]):
        close(a, b, tol=1e-4)
    assert hip.sk_timeouts() == 0


@pytest.mark.parametrize('n,h,cp,cr', [(3, 64, 8, 6), (2, 128, 4, 3), (1, 192, 8, 8)])
def test_few_channel_stride2_conv(n, h, cp, cr):
    """encoder_1 / discriminator layer_1 / decoder_1's data gradient (models_collection.py:454-458, 798-801, 529-534): 4x4
    stride-2 conv over a 4- or 8-channel tensor of which cr are real -- the persistent few-channel kernel (fewchan.hip)."""
    hip = _hip()
    x = rnd(n, cr, h, h, seed=81)
    xp = torch.zeros(n, h, h, cp)
    xp[..., :cr] = nhwc(x)
    xp[..., cr:] = 7.0          # padding channels may hold anything: the filter has no rows for them
    w = rnd(4, 4, cr, 64, seed=82, std=0.05)
    ref = T.conv2d_valid_pad(x, w, 2, 1)
    out = torch.full((n, h // 2, h // 2, 64), float('nan'), device='cuda')
    hip.PROFILE = []
    try:
        hip.conv_forward(hip.View(xp.cuda()), w.cuda(), 2, 1, out)
        names = [p[0] for p in hip.PROFILE]
    finally:
        hip.PROFILE = None
    assert names == ['conv_fewchan<%d>' % cp], names
    close(nchw(out), ref)
    # the conv form of the transposed conv's data gradient: filter [4,4,co,ci] read as HWIO, channel sub-range [64, 128)
    if cp == 4:
        f = rnd(4, 4, cr, 128, seed=83, std=0.05)
        xin = rnd(n, 128, h // 2, h // 2, seed=84).requires_grad_(True)
        y = T.conv2d_transpose_same_s2(xin, f)
        dy = rnd(*y.shape, seed=85)
        y.backward(dy)
        dyp = torch.zeros(n, h, h, cp)
        dyp[..., :cr] = nhwc(dy)
        g1 = torch.full((n, h // 2, h // 2, 64), float('nan'), device='cuda')
        hip.deconv_dgrad(hip.View(dyp.cuda()), f.cuda(), g1, n_off=64, nn=64)
        close(nchw(g1), xin.grad[:, 64:])


@pytest.mark.parametrize('rows,div2,form', [(200, 1, 'k2'), (2592, 36, 'plain'), (72, 36, 'k2'), (200, 1, 'bf16x6'),
                                            (2592, 36, 'bf16x6'), (72, 36, 'bf16x6'), (16, 1, 'bf16x6')])
def test_recurrent_step_all_kernel_forms(rows, div2, form):
    """ssc_lstm_step_fwd = h.K_h + the BasicLSTMCell gate math with the tf.cond pad skip (models_collection.py:184-236) in one
    launch; exact fp32: few rows take the K-split kernel, many rows the plain one (2592 rows x 512 units = 1312 workgroups >=
    1200); default arithmetic: ssc_lstm_step_fwd_bf on the planes of K_h (six bf16 products per fp32 product), same tolerance."""
    hip = _hip()
    C = 512
    if form != 'bf16x6':
        assert ((rows + 63) // 64) * (C // 16) >= 1200 if form == 'plain' else ((rows + 63) // 64) * (C // 16) < 1200
    elif not hip.ARITH_BF16:
        pytest.skip('SSC_ARITH=fp32')
    h = rnd(rows, C, seed=91, std=0.5)
    c = rnd(rows, C, seed=92, std=0.5)
    K = rnd(3 * C, 4 * C, seed=93, std=0.04)       # the recurrent rows are a slice of a wider kernel: ldk = 4C, rows [C, 2C)
    g1 = rnd(rows, 4 * C, seed=94, std=0.5)
    g2 = rnd(rows // div2, 4 * C, seed=95, std=0.5)
    mdiv = div2
    mask = (torch.arange(rows // mdiv) % 3 != 1).to(torch.int32)
    z = h.double() @ K[C:2 * C].double() + g1.double() + g2.double().repeat_interleave(div2, dim=0)
    i, j, f, o = z[:, :C], z[:, C:2 * C], z[:, 2 * C:3 * C], z[:, 3 * C:]
    c1 = c.double() * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
    h1 = torch.tanh(c1) * torch.sigmoid(o)
    keep = mask.repeat_interleave(mdiv).bool().unsqueeze(1)
    c_ref, h_ref = torch.where(keep, c1, c.double()), torch.where(keep, h1, h.double())
    co = torch.full((rows, C), float('nan'), device='cuda')
    ho = torch.full((rows, C), float('nan'), device='cuda')
    acts = torch.zeros(rows, 4 * C, device='cuda')
    Kd = K.cuda()
    hip.lstm_step_fwd(h.cuda(), Kd[C:2 * C], 4 * C, g1.cuda(), g2.cuda(), div2, mask.cuda(), mdiv, c.cuda(), rows, C, True, co, ho, acts,
                      exact=form != 'bf16x6')
    close(co, c_ref, tol=2e-5)
    close(ho, h_ref, tol=2e-5)
    a_ref = torch.cat([torch.sigmoid(i), torch.tanh(j), torch.sigmoid(f + 1.0), torch.sigmoid(o)], dim=1)
    close(acts[keep.squeeze(1).cuda()], a_ref[keep.squeeze(1)], tol=2e-5)
    if form == 'bf16x6':
        # the planes of h_out a step leaves for the next one (hp_out) are the planes ssc_lstm_hsplit makes of h_out: a second
        # step gives bit-identical results either way; and a first step (h = 0: no product, hp_in = None) writes them too
        hp = torch.zeros(2, hip.lstm_hplanes_floats(rows, C), device='cuda')
        z0 = torch.zeros(rows, C, device='cuda')
        c1d, h1d = torch.empty(rows, C, device='cuda'), torch.empty(rows, C, device='cuda')
        hip.lstm_step_fwd(z0, Kd[C:2 * C], 4 * C, g1.cuda(), g2.cuda(), div2, mask.cuda(), mdiv, c.cuda(), rows, C, False, c1d, h1d, acts,
                          hp_out=hp[0])
        outs = []
        for hp_in in (hp[0], None):
            c2, h2 = torch.empty(rows, C, device='cuda'), torch.empty(rows, C, device='cuda')
            hip.lstm_step_fwd(h1d, Kd[C:2 * C], 4 * C, g1.cuda(), g2.cuda(), div2, mask.cuda(), mdiv, c1d, rows, C, True, c2, h2, acts,
                              hp_in=hp_in, hp_out=hp[1])
            outs.append((c2, h2))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        z2 = h1d.double().cpu() @ K[C:2 * C].double() + g1.double() + g2.double().repeat_interleave(div2, dim=0)
        c2r = c1d.double().cpu() * torch.sigmoid(z2[:, 2 * C:3 * C] + 1.0) + torch.sigmoid(z2[:, :C]) * torch.tanh(z2[:, C:2 * C])
        close(outs[0][0], torch.where(keep, c2r, c1d.double().cpu()), tol=2e-5)


def test_handoff_timeout_is_reported_and_fatal():
    """igemm.hip, owner side of the in-launch K-slice hand-off: an owner that gives up waiting stores a partial sum, so it must
    say so -- the timeout word carries the launch's tag, hip.check_sk raises HandoffTimeout naming the launch and zeroes the
    flags, a flag that was never seen set is not cleared by the owner, and the next launch (hook off) is exact again."""
    from sketchyscenecolorization_amd import hip
    from sketchyscenecolorization_amd.hip import ACT_NONE, View
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(32, 48, 48, 128, device='cuda', generator=g)      # encoder_3 at batch 32: 512 whole tiles + 64 x 4 slices
    w = torch.randn(4, 4, 128, 256, device='cuda', generator=g) * 0.05
    v = View(x, None, None, ACT_NONE)
    hip.SK_ENABLED = False
    ref = torch.zeros(32, 24, 24, 256, device='cuda')
    hip.conv_forward(v, w, 2, 1, ref)
    hip.SK_ENABLED = True
    hip.sk_flags()
    hip.check_sk()
    try:
        assert hip.lib().ssc_sk_configure(30, 1) == 0       # 30 ms bound, producers withhold their flags
        out = torch.zeros_like(ref)
        hip.conv_forward(v, w, 2, 1, out)
        torch.cuda.synchronize()
        assert hip.sk_timeouts() == 1
        # (with the hook the producers did write their partial tiles -- only the flags are withheld -- so `out` may even be
        # right here; after a real timeout it is a partial sum, which is why the report must be fatal)
        with pytest.raises(hip.HandoffTimeout) as ei:
            hip.check_sk('unit test')
        assert ('conv_fwd<' in str(ei.value) or 'conv_bf16x6<' in str(ei.value)) and 'M=18432 N=256 K=2048' in str(ei.value), str(ei.value)
    finally:
        assert hip.lib().ssc_sk_configure(0, 0) == 0
    assert hip.sk_timeouts() == 0 and int(hip.sk_flags().abs().sum()) == 0     # zeroed by check_sk
    out = torch.zeros_like(ref)
    hip.conv_forward(v, w, 2, 1, out)
    torch.cuda.synchronize()
    hip.check_sk()
    assert float((out - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def _wgrad_name(hip, d):
    import ctypes
    buf = ctypes.create_string_buffer(64)
    hip.lib().ssc_conv_wgrad_kernel_name(ctypes.byref(d), buf, 64)
    return buf.value.decode()


@pytest.mark.parametrize('case', [
    # n, h, ci, co, k, stride, pad, act of x, norm on x
    (3, 20, 128, 256, 4, 2, 1, 2, True),        # one tap per 128-column tile (encoder_3's shape), K tail (300 pixels)
    (2, 24, 64, 128, 4, 2, 1, 2, False),        # 64 gathered channels: two taps per tile (encoder_2, discriminator layer_2)
    (2, 15, 64, 192, 3, 1, 1, 1, True),         # 9 taps of 64 channels: the last tile holds one tap; Nn = 1.5 tiles
    (1, 24, 256, 128, 4, 1, 1, 2, True),        # stride 1, 23 x 23 lattice (discriminator layer_4's form)
    (2, 12, 128, 130, 1, 1, 0, 0, False),       # 1 x 1, plain gathered side, ragged dense width (even, not a multiple of 4 x 32)
])
def test_wgrad128_conv(case):
    """conv filter gradient on the 128 x 128 kernel (wgrad128.hip; graph_single.py:24-30) vs the oracle's autograd: folded norm +
    activation on the gathered side, dy by LDS-DMA, out-of-image taps / pixels beyond the last / columns beyond the tensor
    through the descriptors' range check."""
    hip = _hip()
    n, h, ci, co, k, stride, pad, act, norm = case
    x = rnd(n, ci, h, h, seed=81)
    w = rnd(k, k, ci, co, seed=82, std=0.05).requires_grad_(True)
    ab = torch.cat([1.0 + 0.1 * rnd(ci, seed=83), 0.2 * rnd(ci, seed=84)]) if norm else None
    xa = x * ab[:ci].view(1, -1, 1, 1) + ab[ci:].view(1, -1, 1, 1) if norm else x
    xa = act_ref(xa, act)
    y = T.conv2d_valid_pad(xa, w, stride, pad)
    dy = rnd(*y.shape, seed=85)
    y.backward(dy)
    cop = (co + 3) // 4 * 4
    dyp = torch.zeros(n, y.shape[2], y.shape[3], cop)
    dyp[..., :co] = nhwc(dy)
    xv = hip.View(nhwc(x).cuda(), None, ab.cuda() if norm else None, act)
    dw = torch.full((k, k, ci, co), float('nan'), device='cuda')
    hip.conv_wgrad(xv, hip.View(dyp.cuda()), dw, stride, pad)
    close(dw, w.grad)
    d = hip.WgradDesc()
    d.g, d.d = xv.c(), hip.View(dyp.cuda()).c()
    d.out = dw.data_ptr()
    d.NB, d.PH, d.PW, d.TH, d.TW, d.in_stride, d.ioff_y, d.ioff_x = n, y.shape[2], y.shape[3], k, k, stride, -pad, -pad
    d.Cg_real, d.Nn, d.ldc, d.accumulate = ci, co, co, 0
    assert _wgrad_name(hip, d) in ('conv_wgrad128<128x128>', 'conv_wgrad128_bf16x6<128x128>')
    # accumulate: out += (splitk == 1 path and the reduce kernel's)
    dw2 = dw.clone()
    hip.conv_wgrad(xv, hip.View(dyp.cuda()), dw2, stride, pad, accumulate=True)
    close(dw2, 2 * w.grad)


def test_wgrad128_deconv_two_sources():
    """Transposed-conv filter gradient (decoder_3's form): gathered side = dy (plain), dense side = concat[decoder, encoder] with a
    folded norm + relu on each source (the transform runs on the dense tile; a column tile lies inside one source)."""
    hip = _hip()
    n, h, c0, c1, co = 2, 10, 128, 128, 128
    ci = c0 + c1
    x = rnd(n, ci, h, h, seed=91)
    f = rnd(4, 4, co, ci, seed=92, std=0.05).requires_grad_(True)
    ab = torch.cat([1.0 + 0.1 * rnd(ci, seed=93), 0.2 * rnd(ci, seed=94)])
    xa = torch.relu(x * ab[:ci].view(1, -1, 1, 1) + ab[ci:].view(1, -1, 1, 1))
    y = T.conv2d_transpose_same_s2(xa, f)
    dy = rnd(*y.shape, seed=95)
    y.backward(dy)
    ab0 = torch.cat([ab[:c0], ab[ci:ci + c0]]).cuda()
    ab1 = torch.cat([ab[c0:ci], ab[ci + c0:]]).cuda()
    v = hip.View(nhwc(x[:, :c0]).cuda(), nhwc(x[:, c0:]).cuda(), ab0, 1, ab1)
    df = torch.full((4, 4, co, ci), float('nan'), device='cuda')
    hip.deconv_wgrad(v, hip.View(nhwc(dy).cuda()), df)
    close(df, f.grad)
    # decoder_5's form: 512 + 64 dense channels (the last column tile is half empty and starts the second source)
    c0, c1, co, h = 512, 64, 128, 6
    ci = c0 + c1
    x = rnd(n, ci, h, h, seed=96)
    f = rnd(4, 4, co, ci, seed=97, std=0.05).requires_grad_(True)
    xa = torch.relu(x)
    y = T.conv2d_transpose_same_s2(xa, f)
    dy = rnd(*y.shape, seed=98)
    y.backward(dy)
    v = hip.View(nhwc(x[:, :c0]).cuda(), nhwc(x[:, c0:]).cuda(), None, 1, None)
    df = torch.full((4, 4, co, ci), float('nan'), device='cuda')
    hip.deconv_wgrad(v, hip.View(nhwc(dy).cuda()), df)
    close(df, f.grad)


@pytest.mark.parametrize('case', [
    # n, h, k, pad, act of x, norm on x
    (3, 23, 4, 1, 2, True),         # discriminator layer_5 (models_collection.py:833-835): 23 x 23 x 512 -> 22 x 22 x 1
    (2, 9, 3, 1, 1, False),         # 3 x 3 SAME-like, relu, no norm: 9 taps (lanes 9..15 idle)
    (1, 7, 4, 0, 0, True),          # no padding: every tap in range
])
def test_head1_patch_head(case):
    """The one-output patch head (head1.hip) in its three forms -- forward, data gradient, filter gradient -- vs the oracle conv
    and its autograd; the launcher must pick the streaming kernels for these descriptors."""
    import ctypes
    hip = _hip()
    n, h, k, pad, act, norm = case
    ci = 512
    x = rnd(n, ci, h, h, seed=101).requires_grad_(True)
    w = rnd(k, k, ci, 1, seed=102, std=0.05).requires_grad_(True)
    ab = torch.cat([1.0 + 0.1 * rnd(ci, seed=103), 0.2 * rnd(ci, seed=104)]) if norm else None
    xa = x * ab[:ci].view(1, -1, 1, 1) + ab[ci:].view(1, -1, 1, 1) if norm else x
    xa = act_ref(xa, act)
    xa.retain_grad()
    y = T.conv2d_valid_pad(xa, w, 1, pad)
    dy = rnd(*y.shape, seed=105)
    y.backward(dy)
    oh = y.shape[2]
    xv = hip.View(nhwc(x.detach()).cuda(), None, ab.cuda() if norm else None, act)
    # forward: channel 0 of a 4-float row, the other stored columns 0
    out = torch.full((n, oh, oh, 4), float('nan'), device='cuda')
    hip.conv_forward(xv, w.detach().cuda(), 1, pad, out, nstore=4)
    close(out[..., 0], y.detach()[:, 0])
    assert float(out[..., 1:].abs().max()) == 0.0
    # data gradient w.r.t. the activated tensor
    dyp = torch.zeros(n, oh, oh, 4)
    dyp[..., 0] = dy[:, 0]
    g = torch.full((n, h, h, ci), float('nan'), device='cuda')
    hip.conv_dgrad(hip.View(dyp.cuda()), w.detach().cuda(), 1, pad, g, k_real=1)
    close(g, nhwc(xa.grad))
    # filter gradient, then accumulated on top of itself
    dw = torch.full((k, k, ci, 1), float('nan'), device='cuda')
    hip.conv_wgrad(xv, hip.View(dyp.cuda()), dw, 1, pad)
    close(dw, w.grad)
    hip.conv_wgrad(xv, hip.View(dyp.cuda()), dw, 1, pad, accumulate=True)
    close(dw, 2 * w.grad)
    # the launcher's choice
    d = hip.WgradDesc()
    d.g, d.d = xv.c(), hip.View(dyp.cuda()).c()
    d.out = dw.data_ptr()
    d.NB, d.PH, d.PW, d.TH, d.TW, d.in_stride, d.ioff_y, d.ioff_x = n, oh, oh, k, k, 1, -pad, -pad
    d.Cg_real, d.Nn, d.ldc, d.accumulate = ci, 1, 1, 0
    assert _wgrad_name(hip, d) == 'head1_wgrad'
    assert hip.lib().ssc_head1_wgrad_supported(ctypes.byref(d)) == 1


@pytest.mark.parametrize('with_rowb', [True, False])
def test_head1_dgrad_fused_with_norm_backward(with_rowb):
    """ssc_head1_dgrad_bn_backward (the head's data gradient recomputed inside layer_4's norm backward, never stored) against
    the two separate launches (conv_dgrad, then bn_act_backward with the class head's per-image term) and the oracle's autograd."""
    hip = _hip()
    n, h, k, pad, ci = 3, 23, 4, 1, 512
    x = rnd(n, ci, h, h, seed=111).requires_grad_(True)
    w = rnd(k, k, ci, 1, seed=112, std=0.05)
    scale = (1.0 + 0.1 * rnd(ci, seed=113)).requires_grad_(True)
    offset = (0.2 * rnd(ci, seed=114)).requires_grad_(True)
    y4 = T.lrelu(T.batchnorm(x, scale, offset), 0.2)
    out = T.conv2d_valid_pad(y4, w, 1, pad)
    dy = rnd(*out.shape, seed=115)
    v = rnd(n, ci, seed=116) if with_rowb else None
    loss = (out * dy).sum()
    if with_rowb:       # a term through the spatial mean of y4, as the class head's
        loss = loss + (y4.mean(dim=(2, 3)) * v).sum()
    loss.backward()
    oh = out.shape[2]
    xd = nhwc(x.detach()).cuda()
    ab, st = torch.empty(2 * ci, device='cuda'), torch.empty(2 * ci, device='cuda')
    hip.bn_stats(xd.view(-1, ci), scale.detach().cuda(), offset.detach().cuda(), ab, st)
    dyp = torch.zeros(n, oh, oh, 4)
    dyp[..., 0] = dy[:, 0]
    dyv = hip.View(dyp.cuda())
    wd = w.cuda()
    rowb = (v.cuda(), 1.0 / (h * h)) if with_rowb else None
    dx = torch.full((n, h, h, ci), float('nan'), device='cuda')
    ds, do = torch.full((ci,), float('nan'), device='cuda'), torch.full((ci,), float('nan'), device='cuda')
    assert hip.head1_dgrad_bn_backward(dyv, wd, pad, xd, ab, st, 2, dx, dscale=ds, doffset=do, rowb=rowb)
    close(dx, nhwc(x.grad))
    close(ds, scale.grad)
    close(do, offset.grad)
    # the separate launches
    g4 = torch.empty(n, h, h, ci, device='cuda')
    hip.conv_dgrad(dyv, wd, 1, pad, g4, k_real=1)
    dx2, ds2, do2 = torch.empty_like(dx), torch.empty_like(ds), torch.empty_like(do)
    hip.bn_act_backward(xd.view(-1, ci), ab, st, g4.view(-1, ci), 2, dx2.view(-1, ci), dscale=ds2, doffset=do2,
                        rowb=(rowb + (h * h,) if rowb else None))
    close(dx, dx2, tol=2e-5)
    close(ds, ds2, tol=2e-5)
    close(do, do2, tol=2e-5)
