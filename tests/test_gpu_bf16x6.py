"""The bf16-split implicit GEMM (igemm_bf16.hip: six bf16 products per fp32 product, fp32 accumulate) against float64, beside the
exact-fp32 MFMA kernel on the same inputs: one case per kernel form (conv with folded norm + activation, transposed conv over a
two-source concat, both data gradients, a column sub-range, the dense matmuls, long K).  The split kernel must be within 1.5 x
the exact kernel's own distance from float64 (plus one unit of fp32 rounding of the output scale), and the dispatcher must
really have taken it.  Reference semantics: models_collection.py:380-405 (nchw_conv / nchw_deconv)."""
import pytest
import torch

from oracle import tf_ops as T
from conftest import parity_log

pytestmark = pytest.mark.gpu


def _hip():
    from sketchyscenecolorization_amd import hip
    return hip


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * std


def act_ref(x, act):
    return torch.relu(x) if act == 1 else (T.lrelu(x, 0.2) if act == 2 else x)


def _both(run, ref64, name, config):
    """run() -> output tensor (device); evaluated with the bf16 split and with SSC_ARITH=fp32 semantics; returns the errors."""
    hip = _hip()
    errs, names = {}, {}
    for mode in ('bf16x6', 'fp32'):
        hip.ARITH_BF16 = mode == 'bf16x6'
        hip.PROFILE = []
        try:
            out = run()
            torch.cuda.synchronize()
            names[mode] = [p[0] for p in hip.PROFILE]
        finally:
            hip.PROFILE = None
            hip.ARITH_BF16 = True
        errs[mode] = float((out.detach().cpu().double() - ref64).abs().max())
    scale = float(ref64.abs().max())
    assert any(n.startswith('conv_bf16x6') for n in names['bf16x6']), names
    assert not any(n.startswith('conv_bf16x6') for n in names['fp32']), names
    bound = 1.5 * errs['fp32'] + 1.2e-7 * scale
    parity_log(name, config, errs['bf16x6'], bound, exact_fp32_err=errs['fp32'], scale=scale, kernels=names['bf16x6'])
    assert errs['bf16x6'] <= bound, (errs, scale)
    assert errs['bf16x6'] <= 1e-3 * max(1.0, scale)
    return errs


@pytest.mark.parametrize('n,h,ci,co,stride,act', [(4, 24, 64, 128, 2, 2), (2, 23, 256, 512, 1, 2), (8, 6, 512, 512, 2, 2),
                                                   (3, 16, 64, 64, 2, 0)])
def test_conv_forward_split_vs_exact(n, h, ci, co, stride, act):
    hip = _hip()
    x = rnd(n, ci, h, h, seed=1)
    w = rnd(4, 4, ci, co, seed=2, std=0.05)
    ab = torch.cat([1.0 + 0.1 * rnd(ci, seed=3), 0.2 * rnd(ci, seed=4)])
    xd, wd, abd = x.double(), w.double(), ab.double()
    ref = T.conv2d_valid_pad(act_ref(xd * abd[:ci].view(1, -1, 1, 1) + abd[ci:].view(1, -1, 1, 1), act), wd, stride, 1)
    oh = ref.shape[2]
    xg, wg, abg = nhwc(x).cuda(), w.cuda(), ab.cuda()

    def run():
        out = torch.full((n, oh, oh, co), float('nan'), device='cuda')
        hip.conv_forward(hip.View(xg, None, abg, act), wg, stride, 1, out)
        return nchw(out)
    _both(run, ref, 'conv_forward_split_vs_exact', dict(n=n, h=h, ci=ci, co=co, stride=stride, act=act))


@pytest.mark.parametrize('n,h,ch,extra,co,k,table', [(3, 24, 64, 3, 64, 3, False), (2, 12, 128, 3, 128, 3, False),
                                                      (2, 10, 256, 131, 256, 3, False), (2, 9, 64, 3, 128, 3, True),
                                                      (1, 8, 512, 3, 64, 1, False)])
def test_conv_partial_chunk_split_vs_exact(n, h, ch, extra, co, k, table):
    """The MRU blocks' SAME convs over a materialised concat [state | image(3) (| skip)] (mru.py:400-411, 555-575): ch + extra real
    channels in a buffer whose rows are padded to a multiple of 4 -- the last 32-wide chunk of every tap is partly empty
    (conv_bf_kernel<KM>).  The buffer's padding channels hold garbage here: the filter planes' zeros must cancel them."""
    hip = _hip()
    ci = ch + extra
    cp = (ci + 3) // 4 * 4
    x = rnd(n, ci, h, h, seed=41)
    w = rnd(k, k, ci, co, seed=42, std=0.05)
    bias = rnd(co, seed=43, std=0.1)
    ab = torch.cat([1.0 + 0.1 * rnd(cp, seed=44), 0.2 * rnd(cp, seed=45)])
    xd = x.double()
    if table:
        xd = act_ref(xd * ab[:ci].double().view(1, -1, 1, 1) + ab[cp:cp + ci].double().view(1, -1, 1, 1), 2)
    ref = T.lrelu(T.conv2d_same(xd, w.double(), 1, bias.double()), 0.2)
    xp = torch.full((n, h, h, cp), 7.5)
    xp[..., :ci] = nhwc(x)
    xg, wg, bg, abg = xp.cuda(), w.cuda(), bias.cuda(), ab.cuda()

    def run():
        out = torch.full((n, h, h, co), float('nan'), device='cuda')
        v = hip.View(xg, None, abg, 2) if table else hip.View(xg)
        hip.conv_forward(v, wg, 1, 0, out, bias=bg, epi=2, same=True)
        return nchw(out)
    _both(run, ref, 'conv_partial_chunk_split_vs_exact', dict(n=n, h=h, ch=ch, extra=extra, co=co, k=k, table=table))


@pytest.mark.parametrize('n,h,c0,c1,co', [(4, 12, 64, 64, 64), (2, 6, 512, 512, 256)])
def test_deconv_forward_concat_split_vs_exact(n, h, c0, c1, co):
    """relu(concat[decoder_{k+1}, encoder_k]) -> conv2d_transpose (models_collection.py:512-531): two sources, two norm tables,
    four sub-pixel phases, the [n][k] filter orientation."""
    hip = _hip()
    a, b = rnd(n, c0, h, h, seed=11), rnd(n, c1, h, h, seed=12)
    f = rnd(4, 4, co, c0 + c1, seed=13, std=0.05)
    ab0 = torch.cat([1.0 + 0.1 * rnd(c0, seed=14), 0.2 * rnd(c0, seed=15)])
    ab1 = torch.cat([1.0 + 0.1 * rnd(c1, seed=16), 0.2 * rnd(c1, seed=17)])
    ad, bd = a.double(), b.double()
    xin = torch.relu(torch.cat([ad * ab0[:c0].double().view(1, -1, 1, 1) + ab0[c0:].double().view(1, -1, 1, 1),
                                bd * ab1[:c1].double().view(1, -1, 1, 1) + ab1[c1:].double().view(1, -1, 1, 1)], 1))
    ref = T.conv2d_transpose_same_s2(xin, f.double())
    ag, bg, fg, ab0g, ab1g = nhwc(a).cuda(), nhwc(b).cuda(), f.cuda(), ab0.cuda(), ab1.cuda()

    def run():
        out = torch.full((n, 2 * h, 2 * h, co), float('nan'), device='cuda')
        hip.deconv_forward(hip.View(ag, bg, ab0g, 1, ab1g), fg, out)
        return nchw(out)
    _both(run, ref, 'deconv_forward_concat_split_vs_exact', dict(n=n, h=h, c0=c0, c1=c1, co=co))


@pytest.mark.parametrize('n,h,ci,co,stride', [(4, 24, 64, 128, 2), (2, 23, 256, 512, 1)])
def test_conv_dgrad_split_vs_exact(n, h, ci, co, stride):
    hip = _hip()
    x = rnd(n, ci, h, h, seed=21).double().requires_grad_(True)
    w = rnd(4, 4, ci, co, seed=22, std=0.05)
    y = T.conv2d_valid_pad(x, w.double(), stride, 1)
    dy = rnd(*y.shape, seed=23)
    y.backward(dy.double())
    ref = x.grad.detach()
    dyg, wg = nhwc(dy).cuda(), w.cuda()

    def run():
        out = torch.full((n, h, h, ci), float('nan'), device='cuda')
        hip.conv_dgrad(hip.View(dyg), wg, stride, 1, out)
        return nchw(out)
    _both(run, ref, 'conv_dgrad_split_vs_exact', dict(n=n, h=h, ci=ci, co=co, stride=stride))


@pytest.mark.parametrize('n,h,ci,co,n_off,nn', [(4, 12, 128, 64, 0, 128), (4, 12, 128, 64, 64, 64), (2, 6, 1024, 256, 512, 512)])
def test_deconv_dgrad_split_vs_exact(n, h, ci, co, n_off, nn):
    """Gradient of the transposed conv w.r.t. a channel sub-range of its (concatenated) input: a stride-2 conv of dy."""
    hip = _hip()
    x = rnd(n, ci, h, h, seed=31).double().requires_grad_(True)
    f = rnd(4, 4, co, ci, seed=32, std=0.05)
    y = T.conv2d_transpose_same_s2(x, f.double())
    dy = rnd(*y.shape, seed=33)
    y.backward(dy.double())
    ref = x.grad.detach()[:, n_off:n_off + nn]
    dyg, fg = nhwc(dy).cuda(), f.cuda()

    def run():
        out = torch.full((n, h, h, nn), float('nan'), device='cuda')
        hip.deconv_dgrad(hip.View(dyg), fg, out, n_off=n_off, nn=nn)
        return nchw(out)
    _both(run, ref, 'deconv_dgrad_split_vs_exact', dict(n=n, h=h, ci=ci, co=co, n_off=n_off, nn=nn))


@pytest.mark.parametrize('m,k,nn', [(1152, 512, 2048), (96, 1024, 2048)])
def test_matmul_split_vs_exact(m, k, nn):
    hip = _hip()
    a, b = rnd(m, k, seed=41), rnd(k, nn, seed=42, std=0.05)
    ref = a.double() @ b.double()
    ag, bg = a.cuda(), b.cuda()

    def run():
        out = torch.full((m, nn), float('nan'), device='cuda')
        hip.matmul(ag, bg, out)
        return out
    _both(run, ref, 'matmul_split_vs_exact', dict(m=m, k=k, n=nn))

    def run_nt():
        out = torch.full((m, k), float('nan'), device='cuda')
        hip.matmul_nt(refg, bg, out)
        return out
    refg = ref.float().cuda()
    ref_nt = refg.cpu().double() @ b.double().t()
    _both(run_nt, ref_nt, 'matmul_nt_split_vs_exact', dict(m=m, k=nn, n=k))


def _both_wgrad(run, ref64, name, config):
    """Filter gradients: hip.ARITH_BF16 selects the bf16-split 128 x 128 kernel (wgrad128_bf16.hip) or the exact-fp32 one."""
    hip = _hip()
    errs, names = {}, {}
    for mode in ('bf16x6', 'fp32'):
        hip.ARITH_BF16 = mode == 'bf16x6'
        try:
            out = run()
            torch.cuda.synchronize()
        finally:
            hip.ARITH_BF16 = True
        errs[mode] = float((out.detach().cpu().double() - ref64).abs().max())
    scale = float(ref64.abs().max())
    assert errs['bf16x6'] != errs['fp32'] or errs['fp32'] == 0.0, 'both modes ran the same kernel'
    bound = 1.5 * errs['fp32'] + 1.2e-7 * scale
    parity_log(name, config, errs['bf16x6'], bound, exact_fp32_err=errs['fp32'], scale=scale, kernels=['conv_wgrad128_bf16x6'])
    assert errs['bf16x6'] <= bound, (errs, scale)
    return errs


@pytest.mark.parametrize('n,h,ci,co,stride,act,tab', [(8, 48, 128, 256, 2, 2, True), (8, 96, 64, 128, 2, 2, False),
                                                       (4, 24, 256, 512, 1, 2, True), (6, 24, 256, 512, 2, 0, False)])
def test_conv_wgrad_split_vs_exact(n, h, ci, co, stride, act, tab):
    """dW of a 4x4 conv (graph_single.py:24-30): gathered side with folded norm + activation (or plain), dense side dy; the
    64-channel case takes two taps per tile."""
    hip = _hip()
    x = rnd(n, ci, h, h, seed=61)
    ab = torch.cat([1.0 + 0.1 * rnd(ci, seed=62), 0.2 * rnd(ci, seed=63)]) if tab else None
    xin = x.double()
    if tab:
        xin = xin * ab[:ci].double().view(1, -1, 1, 1) + ab[ci:].double().view(1, -1, 1, 1)
    xin = act_ref(xin, act)
    w = torch.zeros(4, 4, ci, co, dtype=torch.float64, requires_grad=True)
    y = T.conv2d_valid_pad(xin, w, stride, 1)
    dy = rnd(*y.shape, seed=64)
    y.backward(dy.double())
    ref = w.grad.detach()
    xg, dyg, abg = nhwc(x).cuda(), nhwc(dy).cuda(), (ab.cuda() if tab else None)

    def run():
        out = torch.full((4, 4, ci, co), float('nan'), device='cuda')
        hip.conv_wgrad(hip.View(xg, None, abg, act), hip.View(dyg), out, stride, 1)
        return out
    _both_wgrad(run, ref, 'conv_wgrad_split_vs_exact', dict(n=n, h=h, ci=ci, co=co, stride=stride, act=act, tab=tab))


@pytest.mark.parametrize('n,h,ch,extra,co,k,tab', [(4, 24, 128, 3, 128, 3, False), (2, 20, 64, 3, 128, 3, False),
                                                    (2, 12, 256, 131, 256, 3, False), (2, 16, 128, 3, 128, 3, True),
                                                    (3, 12, 512, 3, 256, 1, False), (2, 24, 96, 0, 128, 3, False)])
def test_conv_wgrad_partial_gathered_split_vs_exact(n, h, ch, extra, co, k, tab):
    """dW of the MRU blocks' SAME convs over a materialised concat [state | image(3) (| skip)] (mru.py:400-411, 555-575):
    ch + extra real gathered channels in rows padded to a multiple of 4, any channel count from 64 up -- the 128-row tiles of
    wgrad128_bf16.hip run over the padded row space and span two or three taps; the padding channels hold garbage and have no
    row in the result."""
    hip = _hip()
    ci = ch + extra
    cp = (ci + 3) // 4 * 4
    x = rnd(n, ci, h, h, seed=71)
    ab = torch.cat([1.0 + 0.1 * rnd(cp, seed=72), 0.2 * rnd(cp, seed=73)]) if tab else None
    xin = x.double()
    if tab:
        xin = act_ref(xin * ab[:ci].double().view(1, -1, 1, 1) + ab[cp:cp + ci].double().view(1, -1, 1, 1), 2)
    w = torch.zeros(k, k, ci, co, dtype=torch.float64, requires_grad=True)
    y = T.conv2d_same(xin, w, 1)
    dy = rnd(*y.shape, seed=74)
    y.backward(dy.double())
    ref = w.grad.detach()
    xp = torch.full((n, h, h, cp), -3.25)
    xp[..., :ci] = nhwc(x)
    xg, dyg, abg = xp.cuda(), nhwc(dy).cuda(), (ab.cuda() if tab else None)

    def run():
        out = torch.full((k, k, ci, co), float('nan'), device='cuda')
        hip.conv_wgrad(hip.View(xg, None, abg, 2 if tab else 0), hip.View(dyg), out, 1, hip.same_pad_before(h, k, 1))
        return out
    _both_wgrad(run, ref, 'conv_wgrad_partial_gathered_split_vs_exact', dict(n=n, h=h, ch=ch, extra=extra, co=co, k=k, tab=tab))


@pytest.mark.parametrize('n,h,c0,c1,co', [(8, 12, 128, 128, 128), (4, 24, 256, 0, 128)])
def test_deconv_wgrad_split_vs_exact(n, h, c0, c1, co):
    """dF of the transposed conv: gathered side dy (plain), dense side relu(concat of two normed tensors)."""
    hip = _hip()
    a = rnd(n, c0, h, h, seed=71)
    b = rnd(n, c1, h, h, seed=72) if c1 else None
    ab0 = torch.cat([1.0 + 0.1 * rnd(c0, seed=73), 0.2 * rnd(c0, seed=74)])
    ab1 = torch.cat([1.0 + 0.1 * rnd(c1, seed=75), 0.2 * rnd(c1, seed=76)]) if c1 else None
    parts = [a.double() * ab0[:c0].double().view(1, -1, 1, 1) + ab0[c0:].double().view(1, -1, 1, 1)]
    if c1:
        parts.append(b.double() * ab1[:c1].double().view(1, -1, 1, 1) + ab1[c1:].double().view(1, -1, 1, 1))
    xin = torch.relu(torch.cat(parts, 1))
    f = torch.zeros(4, 4, co, c0 + c1, dtype=torch.float64, requires_grad=True)
    y = T.conv2d_transpose_same_s2(xin, f)
    dy = rnd(*y.shape, seed=77)
    y.backward(dy.double())
    ref = f.grad.detach()
    ag, bg, dyg = nhwc(a).cuda(), (nhwc(b).cuda() if c1 else None), nhwc(dy).cuda()
    ab0g, ab1g = ab0.cuda(), (ab1.cuda() if c1 else None)

    def run():
        out = torch.full((4, 4, co, c0 + c1), float('nan'), device='cuda')
        hip.deconv_wgrad(hip.View(ag, bg, ab0g, 1, ab1g), hip.View(dyg), out)
        return out
    _both_wgrad(run, ref, 'deconv_wgrad_split_vs_exact', dict(n=n, h=h, c0=c0, c1=c1, co=co))


def test_filter_planes_follow_the_weights():
    """The planes are refreshed when torch modifies the filter (version counter) and by refresh_splits() after a write torch does
    not see (the optimizer kernels)."""
    hip = _hip()
    n, h, ci, co = 2, 16, 64, 64
    x = nhwc(rnd(n, ci, h, h, seed=51)).cuda()
    w = rnd(4, 4, ci, co, seed=52, std=0.05).cuda()
    rng = hip.register_param_buffer(w)              # a parameter: persistent planes (any other tensor is split before every launch)
    out1 = torch.empty(n, 8, 8, co, device='cuda')
    hip.conv_forward(hip.View(x), w, 2, 1, out1)
    w.mul_(2.0)                                     # torch sees this one
    out2 = torch.empty_like(out1)
    hip.conv_forward(hip.View(x), w, 2, 1, out2)
    assert float((out2 - 2.0 * out1).abs().max()) <= 1e-5 * float(out1.abs().max())
    hip.call('ssc_axpy', w, w, 1.0, w.numel())      # w += w behind torch's back
    hip.refresh_splits(w)
    out3 = torch.empty_like(out1)
    hip.conv_forward(hip.View(x), w, 2, 1, out3)
    assert float((out3 - 4.0 * out1).abs().max()) <= 1e-5 * float(out1.abs().max())
    # without the refresh the planes of a PARAMETER are stale (that is the contract) ...
    hip.call('ssc_axpy', w, w, 1.0, w.numel())
    out4 = torch.empty_like(out1)
    hip.conv_forward(hip.View(x), w, 2, 1, out4)
    assert float((out4 - 4.0 * out1).abs().max()) <= 1e-5 * float(out1.abs().max())
    hip.release_param_buffer(rng)
    # ... while a tensor that is no parameter is split in front of every launch
    out5 = torch.empty_like(out1)
    hip.conv_forward(hip.View(x), w, 2, 1, out5)
    assert float((out5 - 8.0 * out1).abs().max()) <= 1e-5 * float(out1.abs().max())
