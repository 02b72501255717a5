"""The bf16-split implicit GEMM (igemm_bf16.hip: six bf16 products per fp32 product, fp32 accumulate) against float64, beside the
exact-fp32 MFMA kernel on the same inputs: one case per kernel form (conv with folded norm + activation, transposed conv over a
two-source concat, both data gradients, a column sub-range, the dense matmuls, long K).  The split kernel must be within 1.5 x
the exact kernel's own distance from float64 (plus one unit of fp32 rounding of the output scale), and the dispatcher must
really have taken it.  Reference semantics: models_collection.py:380-405 (nchw_conv / nchw_deconv)."""
import os

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


@pytest.mark.parametrize('table', [False, True])
def test_conv_partial_chunk_ignores_nan_in_the_padding_lane(table):
    """conv_bf_kernel<KM>: a NaN / Inf in the stored padding channel of a concat row (channels [k_real, C0)) must not reach the
    output -- the zero filter planes do not cancel it (0 * NaN), the staging clears the lane."""
    hip = _hip()
    if not hip.ARITH_BF16:
        pytest.skip('SSC_ARITH=fp32')
    n, h, ci, co, k = 2, 12, 67, 64, 3
    cp = 68
    x = rnd(n, ci, h, h, seed=46)
    w = rnd(k, k, ci, co, seed=47, std=0.05).cuda()
    ab = torch.cat([1.0 + 0.1 * rnd(cp, seed=48), 0.2 * rnd(cp, seed=49)]).cuda()
    outs = []
    for pad in (0.0, float('nan'), float('inf')):
        xp = torch.full((n, h, h, cp), pad)
        xp[..., :ci] = nhwc(x)
        out = torch.full((n, h, h, co), float('nan'), device='cuda')
        hip.PROFILE = []
        try:
            hip.conv_forward(hip.View(xp.cuda(), None, ab, 2) if table else hip.View(xp.cuda()), w, 1, 0, out, same=True)
            torch.cuda.synchronize()
            assert any(p[0].startswith('conv_bf16x6') for p in hip.PROFILE), [p[0] for p in hip.PROFILE]
        finally:
            hip.PROFILE = None
        outs.append(out)
    assert bool(torch.isfinite(outs[0]).all())
    assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])


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


def _fresh_trainer(img=64, seed=3):
    from sketchyscenecolorization_amd.trainer import Pix2PixTrainer
    from oracle import pix2pix as O
    tr = Pix2PixTrainer(img=img, seed=seed)
    p = O.init_params(seed, img=img)
    tr.store.load_dict(p)
    return tr, O


def test_replayed_inference_follows_weights_loaded_through_torch():
    """A captured inference graph holds the ADDRESSES of the filters' bf16 planes and never looks at a tensor's version: weights
    replaced through torch on a live trainer (ParamStore.load_dict) must reach the planes before the next replay, or the bf16
    layers would keep computing with the old checkpoint while the fp32-path layers read the new one."""
    hip = _hip()
    if not hip.ARITH_BF16:
        pytest.skip('SSC_ARITH=fp32: no planes')
    tr, O = _fresh_trainer()
    b = O.synthetic_batch(4, seed=77, img=64)
    sk, nv, text = b['sketches'].cuda(), b['noise_vec'].cuda(), b['text'].numpy()
    for _ in range(3):              # eager, capture, replay
        out_a = tr.generate(sk, text, nv)
    assert any(k[0] == 'infer' for k in tr._graphs), 'the inference pass was not captured'
    p2 = O.init_params(11, img=64)   # another set of weights
    tr.store.load_dict(p2)
    out_b = tr.generate(sk, text, nv)            # replayed
    ref_b = O.generate_pix2pix(p2, b['sketches'], b['text'], b['noise_vec'])
    err = float((out_b.cpu() - ref_b).abs().max())
    assert err < 1e-3, ('the replayed pass mixes two checkpoints', err, float((out_b - out_a).abs().max()))
    # ... and a write straight into one filter's view (no ParamStore call) is picked up in front of the replay
    w = tr.store['generator/encoder_3/conv/filter']
    w.mul_(0.5)
    p3 = dict(p2)
    p3['generator/encoder_3/conv/filter'] = p2['generator/encoder_3/conv/filter'] * 0.5
    out_c = tr.generate(sk, text, nv)
    ref_c = O.generate_pix2pix(p3, b['sketches'], b['text'], b['noise_vec'])
    assert float((out_c.cpu() - ref_c).abs().max()) < 1e-3


def test_planes_created_after_the_capture_follow_the_replayed_optimizer():
    """A captured train step refreshes the planes that existed at its capture.  A filter that meets its first bf16 launch later
    (here: encoder_4 and decoder_5, whose training launches at batch 1 / 64^2 have fewer than 64 rows -- no bf16 form -- while
    their inference launches at batch 8 take it) must still follow every replayed optimizer step."""
    hip = _hip()
    if not hip.ARITH_BF16:
        pytest.skip('SSC_ARITH=fp32: no planes')
    from sketchyscenecolorization_amd.trainer import Pix2PixTrainer
    from oracle import pix2pix as O
    img = 64
    tr = Pix2PixTrainer(img=img, seed=5)
    b1 = O.synthetic_batch(1, seed=5, img=img)
    dev = {k: (v.cuda() if k != 'text' else v.numpy()) for k, v in b1.items()}
    for it in range(3):             # eager, capture, replay
        tr.train_iteration(dev, dev, it)
    n_before = hip.split_generation()
    b8 = O.synthetic_batch(8, seed=9, img=img)
    sk, nv, text = b8['sketches'].cuda(), b8['noise_vec'].cuda(), b8['text'].numpy()
    tr.generate(sk, text, nv)
    if hip.split_generation() == n_before:
        pytest.skip('no filter met its first bf16 launch in the larger inference pass')
    for it in range(3, 6):          # replays: the weights move, the young planes must follow
        tr.train_iteration(dev, dev, it)
    torch.cuda.synchronize()
    out = tr.generate(sk, text, nv)
    p_now = {n: tr.store[n].detach().cpu() for n in tr.store.names()}
    ref = O.generate_pix2pix(p_now, b8['sketches'], b8['text'], b8['noise_vec'])
    err = float((out.cpu() - ref).abs().max())
    assert err < 1e-3, ('planes created after the capture kept the weights of that moment', err)


def test_volatile_planes_used_in_a_capture_are_never_evicted():
    """Entries of non-parameter filters are dropped once 256 exist -- but never one a captured graph launches into."""
    hip = _hip()
    if not hip.ARITH_BF16:
        pytest.skip('SSC_ARITH=fp32: no planes')
    n, h, ci, co = 2, 16, 64, 64
    x = nhwc(rnd(n, ci, h, h, seed=61)).cuda()
    w = rnd(4, 4, ci, co, seed=62, std=0.05).cuda()
    out = torch.empty(n, 8, 8, co, device='cuda')
    hip.conv_forward(hip.View(x), w, 2, 1, out)     # eager first: kernel attributes, workspace
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        hip.conv_forward(hip.View(x), w, 2, 1, out)
    key = (w.data_ptr(), 16, ci, co, 0)
    assert hip._SPLITS[key].pinned
    keep = []
    for i in range(hip._VOLATILE_MAX + 8):          # force evictions
        wi = torch.zeros(1, 1, 64, 64, device='cuda')
        keep.append(wi)
        hip.filter_split(wi, 0)
    assert key in hip._SPLITS, 'a pinned entry was evicted'
    ref = out.clone()
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_kernel_forms_on_the_128x128_tile_of_16k_stages(tmp_path):
    """conv_bfh_kernel (128 x 128 tile on 16-k LDS stages, two workgroups per CU): the planner picks it for launches of whole
    rounds, which the small parity shapes rarely are -- so the split-vs-exact-vs-float64 cases of this file and the epilogue /
    tail-split / statistics cases of test_gpu_igemm.py run once more in a process whose tile choice is pinned to 128 x 128
    (SSC_FWD_CFG=0, read once per process), and the parity log must show the kernel was really taken."""
    import json
    import subprocess
    import sys
    hip = _hip()
    if not hip.ARITH_BF16:
        pytest.skip('SSC_ARITH=fp32')
    here = os.path.dirname(os.path.abspath(__file__))
    log = str(tmp_path / 'parity.jsonl')
    env = dict(os.environ, SSC_FWD_CFG='0', SSC_PARITY_LOG=log)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(here, 'test_gpu_bf16x6.py'), os.path.join(here, 'test_gpu_igemm.py'),
                        '-m', 'gpu', '-q', '-x', '-k', 'not 128x128_tile and not replayed and not planes and not volatile'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, cwd=os.path.dirname(here))
    assert r.returncode == 0, r.stdout[-2000:]
    recs = [json.loads(l) for l in open(log)]
    taken = [x for x in recs if 'conv_bf16x6<128x128>' in (x.get('kernels') or [])]
    assert len(taken) >= 8, (len(taken), len(recs))
