"""Bottleneck-residual generators (FG --block_type Residual, BG 768 generator) vs the oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import parity_log

pytestmark = pytest.mark.gpu

TOL = 1e-3      # BASELINE.json north_star: outputs within 1e-3 max-abs of the reference on fp32 RGB


def _fg_case(img, n, seed, lstm=True):
    from oracle import residual as R
    from sketchyscenecolorization_amd.params import Buffers, ParamStore
    from sketchyscenecolorization_amd.residual import ResidualGenerator
    p = R.init_params('fg', seed=seed, img=img)
    g = torch.Generator().manual_seed(seed + 1)
    z = torch.rand(n, 3, img, img, generator=g) * 2 - 1
    text = torch.zeros(n, 15, dtype=torch.int32)
    for i in range(n):
        k = 3 + i
        text[i, :k] = torch.randint(1, 58, (k,), generator=g, dtype=torch.int32)
    text[0, 1] = 0          # pad token inside a caption is skipped (tf.cond)
    nv = torch.randn(n, 256, generator=g)
    ref = R.generate_residual(p, z, text, nv, lstm_hybrid=lstm)
    store = ParamStore('Residual', 58, img, 'cuda', 0)
    store.load_dict(p)
    gen = ResidualGenerator(store, Buffers('cuda'), 'fg', lstm)
    ctx = gen.forward(z.cuda(), text.numpy(), nv.cuda())
    out = gen.output_nchw(ctx).cpu()
    return out, ref


@pytest.mark.parametrize('img,n,lstm', [(64, 2, True), (64, 3, False), (192, 2, True)])
def test_fg_residual_generator_forward(img, n, lstm):
    out, ref = _fg_case(img, n, 3, lstm)
    err = (out - ref).abs().max().item()
    assert err <= TOL, err


@pytest.mark.parametrize('img,n', [(128, 2), (256, 1)])
def test_bg_residual_generator_forward(img, n):
    from oracle import residual as R
    from sketchyscenecolorization_amd import bg_colorization as bg
    p = R.init_params('bg', seed=5, img=img)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(n, img, img, 3, generator=g) * 2 - 1
    text = torch.zeros(n, 8, dtype=torch.int32)
    text[:, :3] = torch.randint(1, 18, (n, 3), generator=g, dtype=torch.int32)
    text[0, 4] = 7
    # 53 batch-statistics norms deep with only n*(img/32)^2 samples per channel at the bottleneck, fp32 rounding is
    # amplified: the fp32 CPU restatement itself sits 1.5e-3..2e-3 from the float64 evaluation of the same graph.
    # Ground truth is therefore the float64 oracle, and the bar is "1e-3, or no worse than 1.5x the fp32 CPU path".
    ref_img, ref_seg = R.create_residual_generator(p, x, text)
    img64, seg64 = R.create_residual_generator({k: v.double() for k, v in p.items()}, x.double(), text)
    bg.reset()
    store, _, _ = bg.get_tower(img)
    store.load_dict(p)
    out_img, out_seg = bg.create_residual_generator(x, 3, text)
    assert out_img.shape == (n, img, img, 3) and out_seg.shape == (n, img, img, 3)
    e1 = (out_img.cpu().double() - img64).abs().max().item()
    e2 = (out_seg.cpu().double() - seg64).abs().max().item()
    c1 = (ref_img.double() - img64).abs().max().item()
    c2 = (ref_seg.double() - seg64).abs().max().item()
    parity_log('bg_generator_image_vs_f64', dict(n=n, img=img), e1, max(TOL, 1.5 * c1), cpu_fp32_vs_f64=c1, variant='BG', forward=True)
    parity_log('bg_generator_region_logits_vs_f64', dict(n=n, img=img), e2, max(TOL, 1.5 * c2), cpu_fp32_vs_f64=c2, variant='BG', forward=True)
    assert e1 <= max(TOL, 1.5 * c1) and e2 <= max(TOL, 1.5 * c2), (e1, c1, e2, c2)
    assert (out_img.cpu() - ref_img).abs().max().item() <= 4 * TOL


def test_api_generator_residual_matches_tower():
    from sketchyscenecolorization_amd.obj_lib import models_collection as models
    models.reset_default_graph()
    models.set_param('NCHW')
    z = torch.rand(2, 3, 64, 64) * 2 - 1
    text = np.array([[3, 4, 5] + [0] * 12, [7, 8] + [0] * 13], dtype=np.int32)
    nv = torch.randn(2, 256)
    img, nv2 = models.generator_residual(z, text, True, 3, 25, 58, noise_vec=nv)
    assert img.shape == (2, 3, 64, 64) and torch.isfinite(img).all() and float(img.abs().max()) <= 1.0
    disc, logits = models.discriminator_residual(z, img, 25)
    assert disc.shape == (2, 1, 2, 2) and logits.shape == (2, 25)


# --------------------------------------------------------------------------- Residual training path
def _make_trainer(n, img, seed=0):
    from oracle import pix2pix as O
    from oracle import residual as R
    from sketchyscenecolorization_amd.trainer import GanTrainer
    p = R.init_params('fg', seed=seed, with_discriminator=True, img=img)
    tr = GanTrainer(img=img, seed=seed + 1, block_type='Residual')
    tr.store.load_dict(p)
    b = O.synthetic_batch(n, seed=4321 + n, img=img)
    dev = {k: (v.cuda() if k != 'text' else v.numpy()) for k, v in b.items()}
    return p, tr, b, dev


def test_residual_discriminator_forward_parity():
    from oracle import residual as R
    from sketchyscenecolorization_amd import hip
    p, tr, b, dev = _make_trainer(2, 64)
    disc, logits = R.discriminate_residual(p, b['sketches'], b['images_d'])
    xd = torch.zeros(2, 64, 64, 8, device='cuda')
    hip.nchw_to_nhwc(dev['sketches'], xd, 0)
    hip.nchw_to_nhwc(dev['images_d'], xd, 3)
    sn = tr.D.prepare_sn()
    c = tr.D.forward(xd, sn, 'dr')
    assert c['disc'].shape[:3] == (2, 1, 1) or c['disc'].shape[1] == disc.shape[2]
    assert float((c['disc'][..., 0].cpu() - disc[:, 0]).abs().max()) < 1e-3
    assert float((c['logits'].cpu() - logits).abs().max()) < 1e-3


def _grad_errors(get, ref_grads):
    """Per-variable relative L2 and max errors vs the float64 reference."""
    l2s, mxs = {}, {}
    for name, g in ref_grads.items():
        a = get(name).detach().cpu().double()
        l2s[name] = float((a - g).norm() / max(float(g.norm()), 1e-30))
        mxs[name] = float((a - g).abs().max() / max(float(g.abs().max()), 1e-30))
    return l2s, mxs


def _check_grads(scope, ref64, ref32):
    """~100 chained batch-statistics norms (some over as few as 8 samples) make single gradient entries flip with
    fp32 rounding: the torch-CPU fp32 restatement itself is up to 4e-2 (L2) / 1.4e-1 (max) away from its float64
    evaluation for the worst variable, and 2.4e-2 for the MEDIAN generator variable (one flipped relu in the
    discriminator moves everything upstream).  End-to-end criteria are therefore relative to that fp32 CPU path:
    median L2 <= max(3e-3, 1.5x CPU median), worst variable <= max(floor, 2x CPU worst).  The exact formulas of every
    block's backward are pinned separately, in a well-conditioned setting, by test_bottleneck_blocks_backward."""
    l2, mx = _grad_errors(lambda n: scope.g[n], ref64)
    c_l2, c_mx = _grad_errors(lambda n: ref32[n], ref64)
    med, c_med = float(np.median(list(l2.values()))), float(np.median(list(c_l2.values())))
    worst = max(l2.items(), key=lambda kv: kv[1])
    worst_mx = max(mx.items(), key=lambda kv: kv[1])
    assert med < max(3e-3, 1.5 * c_med), (med, c_med)
    assert worst[1] < max(3e-3, 3 * max(c_l2.values())), (worst, max(c_l2.values()))
    # single entries: a flipped relu deep in the chain moves individual filter taps by O(their size) in ANY fp32
    # evaluation; which taps depends on the summation order (tile shape, split-K), so this bound is loose
    assert worst_mx[1] < max(3e-2, 6 * max(c_mx.values())), (worst_mx, max(c_mx.values()))


@pytest.mark.parametrize('n,img', [(2, 64), (2, 192)])
def test_residual_train_step_gradients_parity(n, img):
    """loss_d / loss_g and every gradient of one Residual tower vs float64 autograd on the oracle."""
    from oracle import residual as R
    p, tr, b, dev = _make_trainer(n, img)
    r = R.build_single_graph_f64(p, **b)
    r32 = R.build_single_graph(p, **b)
    ld = tr.d_step(dev, counter=0)
    assert abs(float(ld) - float(r['loss_d'])) < 1e-4 * max(1.0, abs(float(r['loss_d'])))
    _check_grads(tr.store.discriminator, r['grad_d'], r32['grad_d'])
    tr.store.load_dict(p)
    lg = tr.g_step(dev, counter=0)
    assert abs(float(lg) - float(r['loss_g'])) < 1e-4 * max(1.0, abs(float(r['loss_g'])))
    _check_grads(tr.store.generator, r['grad_g'], r32['grad_g'])


@pytest.mark.parametrize('kind', ['de_pu', 'en_pu'])
def test_bottleneck_blocks_backward(kind):
    """One projection bottleneck + one identity bottleneck, forward and hand-written backward, against float64
    autograd on the oracle's blocks: 2x16x16 positions per channel keep the norms well conditioned, so every
    gradient (filters, scales, offsets, both concatenated inputs) must agree to 1e-4 relative L2."""
    from oracle import residual as R
    from sketchyscenecolorization_amd.hip import ACT_LRELU, ACT_RELU
    from sketchyscenecolorization_amd.params import Buffers, ParamStore
    from sketchyscenecolorization_amd.residual import ResidualGenerator, _Val
    g = torch.Generator().manual_seed(3)
    p = R.init_params('fg', seed=2, img=64)
    store = ParamStore('Residual', 58, 64, 'cuda', 0)
    store.load_dict(p)
    gen = ResidualGenerator(store, Buffers('cuda'), 'fg')
    if kind == 'de_pu':
        pre0, pre1, c0, c1, cout, hw, act = 'generator/decoder_3_0', 'generator/decoder_3_1', 256, 256, 128, 8, ACT_RELU
    else:
        pre0, pre1, c0, c1, cout, hw, act = 'generator/encoder_3_0', 'generator/encoder_3_1', 128, 0, 256, 32, ACT_LRELU
    xa = torch.randn(2, c0, hw, hw, generator=g)
    xb = torch.randn(2, c1, hw, hw, generator=g) if c1 else None
    # ---- float64 reference
    q = {k: v.double().requires_grad_(True) for k, v in p.items() if k.startswith(pre0) or k.startswith(pre1)}
    xa64 = xa.double().requires_grad_(True)
    xb64 = xb.double().requires_grad_(True) if c1 else None
    if kind == 'de_pu':
        o = R.bottleneck_residual_de(q, pre0, torch.cat([xa64, xb64], 1))
        o = R.bottleneck_residual_pu(q, pre1, o, False)
    else:
        o = R.bottleneck_residual_en(q, pre0, xa64, 2)
        o = R.bottleneck_residual_pu(q, pre1, o, True)
    gout = torch.randn(o.shape, generator=g)
    names = list(q.keys())
    inputs = [xa64] + ([xb64] if c1 else [])
    grads = torch.autograd.grad((o * gout.double()).sum(), [q[k] for k in names] + inputs)
    # ---- HIP
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    va = _Val(nhwc(xa))
    srcs = (va, _Val(nhwc(xb))) if c1 else (va,)
    gen._tape = []
    if kind == 'de_pu':
        o1 = gen._de('t', pre0, srcs, cout)
    else:
        o1 = gen._en('t', pre0, srcs, cout)
    o2 = gen._pu('t', pre1, o1, act)
    assert float((o2.t.cpu().permute(0, 3, 1, 2) - o.detach().float()).abs().max()) < 1e-4
    gen._gdone = {}
    slot, _ = gen._gslot(o2.t)
    slot.copy_(nhwc(gout))
    for rec in reversed(gen._tape):
        gen._block_backward(rec, gen._gget(rec['out']))
    worst = ('', 0.0)
    for name, gr in zip(names, grads[:len(names)]):
        a = store.grad(name).cpu().double()
        e = float((a - gr).norm() / gr.norm())
        worst = max(worst, (name, e), key=lambda kv: kv[1])
    for src, gr in zip(srcs, grads[len(names):]):
        a = gen._gget(src.t).cpu().double().permute(0, 3, 1, 2)
        e = float((a - gr).norm() / gr.norm())
        worst = max(worst, ('input', e), key=lambda kv: kv[1])
    assert worst[1] < 1e-4, worst


def test_residual_cli_train_smoke(tmp_path, monkeypatch):
    import os
    import obj_colorization_main as cli
    monkeypatch.chdir(tmp_path)
    cli.main(['--mode', 'train', '-bt', 'Residual', '-si', '1', '-bs', '2', '-mi', '3', '-smf', '2', '-swf', '1'])
    run = os.path.join('outputs', sorted(os.listdir('outputs'))[0])
    assert os.path.exists(os.path.join(run, 'snapshot', 'model_1.ckpt-1'))


# --------------------------------------------------------------------------- Background_Colorization training
def test_bg_train_step_real_pass_beside_generator_forward_equals_in_line(monkeypatch):
    """BGTrainer runs D(real) -- forward, loss term, backward -- on a stream of its own beside the generator forward: same
    and the fake pair's discriminator-gradient pass beside the generator backward: same launches in the same order per buffer, so
    the weights after three (eager, captured, replayed) steps are those of the in-line trainer bit for bit."""
    from oracle import residual as R
    from sketchyscenecolorization_amd.bg_colorization import BGTrainer
    img = 128
    b = R.bg_synthetic_batch(2, img, 3)
    dev = (b['inputs'].cuda(), b['targets'].cuda(), b['text'].numpy(), b['labels_gt'].cuda())
    flats = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('SSC_BG_OVERLAP_REAL', mode)
        tr = BGTrainer(image_size=img, max_steps=100, seed=4)
        assert (tr._real_stream is not None) == (mode == '1')
        for _ in range(3):
            tr.train_step(*dev)
        torch.cuda.synchronize()
        flats[mode] = [sc.flat.clone() for sc in (tr.store.generator, tr.store.discriminator)] + [tr.losses.clone()]
    for a, c in zip(flats['0'][:2], flats['1'][:2]):
        assert torch.equal(a, c)
    # the loss words are sums of per-workgroup partials added atomically in double: equal to double rounding, not bitwise
    la, lc = flats['0'][2], flats['1'][2]
    assert float((la - lc).abs().max()) <= 1e-9 * max(1.0, float(la.abs().max()))


def test_bg_train_step_gradients_and_two_steps():
    """BG module: losses, both gradient sets (vs float64 autograd on the oracle, criteria of _check_grads) and two
    Adam(beta1=0.5) steps with the polynomial lr decay tracking the oracle's weights."""
    from oracle import residual as R
    from sketchyscenecolorization_amd.bg_colorization import BGTrainer
    img = 128
    p = R.init_bg_params(0, img=img)
    b = R.bg_synthetic_batch(2, img, 3)
    tr = BGTrainer(image_size=img, max_steps=100)
    tr.store.load_dict(p)
    r64 = R.bg_build_graph_f64(p, **b)
    r32 = R.bg_build_graph(p, **b)
    dev = (b['inputs'].cuda(), b['targets'].cuda(), b['text'].numpy(), b['labels_gt'].cuda())
    tr.gradients(*dev)
    d, g, gan, l1, seg = tr.loss_values()
    assert abs(d - float(r64['discrim_loss'])) < 1e-4 * max(1.0, abs(float(r64['discrim_loss'])))
    assert abs(g - float(r64['gen_loss'])) < 1e-4 * abs(float(r64['gen_loss']))
    assert abs(l1 - float(r64['parts']['gen_loss_L1'])) < 1e-4 and abs(seg - float(r64['parts']['region_mask_loss'])) < 1e-4
    _check_grads(tr.store.discriminator, r64['grad_d'], r32['grad_d'])
    _check_grads(tr.store.generator, r64['grad_g'], r32['grad_g'])
    # two optimizer steps
    st = R.BGTrainState(p)
    for step in range(2):
        R.bg_train_step(p, st, b, step, max_steps=100)
        tr.train_step(*dev)
    assert tr.learning_rate(75) == pytest.approx(2e-5) and tr.learning_rate(0) == pytest.approx(2e-4)
    worst = max((float((tr.store[n].cpu() - p[n]).abs().max()), n) for n in tr.store.names())
    assert worst[0] < 2e-3, worst        # Adam moves every weight by ~lr per step: a sign flip costs up to 2*lr per step
