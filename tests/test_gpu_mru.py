"""MRU generator (reference default --block_type MRU) vs the oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import parity_log

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _case(img, n, seed, lstm=True):
    from oracle import mru as M
    g = torch.Generator().manual_seed(seed + 1)
    p = M.init_params(seed, img=img)
    z = torch.rand(n, 3, img, img, generator=g) * 2 - 1
    text = torch.zeros(n, 15, dtype=torch.int32)
    for i in range(n):
        k = 3 + i
        text[i, 15 - k:] = torch.randint(1, 58, (k,), generator=g, dtype=torch.int32)
    labels = torch.randint(0, 25, (n,), generator=g, dtype=torch.int32)
    nv = torch.randn(n, 256, generator=g)
    return p, z, text, labels, nv


@pytest.mark.parametrize('img,n,lstm', [(64, 2, True), (64, 3, False), (192, 2, True)])
def test_mru_generator_forward(img, n, lstm):
    from oracle import mru as M
    from sketchyscenecolorization_amd.mru import MRUGenerator
    from sketchyscenecolorization_amd.params import Buffers, ParamStore
    p, z, text, labels, nv = _case(img, n, 11, lstm)
    ref, inter = M.generate_mru(p, z, text, labels, nv, lstm_hybrid=lstm, return_all=True)
    ref64 = M.generate_mru({k: v.double() for k, v in p.items()}, z.double(), text, labels, nv.double(), lstm_hybrid=lstm)
    store = ParamStore('MRU', 58, img, 'cuda', 0)
    store.load_dict(p)
    gen = MRUGenerator(store, Buffers('cuda'), lstm)
    ctx = gen.forward(z.cuda(), text.numpy(), labels.cuda(), nv.cuda())
    out = gen.output_nchw(ctx).cpu()
    for k, (a, b) in enumerate(zip(ctx['enc'], inter['enc'])):
        e = (a.cpu().permute(0, 3, 1, 2) - b).abs().max().item()
        assert e <= 1e-3 * max(1.0, b.abs().max().item()), ('enc', k, e)
    for k, (a, b) in enumerate(zip(ctx['dec'], inter['dec'])):
        e = (a.cpu().permute(0, 3, 1, 2) - b).abs().max().item()
        assert e <= 2e-3 * max(1.0, b.abs().max().item()), ('dec', k, e)
    err = (out.double() - ref64).abs().max().item()
    cpu = (ref.double() - ref64).abs().max().item()
    parity_log('mru_generator_forward_vs_f64', dict(n=int(z.shape[0]), img=int(z.shape[-1])), err, max(TOL, 1.5 * cpu), cpu_fp32_vs_f64=cpu,
               variant='MRU', forward=True)
    assert err <= max(TOL, 1.5 * cpu), (err, cpu)


def test_api_generator_mru_and_single_graph_inference():
    from sketchyscenecolorization_amd.obj_lib import models_collection as models
    from sketchyscenecolorization_amd.obj_lib.graph_single import build_single_graph
    models.reset_default_graph()
    models.set_param('NCHW')
    z = torch.rand(1, 3, 64, 64) * 2 - 1
    text = np.array([[0] * 12 + [3, 4, 5]], dtype=np.int32)
    nv = torch.randn(1, 256)
    img, _ = models.generator_mru(z, text, True, 3, 25, 58, labels=np.array([7]), noise_vec=nv)
    gen, _, sk = build_single_graph(z, z, None, np.array([7]), None, text, batch_size=1, training=False,
                                    LSTM_hybrid=True, vocab_size=58, noise_vec=nv)      # default block_type = 'MRU'
    assert img.shape == (1, 3, 64, 64) and torch.isfinite(img).all()
    assert torch.equal(gen, img)
    with pytest.raises(ValueError):
        models.generator_mru(z, text, True, 3, 25, 58)
    disc, logits = models.discriminator_mru(z, gen, 25)
    assert disc.shape == (1, 1, 4, 4) and logits.shape == (1, 25)


# --------------------------------------------------------------------------- MRU training path
def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def _nhwc(t, pad_to=None):
    t = t.permute(0, 2, 3, 1).contiguous()
    if pad_to is not None and t.shape[-1] < pad_to:
        t = torch.cat([t, torch.zeros(t.shape[:-1] + (pad_to - t.shape[-1],))], -1)
    return t.cuda()


@pytest.mark.parametrize('mode', ['gen_conv', 'gen_deconv', 'gen_deconv_noproj', 'disc_conv'])
def test_mru_blocks_backward(mode):
    """One MRU block, forward + hand-written backward, against float64 autograd on the oracle's block
    (well-conditioned sizes): every parameter gradient and every input gradient to 2e-4 relative L2."""
    from oracle import mru as M
    from sketchyscenecolorization_amd.mru import MRUDiscriminator, MRUGenerator
    from sketchyscenecolorization_amd.params import Buffers, ParamStore
    g = torch.Generator().manual_seed(5)
    p = M.init_params(3, with_discriminator=True, img=64)
    store = ParamStore('MRU', 58, 64, 'cuda', 0)
    store.load_dict(p)
    bufs = Buffers('cuda')
    n = 3
    labels = torch.tensor([4, 17, 4], dtype=torch.int32)
    if mode == 'gen_conv':
        pre, ch, d, hw = 'generator/mru_conv_unit_t_2_layer_0', 64, 128, 16
    elif mode == 'disc_conv':
        pre, ch, d, hw = 'discriminator/mru_conv_unit_t_2_layer_0', 128, 256, 16
    elif mode == 'gen_deconv':
        pre, ch, d, hw, cs = 'generator/mru_deconv_unit_t_4_layer_0', 256, 128, 8, 64
    else:
        pre, ch, d, hw, cs = 'generator/mru_deconv_unit_t_6_layer_0', 128, 128, 8, 8
    q = {k: v.double().requires_grad_(not k.endswith('/u')) for k, v in p.items() if k.startswith(pre)}
    ht = torch.randn(n, ch, hw, hw, generator=g)
    ht64 = ht.double().requires_grad_(True)
    if mode in ('gen_conv', 'disc_conv'):
        x = torch.rand(n, 3, hw, hw, generator=g) * 2 - 1
        x64 = x.double().requires_grad_(True)
        if mode == 'gen_conv':
            o = M.mru_conv_block_v3(q, pre, x64, ht64, d, labels.long(), 2)
        else:
            o = M._d_conv_block(q, pre, x64, ht64, d, {})
        ins64 = [ht64, x64]
    else:
        z = torch.rand(n, 3, 2 * hw, 2 * hw, generator=g) * 2 - 1
        sk = torch.randn(n, cs, 2 * hw, 2 * hw, generator=g)
        sk64 = sk.double().requires_grad_(True)
        o = M.mru_deconv_block_v2(q, pre, torch.cat([z.double(), sk64], 1), ht64, d, labels.long(), 2)
        ins64 = [ht64, sk64]
    gout = torch.randn(o.shape, generator=g)
    names = [k for k in q if not k.endswith('/u')]
    grads = torch.autograd.grad((o * gout.double()).sum(), [q[k] for k in names] + ins64)
    ref = dict(zip(names, grads[:len(names)]))
    # ---- HIP
    net = MRUDiscriminator(store, bufs) if mode == 'disc_conv' else MRUGenerator(store, bufs)
    tape = []
    ht_d = _nhwc(ht)
    lab_d = labels.cuda()
    if mode in ('gen_conv', 'disc_conv'):
        x_d = _nhwc(x, 4)
        if mode == 'disc_conv':
            sn = net.prepare_sn()
            net._sn = sn
            lab_d = None
        out = net._conv_block('t', pre, x_d, ht_d, d, lab_d, tape)
    else:
        z_d, sk_d = _nhwc(z, 4), _nhwc(sk)
        out = net._deconv_block('t', pre, z_d, sk_d, ht_d, d, lab_d, tape)
    assert _rel(out.permute(0, 3, 1, 2), o) < 1e-5
    net._gdone = {}
    slot, _ = net._gslot(out)
    slot.copy_(_nhwc(gout))
    if mode in ('gen_conv', 'disc_conv'):
        g_x = torch.zeros_like(x_d)
        net._conv_block_backward(tape[0], lab_d, True, False, g_x)
        if mode == 'disc_conv':
            net.finish_sn_backward(sn)
        got_inputs = [net._gget(ht_d).permute(0, 3, 1, 2), g_x[..., :3].permute(0, 3, 1, 2)]
    else:
        net._deconv_block_backward(tape[0], lab_d)
        got_inputs = [net._gget(ht_d).permute(0, 3, 1, 2), net._gget(sk_d).permute(0, 3, 1, 2)]
    worst = ('', 0.0)
    big = max(float(v.norm()) for v in ref.values())
    for k in names:
        got = store.grad(k).reshape(ref[k].shape)
        if float(ref[k].norm()) < 1e-7 * big:       # e.g. a bias feeding a batch norm: exactly zero gradient
            assert float(got.norm()) < 1e-4 * big, k
            continue
        worst = max(worst, (k, _rel(got, ref[k])), key=lambda kv: kv[1])
    for i, (a, b) in enumerate(zip(got_inputs, grads[len(names):])):
        worst = max(worst, ('input%d' % i, _rel(a, b)), key=lambda kv: kv[1])
    assert worst[1] < 2e-4, worst


def _make_trainer(n, img, seed=0):
    from oracle import mru as M
    from oracle import pix2pix as O
    from sketchyscenecolorization_amd.trainer import GanTrainer
    p = M.init_params(seed, with_discriminator=True, img=img)
    tr = GanTrainer(img=img, seed=seed + 1, block_type='MRU')
    tr.store.load_dict(p)
    b = O.synthetic_batch(n, seed=987 + n, img=img)
    dev = {k: (v.cuda() if k != 'text' else v.numpy()) for k, v in b.items()}
    return p, tr, b, dev


def test_mru_discriminator_forward_parity():
    from oracle import mru as M
    from sketchyscenecolorization_amd import hip
    p, tr, b, dev = _make_trainer(2, 64)
    disc, logits, us = M.discriminate_mru(p, b['sketches'], b['images_d'], return_u=True)
    xd = torch.zeros(2, 64, 64, 8, device='cuda')
    hip.nchw_to_nhwc(dev['sketches'], xd, 0)
    hip.nchw_to_nhwc(dev['images_d'], xd, 3)
    sn = tr.D.prepare_sn()
    c = tr.D.forward(xd, sn, 'dr')
    assert float((c['disc'][..., 0].cpu() - disc[:, 0]).abs().max()) < 1e-3 * max(1.0, float(disc.abs().max()))
    assert float((c['logits'].cpu() - logits).abs().max()) < 1e-3 * max(1.0, float(logits.abs().max()))
    for k, u in us.items():
        assert _rel(sn[k[:-2]]['u_new'], u) < 1e-4, k


def _grad_errors(get, ref_grads):
    """Relative L2 per variable vs float64, the denominator floored at 1e-4 of the largest gradient norm (biases that
    feed a batch norm have an exactly-zero gradient; scalar prelu leaks can have tiny ones)."""
    l2s = {}
    big = max(float(g.norm()) for g in ref_grads.values())
    for name, g in ref_grads.items():
        a = get(name).reshape(g.shape).detach().cpu().double()
        l2s[name] = float((a - g).norm() / max(float(g.norm()), 1e-4 * big))
    return l2s


@pytest.mark.parametrize('n,img,noise', [(2, 64, True), (2, 192, True), (2, 64, False), (2, 192, False)])
def test_mru_train_step_gradients_parity(n, img, noise):
    """loss_d / loss_g and every gradient of one MRU tower vs float64 autograd on the oracle.

    The min-max gates (mru.py:414-415, 560-568) send gradient to the arg-min / arg-max position of every (sample,
    channel) plane.  On sketches (large flat regions) several positions are within fp32 rounding of the extremum, so
    WHICH one is selected differs between any two fp32 evaluations -- observed on both sides: HIP 5e-3 vs torch-CPU
    fp32 5e-5 on one input, HIP 3.5e-5 vs torch-CPU 4e-4 on another -- and one flipped selection shifts every
    upstream variable by the same relative amount.  On TIE-FREE inputs (uniform noise instead of sketches: every plane
    has a unique extremum) the whole tower is therefore held to the tight bar -- median relative L2 < 2e-3 per scope
    (measured 1e-5 .. 6e-4, the torch-CPU fp32 oracle itself 1e-5 .. 4e-4), worst tensor-valued variable < 1e-2, scalars (prelu leaks) < 3e-2 (see below) -- at 64x64 and at the full 192x192; on sketches the bar stays loose (median < 2e-2; a wrong formula
    gives O(1)) and the exact formulas are pinned at 2e-4 by test_mru_blocks_backward."""
    _mru_gradients_parity(n, img, noise)


def test_mru_train_step_gradients_parity_exact_fp32(monkeypatch):
    """The same check with every contraction on the exact-fp32 kernels (SSC_ARITH=fp32 semantics, switched at run time): there
    the scalar prelu leaks are held to the plain bar (no selection noise of the bf16x6 variants to allow for)."""
    from sketchyscenecolorization_amd import hip
    monkeypatch.setattr(hip, 'ARITH_BF16', False)
    _mru_gradients_parity(2, 64, True)


def _mru_gradients_parity(n, img, noise):
    from oracle import mru as M
    p, tr, b, dev = _make_trainer(n, img)
    if noise:
        b['sketches'] = torch.rand(b['sketches'].shape, generator=torch.Generator().manual_seed(1)) * 2 - 1
        dev['sketches'] = b['sketches'].cuda()
    r = M.build_single_graph_f64(p, **b)
    ld = tr.d_step(dev, counter=0)
    assert abs(float(ld) - float(r['loss_d'])) < 1e-4 * max(1.0, abs(float(r['loss_d'])))
    ed = _grad_errors(lambda k: tr.store.discriminator.g[k], r['grad_d'])
    tr.store.load_dict(p)
    lg = tr.g_step(dev, counter=0)
    assert abs(float(lg) - float(r['loss_g'])) < 1e-4 * max(1.0, abs(float(r['loss_g'])))
    eg = _grad_errors(lambda k: tr.store.generator.g[k], r['grad_g'])
    med_tol, worst_tol = (2e-3, 1e-2) if noise else (2e-2, 2e-1)
    # SCALAR variables (the discriminator's prelu leaks: one number summed over a whole tensor, |g| ~ 1e-4 of the largest gradient
    # norm, i.e. at the floor of _grad_errors' denominator) get 3 x the bar: with ~4000 min-max planes per tower the closest
    # runner-up of an arg-extremum is within fp32 rounding of it even on noise inputs, and one flipped selection moves such a
    # scalar by ~2e-6 of the largest gradient norm = 1e-2 of the floor.  Measured on one input over the arithmetic variants of
    # round 5 (exact fp32, bf16x6 convs, bf16x6 filter gradients, partial-chunk bf16x6): 5.3e-4, 1.0e-3, 2.0e-3, 6.7e-3, 1.05e-2,
    # 1.3e-2 (the torch-CPU fp32 oracle: 6.3e-4 .. 7.8e-4); the discriminator's tensor-valued variables stay below 3e-4
    # (scripts/probes_r05/mru_prelu_grad_probe.py).
    scalars = {k for k, g in list(r['grad_d'].items()) + list(r['grad_g'].items()) if g.numel() == 1}
    for e in (ed, eg):
        assert float(np.median(list(e.values()))) < med_tol, float(np.median(list(e.values())))
        worst = max(((k, v) for k, v in e.items() if k not in scalars), key=lambda kv: kv[1])
        assert worst[1] < worst_tol, worst
        # the exact-fp32 arithmetic (SSC_ARITH=fp32: measured 5e-4 .. 2e-3) keeps the plain bar, so that a regression of the
        # leak-gradient path itself still shows there; the 3 x is the bf16x6 variants' selection noise only
        from sketchyscenecolorization_amd import hip as _h
        scalar_tol = 3 * worst_tol if _h.ARITH_BF16 else worst_tol
        for k in scalars & set(e):
            assert e[k] < scalar_tol, (k, e[k], scalar_tol)
    for k, u in r['u_new'].items():          # the G-step commits every spectral-norm u (graph_single.py:178-210)
        assert _rel(tr.store[k], u) < 1e-3, k


def test_mru_cli_train_smoke(tmp_path, monkeypatch):
    import os
    import obj_colorization_main as cli
    monkeypatch.chdir(tmp_path)
    cli.main(['--mode', 'train', '-si', '1', '-bs', '2', '-mi', '3', '-smf', '2', '-swf', '1'])     # default -bt MRU
    run = os.path.join('outputs', sorted(os.listdir('outputs'))[0])
    assert os.path.exists(os.path.join(run, 'snapshot', 'model_1.ckpt-1'))
