"""Edge cases of the training / inference paths for all three block types: no caption branch (--lstm_hybrid 0),
all-pad captions (every tf.cond takes the skip branch), batch 1, the small 64x64 image mode."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(bt, img, n, seed=0, lstm=True):
    from oracle import mru as M
    from oracle import pix2pix as O
    from oracle import residual as R
    from sketchyscenecolorization_amd.trainer import GanTrainer
    if bt == 'Pix2Pix':
        p = O.init_params(seed, img=img)
        mod = O
    elif bt == 'Residual':
        p = R.init_params('fg', seed=seed, with_discriminator=True, img=img)
        mod = R
    else:
        p = M.init_params(seed, with_discriminator=True, img=img)
        mod = M
    graph = lambda b, f64=True, **kw: (mod.build_single_graph_f64 if f64 else mod.build_single_graph)(p, **b, **kw)
    tr = GanTrainer(img=img, seed=seed + 1, block_type=bt, lstm_hybrid=lstm)
    tr.store.load_dict(p)
    b = O.synthetic_batch(n, seed=555 + n, img=img)
    return p, tr, b, graph


def _dev(b):
    return {k: (v.cuda() if k != 'text' else v.numpy()) for k, v in b.items()}


def _median_rel(get, ref):
    big = max(float(g.norm()) for g in ref.values())
    errs = [float((get(k).detach().cpu().double().reshape(g.shape) - g).norm() / max(float(g.norm()), 1e-4 * big))
            for k, g in ref.items()]
    return float(np.median(errs))


@pytest.mark.parametrize('bt', ['Pix2Pix', 'Residual', 'MRU'])
def test_training_without_caption_branch(bt):
    """--lstm_hybrid 0: the generator skips encode_feat_with_text; its variables get zero gradients."""
    p, tr, b, graph = _setup(bt, 64, 2, lstm=False)
    r = graph(b, lstm_hybrid=False)
    dev = _dev(b)
    ld = float(tr.d_step(dev, counter=0))
    assert abs(ld - float(r['loss_d'])) < 1e-4 * max(1.0, abs(float(r['loss_d'])))
    tr.store.load_dict(p)
    lg = float(tr.g_step(dev, counter=0))
    assert abs(lg - float(r['loss_g'])) < 1e-4 * max(1.0, abs(float(r['loss_g'])))
    # ill-conditioned end to end (see tests/test_gpu_residual.py::_check_grads): relative to the fp32 CPU path
    r32 = graph(b, f64=False, lstm_hybrid=False)
    cpu = _median_rel(lambda k: r32['grad_g'][k], r['grad_g'])
    assert _median_rel(lambda k: tr.store.generator.g[k], r['grad_g']) < max(3e-2, 2 * cpu), cpu
    for k, g in tr.store.generator.g.items():
        if '/TextLSTM/' in k:
            assert float(g.abs().max()) == 0.0, k


@pytest.mark.parametrize('bt', ['Pix2Pix', 'Residual', 'MRU'])
def test_all_pad_captions_and_batch_one(bt):
    """A caption of zeros only: every step takes tf.cond's skip branch and the fused feature is relu(atanh(0)) = 0;
    combined with batch 1 (batch statistics over a single sample)."""
    p, tr, b, graph = _setup(bt, 64, 1)
    b['text'].zero_()
    r = graph(b)
    dev = _dev(b)
    if bt == 'MRU':
        out = tr.generate(dev['sketches'], dev['text'], dev['noise_vec'], labels=dev['class_id'])
    else:
        out = tr.generate(dev['sketches'], dev['text'], dev['noise_vec'])
    assert float((out.cpu().double() - r['gen']).abs().max()) < 1e-3
    ld = float(tr.d_step(dev, counter=0))
    tr.store.load_dict(p)
    lg = float(tr.g_step(dev, counter=0))
    assert abs(ld - float(r['loss_d'])) < 1e-3 * max(1.0, abs(float(r['loss_d'])))
    assert abs(lg - float(r['loss_g'])) < 1e-3 * max(1.0, abs(float(r['loss_g'])))
    assert all(bool(torch.isfinite(g).all()) for g in tr.store.generator.g.values())


@pytest.mark.gpu
def test_device_pre_and_post_processing_match_the_oracle():
    """hip.sketch_preprocess_u8 / image_postprocess_u8 against oracle/image_ops.py (NumPy restatement of main_procedure.py's
    normalisation, thicken_drawings and truncating uint8 cast) -- bit exact (uint8 / float32 arithmetic)."""
    import numpy as np
    from oracle import image_ops as I
    from sketchyscenecolorization_amd import hip
    rng = np.random.RandomState(0)
    u8 = rng.randint(0, 256, (3, 40, 56, 3)).astype(np.uint8)
    u8[1] = np.repeat(((rng.rand(40, 56) > 0.9) * 255).astype(np.uint8)[:, :, None], 3, axis=2)     # a sparse drawing
    dev = torch.from_numpy(u8).cuda()
    got = hip.sketch_preprocess_u8(dev).cpu().numpy()
    assert np.array_equal(got[..., :3], I.sketch_preprocess(u8)) and (got[..., 3] == 0).all()
    got_t = hip.sketch_preprocess_u8(dev, thicken=True).cpu().numpy()
    assert np.array_equal(got_t[..., :3], I.sketch_preprocess(u8, thicken=True))
    x = (rng.rand(2, 24, 32, 8).astype(np.float32) * 2 - 1)
    x[0, 0, 0, 3:6] = (1.0, -1.0, 0.0)
    out = hip.image_postprocess_u8(torch.from_numpy(x).cuda(), coff=3).cpu().numpy()
    assert out.dtype == np.uint8 and np.array_equal(out, I.image_postprocess(x[..., 3:6]))


@pytest.mark.gpu
@pytest.mark.parametrize('block_type', ['Pix2Pix', 'Residual', 'MRU'])
def test_generate_u8_equals_host_pipeline(block_type):
    """GanTrainer.generate_u8 (uint8 in, uint8 out, everything on the device, NHWC straight into the network) equals
    host normalise -> generate (NCHW) -> host post-process."""
    import numpy as np
    from oracle import image_ops as I
    from sketchyscenecolorization_amd.trainer import GanTrainer
    tr = GanTrainer(img=64, seed=2, block_type=block_type)
    rng = np.random.RandomState(1)
    u8 = np.repeat(((rng.rand(2, 64, 64) > 0.85) * 255).astype(np.uint8)[..., None], 3, axis=3)
    text = rng.randint(1, 58, (2, 15)).astype(np.int32)
    noise = torch.randn(2, 256, device='cuda')
    labels = torch.tensor([3, 7], dtype=torch.int32, device='cuda')
    got = tr.generate_u8(torch.from_numpy(u8).cuda(), text, noise, labels=labels).cpu().numpy()
    z = torch.from_numpy(np.ascontiguousarray(I.sketch_preprocess(u8).transpose(0, 3, 1, 2))).cuda()
    ref = I.image_postprocess(tr.generate(z, text, noise, labels=labels).cpu().numpy().transpose(0, 2, 3, 1))
    assert got.shape == (2, 64, 64, 3) and np.array_equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize('size', [192, 64])
def test_device_decode_of_training_records_matches_the_oracle(size):
    """hip.decode_paired_u8 against oracle/image_ops.decode_paired_example (NumPy restatement of get_paired_input,
    input_pipeline.py:77-131) on random 384x384 records, with the same dequantisation noise: bit exact.  The product's own
    host decode (input_pipeline.decode_paired_example) is held to the same oracle."""
    import numpy as np
    from oracle import image_ops as I
    from sketchyscenecolorization_amd import hip
    from sketchyscenecolorization_amd.obj_lib.input_pipeline import RECORD_HW, decode_paired_example
    rng = np.random.RandomState(size)
    n = 3
    img = rng.randint(0, 256, (n, RECORD_HW, RECORD_HW, 3)).astype(np.uint8)
    img[1] = rng.randint(40, 200, (RECORD_HW, RECORD_HW, 3))        # a narrower range: min / max normalisation matters
    sk = (rng.rand(n, RECORD_HW, RECORD_HW, 3) > 0.1).astype(np.uint8) * 255
    noise = rng.uniform(0.0, 1.0 / 256, size=(n, size, size, 3)).astype(np.float32)

    class FixedNoise(object):
        def __init__(self, a):
            self.a = a

        def uniform(self, lo, hi, size=None):
            assert tuple(size) == self.a.shape
            return self.a

    gi, gs = hip.decode_paired_u8(torch.from_numpy(img).cuda(), torch.from_numpy(sk).cuda(), size,
                                  noise=torch.from_numpy(noise).cuda())
    gi, gs = gi.cpu().numpy(), gs.cpu().numpy()
    for k in range(n):
        ri, rs = I.decode_paired_example(img[k], sk[k], size, noise[k])
        assert np.array_equal(gi[k], ri) and np.array_equal(gs[k], rs), k
        feat = {'cartoon_data': [img[k].tobytes()], 'sketch_data': [sk[k].tobytes()], 'Category_id': [1],
                'Text_vocab_indices': [bytes(15)]}
        hi, hs = decode_paired_example(feat, (size, size), FixedNoise(noise[k]))[:2]
        assert np.array_equal(hi, ri) and np.array_equal(hs, rs), k


@pytest.mark.gpu
def test_device_distance_map_matches_scipy_edt():
    """--distance_map 1: hip.distance_map_u8 / decode_paired_u8(distance_map=True) against the host path
    (scipy.ndimage.distance_transform_edt over the [384,384,3] array, input_pipeline.py:86-96): bit exact, also when
    the three channels differ (the channel axis is a spatial axis of the reference's transform)."""
    import numpy as np
    from oracle import image_ops as I
    from sketchyscenecolorization_amd import hip
    from sketchyscenecolorization_amd.obj_lib.input_pipeline import RECORD_HW
    rng = np.random.RandomState(7)
    n, size = 2, 192
    img = rng.randint(0, 256, (n, RECORD_HW, RECORD_HW, 3)).astype(np.uint8)
    sk = np.full((n, RECORD_HW, RECORD_HW, 3), 255, np.uint8)
    sk[0, 100:103, 40:300] = 0
    sk[0, 200:330, 250:252] = 30
    pts = rng.randint(0, RECORD_HW, (60, 2))
    sk[1, pts[:, 0], pts[:, 1], :] = 0
    sk[1, 20:24, 20:24, 1] = 100            # a stroke in one channel only
    noise = np.zeros((n, size, size, 3), np.float32)

    gi, gs = hip.decode_paired_u8(torch.from_numpy(img).cuda(), torch.from_numpy(sk).cuda(), size,
                                  noise=torch.from_numpy(noise).cuda(), distance_map=True)
    for k in range(n):
        ri, rs = I.decode_paired_example(img[k], sk[k], size, noise[k], distance_map=True)
        assert np.array_equal(gs[k].cpu().numpy(), rs), (k, float(np.abs(gs[k].cpu().numpy() - rs).max()))
        assert np.array_equal(gi[k].cpu().numpy(), ri)


@pytest.mark.gpu
def test_device_resize_matches_pillow_fixtures():
    """ssc_resample_u8 (Pillow's 8-bit two-pass resampler as kernels) against the outputs of the Pillow installed in the
    build container, stored as fixtures (tests/golden/resize_goldens.npz, made by make_resize_goldens.py): LANCZOS =
    resize_and_padding_mask_image (input_pipeline.py:199-239), bilinear = reverse_resize_image
    (Pipeline_utils/fg_color_utils.py:137-160).  Bit exact."""
    import os
    import numpy as np
    from sketchyscenecolorization_amd import hip
    from sketchyscenecolorization_amd.obj_lib import input_pipeline as ip
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'resize_goldens.npz'))
    n = int(g['n_cases'])
    for i in range(n):
        src, filt, ref = g['src_%d' % i], str(g['filt_%d' % i]), g['out_%d' % i]
        nh, nw = ref.shape[:2]
        out = hip.resample_u8(torch.from_numpy(src).cuda(), nh, nw, ip._dev_coeffs(src.shape[1], nw, filt),
                              ip._dev_coeffs(src.shape[0], nh, filt)).cpu().numpy()
        assert np.array_equal(out, ref), (i, filt, src.shape, ref.shape)
    for i in range(int(g['n_pad'])):
        src, size, margin, ref = g['pad_src_%d' % i], int(g['pad_size_%d' % i]), int(g['pad_margin_%d' % i]), g['pad_out_%d' % i]
        out = ip.resize_and_padding_mask_image_device(torch.from_numpy(src).cuda(), size, margin).cpu().numpy()
        assert np.array_equal(out, ref), (i, src.shape, size, margin)
    for i in range(int(g['n_rev'])):
        src, bh, bw, margin, ref = (g['rev_src_%d' % i], int(g['rev_bh_%d' % i]), int(g['rev_bw_%d' % i]),
                                    int(g['rev_margin_%d' % i]), g['rev_out_%d' % i])
        out = ip.reverse_resize_image_device(torch.from_numpy(src).cuda(), bh, bw, margin_size=margin).cpu().numpy()
        assert np.array_equal(out, ref), (i, bh, bw, margin)
