"""Pin the torch-CPU oracle (oracle/tf_ops.py) against an independent float64 NumPy loop spec of the
TensorFlow-1 op definitions (SAME padding rule, filter layouts, biased variance,
BasicLSTMCell gate order / forget bias, TF-Adam epsilon placement, spectral norm)."""
import math

import numpy as np
import torch

from oracle import tf_ops as T


def rnd(*shape, seed=0):
    return np.random.RandomState(seed).randn(*shape)


def spec_conv(x, w, stride, pads):
    """x [N,C,H,W], w [kh,kw,Cin,Cout] (HWIO), explicit (top,bottom,left,right) zero pads, VALID after padding."""
    n, c, h, ww = x.shape
    kh, kw, ci, co = w.shape
    xp = np.zeros((n, c, h + pads[0] + pads[1], ww + pads[2] + pads[3]))
    xp[:, :, pads[0]:pads[0] + h, pads[2]:pads[2] + ww] = x
    oh = (xp.shape[2] - kh) // stride + 1
    ow = (xp.shape[3] - kw) // stride + 1
    y = np.zeros((n, co, oh, ow))
    for i in range(oh):
        for j in range(ow):
            patch = xp[:, :, i * stride:i * stride + kh, j * stride:j * stride + kw]      # [n,c,kh,kw]
            y[:, :, i, j] = np.einsum('nchw,hwco->no', patch, w)
    return y


def spec_same_pads(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def test_same_padding_rule():
    assert T.same_pads(192, 7, 2) == (2, 3)       # k=7,s=2,in=192 -> pad (2,3)
    assert T.same_pads(24, 4, 1) == (1, 2)        # k=4,s=1 -> (1,2)
    assert T.same_pads(24, 3, 1) == (1, 1)
    assert T.same_pads(24, 7, 1) == (3, 3)
    assert T.same_pads(12, 4, 2) == (1, 1)


def test_conv_pad1_valid_and_same():
    x, w = rnd(2, 3, 9, 8, seed=1), rnd(4, 4, 3, 5, seed=2)
    for s in (1, 2):
        ref = spec_conv(x, w, s, (1, 1, 1, 1))
        got = T.conv2d_valid_pad(torch.tensor(x), torch.tensor(w), s, 1).numpy()
        assert np.abs(got - ref).max() < 1e-10
    for k, s in ((7, 2), (4, 1), (3, 1)):
        w2 = rnd(k, k, 3, 4, seed=3)
        pt, pb = spec_same_pads(9, k, s)
        pl, pr = spec_same_pads(8, k, s)
        ref = spec_conv(x, w2, s, (pt, pb, pl, pr))
        got = T.conv2d_same(torch.tensor(x), torch.tensor(w2), s).numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-10


def test_conv_transpose_is_gradient_of_same_conv():
    """tf.nn.conv2d_transpose(k4,s2,SAME) == d/dx of conv SAME k4 s2 with filter [kh,kw,out,in] read as HWIO."""
    f = rnd(4, 4, 3, 5, seed=4)            # [kh,kw,Cout=3,Cin=5]
    y = rnd(2, 5, 4, 6, seed=5)            # deconv input [N,Cin=5,4,6] -> output [N,3,8,12]
    out = np.zeros((2, 3, 8, 12))
    # scatter definition: conv SAME k=4 s=2 on 8x12 has pad 1/1: y[o] reads x[2o-1+k]
    for oy in range(4):
        for ox in range(6):
            for ky in range(4):
                for kx in range(4):
                    iy, ix = 2 * oy - 1 + ky, 2 * ox - 1 + kx
                    if 0 <= iy < 8 and 0 <= ix < 12:
                        out[:, :, iy, ix] += np.einsum('nc,oc->no', y[:, :, oy, ox], f[ky, kx])
    got = T.conv2d_transpose_same_s2(torch.tensor(y), torch.tensor(f)).numpy()
    assert np.abs(got - out).max() < 1e-10


def test_batchnorm_biased_variance_and_formula():
    x, sc, of = rnd(3, 4, 5, 6, seed=6) * 2 + 1, rnd(4, seed=7), rnd(4, seed=8)
    mean = x.mean(axis=(0, 2, 3), keepdims=True)
    var = ((x - mean) ** 2).mean(axis=(0, 2, 3), keepdims=True)
    ref = (x - mean) / np.sqrt(var + 1e-5) * sc.reshape(1, -1, 1, 1) + of.reshape(1, -1, 1, 1)
    got = T.batchnorm(torch.tensor(x), torch.tensor(sc), torch.tensor(of)).numpy()
    assert np.abs(got - ref).max() < 1e-10


def test_activations_and_l2norm():
    x = torch.tensor(rnd(3, 7, seed=9))
    assert torch.allclose(T.lrelu(x, 0.2), torch.where(x > 0, x, 0.2 * x))
    assert torch.allclose(T.miu_relu(x), (x + torch.sqrt(0.09 + x * x)) / 2)
    n = T.l2_normalize(x, 1)
    assert torch.allclose((n * n).sum(1), torch.ones(3, dtype=x.dtype))
    assert float(T.l2_normalize(torch.zeros(1, 4), 1).abs().max()) == 0.0


def test_basic_lstm_cell_gate_order_and_forget_bias():
    n, c = 3, 4
    x, st = rnd(n, 5, seed=10), rnd(n, 2 * c, seed=11)
    k, b = rnd(5 + c, 4 * c, seed=12), rnd(4 * c, seed=13)
    cc, h = st[:, :c], st[:, c:]
    g = np.concatenate([x, h], 1) @ k + b
    i, j, f, o = g[:, :c], g[:, c:2 * c], g[:, 2 * c:3 * c], g[:, 3 * c:]
    sig = lambda v: 1 / (1 + np.exp(-v))
    nc = cc * sig(f + 1.0) + sig(i) * np.tanh(j)
    nh = np.tanh(nc) * sig(o)
    h2, st2 = T.basic_lstm_cell(torch.tensor(x), torch.tensor(st), torch.tensor(k), torch.tensor(b))
    assert np.abs(h2.numpy() - nh).max() < 1e-12
    assert np.abs(st2.numpy() - np.concatenate([nc, nh], 1)).max() < 1e-12


def test_tf_adam_three_steps():
    """var -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps); beta1=0 -> m=g (graph_single.py:588)."""
    w = rnd(6, seed=14)
    grads = [rnd(6, seed=20 + t) for t in range(3)]
    v = np.zeros(6)
    wt, vt = torch.tensor(w.copy()), torch.zeros(6, dtype=torch.float64)
    for t, g in enumerate(grads, start=1):
        v = 0.9 * v + 0.1 * g * g
        w = w - 1e-3 * math.sqrt(1 - 0.9 ** t) * g / (np.sqrt(v) + 1e-8)
        T.tf_adam_update(wt, torch.tensor(g), vt, t, 1e-3)
    assert np.abs(wt.numpy() - w).max() < 1e-14


def test_spectral_norm_one_iteration():
    W, u = rnd(5, 4, seed=15), rnd(1, 4, seed=16)
    a = u @ W.T
    v = a / (np.sqrt((a ** 2).sum()) + 1e-12)
    b = v @ W
    u2 = b / (np.sqrt((b ** 2).sum()) + 1e-12)
    sigma = (v @ W @ u2.T)[0, 0]
    wbar, un = T.spectral_normed_weight(torch.tensor(W), torch.tensor(u))
    assert np.abs(wbar.numpy() - W / sigma).max() < 1e-12 and np.abs(un.numpy() - u2).max() < 1e-12


def test_lr_decay_schedule():
    assert T.lr_decay(0, 100000) == 1.0
    assert abs(T.lr_decay(50000, 100000) - 0.55) < 1e-6
    assert abs(T.lr_decay(99999, 100000) - 0.2) < 1e-6


# --------------------------------------------------------------------------- ops of the MRU / Residual / BG oracles
def test_mean_pool_upsample_and_area_resize_specs():
    from oracle import mru as M
    x = rnd(2, 3, 8, 12, seed=4)
    t = torch.tensor(x)
    mp = M.mean_pool(t).numpy()
    up = M.upsample(t).numpy()
    for i in range(4):
        for j in range(6):
            blk = x[:, :, 2 * i:2 * i + 2, 2 * j:2 * j + 2]
            assert np.abs(mp[:, :, i, j] - blk.sum(axis=(2, 3)) / 4.0).max() < 1e-12        # mru.py:15-19
    for i in range(16):
        for j in range(24):
            assert np.array_equal(up[:, :, i, j], x[:, :, i // 2, j // 2])                    # concat x4 + depth_to_space(2)
    assert np.abs(M.mean_pool(M.upsample(t)).numpy() - x).max() < 1e-12
    sq = rnd(1, 3, 8, 8, seed=5)
    area = M.image_resize_area(torch.tensor(sq), 2).numpy()                                  # AREA, integer factor 4
    for i in range(2):
        for j in range(2):
            assert np.abs(area[:, :, i, j] - sq[:, :, 4 * i:4 * i + 4, 4 * j:4 * j + 4].mean(axis=(2, 3))).max() < 1e-12


def test_conditional_batchnorm_and_minmax_gate_specs():
    from oracle import mru as M
    x = rnd(3, 4, 5, 6, seed=6)
    labels = np.array([2, 0, 2])
    p = {'s/offset': torch.tensor(rnd(5, 4, seed=7)), 's/scale': torch.tensor(rnd(5, 4, seed=8))}
    got = M.cond_batchnorm(p, 's', torch.tensor(x), torch.tensor(labels)).numpy()
    mean = x.mean(axis=(0, 2, 3))
    var = ((x - mean[None, :, None, None]) ** 2).mean(axis=(0, 2, 3))                        # biased, over N,H,W
    for n in range(3):
        for c in range(4):
            ref = (x[n, c] - mean[c]) / math.sqrt(var[c] + 1e-5) * p['s/scale'][labels[n], c].item() + \
                p['s/offset'][labels[n], c].item()
            assert np.abs(got[n, c] - ref).max() < 1e-10
    g = M._minmax(torch.tensor(x)).numpy()
    for n in range(3):
        for c in range(4):
            lo, hi = x[n, c].min(), x[n, c].max()
            assert np.abs(g[n, c] - (x[n, c] - lo) / (hi - lo)).max() < 1e-12 and g[n, c].min() == 0.0 and g[n, c].max() == 1.0
    z = np.linspace(-3, 3, 13)
    assert np.abs(T.miu_relu(torch.tensor(z)).numpy() - (z + np.sqrt(0.09 + z * z)) / 2).max() < 1e-12


def test_bottleneck_blocks_against_loop_spec():
    """bottleneck_residual_en / pu / de (residual_util.py:81-171) composed from the loop conv above."""
    from oracle import residual as R
    rs = np.random.RandomState(9)
    cin, cout = 8, 16
    c4 = cout // 4

    def bn(v, scale, offset):
        m = v.mean(axis=(0, 2, 3), keepdims=True)
        var = ((v - m) ** 2).mean(axis=(0, 2, 3), keepdims=True)
        return (v - m) / np.sqrt(var + 1e-5) * scale[None, :, None, None] + offset[None, :, None, None]

    lrelu = lambda v: 0.6 * v + 0.4 * np.abs(v)          # the algebraic form of residual_util.py:45-54
    p = {}
    for blk, shape in (('block_1/conv', (4, 4, cin, c4)), ('block_2/conv_ex', (3, 3, c4, c4)),
                       ('block_3/conv_ex', (1, 1, c4, cout)), ('block_add/conv', (4, 4, cin, cout))):
        p['e/' + blk + '/filter'] = rs.randn(*shape) * 0.2
        b = blk.split('/')[0]
        p['e/%s/batchnorm/scale' % b] = 1 + 0.1 * rs.randn(shape[3])
        p['e/%s/batchnorm/offset' % b] = 0.1 * rs.randn(shape[3])
    x = rs.randn(2, cin, 8, 8)
    y = lrelu(bn(spec_conv(x, p['e/block_1/conv/filter'], 2, (1, 1, 1, 1)), p['e/block_1/batchnorm/scale'],
                 p['e/block_1/batchnorm/offset']))
    y = lrelu(bn(spec_conv(y, p['e/block_2/conv_ex/filter'], 1, (1, 1, 1, 1)), p['e/block_2/batchnorm/scale'],
                 p['e/block_2/batchnorm/offset']))
    y = bn(spec_conv(y, p['e/block_3/conv_ex/filter'], 1, (0, 0, 0, 0)), p['e/block_3/batchnorm/scale'],
           p['e/block_3/batchnorm/offset'])
    sc = bn(spec_conv(x, p['e/block_add/conv/filter'], 2, (1, 1, 1, 1)), p['e/block_add/batchnorm/scale'],
            p['e/block_add/batchnorm/offset'])
    ref = lrelu(y + sc)
    got = R.bottleneck_residual_en({k: torch.tensor(v) for k, v in p.items()}, 'e', torch.tensor(x), 2).numpy()
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-9
    # identity ("pu") unit, decoder flavour: 4x4 s1 SAME pads (1, 2) -- the asymmetric case
    q = {}
    for blk, shape in (('block_1/conv_ex', (4, 4, cout, c4)), ('block_2/conv_ex', (3, 3, c4, c4)),
                       ('block_3/conv_ex', (1, 1, c4, cout))):
        q['u/' + blk + '/filter'] = rs.randn(*shape) * 0.2
        b = blk.split('/')[0]
        q['u/%s/batchnorm/scale' % b] = 1 + 0.1 * rs.randn(shape[3])
        q['u/%s/batchnorm/offset' % b] = 0.1 * rs.randn(shape[3])
    relu = lambda v: np.maximum(v, 0)
    z = relu(bn(spec_conv(ref, q['u/block_1/conv_ex/filter'], 1, (1, 2, 1, 2)), q['u/block_1/batchnorm/scale'],
                q['u/block_1/batchnorm/offset']))
    z = relu(bn(spec_conv(z, q['u/block_2/conv_ex/filter'], 1, (1, 1, 1, 1)), q['u/block_2/batchnorm/scale'],
                q['u/block_2/batchnorm/offset']))
    z = bn(spec_conv(z, q['u/block_3/conv_ex/filter'], 1, (0, 0, 0, 0)), q['u/block_3/batchnorm/scale'],
           q['u/block_3/batchnorm/offset'])
    ref_pu = relu(z + ref)
    got_pu = R.bottleneck_residual_pu({k: torch.tensor(v) for k, v in q.items()}, 'u', torch.tensor(ref), False).numpy()
    assert np.abs(got_pu - ref_pu).max() < 1e-9


def test_bg_losses_and_lr_schedule_specs():
    from oracle import residual as R
    rs = np.random.RandomState(11)
    out, tgt = rs.uniform(-1, 1, (1, 4, 4, 3)), rs.uniform(-1, 1, (1, 4, 4, 3))
    logits = rs.randn(1, 4, 4, 3)
    pr, pf = rs.uniform(0.05, 0.95, (1, 2, 2, 5)), rs.uniform(0.05, 0.95, (1, 2, 2, 5))
    labels = rs.randint(0, 3, (1, 4, 4))
    d, g, parts = R.bg_losses(torch.tensor(out), torch.tensor(logits), torch.tensor(pr), torch.tensor(pf), torch.tensor(tgt),
                              torch.tensor(labels))
    d_ref = np.mean(-(np.log(pr + 1e-12) + np.log(1 - pf + 1e-12)))
    gan_ref = np.mean(-np.log(pf + 1e-12))
    sel = labels.reshape(-1) != 0
    l1_ref = np.abs(tgt - out).reshape(-1, 3)[sel].mean()
    lg = logits.reshape(-1, 3)
    ce = -(lg[np.arange(16), labels.reshape(-1)] - np.log(np.exp(lg).sum(axis=1)))
    assert abs(float(d) - d_ref) < 1e-12 and abs(float(parts['gen_loss_GAN']) - gan_ref) < 1e-12
    assert abs(float(parts['gen_loss_L1']) - l1_ref) < 1e-12 and abs(float(parts['region_mask_loss']) - ce.mean()) < 1e-12
    assert abs(float(g) - (gan_ref + 100 * l1_ref + 100 * ce.mean())) < 1e-9
    # tf.train.polynomial_decay(lr, step, 0.75*max, lr/10, power=0.9), clipped at decay_steps
    assert R.bg_learning_rate(2e-4, 0, 1000) == 2e-4
    assert abs(R.bg_learning_rate(2e-4, 375, 1000) - ((2e-4 - 2e-5) * 0.5 ** 0.9 + 2e-5)) < 1e-18
    assert R.bg_learning_rate(2e-4, 750, 1000) == R.bg_learning_rate(2e-4, 999, 1000) == 2e-5
