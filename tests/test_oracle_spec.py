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
