"""TF1 op semantics restated on torch-CPU fp32 (oracle; test infrastructure only).

Every function names the TensorFlow op it restates and the reference call site
(paths relative to /root/reference/Foreground_Instance_Colorization/obj_lib).
Tensors are NCHW at this level, filters keep their TF layouts:
conv ``[kh, kw, Cin, Cout]`` (HWIO), conv-transpose ``[kh, kw, Cout, Cin]``.
"""
import math

import torch
import torch.nn.functional as F


def same_pads(in_size, k, s):
    """tf SAME padding rule: out=ceil(in/s); extra pad goes after (bottom/right)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2


def conv2d_valid_pad(x, filt_hwio, stride, pad):
    """tf.pad(CONSTANT, pad each side) + tf.nn.conv2d(VALID).

    models_collection.py:380-391 (nchw_conv), residual_util.py:16-25 (conv).
    """
    w = filt_hwio.permute(3, 2, 0, 1)
    return F.conv2d(x, w, None, stride=stride, padding=pad)


def conv2d_same(x, filt_hwio, stride, bias=None):
    """tf.nn.conv2d(padding='SAME') with TF's asymmetric padding (mru.py:125)."""
    kh, kw = filt_hwio.shape[0], filt_hwio.shape[1]
    pt, pb = same_pads(x.shape[2], kh, stride)
    pl, pr = same_pads(x.shape[3], kw, stride)
    x = F.pad(x, (pl, pr, pt, pb))
    y = F.conv2d(x, filt_hwio.permute(3, 2, 0, 1), None, stride=stride)
    if bias is not None:
        y = y + bias.reshape(1, -1, 1, 1)
    return y


def conv2d_transpose_same_s2(x, filt_hw_out_in):
    """tf.nn.conv2d_transpose(k=4, strides 2, SAME), output = 2x input.

    models_collection.py:394-405 (nchw_deconv).  It is the gradient of a SAME
    k=4 s=2 conv (pad 1/1), i.e. torch conv_transpose2d(k=4, s=2, p=1) with
    weight [Cin, Cout, kh, kw].
    """
    assert filt_hw_out_in.shape[0] == 4 and filt_hw_out_in.shape[1] == 4
    w = filt_hw_out_in.permute(3, 2, 0, 1)
    return F.conv_transpose2d(x, w, None, stride=2, padding=1)


def batchnorm(x, scale, offset, eps=1e-5):
    """Batch-statistics norm over (N,H,W), no running averages, same in train
    and inference (models_collection.py:36-46).  tf.nn.batch_normalization
    evaluates ``x*inv + (offset - mean*inv)`` with ``inv = rsqrt(var+eps)*scale``.
    """
    mean = x.mean(dim=(0, 2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
    inv = torch.rsqrt(var + eps) * scale.reshape(1, -1, 1, 1)
    return x * inv + (offset.reshape(1, -1, 1, 1) - mean * inv)


def lrelu(x, leak):
    """tf.maximum(leak*x, x) (models_collection.py:51-53)."""
    return torch.maximum(leak * x, x)


def miu_relu(x, miu=0.7):
    """(x + sqrt((1-miu)^2 + x^2))/2 (models_collection.py:63-65)."""
    return (x + torch.sqrt((1 - miu) ** 2 + x ** 2)) / 2.0


def l2_normalize(x, dim, eps=1e-12):
    """tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), eps))."""
    ss = (x * x).sum(dim=dim, keepdim=True)
    return x * torch.rsqrt(torch.clamp(ss, min=eps))


def basic_lstm_cell(x, state, kernel, bias, forget_bias=1.0):
    """tf.nn.rnn_cell.BasicLSTMCell(state_is_tuple=False).

    state = concat([c, h], 1); gates = [x, h] @ kernel + bias split as i, j, f, o;
    c' = c*sigmoid(f+forget_bias) + sigmoid(i)*tanh(j); h' = tanh(c')*sigmoid(o).
    Returns (h', concat([c', h'], 1)).
    """
    n = state.shape[1] // 2
    c, h = state[:, :n], state[:, n:]
    g = torch.cat([x, h], dim=1) @ kernel + bias
    i, j, f, o = g[:, :n], g[:, n:2 * n], g[:, 2 * n:3 * n], g[:, 3 * n:]
    new_c = c * torch.sigmoid(f + forget_bias) + torch.sigmoid(i) * torch.tanh(j)
    new_h = torch.tanh(new_c) * torch.sigmoid(o)
    return new_h, torch.cat([new_c, new_h], dim=1)


def sn_l2normalize(v, eps=1e-12):
    """sn.py:8-9: v / (sum(v^2)^0.5 + eps)."""
    return v / ((v * v).sum() ** 0.5 + eps)


def spectral_normed_weight(w, u):
    """sn.py:12-52 with num_iters=1 and an update collection.

    Gradients flow through the power iteration (no stop_gradient in the
    reference).  Returns (W_bar, u_new); the caller decides when to persist
    u_new (reference: together with opt_g, graph_single.py:178-210).
    """
    w2 = w.reshape(-1, w.shape[-1])
    v = sn_l2normalize(u @ w2.t())
    u_new = sn_l2normalize(v @ w2)
    sigma = (v @ w2 @ u_new.t())[0, 0]
    return (w2 / sigma).reshape(w.shape), u_new


def softplus(x):
    return F.softplus(x)


def sparse_softmax_ce(logits, labels):
    """tf.nn.sparse_softmax_cross_entropy_with_logits (per-sample, no mean)."""
    return F.cross_entropy(logits, labels.long(), reduction='none')


def tf_adam_update(var, grad, v, t, lr, beta1=0.0, beta2=0.9, eps=1e-8, m=None):
    """tf.train.AdamOptimizer dense apply, step index t (1-based).

    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m=b1*m+(1-b1)g; v=b2*v+(1-b2)g^2;
    var -= lr_t*m/(sqrt(v)+eps)   (eps OUTSIDE the bias-corrected sqrt).
    The reference uses beta1=0 (graph_single.py:588) so m == g.
    Updates ``var`` and ``v`` (and ``m`` if given) in place.
    """
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    if m is None:
        m_t = grad * (1.0 - beta1)
    else:
        m.mul_(beta1).add_(grad, alpha=1.0 - beta1)
        m_t = m
    v.mul_(beta2).addcmul_(grad, grad, value=1.0 - beta2)
    var.sub_(lr_t * m_t / (v.sqrt() + eps))
    return var


def tf_rmsprop_update(var, grad, ms, mom, lr, decay=0.9, momentum=0.0, eps=1e-10):
    """tf.train.RMSPropOptimizer dense apply (graph_single.py:586; ``ms`` starts at ones, ``mom`` at zeros)."""
    ms.add_((grad * grad - ms) * (1.0 - decay))
    mom.mul_(momentum).add_(lr * grad / torch.sqrt(ms + eps))
    var.sub_(mom)
    return var


def tf_adagrad_update(var, grad, acc, lr):
    """tf.train.AdagradOptimizer dense apply (graph_single.py:592-593; accumulator starts at 0.1)."""
    acc.add_(grad * grad)
    var.sub_(lr * grad / torch.sqrt(acc))
    return var


def tf_adadelta_update(var, grad, accum, accum_update, lr, rho=0.95, eps=1e-8):
    """tf.train.AdadeltaOptimizer dense apply (graph_single.py:590-591; both slots start at zeros)."""
    accum.mul_(rho).add_(grad * grad, alpha=1.0 - rho)
    upd = torch.sqrt(accum_update + eps) / torch.sqrt(accum + eps) * grad
    accum_update.mul_(rho).add_(upd * upd, alpha=1.0 - rho)
    var.sub_(lr * upd)
    return var


def lr_decay(counter, max_iter_step):
    """graph_single.py:139: max(0.2, 1 - counter/max_iter*0.9) in fp32."""
    c = torch.tensor(float(counter), dtype=torch.float32)
    d = torch.tensor(1.0, dtype=torch.float32) - c / float(max_iter_step) * 0.9
    return float(torch.clamp(d, min=0.2))
