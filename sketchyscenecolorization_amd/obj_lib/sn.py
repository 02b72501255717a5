"""spectral_normed_weight (reference obj_lib/sn.py:12-52) on the HIP kernels ssc_sn_forward/backward."""
import torch

from .. import hip

NO_OPS = 'NO_OPS'


def spectral_normed_weight(W, u, num_iters=1, update_collection=None, with_sigma=False):
    """W [..., n] (device, fp32), u [1, n].  Returns W_bar (and sigma); when ``update_collection`` is a
    list the new ``u`` is appended to it (the reference collects assign ops the same way) instead of
    being written back immediately."""
    assert num_iters == 1, 'the reference only ever uses one power iteration'
    n = W.shape[-1]
    W2 = W.reshape(-1, n).contiguous()
    m = W2.shape[0]
    v = torch.empty(m, device=W.device)
    u_new = torch.empty(1, n, device=W.device)
    wbar = torch.empty(m, n, device=W.device)
    aux = torch.empty(4, device=W.device)
    hip.call('ssc_sn_forward', W2, u, m, n, v, u_new, wbar, aux)
    if update_collection is None:
        u.copy_(u_new)
    elif update_collection != NO_OPS:
        update_collection.append((u, u_new))
    wbar = wbar.reshape(W.shape)
    return (wbar, aux[0]) if with_sigma else wbar
