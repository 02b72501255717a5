"""MRU generator (reference default --block_type MRU) vs the oracle, through the C ABI."""
import numpy as np
import pytest
import torch

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
