"""N>1 path on CPU: 2 processes, gloo backend.  Each process is one tower (its own slice of the global
batch, its own batch statistics); flat gradient sections are summed by GradReducer and scaled by
1/world -- which must equal the reference's average_gradients (graph_single.py:33-68): the per-variable
mean of the towers' gradients, NOT the gradient of one big batch (batch-stat norm is per tower)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pix2pix as O

IMG, PER_TOWER, WORLD = 64, 1, 2
NAMES = ['discriminator/layer_2/conv/filter', 'discriminator/layer_3/scale', 'discriminator/fully_connected/weights']


def _tower_grads(p, b, lo, hi):
    sl = {k: v[lo:hi] for k, v in b.items()}
    r = O.build_single_graph(p, **sl)
    return r


def _worker(rank, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    from sketchyscenecolorization_amd.dist_utils import GradReducer, tower_slice
    from sketchyscenecolorization_amd.params import ParamStore
    from oracle import tf_ops as T
    p = O.init_params(0, img=IMG)
    b = O.synthetic_batch(PER_TOWER * WORLD, seed=9, img=IMG)          # the same global batch on every rank
    lo, hi = tower_slice(PER_TOWER * WORLD, PER_TOWER, rank, WORLD)
    r = _tower_grads(p, b, lo, hi)
    st = ParamStore('Pix2Pix', 58, IMG, device='cpu', seed=5)
    st.load_dict(p)
    sc = st.discriminator
    for n, g in r['grad_d'].items():
        sc.g[n].copy_(g)
    red = GradReducer(dist.group.WORLD)
    half = sc.numel // 2 // 64 * 64
    red.reduce_async(sc.grad, 0, half)            # two "sections", like the generator's bucketed exchange
    red.reduce_async(sc.grad, half, sc.numel)
    red.wait()
    assert red.world == WORLD and red.grad_scale == 0.5
    avg = {n: (sc.g[n] * red.grad_scale).clone() for n in NAMES}
    # identical TF-Adam update on every rank keeps the replicas bit-identical
    T.tf_adam_update(sc.flat, sc.grad * red.grad_scale, sc.adam_v, 1, 1e-4)
    torch.save({'avg': avg, 'flat': sc.flat.clone(), 'loss_d': r['loss_d']}, os.path.join(out_dir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_towers_gloo_match_average_gradients(tmp_path):
    port = 29000 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'r0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'r1.pt'))
    assert torch.equal(r0['flat'], r1['flat'])              # replicas stay bit-identical
    # reference semantics computed in one process: mean over towers of per-tower gradients
    p = O.init_params(0, img=IMG)
    b = O.synthetic_batch(PER_TOWER * WORLD, seed=9, img=IMG)
    towers = [_tower_grads(p, b, i * PER_TOWER, (i + 1) * PER_TOWER) for i in range(WORLD)]
    for n in NAMES:
        mean = sum(t['grad_d'][n] for t in towers) / WORLD
        assert float((r0['avg'][n] - mean).abs().max()) <= 1e-6 * max(1.0, float(mean.abs().max())), n
    # and it is NOT the big-batch gradient (norm statistics are per tower)
    big = O.build_single_graph(p, **b)
    n = NAMES[0]
    mean = sum(t['grad_d'][n] for t in towers) / WORLD
    assert float((big['grad_d'][n] - mean).abs().max()) > 1e-4 * float(mean.abs().max())


def test_tower_slice_matches_split_inputs():
    from sketchyscenecolorization_amd.dist_utils import tower_slice
    from sketchyscenecolorization_amd.obj_lib.input_pipeline import split_inputs
    x = np.arange(8)
    for r in range(4):
        lo, hi = tower_slice(8, 2, r, 4)
        assert (split_inputs(x, 2, [1, 1, 1, 1], 4)[r] == x[lo:hi]).all()
