"""Body of test_gpu_two_ranks.py: the N > 1 train step with TWO real ranks on one GPU.  RCCL refuses two ranks on one device,
so the process group is gloo (it reduces device tensors through host memory); everything else is what N GPUs run: one process
per tower, its own batch and batch statistics, steps captured as graph segments, gradient sections all-reduced on the side stream
between them, 1/world folded into TF-Adam.  Checks, per rank:
  1. after iteration 0 every weight equals a single-process reference that computes both towers' gradients separately, averages
     them per variable (average_gradients, graph_single.py:33-68) and applies TF-Adam once;
  2. the graph-segment trainer equals an eager world-2 trainer over three iterations (capture + replay);
  3. both ranks end with bit-identical weights."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

IMG, NB, WORLD = 64, 2, 2


def worker(rank, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    torch.cuda.set_device(0)
    from sketchyscenecolorization_amd.synthetic import synthetic_batch
    from sketchyscenecolorization_amd.trainer import GanTrainer
    bd = [synthetic_batch(NB, 11 + r, IMG) for r in range(WORLD)]
    bg = [synthetic_batch(NB, 21 + r, IMG) for r in range(WORLD)]
    kw = dict(img=IMG, seed=5, max_iter_step=50)
    T = GanTrainer(use_graphs=True, process_group=dist.group.WORLD, **kw)         # segments: the default for world > 1
    E = GanTrainer(use_graphs=False, process_group=dist.group.WORLD, **kw)
    assert T.world == WORLD and T.segment_graphs and T.reducer.world == WORLD
    refs = [GanTrainer(use_graphs=False, **kw) for _ in range(WORLD)]            # one per tower, world 1

    # 1. iteration 0 against averaged tower gradients
    T.train_iteration(bd[rank], bg[rank], 0)
    for scope, grads, apply, batches in (('discriminator', 'd_gradients', 'apply_d', bd), ('generator', 'g_gradients', 'apply_g', bg)):
        tower = []
        for r in range(WORLD):
            getattr(refs[r], grads)(batches[r])
            tower.append(getattr(refs[r].store, scope).grad.clone())
        mean = sum(tower) / WORLD
        for r in range(WORLD):      # every reference tower applies the same averaged gradient: replicas stay equal
            getattr(refs[r].store, scope).grad.copy_(mean)
            getattr(refs[r], apply)(0)
    torch.cuda.synchronize()
    worst = 0.0
    for n in T.store.names():
        a, b = T.store[n], refs[0].store[n]
        worst = max(worst, float((a - b).abs().max()) / max(1e-3, float(b.abs().max())))
    assert worst < 2e-5, ('iteration 0 vs averaged tower gradients', worst)

    # 2. graph segments (eager, capture, replay) == eager world-2 trainer
    E.train_iteration(bd[rank], bg[rank], 0)
    for it in (1, 2, 3):
        T.train_iteration(bd[rank], bg[rank], it)
        E.train_iteration(bd[rank], bg[rank], it)
    torch.cuda.synchronize()
    segs = [g for g in T._graphs.values() if isinstance(g, list)]
    assert len(segs) == 2, len(segs)
    # the discriminator's exchange is sectioned: layer_4 + layer_5 + class head (the end of the flat buffer) first, beside the
    # backward of layers 3..1, then the rest; together exactly the flat buffer, and what allreduce_plan() declares
    dn = T.store.discriminator.numel
    d_red = [[(op[2], op[3]) for op in g if op[0] == 'reduce' and op[1] is T.store.discriminator.grad] for g in segs]
    d_red = [r for r in d_red if r]
    assert d_red == [[(T._d_late, dn), (0, T._d_late)]], d_red
    plan = T.allreduce_plan()
    assert sum(plan['discriminator_sections_bytes'].values()) == 4 * dn
    g_red = [[(op[2], op[3]) for op in g if op[0] == 'reduce' and op[1] is T.store.generator.grad] for g in segs]
    g_red = [r for r in g_red if r][0]
    assert sum(hi - lo for lo, hi in g_red) * 4 == sum(plan['generator_sections_bytes'].values())
    w2 = max(float((T.store[n] - E.store[n]).abs().max()) for n in T.store.names())
    assert w2 == 0.0, ('segmented graphs vs eager, world 2', w2)

    # 3. replicas
    flat = torch.cat([T.store.generator.flat, T.store.discriminator.flat]).cpu()
    torch.save(flat, os.path.join(out_dir, 'flat%d.pt' % rank))
    dist.barrier()
    if rank == 0:
        other = torch.load(os.path.join(out_dir, 'flat1.pt'))
        assert torch.equal(flat, other), 'replicas diverged'
        print('TWO_RANK_OK worst_vs_reference=%g' % worst, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    import tempfile
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(worker, args=(port, tmp), nprocs=WORLD, join=True)
