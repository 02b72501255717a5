"""Body of test_gpu_pix2pix.py::test_segmented_graphs_with_rccl_world1_match_eager, run in its own interpreter."""
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd.synthetic import synthetic_batch      # noqa: E402
from sketchyscenecolorization_amd.trainer import GanTrainer             # noqa: E402


def main():
    with socket.socket() as sock:           # any free port: the box may already use the default one
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    try:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                                device_id=torch.device('cuda', 0))
    except Exception as e:                  # no RCCL on this box: the protocol itself is covered by the gloo test
        print('SKIP cannot create a 1-rank RCCL group here: %r' % (e,))
        return 0
    a = GanTrainer(img=64, seed=5, max_iter_step=50)
    b = GanTrainer(img=64, seed=5, max_iter_step=50, use_graphs=True, segment_graphs=True,
                   process_group=dist.group.WORLD)
    b.reducer.world = 2          # exercise the collective calls; a 1-rank all-reduce leaves the data unchanged
    b.reducer.stream = torch.cuda.Stream()
    b.world = 1
    bd, bg = synthetic_batch(2, 11, 64), synthetic_batch(2, 12, 64)
    for it in range(4):
        la = (float(a.d_step(bd, it)), float(a.g_step(bg, it)))
        lg, ld = b.train_iteration(bd, bg, it)      # with the G-step's generator forward run ahead inside the D-step
        lb = (float(ld), float(lg))
        assert abs(la[0] - lb[0]) < 1e-4 * max(1.0, abs(la[0])) and abs(la[1] - lb[1]) < 1e-4 * max(1.0, abs(la[1])), \
            (it, la, lb)
    segs = [g for g in b._graphs.values() if isinstance(g, list)]
    assert len(segs) == 2 and all(sum(1 for op in ops if op[0] == 'reduce') >= 1 for ops in segs)
    assert sum(1 for op in [o for ops in segs for o in ops] if op[0] == 'reduce') == 5      # D: 1, G: 4 sections
    worst = max(float((a.store[n] - b.store[n]).abs().max()) for n in a.store.names())
    assert worst < 4e-3, worst
    torch.cuda.synchronize()
    print('RCCL_WORLD1_OK worst=%g' % worst)
    sys.stdout.flush()
    del a, b
    torch.cuda.synchronize()
    dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
