"""Throughput of the multi-GPU step protocol on ONE GPU: 1-rank RCCL group, steps captured as graph segments with the
all-reduces issued eagerly between them (reducer.world faked to 2 so that the collectives are really called).  The gap
to the plain single-GPU number is what the protocol itself costs before any xGMI time."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd.synthetic import synthetic_batch      # noqa: E402
from sketchyscenecolorization_amd.trainer import GanTrainer             # noqa: E402

with socket.socket() as sock:
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
NODIST = os.environ.get('SSC_PROBE_NODIST') == '1'      # no process group at all (the collectives become no-ops): for rocprofv3
if not NODIST:
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
for seg in ((True,) if os.environ.get('SSC_PROBE_SEG_ONLY') == '1' else (False, True)):
    tr = GanTrainer(img=192, seed=0, use_graphs=True, segment_graphs=seg, process_group=dist.group.WORLD if (seg and not NODIST) else None)
    if seg:
        tr.reducer.world = 2
        tr.reducer.stream = torch.cuda.Stream()
        tr.world = int(os.environ.get("SSC_PROBE_WORLD", "2"))     # 2: the many-tower code paths (sectioned discriminator all-reduce)
    if seg and (NODIST or os.environ.get('SSC_PROBE_NOREDUCE') == '1'):      # the same segments, no collective call in between
        tr.reducer.reduce_async = lambda *a: None
        tr.reducer.wait = lambda: None
    bd, bg = synthetic_batch(32, 1, 192), synthetic_batch(32, 2, 192)
    bd, bg = tr.input_buffers('d', bd), tr.input_buffers('g', bg)
    for i in range(60):
        tr.train_iteration(bd, bg, counter=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for i in range(n):
        tr.train_iteration(bd, bg, counter=60 + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('segmented graphs + eager all-reduces' if seg else 'one graph per step', '%.1f images/s  %.2f ms/step' % (32 * n / dt, dt / n * 1e3))
    sys.stdout.flush()
    del tr
torch.cuda.synchronize()
os._exit(0)
