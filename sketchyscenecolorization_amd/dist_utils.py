"""Data-parallel plumbing: one process per GPU, each process = one "tower" of the reference
(graph_single.py:128-173).  ``average_gradients`` (:33-68: per-variable mean over towers) becomes a
sum all-reduce of flat gradient sections (RCCL over xGMI on GPUs, gloo in CPU tests) issued on a side
stream as soon as the backward pass has finished a section; the 1/world factor is applied by the
consumer (the Adam kernel's gradient scale)."""
import os

import torch


def init_distributed():
    """Process group of a multi-GPU run (one process per GPU under torch.distributed.run): backend "nccl" = RCCL, the
    process's GPU = LOCAL_RANK.  SSC_DIST_ONE_DEVICE=1 (tests on a single-GPU box): every rank on cuda:0 over gloo, which
    reduces device tensors through host memory -- RCCL refuses two ranks on one device."""
    import torch.distributed as dist
    one = os.environ.get('SSC_DIST_ONE_DEVICE') == '1'
    torch.cuda.set_device(0 if one else int(os.environ.get('LOCAL_RANK', 0)))
    if not dist.is_initialized():
        dist.init_process_group('gloo' if one else 'nccl')
    return dist


def launch_towers(num_gpu, argv=None):
    """``--num_gpu N`` without a launcher: the reference loops its N towers inside one process (obj_colorization_main.py:
    189-190, graph_single.py:146-166); here a tower is a process, so the command re-executes itself as N ranks under
    torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and returns their exit code.  Returns None when there is
    nothing to do: one tower, or already inside a launcher (WORLD_SIZE set).  ``argv``: [script, args...], default sys.argv."""
    import subprocess
    import sys
    if num_gpu <= 1 or 'WORLD_SIZE' in os.environ:
        return None
    argv = list(sys.argv if argv is None else argv)
    if os.environ.get('SSC_DIST_ONE_DEVICE') != '1':
        n_dev = torch.cuda.device_count()
        if n_dev < num_gpu:
            raise SystemExit('--num_gpu %d but only %d GPU(s) visible' % (num_gpu, n_dev))
    # --standalone: torch.distributed.run picks (and holds) a free rendezvous port itself; a port found here by bind / close
    # could be taken before the launcher binds it
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(num_gpu), os.path.abspath(argv[0])] + argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC (RCCL across processes)
    env.setdefault('OMP_NUM_THREADS', '8')
    print('num_gpu=%d: starting %d ranks (one per GPU): %s' % (num_gpu, num_gpu, ' '.join(cmd)), flush=True)
    return subprocess.call(cmd, env=env)


def tower_slice(global_n, batch_size, rank, world, batch_portion=None):
    """Sample range [lo, hi) of tower ``rank`` (input_pipeline.split_inputs semantics)."""
    portion = [1] * world if batch_portion is None else list(batch_portion)
    lo = sum(int(batch_size * p) for p in portion[:rank])
    hi = lo + int(batch_size * portion[rank])
    assert hi <= global_n
    return lo, hi


class GradReducer(object):
    def __init__(self, process_group=None):
        self.pg = process_group
        self.world = 1
        self.stream = None
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
            if self.world > 1 and torch.cuda.is_available():
                self.stream = torch.cuda.Stream()

    def reduce_async(self, flat, lo, hi):
        """Sum flat[lo:hi] over all towers; asynchronous w.r.t. the compute stream on GPUs."""
        if self.world == 1:
            return
        import torch.distributed as dist
        if self.stream is None:         # CPU / gloo: synchronous
            dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)

    def wait(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    @property
    def grad_scale(self):
        return 1.0 / self.world
