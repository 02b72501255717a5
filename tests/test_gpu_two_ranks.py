"""Two real ranks on one GPU (gloo process group over device tensors): the data-parallel train step of BASELINE config 4 as far
as a single-GPU box allows -- tests/two_rank_gloo_gpu_check.py states what is compared."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_rank_train_step_matches_averaged_tower_gradients():
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, 'two_rank_gloo_gpu_check.py')], capture_output=True, text=True,
                       timeout=900, cwd=os.path.dirname(here))
    assert 'TWO_RANK_OK' in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])


def test_two_rank_cli_nan_restart(tmp_path):
    """The CLI with -gpu 2 as two real ranks (one GPU, gloo): one run directory for both ranks (rank 0's stamp), snapshots and
    scalars written by rank 0 only, and a NaN that only rank 1 sees locally restarts BOTH ranks from the last snapshot."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['SSC_DIST_ONE_DEVICE'] = '1'
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), os.path.join(here, 'two_rank_cli_check.py')],
                       capture_output=True, text=True, timeout=900, cwd=str(tmp_path), env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert 'RANK0_DONE fired=False' in out and 'RANK1_DONE fired=True' in out, out[-4000:]
    assert out.count('NaN occurred during training G') == 2, out[-4000:]           # both ranks saw the tower-mean NaN
    assert out.count('Training ended with status -1. Restarting..') == 2
    runs = sorted(os.listdir(os.path.join(tmp_path, 'outputs')))
    assert len(runs) == 1, runs                                                     # rank 0's stamp on every rank
    run = os.path.join(tmp_path, 'outputs', runs[0])
    assert os.path.exists(os.path.join(run, 'log', 'param_0.json')) and os.path.exists(os.path.join(run, 'log', 'param_2.json'))
    assert os.path.exists(os.path.join(run, 'snapshot', 'model_1.ckpt-1')) and os.path.exists(os.path.join(run, 'snapshot', 'model_3.ckpt-3'))
    steps = [json.loads(l)['step'] for l in open(os.path.join(run, 'log', 'scalars.jsonl'))]
    assert steps == [0, 1, 2, 3, 4], steps        # iteration 2 failed before its summary; the restart repeats it


def test_cli_self_launch_two_ranks(tmp_path):
    """`-gpu 2` WITHOUT torch.distributed.run: the command starts its two ranks itself (dist_utils.launch_towers; the reference
    loops its towers inside one process, obj_colorization_main.py:189-190, graph_single.py:146-166).  Same run as
    test_two_rank_cli_nan_restart -- one run directory, snapshots by rank 0, a NaN on rank 1 restarts both -- started as a plain
    `python <script>`."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['SSC_DIST_ONE_DEVICE'] = '1'
    r = subprocess.run([sys.executable, os.path.join(here, 'two_rank_cli_check.py')], capture_output=True, text=True,
                       timeout=900, cwd=str(tmp_path), env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert 'num_gpu=2: starting 2 ranks' in out, out[-4000:]
    assert 'RANK0_DONE fired=False' in out and 'RANK1_DONE fired=True' in out and 'RANK-1_DONE' in out, out[-4000:]
    assert out.count('Training ended with status -1. Restarting..') == 2
    runs = sorted(os.listdir(os.path.join(tmp_path, 'outputs')))
    assert len(runs) == 1, runs
    run = os.path.join(tmp_path, 'outputs', runs[0])
    assert os.path.exists(os.path.join(run, 'snapshot', 'model_3.ckpt-3'))
    steps = [json.loads(l)['step'] for l in open(os.path.join(run, 'log', 'scalars.jsonl'))]
    assert steps == [0, 1, 2, 3, 4], steps
