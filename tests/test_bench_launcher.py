"""bench.py --gpus N must start N ranks itself (the flag used to be parsed and ignored).
The CPU test drives the real launcher path (torch.distributed.run, env bootstrap, barrier, max-over-ranks timing, one
JSON line from rank 0) with a gloo stub step; the GPU test goes through the same path with 1 rank on RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    return r, lines


def test_gpus_flag_starts_that_many_ranks_gloo_stub():
    r, lines = _run(['--gpus', '2', '--steps', '3', '--warmup', '1', '--stub-cpu'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout          # exactly one JSON line, from rank 0
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['parallelism'] == 'dp2' and j['config']['ranks_observed'] == 2
    assert j['config']['global_batch'] == 64 and j['steps'] == 3 and j['warmup'] == 1
    # sum all-reduce x 1/world of rank+1 = mean(1, 2) = 1.5, a fixed point of further steps
    assert abs(j['config']['mean_value'] - 1.5) < 1e-6


def test_world_size_mismatch_is_refused():
    r, lines = _run(['--gpus', '4', '--stub-cpu'], {'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=2' in (r.stderr + r.stdout) and not lines


def test_default_is_single_rank_in_process(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    import argparse
    a = argparse.Namespace(gpus=1, launcher=False, stub_cpu=False)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    assert bench.self_launch(a, []) is None     # returns without spawning


@pytest.mark.gpu
def test_launcher_path_one_rank_rccl_matches_in_process():
    common = ['--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-kernel-events']
    r1, l1 = _run(common)
    assert r1.returncode == 0 and len(l1) == 1, r1.stderr[-2000:]
    r2, l2 = _run(['--gpus', '1', '--launcher'] + common)
    assert r2.returncode == 0 and len(l2) == 1, r2.stderr[-2000:]
    a, b = json.loads(l1[0]), json.loads(l2[0])
    assert b['n_gpus'] == 1 and b['config']['ranks_observed_by_allreduce'] == 1
    assert 'torch.distributed.run' in b['config']['launcher'] and a['config']['launcher'] == 'in-process'
    assert abs(a['value'] - b['value']) / a['value'] < 0.05, (a['value'], b['value'])


@pytest.mark.gpu
def test_two_rank_rehearsal_on_one_gpu():
    """The whole N = 2 bench path on one GPU: `bench.py --gpus 2` starts two ranks, each runs the segmented-graph train step
    with its own batch, gradients are all-reduced (gloo over device tensors: RCCL refuses two ranks on one device), timing is the
    max over ranks, rank 0 prints one line.  A rehearsal of what the driver launches on an 8-GPU node, labelled as such."""
    r, lines = _run(['--gpus', '2', '--steps', '4', '--warmup', '3', '--batch', '2', '--img', '64', '--preheat-seconds', '0.2',
                     '--no-cpu-baseline', '--prof-steps', '1'], {'SSC_BENCH_ONE_DEVICE': '1'}, timeout=900)
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['parallelism'] == 'dp2' and j['config']['ranks_observed_by_allreduce'] == 2
    assert j['config']['global_batch'] == 4 and 'REHEARSAL' in j['config']['launcher']
    assert j['value'] > 0 and j['scaling'] == 'weak'
    import math
    assert math.isfinite(j['config']['loss_g']) and math.isfinite(j['config']['loss_d'])


def test_cli_num_gpu_self_launch_on_cpu(tmp_path):
    """obj_colorization_main.py -gpu N outside a launcher re-executes itself as N ranks (the reference loops its towers in
    process, obj_colorization_main.py:189-190): the real torch.distributed.run path on CPU, with a script that uses
    dist_utils.launch_towers the way the CLI does and whose ranks report back through files."""
    script = tmp_path / 'towers.py'
    script.write_text(
        'import os, sys\n'
        'sys.path.insert(0, %r)\n'
        'from sketchyscenecolorization_amd.dist_utils import launch_towers\n'
        'rc = launch_towers(int(sys.argv[1]))\n'
        'if rc is not None:\n'
        '    print("LAUNCHER_DONE rc=%%d" %% rc); raise SystemExit(rc)\n'
        'open(os.path.join(%r, "rank%%s_of%%s" %% (os.environ["RANK"], os.environ["WORLD_SIZE"])), "w").write(" ".join(sys.argv[1:]))\n'
        % (ROOT, str(tmp_path)))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['SSC_DIST_ONE_DEVICE'] = '1'        # no device-count check: this box has no GPU
    r = subprocess.run([sys.executable, str(script), '2', '--flag', 'x y'], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and 'LAUNCHER_DONE rc=0' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert sorted(f for f in os.listdir(str(tmp_path)) if f.startswith('rank')) == ['rank0_of2', 'rank1_of2']
    assert (tmp_path / 'rank1_of2').read_text() == '2 --flag x y'      # the ranks get the command's own arguments
    # one tower, or already under a launcher: nothing to start
    from sketchyscenecolorization_amd.dist_utils import launch_towers
    assert launch_towers(1) is None
    os.environ['WORLD_SIZE'] = '2'
    try:
        assert launch_towers(2) is None
    finally:
        del os.environ['WORLD_SIZE']


def test_cli_refuses_more_towers_than_gpus():
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'SSC_DIST_ONE_DEVICE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'obj_colorization_main.py'), '--mode', 'train', '-gpu', '64'],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and '--num_gpu 64 but only' in (r.stderr + r.stdout)
