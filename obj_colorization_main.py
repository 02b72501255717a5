#!/usr/bin/env python
"""Command line of the Foreground_Instance_Colorization module on the MI355X-native HIP path.

Drop-in for the reference CLI (obj_colorization_main.py:159-257): the same flags, short aliases, defaults and
choices, the same run directory ``outputs/<UTC Y-m-d-H-M-S>/{log,snapshot,validation_results,test_results,
inference_results}``, ``log/param_<first iteration>.json`` and the restart-after-NaN loop.

    python obj_colorization_main.py --mode train --block_type Pix2Pix --batch_size 32 --max_iter 1000
    python obj_colorization_main.py --mode train -bt Pix2Pix -gpu 8      # starts itself as 8 ranks, one per GPU
    python -m torch.distributed.run --nproc-per-node 8 obj_colorization_main.py --mode train -bt Pix2Pix -gpu 8   # same
    python obj_colorization_main.py --mode inference -rf <timestamp> --infer_name car.png \
        --instruction 'the car is yellow with blue window'
"""
import argparse
import json
import os

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')     # kernel arguments in device memory (measured: 1570 vs 1540 images/s with 0)
import time

from sketchyscenecolorization_amd.obj_lib import main_procedure
from sketchyscenecolorization_amd.obj_lib.config import Config

# (long flag, short flag, type, default, choices, Config / d_params key, help)
FLAGS = [
    ('mode', 'md', str, 'train', ['train', 'val', 'test', 'inference'], 'dataset_type', 'what to run'),
    ('resume_from', 'rf', str, '', None, 'resume_from', 'timestamp of an earlier run under outputs/ to continue or evaluate'),
    ('batch_size', 'bs', int, 2, None, 'batch_size', 'samples per GPU and step'),
    ('max_iter', 'mi', int, 100000, None, 'max_iter_step', 'last training iteration'),
    ('optimizer', 'opt', str, 'Adam', ['RMSprop', 'Adam', 'AdaDelta', 'AdaGrad'], 'optimizer', 'optimizer family'),
    ('lr_G', 'lrg', float, 2e-4, None, 'lr_G', 'generator step size'),
    ('lr_D', 'lrd', float, 1e-4, None, 'lr_D', 'discriminator step size'),
    ('small_img', 'si', int, 0, [0, 1], 'small_img', '1 = 64x64 instances instead of 192x192'),
    ('lstm_hybrid', 'lh', int, 1, [0, 1], 'LSTM_hybrid', '1 = the caption steers the colours'),
    ('distance_map', 'dm', int, 0, [0, 1], 'distance_map', '1 = sketches as distance maps'),
    ('block_type', 'bt', str, 'MRU', ['MRU', 'Pix2Pix', 'Residual'], 'block_type', 'network family'),
    ('vocab_size', 'vs', int, 58, None, 'vocab_size', 'caption vocabulary size'),
    ('disc_iterations', 'di', int, 1, None, 'disc_iterations', 'discriminator updates per generator update'),
    ('ld', 'ld', int, 10, None, 'ld', 'gradient-penalty weight (unused by the live loss)'),
    ('num_gpu', 'gpu', int, 1, None, 'num_gpu', 'towers (one process per GPU under torch.distributed.run)'),
    ('extra_info', 'ei', str, '', None, 'extra_info', 'free text stored with the run parameters'),
    ('summary_write_freq', 'swf', int, 100, None, 'summary_write_freq', 'iterations between scalar summaries'),
    ('save_model_freq', 'smf', int, 10000, None, 'save_model_freq', 'iterations between snapshots'),
    ('count_left_time_freq', 'clt', int, 100, None, 'count_left_time_freq', 'iterations between ETA prints'),
    ('count_inception_score_freq', 'cis', int, -1, None, 'count_inception_score_freq', '-1 = never'),
    ('infer_name', 'in', str, '', None, 'infer_name', 'sketch file under examples/ (inference mode)'),
    ('instruction', 'ins', str, '', None, 'instruction', 'caption for that sketch (inference mode)'),
]
RESULT_DIRS = {'val': 'validation_results', 'test': 'test_results', 'inference': 'inference_results'}


def build_parser():
    parser = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    for name, short, typ, default, choices, _key, text in FLAGS:
        parser.add_argument('--' + name, '-' + short, type=typ, default=default, choices=choices, help=text)
    return parser


def run_dirs(stamp):
    root = os.path.join('outputs', stamp)
    return os.path.join(root, 'log'), os.path.join(root, 'snapshot'), root


def is_stamp(text):
    return bool(text) and len(text.split('-')) == 6


def start_or_resume_training(params):
    """One call of main_procedure.train: fresh run (new UTC stamp) or continuation of ``resume_from``."""
    stamp = params['resume_from']
    fresh = stamp is None or stamp == ''
    rank, multi = int(os.environ.get('RANK', 0)), params['num_gpu'] > 1
    if multi:       # one process per GPU: rank 0 decides the run directory and the first iteration for everyone
        from sketchyscenecolorization_amd.dist_utils import init_distributed
        dist = init_distributed()
    if fresh:
        stamp = time.strftime('%Y-%m-%d-%H-%M-%S', time.gmtime())
        first_iter = 0
    elif not is_stamp(stamp):
        print('Invalid resume folder')
        return None, stamp
    if multi and fresh:     # ranks that straddle a second boundary would otherwise write to different directories
        box = [stamp]
        dist.broadcast_object_list(box, src=0)
        stamp = box[0]
    log_dir, ckpt_dir, _ = run_dirs(stamp)
    if fresh:
        for d in (log_dir, ckpt_dir):
            os.makedirs(d, exist_ok=True)
    else:
        box = [None]
        if rank == 0:
            latest = main_procedure.latest_checkpoint(ckpt_dir)
            box = [None if latest is None else int(os.path.basename(latest).split('-')[1]) + 1]
        if multi:
            dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise RuntimeError('no snapshot under %s' % ckpt_dir)
        first_iter = box[0]
    params.update(log_dir=log_dir, ckpt_dir=ckpt_dir, iter_from=first_iter)
    if int(os.environ.get('RANK', 0)) == 0:
        os.makedirs(log_dir, exist_ok=True)     # a released model dropped into outputs/<stamp>/snapshot has no log/ yet
        with open(os.path.join(log_dir, 'param_%d.json' % first_iter), 'w') as fp:
            json.dump(params, fp, indent=4)
    Config.set_from_dict(params)
    print(('Launching new train: %s' if fresh else 'Launching training from checkpoint: %s') % stamp)
    return main_procedure.train(**params), stamp


def evaluate(mode, params):
    stamp = params['resume_from']
    if not is_stamp(stamp):
        print('Invalid resume folder')
        return
    log_dir, ckpt_dir, root = run_dirs(stamp)
    params.update(log_dir=log_dir, ckpt_dir=ckpt_dir, results_dir=os.path.join(root, RESULT_DIRS[mode]))
    Config.set_from_dict(params)
    print('Launching %s from checkpoint: %s' % ({'val': 'validation', 'test': 'testing', 'inference': 'inference'}[mode],
                                                 stamp))
    if mode == 'val':
        main_procedure.validation(**params)
    elif mode == 'test':
        main_procedure.test()
    else:
        main_procedure.inference(params['infer_name'], params['instruction'])


def main(argv=None):
    args = build_parser().parse_args(argv)
    params = {key: getattr(args, name) for name, _s, _t, _d, _c, key, _h in FLAGS}
    if args.mode != 'train':
        if args.mode == 'inference':
            assert args.infer_name != '' and args.instruction != '', '--infer_name and --instruction are required'
        return evaluate(args.mode, params)
    # -gpu N without a launcher: the towers are processes here, so the command starts itself once per GPU
    from sketchyscenecolorization_amd.dist_utils import launch_towers
    # a programmatic caller's explicit argv is what the ranks must see, under the host script (which may wrap this module)
    import sys
    rc = launch_towers(args.num_gpu, None if argv is None else [sys.argv[0]] + list(argv))
    if rc is not None:
        if rc != 0:
            raise SystemExit(rc)
        return
    status, stamp = start_or_resume_training(params)
    while status == -1:         # a NaN loss ends train() with -1: continue from the last snapshot
        print('Training ended with status -1. Restarting..')
        params['resume_from'] = stamp
        status, stamp = start_or_resume_training(params)


if __name__ == '__main__':
    main()
