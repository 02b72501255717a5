"""Diagnostic only: how much of the train step does a class of launches cost?  The named C-ABI entry points are replaced by
no-ops (results are WRONG), the step is captured and replayed as usual, and the step time is compared with the full step.
An upper bound for what fusing / removing that class of launches can win.  usage: ablate_step.py [group ...]"""
import os
import sys
import time
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd import hip
from sketchyscenecolorization_amd.synthetic import synthetic_batch
from sketchyscenecolorization_amd.trainer import Pix2PixTrainer

GROUPS = {
    'none': [],
    'none2': [],
    # convs with <= 8 input channels and a full-width output (the three first layers and decoder_1's data gradient)
    'fewchan': [lambda d: d.x.C0 + d.x.C1 <= 8 and d.Nn >= 32],
    'narrow': [lambda d: d.Nn <= 4],
    'bn_bwd': ['ssc_bn_act_backward'],
    'bn_fin': ['ssc_bn_finalize', 'ssc_bn_stats'],
    'text': ['ssc_lstm_step_fwd', 'ssc_lstm_pointwise_fwd', 'ssc_lstm_pointwise_bwd', 'ssc_embedding_gather',
             'ssc_embedding_scatter_add', 'ssc_row_l2norm_fwd', 'ssc_row_l2norm_bwd', 'ssc_squash_fwd', 'ssc_squash_bwd',
             'ssc_group_rowsum', 'ssc_miu_permute_fwd', 'ssc_miu_permute_bwd', 'ssc_add_row_bcast'],
    'lstm_fwd': ['ssc_lstm_step_fwd'],
    'wgrad': ['ssc_conv_wgrad'],
    'adam': ['ssc_adam_tf'],
    'small': ['ssc_fill', 'ssc_axpy', 'ssc_sn_forward', 'ssc_sn_backward', 'ssc_act_mean_hw', 'ssc_fc_small_fwd', 'ssc_fc_small_bwd',
              'ssc_softplus_loss', 'ssc_acgan_loss', 'ssc_gen_output_grad', 'ssc_l2_reg', 'ssc_nchw_to_nhwc', 'ssc_nhwc_to_nchw'],
}


def measure(names, n=32, steps=80):
    L = hip.lib()
    convpred = None
    if names and callable(names[0]):
        convpred, names = names[0], []
    saved = {k: getattr(L, k) for k in names}
    for k in names:
        setattr(L, k, lambda *a: 0)
    run_conv = hip._run_conv
    if convpred is not None:        # skip the conv launches the predicate selects
        hip._run_conv = lambda d, bn=None, bnbwd=None: None if convpred(d) else run_conv(d, bn=bn, bnbwd=bnbwd)
    tr = Pix2PixTrainer(img=192, seed=0)
    bd, bg = synthetic_batch(n, 1, 192), synthetic_batch(n, 2, 192)
    for i in range(40):
        tr.train_iteration(bd, bg, i)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(steps):
        tr.train_iteration(bd, bg, 6 + i)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    for k, v in saved.items():
        setattr(L, k, v)
    hip._run_conv = run_conv
    del tr
    return ms


if __name__ == '__main__':
    groups = sys.argv[1:] or list(GROUPS)
    base = None
    for g in groups:
        ms = measure(GROUPS[g])
        if g == 'none':
            base = ms
        print('%-10s %7.3f ms/step%s' % (g, ms, '' if base is None or g == 'none' else '   (full step - this = %.3f ms)' % (base - ms)), flush=True)
