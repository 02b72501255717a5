"""Does the Residual gradient-parity statistic depend on the host thread count (i.e. on the float64 / float32 CPU
references) or on the device result?  Prints per thread count: device-gradient checksum, float64-reference checksum,
median relative L2 of device vs float64 and of CPU-fp32 vs float64."""
import os, sys, torch
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_residual as T
from oracle import residual as R
for thr in (None, 32, 8):
    if thr: torch.set_num_threads(thr)
    p, tr, b, dev = T._make_trainer(2, 64)
    r = R.build_single_graph_f64(p, **b)
    r32 = R.build_single_graph(p, **b)
    tr.d_step(dev, counter=0)
    sc = tr.store.discriminator
    l2, _ = T._grad_errors(lambda n: sc.g[n], r['grad_d'])
    c2, _ = T._grad_errors(lambda n: r32['grad_d'][n], r['grad_d'])
    dsum = sum(float(sc.g[n].double().abs().sum()) for n in r['grad_d'])
    rsum = sum(float(g.abs().sum()) for g in r['grad_d'].values())
    print('threads', torch.get_num_threads(), 'D: device checksum %.10e  f64 ref checksum %.10e  med %.3e  cpu32 med %.3e'
          % (dsum, rsum, np.median(list(l2.values())), np.median(list(c2.values()))))
    tr.store.load_dict(p)
    tr.g_step(dev, counter=0)
    sc = tr.store.generator
    l2, _ = T._grad_errors(lambda n: sc.g[n], r['grad_g'])
    c2, _ = T._grad_errors(lambda n: r32['grad_g'][n], r['grad_g'])
    dsum = sum(float(sc.g[n].double().abs().sum()) for n in r['grad_g'])
    rsum = sum(float(g.abs().sum()) for g in r['grad_g'].values())
    print('threads', torch.get_num_threads(), 'G: device checksum %.10e  f64 ref checksum %.10e  med %.3e  cpu32 med %.3e'
          % (dsum, rsum, np.median(list(l2.values())), np.median(list(c2.values()))))
