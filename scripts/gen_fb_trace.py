"""The generator forward + backward graph alone (bench.generator_fwd_bwd), replayed 300 times: target of a kernel trace
(scripts/gen_fb_trace.sh) that shows what the single chain leaves exposed."""
import os, sys, argparse
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sketchyscenecolorization_amd.synthetic import synthetic_batch
from sketchyscenecolorization_amd.trainer import GanTrainer
tr = GanTrainer(img=192, seed=0)
bg = synthetic_batch(32, 5678, 192)
bg = tr.input_buffers('g', bg)
r = bench.generator_fwd_bwd(tr, bg, argparse.Namespace(), iters=300)
print(r['ms'], r['frac_of_fp32_mfma_peak_executed'])
