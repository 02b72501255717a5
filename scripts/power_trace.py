"""Socket power and shader clock (sysfs hwmon, polled every ~10 ms by a thread) while the replayed train iteration runs: is the step
at the power limit, and does a change in kernel mix move the clock?  usage: power_trace.py [seconds]"""
import glob
import os
import sys
import threading
import time
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd.synthetic import synthetic_batch
from sketchyscenecolorization_amd.trainer import Pix2PixTrainer

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
hw = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'))
files = {}
for h in hw:
    for name in ('power1_average', 'power1_input', 'freq1_input', 'temp1_input', 'power1_cap'):
        p = os.path.join(h, name)
        if os.path.exists(p) and name not in files:
            files[name] = p
print('hwmon files:', files)
samples = []
stop = False


def poll():
    while not stop:
        row = {}
        for k, p in files.items():
            try:
                row[k] = float(open(p).read().strip())
            except Exception:
                pass
        samples.append((time.time(), row))
        time.sleep(0.01)


tr = Pix2PixTrainer(img=192, seed=0, use_graphs=True)
bd, bg = synthetic_batch(32, 1, 192), synthetic_batch(32, 2, 192)
bd, bg = tr.input_buffers('d', bd), tr.input_buffers('g', bg)
for i in range(10):
    tr.train_iteration(bd, bg, i, next_batch_d=bd)
torch.cuda.synchronize()
th = threading.Thread(target=poll)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(20):
        tr.train_iteration(bd, bg, 10 + n, next_batch_d=bd)
        n += 1
    torch.cuda.synchronize()
dt = time.time() - t0
stop = True
th.join()
keep = [r for t, r in samples if t - t0 > 1.0]


def mean(k):
    v = [r[k] for r in keep if k in r]
    return sum(v) / len(v) if v else float('nan')


pw = mean('power1_average') if 'power1_average' in files else mean('power1_input')
print('%.1f img/s  %.3f ms/step  power %.0f W (cap %.0f W)  sclk %.0f MHz  temp %.1f C  samples %d' % (
    32 * n / dt, dt / n * 1e3, pw / 1e6, mean('power1_cap') / 1e6, mean('freq1_input') / 1e6, mean('temp1_input') / 1e3, len(keep)))
