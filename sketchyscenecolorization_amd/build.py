"""Build libsketchycolor_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libsketchycolor_hip.so')
EXTRA_FLAGS = {'igemm.hip': ['-Xclang', '-target-feature', '-Xclang', '-load-store-opt']} if os.environ.get('SSC_NO_LSOPT') == '1' else {}     # per-source compiler flags
SOURCES = ['igemm.hip', 'wgrad128.hip', 'narrow.hip', 'head1.hip', 'fewchan.hip', 'elementwise.hip', 'text_lstm.hip', 'losses_optim.hip', 'mru_ops.hip']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'igemm_util.h'), os.path.join(ROOT, 'include', 'sketchycolor_hip.h')]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the C-ABI shared library."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(LIB_DIR, s.replace('.hip', '.o'))
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
               '-Wno-unused-value', '-Wno-unused-function'] + os.environ.get('SSC_EXTRA_HIPCC_FLAGS', '').split() + \
              EXTRA_FLAGS.get(s, []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
    print(LIB_PATH)
