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
# -fno-slp-vectorize: the compiler must not pack scalar fp32 math into v_pk_*_f32 instructions -- beside another wave's bf16
# MFMAs they corrupt the results of a wave that issues fp32 MFMAs (csrc/igemm_util.h).  narrow.hip (no MFMA of its own) keeps
# its explicit packed FMAs.
NO_PACKED_FP32 = {'narrow.hip': []}
# MFMA results in VGPRs instead of AGPRs for the bf16-split kernels: fewer registers in total (no copies between the files), 1-2 %
VGPR_FORM = ['-mllvm', '-amdgpu-mfma-vgpr-form=1']
LAST_BUILD = None       # 'rebuilt' | 'reused' after build_library()
SOURCES = ['igemm.hip', 'igemm_bf16.hip', 'wgrad128.hip', 'wgrad128_bf16.hip', 'wgn16.hip', 'narrow.hip', 'head1.hip', 'fewchan.hip', 'fewchan7.hip', 'pw1x1.hip', 'c3x3.hip', 's2n16.hip', 'tr4tiny.hip', 'tr4n16.hip', 'elementwise.hip', 'text_lstm.hip', 'losses_optim.hip', 'mru_ops.hip']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


HASH_MARKER = b'SSC_CSRC_HASH='


def tree_hash(csrc=None, header=None):
    """sha256 (16 hex digits) over the kernel sources + the C-ABI header.  It is compiled INTO the library
    (-DSSC_CSRC_HASH, exported as ssc_build_hash) so that a binary can be tied to the sources it was built from: hip.lib()
    refuses a library whose hash is not the tree's, bench.py / scripts/pmc_summary.py stamp their output with it."""
    import hashlib
    csrc = csrc or CSRC
    header = header or os.path.join(ROOT, 'include', 'sketchycolor_hip.h')
    h = hashlib.sha256()
    for fp in [os.path.join(csrc, f) for f in sorted(os.listdir(csrc))] + [header]:
        if fp.endswith(('.hip', '.h')):
            with open(fp, 'rb') as fh:
                h.update(os.path.basename(fp).encode() + b'\0' + fh.read())
    return h.hexdigest()[:16]


def library_hash(path=None):
    """The source hash a built library carries (read from the file, no dlopen); None when it has none."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    with open(path, 'rb') as fh:
        blob = fh.read()
    i = blob.find(HASH_MARKER)
    if i < 0:
        return None
    return blob[i + len(HASH_MARKER):i + len(HASH_MARKER) + 16].decode('ascii', 'replace')


def is_stale():
    """The library is rebuilt when it is missing or was built from other sources (by content hash, not by mtime: a snapshot
    copy or a checkout changes every mtime, and a stale binary newer than its sources would pass an mtime test)."""
    return library_hash() != tree_hash()


def build_library(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the C-ABI shared library."""
    global LAST_BUILD
    if not force and not is_stale():
        LAST_BUILD = 'reused'
        return LIB_PATH
    LAST_BUILD = 'rebuilt'
    os.makedirs(LIB_DIR, exist_ok=True)
    th = tree_hash()
    import hashlib
    objs = []
    procs = []
    hdrs = b''
    for hp in sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join(ROOT, 'include', 'sketchycolor_hip.h')]:
        with open(hp, 'rb') as fh:
            hdrs += fh.read()
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(LIB_DIR, s.replace('.hip', '.o'))
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
               '-Wno-unused-value', '-Wno-unused-function'] + NO_PACKED_FP32.get(s, ['-fno-slp-vectorize']) + \
              os.environ.get('SSC_EXTRA_HIPCC_FLAGS', '').split() + \
              EXTRA_FLAGS.get(s, []) + (VGPR_FORM if s in ('igemm_bf16.hip', 'wgrad128_bf16.hip') else []) + (['-DSSC_CSRC_HASH="%s"' % th] if s == 'elementwise.hip' else []) + ['-c', src, '-o', obj]
        objs.append(obj)
        # an object is reused when its source, the shared headers and its command line are what it was compiled from
        with open(src, 'rb') as fh:
            key = hashlib.sha256(fh.read() + b'\0' + hdrs + b'\0' + ' '.join(cmd).encode()).hexdigest()
        kf = obj + '.key'
        if not force and os.path.exists(obj) and os.path.exists(kf) and open(kf).read() == key:
            continue
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd, kf, key))
    for p, cmd, kf, key in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
        with open(kf, 'w') as fh:
            fh.write(key)
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
    print(LIB_PATH)
