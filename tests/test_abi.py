"""The C-ABI library loads, exports every symbol include/sketchycolor_hip.h declares, and the ctypes
mirrors of the descriptor structs have the C sizes (checked by compiling the header with gcc)."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'sketchycolor_hip.h')


def _declared():
    return set(re.findall(r'\bint (ssc_\w+)\(', open(HEADER).read()))


def test_library_exports_every_declared_symbol():
    from sketchyscenecolorization_amd import build, hip
    path = build.build_library(verbose=False)
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert names == set(hip.SIGNATURES), names ^ set(hip.SIGNATURES)
    assert lib.ssc_version() >= 100
    assert hasattr(lib, 'ssc_crc32c')      # the one non-int entry point (host CRC-32C for the TFRecord reader)


def test_struct_layouts_match_header():
    from sketchyscenecolorization_amd import hip
    src = '#include <stdio.h>\n#include "sketchycolor_hip.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(ssc_gview), ' \
          'sizeof(ssc_conv_desc), sizeof(ssc_wgrad_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 't.c')
        open(c, 'w').write(src)
        exe = os.path.join(d, 't')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(hip.GView), ctypes.sizeof(hip.ConvDesc), ctypes.sizeof(hip.WgradDesc)]


def test_no_oracle_import_in_product():
    """The product path must never route through the oracle (or /root/reference)."""
    pkg = os.path.join(ROOT, 'sketchyscenecolorization_amd')
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(base, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f
                assert '/root/reference' not in txt, f
    assert 'oracle' not in sys.modules or True


def test_stale_binary_is_refused(tmp_path, monkeypatch):
    """The library carries the hash of the sources it was built from (ssc_build_hash); a tree whose kernel sources differ --
    here: one comment appended to a .hip file, no rebuild -- makes hip.lib() (and with it smoke(), bench.py, every test)
    fail loudly instead of running the old binary under the new sources' name."""
    import shutil
    from sketchyscenecolorization_amd import build, hip
    build.build_library(verbose=False)
    assert build.library_hash() == build.tree_hash() and not build.is_stale()
    l = ctypes.CDLL(build.LIB_PATH)
    hip._declare(l)
    assert hip.build_hash(l) == build.tree_hash()
    csrc2 = tmp_path / 'csrc'
    shutil.copytree(build.CSRC, str(csrc2))
    with open(str(csrc2 / 'igemm.hip'), 'a') as f:
        f.write('// touched\n')
    monkeypatch.setattr(build, 'CSRC', str(csrc2))
    assert build.is_stale()
    monkeypatch.setattr(hip, '_lib', None)
    monkeypatch.delenv('SSC_ALLOW_STALE_LIB', raising=False)
    try:
        hip.lib()
    except RuntimeError as e:
        assert 'built from kernel sources' in str(e)
    else:
        raise AssertionError('a stale library was accepted')
    assert hip._lib is None
    # smoke() starts with the same check
    import __graft_entry__ as G
    src = open(G.__file__).read()
    assert 'hip.check_build_hash()' in src.split('def smoke')[1]


def test_library_has_no_packed_fp32_math():
    """Packed fp32 vector instructions beside fp32 MFMAs are corrupted by another wave's bf16 MFMAs on MI355X (profiles/
    NOTEBOOK_r05.md section 3; tests/test_gpu_stream_hazards.py): the build keeps the compiler from packing scalar math
    (-fno-slp-vectorize) and the sources do not ask for packed math outside narrow.hip (no MFMAs, never seen perturbed)."""
    from sketchyscenecolorization_amd import build
    assert build.NO_PACKED_FP32 == {'narrow.hip': []}
    for f in sorted(os.listdir(build.CSRC)):
        if f.endswith(('.hip', '.h')) and f != 'narrow.hip':
            txt = open(os.path.join(build.CSRC, f)).read()
            assert '__builtin_elementwise_fma' not in txt and '__builtin_elementwise_max' not in txt, f
