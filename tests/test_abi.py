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
