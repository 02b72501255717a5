"""What the launch planner picks for the conv layers of the Pix2Pix train step (host only, no GPU needed)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchyscenecolorization_amd import hip

CFG = {0: '128x128', 1: '64x128', 2: '128x64', 3: '128x32', 4: '64x64', -1: 'narrow'}


def plan(NB, PH, PW, TH, TW, C0, C1, Nn, nphase=1, bmode=0, flags=True):
    d = hip.ConvDesc()
    d.x.C0, d.x.C1, d.x.H, d.x.W = C0, C1, PH * 2, PW * 2
    d.NB, d.PH, d.PW, d.TH, d.TW, d.nphase = NB, PH, PW, TH, TW, nphase
    d.in_stride, d.kstep, d.KH, d.KW = 2, 1, 4, 4
    d.wC0, d.wC1 = (C0 + C1, Nn) if bmode == 0 else (Nn, C0 + C1)
    d.bmode, d.k_real, d.n_off, d.Nn, d.Nstore = bmode, C0 + C1, 0, Nn, Nn
    d.OH, d.OW, d.ldc, d.out_stride = PH, PW, Nn, 1
    d.sk_flags = 1 if flags else None
    out = (C.c_int * 5)()
    hip.lib().ssc_conv_forward_plan(C.byref(d), 256 << 20, out)
    return list(out)


if __name__ == '__main__':
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    rows = [('enc2', N, 48, 48, 4, 4, 64, 0, 128, 1, 0), ('enc3', N, 24, 24, 4, 4, 128, 0, 256, 1, 0),
            ('enc4', N, 12, 12, 4, 4, 256, 0, 512, 1, 0), ('enc5', N, 6, 6, 4, 4, 512, 0, 512, 1, 0),
            ('dec5', N, 6, 6, 2, 2, 512, 64, 512, 4, 1), ('dec4', N, 12, 12, 2, 2, 512, 512, 256, 4, 1),
            ('dec3', N, 24, 24, 2, 2, 256, 256, 128, 4, 1), ('dec2', N, 48, 48, 2, 2, 128, 128, 64, 4, 1),
            ('d_l4', N, 23, 23, 4, 4, 256, 0, 512, 1, 0)]
    for name, *a in rows:
        for fl in (True, False):
            c, sk, full, s, kc = plan(*a, flags=fl)
            M = a[0] * a[1] * a[2]
            print('%-5s M=%6d N=%4d K=%5d phases %d  %-8s split-K %2d  whole tiles %5d  slices/tile %d  model %6d kcyc  %s'
                  % (name, M, a[7], a[3] * a[4] * (a[5] + a[6]), a[8], CFG[c], sk, full, s, kc, 'flags' if fl else 'no flags'))
