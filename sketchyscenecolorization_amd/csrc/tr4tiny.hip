// tr4tiny.hip -- the k = 4 stride-2 transposed convs of the Background generator's region branch (region_br_k: 3 mask channels ->
// 3, + norm + relu; bg_colorization_main.py:392-397, 411-416) -- at most 4 channels in, at most 4 out.
//
// 12 multiply-adds per output value: on the MFMA tile kernel this launch is a 32-wide K chunk holding 3 real channels times a
// 32-column tile holding 3 real columns -- 176 us at 768^2 for 9 MB read and 28 MB written.  Here a thread owns one lattice pixel:
// its 3 x 3 neighbourhood is nine 16-byte loads (folded norm + activation applied, zeros outside the image), the 16 x 4 x 4
// filter values are wave-uniform (scalar loads), the four sub-pixel phases leave as four 16-byte stores, and the batch statistics
// of the output (its norm follows) are per-thread sums folded once per workgroup: one row of partials per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

__global__ __launch_bounds__(256) void tr4_tiny_kernel(const ssc_conv_desc d, long npix, float* __restrict__ stat) {
    __shared__ float red[8][256];
    const int tid = threadIdx.x;
    const long pix = (long)blockIdx.x * 256 + tid;
    const int H = d.x.H, W = d.x.W;
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    if (pix < npix) {
        const int px = (int)(pix % W);
        const long r = pix / W;
        const int py = (int)(r % H), n = (int)(r / H);
        float4 ta = make_float4(1.f, 1.f, 1.f, 1.f), tb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.x.ab0 != nullptr) {
            ta = *reinterpret_cast<const float4*>(d.x.ab0);
            tb = *reinterpret_cast<const float4*>(d.x.ab0 + 4);
        }
        const float slope = d.x.act == SSC_ACT_RELU ? 0.f : (d.x.act == SSC_ACT_LRELU ? 0.2f : 1.f);
        float xin[3][3][4];
#pragma unroll
        for (int oy = 0; oy < 3; ++oy)
#pragma unroll
            for (int ox = 0; ox < 3; ++ox) {
                const int iy = py - 1 + oy, ix = px - 1 + ox;
                const bool ok = ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
                const float4 v = *reinterpret_cast<const float4*>(d.x.s0 + (ok ? (((long)n * H + iy) * W + ix) * 4 : 0));
                float t[4] = {fmaf(ta.x, v.x, tb.x), fmaf(ta.y, v.y, tb.y), fmaf(ta.z, v.z, tb.z), fmaf(ta.w, v.w, tb.w)};
#pragma unroll
                for (int c = 0; c < 4; ++c) xin[oy][ox][c] = (ok && c < d.k_real) ? fmaxf(t[c], slope * t[c]) : 0.f;
            }
        // phase (ry, rx), tap (ty, tx): input offset (ry + ty, rx + tx), filter tap (3 - ry - 2 ty, 3 - rx - 2 tx); f[tap][n][c]
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int ry = ph >> 1, rx = ph & 1;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ty = t >> 1, tx = t & 1;
                const int tap = (3 - ry - 2 * ty) * 4 + (3 - rx - 2 * tx);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < d.Nn) {
                        const float* wp = d.w + ((long)tap * d.wC0 + d.n_off + j) * d.wC1;      // wave-uniform: scalar loads
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c < d.k_real) o[j] = fmaf(xin[ry + ty][rx + tx][c], wp[c], o[j]);
                    }
                }
            }
            float* op = d.out + (((long)n * d.OH + 2 * py + ry) * d.OW + 2 * px + rx) * d.ldc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < d.Nstore) {
                    float v = 0.f;          // columns in [Nn, Nstore) are channel padding: written as 0
                    if (j < d.Nn) {
                        v = o[j];
                        if (d.epi == 1) v = tanhf(v);
                        else if (d.epi == 2) v = fmaxf(v, 0.2f * v);
                    }
                    op[j] = v;
                    ssum[j] += v;
                    ssq[j] += v * v;
                }
            }
        }
    }
    if (stat != nullptr) {      // one row [sum | sum of squares] per workgroup, the threads folded in a fixed tree
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[j][tid] = ssum[j]; red[4 + j][tid] = ssq[j]; }
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
#pragma unroll
                for (int q = 0; q < 8; ++q) red[q][tid] += red[q][tid + s];
            }
            __syncthreads();
        }
        if (tid < d.Nstore) {
            float* sp = stat + (long)blockIdx.x * 2 * d.Nstore;
            sp[tid] = red[tid][0];
            sp[d.Nstore + tid] = red[4 + tid][0];
        }
    }
}

static bool t4t_on() {
    static int on = -1;         // SSC_TR4_TINY=0: the tile kernel (A/B)
    if (on < 0) {
        const char* e = ssc_dev_getenv("SSC_TR4_TINY");
        on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

extern "C" int ssc_conv_tr4_tiny_supported(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    if (!t4t_on()) return 0;
    if (d.nphase != 4 || d.TH != 2 || d.TW != 2 || d.KH != 4 || d.KW != 4 || d.bmode != 1 || d.out_stride != 2 || d.in_stride != 1 ||
        d.ky0 != 0 || d.kx0 != 0 || d.kstep != -2 || d.ioff_y != 0 || d.ioff_x != 0 || d.ooff_y != 0 || d.ooff_x != 0)
        return 0;
    if (d.x.C1 != 0 || d.x.C0 != 4 || d.k_real < 1 || d.k_real > 4 || d.wC1 < d.k_real) return 0;
    if (d.Nn < 1 || d.Nn > 4 || d.Nstore > 4 || d.Nstore < d.Nn || d.Nstore > d.ldc || d.n_off + d.Nn > d.wC0 || d.accumulate ||
        d.bias != nullptr || d.epi > 2)
        return 0;
    if (d.x.H != d.PH || d.x.W != d.PW || d.OH != 2 * d.PH || d.OW != 2 * d.PW) return 0;
    if ((reinterpret_cast<uintptr_t>(d.x.s0) & 15) != 0 || (d.x.ab0 != nullptr && (reinterpret_cast<uintptr_t>(d.x.ab0) & 15) != 0))
        return 0;
    if (d.x.act != SSC_ACT_NONE && d.x.act != SSC_ACT_RELU && d.x.act != SSC_ACT_LRELU) return 0;
    if (d.sb_x != nullptr || d.sb2_x != nullptr || d.stat_mode != 0) return 0;
    const long npix = (long)d.NB * d.PH * d.PW;
    if (npix < 1 || (npix + 255) / 256 >= 0x7fffffffL) return 0;
    return 1;
}

// workgroups (= rows of partial sums)
int ssc_conv_tr4_tiny_blocks(const ssc_conv_desc* dp) {
    return (int)(((long)dp->NB * dp->PH * dp->PW + 255) / 256);
}

int ssc_conv_tr4_tiny_forward(const ssc_conv_desc* dp, float* stat, void* stream) {
    if (!ssc_conv_tr4_tiny_supported(dp)) return -1;
    const ssc_conv_desc& d = *dp;
    const long npix = (long)d.NB * d.PH * d.PW;
    hipLaunchKernelGGL(tr4_tiny_kernel, dim3((unsigned)ssc_conv_tr4_tiny_blocks(dp)), dim3(256), 0, (hipStream_t)stream, d, npix, stat);
    return (int)hipGetLastError();
}
