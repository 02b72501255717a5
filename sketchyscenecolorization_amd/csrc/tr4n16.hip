// tr4n16.hip -- the k = 4 stride-2 transposed conv from 256 channels (concat of two 128-channel tensors) to 16: block_1 of the last
// decoder bottleneck (residual_util.py:122-126 with C/4 = 16; the Background generator's copy at 768^2) on the 16-column MFMA.
//
// Per sub-pixel phase this is M lattice pixels by 16 columns with K = 2x2 taps x 256 channels = 1024; on the tile kernel (32 columns)
// half of the matrix work multiplies padding: 203 us at batch 32 / 192^2, 403 us at 768^2 (47 TFLOP/s on the real FLOPs).  The
// construction of s2n16.hip, one PHASE per workgroup:
//   * v_mfma_f32_16x16x4_f32: 16 lattice pixels x 16 channels x 4 k per instruction, no padded columns;
//   * a workgroup serves one phase (blockIdx.y) for all the tiles it walks; its four wavefronts split K by TAP (256 channels = 64 MFMA
//     steps each), so a lane's share of the filter is 64 registers loaded once; the four partial sums meet in LDS;
//   * a tile is 2 lattice rows x 16 pixels; the (3 x 17)-pixel x 256-channel patch the phase needs is staged once in LDS with each
//     source's folded norm + activation applied on the way (zeros outside the image); an A operand is one ds_read_b32 with an
//     immediate offset; pixel stride 260 floats -> the 16 pixels x 4 k of an operand hit 64 distinct banks;
//   * the next tile's patch is in flight in registers during the MFMAs;
//   * the batch statistics of the output are per-thread sums over the tiles a workgroup walks, one row of partials per workgroup
//     (4 x walkers rows: ssc_conv_forward_bn folds them).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define T16_TR 2
#define T16_TC 16
#define T16_PR (T16_TR + 1)
#define T16_PC (T16_TC + 1)
#define T16_C 256
#define T16_PST 260
#define T16_PSZ (T16_PR * T16_PC * T16_PST)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void tr4n16_kernel(const ssc_conv_desc d, int tiles, int tiles_x,
                                                                                             int tiles_y, float* __restrict__ stat) {
    __shared__ __attribute__((aligned(16))) float patch[T16_PSZ];           // 53.0 KB
    __shared__ __attribute__((aligned(16))) float red[4 * T16_TR * 256];    // [wave][row][lane][4]: 8 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int H = d.x.H, W = d.x.W;
    const bool colv = l15 < d.Nn;
    const int ph = blockIdx.y, ry = ph >> 1, rx = ph & 1;       // this workgroup's sub-pixel phase
    const int ty = wave >> 1, tx = wave & 1;                    // this wave's tap of the phase

    // ---- filter: bf[s] = f[tap][n_off + l15][4 s + kq]   (bmode 1: f[ky][kx][n][c], c contiguous) ----
    float bf[64];
    {
        const int tap = (3 - ry - 2 * ty) * 4 + (3 - rx - 2 * tx);
        const float* wp = d.w + ((long)tap * d.wC0 + d.n_off + (colv ? l15 : 0)) * d.wC1 + kq;
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const bool v = colv & (4 * s + kq < d.k_real);
            const float wv = wp[v ? 4 * s : 0];
            bf[s] = v ? wv : 0.f;
        }
    }

    // ---- patch staging: thread -> (pixel tid / 64 + 4 q, 16-byte chunk tid % 64): its chunk, source and constants are fixed ----
    const int c4 = (tid & 63) * 4;
    const bool first = c4 < d.x.C0;
    const float* const src = first ? d.x.s0 : d.x.s1;
    const int cs = first ? d.x.C0 : d.x.C1;
    const int cc = first ? c4 : c4 - d.x.C0;
    const int act = (!first && d.x.act1 >= 0) ? d.x.act1 : d.x.act;
    const float slope = act == SSC_ACT_RELU ? 0.f : (act == SSC_ACT_LRELU ? 0.2f : 1.f);
    float4 ta = make_float4(1.f, 1.f, 1.f, 1.f), tb = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const float* abp = first ? d.x.ab0 : d.x.ab1;
        if (abp != nullptr) {
            ta = *reinterpret_cast<const float4*>(abp + cc);
            tb = *reinterpret_cast<const float4*>(abp + cs + cc);
        }
    }
    constexpr int NQ = (T16_PR * T16_PC + 3) / 4;       // 13
    float4 rv[NQ];
    auto load_patch = [&](int tile) {
        const int txi = tile % tiles_x;
        const int r = tile / tiles_x;
        const int tyi = r % tiles_y, n = r / tiles_y;
        const int iy0 = T16_TR * tyi - 1 + ry, ix0 = T16_TC * txi - 1 + rx;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pos = (tid >> 6) + 4 * q;
            const int pr = pos / T16_PC, pc = pos - pr * T16_PC;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = (pos < T16_PR * T16_PC) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            const float4 v = *reinterpret_cast<const float4*>(src + (ok ? (((long)n * H + iy) * W + ix) * cs : 0) + cc);
            float4 t;
            t.x = fmaf(ta.x, v.x, tb.x); t.y = fmaf(ta.y, v.y, tb.y); t.z = fmaf(ta.z, v.z, tb.z); t.w = fmaf(ta.w, v.w, tb.w);
            t.x = fmaxf(t.x, slope * t.x); t.y = fmaxf(t.y, slope * t.y); t.z = fmaxf(t.z, slope * t.z); t.w = fmaxf(t.w, slope * t.w);
            rv[q] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pos = (tid >> 6) + 4 * q;
            if (pos < T16_PR * T16_PC) *reinterpret_cast<float4*>(patch + pos * T16_PST + c4) = rv[q];
        }
    };

    // the thread that finishes output (lattice row g, pixel tid >> 4, column tid & 15) -- for both rows g of a tile
    const int ocol = tid & 15, opix = tid >> 4;
    float ssum = 0.f, ssq = 0.f;

    const int G = gridDim.x;
    int tile = blockIdx.x;
    if (tile < tiles) {
        load_patch(tile);
        store_patch();
    }
    __syncthreads();
    // A operand of lane (pixel l15, kq) for lattice row g, MFMA step s: patch[g + ty][l15 + tx][4 s + kq]
    const float* const A0 = patch + (ty * T16_PC + tx + l15) * T16_PST + kq;
    for (; tile < tiles; tile += G) {
        const int next = tile + G;
        if (next < tiles) load_patch(next);          // in flight across the MFMAs below
        f32x4 acc[T16_TR];
#pragma unroll
        for (int g = 0; g < T16_TR; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 64; ++s) {
#pragma unroll
            for (int g = 0; g < T16_TR; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[g * T16_PC * T16_PST + 4 * s], bf[s], acc[g], 0, 0, 0);
        }
        // partial sums of this wave's tap: acc[g][r] is pixel 4 * kq + r, column l15
#pragma unroll
        for (int g = 0; g < T16_TR; ++g) *reinterpret_cast<f32x4*>(red + ((wave * T16_TR + g) * 64 + lane) * 4) = acc[g];
        __syncthreads();            // every wave is done with the patch; the partial sums are visible
        if (next < tiles) store_patch();
        {
            const int txi = tile % tiles_x;
            const int rr = tile / tiles_x;
            const int tyi = rr % tiles_y, n = rr / tiles_y;
            const int px = T16_TC * txi + opix;
            const float* rp = red + ((opix >> 2) * 16 + ocol) * 4 + (opix & 3);
#pragma unroll
            for (int g = 0; g < T16_TR; ++g) {
                const int py = T16_TR * tyi + g;
                const float v = ((rp[(0 * T16_TR + g) * 256] + rp[(1 * T16_TR + g) * 256]) + rp[(2 * T16_TR + g) * 256]) +
                                rp[(3 * T16_TR + g) * 256];
                if ((ocol < d.Nn) & (py < d.PH) & (px < d.PW)) {
                    d.out[(((long)n * d.OH + 2 * py + ry) * d.OW + 2 * px + rx) * d.ldc + ocol] = v;
                    ssum += v;
                    ssq += v * v;
                }
            }
        }
        __syncthreads();            // the next patch is in place; `red` may be overwritten
    }
    if (stat != nullptr) {          // one row [sum | sum of squares] per workgroup: the 16 threads of a column folded in order
        __syncthreads();
        red[tid] = ssum;
        red[256 + tid] = ssq;
        __syncthreads();
        if (tid < 16 && tid < d.Nn) {
            float a = 0.f, b = 0.f;
            for (int p = 0; p < 16; ++p) {
                a += red[p * 16 + tid];
                b += red[256 + p * 16 + tid];
            }
            float* sp = stat + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * d.Nstore;
            sp[tid] = a;
            sp[d.Nstore + tid] = b;
        }
    }
}

static bool t16_on() {
    static int on = -1;         // SSC_TR4N16=0: the tile kernel (A/B)
    if (on < 0) {
        const char* e = ssc_dev_getenv("SSC_TR4N16");
        on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

extern "C" int ssc_conv_tr4n16_supported(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    if (!t16_on()) return 0;
    if (d.nphase != 4 || d.TH != 2 || d.TW != 2 || d.KH != 4 || d.KW != 4 || d.bmode != 1 || d.out_stride != 2 || d.in_stride != 1 ||
        d.ky0 != 0 || d.kx0 != 0 || d.kstep != -2 || d.ioff_y != 0 || d.ioff_x != 0 || d.ooff_y != 0 || d.ooff_x != 0)
        return 0;
    if (d.x.C0 + d.x.C1 != T16_C || (d.x.C0 & 3) != 0 || (d.x.C1 & 3) != 0 || d.k_real > T16_C || d.k_real < 4 || d.wC1 < d.k_real) return 0;
    if (d.Nn < 4 || d.Nn > 16 || d.Nn != d.Nstore || d.Nstore > d.ldc || d.n_off + d.Nn > d.wC0 || d.accumulate || d.bias != nullptr ||
        d.epi != 0)
        return 0;
    if (d.x.H != d.PH || d.x.W != d.PW || d.OH != 2 * d.PH || d.OW != 2 * d.PW) return 0;
    if ((reinterpret_cast<uintptr_t>(d.x.s0) & 15) != 0 || (d.x.C1 > 0 && (reinterpret_cast<uintptr_t>(d.x.s1) & 15) != 0)) return 0;
    if ((d.x.ab0 != nullptr && (reinterpret_cast<uintptr_t>(d.x.ab0) & 15) != 0) ||
        (d.x.ab1 != nullptr && (reinterpret_cast<uintptr_t>(d.x.ab1) & 15) != 0))
        return 0;
    if (d.x.act != SSC_ACT_NONE && d.x.act != SSC_ACT_RELU && d.x.act != SSC_ACT_LRELU) return 0;
    if (d.x.act1 > SSC_ACT_LRELU) return 0;
    if (d.sb_x != nullptr || d.sb2_x != nullptr || d.stat_mode != 0) return 0;
    const long M = (long)d.NB * d.PH * d.PW;
    if (M < 16384 || M >= 0x7fffffffL / 64) return 0;
    return 1;
}

// walkers per phase; the launch has 4 x walkers workgroups (= rows of partial sums)
static int t16_walkers(const ssc_conv_desc& d) {
    const long tiles = (long)d.NB * ((d.PH + T16_TR - 1) / T16_TR) * ((d.PW + T16_TC - 1) / T16_TC);
    const long g = (long)ssc_num_cu() / 2;      // 4 phases x CUs / 2 = two workgroups per CU (61 KB of LDS each)
    return (int)(tiles < g ? tiles : g);
}
int ssc_conv_tr4n16_rows(const ssc_conv_desc* dp) { return 4 * t16_walkers(*dp); }

int ssc_conv_tr4n16_forward(const ssc_conv_desc* dp, float* stat, void* stream) {
    if (!ssc_conv_tr4n16_supported(dp)) return -1;
    const ssc_conv_desc& d = *dp;
    const int tiles_x = (d.PW + T16_TC - 1) / T16_TC, tiles_y = (d.PH + T16_TR - 1) / T16_TR;
    const int tiles = d.NB * tiles_y * tiles_x;
    hipLaunchKernelGGL(tr4n16_kernel, dim3(t16_walkers(d), 4), dim3(256), 0, (hipStream_t)stream, d, tiles, tiles_x, tiles_y, stat);
    return (int)hipGetLastError();
}
