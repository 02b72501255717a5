// s2n16.hip -- the 4x4 convs from 64 channels to 16 on the 16-column fp32 MFMA: stride 1 (block_1 of the plain bottlenecks at the
// highest resolution: conv_ex 4x4 SAME, residual_util.py:147-151 with C/4 = 16; the Background generator's copy at 768^2) and
// stride 2 (the same geometry as the data gradient of the k = 4 stride-2 transposed conv 16 -> 64).
//
// As an implicit GEMM this is M = N*H*W/4 rows by 16 columns with K = 16 taps x 64 channels = 1024.  The tile kernel's narrowest
// tile is 32 columns wide (v_mfma_f32_32x32x2_f32): half of its matrix work multiplies padding -- measured 205 us at batch 32
// (47 TFLOP/s on the real FLOPs), 394 us at 768^2.  Here:
//   * v_mfma_f32_16x16x4_f32: 16 output pixels x 16 channels x 4 k per instruction at the same 64 FLOP/clk/SIMD -- no padded
//     columns;
//   * K is split over the workgroup's four wavefronts by filter ROW (ky = wave: 4 taps x 64 channels = 256 k), so a lane's share
//     of the filter is 64 registers, loaded once per (persistent) workgroup; the four partial sums of a pixel group meet in LDS;
//   * a tile is 2 output rows x 16 pixels; its (6 x 34)-pixel input patch is staged once in LDS with the folded norm +
//     activation applied on the way (zeros outside the image: the padding is of the ACTIVATED tensor); an A operand is one
//     ds_read_b32 with an immediate offset; pixel stride 66 floats -> the 16 pixels x 4 k of an operand hit 64 distinct banks;
//   * the next tile's patch is in flight in registers during the MFMAs;
//   * the batch statistics of the output (its norm follows: ssc_conv_forward_bn) -- or, when the launch is the data gradient of the
//     k = 4 stride-2 transposed conv 16 -> 64 of the last decoder bottleneck (the same geometry), the two sums of the backward of
//     the norm its output is the gradient of (ssc_conv_forward_bnbwd) -- are per-thread sums over the tiles a workgroup walks (a
//     thread owns one output column), one row of partials per workgroup.
// Arithmetic: 2 * 1024 * 16 = 32 KFLOP per output pixel: 9.7 GFLOP at batch 32 / 192^2 = 61 us of fp32 MFMA; the input is 302 MB.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define S2_TC 16          // output pixels per row of a tile (one MFMA's rows)
#define S2_C 64

// STRIDE 2: tiles of 2 output rows; STRIDE 1: 4 rows.  Patch pixel stride PST with STRIDE * PST = 4 (mod 64): the 16 pixels x 4 k
// of an operand read hit 64 distinct banks
template <int STRIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void s2n16_kernel(const ssc_conv_desc d, int tiles, int tiles_x,
                                                                                           int tiles_y, float* __restrict__ stat) {
    constexpr int S2_TR = STRIDE == 2 ? 2 : 4;
    constexpr int S2_PR = STRIDE * S2_TR + 4 - STRIDE, S2_PC = STRIDE * S2_TC + 4 - STRIDE;
    constexpr int S2_PST = STRIDE == 2 ? 66 : 68;
    constexpr int S2_PSZ = S2_PR * S2_PC * S2_PST;
    __shared__ __attribute__((aligned(16))) float patch[S2_PSZ];            // 53.9 KB (stride 2) / 36.2 KB (stride 1)
    __shared__ __attribute__((aligned(16))) float red[4 * S2_TR * 256];     // [wave][row][lane][4]: 8 / 16 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int H = d.x.H, W = d.x.W;
    const bool colv = l15 < d.Nn;

    // ---- filter: bf[s] = w[ky = wave][kx = s / 16][c = 4 * (s % 16) + kq][n_off + l15]   (bmode 0: [ky][kx][c][n]) ----
    float bf[64];
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const int kx = s >> 4, c = 4 * (s & 15) + kq;
        const long idx = ((long)((d.ky0 + wave * d.kstep) * 4 + d.kx0 + kx * d.kstep) * d.wC0 + c) * d.wC1 + d.n_off + (colv ? l15 : 0);
        const float wv = d.w[idx];
        bf[s] = colv ? wv : 0.f;
    }

    // ---- patch staging: thread -> (pixel tid / 16 + 16 q, 16-byte chunk tid % 16) ----
    const int c4 = (tid & 15) * 4;
    float4 ta = make_float4(1.f, 1.f, 1.f, 1.f), tb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.x.ab0 != nullptr) {
        ta = *reinterpret_cast<const float4*>(d.x.ab0 + c4);
        tb = *reinterpret_cast<const float4*>(d.x.ab0 + S2_C + c4);
    }
    const float slope = d.x.act == SSC_ACT_RELU ? 0.f : (d.x.act == SSC_ACT_LRELU ? 0.2f : 1.f);
    constexpr int NQ = (S2_PR * S2_PC + 15) / 16;       // 13
    float4 rv[NQ];
    auto load_patch = [&](int tile) {
        const int tx = tile % tiles_x;
        const int r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const int iy0 = STRIDE * S2_TR * ty - 1, ix0 = STRIDE * S2_TC * tx - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pos = (tid >> 4) + 16 * q;
            const int pr = pos / S2_PC, pc = pos - pr * S2_PC;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = (pos < S2_PR * S2_PC) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            const float4 v = *reinterpret_cast<const float4*>(d.x.s0 + (ok ? (((long)n * H + iy) * W + ix) * S2_C : 0) + c4);
            float4 t;
            t.x = fmaf(ta.x, v.x, tb.x); t.y = fmaf(ta.y, v.y, tb.y); t.z = fmaf(ta.z, v.z, tb.z); t.w = fmaf(ta.w, v.w, tb.w);
            t.x = fmaxf(t.x, slope * t.x); t.y = fmaxf(t.y, slope * t.y); t.z = fmaxf(t.z, slope * t.z); t.w = fmaxf(t.w, slope * t.w);
            rv[q] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pos = (tid >> 4) + 16 * q;
            if (pos < S2_PR * S2_PC) {
                float* p = patch + pos * S2_PST + c4;       // 8-byte aligned (the pixel stride is even): two 8-byte stores
                *reinterpret_cast<float2*>(p) = make_float2(rv[q].x, rv[q].y);
                *reinterpret_cast<float2*>(p + 2) = make_float2(rv[q].z, rv[q].w);
            }
        }
    };

    // the thread that finishes output (row g, pixel (tid >> 4) & 15, column tid & 15) -- for both rows g of a tile
    const int ocol = tid & 15, opix = tid >> 4;
    float ssum = 0.f, ssq = 0.f;
    // data-gradient use (the gradient of the k = 4 stride-2 transposed conv w.r.t. its input): the output is the gradient w.r.t.
    // act(norm(x)); the two sums of that norm's backward instead of the batch statistics (ssc_conv_forward_bnbwd)
    const bool bwd = stat != nullptr && d.sb_x != nullptr;
    float sa = 1.f, sb = 0.f, smu = 0.f, srs = 1.f, sneg = 1.f;
    if (bwd && ocol < d.Nn) {
        sa = d.sb_ab[ocol]; sb = d.sb_ab[d.Nstore + ocol];
        smu = d.sb_stats[ocol]; srs = d.sb_stats[d.Nstore + ocol];
        sneg = d.sb_act == SSC_ACT_RELU ? 0.f : (d.sb_act == SSC_ACT_LRELU ? 0.2f : 1.f);
    }

    const int G = gridDim.x;
    int tile = blockIdx.x;
    if (tile < tiles) {
        load_patch(tile);
        store_patch();
    }
    __syncthreads();
    // A operand of lane (pixel l15, kq) for output row g, MFMA step s = (kx, c4'): patch[STRIDE g + wave][STRIDE l15 + kx][4 c4' + kq]
    const float* const A0 = patch + (wave * S2_PC + STRIDE * l15) * S2_PST + kq;
    for (; tile < tiles; tile += G) {
        const int next = tile + G;
        if (next < tiles) load_patch(next);          // in flight across the MFMAs below
        f32x4 acc[S2_TR];
#pragma unroll
        for (int g = 0; g < S2_TR; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const int off = (s >> 4) * S2_PST + 4 * (s & 15);       // compile-time after unrolling
#pragma unroll
            for (int g = 0; g < S2_TR; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[STRIDE * g * S2_PC * S2_PST + off], bf[s], acc[g], 0, 0, 0);
        }
        // partial sums of this wave's filter row: acc[g][r] is pixel 4 * kq + r, column l15
#pragma unroll
        for (int g = 0; g < S2_TR; ++g) *reinterpret_cast<f32x4*>(red + ((wave * S2_TR + g) * 64 + lane) * 4) = acc[g];
        __syncthreads();            // every wave is done with the patch; the partial sums are visible
        if (next < tiles) store_patch();
        {
            const int tx = tile % tiles_x;
            const int rr = tile / tiles_x;
            const int ty = rr % tiles_y, n = rr / tiles_y;
            const int ox = S2_TC * tx + opix;
            // (pixel opix, column ocol) sits in lane (opix >> 2) * 16 + ocol, register opix & 3
            const float* rp = red + ((opix >> 2) * 16 + ocol) * 4 + (opix & 3);
#pragma unroll
            for (int g = 0; g < S2_TR; ++g) {
                const int oy = S2_TR * ty + g;
                const float v = ((rp[(0 * S2_TR + g) * 256] + rp[(1 * S2_TR + g) * 256]) + rp[(2 * S2_TR + g) * 256]) +
                                rp[(3 * S2_TR + g) * 256];
                if ((ocol < d.Nn) & (oy < d.PH) & (ox < d.PW)) {
                    const long pix = ((long)n * d.PH + oy) * d.PW + ox;
                    d.out[pix * d.ldc + ocol] = v;
                    if (bwd) {
                        const float xv = d.sb_x[pix * d.sb_ldx + ocol];
                        const float dz = v * (fmaf(sa, xv, sb) > 0.f ? 1.f : sneg);
                        ssum += dz;
                        ssq += dz * (xv - smu) * srs;
                    } else {
                        ssum += v;
                        ssq += v * v;
                    }
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
            float* sp = stat + (long)blockIdx.x * 2 * d.Nstore;
            sp[tid] = a;
            sp[d.Nstore + tid] = b;
        }
    }
}

static bool s2_on() {
    static int on = -1;         // SSC_S2N16=0: the tile kernel (A/B)
    if (on < 0) {
        const char* e = ssc_dev_getenv("SSC_S2N16");
        on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

extern "C" int ssc_conv_s2n16_supported(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    if (!s2_on()) return 0;
    if (d.x.C1 != 0 || d.x.C0 != S2_C || d.k_real != S2_C || d.wC0 != S2_C) return 0;
    if (d.nphase != 1 || d.TH != 4 || d.TW != 4 || d.KH != 4 || d.KW != 4 || (d.in_stride != 1 && d.in_stride != 2) || d.ioff_y != -1 || d.ioff_x != -1 ||
        d.out_stride != 1 || d.ooff_y != 0 || d.ooff_x != 0 || d.bmode != 0 || d.ky0 != 0 || d.kx0 != 0 || d.kstep != 1)
        return 0;
    if (d.bias != nullptr || d.epi != 0 || d.accumulate || d.Nn < 4 || d.Nn > 16 || d.Nn != d.Nstore || d.Nstore > d.ldc ||
        d.n_off + d.Nn > d.wC1)
        return 0;
    if (d.x.act != SSC_ACT_NONE && d.x.act != SSC_ACT_RELU && d.x.act != SSC_ACT_LRELU) return 0;
    if (d.OH != d.PH || d.OW != d.PW || d.x.H != d.in_stride * d.PH || d.x.W != d.in_stride * d.PW) return 0;
    if ((reinterpret_cast<uintptr_t>(d.x.s0) & 15) != 0 || (d.x.ab0 != nullptr && (reinterpret_cast<uintptr_t>(d.x.ab0) & 15) != 0))
        return 0;
    const long M = (long)d.NB * d.PH * d.PW;
    if (M < 32768 || M >= 0x7fffffffL / 64) return 0;
    if (d.sb2_x != nullptr || d.stat_mode != 0) return 0;
    return 1;
}

// persistent workgroups (= rows of partial sums)
int ssc_conv_s2n16_walkers(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    const int tr = d.in_stride == 2 ? 2 : 4;
    const long tiles = (long)d.NB * ((d.PH + tr - 1) / tr) * ((d.PW + S2_TC - 1) / S2_TC);
    const long g = (long)ssc_num_cu() * 2;      // 62 / 52 KB of LDS per workgroup, 2 waves per SIMD
    return (int)(tiles < g ? tiles : g);
}

int ssc_conv_s2n16_forward(const ssc_conv_desc* dp, float* stat, void* stream) {
    if (!ssc_conv_s2n16_supported(dp)) return -1;
    const ssc_conv_desc& d = *dp;
    const int tr = d.in_stride == 2 ? 2 : 4;
    const int tiles_x = (d.PW + S2_TC - 1) / S2_TC, tiles_y = (d.PH + tr - 1) / tr;
    const int tiles = d.NB * tiles_y * tiles_x;
    const int G = ssc_conv_s2n16_walkers(dp);
    if (d.in_stride == 2)
        hipLaunchKernelGGL(s2n16_kernel<2>, dim3(G), dim3(256), 0, (hipStream_t)stream, d, tiles, tiles_x, tiles_y, stat);
    else
        hipLaunchKernelGGL(s2n16_kernel<1>, dim3(G), dim3(256), 0, (hipStream_t)stream, d, tiles, tiles_x, tiles_y, stat);
    return (int)hipGetLastError();
}
