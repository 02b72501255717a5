// fewchan7.hip -- the Residual / Background generators' first conv: 7x7 stride 2 SAME over the 3 image channels (padded to 4) to 64
// outputs + batch-statistics norm (encoder_1, bg_colorization_main.py:217-236 / residual generator; SAME = 2 before, 3 after).
//
// As an implicit GEMM K = 49 taps x 3 = 147: on the tile kernel every tap is a 32-wide K chunk holding 3 real channels -- measured
// 251 us at 768^2 (44 TFLOP/s on the real FLOPs) for 9 MB read and 151 MB written.  The construction of fewchan.hip with a 7x7
// footprint: the whole filter slice in registers (98 values per lane and 32-column block), persistent workgroups, 4 x 32 output
// tiles whose (13 x 69)-pixel input patch is staged once in LDS, every MFMA A operand one ds_read_b32 with an immediate offset,
// the next patch in flight during the MFMAs -- and the batch statistics of the output as per-lane sums, one row of partials per
// workgroup (ssc_conv_forward_bn).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define F7_TR 4          // output rows per tile
#define F7_TC 32         // output columns per tile
#define F7_KT 7
#define F7_PR (2 * F7_TR + F7_KT - 2)
#define F7_PC (2 * F7_TC + F7_KT - 2)

__global__ __launch_bounds__(256) void fewchan7_conv_kernel(const ssc_conv_desc d, int tiles, int tiles_x, int tiles_y,
                                                            float* __restrict__ stat) {
    constexpr int C = 4;
    constexpr int K = F7_KT * F7_KT * C, KS = K / 2;    // 196, 98
    constexpr int CP = C + 1;                   // floats per patch pixel in LDS
    constexpr int PSZ = F7_PR * F7_PC * CP;
    constexpr int NQ = (F7_PR * F7_PC + 255) / 256;
    __shared__ float patch[2][PSZ];
    __shared__ float red[2][2][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wp = wave >> 1, wj = wave & 1;    // this wave: output rows 2*wp, 2*wp+1 of the tile, columns [32*wj, 32*wj+32)
    const int col = wj * 32 + l31;

    // ---- filter fragments: B[k][n] for k = 2s + lhi = (tap, c), n = col ----
    float bf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + lhi;
        const int tap = k / C, c = k - tap * C;
        const bool v = (c < d.k_real) & (col < d.Nn);
        const float w = d.w[v ? ((long)tap * d.wC0 + c) * d.wC1 + d.n_off + col : 0];
        bf[s] = v ? w : 0.f;
    }

    // ---- patch staging: one 16-byte pixel per thread and pass ----
    float4 rv[NQ];
    auto load_patch = [&](int tile) {
        const int tx = tile % tiles_x;
        const int r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const int iy0 = 2 * F7_TR * ty - 2, ix0 = 2 * F7_TC * tx - 2;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pix = tid + 256 * q;
            const int pr = pix / F7_PC, pc = pix - pr * F7_PC;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = (pix < F7_PR * F7_PC) & ((unsigned)iy < (unsigned)d.x.H) & ((unsigned)ix < (unsigned)d.x.W);
            const float4 v = *reinterpret_cast<const float4*>(d.x.s0 + (ok ? (((long)n * d.x.H + iy) * d.x.W + ix) * C : 0));
            rv[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_patch = [&](float* P) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pix = tid + 256 * q;
            if (pix < F7_PR * F7_PC) {
                float* p = P + pix * CP;
                p[0] = rv[q].x; p[1] = rv[q].y; p[2] = rv[q].z; p[3] = rv[q].w;
            }
        }
    };

    float ssum = 0.f, ssq = 0.f;
    const int G = gridDim.x;
    int tile = blockIdx.x;
    if (tile < tiles) {
        load_patch(tile);
        store_patch(patch[0]);
    }
    __syncthreads();
    int buf = 0;
    // lane bases of the two output rows of this wave: pixel (2*(2*wp+r), 2*l31) of the patch, + lhi (k parity = channel parity)
    const int ab0 = ((2 * (2 * wp)) * F7_PC + 2 * l31) * CP + lhi;
    const int ab1 = ((2 * (2 * wp + 1)) * F7_PC + 2 * l31) * CP + lhi;
    for (; tile < tiles; tile += G) {
        const int next = tile + G;
        if (next < tiles) load_patch(next);          // in flight across the MFMAs below
        const float* P = patch[buf];
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = 2 * s;
            const int tap = k0 / C, c0 = k0 - tap * C;
            const int off = ((tap / F7_KT) * F7_PC + (tap % F7_KT)) * CP + c0;      // compile-time after unrolling
            const float a0 = P[ab0 + off], a1 = P[ab1 + off];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf[s], acc1, 0, 0, 0);
        }
        // ---- epilogue: acc[r] is row (r & 3) + 8 * (r >> 2) + 4 * lhi (= output x inside the tile), column l31 ----
        {
            const int tx = tile % tiles_x;
            const int rr = tile / tiles_x;
            const int ty = rr % tiles_y, n = rr / tiles_y;
            if (col < d.Nstore) {
                const int oy = F7_TR * ty + 2 * wp;
                float* o0 = d.out + (((long)n * d.OH + oy) * d.OW + F7_TC * tx + 4 * lhi) * d.ldc + col;
                float* o1 = o0 + (long)d.OW * d.ldc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int x = (r & 3) + 8 * (r >> 2);
                    o0[(long)x * d.ldc] = acc0[r];
                    o1[(long)x * d.ldc] = acc1[r];
                    ssum += acc0[r] + acc1[r];
                    ssq += acc0[r] * acc0[r] + acc1[r] * acc1[r];
                }
            }
        }
        if (next < tiles) store_patch(patch[buf ^ 1]);
        __syncthreads();        // one barrier per tile: the image written above was last read before the previous barrier
        buf ^= 1;
    }
    if (stat != nullptr) {      // one row [sum | sum of squares] per workgroup: lane halves, then the two row pairs, in order
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (wp == 1 && lhi == 0) { red[0][wj][l31] = ssum; red[1][wj][l31] = ssq; }
        __syncthreads();
        if (wp == 0 && lhi == 0 && col < d.Nstore) {
            float* sp = stat + (long)blockIdx.x * 2 * d.Nstore;
            sp[col] = ssum + red[0][wj][l31];
            sp[d.Nstore + col] = ssq + red[1][wj][l31];
        }
    }
}

static bool f7_on() {
    static int on = -1;         // SSC_FEWCHAN7=0: the tile kernel (A/B)
    if (on < 0) {
        const char* e = ssc_dev_getenv("SSC_FEWCHAN7");
        on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

extern "C" int ssc_conv_fewchan7_supported(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    if (!f7_on()) return 0;
    if (d.x.C1 != 0 || d.x.C0 != 4 || d.x.ab0 != nullptr || d.x.act != SSC_ACT_NONE) return 0;
    if (d.nphase != 1 || d.TH != 7 || d.TW != 7 || d.KH != 7 || d.KW != 7 || d.in_stride != 2 || d.ioff_y != -2 || d.ioff_x != -2 ||
        d.ky0 != 0 || d.kx0 != 0 || d.kstep != 1 || d.bmode != 0)
        return 0;
    if (d.k_real < 1 || d.k_real > 4 || d.wC0 < d.k_real || d.Nn < 33 || d.Nn > 64 || d.Nstore != d.Nn || d.Nstore > d.ldc ||
        d.n_off + d.Nn > d.wC1)
        return 0;
    if (d.bias != nullptr || d.epi != 0 || d.accumulate || d.out_stride != 1 || d.ooff_y != 0 || d.ooff_x != 0) return 0;
    if (d.OH != d.PH || d.OW != d.PW || (d.PH % F7_TR) != 0 || (d.PW % F7_TC) != 0) return 0;
    if (2 * d.PH != d.x.H || 2 * d.PW != d.x.W) return 0;
    if ((reinterpret_cast<uintptr_t>(d.x.s0) & 15) != 0) return 0;
    if ((long)d.NB * (d.PH / F7_TR) * (d.PW / F7_TC) >= 0x7fffffffL) return 0;
    if (d.sb_x != nullptr || d.sb2_x != nullptr || d.stat_mode != 0) return 0;
    return 1;
}

// persistent workgroups (= rows of partial sums): 2 per CU (36 KB of LDS, ~150 registers per lane)
int ssc_conv_fewchan7_walkers(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    const long tiles = (long)d.NB * (d.PH / F7_TR) * (d.PW / F7_TC);
    const long g = (long)ssc_num_cu() * 2;
    return (int)(tiles < g ? tiles : g);
}

int ssc_conv_fewchan7_forward(const ssc_conv_desc* dp, float* stat, void* stream) {
    if (!ssc_conv_fewchan7_supported(dp)) return -1;
    const ssc_conv_desc& d = *dp;
    const int tiles_x = d.PW / F7_TC, tiles_y = d.PH / F7_TR;
    const int tiles = d.NB * tiles_y * tiles_x;
    hipLaunchKernelGGL(fewchan7_conv_kernel, dim3(ssc_conv_fewchan7_walkers(dp)), dim3(256), 0, (hipStream_t)stream, d, tiles, tiles_x,
                       tiles_y, stat);
    return (int)hipGetLastError();
}
