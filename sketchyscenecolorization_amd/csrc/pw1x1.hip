// pw1x1.hip -- the 1x1 expansion conv of the bottleneck blocks (residual_util.py:97-101, 132-136, 161-163; the Background
// module's copy bg_colorization_main.py:236-240): C/4 channels -> C, i.e. out[M][N] = act(a x + b)[M][K] W[K][N] with K = 16 ..
// 128 and N = 4 K.
//
// On the tile kernel such a launch is one to four K steps per tile between a prologue (index decode, first loads) and an
// epilogue through LDS: 22-47 us for launches whose output is 9-75 MB (1.6-12 us of HBM time) -- 61 of them per Residual
// iteration, 29 per Background forward.  Here, in the manner of fewchan.hip:
//   * the filter slice of a workgroup's 128 columns lives in REGISTERS for its life (K / 2 values per lane and 32-column
//     block), loaded once; workgroups are persistent and walk tiles of 64 rows;
//   * a tile's [64][K] input block is staged once in LDS with the folded norm + activation applied on the way (a thread keeps
//     one group of 4 channels: its constants are loaded once), every MFMA A operand is one ds_read_b32;
//   * the next tile's block is in flight (registers) while the current one is multiplied; two LDS images, one barrier per tile;
//   * accumulators leave as 128-byte row segments straight from the MFMA layout;
//   * the batch statistics of the output (its norm follows: `bn=`) are per-lane sums over the tiles a workgroup walks, one row
//     of partials per workgroup at the end, folded by bn_stats_finalize_kernel like the tile kernel's rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PW_TR 64         // rows per tile
#define PW_BN 128        // columns per workgroup

// NARROW (at most 64 outputs: the expansion 16 -> 64): the four wavefronts are 2 row blocks x 2 column blocks of the 64-row tile instead of
// 4 column blocks of which two would be empty
template <int K, bool NARROW>
__global__ __launch_bounds__(256) void pw1x1_kernel(const ssc_conv_desc d, long M, int tiles, float* __restrict__ stat) {
    constexpr int KS = K / 2;                   // MFMA steps (2 k per v_mfma_f32_32x32x2_f32)
    constexpr int LD = K + 1;                   // floats per staged row (odd: the 32 lanes of an operand read hit 32 banks)
    constexpr int K4 = K / 4;                   // 16-byte chunks per row
    constexpr int NQ = (PW_TR * K4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float pw_smem[];
    float* const img0 = pw_smem;
    float* const img1 = pw_smem + PW_TR * LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n0 = NARROW ? (wave & 1) * 32 : blockIdx.y * PW_BN + wave * 32;          // this wave's 32 columns
    const int wr = NARROW ? (wave >> 1) : 0;                // NARROW: this wave's row block
    const int col = n0 + l31;
    const bool colv = col < d.Nn;

    // ---- filter fragments: B[k][n] for k = 2 s + lhi, n = col ----
    float bf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + lhi;
        bf[s] = colv ? d.w[(long)k * d.wC1 + d.n_off + col] : 0.f;
    }

    // ---- staging: thread -> (row q * (256 / K4) + tid / K4, chunk tid % K4): the chunk is the same for every q ----
    const int ch = tid % K4, r0 = tid / K4;
    constexpr int RP = 256 / K4;                // rows per pass
    float4 ta = make_float4(1.f, 1.f, 1.f, 1.f), tb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.x.ab0 != nullptr) {
        ta = *reinterpret_cast<const float4*>(d.x.ab0 + 4 * ch);
        tb = *reinterpret_cast<const float4*>(d.x.ab0 + K + 4 * ch);
    }
    const float slope = d.x.act == SSC_ACT_RELU ? 0.f : (d.x.act == SSC_ACT_LRELU ? 0.2f : 1.f);
    float4 rv[NQ];
    auto load_tile = [&](int tile) {
        const long m0 = (long)tile * PW_TR;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const long r = m0 + r0 + RP * q;
            const bool ok = (r0 + RP * q < PW_TR) & (r < M);
            const float4 v = *reinterpret_cast<const float4*>(d.x.s0 + (ok ? r : 0) * K + 4 * ch);
            float4 t;
            t.x = fmaf(ta.x, v.x, tb.x); t.y = fmaf(ta.y, v.y, tb.y); t.z = fmaf(ta.z, v.z, tb.z); t.w = fmaf(ta.w, v.w, tb.w);
            t.x = fmaxf(t.x, slope * t.x); t.y = fmaxf(t.y, slope * t.y); t.z = fmaxf(t.z, slope * t.z); t.w = fmaxf(t.w, slope * t.w);
            rv[q] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](float* P) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int rr = r0 + RP * q;
            if (rr < PW_TR) {
                float* p = P + rr * LD + 4 * ch;
                p[0] = rv[q].x; p[1] = rv[q].y; p[2] = rv[q].z; p[3] = rv[q].w;
            }
        }
    };

    float ssum = 0.f, ssq = 0.f;                // this lane's column, over the rows of its two blocks and all its tiles
    const int G = gridDim.x;
    int tile = blockIdx.x;
    if (tile < tiles) {
        load_tile(tile);
        store_tile(img0);
    }
    __syncthreads();
    int buf = 0;
    for (; tile < tiles; tile += G) {
        const int next = tile + G;
        if (next < tiles) load_tile(next);          // in flight across the MFMAs below
        const float* P = buf ? img1 : img0;
        const float* a0p = P + (l31 + 32 * wr) * LD + lhi;
        const float* a1p = a0p + 32 * LD;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float a0 = a0p[2 * s];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf[s], acc0, 0, 0, 0);
            if (!NARROW) {
                const float a1 = a1p[2 * s];
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf[s], acc1, 0, 0, 0);
            }
        }
        // ---- epilogue: acc[r] is row (r & 3) + 8 * (r >> 2) + 4 * lhi of its 32-row block, column l31 ----
        if (colv) {
            const long m0 = (long)tile * PW_TR + 32 * wr;
            float* o = d.out + (m0 + 4 * lhi) * d.ldc + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = (r & 3) + 8 * (r >> 2);
                if (m0 + 4 * lhi + x < M) {
                    o[(long)x * d.ldc] = acc0[r];
                    ssum += acc0[r];
                    ssq += acc0[r] * acc0[r];
                }
                if (!NARROW && m0 + 32 + 4 * lhi + x < M) {
                    o[(long)(32 + x) * d.ldc] = acc1[r];
                    ssum += acc1[r];
                    ssq += acc1[r] * acc1[r];
                }
            }
        }
        if (next < tiles) store_tile(buf ? img0 : img1);
        __syncthreads();        // one barrier per tile: the image written above was last read before the previous barrier
        buf ^= 1;
    }
    if (stat != nullptr && NARROW) {       // the two row blocks of a column fold through LDS (row block 0 first)
        __shared__ float nred[2][2][32];
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (wr == 1 && lhi == 0) { nred[0][wave & 1][l31] = ssum; nred[1][wave & 1][l31] = ssq; }
        __syncthreads();
        if (wr == 0 && lhi == 0 && colv) {
            float* sp = stat + (long)blockIdx.x * 2 * d.Nstore;
            sp[col] = ssum + nred[0][wave & 1][l31];
            sp[d.Nstore + col] = ssq + nred[1][wave & 1][l31];
        }
    } else if (stat != nullptr) {      // one row [sum | sum of squares] per workgroup of a column group: rows blockIdx.x, width Nstore
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (lhi == 0 && colv) {
            float* sp = stat + (long)blockIdx.x * 2 * d.Nstore;
            sp[col] = ssum;
            sp[d.Nstore + col] = ssq;
        }
    }
}

static bool pw_on() {
    static int on = -1;         // SSC_PW1X1=0: the tile kernel (A/B)
    if (on < 0) {
        const char* e = ssc_dev_getenv("SSC_PW1X1");
        on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

extern "C" int ssc_conv_pw1x1_supported(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    const int K = d.x.C0;
    if (!pw_on()) return 0;
    if (d.x.C1 != 0 || (K != 16 && K != 32 && K != 64 && K != 128) || d.k_real != K || d.wC0 != K) return 0;
    if (d.nphase != 1 || d.TH != 1 || d.TW != 1 || d.KH != 1 || d.KW != 1 || d.in_stride != 1 || d.ioff_y != 0 || d.ioff_x != 0 ||
        d.bmode != 0 || d.out_stride != 1 || d.ooff_y != 0 || d.ooff_x != 0)
        return 0;
    if (d.bias != nullptr || d.epi != 0 || d.accumulate || d.n_off != 0 || d.Nn != d.Nstore || (d.Nn & 31) != 0 || d.Nn < 64 ||
        d.Nstore > d.ldc)
        return 0;
    if (d.x.act != SSC_ACT_NONE && d.x.act != SSC_ACT_RELU && d.x.act != SSC_ACT_LRELU) return 0;
    if (d.OH != d.PH || d.OW != d.PW || d.x.H != d.PH || d.x.W != d.PW) return 0;
    if ((reinterpret_cast<uintptr_t>(d.x.s0) & 15) != 0 || (d.x.ab0 != nullptr && (reinterpret_cast<uintptr_t>(d.x.ab0) & 15) != 0))
        return 0;
    const long M = (long)d.NB * d.PH * d.PW;
    if (M < 2048 || M >= 0x7fffffffL) return 0;
    if (d.sb_x != nullptr || d.stat_mode != 0) return 0;
    return 1;
}

// workgroups that walk the row tiles (per column group): as many as are resident at once
int ssc_conv_pw1x1_walkers(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    const long M = (long)d.NB * d.PH * d.PW;
    const int tiles = (int)((M + PW_TR - 1) / PW_TR);
    const int colg = (d.Nn + PW_BN - 1) / PW_BN;
    const int K = d.x.C0;
    const int per_cu = K == 128 ? 2 : 4;        // LDS: 2 x 64 x (K + 1) floats per workgroup
    int g = ssc_num_cu() * per_cu / colg;
    if (g < 1) g = 1;
    return tiles < g ? tiles : g;
}

// stat != NULL: rows of [sum | sum of squares] of the output columns, one per walker (ssc_conv_pw1x1_walkers rows of 2 x Nstore)
int ssc_conv_pw1x1_forward(const ssc_conv_desc* dp, float* stat, void* stream) {
    if (!ssc_conv_pw1x1_supported(dp)) return -1;
    const ssc_conv_desc& d = *dp;
    const long M = (long)d.NB * d.PH * d.PW;
    const int tiles = (int)((M + PW_TR - 1) / PW_TR);
    const dim3 grid((unsigned)ssc_conv_pw1x1_walkers(dp), (unsigned)((d.Nn + PW_BN - 1) / PW_BN));
    hipStream_t st = (hipStream_t)stream;
    const bool narrow = d.Nn <= 64;
    const size_t lds = (size_t)2 * PW_TR * (d.x.C0 + 1) * sizeof(float);
#define PW_LAUNCH(KK)                                                                                              \
    {                                                                                                              \
        static unsigned long long attr_done = 0;                                                                   \
        if (narrow) {                                                                                              \
            const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&pw1x1_kernel<KK, true>), (int)lds, &attr_done);   \
            if (arc != 0) return arc;                                                                              \
            hipLaunchKernelGGL((pw1x1_kernel<KK, true>), grid, dim3(256), lds, st, d, M, tiles, stat);             \
        } else {                                                                                                   \
            static unsigned long long attr_done2 = 0;                                                              \
            const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&pw1x1_kernel<KK, false>), (int)lds, &attr_done2); \
            if (arc != 0) return arc;                                                                              \
            hipLaunchKernelGGL((pw1x1_kernel<KK, false>), grid, dim3(256), lds, st, d, M, tiles, stat);            \
        }                                                                                                          \
    }
    switch (d.x.C0) {
        case 16: PW_LAUNCH(16) break;
        case 32: PW_LAUNCH(32) break;
        case 64: PW_LAUNCH(64) break;
        default: PW_LAUNCH(128) break;
    }
#undef PW_LAUNCH
    return (int)hipGetLastError();
}
