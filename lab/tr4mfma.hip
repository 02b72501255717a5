// tr4mfma.hip -- the generator's last layer: the k = 4, stride-2 transposed conv from 128 channels to the 3 image channels (+ bias,
// tanh; models_collection.py:529-534, the Residual / Background generators' copy) on the matrix pipe.
//
// 3 output channels waste 29 of a 32-column MFMA tile's 32 columns, so this layer ran on the vector ALUs (narrow.hip: 124 us at batch
// 32, 24 TFLOP/s, bound by the scalar filter fetches).  v_mfma_f32_4x4x1_16b_f32 multiplies 16 independent 4 x 1 by 1 x 4 blocks: here
//   * the 4 rows of a block are 4 neighbouring lattice pixels, its 4 columns the output channels (3 + 1 of padding), and the 16
//     BLOCKS are 16 different input channels -- a 16-way split of K whose partial sums are added across lanes at the end;
//   * so a lane's B operands -- filter values of ITS channel residue and output channel for every (sub-pixel phase, tap, group of 16
//     channels) -- are 4 x 4 x 8 = 128 registers loaded once per workgroup: the filter costs no LDS bandwidth and no scalar fetch;
//   * an A operand is one ds_read_b32 of the staged input patch (folded norm + activation applied once per element on the way in)
//     and feeds every phase that uses that input offset: 72 reads for 128 MFMAs per 4 pixels -- 72 of the 128 B/clk of LDS;
//   * persistent workgroups (one per CU: 256 filter + accumulator registers per lane are the price), 4 x 16-pixel tiles, the next
//     tile's patch in flight in registers during the MFMAs, two LDS images, one barrier per tile;
//   * the 16 partial sums meet through two DPP adds (row_ror 8, 4) and one pass through 1 KB of LDS per wavefront, after which
//     lane e = (phase, pixel, channel) owns one output element: bias, tanh, store (128 contiguous bytes per output row).
// Arithmetic: 2 * 4 phases * 4 taps * 128 * 4 columns = 16 KFLOP per lattice pixel; 4.8 GFLOP at batch 32 = 31 us of fp32 MFMA; the
// input is 151 MB (38 us of HBM at 4 TB/s).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define T4_TY 4          // lattice rows per tile (one per wavefront)
#define T4_TX 16         // lattice columns per tile (4 groups of 4 pixels)
#define T4_PY (T4_TY + 2)
#define T4_PX (T4_TX + 2)
#define T4_C 128
#define T4_PST 144       // floats per patch pixel: 128 + 16 -> the 4 pixels x 16 channels of an operand read hit 64 distinct banks
#define T4_PSZ (T4_PY * T4_PX * T4_PST)

// v + (v rotated right by 8 / 4 lanes within its row of 16): folds into one v_add_f32_dpp each
__device__ __forceinline__ float t4_add_ror8(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));
}
__device__ __forceinline__ float t4_add_ror4(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));
}

__global__ __launch_bounds__(256) void tr4_mfma_kernel(const ssc_conv_desc d, int tiles, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float t4_smem[];
    float* const img0 = t4_smem;
    float* const img1 = t4_smem + T4_PSZ;
    float* const red = t4_smem + 2 * T4_PSZ;        // [wave][row of 16 lanes][phase][channel][pixel]: 256 floats per wave

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = lane >> 2, q4 = lane & 3;         // MFMA block (= channel residue mod 16); row (A: pixel) / column (B: channel)
    const int H = d.x.H, W = d.x.W;

    // ---- filter: bw[phase][tap][c16] = f[tap index][q4][16 * c16 + b]  (bmode 1: f[ky][kx][n][c], c contiguous) ----
    float bw[4][4][8];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ry = ph >> 1, rx = ph & 1, ty = t >> 1, tx = t & 1;
            const int tap = (3 - ry - 2 * ty) * 4 + (3 - rx - 2 * tx);
#pragma unroll
            for (int c16 = 0; c16 < 8; ++c16) {
                const int c = 16 * c16 + b;
                const bool v = (q4 < d.Nn) & (c < d.k_real);        // branch-free: a clamped address, the value selected
                const long idx = ((long)tap * d.wC0 + d.n_off + (v ? q4 : 0)) * d.wC1 + (v ? c : 0);
                const float wv = d.w[idx];
                bw[ph][t][c16] = v ? wv : 0.f;
            }
        }

    // ---- patch staging: thread -> (pixel tid / 32 + 8 q, 16-byte chunk tid % 32): its chunk, source and constants are fixed ----
    const int c4 = (tid & 31) * 4;
    const bool first = c4 < d.x.C0;
    const float* const src = first ? d.x.s0 : d.x.s1;
    const int cs = first ? d.x.C0 : d.x.C1;
    const int cc = first ? c4 : c4 - d.x.C0;
    const int act = (!first && d.x.act1 >= 0) ? d.x.act1 : d.x.act;
    const float slope = act == SSC_ACT_RELU ? 0.f : (act == SSC_ACT_LRELU ? 0.2f : 1.f);     // act(t) = max(t, slope * t), branch-free
    float4 ta = make_float4(1.f, 1.f, 1.f, 1.f), tb = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const float* abp = first ? d.x.ab0 : d.x.ab1;
        if (abp != nullptr) {
            ta = *reinterpret_cast<const float4*>(abp + cc);
            tb = *reinterpret_cast<const float4*>(abp + cs + cc);
        }
    }
    constexpr int NQ = (T4_PY * T4_PX + 7) / 8;     // 14
    float4 rv[NQ];
    auto load_patch = [&](int tile) {
        const int tx = tile % tiles_x;
        const int r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const int iy0 = T4_TY * ty - 1, ix0 = T4_TX * tx - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pos = (tid >> 5) + 8 * q;
            const int pr = pos / T4_PX, pc = pos - pr * T4_PX;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = (pos < T4_PY * T4_PX) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            const float4 v = *reinterpret_cast<const float4*>(src + (ok ? (((long)n * H + iy) * W + ix) * cs : 0) + cc);
            float4 t;
            t.x = fmaf(ta.x, v.x, tb.x); t.y = fmaf(ta.y, v.y, tb.y); t.z = fmaf(ta.z, v.z, tb.z); t.w = fmaf(ta.w, v.w, tb.w);
            t.x = fmaxf(t.x, slope * t.x); t.y = fmaxf(t.y, slope * t.y); t.z = fmaxf(t.z, slope * t.z); t.w = fmaxf(t.w, slope * t.w);
            rv[q] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_patch = [&](float* P) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pos = (tid >> 5) + 8 * q;
            if (pos < T4_PY * T4_PX) *reinterpret_cast<float4*>(P + pos * T4_PST + c4) = rv[q];
        }
    };

    const float bias = (d.bias != nullptr && q4 < d.Nn) ? d.bias[q4] : 0.f;     // lane e = (phase, pixel, channel q4) at the end
    float* const rw = red + wave * 256;

    const int G = gridDim.x;
    int tile = blockIdx.x;
    if (tile < tiles) {
        load_patch(tile);
        store_patch(img0);
    }
    __syncthreads();
    int buf = 0;
    for (; tile < tiles; tile += G) {
        const int next = tile + G;
        if (next < tiles) load_patch(next);         // in flight across the MFMAs below
        const float* const P = buf ? img1 : img0;
        const int txi = tile % tiles_x;
        const int rr = tile / tiles_x;
        const int tyi = rr % tiles_y, n = rr / tiles_y;
        const int py = T4_TY * tyi + wave;
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            // A operand of lane (b, q4): pixel 4 g + q4 of the wave's row, channel 16 c16 + b, at patch offset (oy, ox)
            const float* const A = P + (wave * T4_PX + 4 * g + q4) * T4_PST + b;
            f32x4 acc[4];
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) acc[ph] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // per group of 16 channels: the 9 offsets' operands (read one group ahead: a wavefront is alone on its SIMD), then 16
            // MFMAs that walk the four phases' accumulators in turn
            float an[9];
#pragma unroll
            for (int o = 0; o < 9; ++o) an[o] = A[((o / 3) * T4_PX + (o % 3)) * T4_PST];
#pragma unroll
            for (int c16 = 0; c16 < 8; ++c16) {
                float a[9];
#pragma unroll
                for (int o = 0; o < 9; ++o) a[o] = an[o];
                if (c16 < 7) {
#pragma unroll
                    for (int o = 0; o < 9; ++o) an[o] = A[((o / 3) * T4_PX + (o % 3)) * T4_PST + 16 * (c16 + 1)];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph) {
                        const int ry = ph >> 1, rx = ph & 1, ty = t >> 1, tx = t & 1;
                        acc[ph] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[(ry + ty) * 3 + rx + tx], bw[ph][t][c16], acc[ph], 0, 0, 0);
                    }
            }
            // ---- the 16 blocks' partial sums: within a row of 16 lanes by DPP, across the 4 rows through LDS ----
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                f32x4 s;
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] = t4_add_ror4(t4_add_ror8(acc[ph][i]));
                if ((lane & 12) == 0)       // one block per row writes: [row][phase][channel q4][pixel i]
                    *reinterpret_cast<f32x4*>(rw + (((lane >> 4) * 4 + ph) * 4 + q4) * 4) = s;
            }
            // lane e = (phase e >> 4, pixel (e >> 2) & 3, channel e & 3)
            {
                const int ph = lane >> 4, pi = (lane >> 2) & 3;
                const float* r0 = rw + ((ph * 4 + q4) * 4) + pi;
                float v = ((r0[0] + r0[64]) + r0[128]) + r0[192];
                const int px = T4_TX * txi + 4 * g + pi;
                if ((py < d.PH) & (px < d.PW) & (q4 < d.Nstore)) {
                    float o = 0.f;          // columns in [Nn, Nstore) are channel padding: written as 0
                    if (q4 < d.Nn) {
                        o = v + bias;
                        if (d.epi == 1) o = tanhf(o);
                        else if (d.epi == 2) o = fmaxf(o, 0.2f * o);
                    }
                    d.out[(((long)n * d.OH + 2 * py + (ph >> 1)) * d.OW + 2 * px + (ph & 1)) * d.ldc + q4] = o;
                }
            }
        }
        if (next < tiles) store_patch(buf ? img0 : img1);
        __syncthreads();        // one barrier per tile: the image written above was last read before the previous barrier
        buf ^= 1;
    }
}

// OFF by default (SSC_TR4_MFMA=1 turns it on; read at every call so that a test can): measured 70 us against the vector-ALU kernel's
// 80 us at batch 16 (generator inference + 0.5 %), but 17.89 against 17.77 ms in the train step at batch 32 -- a workgroup holds 128 KB
// of LDS and 256 registers per lane, so nothing of the chains beside it shares its CU, and one wavefront per SIMD leaves the patch
// staging and the cross-lane sums (2 x the MFMA time) uncovered (profiles/NOTEBOOK_r04.md)
static bool t4_on() {
    const char* e = getenv("SSC_TR4_MFMA");
    return e != nullptr && e[0] == '1';
}

extern "C" int ssc_conv_tr4_mfma_supported(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    if (d.nphase != 4 || d.TH != 2 || d.TW != 2 || d.KH != 4 || d.KW != 4 || d.bmode != 1 || d.out_stride != 2 || d.in_stride != 1 ||
        d.ky0 != 0 || d.kx0 != 0 || d.kstep != -2 || d.ioff_y != 0 || d.ioff_x != 0 || d.ooff_y != 0 || d.ooff_x != 0)
        return 0;
    if (d.x.C0 + d.x.C1 != T4_C || (d.x.C0 & 3) != 0 || (d.x.C1 & 3) != 0 || d.k_real > T4_C || d.wC1 < d.k_real) return 0;
    if (d.Nn < 1 || d.Nn > 4 || d.Nstore > 4 || d.Nstore < d.Nn || d.n_off + d.Nn > d.wC0 || d.accumulate || d.epi > 2) return 0;
    if (d.x.H != d.PH || d.x.W != d.PW || d.OH != 2 * d.PH || d.OW != 2 * d.PW) return 0;
    if ((reinterpret_cast<uintptr_t>(d.x.s0) & 15) != 0 || (d.x.C1 > 0 && (reinterpret_cast<uintptr_t>(d.x.s1) & 15) != 0)) return 0;
    if ((d.x.ab0 != nullptr && (reinterpret_cast<uintptr_t>(d.x.ab0) & 15) != 0) ||
        (d.x.ab1 != nullptr && (reinterpret_cast<uintptr_t>(d.x.ab1) & 15) != 0))
        return 0;
    if (d.x.act != SSC_ACT_NONE && d.x.act != SSC_ACT_RELU && d.x.act != SSC_ACT_LRELU) return 0;
    if (d.x.act1 > SSC_ACT_LRELU) return 0;
    if (d.stat_partial != nullptr || d.sb_x != nullptr || d.sb2_x != nullptr || d.fin_cnt != nullptr || d.stat_mode != 0) return 0;
    const long tiles = (long)d.NB * ((d.PH + T4_TY - 1) / T4_TY) * ((d.PW + T4_TX - 1) / T4_TX);
    if (tiles < 256 || tiles >= 0x7fffffffL) return 0;     // fewer tiles than CUs: the vector-ALU kernel's finer grid
    return t4_on() ? 1 : 0;
}

int ssc_conv_tr4_mfma_forward(const ssc_conv_desc* dp, void* stream) {
    if (!ssc_conv_tr4_mfma_supported(dp)) return -1;
    const ssc_conv_desc& d = *dp;
    const int tiles_x = (d.PW + T4_TX - 1) / T4_TX, tiles_y = (d.PH + T4_TY - 1) / T4_TY;
    const int tiles = d.NB * tiles_y * tiles_x;
    const int ncu = ssc_num_cu();
    const int G = tiles < ncu ? tiles : ncu;
    const size_t lds = (size_t)(2 * T4_PSZ + 4 * 256) * sizeof(float);
    static unsigned long long attr_done = 0;
    const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&tr4_mfma_kernel), (int)lds, &attr_done);
    if (arc != 0) return arc;
    hipLaunchKernelGGL(tr4_mfma_kernel, dim3(G), dim3(256), lds, (hipStream_t)stream, d, tiles, tiles_x, tiles_y);
    return (int)hipGetLastError();
}
