// c3x3.hip -- the 3x3 stride-1 convs of the bottleneck blocks at their two highest resolutions (residual_util.py:92-96, 127-131,
// 156-160: block_2, C/4 -> C/4 with C/4 = 16 or 32 channels) and their data gradients.
//
// As implicit GEMMs they are M = N*H*W rows by 16 / 32 columns with K = 9 * C = 144 / 288: a handful of K steps per tile, half
// of every 32-wide K chunk empty at 16 channels, on a 128 x 32 tile.  Measured on the tile kernel: 66-116 us (16 channels, 96^2 /
// 384^2) and 35-55 us (32 channels) for launches that move 19-75 MB -- 20-40 TFLOP/s; 26 such launches per Residual iteration
// forward and as many data gradients, 10 per Background forward.  Here, as in fewchan.hip / pw1x1.hip:
//   * the whole filter lives in REGISTERS for the life of the workgroup (K / 2 values per lane), workgroups are persistent;
//   * an output tile is 4 rows x 32 pixels (a wavefront per row); its (6 x 34)-pixel input patch is staged once in LDS with
//     the folded norm + activation applied on the way, zeros outside the image, and every MFMA A operand is one ds_read_b32
//     with an immediate offset -- the im2col happens in the LDS address;
//   * the next tile's patch is in flight (registers) while the current one is multiplied;
//   * accumulators leave as 128-byte row segments straight from the MFMA layout;
//   * the epilogue's per-column sums -- the batch statistics of the output (forward) or the two sums of the backward of the
//     norm the output is the gradient of (data gradient: ssc_conv_forward_bnbwd) -- are per-lane sums over the tiles a
//     workgroup walks: one row of partials per workgroup.
// Both filter orientations (KN forward, NK flipped for the data gradient) through the fragment load only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// tile: 4 wavefronts x RPW rows x TCW pixels, RPW * TCW = 32 (the MFMA's rows): 4 x 32, or 8 x 16 where the width is an odd
// multiple of 16 (48 x 48 at 32 channels)
template <int C, int TCW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void c3x3_kernel(const ssc_conv_desc d, int tiles, int tiles_x, int tiles_y,
                                                   float* __restrict__ stat) {
    constexpr int K = 9 * C, KS = K / 2;
    constexpr int RPW = 32 / TCW, C3_TR = 4 * RPW, C3_TC = TCW, C3_PR = C3_TR + 2, C3_PC = C3_TC + 2;
    constexpr int CP = C + 1;                   // floats per patch pixel in LDS
    constexpr int PSZ = C3_PR * C3_PC * CP;
    constexpr int C4 = C / 4;
    constexpr int NQ = (C3_PR * C3_PC * C4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float c3_smem[];
    float* const img0 = c3_smem;
    float* const img1 = c3_smem + PSZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int col = l31;
    const bool colv = col < d.Nn;
    const int H = d.x.H, W = d.x.W;

    // ---- filter fragments: B[k][n] for k = 2 s + lhi = (tap, c), n = col.  The filter passes through LDS once (the patch images'
    // space): read from memory along its contiguous axis in either orientation, picked up as [k][n] rows ----
    float bf[KS];
    {
        constexpr int FLD = 33;
        float* const F = c3_smem;                   // [K][33] <= 2 * PSZ floats
        const int Nn = d.Nn;
        constexpr int NF = K * C / 256;             // filter elements per thread: all loads issued, then all LDS writes
        float fv[NF];
        int fo[NF];
        const int kn = K * Nn;
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            const int idx = tid + 256 * q;
            int tap, c, n;
            if (d.bmode == 0) {                     // KN: w[ky][kx][c][n_off + n], n contiguous
                const int k = idx / Nn;
                n = idx - k * Nn;
                tap = k / C;
                c = k - tap * C;
            } else {                                // NK: w[ky][kx][n_off + n][c], c contiguous
                c = idx % C;
                const int r = idx / C;
                tap = r / Nn;
                n = r - tap * Nn;
            }
            const int ky = d.ky0 + (tap / 3) * d.kstep, kx = d.kx0 + (tap % 3) * d.kstep;
            const long g = d.bmode == 0 ? ((long)(ky * 3 + kx) * d.wC0 + c) * d.wC1 + d.n_off + n
                                        : ((long)(ky * 3 + kx) * d.wC0 + d.n_off + n) * d.wC1 + c;
            fo[q] = idx < kn ? (tap * C + c) * FLD + n : -1;
            fv[q] = idx < kn ? d.w[g] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NF; ++q)
            if (fo[q] >= 0) F[fo[q]] = fv[q];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KS; ++s) bf[s] = colv ? F[(2 * s + lhi) * FLD + col] : 0.f;
        __syncthreads();
    }

    // ---- patch staging: thread -> (pixel (tid + 256 q) / C4, chunk tid % C4): the chunk is the same for every q ----
    const int ch = tid % C4;
    float4 ta = make_float4(1.f, 1.f, 1.f, 1.f), tb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.x.ab0 != nullptr) {
        ta = *reinterpret_cast<const float4*>(d.x.ab0 + 4 * ch);
        tb = *reinterpret_cast<const float4*>(d.x.ab0 + C + 4 * ch);
    }
    const float slope = d.x.act == SSC_ACT_RELU ? 0.f : (d.x.act == SSC_ACT_LRELU ? 0.2f : 1.f);
    float4 rv[NQ];
    auto load_patch = [&](int tile) {
        const int tx = tile % tiles_x;
        const int r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const int iy0 = C3_TR * ty - 1, ix0 = C3_TC * tx - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = tid + 256 * q;
            const int pix = idx / C4;
            const int pr = pix / C3_PC, pc = pix - pr * C3_PC;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = (pix < C3_PR * C3_PC) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            const float4 v = *reinterpret_cast<const float4*>(d.x.s0 + (ok ? (((long)n * H + iy) * W + ix) * C + 4 * ch : 0));
            float4 t;
            t.x = fmaf(ta.x, v.x, tb.x); t.y = fmaf(ta.y, v.y, tb.y); t.z = fmaf(ta.z, v.z, tb.z); t.w = fmaf(ta.w, v.w, tb.w);
            t.x = fmaxf(t.x, slope * t.x); t.y = fmaxf(t.y, slope * t.y); t.z = fmaxf(t.z, slope * t.z); t.w = fmaxf(t.w, slope * t.w);
            rv[q] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);       // zero padding of the ACTIVATED tensor
        }
    };
    auto store_patch = [&](float* P) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = tid + 256 * q;
            const int pix = idx / C4;
            if (pix < C3_PR * C3_PC) {
                float* p = P + pix * CP + 4 * ch;
                p[0] = rv[q].x; p[1] = rv[q].y; p[2] = rv[q].z; p[3] = rv[q].w;
            }
        }
    };

    // epilogue sums of this lane's column
    const bool bwd = stat != nullptr && d.sb_x != nullptr;
    float sa = 1.f, sb = 0.f, smu = 0.f, srs = 1.f, sneg = 1.f;
    if (bwd && colv) {
        sa = d.sb_ab[col]; sb = d.sb_ab[d.Nstore + col];
        smu = d.sb_stats[col]; srs = d.sb_stats[d.Nstore + col];
        sneg = d.sb_act == SSC_ACT_RELU ? 0.f : (d.sb_act == SSC_ACT_LRELU ? 0.2f : 1.f);
    }
    float ssum = 0.f, ssq = 0.f;

    const int G = gridDim.x;
    int tile = blockIdx.x;
    if (tile < tiles) {
        load_patch(tile);
        store_patch(img0);
    }
    __syncthreads();
    int buf = 0;
    const int abase = ((wave * RPW + l31 / TCW) * C3_PC + l31 % TCW) * CP + lhi;     // this lane's pixel of the patch, + k parity
    for (; tile < tiles; tile += G) {
        const int next = tile + G;
        if (next < tiles) load_patch(next);          // in flight across the MFMAs below
        const float* P = (buf ? img1 : img0) + abase;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = 2 * s;
            const int tap = k0 / C, c0 = k0 - tap * C;
            const int off = ((tap / 3) * C3_PC + (tap % 3)) * CP + c0;      // compile-time after unrolling
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(P[off], bf[s], acc, 0, 0, 0);
        }
        // ---- epilogue: acc[r] is MFMA row m = (r & 3) + 8 * (r >> 2) + 4 * lhi = pixel (m / TCW, m % TCW) of the wave's rows, column
        // l31; 4 * lhi never crosses a multiple of TCW, so the row of the pixel is known at compile time ----
        {
            const int tx = tile % tiles_x;
            const int rr = tile / tiles_x;
            const int ty = rr % tiles_y, n = rr / tiles_y;
            const int oy = C3_TR * ty + wave * RPW;
            if (colv & (oy < H)) {
                const long prow = ((long)n * H + oy) * W + C3_TC * tx + 4 * lhi;
                const int wleft = W - (C3_TC * tx + 4 * lhi), hleft = H - oy;
                const float* xp = d.sb_x + prow * d.sb_ldx + col;
                float* o = d.out + prow * d.ldc + col;
#pragma unroll
                for (int h = 0; h < 2; ++h) {           // two batches of 8: the loads of the normed tensor issued ahead of the stores
                    float xv[8];
                    if (bwd) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int r = 8 * h + q;
                            const int x = (r & 3) + 8 * (r >> 2);
                            const int dy = x / TCW, dx = x % TCW;
                            xv[q] = (dx < wleft) & (dy < hleft) ? xp[((long)dy * W + dx) * d.sb_ldx] : 0.f;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = 8 * h + q;
                        const int x = (r & 3) + 8 * (r >> 2);
                        const int dy = x / TCW, dx = x % TCW;
                        if ((dx < wleft) & (dy < hleft)) {
                            const float v = acc[r];
                            o[((long)dy * W + dx) * d.ldc] = v;
                            if (bwd) {
                                const float dz = v * (fmaf(sa, xv[q], sb) > 0.f ? 1.f : sneg);
                                ssum += dz;
                                ssq += dz * (xv[q] - smu) * srs;
                            } else {
                                ssum += v;
                                ssq += v * v;
                            }
                        }
                    }
                }
            }
        }
        if (next < tiles) store_patch(buf ? img0 : img1);
        __syncthreads();        // one barrier per tile: the image written above was last read before the previous barrier
        buf ^= 1;
    }
    if (stat != nullptr) {      // one row of the two sums per workgroup: the four waves' rows and the two lane halves folded in order
        __shared__ float red[2][4][32];
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (lhi == 0) { red[0][wave][l31] = ssum; red[1][wave][l31] = ssq; }
        __syncthreads();
        if (wave == 0 && lhi == 0 && colv) {
            float* sp = stat + (long)blockIdx.x * 2 * d.Nstore;
            sp[col] = ((red[0][0][l31] + red[0][1][l31]) + red[0][2][l31]) + red[0][3][l31];
            sp[d.Nstore + col] = ((red[1][0][l31] + red[1][1][l31]) + red[1][2][l31]) + red[1][3][l31];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 16 channels on the 16-column MFMA (v_mfma_f32_16x16x4_f32): the 32-column instruction above multiplies padding in half of its
// columns when the conv has 16 outputs.  Same structure (4 x 32-pixel tiles, a wavefront per row, two LDS images, persistent
// workgroups, per-lane sums); a lane's share of the filter is 36 registers, a row is two MFMA row groups of 16 pixels, pixel stride
// 20 floats in LDS (the 16 pixels x 4 k of an operand read hit 64 distinct banks).
typedef float c3f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void c3x3n16_kernel(const ssc_conv_desc d, int tiles, int tiles_x, int tiles_y, float* __restrict__ stat) {
    constexpr int C = 16, K = 144, KS = 36, TR = 4, TC = 32, PR = TR + 2, PC = TC + 2, CP = 20;
    constexpr int PSZ = PR * PC * CP;
    constexpr int NQ = (PR * PC * 4 + 255) / 256;       // 4
    __shared__ __attribute__((aligned(16))) float simg[2 * PSZ];       // 32.6 KB
    __shared__ float red[2][4][16];
    float* const img0 = simg;
    float* const img1 = simg + PSZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const bool colv = l15 < d.Nn;
    const int H = d.x.H, W = d.x.W;

    // ---- filter fragments through LDS once: F[k][n] (k = (tap, c)), bf[s] = F[4 s + kq][l15] ----
    float bf[KS];
    {
        constexpr int FLD = 17;
        float* const F = simg;                      // [144][17] <= 2 * PSZ floats
        const int Nn = d.Nn;
        constexpr int NF = K * C / 256;             // 9
        float fv[NF];
        int fo[NF];
        const int kn = K * Nn;
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            const int idx = tid + 256 * q;
            int tap, c, n;
            if (d.bmode == 0) {                     // KN: w[ky][kx][c][n_off + n], n contiguous
                const int k = idx / Nn;
                n = idx - k * Nn;
                tap = k / C;
                c = k - tap * C;
            } else {                                // NK: w[ky][kx][n_off + n][c], c contiguous
                c = idx % C;
                const int r = idx / C;
                tap = r / Nn;
                n = r - tap * Nn;
            }
            const int ky = d.ky0 + (tap / 3) * d.kstep, kx = d.kx0 + (tap % 3) * d.kstep;
            const long g = d.bmode == 0 ? ((long)(ky * 3 + kx) * d.wC0 + c) * d.wC1 + d.n_off + n
                                        : ((long)(ky * 3 + kx) * d.wC0 + d.n_off + n) * d.wC1 + c;
            fo[q] = idx < kn ? (tap * C + c) * FLD + n : -1;
            fv[q] = idx < kn ? d.w[g] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NF; ++q)
            if (fo[q] >= 0) F[fo[q]] = fv[q];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KS; ++s) bf[s] = colv ? F[(4 * s + kq) * FLD + l15] : 0.f;
        __syncthreads();
    }

    // ---- patch staging: thread -> (pixel (tid + 256 q) / 4, chunk tid % 4) ----
    const int ch = tid & 3;
    float4 ta = make_float4(1.f, 1.f, 1.f, 1.f), tb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.x.ab0 != nullptr) {
        ta = *reinterpret_cast<const float4*>(d.x.ab0 + 4 * ch);
        tb = *reinterpret_cast<const float4*>(d.x.ab0 + C + 4 * ch);
    }
    const float slope = d.x.act == SSC_ACT_RELU ? 0.f : (d.x.act == SSC_ACT_LRELU ? 0.2f : 1.f);
    float4 rv[NQ];
    auto load_patch = [&](int tile) {
        const int tx = tile % tiles_x;
        const int r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const int iy0 = TR * ty - 1, ix0 = TC * tx - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pix = (tid >> 2) + 64 * q;
            const int pr = pix / PC, pc = pix - pr * PC;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = (pix < PR * PC) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            const float4 v = *reinterpret_cast<const float4*>(d.x.s0 + (ok ? (((long)n * H + iy) * W + ix) * C + 4 * ch : 0));
            float4 t;
            t.x = fmaf(ta.x, v.x, tb.x); t.y = fmaf(ta.y, v.y, tb.y); t.z = fmaf(ta.z, v.z, tb.z); t.w = fmaf(ta.w, v.w, tb.w);
            t.x = fmaxf(t.x, slope * t.x); t.y = fmaxf(t.y, slope * t.y); t.z = fmaxf(t.z, slope * t.z); t.w = fmaxf(t.w, slope * t.w);
            rv[q] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);       // zero padding of the ACTIVATED tensor
        }
    };
    auto store_patch = [&](float* P) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pix = (tid >> 2) + 64 * q;
            if (pix < PR * PC) *reinterpret_cast<float4*>(P + pix * CP + 4 * ch) = rv[q];
        }
    };

    const bool bwd = stat != nullptr && d.sb_x != nullptr;
    float sa = 1.f, sb = 0.f, smu = 0.f, srs = 1.f, sneg = 1.f;
    if (bwd && colv) {
        sa = d.sb_ab[l15]; sb = d.sb_ab[d.Nstore + l15];
        smu = d.sb_stats[l15]; srs = d.sb_stats[d.Nstore + l15];
        sneg = d.sb_act == SSC_ACT_RELU ? 0.f : (d.sb_act == SSC_ACT_LRELU ? 0.2f : 1.f);
    }
    float ssum = 0.f, ssq = 0.f;

    const int G = gridDim.x;
    int tile = blockIdx.x;
    if (tile < tiles) {
        load_patch(tile);
        store_patch(img0);
    }
    __syncthreads();
    int buf = 0;
    const int abase = (wave * PC + l15) * CP + kq;       // pixel (row `wave`, column l15 of a half) of the patch, + k within a step
    for (; tile < tiles; tile += G) {
        const int next = tile + G;
        if (next < tiles) load_patch(next);
        const float* P = (buf ? img1 : img0) + abase;
        c3f4 acc[2];
        acc[0] = (c3f4){0.f, 0.f, 0.f, 0.f};
        acc[1] = (c3f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int tap = s >> 2;
            const int off = ((tap / 3) * PC + (tap % 3)) * CP + 4 * (s & 3);        // compile-time after unrolling
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(P[off], bf[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(P[off + 16 * CP], bf[s], acc[1], 0, 0, 0);
        }
        // ---- epilogue: acc[h][r] is pixel 16 h + 4 kq + r of the wave's row, column l15 ----
        {
            const int tx = tile % tiles_x;
            const int rr = tile / tiles_x;
            const int ty = rr % tiles_y, n = rr / tiles_y;
            const int oy = TR * ty + wave;
            if (colv & (oy < H)) {
                const long prow = ((long)n * H + oy) * W + TC * tx + 4 * kq;
                const int wleft = W - (TC * tx + 4 * kq);
                const float* xp = d.sb_x + prow * d.sb_ldx + l15;
                float* o = d.out + prow * d.ldc + l15;
                float xv[8];
                if (bwd) {          // the loads of the normed tensor issued ahead of the stores
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int x = 16 * (q >> 2) + (q & 3);
                        xv[q] = x < wleft ? xp[(long)x * d.sb_ldx] : 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int x = 16 * (q >> 2) + (q & 3);
                    if (x < wleft) {
                        const float v = acc[q >> 2][q & 3];
                        o[(long)x * d.ldc] = v;
                        if (bwd) {
                            const float dz = v * (fmaf(sa, xv[q], sb) > 0.f ? 1.f : sneg);
                            ssum += dz;
                            ssq += dz * (xv[q] - smu) * srs;
                        } else {
                            ssum += v;
                            ssq += v * v;
                        }
                    }
                }
            }
        }
        if (next < tiles) store_patch(buf ? img0 : img1);
        __syncthreads();
        buf ^= 1;
    }
    if (stat != nullptr) {      // one row of the two sums per workgroup: the four k-quarters of a wave, then the four waves, in order
        ssum += __shfl_xor(ssum, 16, 64);
        ssq += __shfl_xor(ssq, 16, 64);
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (kq == 0) { red[0][wave][l15] = ssum; red[1][wave][l15] = ssq; }
        __syncthreads();
        if (wave == 0 && kq == 0 && colv) {
            float* sp = stat + (long)blockIdx.x * 2 * d.Nstore;
            sp[l15] = ((red[0][0][l15] + red[0][1][l15]) + red[0][2][l15]) + red[0][3][l15];
            sp[d.Nstore + l15] = ((red[1][0][l15] + red[1][1][l15]) + red[1][2][l15]) + red[1][3][l15];
        }
    }
}

static bool c3_on() {
    static int on = -1;         // SSC_C3X3=0: the tile kernel (A/B)
    if (on < 0) {
        const char* e = ssc_dev_getenv("SSC_C3X3");
        on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

extern "C" int ssc_conv_c3x3_supported(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    const int C = d.x.C0;
    if (!c3_on()) return 0;
    if (d.x.C1 != 0 || (C != 16 && C != 32) || d.k_real != C) return 0;
    if (d.nphase != 1 || d.TH != 3 || d.TW != 3 || d.KH != 3 || d.KW != 3 || d.in_stride != 1 || d.ioff_y != -1 || d.ioff_x != -1 ||
        d.out_stride != 1 || d.ooff_y != 0 || d.ooff_x != 0)
        return 0;
    if (!((d.ky0 == 0 && d.kx0 == 0 && d.kstep == 1) || (d.ky0 == 2 && d.kx0 == 2 && d.kstep == -1))) return 0;
    if (d.bias != nullptr || d.epi != 0 || d.accumulate || d.Nn < 4 || d.Nn > 32 || d.Nn != d.Nstore || d.Nstore > d.ldc) return 0;
    if ((d.bmode == 0 && (d.wC0 != C || d.n_off + d.Nn > d.wC1)) || (d.bmode == 1 && (d.wC1 != C || d.n_off + d.Nn > d.wC0))) return 0;
    if (d.x.act != SSC_ACT_NONE && d.x.act != SSC_ACT_RELU && d.x.act != SSC_ACT_LRELU) return 0;
    if (d.OH != d.PH || d.OW != d.PW || d.x.H != d.PH || d.x.W != d.PW) return 0;
    if ((reinterpret_cast<uintptr_t>(d.x.s0) & 15) != 0 || (d.x.ab0 != nullptr && (reinterpret_cast<uintptr_t>(d.x.ab0) & 15) != 0))
        return 0;
    const long M = (long)d.NB * d.PH * d.PW;
    if (M < 16384 || M >= 0x7fffffffL / 64) return 0;
    if (d.stat_mode != 0 || d.sb2_x != nullptr) return 0;
    return 1;
}

static int c3_tcw(const ssc_conv_desc& d) {          // 16: the 8 x 16 tile wastes fewer pixels than the 4 x 32 one
    const long c32 = (long)((d.PW + 31) / 32 * 32) * ((d.PH + 3) / 4 * 4);
    const long c16 = (long)((d.PW + 15) / 16 * 16) * ((d.PH + 7) / 8 * 8);
    return c16 < c32 ? 16 : 32;
}

// persistent workgroups (= rows of partial sums)
int ssc_conv_c3x3_walkers(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    const int tcw = c3_tcw(d), tr = 4 * (32 / tcw);
    const long tiles = (long)d.NB * ((d.PH + tr - 1) / tr) * ((d.PW + tcw - 1) / tcw);
    const int per_cu = d.x.C0 == 32 ? 2 : 3;        // LDS: 2 patch images per workgroup; K / 2 filter registers
    const long g = (long)ssc_num_cu() * per_cu;
    return (int)(tiles < g ? tiles : g);
}

int ssc_conv_c3x3_forward(const ssc_conv_desc* dp, float* stat, void* stream) {
    if (!ssc_conv_c3x3_supported(dp)) return -1;
    const ssc_conv_desc& d = *dp;
    const int tcw = c3_tcw(d), tr = 4 * (32 / tcw);
    const int tiles_x = (d.PW + tcw - 1) / tcw, tiles_y = (d.PH + tr - 1) / tr;
    const int tiles = d.NB * tiles_y * tiles_x;
    const int G = ssc_conv_c3x3_walkers(dp);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)2 * (tr + 2) * (tcw + 2) * (d.x.C0 + 1) * sizeof(float);
    static int n16 = -1;        // SSC_C3X3_N16=0: the 32-column instruction also at 16 outputs (A/B)
    if (n16 < 0) {
        const char* e = ssc_dev_getenv("SSC_C3X3_N16");
        n16 = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    if (d.x.C0 == 16 && d.Nn <= 16 && tcw == 32 && n16) {
        hipLaunchKernelGGL(c3x3n16_kernel, dim3(G), dim3(256), 0, st, d, tiles, tiles_x, tiles_y, stat);
    } else if (d.x.C0 == 16) {
        if (tcw == 32)
            hipLaunchKernelGGL((c3x3_kernel<16, 32>), dim3(G), dim3(256), lds, st, d, tiles, tiles_x, tiles_y, stat);
        else
            hipLaunchKernelGGL((c3x3_kernel<16, 16>), dim3(G), dim3(256), lds, st, d, tiles, tiles_x, tiles_y, stat);
    } else {
        if (tcw == 32)
            hipLaunchKernelGGL((c3x3_kernel<32, 32>), dim3(G), dim3(256), lds, st, d, tiles, tiles_x, tiles_y, stat);
        else
            hipLaunchKernelGGL((c3x3_kernel<32, 16>), dim3(G), dim3(256), lds, st, d, tiles, tiles_x, tiles_y, stat);
    }
    return (int)hipGetLastError();
}
