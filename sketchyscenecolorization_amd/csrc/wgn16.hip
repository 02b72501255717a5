// wgn16.hip -- filter gradients of the bottlenecks' convs with 16 output channels at the highest resolution: the 3x3 conv 16 -> 16
// (block_2, residual_util.py:92-96: dW [144][16]) and the 4x4 stride-1 conv 64 -> 16 (block_1 of the plain bottlenecks,
// residual_util.py:147-151: dW [1024][16]).
//
// dW = patch(x)^T dy is a GEMM with 144 / 1024 rows, 16 columns and K = every pixel of the batch (294 912 at batch 32): the general
// filter-gradient kernel's narrowest tile is 32 columns and its rows come in 128s -- measured 84 us (16 TFLOP/s) and 220 us
// (44 TFLOP/s).  Here v_mfma_f32_16x16x4_f32 takes 16 filter rows x 16 output channels x 4 pixels per instruction:
//   * A operand = x at (pixel + tap offset), lane (channel within a block of 16, pixel of the quad); B operand = dy at the pixel, lane
//     (output channel, pixel of the quad): both one ds_read_b32 from dense [pixel][16]-strided LDS images (64 distinct banks);
//   * 3x3, 16 channels: a wavefront owns a row of the 4 x 32-pixel tile and all 9 taps (36 accumulator registers);
//     4x4, 64 channels: the four wavefronts own one filter ROW each (4 taps x 4 channel blocks: 64 accumulator registers) and walk
//     the same pixels;
//   * persistent workgroups accumulate over all the tiles they walk and write one slab per wavefront / workgroup at the end; a second
//     launch adds the slabs in a fixed order (+ the accumulate of the second discriminator pass).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 wg16_xform(const float4& v, const float4& a, const float4& b, float slope, bool ok) {
    float4 t;
    t.x = fmaf(a.x, v.x, b.x); t.y = fmaf(a.y, v.y, b.y); t.z = fmaf(a.z, v.z, b.z); t.w = fmaf(a.w, v.w, b.w);
    t.x = fmaxf(t.x, slope * t.x); t.y = fmaxf(t.y, slope * t.y); t.z = fmaxf(t.z, slope * t.z); t.w = fmaxf(t.w, slope * t.w);
    return ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------- 3x3, 16 gathered channels: tile 4 rows x 32 pixels, a wave per row
__global__ __launch_bounds__(256) void wgn16_k3_kernel(const ssc_wgrad_desc d, int tiles, int tiles_x, int tiles_y, float* __restrict__ slabs) {
    constexpr int TR = 4, TC = 32, PR = TR + 2, PC = TC + 2, C = 16;
    __shared__ __attribute__((aligned(16))) float xs[PR * PC * C];      // 13.1 KB
    __shared__ __attribute__((aligned(16))) float ys[TR * TC * 16];     // 8 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int H = d.g.H, W = d.g.W;
    const int c4 = (tid & 3) * 4;       // staging: 4 chunks of 16 B per pixel of either image
    float4 ta = make_float4(1.f, 1.f, 1.f, 1.f), tb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.g.ab0 != nullptr) {
        ta = *reinterpret_cast<const float4*>(d.g.ab0 + c4);
        tb = *reinterpret_cast<const float4*>(d.g.ab0 + C + c4);
    }
    const float slope = d.g.act == SSC_ACT_RELU ? 0.f : (d.g.act == SSC_ACT_LRELU ? 0.2f : 1.f);
    const int ncol = d.Nn;

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NQX = (PR * PC + 63) / 64, NQY = (TR * TC) / 64;      // 4, 2
    float4 rx[NQX], ry[NQY];
    auto load_tile = [&](int tile) {
        const int tx = tile % tiles_x;
        const int r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const int y0 = TR * ty, x0 = TC * tx;
#pragma unroll
        for (int q = 0; q < NQX; ++q) {
            const int pos = (tid >> 2) + 64 * q;
            const int pr = pos / PC, pc = pos - pr * PC;
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            const bool ok = (pos < PR * PC) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            const float4 v = *reinterpret_cast<const float4*>(d.g.s0 + (ok ? (((long)n * H + iy) * W + ix) * C : 0) + c4);
            rx[q] = wg16_xform(v, ta, tb, slope, ok);
        }
#pragma unroll
        for (int q = 0; q < NQY; ++q) {
            const int pos = (tid >> 2) + 64 * q;
            const int pr = pos / TC, pc = pos - pr * TC;
            const int oy = y0 + pr, ox = x0 + pc;
            const bool ok = (oy < d.PH) & (ox < d.PW) & (c4 < ncol);
            const float4 v = *reinterpret_cast<const float4*>(d.d.s0 + (ok ? (((long)n * d.PH + oy) * d.PW + ox) * d.d.C0 : 0) + c4);
            ry[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int q = 0; q < NQX; ++q) {
            const int pos = (tid >> 2) + 64 * q;
            if (pos < PR * PC) *reinterpret_cast<float4*>(xs + pos * C + c4) = rx[q];
        }
#pragma unroll
        for (int q = 0; q < NQY; ++q) *reinterpret_cast<float4*>(ys + ((tid >> 2) + 64 * q) * 16 + c4) = ry[q];
    };

    const int G = gridDim.x;
    for (int tile = blockIdx.x; tile < tiles; tile += G) {
        load_tile(tile);
        __syncthreads();            // the previous tile's images are no longer read
        store_tile();
        __syncthreads();
        // lane (m = l15: gathered channel, kq: pixel of the quad) / (n = l15: output channel, kq)
        const float* xa = xs + (wave * PC + kq) * C + l15;
        const float* yb = ys + (wave * TC + kq) * 16 + l15;
#pragma unroll
        for (int q = 0; q < TC / 4; ++q) {
            const float b = yb[4 * q * 16];
#pragma unroll
            for (int t = 0; t < 9; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[((t / 3) * PC + 4 * q + t % 3) * C], b, acc[t], 0, 0, 0);
        }
    }
    // one slab [144][16] per wavefront: acc[t][r] is filter row 16 t + 4 kq + r (= tap t, channel 4 kq + r), column l15
    float* sl = slabs + ((long)blockIdx.x * 4 + wave) * 144 * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sl[(16 * t + 4 * kq + r) * 16 + l15] = acc[t][r];
}

// ---------------------------------------------------------------- 4x4 stride 1, 64 gathered channels: tile 4 rows x 16 pixels, a wave per filter row
__global__ __launch_bounds__(256) void wgn16_k4_kernel(const ssc_wgrad_desc d, int tiles, int tiles_x, int tiles_y, float* __restrict__ slabs) {
    constexpr int TR = 4, TC = 16, PR = TR + 3, PC = TC + 3, C = 64, PST = 80;    // 80 = 16 (mod 64): pixel k of a quad -> banks 16 k + m
    __shared__ __attribute__((aligned(16))) float xs[PR * PC * PST];    // 42.6 KB
    __shared__ __attribute__((aligned(16))) float ys[TR * TC * 16];     // 4 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int H = d.g.H, W = d.g.W;
    const int c4 = (tid & 15) * 4;      // x staging: 16 chunks of 16 B per pixel
    float4 ta = make_float4(1.f, 1.f, 1.f, 1.f), tb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.g.ab0 != nullptr) {
        ta = *reinterpret_cast<const float4*>(d.g.ab0 + c4);
        tb = *reinterpret_cast<const float4*>(d.g.ab0 + C + c4);
    }
    const float slope = d.g.act == SSC_ACT_RELU ? 0.f : (d.g.act == SSC_ACT_LRELU ? 0.2f : 1.f);
    const int ncol = d.Nn;

    f32x4 acc[4][4];        // [kx][channel block]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NQX = (PR * PC + 15) / 16;        // 9
    float4 rx[NQX], ry;
    auto load_tile = [&](int tile) {
        const int tx = tile % tiles_x;
        const int r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const int y0 = TR * ty, x0 = TC * tx;
#pragma unroll
        for (int q = 0; q < NQX; ++q) {
            const int pos = (tid >> 4) + 16 * q;
            const int pr = pos / PC, pc = pos - pr * PC;
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            const bool ok = (pos < PR * PC) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            const float4 v = *reinterpret_cast<const float4*>(d.g.s0 + (ok ? (((long)n * H + iy) * W + ix) * C : 0) + c4);
            rx[q] = wg16_xform(v, ta, tb, slope, ok);
        }
        {
            const int pos = tid >> 2, cy = (tid & 3) * 4;
            const int pr = pos / TC, pc = pos - pr * TC;
            const int oy = y0 + pr, ox = x0 + pc;
            const bool ok = (oy < d.PH) & (ox < d.PW) & (cy < ncol);
            const float4 v = *reinterpret_cast<const float4*>(d.d.s0 + (ok ? (((long)n * d.PH + oy) * d.PW + ox) * d.d.C0 : 0) + cy);
            ry = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int q = 0; q < NQX; ++q) {
            const int pos = (tid >> 4) + 16 * q;
            if (pos < PR * PC) *reinterpret_cast<float4*>(xs + pos * PST + c4) = rx[q];
        }
        *reinterpret_cast<float4*>(ys + (tid >> 2) * 16 + (tid & 3) * 4) = ry;
    };

    const int G = gridDim.x;
    for (int tile = blockIdx.x; tile < tiles; tile += G) {
        load_tile(tile);
        __syncthreads();
        store_tile();
        __syncthreads();
        // this wave: filter row ky = wave.  A: lane (m = l15: channel within its block, kq: pixel of the quad); B: (n = l15, kq)
        const float* xa = xs + (wave * PC + kq) * PST + l15;
        const float* yb = ys + kq * 16 + l15;
#pragma unroll 1
        for (int row = 0; row < TR; ++row) {
#pragma unroll
            for (int q = 0; q < TC / 4; ++q) {
                const float b = yb[(row * TC + 4 * q) * 16];
#pragma unroll
                for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[kx][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[(row * PC + 4 * q + kx) * PST + 16 * cb], b, acc[kx][cb], 0, 0, 0);
            }
        }
    }
    // one slab [1024][16] per workgroup: acc[kx][cb][r] is filter row ((wave * 4 + kx) * 64 + 16 cb + 4 kq + r), column l15
    float* sl = slabs + (long)blockIdx.x * 1024 * 16;
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sl[((wave * 4 + kx) * 64 + 16 * cb + 4 * kq + r) * 16 + l15] = acc[kx][cb][r];
}

// out[r][n] (+)= sum over the slabs in a fixed order: a workgroup per filter row, thread (sg, n) adds the slabs sg, sg + 16, ...,
// the 16 partial sums of a column are folded in order; slab columns beyond Nn are zero by construction and not stored
__global__ __launch_bounds__(256) void wgn16_reduce_kernel(const float* __restrict__ slabs, int nslab, int rows, int Nn, int ldc,
                                                           float* __restrict__ out, int accumulate) {
    __shared__ float part[16][16];
    const int r = blockIdx.x, n = threadIdx.x & 15, sg = threadIdx.x >> 4;
    const long st = (long)rows * 16;
    const float* p = slabs + (long)r * 16 + n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = sg;
    for (; g + 48 < nslab; g += 64) {
        s0 += p[(long)g * st];
        s1 += p[(long)(g + 16) * st];
        s2 += p[(long)(g + 32) * st];
        s3 += p[(long)(g + 48) * st];
    }
    for (; g < nslab; g += 16) s0 += p[(long)g * st];
    part[sg][n] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sg == 0 && n < Nn) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += part[q][n];
        float* o = out + (long)r * ldc + n;
        if (accumulate) v += *o;
        *o = v;
    }
}

static bool wg16_on() {
    static int on = -1;         // SSC_WGN16=0: the general filter-gradient kernel (A/B)
    if (on < 0) {
        const char* e = ssc_dev_getenv("SSC_WGN16");
        on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return on != 0;
}

// 3: the 3x3 form, 4: the 4x4 form, 0: not this kernel's
static int wg16_form(const ssc_wgrad_desc& d) {
    if (!wg16_on()) return 0;
    if (d.g.C1 != 0 || d.d.C1 != 0 || d.in_stride != 1 || d.ioff_y != -1 || d.ioff_x != -1) return 0;
    if (d.Nn < 4 || d.Nn > 16 || (d.Nn & 3) != 0 || d.d.C0 < d.Nn || d.ldc != d.Nn) return 0;
    if (d.g.H != d.PH || d.g.W != d.PW || d.d.H != d.PH || d.d.W != d.PW) return 0;
    if (d.g.act != SSC_ACT_NONE && d.g.act != SSC_ACT_RELU && d.g.act != SSC_ACT_LRELU) return 0;
    if (d.d.ab0 != nullptr || d.d.act != SSC_ACT_NONE) return 0;
    if ((reinterpret_cast<uintptr_t>(d.g.s0) & 15) != 0 || (reinterpret_cast<uintptr_t>(d.d.s0) & 15) != 0 ||
        (d.g.ab0 != nullptr && (reinterpret_cast<uintptr_t>(d.g.ab0) & 15) != 0))
        return 0;
    const long P = (long)d.NB * d.PH * d.PW;
    if (P < 32768 || P >= 0x7fffffffL / 64) return 0;
    if (d.TH == 3 && d.TW == 3 && d.g.C0 == 16 && d.Cg_real == 16) return 3;
    if (d.TH == 4 && d.TW == 4 && d.g.C0 == 64 && d.Cg_real == 64) return 4;
    return 0;
}

extern "C" int ssc_conv_wgn16_supported(const ssc_wgrad_desc* dp) { return wg16_form(*dp) != 0 ? 1 : 0; }

int ssc_conv_wgn16(const ssc_wgrad_desc* dp, float* ws, int64_t ws_bytes, void* stream) {
    const ssc_wgrad_desc& d = *dp;
    const int form = wg16_form(d);
    if (form == 0 || ws == nullptr) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int tc = form == 3 ? 32 : 16;
    const int tiles_x = (d.PW + tc - 1) / tc, tiles_y = (d.PH + 3) / 4;
    const long tiles = (long)d.NB * tiles_y * tiles_x;
    const int rows = form == 3 ? 144 : 1024;
    const int per_wg = form == 3 ? 4 : 1;               // slabs per workgroup
    long G = (long)ssc_num_cu() * 3;
    if (G > tiles) G = tiles;
    while (G > 1 && (int64_t)G * per_wg * rows * 16 * 4 > ws_bytes) G /= 2;
    if ((int64_t)G * per_wg * rows * 16 * 4 > ws_bytes) return -2;
    if (form == 3)
        hipLaunchKernelGGL(wgn16_k3_kernel, dim3((unsigned)G), dim3(256), 0, st, d, (int)tiles, tiles_x, tiles_y, ws);
    else
        hipLaunchKernelGGL(wgn16_k4_kernel, dim3((unsigned)G), dim3(256), 0, st, d, (int)tiles, tiles_x, tiles_y, ws);
    hipLaunchKernelGGL(wgn16_reduce_kernel, dim3((unsigned)rows), dim3(256), 0, st, ws, (int)(G * per_wg), rows, d.Nn,
                       d.ldc, d.out, d.accumulate);
    return (int)hipGetLastError();
}
