// fewchan.hip -- 4x4 stride-2 convolutions over an input with very few channels (4 or 8, of which 3 or 6 real): the
// generator's first conv (models_collection.py:454-458), the discriminator's first conv (:798-801) and the data gradient of
// the generator's last transposed conv w.r.t. its inputs (the conv form of conv2d_transpose's gradient, :529-534).
//
// As implicit GEMMs they are M = N*96*96 rows by 64 columns with K = 16 taps * C = 64 or 128: two to four K-tiles per output
// tile, so on the general tile kernel a launch is almost only prologue and epilogue (the non-uniform-tap form, ~385 vector
// ALU instructions per K-tile; measured 33-43 TFLOP/s, 5.3 % of the step's vector-ALU time for 1.3 % of its matrix work).
// Here:
//   * the whole filter slice lives in REGISTERS for the life of the workgroup (K/2 values per lane and 32-column block),
//     loaded once; workgroups are persistent and walk output tiles;
//   * an output tile is 4 output rows x 32 columns; its (10 x 66)-pixel input patch is staged once in LDS (pixel stride C+1
//     floats: the stride-2 reads of the 32 lanes of an MFMA row fall in 32 different banks) and every MFMA A operand is one
//     ds_read_b32 with an immediate offset -- the im2col happens in the LDS address, no per-tap decode;
//   * the next tile's patch is in flight (registers) while the current one is multiplied; two LDS images, one barrier per tile;
//   * accumulators leave as 128-byte row segments straight from the MFMA layout.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FC_TR 4          // output rows per tile
#define FC_TC 32         // output columns per tile
#define FC_PR (2 * FC_TR + 2)
#define FC_PC (2 * FC_TC + 2)

template <int C>
__global__ __launch_bounds__(256) void fewchan_conv_kernel(const ssc_conv_desc d, int tiles, int tiles_x, int tiles_y) {
    constexpr int K = 16 * C, KS = K / 2;       // GEMM depth, MFMA steps (2 k per v_mfma_f32_32x32x2_f32)
    constexpr int CP = C + 1;                   // floats per patch pixel in LDS
    constexpr int PSZ = FC_PR * FC_PC * CP;
    constexpr int C4 = C / 4;                   // 16-byte chunks per pixel
    constexpr int NQ = (FC_PR * FC_PC * C4 + 255) / 256;
    __shared__ float patch[2][PSZ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wp = wave >> 1, wj = wave & 1;    // this wave: output rows 2*wp, 2*wp+1 of the tile, columns [32*wj, 32*wj+32)

    // ---- filter fragments: B[k][n] for k = 2s + lhi, n = 32*wj + l31 ----
    float bf[KS];
    {
        const int n = wj * 32 + l31;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 2 * s + lhi;
            const int tap = k / C, c = k - tap * C;
            const bool v = (c < d.k_real) & (n < d.Nn);
            const float w = d.w[v ? ((long)tap * d.wC0 + c) * d.wC1 + d.n_off + n : 0];
            bf[s] = v ? w : 0.f;
        }
    }

    // ---- patch staging ----
    float4 rv[NQ];
    auto load_patch = [&](int tile) {
        const int tx = tile % tiles_x;
        const int r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const int iy0 = 2 * FC_TR * ty - 1, ix0 = 2 * FC_TC * tx - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = tid + 256 * q;
            const int pix = idx / C4, ch = idx - pix * C4;
            const int pr = pix / FC_PC, pc = pix - pr * FC_PC;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = (idx < FC_PR * FC_PC * C4) & ((unsigned)iy < (unsigned)d.x.H) & ((unsigned)ix < (unsigned)d.x.W);
            const float4 v = *reinterpret_cast<const float4*>(d.x.s0 + (ok ? (((long)n * d.x.H + iy) * d.x.W + ix) * C + 4 * ch : 0));
            rv[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_patch = [&](float* P) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = tid + 256 * q;
            if (idx < FC_PR * FC_PC * C4) {
                const int pix = idx / C4, ch = idx - pix * C4;
                float* p = P + pix * CP + 4 * ch;
                p[0] = rv[q].x; p[1] = rv[q].y; p[2] = rv[q].z; p[3] = rv[q].w;
            }
        }
    };

    const int G = gridDim.x;
    int tile = blockIdx.x;
    if (tile < tiles) {
        load_patch(tile);
        store_patch(patch[0]);
    }
    __syncthreads();
    int buf = 0;
    // lane bases of the two output rows of this wave: pixel (2*(2*wp+r), 2*l31) of the patch, + lhi (k parity = channel parity)
    const int ab0 = ((2 * (2 * wp)) * FC_PC + 2 * l31) * CP + lhi;
    const int ab1 = ((2 * (2 * wp + 1)) * FC_PC + 2 * l31) * CP + lhi;
    for (; tile < tiles; tile += G) {
        const int next = tile + G;
        if (next < tiles) load_patch(next);          // in flight across the MFMAs below
        const float* P = patch[buf];
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        // (issuing the ds_reads of a group of k-steps ahead of the previous group's MFMAs by hand measured 5 % slower than the
        // compiler's order: two reads, wait, four MFMAs -- two waves per SIMD already cover that latency)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = 2 * s;
            const int tap = k0 / C, c0 = k0 - tap * C;
            const int off = ((tap >> 2) * FC_PC + (tap & 3)) * CP + c0;      // compile-time after unrolling
            const float a0 = P[ab0 + off], a1 = P[ab1 + off];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf[s], acc1, 0, 0, 0);
        }
        // ---- epilogue: acc[r] is row (r & 3) + 8 * (r >> 2) + 4 * lhi (= output x inside the tile), column l31 ----
        {
            const int tx = tile % tiles_x;
            const int rr = tile / tiles_x;
            const int ty = rr % tiles_y, n = rr / tiles_y;
            const int col = wj * 32 + l31;
            if (col < d.Nstore) {
                const int oy = FC_TR * ty + 2 * wp;
                float* o0 = d.out + (((long)n * d.OH + oy) * d.OW + FC_TC * tx + 4 * lhi) * d.ldc + col;
                float* o1 = o0 + (long)d.OW * d.ldc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int x = (r & 3) + 8 * (r >> 2);
                    o0[(long)x * d.ldc] = acc0[r];
                    o1[(long)x * d.ldc] = acc1[r];
                }
            }
        }
        if (next < tiles) store_patch(patch[buf ^ 1]);
        __syncthreads();        // one barrier per tile: the image written above was last read before the previous barrier
        buf ^= 1;
    }
}

extern "C" int ssc_conv_fewchan_supported(const ssc_conv_desc* dp) {
    static const bool on = ssc_dev_getenv("SSC_FEWCHAN") == nullptr || atoi(ssc_dev_getenv("SSC_FEWCHAN")) != 0;
    const ssc_conv_desc& d = *dp;
    if (!on) return 0;
    if (d.x.C1 != 0 || (d.x.C0 != 4 && d.x.C0 != 8) || d.x.ab0 != nullptr || d.x.act != SSC_ACT_NONE) return 0;
    if (d.nphase != 1 || d.TH != 4 || d.TW != 4 || d.KH != 4 || d.KW != 4 || d.in_stride != 2 || d.ioff_y != -1 || d.ioff_x != -1 ||
        d.ky0 != 0 || d.kx0 != 0 || d.kstep != 1 || d.bmode != 0)
        return 0;
    if (d.k_real < 1 || d.k_real > d.x.C0 || d.Nn < 33 || d.Nn > 64 || d.Nstore < d.Nn || d.Nstore > 64 || d.Nstore > d.ldc) return 0;
    if (d.bias != nullptr || d.epi != 0 || d.accumulate || d.out_stride != 1 || d.ooff_y != 0 || d.ooff_x != 0) return 0;
    if (d.OH != d.PH || d.OW != d.PW || (d.PH % FC_TR) != 0 || (d.PW % FC_TC) != 0) return 0;
    if (2 * d.PH != d.x.H || 2 * d.PW != d.x.W) return 0;
    if ((reinterpret_cast<uintptr_t>(d.x.s0) & 15) != 0) return 0;
    if ((long)d.NB * (d.PH / FC_TR) * (d.PW / FC_TC) >= 0x7fffffffL) return 0;
    return 1;
}

int ssc_conv_fewchan_forward(const ssc_conv_desc* dp, int num_cu, void* stream) {
    if (!ssc_conv_fewchan_supported(dp)) return -1;
    const ssc_conv_desc& d = *dp;
    const int tiles_x = d.PW / FC_TC, tiles_y = d.PH / FC_TR;
    const int tiles = d.NB * tiles_y * tiles_x;
    // persistent workgroups: as many as are resident at once (2 per CU with 8 channels -- 48 KB of LDS and ~64 filter
    // registers each --, 3 with 4); workgroup b walks tiles b, b + G, ...: with 2304 tiles on 512 workgroups the first half
    // of the ids walks 5 and the second half 4, and the dispatcher pairs one of each on a CU (9 tiles per CU; equal counts on
    // fewer workgroups left a fifth of the CUs with one workgroup: 10)
    const int slots = num_cu * (d.x.C0 == 8 ? 2 : 3);
    const int G = tiles < slots ? tiles : slots;
    hipStream_t st = (hipStream_t)stream;
    if (d.x.C0 == 8)
        hipLaunchKernelGGL(fewchan_conv_kernel<8>, dim3(G), dim3(256), 0, st, d, tiles, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL(fewchan_conv_kernel<4>, dim3(G), dim3(256), 0, st, d, tiles, tiles_x, tiles_y);
    return (int)hipGetLastError();
}
