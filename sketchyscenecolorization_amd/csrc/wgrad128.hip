// wgrad128.hip -- filter gradients of the large layers on a 128 x 128 accumulator tile (gfx950, fp32 MFMA).
//
//   dF[(tap, cg)][cd] = sum_pix G[pix@tap][cg] * D[pix][cd]          (graph_single.py:24-30, 309-312: compute_gradients)
//
// conv_wgrad_kernel (igemm.hip) walks 64 x 128 tiles whose 64 gathered columns may straddle taps: every thread decodes its own
// tap, tests its own bounds and clamps its own dense rows -- 105 vector-ALU instructions per 32-pixel K step against 32 MFMAs,
// and fp32 MFMA shares the vector lanes with them (SQ_VALU_MFMA_COEXEC_CYCLES = 0): the kernel ran at 0.66 of the matrix peak
// where the forward kernel reaches 0.73.  The output of a filter gradient is small and its K (pixels) long, so nothing forces
// small tiles: K is split over workgroups instead.  This kernel therefore
//   * owns a 128 x 128 tile per workgroup, 64 x 64 per wave (4 accumulators, 64 AGPRs): one ds_read_b64 per operand feeds two
//     MFMAs on each side -- 0.5 LDS instructions and ~0.6 vector-ALU instructions per MFMA instead of 0.8 and 3.3;
//   * takes only tiles whose 128 gathered columns lie inside ONE tap and one source (Cg % 128 == 0), or exactly two taps of a
//     64-channel tensor: the tap is a workgroup constant, so "which pixel, is it inside the image" is a per-PIXEL fact, decoded
//     once per 256 pixels by the whole workgroup into an LDS table {byte offset | out-of-range, 1.0 | 0.0};
//   * loads through buffer descriptors: an out-of-image tap, a pixel beyond the last one and a column beyond the tensor are
//     offsets outside the descriptor's range and come back as zeros -- no masks, no clamps, no per-load compares; the dense
//     tile's K advance is a scalar add to the descriptor's base;
//   * moves a dense side without folded norm / activation (every dy) global -> LDS by DMA (buffer_load ... lds).
// Rows of the 32 x 32 MFMA blocks are interleaved (tile row = 2 * lane row + block) so that the two blocks of a wave read one
// 8-byte LDS word; the epilogue undoes it.  Split-K over the pixels: slabs + the deterministic reduce of igemm.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "igemm_util.h"
#include "host_util.h"
#include <type_traits>

#ifndef SSC_WG128_BK
#define SSC_WG128_BK 32    // pixels per K step: 32 (68 KB of LDS, two workgroups per CU) or 16 (36 KB)
#endif
#define BK SSC_WG128_BK
#define NP (BK / 8)      // 16-byte pieces of a staged tile per thread
#define KPB (256 / BK)   // K steps per block of the pixel table
#define NBB 2           // LDS buffers of the dense side
#define TB 128          // tile edge (both sides)

typedef float f32x2v __attribute__((ext_vector_type(2)));

#define SSC_RSRC_FLAGS 0x00020000

#ifdef SSC_WG128_TIMING
// diagnostic build (scripts/wg128_timing.sh): cycle-counter stamps of one wave per workgroup at the phase borders of every K
// step, summed into g_wg128_t: [0] barrier release -> first operands in registers, [1] -> last MFMA issued, [2] -> counted wait
// passed, [3] -> barrier passed, [4] K steps counted, [5] whole kernel (wave 0 of every workgroup).  The stamps perturb the
// kernel (read the SHARES, not the cycles).  Finding: with the dense tile's DMA in the second quarter of the step a wave sat
// 15 % (one workgroup per CU) to 36 % (two) of its time in the counted wait in front of the barrier -- a 16 KB LDS-DMA fill
// takes ~1.1 us from issue to landed (MI355X_MICROARCH.md, ldsdma-fill); issuing the DMA first (every load of the loop by
// inline asm under a hand count, because hipcc's own vmcnt does not know the DMA) moved the wait to the register loads of the
// gathered tile and the kernel's rate did not change (113.7 vs 115.9 TFLOP/s): the step is bound by the latency of its
// ~32 KB of loads per workgroup against ~1.8 us of MFMA work, not by their placement.
__device__ unsigned long long g_wg128_t[8];
#define TSTAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define TSTAMP(v)
#endif

// 16 bytes per lane through a buffer descriptor: an offset at or beyond its num_records comes back as zeros.  (The builtin's
// result is taken with `auto`: assigned to an ext_vector_type it is converted as a SCALAR -- a splat of the first dword.)
__device__ __forceinline__ float4 bload16(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    static_assert(sizeof(v) == 16, "four dwords");
    return __builtin_bit_cast(float4, v);
}

// act(a*v+b) without a mask (dense side: rows beyond the last pixel meet zeros on the gathered side)
__device__ __forceinline__ float4 xform4_nomask(float4 v, const float4& a, const float4& b, float slope) {
    float t;        // scalar on purpose: no packed fp32 math beside fp32 MFMAs (igemm_util.h)
    t = fmaf(a.x, v.x, b.x); v.x = fmaxf(t, t * slope);
    t = fmaf(a.y, v.y, b.y); v.y = fmaxf(t, t * slope);
    t = fmaf(a.z, v.z, b.z); v.z = fmaxf(t, t * slope);
    t = fmaf(a.w, v.w, b.w); v.w = fmaxf(t, t * slope);
    return v;
}

// GPLAIN: the gathered side has no folded norm / activation.  DMODE: dense side 0 = plain by LDS-DMA, 1 = plain through
// registers (A/B), 2 = folded norm + activation through registers.  TPT: taps per tile (2: a 64-channel gathered tensor).
template <bool GPLAIN, int DMODE, int TPT>
__global__ __launch_bounds__(256, BK == 32 ? 2 : 3) void conv_wgrad128_kernel(const ssc_wgrad_desc d, const Magics mg,
                                                                float* __restrict__ slab_base, long slab_stride, int splitk,
                                                                int xcd) {
    constexpr int T_SZ = BK * TB;          // floats per operand tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][BK][TB]  gathered side, [pixel][column]
    float* Bs = smem + 2 * T_SZ;           // [NBB][BK][TB]  dense side
    int2* ptab = reinterpret_cast<int2*>(smem + (2 + NBB) * T_SZ);      // [2][TPT][256]: {byte offset of pixel@tap | 0x80000000, 1.0f | 0}

    TSTAMP(t_kernel0);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    const int gC0 = d.g.C0, gC1 = d.g.C1, Cg = gC0 + gC1;
    const int ntap = d.TH * d.TW;
    const int Mtot = ntap * Cg;
    const unsigned P = (unsigned)((long)d.NB * d.PH * d.PW);
    const int PHW = d.PH * d.PW;
    // workgroup -> (row tile, column tile, K slice).  All tiles of a K slice read the same pixels; dealt round robin to the 8
    // XCDs (id % 8) each L2 fetches every slice.  xcd (host flag, grid a multiple of 8): ids with the same id % 8 walk a
    // contiguous run of (row tile, column tile, slice) order -- whole slices per XCD.
    // (1-D grid: row tile fastest, column tile, slice)
    int mt_i, nt_i, ks;
    {
        const unsigned gx = (unsigned)((Mtot + TB - 1) / TB), gy = (unsigned)((d.Nn + TB - 1) / TB);
        const unsigned per_slice = gx * gy, total = per_slice * (unsigned)splitk;
        const unsigned lin = blockIdx.x;
        const unsigned t2 = xcd ? (lin & 7u) * (total >> 3) + (lin >> 3) : lin;
        ks = (int)(t2 / per_slice);
        const unsigned r = t2 - (unsigned)ks * per_slice;
        nt_i = (int)(r / gx);
        mt_i = (int)(r - (unsigned)nt_i * gx);
    }
    const int m0 = mt_i * TB, n0 = nt_i * TB;

    // ---- gathered side: the tile's tap(s), source, descriptor ----
    int tap0, c0;
    if (TPT == 1) {
        tap0 = div32(m0, mg.mC, mg.oneC);
        c0 = m0 - tap0 * Cg;
    } else {
        tap0 = mt_i * 2;
        c0 = 0;
    }
    const bool g_first = c0 < gC0;
    const int g_cs = g_first ? gC0 : gC1;
    const int g_coff = g_first ? c0 : c0 - gC0;
    const float* const g_base = (g_first ? d.g.s0 : d.g.s1) + g_coff;
    const int g_bytes = d.NB * d.g.H * d.g.W * g_cs * 4 - g_coff * 4;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g_base, 0, g_bytes, SSC_RSRC_FLAGS);
    int ty[TPT], tx[TPT];
    bool tapv[TPT];
#pragma unroll
    for (int t = 0; t < TPT; ++t) {
        const int tp = tap0 + t;
        tapv[t] = tp < ntap;
        ty[t] = div32(tapv[t] ? tp : 0, mg.mTW, mg.oneTW);
        tx[t] = (tapv[t] ? tp : 0) - ty[t] * d.TW;
    }

    // per-thread piece of a staged tile: rows a_r + 8 s (s = 0..3), 16-byte column a_q
    const int a_q = tid & 31, a_r = tid >> 5;
    const int a_tsel = (TPT == 2) ? (a_q >> 4) : 0;
    const int a_cq = (TPT == 2) ? (a_q & 15) : a_q;     // 16-byte column inside the tap's channels
    const unsigned a_cb = (unsigned)a_cq * 16u;
    float4 aa = make_float4(1.f, 1.f, 1.f, 1.f), ab = make_float4(0.f, 0.f, 0.f, 0.f);
    float g_slope = 1.f;
    if (!GPLAIN) {
        gview_affine4(d.g, c0 + a_cq * 4, aa, ab);
        g_slope = act_slope((!g_first && d.g.act1 >= 0) ? d.g.act1 : d.g.act);
    }

    // ---- dense side ----
    const int dC0 = d.d.C0, dC1 = d.d.C1;
    const bool d_first = n0 < dC0;
    const int d_cs = d_first ? dC0 : dC1;
    const int d_coff = d_first ? n0 : n0 - dC0;
    const float* const d_base = (d_first ? d.d.s0 : d.d.s1) + d_coff;
    const unsigned d_rowb = (unsigned)d_cs * 4u;
    const int d_bytes = (int)(P * d_rowb) - d_coff * 4;
    const unsigned d_step = (unsigned)BK * d_rowb;      // bytes per K-tile
    // a column beyond the source's channels: an offset no descriptor holds (the lane's 16 bytes are zeros)
    const bool b_cv = d_coff + a_q * 4 < d_cs;
    float4 ba = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
    float d_slope = 1.f;
    if (DMODE == 2) {
        gview_affine4(d.d, b_cv ? n0 + a_q * 4 : 0, ba, bb);
        d_slope = act_slope((!d_first && d.d.act1 >= 0) ? d.d.act1 : d.d.act);
    }
    unsigned b_voff[NP];        // registers: row a_r + 8 s; DMA: instruction wave * NP + s covers rows 2 * ins + (lane >> 5)
#pragma unroll
    for (int s = 0; s < NP; ++s) {
        const int row = (DMODE == 0) ? (wave * NP + s) * 2 + lhi : a_r + 8 * s;
        b_voff[s] = b_cv ? (unsigned)row * d_rowb + (unsigned)a_q * 16u : 0x80000000u;
    }

    // ---- K range of this slice ----
    const int nkt = (int)((P + BK - 1) / BK);
    const int per = (nkt + splitk - 1) / splitk;
    const int kt_begin = ks * per;
    const int nk = min(nkt, kt_begin + per) - kt_begin;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- the pixel table: block b = pixels [(kt_begin + 8 b) * 32, + 256), one pixel per thread ----
    const int gH = d.g.H, gW = d.g.W;
    auto fill_ptab = [&](int blk) {
        const unsigned p = (unsigned)(kt_begin + blk * KPB) * BK + (unsigned)tid;
        const bool pv = p < P;
        const unsigned pp = pv ? p : 0u;
        int n, rem, py;
        if (mg.use32) {
            n = (int)__umulhi(pp, mg.mPHPW32) + (int)(pp & (unsigned)mg.onePHPW);
            rem = (int)pp - n * PHW;
            py = (int)__umulhi((unsigned)rem, mg.mPW32) + (int)((unsigned)rem & (unsigned)mg.onePW);
        } else {
            n = (int)div64((long)pp, mg.mPHPW, mg.onePHPW);
            rem = (int)pp - n * PHW;
            py = (int)div64((long)rem, mg.mPW, mg.onePW);
        }
        const int px = rem - py * d.PW;
        const int iy0 = py * d.in_stride + d.ioff_y, ix0 = px * d.in_stride + d.ioff_x;
#pragma unroll
        for (int t = 0; t < TPT; ++t) {
            const int iy = iy0 + ty[t], ix = ix0 + tx[t];
            const bool v = pv & tapv[t] & ((unsigned)iy < (unsigned)gH) & ((unsigned)ix < (unsigned)gW);
            const int off = ((n * gH + iy) * gW + ix) * (g_cs * 4);
            ptab[((blk & 1) * TPT + t) * 256 + tid] = make_int2(v ? off : (int)0x80000000, v ? 0x3f800000 : 0);
        }
    };

    float4 ra[NP], rb[NP];
    float ram[NP];
    // The staging work of a K step is cut into PIECES that the main loop places by hand between the MFMA groups (one piece per
    // group of four MFMAs, scheduling barriers in between): piece s of the gathered / dense tile = rows a_r + 8 s.
    int2 pe[NP];        // table entries of the pieces about to be loaded (read a few MFMA groups ahead of the loads)
    auto ptab_piece = [&](int j, int s) {
        pe[s] = ptab[(((j / KPB) & 1) * TPT + a_tsel) * 256 + (j % KPB) * BK + a_r + 8 * s];
    };
    auto load_a_piece = [&](int j, int s) {         // K-tile j (relative) of the gathered side -> registers
        ra[s] = bload16(rsA, (unsigned)pe[s].x + a_cb);
        ram[s] = __builtin_bit_cast(float, pe[s].y);
    };
    auto rsrc_b = [&](int j) {          // the dense tensor from K-tile j on (scalar arithmetic only)
        const unsigned off = (unsigned)(kt_begin + j) * d_step;
        const int left = d_bytes - (int)off;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(d_base) + off), 0, left > 0 ? left : 0,
                                                 SSC_RSRC_FLAGS);
    };
    auto load_b_piece = [&](int j, int s) {
        if (DMODE != 0) rb[s] = bload16(rsrc_b(j), b_voff[s]);
    };
    auto dma_b_piece = [&](int j, int buf, int s) {
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)((2 * T_SZ + buf * T_SZ + wave * NP * 256) * 4));
        glds16_buf(rsrc_b(j), b_voff[s], 0u, dst + s * 1024);
    };
    auto stage_a_piece = [&](int buf, int s) {
        *reinterpret_cast<float4*>(As + buf * T_SZ + (a_r + 8 * s) * TB + a_q * 4) =
            GPLAIN ? ra[s] : xform4(ra[s], aa, ab, g_slope, ram[s]);
    };
    auto stage_b_piece = [&](int buf, int s) {
        if (DMODE != 0)
            *reinterpret_cast<float4*>(Bs + buf * T_SZ + (a_r + 8 * s) * TB + a_q * 4) =
                (DMODE == 2) ? xform4_nomask(rb[s], ba, bb, d_slope) : rb[s];
    };

    if (nk > 0) {
        fill_ptab(0);
        if (DMODE == 0) {
#pragma unroll
            for (int s = 0; s < NP; ++s) dma_b_piece(0, 0, s);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NP; ++s) { ptab_piece(0, s); load_a_piece(0, s); load_b_piece(0, s); }
#pragma unroll
        for (int s = 0; s < NP; ++s) { stage_a_piece(0, s); stage_b_piece(0, s); }
#pragma unroll
        for (int s = 0; s < NP; ++s) { ptab_piece(1, s); load_a_piece(1, s); load_b_piece(1, s); }
        if (DMODE == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP) : "memory");       // the first dense tile has landed
        __syncthreads();
        // operand words of this lane: tile row / column 2 * l31 + {0, 1} of the wave's 64, K row 2 kk + lhi
        const float* a_rd = As + lhi * TB + wm * 64 + 2 * l31;
        const float* b_rd = Bs + lhi * TB + wn * 64 + 2 * l31;
        for (int j = 0; j < nk; ++j) {
            const int cur = j & 1;
            const float* Ab = a_rd + cur * T_SZ;
            const float* Bb = b_rd + cur * T_SZ;
            if ((j % KPB) == KPB / 2) fill_ptab(j / KPB + 1);      // wave-uniform; the table of the next 256 pixels (read from j + 2 on)
            // BK / 2 groups of four MFMAs (one per accumulator: consecutive MFMAs never share one), operands fetched PF groups
            // ahead into a ring of registers; between the groups, in this order: K-tile j + 1 registers -> LDS (first quarter of
            // the groups), its dense tile by DMA or from registers + the table entries of K-tile j + 2 (second), its loads into
            // the drained registers (second half).  Full scheduling barriers between groups: the compiler orders only inside a
            // group.  Measured on encoder_3's filter gradient (19.3 GFLOP, sustained): compiler-interleaved staging with runs
            // of MFMAs on one accumulator 108.8 TFLOP/s, this placement 115.9, all vector-ALU work in front of the groups
            // 114.7, all memory operations in front 113.0; one workgroup per CU 113.3 against 115.6 for two -- the kernel's
            // speed is the speed of ONE wave's instruction stream.
            constexpr int NG = BK / 2, PF = 4;
            f32x2v av[8], bv[8];
            TSTAMP(t_a);
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                av[q] = *reinterpret_cast<const f32x2v*>(Ab + q * 2 * TB);
                bv[q] = *reinterpret_cast<const f32x2v*>(Bb + q * 2 * TB);
            }
#ifdef SSC_WG128_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#endif
            TSTAMP(t_b);
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                if (q + PF < NG) {
                    av[(q + PF) & 7] = *reinterpret_cast<const f32x2v*>(Ab + (q + PF) * 2 * TB);
                    bv[(q + PF) & 7] = *reinterpret_cast<const f32x2v*>(Bb + (q + PF) * 2 * TB);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q & 7][i], bv[q & 7][jj], acc[i][jj], 0, 0, 0);
                if (q < NP) {
                    stage_a_piece(cur ^ 1, q);
                } else if (q < 2 * NP) {
                    if (DMODE == 0) dma_b_piece(j + 1, cur ^ 1, q - NP);
                    else stage_b_piece(cur ^ 1, q - NP);
                    ptab_piece(j + 2, q - NP);
                } else if (q < 3 * NP) {
                    load_a_piece(j + 2, q - 2 * NP);
                } else {
                    load_b_piece(j + 2, q - 3 * NP);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            TSTAMP(t_c);
            if (DMODE == 0) {       // counted wait + bare barrier: the gathered loads of K-tile j + 2 stay in flight
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NP) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            TSTAMP(t_d);
            __builtin_amdgcn_s_barrier();
#ifdef SSC_WG128_TIMING
            {
                TSTAMP(t_e);
                if (tid == 0) {
                    atomicAdd(&g_wg128_t[0], t_b - t_a);
                    atomicAdd(&g_wg128_t[1], t_c - t_b);
                    atomicAdd(&g_wg128_t[2], t_d - t_c);
                    atomicAdd(&g_wg128_t[3], t_e - t_d);
                    atomicAdd(&g_wg128_t[4], 1ull);
                }
            }
#endif
        }
    }

#ifdef SSC_WG128_TIMING
    if (tid == 0) atomicAdd(&g_wg128_t[5], __builtin_readcyclecounter() - t_kernel0);
#endif
    // ---- epilogue: tile row 2 * (row of the 32 x 32 block) + i, tile column 2 * l31 + jj ----
    float* outp = (splitk > 1) ? (slab_base + (long)ks * slab_stride) : d.out;
    const bool accum = (splitk == 1) && d.accumulate;
    const int col = n0 + wn * 64 + 2 * l31;
    if (col < d.Nn) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lhi) + i;
                if (m >= Mtot) continue;
                float2* o = reinterpret_cast<float2*>(outp + (long)m * d.ldc + col);
                float2 v = make_float2(acc[i][0][r], acc[i][1][r]);
                if (accum) {
                    const float2 t = *o;
                    v.x += t.x; v.y += t.y;
                }
                *o = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool view_plain(const ssc_gview& g) {
    return g.ab0 == nullptr && g.act == SSC_ACT_NONE &&
           (g.C1 == 0 || (g.ab1 == nullptr && (g.act1 >= 0 ? g.act1 : g.act) == SSC_ACT_NONE));
}

static bool wg128_use_bf();
// taps a 128-row tile of the gathered side can span (the kernels' TPT), 0 = not this kernel.  The exact-fp32 kernel takes whole
// 128-channel blocks of a tap (1) or a 64-channel tensor (2); the bf16 kernel also any single gathered source with a multiple of
// 4 channels from 64 up, real channels <= stored ones (wgrad128_bf16.hip: tiles over the padded row space)
static int wg128_tpt(const ssc_wgrad_desc& d) {
    const int Cg = d.g.C0 + d.g.C1;
    if (d.Cg_real == Cg) {
        if (Cg % TB == 0 && (d.g.C1 == 0 || d.g.C0 % TB == 0)) return 1;
        if (d.g.C1 == 0 && d.g.C0 == 64) return 2;
    }
    static int span = -1;       // SSC_WG128_SPAN=0: no tap-spanning tiles (A/B)
    if (span < 0) {
        const char* e = ssc_dev_getenv("SSC_WG128_SPAN");
        span = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    if (span && wg128_use_bf() && d.exact == 0 && d.g.C1 == 0 && (Cg & 3) == 0 && Cg >= 64 && d.Cg_real >= 1 && d.Cg_real <= Cg &&
        (long)d.TH * d.TW * Cg < 0x7fffff00L)
        return Cg >= TB ? 2 : 3;
    return 0;
}

extern "C" int ssc_conv_wgrad128_supported(const ssc_wgrad_desc* dp) {
    const ssc_wgrad_desc& d = *dp;
    static int off = -1;        // SSC_WGRAD128=0: always the 64-column kernel of igemm.hip (A/B)
    if (off < 0) {
        const char* e = ssc_dev_getenv("SSC_WGRAD128");
        off = (e != nullptr && e[0] == '0') ? 1 : 0;
    }
    if (off) return 0;
    const int Cg = d.g.C0 + d.g.C1, Cd = d.d.C0 + d.d.C1;
    const long P = (long)d.NB * d.PH * d.PW;
    if (wg128_tpt(d) == 0) return 0;
    if ((d.g.C0 & 3) || (d.g.C1 & 3) || (d.d.C0 & 3) || (d.d.C1 & 3)) return 0;
    if (d.Nn < TB || (d.Nn & 1) || d.ldc != d.Nn || d.Nn > Cd) return 0;
    if (d.d.C1 != 0 && d.d.C0 % TB != 0) return 0;          // a column tile inside one source
    if ((long)d.TH * d.TW * Cg < TB) return 0;
    if ((reinterpret_cast<unsigned long>(d.out) & 7) != 0) return 0;
    const long gmax = (long)d.NB * d.g.H * d.g.W * (d.g.C0 > d.g.C1 ? d.g.C0 : d.g.C1) * 4;
    const long dmax = (P + 4 * BK) * (d.d.C0 > d.d.C1 ? d.d.C0 : d.d.C1) * 4;
    if (gmax >= 0x7fffffffL || dmax >= 0x7fffffffL || P >= 0x7fffff00L || P < 8 * BK) return 0;
    return 1;
}

static int wg128_num_cu() { return ssc_num_cu(); }

// K slices: every workgroup carries the same number of K-tiles; two workgroups fit a CU (LDS).  Cost of a layout = K-tiles on
// the busiest CU (+ a fixed cost per workgroup: table, first tiles, 64 accumulator registers to store) + the slab traffic.
static int wg128_splitk(const ssc_wgrad_desc& d, int64_t ws_bytes, bool have_ws) {
    const int Cg = d.g.C0 + d.g.C1;
    const long Mtot = (long)d.TH * d.TW * Cg;
    const long P = (long)d.NB * d.PH * d.PW;
    const long nkt = (P + BK - 1) / BK;
    const long tiles = ((Mtot + TB - 1) / TB) * ((d.Nn + TB - 1) / TB);
    const long out_elems = (long)d.TH * d.TW * d.Cg_real * d.ldc;
    const int ncu = wg128_num_cu();
    static int force = -2;
    if (force == -2) {
        const char* e = ssc_dev_getenv("SSC_WG128_SPLITK");
        force = (e != nullptr) ? atoi(e) : -1;
    }
    long best = 1;
    double best_cost = 1e300;
    for (long sk = 1; sk <= 1024; ++sk) {
        if (sk > 1 && (!have_ws || nkt / sk < 6 || (int64_t)sk * out_elems * 4 > ws_bytes)) break;
        const long per = (nkt + sk - 1) / sk;
        if ((nkt + per - 1) / per != sk) continue;      // would leave empty trailing slices
        const long wgs = tiles * sk;
        const long on_cu = (wgs + ncu - 1) / ncu;                       // workgroups on the busiest CU
        double cost = (double)on_cu * (double)(per + 5);
        static double occ1 = -1.0;      // SSC_WG128_OCC1: penalty of one workgroup per CU (tuning aid)
        if (occ1 < 0.0) {
            const char* e = ssc_dev_getenv("SSC_WG128_OCC1");
            occ1 = (e != nullptr) ? atof(e) : 0.9;      // in the train step one workgroup per CU co-runs better: 1824 vs 1818 images/s (1.25)
        }
        if (on_cu == 1) cost *= occ1;
        if (sk > 1) cost += (double)sk * (double)out_elems * 4.0 * 2.0 / 3.0e6 / 3.4 + 2.0;     // slab bytes at ~3 TB/s in K-tile units (3.4 us)
        if (force > 0) cost = (double)(sk > force ? sk - force : force - sk);
        if (cost < best_cost) { best_cost = cost; best = sk; }
    }
    return (int)best;
}

void ssc_launch_wgrad_reduce(const float* ws, long count, int splitk, float* out, int accumulate, hipStream_t st);   // igemm.hip

template <bool GPLAIN, int DMODE, int TPT>
static int launch_wg128(const ssc_wgrad_desc& d, int splitk, float* ws, hipStream_t st) {
    constexpr size_t lds = (2 + NBB) * BK * TB * sizeof(float) + 2 * TPT * 256 * sizeof(int2);
    const int Cg = d.g.C0 + d.g.C1;
    const long Mtot = (long)d.TH * d.TW * Cg;
    const long P = (long)d.NB * d.PH * d.PW;
    const Magics mg = make_magics((unsigned)Cg, (unsigned)d.TW, (unsigned long)d.PW, (unsigned long)d.PH * d.PW, (unsigned long)P);
    static unsigned long long attr_done = 0;
    {
        const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&conv_wgrad128_kernel<GPLAIN, DMODE, TPT>), (int)lds, &attr_done);
        if (arc != 0) return arc;
    }
    const long out_count = Mtot * d.ldc;
    const long wgs = ((Mtot + TB - 1) / TB) * ((d.Nn + TB - 1) / TB) * splitk;
    static int xcd_on = -1;     // SSC_WG128_XCD=0: plain grid order (A/B)
    if (xcd_on < 0) {
        const char* e = ssc_dev_getenv("SSC_WG128_XCD");
        xcd_on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    const int xcd = (xcd_on && splitk > 1 && (wgs & 7) == 0) ? 1 : 0;
    hipLaunchKernelGGL((conv_wgrad128_kernel<GPLAIN, DMODE, TPT>), dim3((unsigned)wgs), dim3(256), lds, st, d, mg,
                       ws, out_count, splitk, xcd);
    if (splitk > 1) ssc_launch_wgrad_reduce(ws, out_count, splitk, d.out, d.accumulate, st);
    return (int)hipGetLastError();
}

template <int TPT>
static int launch_wg128_t(const ssc_wgrad_desc& d, int splitk, float* ws, hipStream_t st) {
    const bool gp = view_plain(d.g), dp = view_plain(d.d);
    static int dma = -1;        // SSC_WGRAD_DMA=0: plain dense tiles through registers (A/B)
    if (dma < 0) {
        const char* e = ssc_dev_getenv("SSC_WGRAD_DMA");
        dma = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    if (dp && dma) return gp ? launch_wg128<true, 0, TPT>(d, splitk, ws, st) : launch_wg128<false, 0, TPT>(d, splitk, ws, st);
    if (dp) return gp ? launch_wg128<true, 1, TPT>(d, splitk, ws, st) : launch_wg128<false, 1, TPT>(d, splitk, ws, st);
    return gp ? launch_wg128<true, 2, TPT>(d, splitk, ws, st) : launch_wg128<false, 2, TPT>(d, splitk, ws, st);
}

#ifdef SSC_WG128_TIMING
extern "C" int ssc_wg128_timing(unsigned long long* out8, int reset) {
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wg128_t), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_wg128_t), 8 * sizeof(unsigned long long));
}
#endif

// wgrad128_bf16.hip: the same tile on the bf16 matrix pipe (3-way split of both operands)
int ssc_launch_wgrad128_bf(const ssc_wgrad_desc& d, int tpt, int splitk, float* ws, hipStream_t st);
static bool wg128_use_bf() {
    static int on = -1;         // SSC_ARITH=fp32 (everything exact) or SSC_WGRAD_BF16=0 (this kernel only): the exact-fp32 MFMA form
    if (on < 0) {
        const char* a = getenv("SSC_ARITH");
        const char* w = ssc_dev_getenv("SSC_WGRAD_BF16");
        on = ((a != nullptr && (a[0] == 'f' || a[0] == 'F')) || (w != nullptr && w[0] == '0')) ? 0 : 1;
    }
    return on != 0;
}

int ssc_conv_wgrad128_bf_selected(const ssc_wgrad_desc* dp) { return (wg128_use_bf() && dp->exact == 0) ? 1 : 0; }

extern "C" int ssc_conv_wgrad128(const ssc_wgrad_desc* dp, float* ws, int64_t ws_bytes, void* stream) {
    const ssc_wgrad_desc& d = *dp;
    if (!ssc_conv_wgrad128_supported(dp)) return -10;
    const int sk = wg128_splitk(d, ws_bytes, ws != nullptr);
    if (wg128_use_bf() && d.exact == 0)
        return ssc_launch_wgrad128_bf(d, wg128_tpt(d), sk, ws, (hipStream_t)stream);
    if (d.Cg_real != d.g.C0 + d.g.C1 || wg128_tpt(d) == 3) return -10;       // (unreachable: those shapes qualify for the bf16 form only)
    return wg128_tpt(d) == 2 ? launch_wg128_t<2>(d, sk, ws, (hipStream_t)stream) : launch_wg128_t<1>(d, sk, ws, (hipStream_t)stream);
}
