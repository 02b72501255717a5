// wgrad128_bf16.hip -- the 128 x 128 filter-gradient tile of wgrad128.hip on the bf16 matrix pipe.
//
//   dF[(tap, cg)][cd] = sum_pix G[pix@tap][cg] * D[pix][cd]          (graph_single.py:24-30, 309-312: compute_gradients)
//
// Both operands are activations, so both are split on the fly (x = h + m + l exactly, three bf16, round to nearest; six products
// per fp32 product, hh into one fp32 accumulator and the five correction products into a second one: igemm_bf16.hip has the
// arithmetic and its error measurements).  The contraction index is the PIXEL, while memory holds [pixel][channel]: an MFMA
// operand wants, per lane, 8 consecutive pixels of ONE channel.  gfx950's transposing LDS read does that for free:
//   ds_read_b64_tr_b16: in a group of 16 lanes, lane i supplies the address of 8 bytes (4 bf16); the 16 x 4 block is transposed,
//   lane c receives word (c % 4) of the lanes 4j + c / 4, j = 0..3 -- with lane i pointing at row i / 4, columns 4 (i % 4) .. + 3 of
//   a row-major [4 pixels][16 channels] block, lane c receives column c of the four rows (measured: scripts/probes_r05/).
// LDS image of one K step (32 pixels) and one side: three planes of four sub-tiles [32 pixels][32 channels] bf16 (rows of 64
// bytes; sub-tile stride 2112 = 2048 + 64 bytes so that the ds_write_b64 of a 16-lane group -- 64 channels of one pixel, two
// sub-tiles -- meets 32 different banks).  A wave's 32 lanes of one half read four whole rows of a sub-tile (256 contiguous
// bytes): conflict-free.  Staging: the thread that loads 4 channels of a pixel transforms them (folded norm + activation +
// validity), splits them and stores 8 bytes per plane.
// Everything around the arithmetic is wgrad128.hip's: tiles whose 128 gathered columns lie in one tap (or two taps of a 64-channel
// tensor), the per-pixel table decoded once per 256 pixels, loads through buffer descriptors whose range check supplies the
// zeros, split-K over the pixels with slabs and the deterministic reduce.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "igemm_util.h"
#include "host_util.h"

#define BK 32
#define NP (BK / 8)
#define KPB (256 / BK)
#define TB 128
#define SSC_RSRC_FLAGS 0x00020000
#define WB_ST 2112                      // bytes between sub-tiles [32][32] bf16
#define WB_PLANE (4 * WB_ST)            // one plane of one side: 128 channels
#define WB_SIDE (3 * WB_PLANE)
#define WB_STAGE (2 * WB_SIDE)          // gathered side, dense side

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float4 wb_bload16(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    static_assert(sizeof(v) == 16, "four dwords");
    return __builtin_bit_cast(float4, v);
}
__device__ __forceinline__ unsigned wb_cvt_pk(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void wb_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = wb_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = wb_cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = wb_cvt_pk(s0, s1);
}
// split a float4 and store it: 8 bytes into each of the three planes at p (plane stride WB_PLANE)
__device__ __forceinline__ void wb_store_split(char* p, const float4& v) {
    unsigned h0, m0, l0, h1, m1, l1;
    wb_split_pair(v.x, v.y, h0, m0, l0);
    wb_split_pair(v.z, v.w, h1, m1, l1);
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + WB_PLANE) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(p + 2 * WB_PLANE) = make_uint2(l0, l1);
}
__device__ __forceinline__ s16x4 wb_tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// GPLAIN / DPLAIN: the gathered / dense side has no folded norm / activation.
// TPT: taps per tile.  1: the gathered channel count is a multiple of 128 (inside one source) -- a tile's 128 rows lie in one tap.
// 2 / 3: ONE gathered source with any multiple-of-4 channel count Cg >= 128 / >= 64 (a 64-channel tensor: two taps per tile;
// MRU's materialised concats [state | image], Cg = 132, 260 ... with Cg_real = Cg - 1 real channels, mru.py:400-411, 555-575):
// the tile is 128 consecutive rows of the PADDED row space tap * Cg + c, which spans up to TPT taps; a thread's float4 column
// lies in one of them (Cg % 4 == 0) and picks that tap's row of the pixel table; the epilogue maps padded rows back to
// tap * Cg_real + c and drops the padding channels.
// DB: two LDS stages used in turn (101 KB: one workgroup per CU) or ONE (51 KB: two workgroups per CU = two waves per SIMD).  A
// step's operands are all in registers after its two fetches, so with one stage the only extra cost is a barrier behind them,
// in front of the first store of the next K-tile; the second workgroup of the CU computes while this one waits there.
template <bool GPLAIN, bool DPLAIN, int TPT, bool DB>
__global__ __launch_bounds__(256) void conv_wgrad128_bf_kernel(const ssc_wgrad_desc d, const Magics mg, float* __restrict__ slab_base,
                                                                long slab_stride, int splitk, int xcd) {
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    int2* ptab = reinterpret_cast<int2*>(smem_b + (DB ? 2 : 1) * WB_STAGE);        // [2][TPT][256]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    const int gC0 = d.g.C0, gC1 = d.g.C1, Cg = gC0 + gC1;
    const int ntap = d.TH * d.TW;
    const int Mtot = ntap * Cg;
    const unsigned P = (unsigned)((long)d.NB * d.PH * d.PW);
    const int PHW = d.PH * d.PW;
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
    const int tap0 = div32(m0, mg.mC, mg.oneC);
    const int c0 = (TPT == 1) ? m0 - tap0 * Cg : 0;      // TPT > 1: the source starts at channel 0, the thread's column says where
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

    // per-thread piece of a staged tile: pixel rows a_r + 8 s (s = 0..3), 16-byte fp32 column a_q (4 channels)
    const int a_q = tid & 31, a_r = tid >> 5;
    int a_tsel = 0, a_cq = a_q;             // which of the tile's taps, float4 column inside that tap's channels
    if (TPT > 1) {
        const int mq = m0 + a_q * 4;
        const int tq = div32(mq, mg.mC, mg.oneC);
        a_tsel = tq - tap0;
        a_cq = (mq - tq * Cg) >> 2;
    }
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
    const unsigned d_step = (unsigned)BK * d_rowb;
    const bool b_cv = d_coff + a_q * 4 < d_cs;
    float4 ba = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
    float d_slope = 1.f;
    if (!DPLAIN) {
        gview_affine4(d.d, b_cv ? n0 + a_q * 4 : 0, ba, bb);
        d_slope = act_slope((!d_first && d.d.act1 >= 0) ? d.d.act1 : d.d.act);
    }
    unsigned b_voff[NP];
#pragma unroll
    for (int s = 0; s < NP; ++s) b_voff[s] = b_cv ? (unsigned)(a_r + 8 * s) * d_rowb + (unsigned)a_q * 16u : 0x80000000u;

    // ---- K range of this slice ----
    const int nkt = (int)((P + BK - 1) / BK);
    const int per = (nkt + splitk - 1) / splitk;
    const int kt_begin = ks * per;
    const int nk = min(nkt, kt_begin + per) - kt_begin;

    f32x16 acc[2][2], accc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = accc[i][j][r] = 0.f;

    // ---- the pixel table (wgrad128.hip): block b = pixels [(kt_begin + 8 b) * 32, + 256), one pixel per thread ----
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
    int2 pe[NP];
    auto ptab_piece = [&](int j, int s) {
        pe[s] = ptab[(((j / KPB) & 1) * TPT + a_tsel) * 256 + (j % KPB) * BK + a_r + 8 * s];
    };
    auto load_a_piece = [&](int j, int s) {
        ra[s] = wb_bload16(rsA, (unsigned)pe[s].x + a_cb);
        ram[s] = __builtin_bit_cast(float, pe[s].y);
    };
    auto rsrc_b = [&](int j) {
        const unsigned off = (unsigned)(kt_begin + j) * d_step;
        const int left = d_bytes - (int)off;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(d_base) + off), 0, left > 0 ? left : 0,
                                                 SSC_RSRC_FLAGS);
    };
    auto load_b_piece = [&](int j, int s) { rb[s] = wb_bload16(rsrc_b(j), b_voff[s]); };
    // this thread's 8 bytes of piece s inside a plane: sub-tile a_q / 8, row a_r + 8 s, 4 channels at (a_q % 8) * 8 bytes
    const int st_off = (a_q >> 3) * WB_ST + a_r * 64 + (a_q & 7) * 8;
    auto stage_a_piece = [&](int buf, int s) {
        float4 v = ra[s];       // (without a transform the descriptor's zeros for out-of-image pixels need no mask)
        if (!GPLAIN) v = xform4(v, aa, ab, g_slope, ram[s]);
        wb_store_split(smem_b + buf * WB_STAGE + st_off + s * (8 * 64), v);
    };
    auto stage_b_piece = [&](int buf, int s) {
        float4 v = rb[s];
        if (!DPLAIN) {
            float t;
            t = fmaf(ba.x, v.x, bb.x); v.x = fmaxf(t, t * d_slope);
            t = fmaf(ba.y, v.y, bb.y); v.y = fmaxf(t, t * d_slope);
            t = fmaf(ba.z, v.z, bb.z); v.z = fmaxf(t, t * d_slope);
            t = fmaf(ba.w, v.w, bb.w); v.w = fmaxf(t, t * d_slope);
        }
        wb_store_split(smem_b + buf * WB_STAGE + WB_SIDE + st_off + s * (8 * 64), v);
    };

    if (nk > 0) {
        fill_ptab(0);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NP; ++s) { ptab_piece(0, s); load_a_piece(0, s); load_b_piece(0, s); }
#pragma unroll
        for (int s = 0; s < NP; ++s) { stage_a_piece(0, s); stage_b_piece(0, s); }
#pragma unroll
        for (int s = 0; s < NP; ++s) { ptab_piece(1, s); load_a_piece(1, s); load_b_piece(1, s); }
        __syncthreads();
        // operand addresses of this lane inside a plane: row (K index) lhi * 8 + t * 4 + (lane & 15) / 4 of chunk kc, 8 bytes at
        // column 16 * ((lane >> 4) & 1) + 4 * (lane & 3) of the 32-channel sub-tile; block i of the wave's 64 = sub-tile 2 w + i
        const int rd_off = (lhi * 8 + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
        const char* a_rd = smem_b + (wm * 2) * WB_ST + rd_off;
        const char* b_rd = smem_b + WB_SIDE + (wn * 2) * WB_ST + rd_off;
        for (int j = 0; j < nk; ++j) {
            const int cur = DB ? (j & 1) : 0;
            const int nxt = DB ? (cur ^ 1) : 0;
            const char* Ab = a_rd + cur * WB_STAGE;
            const char* Bb = b_rd + cur * WB_STAGE;
            if ((j % KPB) == KPB / 2) fill_ptab(j / KPB + 1);      // wave-uniform; the table of the next 256 pixels (read from j + 2 on)
            bf16x8 av[2][2][3], bv[2][2][3];        // [kc][block][plane]
            auto fetch = [&](int kc) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const s16x4 a0 = wb_tr_read(Ab + p * WB_PLANE + i * WB_ST + kc * (16 * 64));
                        const s16x4 a1 = wb_tr_read(Ab + p * WB_PLANE + i * WB_ST + kc * (16 * 64) + 4 * 64);
                        av[kc][i][p] = (bf16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                        const s16x4 b0 = wb_tr_read(Bb + p * WB_PLANE + i * WB_ST + kc * (16 * 64));
                        const s16x4 b1 = wb_tr_read(Bb + p * WB_PLANE + i * WB_ST + kc * (16 * 64) + 4 * 64);
                        bv[kc][i][p] = (bf16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    }
            };
            // one product over the wave's four blocks (consecutive MFMAs go to different accumulators); smallest products first
            auto group = [&](int kc, int t) {
                constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        if (t == 5)
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[kc][i][0], bv[kc][jj][0], acc[i][jj], 0, 0, 0);
                        else
                            accc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[kc][i][pa[t]], bv[kc][jj][pb[t]], accc[i][jj], 0, 0, 0);
                    }
            };
#define WB_SB __builtin_amdgcn_sched_barrier(0)
            // twelve groups of four MFMAs, one piece of the step's other work behind each (wgrad128.hip's placement): K-tile j + 1
            // registers -> LDS in the first eight, the table entries and the loads of K-tile j + 2 behind them
            fetch(0);
            WB_SB;
            group(0, 0); fetch(1); WB_SB;
            if (!DB) {      // every wave holds its operands of this K-tile: the stage may be overwritten
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                WB_SB;
            }
            group(0, 1); stage_a_piece(nxt, 0); WB_SB;
            group(0, 2); stage_a_piece(nxt, 1); WB_SB;
            group(0, 3); stage_a_piece(nxt, 2); WB_SB;
            group(0, 4); stage_a_piece(nxt, 3); WB_SB;
            group(0, 5); stage_b_piece(nxt, 0); WB_SB;
            group(1, 0); stage_b_piece(nxt, 1); WB_SB;
            group(1, 1); stage_b_piece(nxt, 2); WB_SB;
            group(1, 2); stage_b_piece(nxt, 3); WB_SB;
            group(1, 3);
#pragma unroll
            for (int s = 0; s < NP; ++s) ptab_piece(j + 2, s);
            WB_SB;
            group(1, 4);
#pragma unroll
            for (int s = 0; s < NP; ++s) load_a_piece(j + 2, s);
            WB_SB;
            group(1, 5);
#pragma unroll
            for (int s = 0; s < NP; ++s) load_b_piece(j + 2, s);
            WB_SB;
#undef WB_SB
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's ds_writes; the loads of K-tile j + 2 stay in flight
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- epilogue: block (i, jj) of the wave's 64 x 64, row (r & 3) + 8 (r >> 2) + 4 lhi, column l31 ----
    float* outp = (splitk > 1) ? (slab_base + (long)ks * slab_stride) : d.out;
    const bool accum = (splitk == 1) && d.accumulate;
    const int Cgr = d.Cg_real;
    const bool remap = Cgr != Cg;           // padded rows -> tap * Cg_real + c (the padding channels have no row)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int col = n0 + wn * 64 + jj * 32 + l31;
            if (col >= d.Nn) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m >= Mtot) continue;
                if (remap) {
                    const int tp = div32(m, mg.mC, mg.oneC);
                    const int c = m - tp * Cg;
                    if (c >= Cgr) continue;
                    m = tp * Cgr + c;
                }
                float* o = outp + (long)m * d.ldc + col;
                float v = acc[i][jj][r] + accc[i][jj][r];
                if (accum) v += *o;
                *o = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void ssc_launch_wgrad_reduce(const float* ws, long count, int splitk, float* out, int accumulate, hipStream_t st);   // igemm.hip

static bool wb_view_plain(const ssc_gview& g) {
    return g.ab0 == nullptr && g.act == SSC_ACT_NONE &&
           (g.C1 == 0 || (g.ab1 == nullptr && (g.act1 >= 0 ? g.act1 : g.act) == SSC_ACT_NONE));
}

template <bool GPLAIN, bool DPLAIN, int TPT, bool DB>
static int launch_wb(const ssc_wgrad_desc& d, int splitk, float* ws, hipStream_t st) {
    constexpr size_t lds = (DB ? 2 : 1) * WB_STAGE + 2 * TPT * 256 * sizeof(int2);
    const int Cg = d.g.C0 + d.g.C1;
    const long Mtot = (long)d.TH * d.TW * Cg;
    const long P = (long)d.NB * d.PH * d.PW;
    const Magics mg = make_magics((unsigned)Cg, (unsigned)d.TW, (unsigned long)d.PW, (unsigned long)d.PH * d.PW, (unsigned long)P);
    static unsigned long long attr_done = 0;
    {
        const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&conv_wgrad128_bf_kernel<GPLAIN, DPLAIN, TPT, DB>), (int)lds, &attr_done);
        if (arc != 0) return arc;
    }
    const long out_count = (long)d.TH * d.TW * d.Cg_real * d.ldc;       // rows of the OUTPUT (the slabs' and the reduce's extent)
    const long wgs = ((Mtot + TB - 1) / TB) * ((d.Nn + TB - 1) / TB) * splitk;
    const int xcd = (splitk > 1 && (wgs & 7) == 0) ? 1 : 0;
    hipLaunchKernelGGL((conv_wgrad128_bf_kernel<GPLAIN, DPLAIN, TPT, DB>), dim3((unsigned)wgs), dim3(256), lds, st, d, mg, ws, out_count,
                       splitk, xcd);
    if (splitk > 1) ssc_launch_wgrad_reduce(ws, out_count, splitk, d.out, d.accumulate, st);
    return (int)hipGetLastError();
}

int ssc_launch_wgrad128_bf(const ssc_wgrad_desc& d, int tpt, int splitk, float* ws, hipStream_t st) {
    const bool gp = wb_view_plain(d.g), dp = wb_view_plain(d.d);
    static int db = -1;         // SSC_WGBF_DB=1: two LDS stages, one workgroup per CU (A/B)
    if (db < 0) {
        const char* e = ssc_dev_getenv("SSC_WGBF_DB");
        db = (e != nullptr && e[0] == '1') ? 1 : 0;
    }
#define WB_CASE2(T, D)                                                                                              \
    return gp ? (dp ? launch_wb<true, true, T, D>(d, splitk, ws, st) : launch_wb<true, false, T, D>(d, splitk, ws, st)) \
              : (dp ? launch_wb<false, true, T, D>(d, splitk, ws, st) : launch_wb<false, false, T, D>(d, splitk, ws, st))
#define WB_CASE(T) do { if (db) { WB_CASE2(T, true); } else { WB_CASE2(T, false); } } while (0)
    if (tpt == 3) WB_CASE(3);
    if (tpt == 2) WB_CASE(2);
    WB_CASE(1);
#undef WB_CASE
#undef WB_CASE2
}
