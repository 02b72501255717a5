// igemm.hip -- implicit-GEMM convolution kernels for gfx950 (MI355X, CDNA4).
//
// fp32 in / fp32 accumulate on the matrix cores (v_mfma_f32_32x32x2_f32: exact
// f32, 157 TFLOP/s peak), 64-wide wavefronts, 4 waves per workgroup, LDS-staged
// tiles with register prefetch of the next K-tile (one barrier per K-tile).
//
// Two kernels share the tile loaders:
//   conv_fwd_kernel  : out[pix][n]      = sum_{tap,k} X[pix@tap][k] * F(tap,k,n)
//   conv_wgrad_kernel: dF[(tap,cg)][cd] = sum_pix     G[pix@tap][cg] * D[pix][cd]
// X/G/D are "gather views" (ssc_gview): NHWC tensors (optionally the channel
// concat of two) with the folded batch-stat norm a*x+b and the activation
// applied while the tile is loaded, zero outside the image.  This is how the
// reference's  lrelu -> conv -> batchnorm  /  relu(concat) -> deconv -> batchnorm
// blocks (models_collection.py:434-439, 510-526) are fused: the norm+activation
// of layer L is evaluated inside the loads of layer L+1, so normalised tensors
// are never written to HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "igemm_util.h"
#include "igemm_epilogue.h"
#include "host_util.h"

// wgrad128.hip
extern "C" int ssc_conv_wgrad128_supported(const ssc_wgrad_desc* dp);
extern "C" int ssc_conv_wgrad128(const ssc_wgrad_desc* dp, float* ws, int64_t ws_bytes, void* stream);
int ssc_conv_wgrad128_bf_selected(const ssc_wgrad_desc* dp);
// wgn16.hip
extern "C" int ssc_conv_wgn16_supported(const ssc_wgrad_desc* dp);
int ssc_conv_wgn16(const ssc_wgrad_desc* dp, float* ws, int64_t ws_bytes, void* stream);
// narrow.hip
int ssc_conv_narrow_forward_ws(const ssc_conv_desc* dp, float* ws, int64_t ws_bytes, void* stream, int* csplit_out);
// fewchan.hip
int ssc_conv_fewchan_forward(const ssc_conv_desc* dp, int num_cu, void* stream);
// pw1x1.hip
extern "C" int ssc_conv_pw1x1_supported(const ssc_conv_desc* dp);
int ssc_conv_pw1x1_walkers(const ssc_conv_desc* dp);
int ssc_conv_pw1x1_forward(const ssc_conv_desc* dp, float* stat, void* stream);
// tr4n16.hip
extern "C" int ssc_conv_tr4n16_supported(const ssc_conv_desc* dp);
int ssc_conv_tr4n16_rows(const ssc_conv_desc* dp);
int ssc_conv_tr4n16_forward(const ssc_conv_desc* dp, float* stat, void* stream);
// fewchan7.hip
extern "C" int ssc_conv_fewchan7_supported(const ssc_conv_desc* dp);
int ssc_conv_fewchan7_walkers(const ssc_conv_desc* dp);
int ssc_conv_fewchan7_forward(const ssc_conv_desc* dp, float* stat, void* stream);
// tr4tiny.hip
extern "C" int ssc_conv_tr4_tiny_supported(const ssc_conv_desc* dp);
int ssc_conv_tr4_tiny_blocks(const ssc_conv_desc* dp);
int ssc_conv_tr4_tiny_forward(const ssc_conv_desc* dp, float* stat, void* stream);
// s2n16.hip
extern "C" int ssc_conv_s2n16_supported(const ssc_conv_desc* dp);
int ssc_conv_s2n16_walkers(const ssc_conv_desc* dp);
int ssc_conv_s2n16_forward(const ssc_conv_desc* dp, float* stat, void* stream);
// c3x3.hip
extern "C" int ssc_conv_c3x3_supported(const ssc_conv_desc* dp);
int ssc_conv_c3x3_walkers(const ssc_conv_desc* dp);
int ssc_conv_c3x3_forward(const ssc_conv_desc* dp, float* stat, void* stream);

// igemm_bf16.hip
bool ssc_bf_hk_enabled();       // igemm_bf16.hip
int ssc_launch_conv_bf(int cfg, bool plain, const ssc_conv_desc& d, int splitk, float* ws, hipStream_t st, long ts_full, int ts_s,
                       int64_t ws_bytes, int xcd);
int ssc_sk_configure_bf(const unsigned* cfg4);

#define BK 32
// in-launch K-slice hand-off (conv_ut_kernel): {wait bound in ticks of the 100 MHz wall clock (lo, hi), test hook: producers
// withhold their flags, -}.  ssc_sk_configure writes it.
__device__ unsigned g_sk_cfg[4] = {2000000000u, 0u, 0u, 0u};
#ifndef SSC_BDMA
#define SSC_BDMA 1       // filter tiles of conv_ut_kernel by LDS-DMA (global_load_lds) instead of through registers
#endif
#ifndef SSC_ADMA
#define SSC_ADMA 0       // gathered tiles of the launches without norm / activation by LDS-DMA too (buffer_load ... lds):
                         // works, measured equal to the register path (1714-1720 images/s either way) -- off by default
#endif
#ifndef SSC_UT_SGB
#define SSC_UT_SGB 1     // sched_group_barrier interleave hints in conv_ut_kernel (+1-2 % over the compiler's own order)
#endif

// ---------------------------------------------------------------------------------------------
// forward form
// ---------------------------------------------------------------------------------------------
// General form: any channel counts (K-tiles may straddle taps and sources: per-thread tap decode), scalar or float4
// filter loads.  Phases per wave: issue the next K-tile's loads | MFMAs of the current one | transform + LDS write |
// barrier.  The layers whose K-tiles lie inside one tap and one source take conv_ut_kernel below instead.
template <int WM, int WN, int SM, int SN, int BMODE, bool VECB>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ssc_conv_desc d, const Magics mg,
                                                        float* __restrict__ slab_base, long slab_stride,
                                                        int splitk) {
    constexpr int BM = WM * SM * 32;
    constexpr int BN = WN * SN * 32;
    constexpr int A_LD = BK + 1;
    constexpr int A_SZ = BM * A_LD;
    constexpr int B_LD = (BMODE == 0) ? BN : (BK + 1);
    constexpr int B_SZ = (BMODE == 0) ? BK * BN : BN * (BK + 1);
    constexpr int A_ROWS = BM / 32;   // rows of the A tile per thread
    constexpr int B_SLOTS = BN / 32;  // float4 slots of the B tile per thread
    constexpr int B_RP = 1024 / BN;   // filter rows per pass (KN)
    static_assert(WM * WN == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][A_SZ]
    float* Bs = smem + 2 * A_SZ;      // [2][B_SZ]
    long* rowpix = reinterpret_cast<long*>(smem + 2 * A_SZ + 2 * B_SZ);   // [BM] output pixel of each tile row

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int C = d.x.C0 + d.x.C1;
    const int Ktot = d.TH * d.TW * C;
    const long M = (long)d.NB * d.PH * d.PW;
    const int PHW = d.PH * d.PW;
    const int phase = blockIdx.z / splitk;
    const int ks = blockIdx.z % splitk;
    const FwdPhase ph = fwd_phase(d, phase);

    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-thread A rows: pixel decode (fixed for the whole kernel) ----
    const int a_col4 = tid & 7;            // float4 column inside the K tile
    int a_iyb[A_ROWS], a_ixb[A_ROWS];
    long a_nb[A_ROWS];
    bool a_mv[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const long m = m0 + row;
        a_mv[i] = m < M;
        const long mm = a_mv[i] ? m : 0;
        const int n = (int)div64(mm, mg.mPHPW, mg.onePHPW);
        const int rem = (int)(mm - (long)n * PHW);
        const int py = (int)div64(rem, mg.mPW, mg.onePW), px = rem - py * d.PW;
        a_iyb[i] = py * d.in_stride + ph.ioff_y;
        a_ixb[i] = px * d.in_stride + ph.ioff_x;
        a_nb[i] = (long)n * d.x.H * d.x.W;
        if (a_col4 == 0)
            rowpix[row] = ((long)n * d.OH + py * d.out_stride + ph.ooff_y) * d.OW + px * d.out_stride + ph.ooff_x;
    }

    const int nkt = (Ktot + BK - 1) / BK;
    const int per = (nkt + splitk - 1) / splitk;
    const int kt_begin = ks * per;
    const int kt_end = min(nkt, kt_begin + per);

    f32x16 acc[SM][SN];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging registers
    float4 ra[A_ROWS];
    float rav[A_ROWS];    // 1.0 / 0.0 validity of each staged A row
    float4 raa, rab;
    float ra_slope = 1.f;    // activation slope of the source this K-tile reads
    float4 rb[B_SLOTS];
    float4 rbm[B_SLOTS];  // per-element 1.0 / 0.0 validity of the staged filter values
    const float x_slope0 = act_slope(d.x.act), x_slope1 = act_slope(d.x.act1 >= 0 ? d.x.act1 : d.x.act);

    auto load_tile = [&](int kt) {
        // A (gather view): every load unconditional from a clamped address
        {
            const int kcol = kt * BK + a_col4 * 4;
            const bool kv = kcol < Ktot;
            const int kk = kv ? kcol : 0;
            const int tap = div32(kk, mg.mC, mg.oneC);
            const int c = kk - tap * C;
            const int ty = div32(tap, mg.mTW, mg.oneTW), tx = tap - ty * d.TW;
            gview_affine4(d.x, c, raa, rab);
            ra_slope = c < d.x.C0 ? x_slope0 : x_slope1;
            const float* base;
            int cs;
            gview_src(d.x, c, base, cs);
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                const int iy = a_iyb[i] + ty, ix = a_ixb[i] + tx;
                const bool v = kv && a_mv[i] && iy >= 0 && iy < d.x.H && ix >= 0 && ix < d.x.W;
                const long pix = v ? a_nb[i] + (long)iy * d.x.W + ix : 0;
                rav[i] = v ? 1.f : 0.f;
                ra[i] = *reinterpret_cast<const float4*>(base + pix * cs);
            }
        }
        if (BMODE == 0) {           // KN: tile [BK rows k][BN cols n], n contiguous in memory
            const int n = n0 + (tid % (BN / 4)) * 4;
#pragma unroll
            for (int s = 0; s < B_SLOTS; ++s) {
                const int krow = kt * BK + tid / (BN / 4) + B_RP * s;
                const bool kv = krow < Ktot;
                const int kk = kv ? krow : 0;
                const int tp = div32(kk, mg.mC, mg.oneC);
                const int kc = kk - tp * C;
                const int tty = div32(tp, mg.mTW, mg.oneTW), ttx = tp - tty * d.TW;
                const int ky = ph.ky0 + tty * d.kstep, kx = ph.kx0 + ttx * d.kstep;
                const bool rowv = kv && kc < d.k_real;
                const float* wp = d.w + ((long)(ky * d.KW + kx) * d.wC0 + (rowv ? kc : 0)) * d.wC1 + d.n_off;
                if (VECB) {
                    const bool cv = n < d.Nn;       // Nn % 4 == 0: the whole float4 is in or out
                    rb[s] = *reinterpret_cast<const float4*>(wp + (cv ? n : 0));
                    const float m = (rowv && cv) ? 1.f : 0.f;
                    rbm[s] = make_float4(m, m, m, m);
                } else {
                    const bool v0 = rowv && n + 0 < d.Nn, v1 = rowv && n + 1 < d.Nn;
                    const bool v2 = rowv && n + 2 < d.Nn, v3 = rowv && n + 3 < d.Nn;
                    rb[s].x = wp[v0 ? n + 0 : 0];
                    rb[s].y = wp[v1 ? n + 1 : 0];
                    rb[s].z = wp[v2 ? n + 2 : 0];
                    rb[s].w = wp[v3 ? n + 3 : 0];
                    rbm[s] = make_float4(v0 ? 1.f : 0.f, v1 ? 1.f : 0.f, v2 ? 1.f : 0.f, v3 ? 1.f : 0.f);
                }
            }
        } else {                    // NK: tile [BN rows n][BK cols k], k contiguous in memory
            const int kc4 = kt * BK + (tid & 7) * 4;
            const bool kvb = kc4 < Ktot;
            const int kk = kvb ? kc4 : 0;
            const int tp = div32(kk, mg.mC, mg.oneC);
            const int kc = kk - tp * C;
            const int tty = div32(tp, mg.mTW, mg.oneTW), ttx = tp - tty * d.TW;
            const int ky = ph.ky0 + tty * d.kstep, kx = ph.kx0 + ttx * d.kstep;
            const long tapoff = (long)(ky * d.KW + kx) * d.wC0 + d.n_off;
#pragma unroll
            for (int s = 0; s < B_SLOTS; ++s) {
                const int n = n0 + (tid >> 3) + 32 * s;
                const bool nv = kvb && n < d.Nn;
                const float* wp = d.w + (tapoff + (nv ? n : 0)) * d.wC1;
                if (VECB) {
                    const bool cv = kc < d.k_real;  // k_real % 4 == 0
                    rb[s] = *reinterpret_cast<const float4*>(wp + (cv ? kc : 0));
                    const float m = (nv && cv) ? 1.f : 0.f;
                    rbm[s] = make_float4(m, m, m, m);
                } else {
                    const bool v0 = nv && kc + 0 < d.k_real, v1 = nv && kc + 1 < d.k_real;
                    const bool v2 = nv && kc + 2 < d.k_real, v3 = nv && kc + 3 < d.k_real;
                    rb[s].x = wp[v0 ? kc + 0 : 0];
                    rb[s].y = wp[v1 ? kc + 1 : 0];
                    rb[s].z = wp[v2 ? kc + 2 : 0];
                    rb[s].w = wp[v3 ? kc + 3 : 0];
                    rbm[s] = make_float4(v0 ? 1.f : 0.f, v1 ? 1.f : 0.f, v2 ? 1.f : 0.f, v3 ? 1.f : 0.f);
                }
            }
        }
    };

    auto store_tile = [&](int buf) {
        float* Ab = As + buf * A_SZ;
        float* Bb = Bs + buf * B_SZ;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const float4 v = xform4(ra[i], raa, rab, ra_slope, rav[i]);
            float* p = Ab + ((tid >> 3) + 32 * i) * A_LD + a_col4 * 4;
            p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
        }
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            float4 v = rb[s];
            v.x *= rbm[s].x; v.y *= rbm[s].y; v.z *= rbm[s].z; v.w *= rbm[s].w;
            if (BMODE == 0) {
                *reinterpret_cast<float4*>(Bb + (tid / (BN / 4) + B_RP * s) * B_LD + (tid % (BN / 4)) * 4) = v;
            } else {
                float* p = Bb + ((tid >> 3) + 32 * s) * B_LD + (tid & 7) * 4;
                p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
            }
        }
    };

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();

    // main loop: the global loads of K-tile kt+1 are issued before the 64 MFMAs of K-tile kt and written to
    // the other LDS buffer after them (one barrier per K-tile)
    const int l31 = lane & 31, lhi = lane >> 5;
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = (kt + 1) < kt_end;
        if (more) load_tile(kt + 1);
        const float* Ab = As + cur * A_SZ + (wm * SM * 32 + l31) * A_LD + lhi;
        const float* Bb = (BMODE == 0) ? (Bs + cur * B_SZ + lhi * B_LD + wn * SN * 32 + l31)
                                       : (Bs + cur * B_SZ + (wn * SN * 32 + l31) * B_LD + lhi);
        // LDS -> register operand fetch, software-pipelined by groups of FG k-steps: the ds_reads of group g+1 are
        // issued before the MFMAs of group g (the compiler otherwise places every read directly in front of its
        // MFMA pair behind an s_waitcnt lgkmcnt(0), exposing the LDS latency 16 times per K-tile)
        constexpr int FG = 4, NFG = BK / 2 / FG;
        float av[2][FG][SM], bv[2][FG][SN];
        auto fetch = [&](int g, int buf) {
#pragma unroll
            for (int q = 0; q < FG; ++q) {
                const int kk = g * FG + q;
#pragma unroll
                for (int i = 0; i < SM; ++i) av[buf][q][i] = Ab[i * 32 * A_LD + kk * 2];
#pragma unroll
                for (int j = 0; j < SN; ++j)
                    bv[buf][q][j] = (BMODE == 0) ? Bb[kk * 2 * B_LD + j * 32] : Bb[j * 32 * B_LD + kk * 2];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int g = 0; g < NFG; ++g) {
            if (g + 1 < NFG) fetch(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < FG; ++q)
#pragma unroll
                for (int i = 0; i < SM; ++i)
#pragma unroll
                    for (int j = 0; j < SN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][q][i], bv[g & 1][q][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue ----
    float* outp = (splitk > 1) ? (slab_base + (long)ks * slab_stride) : d.out;
    const bool final_pass = (splitk == 1);
#pragma unroll
    for (int i = 0; i < SM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * SM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m0 + row >= M) continue;
            const long opix = rowpix[row];
#pragma unroll
            for (int j = 0; j < SN; ++j) {
                const int col = n0 + wn * SN * 32 + j * 32 + l31;
                if (col >= d.Nstore) continue;
                float v = acc[i][j][r];
                float* o = outp + opix * d.ldc + col;
                if (final_pass) {
                    if (d.bias != nullptr && col < d.Nn) v += d.bias[col];
                    if (d.epi == 1) v = tanhf(v);
                    else if (d.epi == 2) v = fmaxf(v, 0.2f * v);
                    if (d.accumulate) v += *o;
                }
                *o = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// uniform-tap form with the staging of later K-tiles interleaved into the MFMA stream of the same wave
// ---------------------------------------------------------------------------------------------
// conv_fwd_kernel alternates phases per wave (issue loads | 32-64 MFMAs | transform + LDS writes | barrier) and relies on
// the 2-3 co-resident waves of a SIMD being in different phases to keep the matrix pipe fed; measured, the pipe sits at
// 60-70 %.  Here one K-tile step is a single branch-free block: the registers that hold K-tile kt+1 (loaded one step
// ago) go to the free LDS buffer and are refilled with the global loads of K-tile kt+2 while the MFMAs of K-tile kt
// issue, so address arithmetic, memory issue and LDS traffic sit in the 64-cycle shadow of each MFMA.
//   * no per-step branches: the K-tile index is clamped instead of tested, selects are arithmetic, the filter columns
//     beyond Nn are zeroed in the epilogue instead of masked every step;
//   * PLAIN (no folded norm, no activation on any source: every data-gradient launch) is a template flag.
#define UT_AV(BM, BN) (!((BM) == 128 && (BN) <= 64))
// KMASK: channel counts that are not multiples of 32 (the MRU concats: 576 + 4, of which 579 real).  K-tiles are then
// enumerated per (tap, source, 32-channel chunk), mg.mC holding the magic of chunks-per-tap, so a tile still lies inside
// one tap and one source; the last chunk of a source is partly empty: its A float4s beyond the source's channels are
// not loaded and its filter rows beyond the real channels are zeroed (the A padding channels need not hold zeros).
// KM = 2 ("row tap"): few channels, TW * C == 32 (the discriminator's first conv: 4 taps x 8 channels).  A filter ROW is
// then one K-tile: its TW taps x C channels are 32 contiguous floats both in the NHWC input (consecutive pixels) and in
// the filter, so the launch runs as TH taps of "32 channels" (the host sets TW = 1); only the x bound differs per thread.
template <int WM, int WN, int SM, int SN, int BMODE, bool PLAIN, int KM>
__global__ __launch_bounds__(256) void conv_ut_kernel(const ssc_conv_desc d, const Magics mg,
                                                       float* __restrict__ slab_base, long slab_stride, int splitk,
                                                       int ts_full, int ts_s, unsigned* __restrict__ flags) {
    constexpr int BM = WM * SM * 32;
    constexpr int BN = WN * SN * 32;
    // AV: A rows padded to 36 floats (16-byte aligned, b128 accesses conflict-free) and the K index of MFMA step kk on
    // lane half lhi permuted to k = 16*lhi + kk (any bijection works when A and B agree): a lane's 16 A operands of a
    // K-tile are then contiguous, 4 ds_read_b128 instead of 16 ds_read_b32 per row block, and each staged float4 is one
    // ds_write_b128.  Not for the 128-row x <=64-column tiles, whose 3 workgroups per CU would no longer fit the LDS.
    constexpr bool KMASK = (KM == 1), ROWTAP = (KM == 2);
    // BDMA: the filter tile goes global memory -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave instruction, lane L
    // lands at +16 L), no staging registers, no ds_write, no vector ALU work per K-tile.  A wave instruction covers
    // 256 / BN consecutive k rows of the [k][n] image (KN) or 8 rows of the swizzled [n][32] image (NK: each lane fetches the
    // 16-byte chunk whose swizzled position is its own slot).  Not with KMASK, whose partial chunks need zeroed rows.
    constexpr bool BDMA = SSC_BDMA && !KMASK;
    constexpr int B_IPW = BK * BN / 1024;          // DMA instructions per wave and K-tile
    // ADMA (launches without norm / activation on any source, i.e. every data gradient): the gathered tile too goes straight
    // into LDS -- buffer_load_dwordx4 ... lds through a buffer descriptor per source, whose range check writes ZEROS for the
    // lanes sent out of range (padding taps, rows beyond M): no mask, no staging registers, no ds_write.  The tile is the
    // swizzled [row][32] image of the NK filter tile (8 rows per wave instruction).
    constexpr bool ADMA = SSC_ADMA && BDMA && PLAIN && KM == 0;
    constexpr bool AV = ADMA || UT_AV(BM, BN);
    constexpr int A_LD = ADMA ? BK : (AV ? BK + 4 : BK + 1);
    constexpr int A_SZ = BM * (UT_AV(BM, BN) ? BK + 4 : BK + 1);       // buffer size as the host allocates it (the ADMA image is smaller)
    // NK filters (k contiguous in memory): the B tile is [n][32] with the 16-byte chunk c of row n stored at chunk
    // c ^ ((n >> 1) & 7) -- b128 writes and b128 operand reads, both conflict-free, no padding (the scalar [n][33] image it
    // replaces cost 34 address adds per K-tile and wave)
    constexpr int B_LD = (BMODE == 0) ? BN : BK;
    constexpr int B_SZ = (BMODE == 0) ? BK * BN : BN * BK;
    constexpr int A_ROWS = BM / 32;
    constexpr int B_SLOTS = BN / 32;
    constexpr int B_RP = 1024 / BN;
    static_assert(WM * WN == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * A_SZ;
    long* rowpix = reinterpret_cast<long*>(smem + 2 * A_SZ + 2 * B_SZ);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    // descriptor fields the K loop needs, in registers (the loop must not go back to the kernarg segment)
    const float* const xs0 = d.x.s0;
    const float* const xs1 = d.x.s1;
    const float* const xab0 = d.x.ab0;
    const float* const xab1 = d.x.ab1;
    const float* const wbase = d.w;
    const int xC0 = d.x.C0, xC1 = d.x.C1, xH = d.x.H, xW = d.x.W;
    const int TWv = d.TW, kstep = d.kstep, KWv = d.KW, wC0 = d.wC0, wC1 = d.wC1;
    const float slope0 = act_slope(d.x.act), slope1 = act_slope(d.x.act1 >= 0 ? d.x.act1 : d.x.act);

    const int C = xC0 + xC1;
    const int nch0 = ROWTAP ? 1 : (xC0 + BK - 1) / BK, nch1 = ROWTAP ? 0 : (xC1 + BK - 1) / BK;   // 32-channel chunks per source
    const int tpt = nch0 + nch1;                                          // K-tiles per tap
    const int kreal0 = min(xC0, d.k_real), kreal1 = d.k_real - xC0;        // real filter channels of each source
    const long M = (long)d.NB * d.PH * d.PW;
    const int PHW = d.PH * d.PW;
    // workgroup -> (tile, K range).  Legacy grid: x = row tile, y = column tile, z = phase * splitk + K slice.
    // Tail split (ts_s > 0, 1-D grid, launches without split-K): tiles [0, ts_full) fill whole rounds of the chip and are
    // finished in place; each of the remaining tiles -- the partly filled last round -- is cut into ts_s K slices that
    // together fill that round.  The slices of a tile are combined INSIDE the launch: slices 0 .. ts_s-2 write raw partial
    // tiles (accumulator order, write-through stores) and raise a flag each; the last slice -- the highest workgroup id of
    // the tile, so everything it waits for was dispatched before it -- adds them to its accumulators in slice order (a
    // fixed summation order) and runs the epilogue.  Agent-scope release / acquire as the CDNA4 guide prescribes: sc1
    // stores, every storing wave drains, one lane stores the flag relaxed; the owner polls relaxed, one acquire fence,
    // then plain loads.  The owner leaves the flags zero; its waits are bounded (flag word SSC_SK_FLAG_WORDS-1 = timeout).
    int phase, ks, sk, n0, slot = 0;
    long m0;
    float* part = nullptr;
    if (ts_s > 0) {
        const int bid = blockIdx.x;
        int tile;
        if (bid < ts_full) {
            tile = bid; ks = 0; sk = 1;
        } else {
            const int r = bid - ts_full;
            const int q = r / (ts_s & 0xffff);
            tile = ts_full + q; ks = r - q * (ts_s & 0xffff); sk = ts_s & 0xffff;
            part = slab_base + (long)r * (BM * BN);
            slot = r;
        }
        const int mt = (int)((M + BM - 1) / BM), nt = (d.Nstore + BN - 1) / BN;
        if (ts_s & 0x10000) {
            // XCD-aware order (ts_s high half set by the host): tiles are numbered column tile fastest, then row tile, then
            // phase, and the whole tiles are dealt so that workgroup ids with the same id % 8 -- the same XCD, the same L2
            // -- walk a contiguous run of that order: the column tiles of a row tile (same gathered rows) and spatially
            // neighbouring row tiles (overlapping taps) meet in one L2 instead of eight
            if (bid < ts_full) tile = (bid & 7) * (ts_full >> 3) + (bid >> 3);
            const int r2 = tile / nt;
            n0 = (tile - r2 * nt) * BN;
            if (ts_s & 0x20000) {       // 4 sub-pixel phases of a row tile next to each other: they gather the same 3x3 input
                phase = r2 & 3;         // neighbourhoods (16 (phase, tap) pairs, 9 distinct pixels), so they share them in one L2
                m0 = (long)(r2 >> 2) * BM;
            } else {
                phase = r2 / mt;
                m0 = (long)(r2 - phase * mt) * BM;
            }
            ts_s &= 0xffff;
            sk = bid < ts_full ? 1 : ts_s;
        } else {
            const int rest = tile / mt;
            m0 = (long)(tile - rest * mt) * BM;
            phase = rest / nt;
            n0 = (rest - phase * nt) * BN;
        }
    } else {
        phase = blockIdx.z / splitk;
        ks = blockIdx.z % splitk;
        sk = splitk;
        m0 = (long)blockIdx.x * BM;
        n0 = blockIdx.y * BN;
    }
    const FwdPhase ph = fwd_phase(d, phase);

    const int a_kx = ROWTAP ? ((tid & 7) * 4) / xC0 : 0;    // row tap: this thread's float4 belongs to pixel ix + a_kx
    int a_iyb[A_ROWS], a_ixb[A_ROWS], a_off0[A_ROWS], a_off1[A_ROWS];
    bool a_mv[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        // tile row and 16-byte chunk of this thread's i-th piece.  Register path: 32 rows per pass, chunk tid & 7.  ADMA: wave
        // instruction w * A_ROWS + i covers 8 rows, lane L lands at slot L & 7 of row L >> 3 and so fetches the chunk whose
        // swizzled position that is.
        const int row = ADMA ? (wave * A_ROWS + i) * 8 + (lane >> 3) : (tid >> 3) + 32 * i;
        const int a_col4 = ADMA ? ((lane & 7) ^ ((row >> 1) & 7)) : (tid & 7);
        const long m = m0 + row;
        a_mv[i] = m < M;
        const long mm = a_mv[i] ? m : 0;
        int n, rem, py;
        if (mg.use32) {     // wave-uniform: every numerator x divisor fits 32 bits
            n = (int)__umulhi((unsigned)mm, mg.mPHPW32) + (int)((unsigned)mm & (unsigned)mg.onePHPW);
            rem = (int)mm - n * PHW;
            py = (int)__umulhi((unsigned)rem, mg.mPW32) + (int)((unsigned)rem & (unsigned)mg.onePW);
        } else {
            n = (int)div64(mm, mg.mPHPW, mg.onePHPW);
            rem = (int)(mm - (long)n * PHW);
            py = (int)div64(rem, mg.mPW, mg.onePW);
        }
        const int px = rem - py * d.PW;
        a_iyb[i] = py * d.in_stride + ph.ioff_y;
        a_ixb[i] = px * d.in_stride + ph.ioff_x;
        const int pix0 = (n * xH + a_iyb[i]) * xW + a_ixb[i];
        a_off0[i] = (pix0 * xC0 + a_col4 * 4) * 4;      // BYTE offsets from a wave-uniform base: the loads then take the
        a_off1[i] = (pix0 * xC1 + a_col4 * 4) * 4;      // scalar-base + 32-bit vector-offset form, no 64-bit address arithmetic
        if ((tid & 7) == 0)
            rowpix[row] = ((long)n * d.OH + py * d.out_stride + ph.ooff_y) * d.OW + px * d.out_stride + ph.ooff_x;
    }
    const int a_col4 = tid & 7;     // register path: chunk of this thread's float4s
    // filter offsets of this thread's float4 slots, split into the part along K (b_k: filter rows of the tile for KN,
    // element offset along the row for NK) and the rest (b_n), so that KMASK can drop the K part of an empty slot
    int b_k[B_SLOTS], b_n[B_SLOTS], b_kk[B_SLOTS];     // b_kk: k index inside the tile
#pragma unroll
    for (int s = 0; s < B_SLOTS; ++s) {
        if (BMODE == 0) {
            const int n = n0 + (tid % (BN / 4)) * 4;
            const bool nv = n < d.Nn;
            b_kk[s] = tid / (BN / 4) + B_RP * s;
            b_k[s] = nv ? b_kk[s] * wC1 * 4 : 0;            // bytes
            b_n[s] = nv ? (d.n_off + n) * 4 : 0;
        } else {
            const int n = n0 + (tid >> 3) + 32 * s;
            const bool nv = n < d.Nn;
            b_kk[s] = (tid & 7) * 4;
            b_k[s] = nv ? b_kk[s] * 4 : 0;
            b_n[s] = nv ? (d.n_off + n) * wC1 * 4 : 0;
        }
    }

    unsigned bd_off[B_IPW];     // BDMA: this lane's byte offset inside the filter for each of its wave's instructions
    if (BDMA) {
#pragma unroll
        for (int q = 0; q < B_IPW; ++q) {
            const int ins = wave * B_IPW + q;
            if (BMODE == 0) {
                constexpr int RPI = 256 / BN;           // k rows per instruction
                const int r = ins * RPI + lane / (BN / 4), n = n0 + (lane % (BN / 4)) * 4;
                bd_off[q] = n < d.Nn ? (unsigned)((r * wC1 + d.n_off + n) * 4) : (unsigned)(r * wC1 * 4);
            } else {
                const int r = ins * 8 + (lane >> 3), n = n0 + r;
                const int c = (lane & 7) ^ ((r >> 1) & 7);      // the chunk whose swizzled slot is lane & 7
                bd_off[q] = n < d.Nn ? (unsigned)(((d.n_off + n) * wC1 + c * 4) * 4) : (unsigned)(c * 16);
            }
        }
    }

    const int nkt = d.TH * d.TW * tpt;
    const int per = (nkt + sk - 1) / sk;
    const int kt_begin = ks * per;
    const int kt_end = min(nkt, kt_begin + per);

    f32x16 acc[SM][SN];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_ROWS];
    float rav[A_ROWS];
    float4 raa, rab;
    float ra_slope = 1.f;
    float4 rb[B_SLOTS];
    float rbv[B_SLOTS];      // KMASK: 1.0 / 0.0 validity of each staged filter float4

    auto issue_loads = [&](int kt) {
        if (ADMA) return;
        const int tap = div32(kt, mg.mC, mg.oneC);      // mC: magic of tpt
        const int chunk = kt - tap * tpt;
        const int ty = div32(tap, mg.mTW, mg.oneTW), tx = tap - ty * TWv;
        const bool first = chunk < nch0;
        const int cs = first ? xC0 : xC1;
        const int cc = (first ? chunk : chunk - nch0) * BK;     // channel offset inside the source
        const int cch = first ? cc : xC0 + cc;                  // channel offset inside the filter's K rows
        const int kreal = (first ? kreal0 : kreal1) - cc;       // real channels left in this source from cc on
        const int cchf = (!KMASK || kreal > 0) ? cch : 0;       // a chunk without real channels reads (and drops) row 0
        const char* sbase = reinterpret_cast<const char*>((first ? xs0 : xs1) + cc);
        const int tapshift = (ty * xW + tx) * cs * 4;
        const int fmask = first ? -1 : 0;
        if (!PLAIN) {
            const float* abp = first ? xab0 : xab1;
            const bool has = abp != nullptr;
            const bool hc = has & (!KMASK || cc + a_col4 * 4 < cs);     // KMASK: no table entries beyond the source
            const float* pa = hc ? abp + cc + a_col4 * 4 : xs0;       // a valid address either way
            const float* pb = hc ? abp + cs + cc + a_col4 * 4 : xs0;
            const float4 va = *reinterpret_cast<const float4*>(pa);
            const float4 vb = *reinterpret_cast<const float4*>(pb);
            raa = has ? va : make_float4(1.f, 1.f, 1.f, 1.f);
            rab = has ? vb : make_float4(0.f, 0.f, 0.f, 0.f);
            ra_slope = first ? slope0 : slope1;
        }
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const int iy = a_iyb[i] + ty, ix = a_ixb[i] + tx;
            const bool v = a_mv[i] & ((unsigned)iy < (unsigned)xH) & ((unsigned)(ix + a_kx) < (unsigned)xW) &
                           (!KMASK || cc + a_col4 * 4 < cs);
            const int osel = (a_off0[i] & fmask) | (a_off1[i] & ~fmask);    // bit select: a ?: here became a scratch array
            // unsigned 32-bit element offset from a wave-uniform base: one shift, no 64-bit vector address arithmetic
            const unsigned off = v ? (unsigned)(osel + tapshift) : (unsigned)(a_col4 * 16);
            rav[i] = v ? 1.f : 0.f;
            ra[i] = *reinterpret_cast<const float4*>(sbase + off);
        }
        if (BDMA) return;
        const int ky = ph.ky0 + ty * kstep, kx = ph.kx0 + tx * kstep;
        const char* wtap = reinterpret_cast<const char*>((BMODE == 0) ? wbase + ((long)(ky * KWv + kx) * wC0 + cchf) * wC1
                                                                      : wbase + (long)(ky * KWv + kx) * wC0 * wC1 + cchf);
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            if (KMASK) {
                const bool bv = b_kk[s] < kreal;        // k_real % 4 == 0 for NK: the float4 is all in or all out
                rbv[s] = bv ? 1.f : 0.f;
                rb[s] = *reinterpret_cast<const float4*>(wtap + (unsigned)((bv ? b_k[s] : 0) + b_n[s]));
            } else {
                rb[s] = *reinterpret_cast<const float4*>(wtap + (unsigned)(b_k[s] + b_n[s]));
            }
        }
    };

    // ADMA: gathered K-tile kt straight into LDS buffer `buf`
    const int a_px = d.NB * xH * xW;        // pixels of a source (host: pixels x channels < 2^29)
    auto dma_a = [&](int kt, int buf) {
        const int tap = div32(kt, mg.mC, mg.oneC);
        const int chunk = kt - tap * tpt;
        const int ty = div32(tap, mg.mTW, mg.oneTW), tx = tap - ty * TWv;
        const bool first = chunk < nch0;
        const int cs = first ? xC0 : xC1;
        const int cc = (first ? chunk : chunk - nch0) * BK;
        const int tapshift = (ty * xW + tx) * cs * 4;
        const int fmask = first ? -1 : 0;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)((buf * A_SZ + wave * A_ROWS * 256) * 4));
        // the descriptor of the source this K-tile reads (scalar ALU only): base, size in bytes = the range the check enforces
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? xs0 : xs1), 0, a_px * cs * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const int iy = a_iyb[i] + ty, ix = a_ixb[i] + tx;
            const bool v = a_mv[i] & ((unsigned)iy < (unsigned)xH) & ((unsigned)ix < (unsigned)xW);
            const int osel = (a_off0[i] & fmask) | (a_off1[i] & ~fmask);
            const unsigned off = v ? (unsigned)(osel + tapshift) : 0x80000000u;     // out of range: the DMA writes zeros
            glds16_buf(rs, off, (unsigned)(cc * 4), dst + i * 1024);
        }
    };

    // BDMA: filter K-tile kt straight into LDS buffer `buf`
    auto dma_b = [&](int kt, int buf) {
        const int tap = div32(kt, mg.mC, mg.oneC);
        const int chunk = kt - tap * tpt;
        const int ty = div32(tap, mg.mTW, mg.oneTW), tx = tap - ty * TWv;
        const int cch = ROWTAP ? 0 : (chunk < nch0 ? chunk * BK : xC0 + (chunk - nch0) * BK);
        const int ky = ph.ky0 + ty * kstep, kx = ph.kx0 + tx * kstep;
        const char* wtap = reinterpret_cast<const char*>((BMODE == 0) ? wbase + ((long)(ky * KWv + kx) * wC0 + cch) * wC1
                                                                      : wbase + (long)(ky * KWv + kx) * wC0 * wC1 + cch);
        // LDS byte address of this wave's first slot (the dynamic LDS segment starts at 0: the kernel has no static LDS)
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(((Bs - smem) + buf * B_SZ + wave * B_IPW * 256) * 4));
#pragma unroll
        for (int q = 0; q < B_IPW; ++q) glds16(wtap, bd_off[q], dst + q * 1024);
    };

    auto stage = [&](int buf) {
        if (ADMA) return;
        float* Ab = As + buf * A_SZ;
        float* Bb = Bs + buf * B_SZ;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const float4 v = PLAIN ? mask4(ra[i], rav[i]) : xform4(ra[i], raa, rab, ra_slope, rav[i]);
            float* p = Ab + ((tid >> 3) + 32 * i) * A_LD + a_col4 * 4;
            if (AV) {
                *reinterpret_cast<float4*>(p) = v;
            } else {
                p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
            }
        }
        if (BDMA) return;
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            const float4 v = KMASK ? mask4(rb[s], rbv[s]) : rb[s];
            if (BMODE == 0) {
                *reinterpret_cast<float4*>(Bb + (tid / (BN / 4) + B_RP * s) * B_LD + (tid % (BN / 4)) * 4) = v;
            } else {
                *reinterpret_cast<float4*>(Bb + ((tid >> 3) + 32 * s) * B_LD + (((tid & 7) ^ ((tid >> 4) & 7)) * 4)) = v;
            }
        }
    };

    if (kt_begin < kt_end) {
        const int last = kt_end - 1;
        if (ADMA) dma_a(kt_begin, 0);
        if (BDMA) dma_b(kt_begin, 0);
        issue_loads(kt_begin);
        stage(0);
        issue_loads(min(kt_begin + 1, last));
        if (BDMA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ADMA ? 0 : A_ROWS) : "memory");   // the first DMA tile(s) have landed
        __syncthreads();
        int cur = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            // K index of MFMA step kk on lane half lhi: 16*lhi + kk (any bijection works when A and B agree; this one makes
            // a lane's 16 operands of a K-tile contiguous in the [row][k] images)
            constexpr int KS = 1;                       // k stride between consecutive steps
            const int kl = lhi * (BK / 2);              // k of step 0
            const float* Ab = As + cur * A_SZ + (wm * SM * 32 + l31) * A_LD + (ADMA ? 0 : kl);
            // ADMA: every wave is past the barrier that ended K-tile kt-1, so buffer cur^1 is free: both tiles of K-tile kt+1
            // are issued now and have this whole K step to land
            if (ADMA) {
                dma_a(min(kt + 1, last), cur ^ 1);
                dma_b(min(kt + 1, last), cur ^ 1);
            }
            const float* Bb = (BMODE == 0) ? (Bs + cur * B_SZ + kl * B_LD + wn * SN * 32 + l31)
                                           : (Bs + cur * B_SZ + (wn * SN * 32 + l31) * B_LD);
            const int bsw = (l31 >> 1) & 7;         // NK: chunk swizzle of this lane's rows (the same for every j)
            constexpr int FG = 4, NFG = BK / 2 / FG;
            float av[2][FG][SM], bv[2][FG][SN];
            auto fetch = [&](int g, int buf) {
                if (AV) {
#pragma unroll
                    for (int i = 0; i < SM; ++i) {
                        const float4 v = *reinterpret_cast<const float4*>(Ab + i * 32 * A_LD +
                                                                          (ADMA ? ((lhi * 4 + g) ^ bsw) * 4 : g * FG));
                        av[buf][0][i] = v.x; av[buf][1][i] = v.y; av[buf][2][i] = v.z; av[buf][3][i] = v.w;
                    }
                }
                if (BMODE == 1) {       // K index of MFMA step kk on lane half lhi is 16*lhi + kk (as for A when AV): chunk 4*lhi + g
#pragma unroll
                    for (int j = 0; j < SN; ++j) {
                        const float4 v = *reinterpret_cast<const float4*>(Bb + j * 32 * B_LD + (((lhi * 4 + g) ^ bsw) * 4));
                        bv[buf][0][j] = v.x; bv[buf][1][j] = v.y; bv[buf][2][j] = v.z; bv[buf][3][j] = v.w;
                    }
                }
#pragma unroll
                for (int q = 0; q < FG; ++q) {
                    const int kk = g * FG + q;
                    if (!AV) {
#pragma unroll
                        for (int i = 0; i < SM; ++i) av[buf][q][i] = Ab[i * 32 * A_LD + kk * KS];
                    }
                    if (BMODE == 0) {
#pragma unroll
                        for (int j = 0; j < SN; ++j) bv[buf][q][j] = Bb[kk * KS * B_LD + j * 32];
                    }
                }
            };
            auto mfmas = [&](int g) {
#pragma unroll
                for (int q = 0; q < FG; ++q)
#pragma unroll
                    for (int i = 0; i < SM; ++i)
#pragma unroll
                        for (int j = 0; j < SN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][q][i], bv[g & 1][q][j], acc[i][j], 0, 0, 0);
            };
            fetch(0, 0);
            fetch(1, 1);
            mfmas(0);
            stage(cur ^ 1);                         // K-tile kt+1: registers -> the LDS buffer nobody reads now
            // the filter tile of K-tile kt+1 into the same free buffer; issued after the staging (whose wait on the register
            // loads would otherwise also wait for the DMA) and before the loads of K-tile kt+2 (so that the counted wait
            // at the end of the step can leave exactly those outstanding)
            if (BDMA && !ADMA) dma_b(min(kt + 1, last), cur ^ 1);
            fetch(2, 0);
            mfmas(1);
            issue_loads(min(kt + 2, last));         // K-tile kt+2 into the registers just drained
            fetch(3, 1);
            mfmas(2);
            mfmas(3);
#if SSC_UT_SGB == 1
            // interleave hint: one MFMA, then a few of the independent staging instructions
#pragma unroll
            for (int q = 0; q < NFG * FG * SM * SN; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x220, 1, 0);
            }
#endif
            if (BDMA) {
                // counted wait instead of __syncthreads(): its fence would wait vmcnt(0) and drain the register prefetch of
                // K-tile kt+2 with the DMA.  vmcnt retires in order and the A_ROWS gather loads are the last vector-memory
                // operations issued after the DMA on every path (the norm-table loads, when a source has a table, come before
                // them): at most A_ROWS outstanding means the DMA has landed.  lgkmcnt(0): this wave's ds_writes of the A tile.
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(ADMA ? 0 : A_ROWS) : "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();
            }
            cur ^= 1;
        }
    }

    // ---- epilogue ----
    ut_epilogue<BM, BN, WM, WN, SM, SN, 2 * (A_SZ + B_SZ)>(acc, d, smem, rowpix, part, ks, sk, slot, flags, ts_s, splitk, slab_base,
                                                          slab_stride, m0, n0, phase, M, g_sk_cfg);
}


// streaming 16-byte load (split-K slabs are read once)
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload4(const float4* p) {
    const nt_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// sum split-K slabs; applies the epilogue that the partial passes skipped
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int splitk,
                                   float* __restrict__ out, long count, int ldc, int Nn, int Nstore,
                                   const float* bias, int epi, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int col = (int)(i % ldc);
    if (col >= Nstore) return;   // columns the partial passes never wrote
    float v = 0.f;
    for (int s = 0; s < splitk; ++s) v += slabs[(long)s * slab_stride + i];
    if (bias != nullptr && col < Nn) v += bias[col];
    if (epi == 1) v = tanhf(v);
    else if (epi == 2) v = fmaxf(v, 0.2f * v);
    if (accumulate) v += out[i];
    out[i] = v;
}

// the same sum, four columns per thread (16-byte loads, four slabs in flight; the additions keep the scalar kernel's order,
// so both give the same bits): ldc, Nn, Nstore multiples of 4 and 16-byte aligned bases
__global__ __launch_bounds__(256) void slab_reduce4_kernel(const float4* __restrict__ slabs, long slab_stride4, int splitk,
                                                            float4* __restrict__ out, long count4, int ldc4, int Nn4,
                                                            int Nstore4, const float4* bias, int epi, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count4) return;
    const int col = (int)(i % ldc4);
    if (col >= Nstore4) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* p = slabs + i;
    int s = 0;
    for (; s + 4 <= splitk; s += 4) {
        const float4 a = ntload4(p), b = ntload4(p + slab_stride4),
                     c = ntload4(p + 2 * slab_stride4), e = ntload4(p + 3 * slab_stride4);
        p += 4 * slab_stride4;
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
        v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
    }
    for (; s < splitk; ++s, p += slab_stride4) {
        const float4 a = ntload4(p);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (bias != nullptr && col < Nn4) {
        const float4 b = bias[col];
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (epi == 1) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
    else if (epi == 2) { v.x = fmaxf(v.x, 0.2f * v.x); v.y = fmaxf(v.y, 0.2f * v.y); v.z = fmaxf(v.z, 0.2f * v.z); v.w = fmaxf(v.w, 0.2f * v.w); }
    if (accumulate) {
        const float4 o = out[i];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    out[i] = v;
}

// The slab sum of a conv whose output feeds a batch-statistics norm: the sum touches every output element once, so it takes
// the per-column sum and sum of squares on the way (rows of partials per block, folded by bn_stats_finalize) instead of a
// statistics pass of its own over the tensor.  A thread keeps one group of 4 columns (256 % (ldc / 4) == 0) and walks rows;
// the additions over the slabs keep slab_reduce4_kernel's order (same output bits).
__global__ __launch_bounds__(256) void slab_reduce4_stats_kernel(const float4* __restrict__ slabs, long slab_stride4, int splitk,
                                                                  float4* __restrict__ out, long rows, int cg,
                                                                  float* __restrict__ stat) {
    __shared__ float4 sh[2][256];
    const int R = 256 / cg;
    const int c = threadIdx.x % cg, rl = threadIdx.x / cg;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    for (long r = (long)blockIdx.x * R + rl; r < rows; r += (long)gridDim.x * R) {
        const long i = r * cg + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* p = slabs + i;
        int k = 0;
        for (; k + 4 <= splitk; k += 4) {
            const float4 a = ntload4(p), b = ntload4(p + slab_stride4), c2 = ntload4(p + 2 * slab_stride4),
                         e = ntload4(p + 3 * slab_stride4);
            p += 4 * slab_stride4;
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            v.x += c2.x; v.y += c2.y; v.z += c2.z; v.w += c2.w;
            v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        }
        for (; k < splitk; ++k, p += slab_stride4) {
            const float4 a = ntload4(p);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        out[i] = v;
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
    }
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < R; ++k) {
            const float4 a = sh[0][k * cg + c], b = sh[1][k * cg + c];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
        }
        float* sp = stat + (long)blockIdx.x * 2 * (4 * cg);
        *reinterpret_cast<float4*>(sp + 4 * c) = s;
        *reinterpret_cast<float4*>(sp + 4 * cg + 4 * c) = q;
    }
}

static bool slab_stats_ok(const ssc_conv_desc& d) {
    static int off = -1;        // SSC_SLAB_STATS=0: statistics by a pass of their own after the slab sum (A/B)
    if (off < 0) {
        const char* e = ssc_dev_getenv("SSC_SLAB_STATS");
        off = (e != nullptr && e[0] == '0') ? 1 : 0;
    }
    const int cg = d.ldc / 4;
    return !off && (d.ldc & 3) == 0 && cg >= 1 && cg <= 256 && (256 % cg) == 0 && d.Nn == d.ldc && d.Nstore == d.ldc &&
           d.bias == nullptr && d.epi == 0 && !d.accumulate && ((uintptr_t)d.out & 15) == 0;
}
// rows of partial sums the statistics form of the slab sum writes (one per block)
static int slab_stats_blocks(long rows, int ldc) {
    const int R = 256 / (ldc / 4);
    long b = ((rows + R - 1) / R + 3) / 4;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

static void launch_slab_reduce(const float* ws, long out_count, int splitk, const ssc_conv_desc& d, hipStream_t st) {
    const int thr = 256;
    if (d.stat_partial != nullptr && slab_stats_ok(d) && ((uintptr_t)ws & 15) == 0 && (out_count % 4) == 0) {
        const long rows = out_count / d.ldc;
        hipLaunchKernelGGL(slab_reduce4_stats_kernel, dim3((unsigned)slab_stats_blocks(rows, d.ldc)), dim3(256), 0, st,
                           (const float4*)ws, out_count / 4, splitk, (float4*)d.out, rows, d.ldc / 4, d.stat_partial);
        return;
    }
    const bool v4 = (out_count % 4) == 0 && (d.ldc % 4) == 0 && (d.Nn % 4) == 0 && (d.Nstore % 4) == 0 &&
                    (((uintptr_t)ws | (uintptr_t)d.out | (uintptr_t)d.bias) & 15) == 0;
    if (v4) {
        const long c4 = out_count / 4;
        hipLaunchKernelGGL(slab_reduce4_kernel, dim3((unsigned)((c4 + thr - 1) / thr)), dim3(thr), 0, st, (const float4*)ws,
                           c4, splitk, (float4*)d.out, c4, d.ldc / 4, d.Nn / 4, d.Nstore / 4, (const float4*)d.bias, d.epi,
                           d.accumulate);
    } else {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((out_count + thr - 1) / thr)), dim3(thr), 0, st, ws,
                           out_count, splitk, d.out, out_count, d.ldc, d.Nn, d.Nstore, d.bias, d.epi, d.accumulate);
    }
}

// igemm_bf16.hip sums its split-K slabs with the same kernels
void ssc_launch_slab_reduce(const float* ws, long out_count, int splitk, const ssc_conv_desc& d, hipStream_t st) {
    launch_slab_reduce(ws, out_count, splitk, d, st);
}

// ---------------------------------------------------------------------------------------------
// filter-gradient form
// ---------------------------------------------------------------------------------------------
// One K-tile step is a single branch-free block, as in conv_ut_kernel: K-tile kt+1 goes from registers to the free LDS
// buffer and K-tile kt+2 is loaded into the drained registers between the MFMAs of K-tile kt.  GPLAIN / DPLAIN: the
// gathered / dense view has no folded norm and no activation on any source (decided per launch).
template <int WM, int WN, int SM, int SN, bool GPLAIN, bool DPLAIN, bool DDMA>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const ssc_wgrad_desc d, const Magics mg,
                                                          float* __restrict__ slab_base, long slab_stride,
                                                          int splitk, int xcd) {
    constexpr int BM = WM * SM * 32;
    constexpr int BN = WN * SN * 32;
    constexpr int A_LD = BM;
    constexpr int B_LD = BN;
    constexpr int A_SZ = BK * BM;
    constexpr int B_SZ = BK * BN;
    constexpr int A_SLOTS = BM / 32;
    constexpr int B_SLOTS = BN / 32;
    constexpr int A_RP = 1024 / BM;   // pixel rows per pass
    constexpr int B_RP = 1024 / BN;
    // DDMA (dense side = one tensor without norm / activation, e.g. every dy): its [pixel][channel] tile goes to LDS by
    // LDS-DMA as the filter tiles of conv_ut_kernel do -- a wave instruction covers 256 / BN pixel rows
    constexpr int B_IPW = BK * BN / 1024;
    static_assert(!DDMA || DPLAIN, "DMA tiles cannot be transformed");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * A_SZ;
    // per-pixel decode of the K-tile after next, shared by the whole workgroup: {pixel index of (n, iy0, ix0) in the
    // gathered tensor, iy0, ix0} with iy0 = py*stride + ioff_y (tap offsets are per-thread constants)
    int4* ptab = reinterpret_cast<int4*>(smem + 2 * A_SZ + 2 * B_SZ);      // [2][BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int Cg = d.g.C0 + d.g.C1;
    const int Cd = d.d.C0 + d.d.C1;
    const int Mtot = d.TH * d.TW * Cg;
    const long P = (long)d.NB * d.PH * d.PW;
    const int PHW = d.PH * d.PW;
    // workgroup -> (row tile, column tile, K slice).  All tiles of a K slice read the same pixels (every (tap, channel) block of
    // the gathered side and every block of the dense side of that pixel range), so they should meet in ONE L2: with the plain
    // grid order a slice's tiles are dealt round-robin to the 8 XCDs and each of them fetches the range.  xcd (host flag): ids
    // with the same id % 8 walk a contiguous run of (row tile, column tile, slice) order -- whole slices per XCD.
    int ks = blockIdx.z, mtile = blockIdx.x, ntile = blockIdx.y;
    if (xcd) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned t2 = (lin & 7u) * (total >> 3) + (lin >> 3);
        const unsigned per_slice = gridDim.x * gridDim.y;
        ks = (int)(t2 / per_slice);
        const unsigned r = t2 - (unsigned)ks * per_slice;
        ntile = (int)(r / gridDim.x);
        mtile = (int)(r - (unsigned)ntile * gridDim.x);
    }
    const int m0 = mtile * BM;
    const int n0 = ntile * BN;

    // fixed per-thread columns: (tap, channel) of the gathered side, channel of the dense side
    const int a_col = m0 + (tid % (BM / 4)) * 4;
    const bool a_cv = a_col < Mtot;
    const int a_cc = a_cv ? a_col : 0;
    const int a_tap = div32(a_cc, mg.mC, mg.oneC);
    const int a_c = a_cc - a_tap * Cg;
    const int a_ty = div32(a_tap, mg.mTW, mg.oneTW), a_tx = a_tap - a_ty * d.TW;
    float4 aa, ab;
    gview_affine4(d.g, a_c, aa, ab);
    const float* a_base;
    int a_cs;
    gview_src(d.g, a_c, a_base, a_cs);
    const int a_act = (a_c >= d.g.C0 && d.g.act1 >= 0) ? d.g.act1 : d.g.act;
    const int b_col = n0 + (tid % (BN / 4)) * 4;
    const bool b_cv = b_col < Cd;
    const int b_c = b_cv ? b_col : 0;
    float4 ba, bb;
    gview_affine4(d.d, b_c, ba, bb);
    const float* b_base;
    int b_cs;
    gview_src(d.d, b_c, b_base, b_cs);
    const int b_act = (b_c >= d.d.C0 && d.d.act1 >= 0) ? d.d.act1 : d.d.act;

    // 32-bit K-tile arithmetic (the host checks P < 2^31 and every tensor < 2^29 elements: byte offsets fit 32 bits)
    const int nkt = (int)((P + BK - 1) / BK);
    const int per = (nkt + splitk - 1) / splitk;
    const int kt_begin = ks * per;
    const int kt_end = min(nkt, kt_begin + per);

    f32x16 acc[SM][SN];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_SLOTS], rb[B_SLOTS];
    float rav[A_SLOTS];
    const float g_slope = act_slope(a_act), d_slope = act_slope(b_act);

    // one thread per pixel of K-tile `kt` fills ptab[kt & 1] (threads 0..BK-1): {byte offset of (n, iy0, ix0) in source 0,
    // iy0, ix0, byte offset in source 1} with iy0 = py*stride + ioff_y (tap offsets are per-thread constants)
    const int gC0b = d.g.C0 * 4, gC1b = d.g.C1 * 4;
    const unsigned Pu = (unsigned)P;
    auto fill_ptab = [&](int kt) {
        // the waves take turns (K-tile kt is decoded by wave kt & 3): the decode is ~35 vector instructions that only
        // 32 lanes need, and a wave that did it every step would be the one the barrier waits for
        if ((wave == (kt & 3)) & (lane < BK)) {
            const unsigned p = (unsigned)kt * BK + lane;
            const bool pv = p < Pu;
            const long pp = pv ? (long)p : 0;
            int n, py, px;
            if (mg.use32) {     // wave-uniform: numerators fit the 32-bit multiply-high
                n = (int)__umulhi((unsigned)pp, mg.mPHPW32) + (int)((unsigned)pp & (unsigned)mg.onePHPW);
                const int rem = (int)pp - n * PHW;
                py = (int)__umulhi((unsigned)rem, mg.mPW32) + (int)((unsigned)rem & (unsigned)mg.onePW);
                px = rem - py * d.PW;
            } else {
                n = (int)div64(pp, mg.mPHPW, mg.onePHPW);
                const int rem = (int)(pp - (long)n * PHW);
                py = (int)div64(rem, mg.mPW, mg.onePW);
                px = rem - py * d.PW;
            }
            const int iy0 = py * d.in_stride + d.ioff_y, ix0 = px * d.in_stride + d.ioff_x;
            const int pix = (n * d.g.H + iy0) * d.g.W + ix0;
            // an invalid pixel gets coordinates no tap can bring inside the image
            ptab[(kt & 1) * BK + lane] = make_int4(pix * gC0b, pv ? iy0 : -(1 << 20), ix0, pix * gC1b);
        }
    };
    // per-thread constants of the gathered side: which ptab word holds this thread's source offset, the tap's byte shift
    const bool a_first = a_c < d.g.C0;
    const int a_tapb = (a_ty * d.g.W + a_tx) * a_cs * 4;
    const char* const a_bytes = reinterpret_cast<const char*>(a_base);
    const int gH = d.g.H, gW = d.g.W;
    // dense side: rows beyond the last pixel are CLAMPED to it, not masked -- the gathered side of those rows is staged as
    // zeros (ptab marks them invalid), so what they hold never reaches a sum; every load is unconditional and unmasked
    const char* const b_bytes = reinterpret_cast<const char*>(b_base);
    const unsigned b_rowb = (unsigned)b_cs * 4u;
    const unsigned b_last = (unsigned)(P - 1) * b_rowb;
    unsigned b_off[B_SLOTS];
#pragma unroll
    for (int s = 0; s < B_SLOTS; ++s) b_off[s] = (unsigned)(tid / (BN / 4) + B_RP * s) * b_rowb;
    const unsigned b_step = (unsigned)BK * b_rowb;
    unsigned d_row[B_IPW];      // DDMA: byte offset of this lane's pixel row inside a K-tile, per instruction of its wave
    const unsigned d_col = (n0 + (lane % (BN / 4)) * 4) < Cd ? (unsigned)((n0 + (lane % (BN / 4)) * 4) * 4) : 0u;
#pragma unroll
    for (int q = 0; q < B_IPW; ++q) d_row[q] = (unsigned)((wave * B_IPW + q) * (256 / BN) + lane / (BN / 4)) * b_rowb;
    auto dma_d = [&](int kt, int buf) {
        const unsigned kb = (unsigned)kt * b_step;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)((2 * A_SZ + buf * B_SZ + wave * B_IPW * 256) * 4));
#pragma unroll
        for (int q = 0; q < B_IPW; ++q)
            glds16(reinterpret_cast<const char*>(d.d.s0), min(kb + d_row[q], b_last) + d_col, dst + q * 1024);
    };
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) {
            const int4 e = ptab[(kt & 1) * BK + tid / (BM / 4) + A_RP * s];
            const int iy = e.y + a_ty, ix = e.z + a_tx;
            const bool v = a_cv & ((unsigned)iy < (unsigned)gH) & ((unsigned)ix < (unsigned)gW);
            const unsigned off = v ? (unsigned)((a_first ? e.x : e.w) + a_tapb) : 0u;
            rav[s] = v ? 1.f : 0.f;
            ra[s] = *reinterpret_cast<const float4*>(a_bytes + off);
        }
        if (DDMA) return;
        const unsigned kb = (unsigned)kt * b_step;
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            const unsigned off = min(kb + b_off[s], b_last);
            rb[s] = *reinterpret_cast<const float4*>(b_bytes + (b_cv ? off : 0u));
        }
    };
    auto store_tile = [&](int buf) {
        float* Ab = As + buf * A_SZ;
        float* Bb = Bs + buf * B_SZ;
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) {
            const int row = tid / (BM / 4) + A_RP * s;
            *reinterpret_cast<float4*>(Ab + row * A_LD + (tid % (BM / 4)) * 4) =
                GPLAIN ? mask4(ra[s], rav[s]) : xform4(ra[s], aa, ab, g_slope, rav[s]);
        }
        if (DDMA) return;
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            const int row = tid / (BN / 4) + B_RP * s;
            *reinterpret_cast<float4*>(Bb + row * B_LD + (tid % (BN / 4)) * 4) =
                DPLAIN ? rb[s] : xform4(rb[s], ba, bb, d_slope, 1.f);
        }
    };

    const int l31 = lane & 31, lhi = lane >> 5;
    if (kt_begin < kt_end) {
        // K-tiles past kt_end (another split's, or past the last pixel: every row masked) are staged but never multiplied
        if (DDMA) dma_d(kt_begin, 0);
        fill_ptab(kt_begin);
        fill_ptab(kt_begin + 1);
        __syncthreads();
        load_tile(kt_begin);
        store_tile(0);
        load_tile(kt_begin + 1);
        if (DDMA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_SLOTS) : "memory");      // the first dense tile has landed
        __syncthreads();                    // every thread has read ptab slots kt_begin, kt_begin + 1
        fill_ptab(kt_begin + 2);
        __syncthreads();
        int cur = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const float* Ab = As + cur * A_SZ + lhi * A_LD + wm * SM * 32 + l31;
            const float* Bb = Bs + cur * B_SZ + lhi * B_LD + wn * SN * 32 + l31;
            constexpr int FG = 4, NFG = BK / 2 / FG;
            float av[2][FG][SM], bv[2][FG][SN];
            auto fetch = [&](int g, int buf) {
#pragma unroll
                for (int q = 0; q < FG; ++q) {
                    const int kk = g * FG + q;
#pragma unroll
                    for (int i = 0; i < SM; ++i) av[buf][q][i] = Ab[kk * 2 * A_LD + i * 32];
#pragma unroll
                    for (int j = 0; j < SN; ++j) bv[buf][q][j] = Bb[kk * 2 * B_LD + j * 32];
                }
            };
            auto mfmas = [&](int g) {
#pragma unroll
                for (int q = 0; q < FG; ++q)
#pragma unroll
                    for (int i = 0; i < SM; ++i)
#pragma unroll
                        for (int j = 0; j < SN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][q][i], bv[g & 1][q][j], acc[i][j], 0, 0, 0);
            };
            fill_ptab(kt + 3);              // overwrites ptab[(kt+1)&1], last read before the previous barrier; its
                                            // lane-divergent branch stays in front of the MFMA block, not inside it
            fetch(0, 0);
            fetch(1, 1);
            mfmas(0);
            store_tile(cur ^ 1);            // K-tile kt+1: registers -> the LDS buffer nobody reads now
            if (DDMA) dma_d(kt + 1, cur ^ 1);       // its dense tile: DMA into the same free buffer (see conv_ut_kernel)
            fetch(2, 0);
            mfmas(1);
            load_tile(kt + 2);              // reads ptab[(kt+2)&1], published by the previous barrier
            fetch(3, 1);
            mfmas(2);
            mfmas(3);
#if SSC_UT_SGB
#pragma unroll
            for (int q = 0; q < NFG * FG * SM * SN; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x220, 1, 0);
            }
#endif
            if (DDMA) {     // counted wait + bare barrier: the gathered loads of K-tile kt+2 stay in flight (conv_ut_kernel)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(A_SLOTS) : "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();
            }
            cur ^= 1;
        }
    }

    float* outp = (splitk > 1) ? (slab_base + (long)ks * slab_stride) : d.out;
#pragma unroll
    for (int i = 0; i < SM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * SM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m >= Mtot) continue;
            const int tap = div32(m, mg.mC, mg.oneC);
            const int c = m - tap * Cg;
            if (c >= d.Cg_real) continue;
            const long orow = (long)tap * d.Cg_real + c;
#pragma unroll
            for (int j = 0; j < SN; ++j) {
                const int col = n0 + wn * SN * 32 + j * 32 + l31;
                if (col >= d.Nn) continue;
                float v = acc[i][j][r];
                float* o = outp + orow * d.ldc + col;
                if (splitk == 1 && d.accumulate) v += *o;
                *o = v;
            }
        }
    }
}

// Sum of the split-K slabs of a filter gradient.  Small filters (1x1 / 3x3 convs of the residual blocks: 10^4-10^5
// elements) are split hundreds of ways over the pixels, so one thread per element would leave a few dozen workgroups
// walking hundreds of slabs each: L threads share an element (slabs s, s+L, ...) and combine through LDS in a fixed order.
template <int L>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int splitk,
                                                           float* __restrict__ out, long count, int accumulate) {
    constexpr int EPB = 256 / L;        // elements per block
    __shared__ float part[256];
    const int e = threadIdx.x % EPB, ls = threadIdx.x / EPB;
    const long i = (long)blockIdx.x * EPB + e;
    float v = 0.f;
    if (i < count)
        for (int s = ls; s < splitk; s += L) v += slabs[(long)s * slab_stride + i];
    if (L > 1) {
        part[threadIdx.x] = v;
        __syncthreads();
        if (ls == 0)
            for (int l = 1; l < L; ++l) v += part[l * EPB + e];
    }
    if (ls == 0 && i < count) {
        if (accumulate) v += out[i];
        out[i] = v;
    }
}

// one thread per four elements: 16-byte loads, four slabs in flight, the scalar kernel's order of additions (same bits)
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float4* __restrict__ slabs, long slab_stride4, int splitk,
                                                             float4* __restrict__ out, long count4, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* p = slabs + i;
    int s = 0;
    for (; s + 4 <= splitk; s += 4) {
        const float4 a = ntload4(p), b = ntload4(p + slab_stride4),
                     c = ntload4(p + 2 * slab_stride4), e = ntload4(p + 3 * slab_stride4);
        p += 4 * slab_stride4;
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
        v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
    }
    for (; s < splitk; ++s, p += slab_stride4) {
        const float4 a = ntload4(p);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (accumulate) {
        const float4 o = out[i];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    out[i] = v;
}

static void launch_wgrad_reduce(const float* ws, long count, int splitk, float* out, int accumulate, hipStream_t st) {
    static int skip = -1;       // SSC_DIAG_SKIP_WGRAD_REDUCE=1: timing diagnostic only (wrong gradients): what do the slab sums cost
    if (skip < 0) {
        const char* e = ssc_dev_getenv("SSC_DIAG_SKIP_WGRAD_REDUCE");
        skip = (e != nullptr && e[0] == '1') ? 1 : 0;
    }
    if (skip) return;
    if ((count >= 262144 || splitk < 8) && (count % 4) == 0 && (((uintptr_t)ws | (uintptr_t)out) & 15) == 0) {
        const long c4 = count / 4;
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((c4 + 255) / 256)), dim3(256), 0, st, (const float4*)ws, c4,
                           splitk, (float4*)out, c4, accumulate);
    } else if (count >= 262144 || splitk < 8) {
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ws, count,
                           splitk, out, count, accumulate);
    } else if (count >= 65536 || splitk < 32) {
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, st, ws, count, splitk,
                           out, count, accumulate);
    } else {
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((count + 15) / 16)), dim3(256), 0, st, ws, count, splitk,
                           out, count, accumulate);
    }
}

// wgrad128.hip sums its split-K slabs with the same kernels
void ssc_launch_wgrad_reduce(const float* ws, long count, int splitk, float* out, int accumulate, hipStream_t st) {
    launch_wgrad_reduce(ws, count, splitk, out, accumulate, st);
}

#ifdef SSC_ISA_ONLY
// scripts/isa_one.sh: compile a single instantiation to look at its ISA (the whole file takes over a minute)
template __global__ void conv_ut_kernel<2, 2, 1, 2, 0, false, 0>(const ssc_conv_desc, const Magics, float*, long, int, int, int, unsigned*);
template __global__ void conv_ut_kernel<2, 2, 1, 2, 1, true, 0>(const ssc_conv_desc, const Magics, float*, long, int, int, int, unsigned*);
template __global__ void conv_wgrad_kernel<1, 4, 2, 1, false, true, true>(const ssc_wgrad_desc, const Magics, float*, long, int, int);
#else
// ---------------------------------------------------------------------------------------------
// host launchers (C ABI)
// ---------------------------------------------------------------------------------------------
static int num_cu() { return ssc_num_cu(); }

// Tile configurations.  `res` = workgroups resident per CU (LDS-limited), `penalty` = relative cost of the
// staging instructions per MFMA (fp32 MFMA does not overlap VALU on gfx950).
struct TileCfg { int id, BM, BN, res; double penalty; };
// penalties from the measured instruction mix: ~16 VALU per staged A row-float4 (bounds + transform), ~3 per
// filter float4, 4 cycles each, against 64 cycles per MFMA: (MFMA + VALU) / MFMA, normalised to 128x128
static const TileCfg FWD_CFGS[5] = {{0, 128, 128, 2, 1.00}, {1, 64, 128, 3, 1.03}, {2, 128, 64, 3, 1.11}, {3, 128, 32, 3, 1.33},
                                    {4, 64, 64, 4, 1.30}};    // 64x64: small-M GEMMs (LSTM steps) that leave CUs under-filled
// the bf16-split form (igemm_bf16.hip): larger LDS images (fewer resident workgroups), the same ids; 128x32 is not built
static const TileCfg BF_CFGS[5] = {{0, 128, 128, 1, 1.05}, {1, 64, 128, 2, 1.00}, {2, 128, 64, 2, 1.12}, {3, 128, 32, 1, 9.9},
                                   {4, 64, 64, 3, 1.40}};     // measured on the batch-32 layers: 64x128 ahead of 128x128 / 128x64 by ~9 %, 64x64 behind by ~15 %
static const TileCfg WG_CFGS[5] = {{0, 128, 128, 2, 1.00}, {1, 64, 128, 3, 1.00}, {2, 128, 64, 3, 1.10}, {3, 64, 64, 4, 1.2},
                                   {4, 128, 32, 4, 1.3}};

// Makespan model of one launch: `blocks` equal workgroups of `w` MFMA-cycles each on ncu CUs that hold `res`
// of them at a time and are matrix-pipe bound (co-resident workgroups share the pipe).  Split-K by s divides w
// and multiplies the block count, at the price of writing + re-reading s partial copies of the output.
// planner constants, overridable from the environment while tuning (read once)
static double plan_const(const char* name, double dflt) {
    static const char* names[16];
    static double vals[16];
    static int n = 0;
    for (int i = 0; i < n; ++i)
        if (names[i] == name) return vals[i];
    const char* e = getenv(name);
    const double v = (e != nullptr) ? atof(e) : dflt;
    if (n < 16) { names[n] = name; vals[n] = v; ++n; }
    return v;
}

static double makespan(long blocks, double w, int res, int ncu) {
    const long slots = (long)ncu * res;
    const long full = blocks / slots, rem = blocks % slots;
    double t = (double)(full * res + (rem + ncu - 1) / ncu) * w;
    // a lone wave per SIMD cannot hide its own LDS / barrier / load waits (measured: 1 workgroup per CU runs
    // ~25 % below 3 per CU at equal work)
    const long per_cu = (blocks + ncu - 1) / ncu;
    const long occ = per_cu < res ? per_cu : res;
    if (occ <= 1) t *= plan_const("SSC_PLAN_OCC1", 1.30);
    else if (occ == 2) t *= plan_const("SSC_PLAN_OCC2", 1.08);
    return t;
}

struct Plan { int cfg; int splitk; double cost; long ts_full; int ts_s; };

static Plan plan_launch(const TileCfg* cfgs, int ncfg, const bool* allowed, long M, long N, long nphase, long nkt,
                        long out_elems, int64_t ws_bytes, bool have_ws, bool can_ts = false, double cyc_scale = 1.0) {
    const int ncu = num_cu();
    Plan best = {-1, 1, 1e300, 0, 1};
    static int force = -2;      // SSC_FWD_CFG=n: tuning aid, restricts the search to tile configuration n where allowed
    if (force == -2) {
        const char* e = getenv("SSC_FWD_CFG");
        force = (e != nullptr) ? atoi(e) : -1;
    }
    for (int c = 0; c < ncfg; ++c) {
        if (!allowed[c]) continue;
        if (force >= 0 && c != force && allowed[force]) continue;
        const TileCfg& t = cfgs[c];
        const long mt = (M + t.BM - 1) / t.BM, nt = (N + t.BN - 1) / t.BN;
        const long blocks = mt * nt * nphase;
        const double wfull = (double)nkt * 16.0 * (t.BM / 32) * (t.BN / 32) / 4.0 * 64.0 * t.penalty * cyc_scale;  // cycles
        for (int sk = 1; sk <= 16; ++sk) {
            if (sk > 1 && (!have_ws || nkt / sk < 4 || (int64_t)sk * out_elems * 4 > ws_bytes)) break;
            const long per = (nkt + sk - 1) / sk;
            if ((nkt + per - 1) / per != sk) continue;      // would leave empty trailing splits
            double cost = makespan(blocks * sk, wfull * (double)per / (double)nkt, t.res, ncu) + 2500.0;
            if (sk > 1) cost += 2.0 * sk * (double)out_elems * 4.0 / 1500.0 + plan_const("SSC_PLAN_REDUCE", 6000.0);   // slab traffic + reduce launch
            const long bf = blocks;
            const int bs = 1;
            if (cost < best.cost) best = {c, sk, cost, bf, bs};
        }
    }
    // Whole tiles + K slices combined inside the launch (launch_fwd_ut).  Measured on the batch-32 layers with 64x128 tiles
    // (scripts/ts_sweep.sh): the best layout is as many whole tiles as fill whole rounds of one workgroup per CU, the rest
    // cut into about (CUs / remaining tiles) slices, i.e. one more workgroup per CU -- encoder_3 (576 tiles) 97 -> 115
    // TFLOP/s as 512 + 64 x 4, encoder_2 (1152) 105 -> 115 as 1024 + 128 x 2.  Slicing more tiles than that loses to the
    // fixed cost of a workgroup (first loads, epilogue, hand-off), split-K slabs + reduce kernel lose to both.
    if (can_ts && have_ws && best.cfg >= 0) {
        const TileCfg& t = cfgs[best.cfg];
        const long blocks = ((M + t.BM - 1) / t.BM) * ((N + t.BN - 1) / t.BN) * nphase;
        long full = (blocks / ncu) * ncu, tail = blocks - full;
        long sl = 1;
        static int ff = -2, fs = -2;        // tuning aid: SSC_TS_FORCE="whole tiles per CU,slices" pins the layout
        if (ff == -2) {
            const char* e = getenv("SSC_TS_FORCE");
            ff = fs = -1;
            if (e != nullptr) sscanf(e, "%d,%d", &ff, &fs);
        }
        if (ff >= 0 && fs >= 1) {
            full = (long)ff * ncu;
            if (full > blocks) full = (blocks / ncu) * ncu;
            tail = blocks - full;
            sl = tail > 0 ? fs : 1;
        } else if (full > 0 && tail > 0 && tail * 4 <= (long)ncu * 3) {
            sl = (ncu + tail / 2) / tail;
            const long smax = (long)plan_const("SSC_TS_MAXS", 8.0);
            if (sl > smax) sl = smax;
            while (sl > 1 && nkt / sl < 4) --sl;
        }
        if (sl > 1 && tail * sl < SSC_SK_FLAG_WORDS - 1 && (int64_t)tail * sl * t.BM * t.BN * 4 <= ws_bytes) {
            best.splitk = 1;
            best.ts_full = full;
            best.ts_s = (int)sl;
        }
    }
    return best;
}

template <int WM, int WN, int SM, int SN, int BMODE, bool VECB>
static int launch_fwd_v(const ssc_conv_desc& d, int splitk, float* ws, hipStream_t st) {
    constexpr int BM = WM * SM * 32, BN = WN * SN * 32;
    constexpr int A_SZ = BM * (BK + 1);
    constexpr int B_SZ = (BMODE == 0) ? BK * BN : BN * (BK + 1);
    constexpr size_t lds = 2 * (A_SZ + B_SZ) * sizeof(float) + BM * sizeof(long);
    const long M = (long)d.NB * d.PH * d.PW;
    const int C = d.x.C0 + d.x.C1;
    const Magics mg = make_magics((unsigned)C, (unsigned)d.TW, (unsigned long)d.PW, (unsigned long)d.PH * d.PW,
                                  (unsigned long)M);
    const long mt = (M + BM - 1) / BM;
    const int nt = (d.Nstore + BN - 1) / BN;
    const long out_count = (long)d.NB * d.OH * d.OW * d.ldc;
    static unsigned long long attr_done = 0;
    {
        const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&conv_fwd_kernel<WM, WN, SM, SN, BMODE, VECB>), (int)lds, &attr_done);
        if (arc != 0) return arc;
    }
    dim3 grid((unsigned)mt, (unsigned)nt, (unsigned)(d.nphase * splitk));
    hipLaunchKernelGGL((conv_fwd_kernel<WM, WN, SM, SN, BMODE, VECB>), grid, dim3(256), lds, st, d, mg, ws,
                       out_count, splitk);
    if (splitk > 1) {
        launch_slab_reduce(ws, out_count, splitk, d, st);
    }
    return (int)hipGetLastError();
}

// set by ssc_conv_forward for the launch it is about to make: resident workgroups per CU of the chosen tile, workspace size
static thread_local int g_launch_res = 2;
static thread_local int64_t g_launch_ws_bytes = 0;
static thread_local long g_launch_ts_full = 0;      // the planner's tail split: whole tiles, slices per remaining tile
static thread_local int g_launch_ts_s = 1;
static int tail_split_mode() {
    static int mode = -1;
    if (mode < 0) {
        const char* e = ssc_dev_getenv("SSC_TAIL_SPLIT");
        mode = (e != nullptr) ? atoi(e) : 1;
    }
    return mode;
}

// float4 filter loads need 16-byte aligned, fully in-range groups of 4
static bool fwd_is_vec(const ssc_conv_desc& d) {
    return (d.bmode == 0) ? (((d.wC1 | d.n_off) & 3) == 0 && (d.Nn & 3) == 0) : ((d.wC1 & 3) == 0 && (d.k_real & 3) == 0);
}

// uniform-tap fast path: every 32-wide K-tile inside one tap and one source, no channel padding
static bool fwd_is_ut(const ssc_conv_desc& d) {
    const bool vec = fwd_is_vec(d);
    const int C = d.x.C0 + d.x.C1;
    return vec && (C % BK) == 0 && (d.x.C0 % BK) == 0 && d.k_real == C &&
           (long)d.NB * d.x.H * d.x.W * (d.x.C0 > d.x.C1 ? d.x.C0 : d.x.C1) < 0x1fffffffL &&
           (long)d.KH * d.KW * d.wC0 * d.wC1 < 0x1fffffffL;
}

// row-tap form (conv_ut_kernel<KM = 2>): plain single-source view, TW * C == 32, unflipped taps, vector filter loads
static bool fwd_is_rowtap(const ssc_conv_desc& d) {
    static int off = -1;
    if (off < 0) {
        const char* e = ssc_dev_getenv("SSC_ROWTAP");
        off = (e != nullptr && e[0] == '0') ? 1 : 0;
    }
    return !off && d.bmode == 0 && d.nphase == 1 && d.kstep == 1 && d.kx0 == 0 && d.x.C1 == 0 && d.TW * d.x.C0 == BK &&
           d.TW == d.KW && d.k_real == d.x.C0 && d.wC0 == d.x.C0 && fwd_is_vec(d) && d.x.ab0 == nullptr &&
           d.x.act == SSC_ACT_NONE && (long)d.NB * d.x.H * d.x.W * d.x.C0 < 0x1fffffffL &&
           (long)d.KH * d.KW * d.wC0 * d.wC1 < 0x1fffffffL;
}

// chunked uniform-tap form (conv_ut_kernel<KMASK>): vector filter loads possible and at most a third of the K-tiles' width
// wasted on the partly empty last chunk of each source (padded <= 1.5 x real)
static bool fwd_is_utg(const ssc_conv_desc& d) {
    const bool vec = fwd_is_vec(d);
    const int C = d.x.C0 + d.x.C1;
    const int padded = ((d.x.C0 + BK - 1) / BK + (d.x.C1 + BK - 1) / BK) * BK;
    static int off = -1;
    if (off < 0) {
        const char* e = ssc_dev_getenv("SSC_UTG");
        off = (e != nullptr && e[0] == '0') ? 1 : 0;
    }
    static double waste = -1.0;     // SSC_UTG_WASTE: largest padded / real K width taken (tuning aid)
    if (waste < 0.0) {
        const char* e = ssc_dev_getenv("SSC_UTG_WASTE");
        waste = (e != nullptr) ? atof(e) : 1.5;     // measured 1.2 / 1.25 / 1.5: MRU train 258.9 / 257.8 / 256.9 ms, MRU forward 13.47 / - / 13.20 ms
    }
    // ... or no more K-tiles than the generic kernel's walk over taps * C would take (few channels: one chunk per tap either way)
    const long kt_chunks = (long)d.TH * d.TW * (padded / BK), kt_flat = ((long)d.TH * d.TW * C + BK - 1) / BK;
    return !off && vec && d.k_real >= 1 && d.k_real <= C && ((double)padded <= waste * (double)C || kt_chunks <= kt_flat) &&
           (long)d.NB * d.x.H * d.x.W * (d.x.C0 > d.x.C1 ? d.x.C0 : d.x.C1) < 0x1fffffffL &&
           (long)d.KH * d.KW * d.wC0 * d.wC1 < 0x1fffffffL;
}

static int xcd_order() {
    static int on = -1;
    if (on < 0) {
        const char* e = ssc_dev_getenv("SSC_XCD_ORDER");
        on = (e != nullptr) ? atoi(e) : 2;     // 0: off, 1: XCD-aware order, 2: + the 4 phases of a row tile adjacent
    }
    return on;
}

template <int WM, int WN, int SM, int SN, int BMODE, bool PLAIN, int KM>
static int launch_fwd_ut(const ssc_conv_desc& d, int splitk, float* ws, hipStream_t st) {
    constexpr int BM = WM * SM * 32, BN = WN * SN * 32;
    constexpr int A_SZ = BM * (UT_AV(BM, BN) ? BK + 4 : BK + 1);
    constexpr int B_SZ = BK * BN;
    constexpr size_t lds = 2 * (A_SZ + B_SZ) * sizeof(float) + BM * sizeof(long);
    const long M = (long)d.NB * d.PH * d.PW;
    const int C = d.x.C0 + d.x.C1;
    const int tpt = (KM == 2) ? 1 : (d.x.C0 + BK - 1) / BK + (d.x.C1 + BK - 1) / BK;   // K-tiles per tap (mg.mC divides by it)
    const Magics mg = make_magics((unsigned)tpt, (unsigned)d.TW, (unsigned long)d.PW, (unsigned long)d.PH * d.PW,
                                  (unsigned long)M);
    const long mt = (M + BM - 1) / BM;
    const int nt = (d.Nstore + BN - 1) / BN;
    const long out_count = (long)d.NB * d.OH * d.OW * d.ldc;
    static unsigned long long attr_done = 0;
    {
        const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&conv_ut_kernel<WM, WN, SM, SN, BMODE, PLAIN, KM>), (int)lds, &attr_done);
        if (arc != 0) return arc;
    }
    // whole tiles + K slices combined inside the launch, as the planner laid them out (plan_launch / cu_timeline)
    if (splitk == 1 && ws != nullptr && d.sk_flags != nullptr && g_launch_ts_s > 1) {
        const long tiles = mt * nt * d.nphase;
        const long full = g_launch_ts_full, tail = tiles - full, s = g_launch_ts_s;
        if (full >= 0 && tail > 0 && (int64_t)tail * s * BM * BN * 4 <= g_launch_ws_bytes && tail * s < SSC_SK_FLAG_WORDS - 1 &&
            full + tail * s < 0x7fffffffL) {
            hipLaunchKernelGGL((conv_ut_kernel<WM, WN, SM, SN, BMODE, PLAIN, KM>), dim3((unsigned)(full + tail * s)),
                               dim3(256), lds, st, d, mg, ws, out_count, 1, (int)full,
                               (int)s | ((xcd_order() && (full & 7) == 0) ? (0x10000 | ((xcd_order() >= 2 && d.nphase == 4) ? 0x20000 : 0)) : 0),
                               d.sk_flags);
            return (int)hipGetLastError();
        }
    }
    if (splitk == 1 && xcd_order()) {        // whole tiles only, 1-D grid in the XCD-aware order (no flags needed)
        const long tiles = mt * nt * d.nphase;
        const long full = tiles & ~7L;
        if (tiles < 0x7fffffffL && full > 0) {
            hipLaunchKernelGGL((conv_ut_kernel<WM, WN, SM, SN, BMODE, PLAIN, KM>), dim3((unsigned)tiles), dim3(256), lds, st, d,
                               mg, ws, out_count, 1, (int)full, 1 | 0x10000 | ((xcd_order() >= 2 && d.nphase == 4) ? 0x20000 : 0),
                               (unsigned*)nullptr);
            return (int)hipGetLastError();
        }
    }
    dim3 grid((unsigned)mt, (unsigned)nt, (unsigned)(d.nphase * splitk));
    hipLaunchKernelGGL((conv_ut_kernel<WM, WN, SM, SN, BMODE, PLAIN, KM>), grid, dim3(256), lds, st, d, mg, ws, out_count,
                       splitk, 0, 0, (unsigned*)nullptr);
    if (splitk > 1) {
        launch_slab_reduce(ws, out_count, splitk, d, st);
    }
    return (int)hipGetLastError();
}

template <int WM, int WN, int SM, int SN, int BMODE>
static int launch_fwd(const ssc_conv_desc& d, int splitk, float* ws, hipStream_t st) {
    const bool vec = fwd_is_vec(d);
    const bool ut = fwd_is_ut(d);
    if (ut || fwd_is_utg(d)) {
        const bool plain0 = d.x.ab0 == nullptr && d.x.act == SSC_ACT_NONE;
        const bool plain1 = d.x.C1 == 0 || (d.x.ab1 == nullptr && (d.x.act1 >= 0 ? d.x.act1 : d.x.act) == SSC_ACT_NONE);
        const bool plain = plain0 && plain1;
        if (ut) return plain ? launch_fwd_ut<WM, WN, SM, SN, BMODE, true, 0>(d, splitk, ws, st)
                             : launch_fwd_ut<WM, WN, SM, SN, BMODE, false, 0>(d, splitk, ws, st);
        return plain ? launch_fwd_ut<WM, WN, SM, SN, BMODE, true, 1>(d, splitk, ws, st)
                     : launch_fwd_ut<WM, WN, SM, SN, BMODE, false, 1>(d, splitk, ws, st);
    }
    if (BMODE == 0 && fwd_is_rowtap(d)) {       // a filter row per K-tile: run as TH taps of 32 "channels"
        ssc_conv_desc dr = d;
        dr.TW = 1;
        return launch_fwd_ut<WM, WN, SM, SN, 0, true, 2>(dr, splitk, ws, st);
    }
    return vec ? launch_fwd_v<WM, WN, SM, SN, BMODE, true>(d, splitk, ws, st)
               : launch_fwd_v<WM, WN, SM, SN, BMODE, false>(d, splitk, ws, st);
}

// bf16-split form (igemm_bf16.hip): the caller supplied the filter's planes, every K-tile lies inside one tap and one source,
// the column range starts at a 32-column block.  SSC_ARITH=fp32 keeps everything on the exact-fp32 MFMA.
static bool fwd_is_bf(const ssc_conv_desc& d) {
    static int off = -1;
    if (off < 0) {
        const char* e = getenv("SSC_ARITH");
        off = (e != nullptr && (e[0] == 'f' || e[0] == 'F')) ? 1 : 0;
    }
    // ... or ONE source whose channel count is any multiple of 4 above 32 (MRU's materialised concats: state + image channels),
    // the partly empty last chunk of every tap masked in the staging (conv_bf_kernel<KM>); the planes pad k >= K with zeros
    const bool km = d.x.C1 == 0 && (d.x.C0 & 3) == 0 && d.x.C0 > BK && (d.x.C0 % BK) != 0 && d.k_real >= 1 && d.k_real <= d.x.C0 &&
                    d.ws_kc == 2 * ((d.x.C0 + BK - 1) / BK) && (long)d.NB * d.x.H * d.x.W * d.x.C0 < 0x1fffffffL &&
                    (long)d.KH * d.KW * d.wC0 * d.wC1 < 0x1fffffffL;
    return !off && d.wsplit != nullptr && d.ws_kc > 0 && d.ws_nbp > 0 && (fwd_is_ut(d) || km) && (d.n_off & 31) == 0 && d.Nstore > 32 &&
           d.TH * d.TW <= 32;
}

static Plan plan_fwd(const ssc_conv_desc& d, int64_t ws_bytes, bool have_ws) {
    const long M = (long)d.NB * d.PH * d.PW;
    const int C = d.x.C0 + d.x.C1;
    long nkt = ((long)d.TH * d.TW * C + BK - 1) / BK;
    if (fwd_is_ut(d) || fwd_is_utg(d))       // conv_ut_kernel walks whole chunks per tap and source
        nkt = (long)d.TH * d.TW * ((d.x.C0 + BK - 1) / BK + (d.x.C1 + BK - 1) / BK);
    else if (d.bmode == 0 && fwd_is_rowtap(d))
        nkt = d.TH;
    // column tile no wider than needed: <=32 -> 128x32, <=64 -> 128x64, else 128x128 / 64x128 / 128x64
    const bool allowed[5] = {d.Nstore > 64, d.Nstore > 64, d.Nstore > 32, d.Nstore <= 32, d.Nstore > 32};
    const bool can_ts = d.sk_flags != nullptr && tail_split_mode() != 0 &&
                        (fwd_is_ut(d) || fwd_is_utg(d) || (d.bmode == 0 && fwd_is_rowtap(d)));
    if (fwd_is_bf(d)) {
        const bool allowed_bf[5] = {d.Nstore > 64, d.Nstore > 64, true, false, true};
        // six bf16 passes = 6/16 of the fp32 MFMA's cycles; the staging beside them and the shorter K steps make it ~0.45
        // 128 x 128 on 16-k stages (round 6): two workgroups per CU, measured 3-10 % ahead of 64 x 128 on launches of whole
        // rounds (d4 +9.5 %, dec3 +6.5 %, enc2 +6 %); the uniform form only (the partial-chunk launches keep the 32-k kernel)
        TileCfg cfgs[5];
        for (int i = 0; i < 5; ++i) cfgs[i] = BF_CFGS[i];
        // (Also for launches that share the chip with other streams' -- lds_hint, the Pix2Pix train step -- although the 42 KB
        // one-stage 64 x 128 form is what lets three workgroups of different chains share a CU: 13.13 / 12.91 vs 13.00 / 13.05 ms
        // per step, equal.  The tile choice must not depend on the hint: it changes the summation order, and a trainer that
        // overlaps its chains must stay bit-identical with one that does not.)
        if (ssc_bf_hk_enabled()) {
            // ... for launches of at least SSC_PLAN_BF_HK_MIN x (2 x CUs) such tiles (below that the 64 x 128 tile's twice as many
            // workgroups balance better).  Swept over 0 / 0.5 / 1 / 2 / never (scripts/hk_min_sweep.sh, same box, twice): generator
            // inference at batch 16 1.325 / 1.300 / 1.300 / 1.308 / 1.293 ms, Background train step 22.3 / 21.8 / 22.1 / 22.1 / 22.0,
            // Pix2Pix train step 12.35 / 12.38 / 12.54 / 12.84 / 12.90, MRU 183.9 / 183.6 / 182.9 / 184.0 / 190.2
            const long t128 = ((M + 127) / 128) * ((d.Nstore + 127) / 128) * d.nphase;
            if ((double)t128 < plan_const("SSC_PLAN_BF_HK_MIN", 0.5) * 2.0 * num_cu()) {
                return plan_launch(cfgs, 5, allowed_bf, M, d.Nstore, d.nphase, nkt, (long)d.NB * d.OH * d.OW * d.ldc, ws_bytes,
                                   have_ws, can_ts, plan_const("SSC_PLAN_BF_SCALE", 0.45));
            }
            cfgs[0].res = 2;
            // 0.78: swept on the train steps (scripts/hk_plan_sweep.sh): Pix2Pix 12.57 -> 12.2 ms between 0.86 and 0.78 (the tile then
            // takes most whole-round launches), MRU / Residual / Background flat from 0.62 to 0.86
            cfgs[0].penalty = plan_const("SSC_PLAN_BF_HK", 0.78);
            // the partial-chunk form walks ceil(C / 16) chunks of 16 per tap there, 2 * ceil(C / 32) on the 32-k tiles
            if (!fwd_is_ut(d)) cfgs[0].penalty *= (double)((d.x.C0 + 15) / 16) / (double)(2 * ((d.x.C0 + 31) / 32));
        }
        return plan_launch(cfgs, 5, allowed_bf, M, d.Nstore, d.nphase, nkt, (long)d.NB * d.OH * d.OW * d.ldc, ws_bytes,
                           have_ws, can_ts, plan_const("SSC_PLAN_BF_SCALE", 0.45));
    }
    return plan_launch(FWD_CFGS, 5, allowed, M, d.Nstore, d.nphase, nkt, (long)d.NB * d.OH * d.OW * d.ldc, ws_bytes,
                       have_ws, can_ts);
}

static void copy_name(const char* src, char* dst, int len) {
    int i = 0;
    for (; i < len - 1 && src[i]; ++i) dst[i] = src[i];
    if (len > 0) dst[i] = 0;
}

extern "C" int ssc_conv_forward_kernel_name(const ssc_conv_desc* dp, char* buf, int len) {
    if (ssc_head1_forward_supported(dp) || ssc_head1_dgrad_supported(dp)) {
        copy_name(ssc_head1_dgrad_supported(dp) ? "head1_dgrad" : "head1_fwd", buf, len);
        return 0;
    }
    if (ssc_conv_narrow_supported(dp)) {
        copy_name(dp->nphase == 4 ? "narrow_fwd<transposed>" : "narrow_fwd<conv>", buf, len);
        return 0;
    }
    if (ssc_conv_fewchan_supported(dp)) {
        copy_name(dp->x.C0 == 8 ? "conv_fewchan<8>" : "conv_fewchan<4>", buf, len);
        return 0;
    }
    if (ssc_conv_pw1x1_supported(dp)) {
        copy_name("conv_pw1x1", buf, len);
        return 0;
    }
    if (ssc_conv_c3x3_supported(dp)) {
        copy_name(dp->bmode ? "conv_c3x3<NK>" : "conv_c3x3<KN>", buf, len);
        return 0;
    }
    if (ssc_conv_s2n16_supported(dp)) {
        copy_name("conv_s2n16", buf, len);
        return 0;
    }
    if (ssc_conv_tr4_tiny_supported(dp)) {
        copy_name("deconv_tr4_tiny", buf, len);
        return 0;
    }
    if (ssc_conv_fewchan7_supported(dp)) {
        copy_name("conv_fewchan7", buf, len);
        return 0;
    }
    if (ssc_conv_tr4n16_supported(dp)) {
        copy_name("deconv_tr4n16", buf, len);
        return 0;
    }
    static const char* names[2][5] = {
        {"conv_fwd<128x128,KN>", "conv_fwd<64x128,KN>", "conv_fwd<128x64,KN>", "conv_fwd<128x32,KN>", "conv_fwd<64x64,KN>"},
        {"conv_fwd<128x128,NK>", "conv_fwd<64x128,NK>", "conv_fwd<128x64,NK>", "conv_fwd<128x32,NK>", "conv_fwd<64x64,NK>"}};
    const Plan p = plan_fwd(*dp, (int64_t)1 << 40, true);
    if (fwd_is_bf(*dp)) {
        static const char* bfn[5] = {"conv_bf16x6<128x128>", "conv_bf16x6<64x128>", "conv_bf16x6<128x64>", "conv_bf16x6<128x32>",
                                     "conv_bf16x6<64x64>"};
        copy_name(bfn[p.cfg < 0 ? 0 : p.cfg], buf, len);
        return 0;
    }
    copy_name(names[dp->bmode ? 1 : 0][p.cfg < 0 ? 0 : p.cfg], buf, len);
    return 0;
}

// elementwise.hip
extern "C" int ssc_bn_stats(const float* x, int64_t M, int C, int ldx, const float* scale, const float* offset, float eps,
                            float* ab, float* stats, float* ws, int64_t ws_bytes, void* stream);
extern "C" int ssc_bn_finalize(const float* partial, int nblk, int C, int64_t M, const float* scale, const float* offset,
                               float eps, float* ab, float* stats, void* stream);

// conv + batch-statistics norm fold of its output.  When the launch finishes every output tile in one workgroup with the
// vector epilogue (uniform-tap kernel, no split-K slabs, 16-byte aligned rows, not the 128x128 tile whose C image leaves no
// LDS for the reduction) the per-column sums come out of the conv epilogue; otherwise the output is read back once.
extern "C" int ssc_conv_forward_bn(const ssc_conv_desc* dp, float* ws, int64_t ws_bytes, const float* scale,
                                   const float* offset, float eps, float* ab, float* stats, void* stream) {
    ssc_conv_desc d = *dp;
    d.stat_partial = nullptr;
    d.sb_x = nullptr;
    d.sb2_x = nullptr;
    d.stat_mode = 0;
    const long M = (long)d.NB * d.PH * d.PW;
    const long Mall = M * d.nphase;
    static int off = -1;        // SSC_FUSE_STATS=0: always the separate pass (A/B)
    if (off < 0) {
        const char* e = ssc_dev_getenv("SSC_FUSE_STATS");
        off = (e != nullptr && e[0] == '0') ? 1 : 0;
    }
    bool fused = false;
    int64_t ws_conv = ws_bytes;
    // (the streaming kernels are dispatched BEHIND the one-output head, the few-output and the few-channel kernels in
    // ssc_conv_forward: a descriptor those take must not be promised rows only a streaming kernel writes)
    const bool earlier = ssc_head1_forward_supported(dp) || ssc_head1_dgrad_supported(dp) || ssc_conv_narrow_supported(dp) ||
                         ssc_conv_fewchan_supported(dp);
    if (!off && !earlier && ws != nullptr && d.Nstore == d.ldc &&
        (ssc_conv_pw1x1_supported(&d) || ssc_conv_c3x3_supported(&d) || ssc_conv_s2n16_supported(&d) ||
         ssc_conv_tr4_tiny_supported(&d) || ssc_conv_fewchan7_supported(&d) || ssc_conv_tr4n16_supported(&d))) {
        // the streaming kernels take the statistics as per-lane sums over the tiles a workgroup walks: one row per walker
        const int nblk = ssc_conv_pw1x1_supported(&d) ? ssc_conv_pw1x1_walkers(&d)
                         : (ssc_conv_c3x3_supported(&d) ? ssc_conv_c3x3_walkers(&d)
                            : (ssc_conv_s2n16_supported(&d) ? ssc_conv_s2n16_walkers(&d)
                               : (ssc_conv_tr4_tiny_supported(&d) ? ssc_conv_tr4_tiny_blocks(&d)
                                  : (ssc_conv_fewchan7_supported(&d) ? ssc_conv_fewchan7_walkers(&d) : ssc_conv_tr4n16_rows(&d)))));
        const int64_t need = (int64_t)nblk * 2 * d.Nstore * 4;
        if (need <= ws_bytes) {
            d.stat_partial = ws;
            const int rc = ssc_conv_forward(&d, ws, ws_bytes, stream);
            if (rc != 0) return rc;
            return ssc_bn_finalize(d.stat_partial, nblk, d.Nstore, Mall, scale, offset, eps, ab, stats, stream);
        }
    }
    const bool streaming = ssc_conv_pw1x1_supported(&d) || ssc_conv_c3x3_supported(&d) || ssc_conv_s2n16_supported(&d) ||
                           ssc_conv_tr4_tiny_supported(&d) || ssc_conv_fewchan7_supported(&d) ||
                           ssc_conv_tr4n16_supported(&d);     // rows per walker, not per tile
    if (!off && !streaming && ws != nullptr && !ssc_conv_narrow_supported(dp) && !ssc_conv_fewchan_supported(dp) && d.epi == 0 &&
        !d.accumulate && d.Nstore == d.ldc && ((d.Nstore & 3) == 0) && ((reinterpret_cast<unsigned long>(d.out) & 15) == 0) &&
        (fwd_is_ut(d) || fwd_is_utg(d) || (d.bmode == 0 && fwd_is_rowtap(d)))) {
        const Plan p = plan_fwd(d, ws_bytes, true);
        if (p.cfg >= 0 && p.splitk == 1) {
            const long mt = (M + FWD_CFGS[p.cfg].BM - 1) / FWD_CFGS[p.cfg].BM;
            const int64_t need = (int64_t)mt * d.nphase * 2 * d.Nstore * 4;
            const int64_t reserve = (need + 255) & ~(int64_t)255;
            if (reserve * 4 <= ws_bytes) {     // the partial rows sit at the end of the workspace, the conv keeps the rest
                ws_conv = (ws_bytes - reserve) & ~(int64_t)255;
                const Plan p2 = plan_fwd(d, ws_conv, true);
                if (p2.cfg == p.cfg && p2.splitk == 1) {
                    d.stat_partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ws_conv);
                    fused = true;
                    const int rc = ssc_conv_forward(&d, ws, ws_conv, stream);
                    if (rc != 0) return rc;
                    return ssc_bn_finalize(d.stat_partial, (int)(mt * d.nphase), d.Nstore, Mall, scale, offset, eps, ab, stats,
                                           stream);
                }
            }
        }
    }
    // split-K launches (few rows, long K: the bottlenecks' 3x3 / 4x4 convs at the low resolutions, encoder_5): the statistics ride
    // in the slab sum (slab_reduce4_stats_kernel) -- slab sum + fold instead of slab sum + statistics pass + fold
    if (!off && !streaming && ws != nullptr && !ssc_conv_narrow_supported(dp) && !ssc_conv_fewchan_supported(dp) && slab_stats_ok(d) &&
        !(ssc_head1_forward_supported(dp) || ssc_head1_dgrad_supported(dp))) {
        const Plan p = plan_fwd(d, ws_bytes, true);
        if (p.cfg >= 0 && p.splitk > 1 && p.ts_s <= 1) {
            const int nblk = slab_stats_blocks(Mall, d.ldc);
            const int64_t reserve = (((int64_t)nblk * 2 * d.Nstore * 4) + 255) & ~(int64_t)255;
            ws_conv = (ws_bytes - reserve) & ~(int64_t)255;
            const Plan p2 = plan_fwd(d, ws_conv, true);
            if (ws_conv > 0 && p2.cfg == p.cfg && p2.splitk == p.splitk && p2.ts_s <= 1) {
                d.stat_partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ws_conv);
                const int rc = ssc_conv_forward(&d, ws, ws_conv, stream);
                if (rc != 0) return rc;
                return ssc_bn_finalize(d.stat_partial, nblk, d.Nstore, Mall, scale, offset, eps, ab, stats, stream);
            }
            d.stat_partial = nullptr;
        }
    }
    (void)fused;
    {
        static int dbg = -1;        // SSC_DEBUG_BNPATH=1: why a conv + norm call took the separate statistics pass (stderr)
        if (dbg < 0) {
            const char* e = ssc_dev_getenv("SSC_DEBUG_BNPATH");
            dbg = (e != nullptr && e[0] == '1') ? 1 : 0;
        }
        if (dbg) {
            const Plan p = plan_fwd(d, ws_bytes, ws != nullptr);
            fprintf(stderr, "bnpath separate: M=%ld N=%d K=%d ut=%d utg=%d narrow=%d fewchan=%d cfg=%d splitk=%d epi=%d acc=%d\n", M * d.nphase,
                    d.Nstore, d.TH * d.TW * (d.x.C0 + d.x.C1), (int)fwd_is_ut(d), (int)fwd_is_utg(d), ssc_conv_narrow_supported(dp),
                    ssc_conv_fewchan_supported(dp), p.cfg, p.splitk, d.epi, d.accumulate);
        }
    }
    const int rc = ssc_conv_forward(&d, ws, ws_bytes, stream);
    if (rc != 0) return rc;
    return ssc_bn_stats(d.out, Mall, d.Nstore, d.ldc, scale, offset, eps, ab, stats, ws, ws_bytes, stream);
}

// conv whose output g is the gradient w.r.t. act(a*x+b) of a batch-statistics-normed tensor x ([rows][ldx], laid out like
// the output): when the launch qualifies (as ssc_conv_forward_bn) the two per-channel sums of the norm's backward come out of
// the epilogue as rows of `partial` ([rows][2][Nstore], caller's buffer of at least nphase * ceil(M / 64) rows) and *nrows is
// their count; otherwise *nrows = 0 and the caller takes the sums with the separate pass (ssc_bn_act_backward).
extern "C" int ssc_conv_forward_bnbwd(const ssc_conv_desc* dp, float* ws, int64_t ws_bytes, const float* x, int ldx,
                                      const float* ab, const float* stats, int act, float* partial, int64_t partial_bytes,
                                      int* nrows, void* stream) {
    ssc_conv_desc d = *dp;
    d.stat_partial = nullptr;
    d.sb_x = nullptr;
    d.sb2_x = nullptr;
    d.stat_mode = 0;
    *nrows = 0;
    const long M = (long)d.NB * d.PH * d.PW;
    static int off = -1;        // SSC_FUSE_BNBWD=0: always the separate pass (A/B)
    if (off < 0) {
        const char* e = ssc_dev_getenv("SSC_FUSE_BNBWD");
        off = (e != nullptr && e[0] == '0') ? 1 : 0;
    }
    const bool earlier = ssc_head1_forward_supported(dp) || ssc_head1_dgrad_supported(dp) || ssc_conv_narrow_supported(dp) ||
                         ssc_conv_fewchan_supported(dp);        // dispatched in front of the streaming kernels
    if (!off && !earlier && partial != nullptr && x != nullptr && ab != nullptr && stats != nullptr && d.Nstore == d.ldc &&
        (ssc_conv_c3x3_supported(&d) || ssc_conv_s2n16_supported(&d))) {
        // streaming kernels: the two sums as per-lane sums, one row per walker
        const int nblk = ssc_conv_c3x3_supported(&d) ? ssc_conv_c3x3_walkers(&d) : ssc_conv_s2n16_walkers(&d);
        if ((int64_t)nblk * 2 * d.Nstore * 4 <= partial_bytes) {
            d.stat_partial = partial;
            d.sb_x = x; d.sb_ldx = ldx; d.sb_ab = ab; d.sb_stats = stats; d.sb_act = act;
            *nrows = nblk;
            return ssc_conv_forward(&d, ws, ws_bytes, stream);
        }
    }
    if (!off && ws != nullptr && partial != nullptr && x != nullptr && ab != nullptr && stats != nullptr &&
        !ssc_conv_c3x3_supported(&d) && !ssc_conv_s2n16_supported(&d) && !ssc_conv_narrow_supported(dp) && !ssc_conv_fewchan_supported(dp) && d.epi == 0 && !d.accumulate &&
        d.bias == nullptr && d.Nstore == d.ldc &&
        d.Nn == d.Nstore && ((d.Nstore & 3) == 0) && ((ldx & 3) == 0) && ((reinterpret_cast<unsigned long>(d.out) & 15) == 0) &&
        ((reinterpret_cast<unsigned long>(x) & 15) == 0) &&
        (fwd_is_ut(d) || fwd_is_utg(d) || (d.bmode == 0 && fwd_is_rowtap(d)))) {
        const Plan p = plan_fwd(d, ws_bytes, true);
        if (p.cfg >= 0 && p.splitk == 1) {
            const long mt = (M + FWD_CFGS[p.cfg].BM - 1) / FWD_CFGS[p.cfg].BM;
            if ((int64_t)mt * d.nphase * 2 * d.Nstore * 4 <= partial_bytes) {
                d.stat_partial = partial;
                d.sb_x = x; d.sb_ldx = ldx; d.sb_ab = ab; d.sb_stats = stats; d.sb_act = act;
                *nrows = (int)(mt * d.nphase);
            }
        }
    }
    return ssc_conv_forward(&d, ws, ws_bytes, stream);
}

// mru_ops.hip
extern "C" int ssc_minmax_hw(const float* x, int ld, int N, int P, int C, float* mnmx, float* workspace, int64_t workspace_bytes,
                             void* stream);
extern "C" int ssc_minmax_finalize(const float* part, int nsplit, int N, int C, float* mnmx, void* stream);

// conv (+ bias + lrelu) whose output's per-sample, per-channel extrema are wanted (the MRU gates): the per-tile minima / maxima
// out of the epilogue when the launch qualifies as for ssc_conv_forward_bn and no tile straddles two samples
extern "C" int ssc_conv_forward_minmax(const ssc_conv_desc* dp, float* ws, int64_t ws_bytes, float* mnmx, void* stream) {
    ssc_conv_desc d = *dp;
    d.stat_partial = nullptr;
    d.sb_x = nullptr;
    d.sb2_x = nullptr;
    d.stat_mode = 0;
    const long P = (long)d.PH * d.PW;
    const long M = (long)d.NB * P;
    static int off = -1;        // SSC_FUSE_MINMAX=0: always the separate pass (A/B)
    if (off < 0) {
        const char* e = ssc_dev_getenv("SSC_FUSE_MINMAX");
        off = (e != nullptr && e[0] == '0') ? 1 : 0;
    }
    if (!off && ws != nullptr && d.nphase == 1 && !ssc_conv_narrow_supported(dp) && !ssc_conv_fewchan_supported(dp) &&
        (d.epi == 0 || d.epi == 2) && !d.accumulate && d.Nstore == d.ldc && d.Nn == d.Nstore && ((d.Nstore & 3) == 0) &&
        ((reinterpret_cast<unsigned long>(d.out) & 15) == 0) && (fwd_is_ut(d) || fwd_is_utg(d))) {
        const Plan p = plan_fwd(d, ws_bytes, true);
        if (p.cfg >= 0 && p.splitk == 1 && (P % FWD_CFGS[p.cfg].BM) == 0) {
            const long mt = M / FWD_CFGS[p.cfg].BM;
            const int64_t need = (int64_t)mt * 2 * d.Nstore * 4;
            if (need * 4 <= ws_bytes) {
                const int64_t ws_conv = (ws_bytes - need) & ~(int64_t)255;
                const Plan p2 = plan_fwd(d, ws_conv, true);
                if (p2.cfg == p.cfg && p2.splitk == 1) {
                    d.stat_partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ws_conv);
                    d.stat_mode = 1;
                    const int rc = ssc_conv_forward(&d, ws, ws_conv, stream);
                    if (rc != 0) return rc;
                    return ssc_minmax_finalize(d.stat_partial, (int)(P / FWD_CFGS[p.cfg].BM), d.NB, d.Nstore, mnmx, stream);
                }
            }
        }
    }
    const int rc = ssc_conv_forward(&d, ws, ws_bytes, stream);
    if (rc != 0) return rc;
    return ssc_minmax_hw(d.out, d.ldc, d.NB, (int)P, d.Nstore, mnmx, ws, ws_bytes, stream);
}

extern "C" int ssc_conv_forward_bnbwd2(const ssc_conv_desc* dp, float* ws, int64_t ws_bytes, const ssc_bnbwd_site* s0,
                                       const ssc_bnbwd_site* s1, int C0, int* nrows, void* stream) {
    ssc_conv_desc d = *dp;
    d.stat_partial = nullptr;
    d.sb_x = nullptr;
    d.sb2_x = nullptr;
    d.stat_mode = 0;
    *nrows = 0;
    const long M = (long)d.NB * d.PH * d.PW;
    static int off = -1;        // SSC_FUSE_BNBWD=0: always the separate pass (A/B)
    if (off < 0) {
        const char* e = ssc_dev_getenv("SSC_FUSE_BNBWD");
        off = (e != nullptr && e[0] == '0') ? 1 : 0;
    }
    const int C1 = d.Nstore - C0;
    if (!off && ws != nullptr && s0 != nullptr && s1 != nullptr && s0->x && s1->x && s0->ab && s1->ab && s0->stats && s1->stats &&
        s0->partial && s1->partial && C0 > 0 && C1 > 0 && (C0 % 128) == 0 && (C1 & 3) == 0 &&
        !ssc_conv_narrow_supported(dp) && !ssc_conv_fewchan_supported(dp) && d.epi == 0 && !d.accumulate &&
        d.bias == nullptr && d.Nstore == d.ldc && d.Nn == d.Nstore && ((s0->ldx | s1->ldx) & 3) == 0 &&
        ((reinterpret_cast<unsigned long>(d.out) | reinterpret_cast<unsigned long>(s0->x) |
          reinterpret_cast<unsigned long>(s1->x)) & 15) == 0 &&
        (fwd_is_ut(d) || fwd_is_utg(d))) {
        const Plan p = plan_fwd(d, ws_bytes, true);
        if (p.cfg >= 0 && p.splitk == 1 && FWD_CFGS[p.cfg].BN <= 128) {
            const long mt = (M + FWD_CFGS[p.cfg].BM - 1) / FWD_CFGS[p.cfg].BM;
            if ((int64_t)mt * d.nphase * 2 * C0 * 4 <= s0->partial_bytes && (int64_t)mt * d.nphase * 2 * C1 * 4 <= s1->partial_bytes) {
                d.stat_partial = s0->partial;
                d.sb_x = s0->x; d.sb_ldx = s0->ldx; d.sb_ab = s0->ab; d.sb_stats = s0->stats; d.sb_act = s0->act;
                d.stat_partial2 = s1->partial; d.sb2_col0 = C0;
                d.sb2_x = s1->x; d.sb2_ldx = s1->ldx; d.sb2_ab = s1->ab; d.sb2_stats = s1->stats; d.sb2_act = s1->act;
                *nrows = (int)(mt * d.nphase);
            }
        }
    }
    return ssc_conv_forward(&d, ws, ws_bytes, stream);
}

extern "C" int ssc_sk_configure(int timeout_ms, int test_withhold) {
    const unsigned long long ticks = (unsigned long long)(timeout_ms > 0 ? timeout_ms : 20000) * 100000ull;     // 100 MHz
    const unsigned cfg[4] = {(unsigned)(ticks & 0xffffffffu), (unsigned)(ticks >> 32), test_withhold ? 1u : 0u, 0u};
    const int rc = ssc_sk_configure_bf(cfg);
    if (rc != 0) return rc;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sk_cfg), cfg, sizeof(cfg), 0, hipMemcpyHostToDevice);
}

extern "C" int ssc_conv_forward_plan(const ssc_conv_desc* dp, int64_t ws_bytes, int* out5) {
    // host only: {tile configuration, split-K slabs, whole tiles, K slices per remaining tile, modelled cycles / 1000}
    if (ssc_conv_narrow_supported(dp) || ssc_conv_fewchan_supported(dp)) {
        out5[0] = -1; out5[1] = 1; out5[2] = 0; out5[3] = 1; out5[4] = 0;
        return 0;
    }
    const Plan p = plan_fwd(*dp, ws_bytes, true);
    out5[0] = p.cfg; out5[1] = p.splitk; out5[2] = (int)p.ts_full; out5[3] = p.ts_s; out5[4] = (int)(p.cost / 1000.0);
    return 0;
}

extern "C" int ssc_conv_forward(const ssc_conv_desc* dp, float* ws, int64_t ws_bytes, void* stream) {
    const ssc_conv_desc& d = *dp;
    hipStream_t st = (hipStream_t)stream;
    if ((d.x.C0 & 3) || (d.x.C1 & 3) || d.Nstore < d.Nn || d.Nstore > d.ldc) return -1;
    if (d.nphase != 1 && d.nphase != 4) return -2;
    if (d.nphase == 4 && (d.TH != 2 || d.TW != 2 || d.KH != 4 || d.KW != 4 || d.kstep != -2 || d.out_stride != 2 ||
                          d.in_stride != 1))
        return -3;
    // the one-output patch head over a 512-channel tensor and its data gradient (head1.hip): streaming kernels
    if (ws != nullptr && ssc_head1_forward_supported(dp) && (int64_t)d.NB * d.x.H * d.x.W * 16 * 4 <= ws_bytes)
        return ssc_head1_forward(dp, ws, ws_bytes, stream);
    if (ssc_head1_dgrad_supported(dp)) return ssc_head1_dgrad(dp, stream);
    if (ssc_conv_narrow_supported(dp)) {        // <= 4 output channels
        int csplit = 1;
        const int rc = ssc_conv_narrow_forward_ws(dp, ws, ws_bytes, stream, &csplit);
        if (rc != 0 || csplit == 1) return rc;
        const long out_count = (long)d.NB * d.OH * d.OW * d.ldc;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((out_count + 255) / 256)), dim3(256), 0, st, ws, out_count,
                           csplit, d.out, out_count, d.ldc, d.Nn, d.Nstore, d.bias, d.epi, d.accumulate);
        return (int)hipGetLastError();
    }
    if (ssc_conv_fewchan_supported(dp))         // 4x4 stride-2 over 4 or 8 input channels
        return ssc_conv_fewchan_forward(dp, num_cu(), stream);
    if (ssc_conv_pw1x1_supported(dp))           // 1x1 expansion of a bottleneck: K <= 128, streaming kernel
        return ssc_conv_pw1x1_forward(dp, d.stat_partial, stream);
    if (ssc_conv_c3x3_supported(dp))            // 3x3 of a bottleneck at 16 / 32 channels (either filter orientation)
        return ssc_conv_c3x3_forward(dp, d.stat_partial, stream);
    if (ssc_conv_s2n16_supported(dp))           // 4x4 stride-2 conv 64 -> <= 16 channels: 16-column MFMA, K split over the waves
        return ssc_conv_s2n16_forward(dp, d.stat_partial, stream);
    if (ssc_conv_tr4_tiny_supported(dp))        // k = 4 stride-2 transposed conv, <= 4 channels in and out: a thread per lattice pixel
        return ssc_conv_tr4_tiny_forward(dp, d.stat_partial, stream);
    if (ssc_conv_fewchan7_supported(dp))        // 7x7 stride-2 conv over the (padded) image channels
        return ssc_conv_fewchan7_forward(dp, d.stat_partial, stream);
    if (ssc_conv_tr4n16_supported(dp))          // k = 4 stride-2 transposed conv 256 -> <= 16 channels: 16-column MFMA, a phase per workgroup
        return ssc_conv_tr4n16_forward(dp, d.stat_partial, stream);
    const Plan p = plan_fwd(d, ws_bytes, ws != nullptr);
    if (p.cfg < 0) return -4;
    if (fwd_is_bf(d)) {
        const bool plain0 = d.x.ab0 == nullptr && d.x.act == SSC_ACT_NONE;
        const bool plain1 = d.x.C1 == 0 || (d.x.ab1 == nullptr && (d.x.act1 >= 0 ? d.x.act1 : d.x.act) == SSC_ACT_NONE);
        return ssc_launch_conv_bf(p.cfg, plain0 && plain1, d, p.splitk, ws, st, p.ts_full, p.ts_s, ws_bytes, xcd_order());
    }
    g_launch_res = FWD_CFGS[p.cfg].res;
    g_launch_ws_bytes = ws_bytes;
    g_launch_ts_full = p.ts_full;
    g_launch_ts_s = p.ts_s;
    if (d.bmode == 0) {
        switch (p.cfg) {
            case 0: return launch_fwd<2, 2, 2, 2, 0>(d, p.splitk, ws, st);
            case 1: return launch_fwd<2, 2, 1, 2, 0>(d, p.splitk, ws, st);
            case 2: return launch_fwd<2, 2, 2, 1, 0>(d, p.splitk, ws, st);
            case 4: return launch_fwd<2, 2, 1, 1, 0>(d, p.splitk, ws, st);
            default: return launch_fwd<4, 1, 1, 1, 0>(d, p.splitk, ws, st);
        }
    } else {
        switch (p.cfg) {
            case 0: return launch_fwd<2, 2, 2, 2, 1>(d, p.splitk, ws, st);
            case 1: return launch_fwd<2, 2, 1, 2, 1>(d, p.splitk, ws, st);
            case 2: return launch_fwd<2, 2, 2, 1, 1>(d, p.splitk, ws, st);
            case 4: return launch_fwd<2, 2, 1, 1, 1>(d, p.splitk, ws, st);
            default: return launch_fwd<4, 1, 1, 1, 1>(d, p.splitk, ws, st);
        }
    }
}

static bool gview_plain(const ssc_gview& g) {
    return g.ab0 == nullptr && g.act == SSC_ACT_NONE &&
           (g.C1 == 0 || (g.ab1 == nullptr && (g.act1 >= 0 ? g.act1 : g.act) == SSC_ACT_NONE));
}

template <int WM, int WN, int SM, int SN, bool GPLAIN, bool DPLAIN, bool DDMA>
static int launch_wgrad_v(const ssc_wgrad_desc& d, int splitk, float* ws, hipStream_t st) {
    constexpr int BM = WM * SM * 32, BN = WN * SN * 32;
    constexpr size_t lds = 2 * (BK * BM + BK * BN) * sizeof(float) + 2 * BK * sizeof(int4);
    const int Cg = d.g.C0 + d.g.C1;
    const int Mtot = d.TH * d.TW * Cg;
    const long P = (long)d.NB * d.PH * d.PW;
    const Magics mg = make_magics((unsigned)Cg, (unsigned)d.TW, (unsigned long)d.PW, (unsigned long)d.PH * d.PW,
                                  (unsigned long)P);
    const int mt = (Mtot + BM - 1) / BM;
    const int nt = (d.Nn + BN - 1) / BN;
    const long out_count = (long)d.TH * d.TW * d.Cg_real * d.ldc;
    static unsigned long long attr_done = 0;
    {
        const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&conv_wgrad_kernel<WM, WN, SM, SN, GPLAIN, DPLAIN, DDMA>), (int)lds, &attr_done);
        if (arc != 0) return arc;
    }
    dim3 grid((unsigned)mt, (unsigned)nt, (unsigned)splitk);
    // SSC_WG_XCD=1: whole K slices per XCD.  Measured (scripts/wg_xcd_ab.sh): FETCH_SIZE of the encoder_3 filter gradient 113 ->
    // 35 MB, the launch itself unchanged (94.5 vs 94.9 TFLOP/s), the train step 0.4 % SLOWER (18.05 vs 17.98 ms) -- the re-reads
    // of the plain order are served by the Infinity Cache while all XCDs walk the same pixel range -- so it stays off
    static int wg_xcd = -1;
    if (wg_xcd < 0) {
        const char* e = ssc_dev_getenv("SSC_WG_XCD");
        wg_xcd = (e != nullptr && e[0] == '1') ? 1 : 0;
    }
    const int xcd = (wg_xcd && splitk > 1 && (((long)mt * nt * splitk) & 7) == 0) ? 1 : 0;
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, SM, SN, GPLAIN, DPLAIN, DDMA>), grid, dim3(256), lds, st, d, mg, ws, out_count,
                       splitk, xcd);
    if (splitk > 1) launch_wgrad_reduce(ws, out_count, splitk, d.out, d.accumulate, st);
    return (int)hipGetLastError();
}

template <int WM, int WN, int SM, int SN>
static int launch_wgrad(const ssc_wgrad_desc& d, int splitk, float* ws, hipStream_t st) {
    const bool gp = gview_plain(d.g), dp = gview_plain(d.d);
    static int ddma_on = -1;       // SSC_WGRAD_DMA=0: dense tiles through registers (A/B)
    if (ddma_on < 0) {
        const char* e = ssc_dev_getenv("SSC_WGRAD_DMA");
        ddma_on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    // LDS-DMA of the dense tile: one plain tensor whose byte offsets fit 32 bits
    const bool ddma = SSC_BDMA && ddma_on && dp && d.d.C1 == 0 && (long)d.NB * d.PH * d.PW * d.d.C0 < 0x1fffffffL;
    if (dp && ddma) return gp ? launch_wgrad_v<WM, WN, SM, SN, true, true, true>(d, splitk, ws, st)
                              : launch_wgrad_v<WM, WN, SM, SN, false, true, true>(d, splitk, ws, st);
    if (dp) return gp ? launch_wgrad_v<WM, WN, SM, SN, true, true, false>(d, splitk, ws, st)
                      : launch_wgrad_v<WM, WN, SM, SN, false, true, false>(d, splitk, ws, st);
    return launch_wgrad_v<WM, WN, SM, SN, false, false, false>(d, splitk, ws, st);   // the full transform covers a plain side
}

// split over the pixel (K) dimension: many more choices than the forward form, so search a wider range
static Plan plan_wgrad(const ssc_wgrad_desc& d, int64_t ws_bytes, bool have_ws) {
    const int Cg = d.g.C0 + d.g.C1;
    const long Mtot = (long)d.TH * d.TW * Cg;
    const long P = (long)d.NB * d.PH * d.PW;
    const long nkt = (P + BK - 1) / BK;
    const long out_elems = (long)d.TH * d.TW * d.Cg_real * d.ldc;
    bool allowed[5];
    allowed[0] = d.Nn > 64 && Mtot > 64;
    allowed[1] = d.Nn > 64;                     // 64 gathered columns x 128: the gathered side is the costly one
    allowed[2] = d.Nn > 32 && Mtot > 64;
    allowed[3] = d.Nn > 32 && d.Nn <= 64 && Mtot <= 64;
    allowed[4] = d.Nn <= 32;
    const int ncu = num_cu();
    Plan best = {-1, 1, 1e300, 0, 1};
    static int force_cfg = -2, force_sk = -2;       // SSC_WG_CFG / SSC_WG_SPLITK: tuning aids
    if (force_cfg == -2) {
        const char* e = ssc_dev_getenv("SSC_WG_CFG");
        force_cfg = (e != nullptr) ? atoi(e) : -1;
        e = ssc_dev_getenv("SSC_WG_SPLITK");
        force_sk = (e != nullptr) ? atoi(e) : -1;
    }
    for (int c = 0; c < 5; ++c) {
        if (!allowed[c]) continue;
        if (force_cfg >= 0 && c != force_cfg && allowed[force_cfg]) continue;
        const TileCfg& t = WG_CFGS[c];
        const long mt = (Mtot + t.BM - 1) / t.BM, nt = (d.Nn + t.BN - 1) / t.BN;
        const long blocks = mt * nt;
        const double wfull = (double)nkt * 16.0 * (t.BM / 32) * (t.BN / 32) / 4.0 * 64.0 * t.penalty;
        for (long sk = 1; sk <= 512; sk = (sk < 16 ? sk + 1 : sk + sk / 8)) {
            if (sk > 1 && (!have_ws || nkt / sk < 4 || (int64_t)sk * out_elems * 4 > ws_bytes)) break;
            const long per = (nkt + sk - 1) / sk;
            if ((nkt + per - 1) / per != sk) continue;
            double cost = makespan(blocks * sk, wfull * (double)per / (double)nkt, t.res, ncu) + 2500.0;
            if (sk > 1) cost += 2.0 * sk * (double)out_elems * 4.0 / 1500.0 + plan_const("SSC_PLAN_REDUCE", 6000.0);
            if (force_sk > 0) cost = (double)(sk > force_sk ? sk - force_sk : force_sk - sk);     // nearest legal split
            if (cost < best.cost) best = {c, (int)sk, cost, 0, 1};
        }
    }
    return best;
}

extern "C" int ssc_conv_wgrad_kernel_name(const ssc_wgrad_desc* dp, char* buf, int len) {
    static const char* names[5] = {"conv_wgrad<128x128>", "conv_wgrad<64x128>", "conv_wgrad<128x64>",
                                   "conv_wgrad<64x64>", "conv_wgrad<128x32>"};
    if (ssc_head1_wgrad_supported(dp)) {
        copy_name("head1_wgrad", buf, len);
        return 0;
    }
    if (ssc_conv_wgrad128_supported(dp)) {
        copy_name(ssc_conv_wgrad128_bf_selected(dp) ? "conv_wgrad128_bf16x6<128x128>" : "conv_wgrad128<128x128>", buf, len);
        return 0;
    }
    if (ssc_conv_wgn16_supported(dp)) {
        copy_name(dp->TH == 3 ? "conv_wgn16<3x3>" : "conv_wgn16<4x4>", buf, len);
        return 0;
    }
    const Plan p = plan_wgrad(*dp, (int64_t)1 << 40, true);
    copy_name(names[p.cfg < 0 ? 0 : p.cfg], buf, len);
    return 0;
}

extern "C" int ssc_conv_wgrad(const ssc_wgrad_desc* dp, float* ws, int64_t ws_bytes, void* stream) {
    const ssc_wgrad_desc& d = *dp;
    hipStream_t st = (hipStream_t)stream;
    if ((d.g.C0 & 3) || (d.g.C1 & 3) || (d.d.C0 & 3) || (d.d.C1 & 3)) return -1;
    if (d.d.H != d.PH || d.d.W != d.PW) return -2;
    // the filter-gradient slab is dense [TH*TW*Cg_real][ldc]; rows/cols skipped by the kernel
    // (padding channels) do not exist in it, so every slab entry is written when ldc == Nn.
    if (d.ldc != d.Nn) return -3;
    if (ws != nullptr && ws_bytes >= (int64_t)16 * 512 * 4 && ssc_head1_wgrad_supported(dp))     // the one-output patch head
        return ssc_head1_wgrad(dp, ws, ws_bytes, stream);
    if (ssc_conv_wgrad128_supported(dp)) return ssc_conv_wgrad128(dp, ws, ws_bytes, stream);     // the large layers
    if (ws != nullptr && ssc_conv_wgn16_supported(dp)) {      // 16 output channels, 3x3 x 16 or 4x4 x 64 gathered: 16-column MFMA
        const int rc = ssc_conv_wgn16(dp, ws, ws_bytes, stream);
        if (rc != -2) return rc;        // -2: the workspace cannot hold a slab
    }
    const Plan p = plan_wgrad(d, ws_bytes, ws != nullptr);
    switch (p.cfg) {
        case 0: return launch_wgrad<2, 2, 2, 2>(d, p.splitk, ws, st);
        case 1: return launch_wgrad<1, 4, 2, 1>(d, p.splitk, ws, st);
        case 2: return launch_wgrad<2, 2, 2, 1>(d, p.splitk, ws, st);
        case 3: return launch_wgrad<2, 2, 1, 1>(d, p.splitk, ws, st);
        case 4: return launch_wgrad<4, 1, 1, 1>(d, p.splitk, ws, st);
        default: return -4;
    }
}
#endif  // SSC_ISA_ONLY
