// igemm_bf16.hip -- the uniform-tap implicit GEMM of igemm.hip on the bf16 matrix pipe with fp32-grade arithmetic.
//
// gfx950 runs the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) at the vector-ALU rate (157 TFLOP/s) and it does not co-issue with
// vector-ALU work; the bf16 MFMA (v_mfma_f32_32x32x16_bf16) runs at 16x that rate beside the vector ALU.  Every fp32 value is
// the EXACT sum of three bf16 values (x = h + m + l: 8 + 8 + 8 significant bits, round-to-nearest at each step, residuals are
// exact in fp32), so
//     a * b = (ah + am + al) * (bh + bm + bl) = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh) + [am*bl + al*bm + al*bl]
// and the six products outside the brackets, each exact in the MFMA's fp32 accumulator, reproduce a * b to 2^-23 relative
// (|am| <= 2^-8 |a|, |bl| <= 2^-16 |b|: the three dropped products are together below one fp32 rounding of the product).  Six
// bf16 MFMA passes cost 6/16 of the fp32 MFMA's cycles.  Accumulation is fp32 in both forms.
//   * The FILTER operand is split ahead of time (ssc_filter_split: once per optimizer step) into three bf16 planes stored
//     fragment-major -- the 64 x 16-byte image of one MFMA B operand (32 columns x 16 k) is one contiguous KiB -- so a K-tile's
//     filter data goes global memory -> LDS by LDS-DMA and is read back as operands with lane-linear ds_read_b128.
//   * The GATHERED operand is split inside the staging that already transforms it (folded norm + activation + padding mask):
//     v_cvt_pk_bf16_f32 + shifts + subtractions beside the MFMAs, three ds_write_b64 per float4.
//   * Prologue (tile order, tail split, XCD-aware mapping, pixel decode) and epilogue are conv_ut_kernel's (igemm_epilogue.h).
// Replaces tf.nn.conv2d / conv2d_transpose and their data gradients (models_collection.py:380-405) on the layers whose channel
// counts are multiples of 32; everything else stays on the exact-fp32 kernels of igemm.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "igemm_util.h"
#include "igemm_epilogue.h"
#include "host_util.h"

#define BK 32
#ifndef SSC_BF_DIAG_BUILD
#define SSC_BF_DIAG_BUILD 0     // diagnostic builds: bit 0 no MFMAs, bit 1 no LDS-DMA, bit 2 no staging stores, bit 3 no gather loads,
                                // bit 4 no 3-way split (the raw bits three times), bit 5 no transform (norm / activation / mask),
                                // bit 6 no operand reads from LDS (scripts/bf_diag_builds.sh: what bounds the K step?)
#endif
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// this translation unit's copy of the hand-off configuration (ssc_sk_configure sets both)
__device__ unsigned g_sk_cfg_bf[4] = {2000000000u, 0u, 0u, 0u};

int ssc_sk_configure_bf(const unsigned* cfg4) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sk_cfg_bf), cfg4, 4 * sizeof(unsigned), 0, hipMemcpyHostToDevice);
}

// N LDS-DMA instructions (16 bytes per lane from sbase + voff[q] to LDS at lds_addr + q * 1024 + 16 * lane) with ONE save / restore
// of M0: the per-instruction form (glds16) spends four scalar instructions on M0 per DMA, and the scalar stream of this kernel
// is as long as its vector stream
template <int N>
__device__ __forceinline__ void glds16_run(const char* sbase, const unsigned* voff, unsigned lds_addr) {
    unsigned keep;
    if (N == 1) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff[0]), "s"(sbase), "s"(lds_addr) : "memory");
    } else if (N == 2) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "s"(sbase), "s"(lds_addr) : "memory", "scc");
    } else {
        static_assert(N >= 1 && N <= 3, "1..3 instructions per run");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %4\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "s"(sbase), "s"(lds_addr) : "memory", "scc");
    }
}

// ---------------------------------------------------------------------------------------------
// the 3-way split
// ---------------------------------------------------------------------------------------------
// two values at a time: v_cvt_pk_bf16_f32 rounds to nearest even and packs; {lo, hi} halves back to fp32 are a shift and a mask
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    if (SSC_BF_DIAG_BUILD & 16) { h = m = l = __float_as_uint(x0) ^ __float_as_uint(x1); return; }
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

// ---------------------------------------------------------------------------------------------
// filter planes
// ---------------------------------------------------------------------------------------------
// A filter W[tap][c0][c1] (fp32, c1 contiguous: tf conv [kh,kw,Cin,Cout] and conv2d_transpose [kh,kw,Cout,Cin] alike) as a GEMM
// operand B[k][n] per tap, orient 0: k = c0, n = c1 ("KN");  orient 1: k = c1, n = c0 ("NK").  Layout of the planes buffer:
//   fragment(tap, kc, plane, nb) at byte (((tap * KC + kc) * 3 + plane) * NBP + nb) * 1024, KC = 2 * ceil(K / 32) chunks of 16 k,
//   NBP = ceil(N / 32) + 3 blocks of 32 columns (three zero blocks: a 128-column tile may start at any block);
//   inside a fragment lane L holds the 8 bf16 B[kc * 16 + (L >> 5) * 8 + e][nb * 32 + (L & 31)], e = 0..7 -- the B operand of
//   v_mfma_f32_32x32x16_bf16.  k >= K and n >= N hold zeros.
struct SplitGeom { int K, N, KC, NBP; };
static inline SplitGeom split_geom(int c0, int c1, int orient) {
    SplitGeom g;
    g.K = orient ? c1 : c0;
    g.N = orient ? c0 : c1;
    g.KC = 2 * ((g.K + 31) / 32);
    g.NBP = (g.N + 31) / 32 + 3;
    return g;
}

static int64_t ssc_filter_split_bytes(int taps, int c0, int c1, int orient) {
    const SplitGeom g = split_geom(c0, c1, orient);
    return (int64_t)taps * g.KC * 3 * g.NBP * 1024;
}

// round-to-nearest-even to bf16 on the integer pipe (this kernel is not on a hot path); finite inputs
__device__ __forceinline__ unsigned rne_bf16_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}

// one thread per (fragment, lane): 8 values read, 3 x 16 bytes written
__device__ __forceinline__ void filter_split_body(const ssc_split_job& jb, long lt) {
    const int K = jb.orient ? jb.c1 : jb.c0, N = jb.orient ? jb.c0 : jb.c1;
    const int KC = 2 * ((K + 31) / 32), NBP = (N + 31) / 32 + 3;
    const long nfrag = (long)jb.taps * KC * NBP;
    if (lt >= nfrag * 64) return;
    const int lane = (int)(lt & 63);
    const long f = lt >> 6;
    const int nb = (int)(f % NBP);
    const long r = f / NBP;
    const int kc = (int)(r % KC), tap = (int)(r / KC);
    const int n = nb * 32 + (lane & 31), k0 = kc * 16 + (lane >> 5) * 8;
    unsigned hb[8], mb[8], lb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        float x = 0.f;
        if (k < K && n < N)
            x = jb.orient ? jb.w[((long)tap * jb.c0 + n) * jb.c1 + k] : jb.w[((long)tap * jb.c0 + k) * jb.c1 + n];
        const unsigned h = rne_bf16_bits(x);
        const float r1 = x - __uint_as_float(h);
        const unsigned m = rne_bf16_bits(r1);
        const float r2 = r1 - __uint_as_float(m);
        hb[e] = h; mb[e] = m; lb[e] = rne_bf16_bits(r2);
    }
    uint4 vh, vm, vl;
    vh.x = (hb[0] >> 16) | hb[1]; vh.y = (hb[2] >> 16) | hb[3]; vh.z = (hb[4] >> 16) | hb[5]; vh.w = (hb[6] >> 16) | hb[7];
    vm.x = (mb[0] >> 16) | mb[1]; vm.y = (mb[2] >> 16) | mb[3]; vm.z = (mb[4] >> 16) | mb[5]; vm.w = (mb[6] >> 16) | mb[7];
    vl.x = (lb[0] >> 16) | lb[1]; vl.y = (lb[2] >> 16) | lb[3]; vl.z = (lb[4] >> 16) | lb[5]; vl.w = (lb[6] >> 16) | lb[7];
    char* base = reinterpret_cast<char*>(jb.dst) + ((((long)tap * KC + kc) * 3) * NBP + nb) * 1024 + lane * 16;
    *reinterpret_cast<uint4*>(base) = vh;
    *reinterpret_cast<uint4*>(base + (long)NBP * 1024) = vm;
    *reinterpret_cast<uint4*>(base + (long)2 * NBP * 1024) = vl;
}

// many filters in one launch: jobs in device memory, first_thread ascending
__global__ __launch_bounds__(256) void filter_split_kernel(const ssc_split_job* __restrict__ jobs, int njobs) {
    // the job of this block (jobs own whole blocks: first_thread is a multiple of 256 -- one search per block, block-uniform):
    // the last one with first_thread <= t, by bisection (a linear walk cost a dependent load per job: the Background generator's
    // table holds some two hundred)
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long tb = (long)blockIdx.x * 256;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_thread <= tb) lo = mid;
        else hi = mid - 1;
    }
    const ssc_split_job jb = jobs[lo];
    filter_split_body(jb, t - jb.first_thread);
}
__global__ __launch_bounds__(256) void filter_split_one_kernel(const ssc_split_job jb) {
    filter_split_body(jb, (long)blockIdx.x * 256 + threadIdx.x);
}

static int64_t ssc_filter_split_threads(int taps, int c0, int c1, int orient) {
    const SplitGeom g = split_geom(c0, c1, orient);
    return (((int64_t)taps * g.KC * g.NBP * 64 + 255) / 256) * 256;     // whole blocks per job
}

extern "C" int ssc_filter_split_geom(int taps, int c0, int c1, int orient, int* kc, int* nbp, int64_t* bytes, int64_t* threads) {
    if (taps <= 0 || c0 <= 0 || c1 <= 0) return -1;
    const SplitGeom g = split_geom(c0, c1, orient);
    if (kc) *kc = g.KC;
    if (nbp) *nbp = g.NBP;
    if (bytes) *bytes = ssc_filter_split_bytes(taps, c0, c1, orient);
    if (threads) *threads = ssc_filter_split_threads(taps, c0, c1, orient);
    return 0;
}

// jobs: DEVICE array of njobs entries, first_thread ascending from 0 in steps of the jobs' thread counts; total_threads = their sum
extern "C" int ssc_filter_split_batch(const ssc_split_job* jobs_dev, int njobs, int64_t total_threads, void* stream) {
    if (njobs <= 0 || total_threads <= 0) return 0;
    if (total_threads / 256 > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(filter_split_kernel, dim3((unsigned)(total_threads / 256)), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// LDS images of one K-tile (32 k):
//   A [BM rows][208 bytes]: three planes of 64 bytes (32 bf16: the 16-byte chunk (kc, lhi) holds k = kc*16 + lhi*8 .. +7) + 16 bytes
//     of padding -- a row stride of 13 x 16 bytes keeps the ds_read_b128 of a wave's 32 rows conflict-free;
//   B [kc 2][plane 3][BN/32 blocks][1 KiB fragment]: the DMA's image, read with lane-linear addresses.
#define BF_A_RS 208
// s_waitcnt vmcnt(0) lgkmcnt(0) through the builtin (simm16: vmcnt [3:0] and [15:14], expcnt [6:4] = 7 = no wait, lgkmcnt [11:8]): the
// compiler's own wait insertion then knows that every load it issued has landed and adds no vmcnt wait of its own in front of the
// first use of the staged registers -- which, counted without the LDS-DMA instructions it cannot see, would wait for those too.
// The inline-asm statement keeps the DMA's LDS writes (which the compiler does not know about) ordered before the barrier.
#define BF_WAIT_ALL() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0070); } while (0)
// SS: ONE stage of each operand instead of two (the K step fetches both 16-k halves of its operands up front and meets a barrier
// before the next K-tile is stored: conv_bf_kernel); the buffers also hold the epilogue's C image + reduction scratch
template <int BM, int BN, bool SS = false> struct BfLds {
    static constexpr int A_BYTES = BM * BF_A_RS;
    static constexpr int B_BYTES = 6 * (BN / 32) * 1024;
    static constexpr int EPI_BYTES = (BM * (BN + 4) + 2 * 256 * 4) * 4;
    static constexpr int ONE = A_BYTES + B_BYTES;
    static constexpr int OPER_BYTES = SS ? (ONE > EPI_BYTES ? ONE : EPI_BYTES) : 2 * ONE;
    static constexpr int TOTAL = OPER_BYTES + BM * 8;
};

// tab: the norm tables of the two sources as {a0, b0, a1, b1} ([C] each), never NULL: a source without a table gets the launcher's
// identity rows (ones, zeros) -- no per-step "has a table" selects, and the pointers stay plain global pointers
struct BfTabs { const float* a0; const float* b0; const float* a1; const float* b1; };
// ONE: the gathered side has ONE source tensor (no channel concat): the per-step selects between the two sources' pointers, strides,
// tables and row offsets (half of the step's scalar instructions) are compiled out
// KM (with ONE): the source's channel count is any multiple of 4 (MRU's materialised concats [state | image] = 64 + 4, 128 + 4 ...,
// mru.py:400-411, 555-575): the last 32-wide chunk of every tap is partly empty -- its float4s beyond the row are staged as zeros
// (never loaded) and the filter planes hold zeros there (ssc_filter_split pads k >= K)
template <int WM, int WN, int SM, int SN, bool PLAIN, bool ONE, bool KM = false, bool SS = false>
__global__ __launch_bounds__(256) void conv_bf_kernel(const ssc_conv_desc d, const Magics mg, float* __restrict__ slab_base,
                                                       long slab_stride, int splitk, int ts_full, int ts_s,
                                                       unsigned* __restrict__ flags, const BfTabs tab, const int korder) {
    constexpr int BM = WM * SM * 32;
    constexpr int BN = WN * SN * 32;
    constexpr int NBT = BN / 32;
    constexpr int A_BYTES = BfLds<BM, BN, SS>::A_BYTES, B_BYTES = BfLds<BM, BN, SS>::B_BYTES;
    constexpr int LDS_FLOATS = BfLds<BM, BN, SS>::OPER_BYTES / 4;
    constexpr int B_BASE = (SS ? 1 : 2) * A_BYTES;      // the filter stage(s) behind the gathered side's
    constexpr int A_ROWS = BM / 32;
    constexpr int B_IPW = 6 * NBT / 4;          // DMA instructions per wave and K-tile
    static_assert(WM * WN == 4 && (6 * NBT) % 4 == 0, "4 waves share the fragments of a K-tile evenly");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm_b = reinterpret_cast<char*>(smem);
    long* rowpix = reinterpret_cast<long*>(smem + LDS_FLOATS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    const float* const xs0 = d.x.s0;
    const float* const xs1 = d.x.s1;
    const int xC0 = d.x.C0, xC1 = d.x.C1, xH = d.x.H, xW = d.x.W;
    const int TWv = d.TW, kstep = d.kstep, KWv = d.KW;
    const float slope0 = act_slope(d.x.act), slope1 = act_slope(d.x.act1 >= 0 ? d.x.act1 : d.x.act);

    static_assert(!KM || ONE, "partial chunks: one source only");
    const int nch0 = KM ? (xC0 + BK - 1) / BK : xC0 / BK, nch1 = xC1 / BK;
    const int tpt = nch0 + nch1;
    const long M = (long)d.NB * d.PH * d.PW;
    const int PHW = d.PH * d.PW;
    // workgroup -> (tile, K range): conv_ut_kernel's layouts (igemm.hip)
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
            if (bid < ts_full) tile = (bid & 7) * (ts_full >> 3) + (bid >> 3);
            const int r2 = tile / nt;
            n0 = (tile - r2 * nt) * BN;
            if (ts_s & 0x20000) {
                phase = r2 & 3;
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

    // gathered side: thread t stages the float4 (4 k) at chunk t & 7 of the rows arow + 32 i.  arow: the 8 rows of a wave with
    // bits 0 and 1 of the row swapped, so that the two rows a 16-lane group of a ds_write_b64 covers lie two rows (416 bytes =
    // 40 banks) apart instead of one (208 bytes = 52 banks: the second row's 16 banks wrapped onto 4 of the first's)
    const int a_col4 = tid & 7;
    const bool a_lastv = !KM || (nch0 - 1) * BK + a_col4 * 4 < xC0;      // KM: this thread's float4 of the LAST chunk exists
    // KM: stored channels in [k_real, C0) (the padding lane of a concat [state | image(3) | pad]) meet zero filter planes, but
    // 0 * NaN = NaN: whatever sits there is cleared by a bit mask on the way to LDS (element e of this thread's float4 of the
    // last chunk is real iff its channel < k_real)
    uint4 a_lastm = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (KM) {
        const int c_ = (nch0 - 1) * BK + a_col4 * 4;
        a_lastm = make_uint4(c_ < d.k_real ? ~0u : 0u, c_ + 1 < d.k_real ? ~0u : 0u, c_ + 2 < d.k_real ? ~0u : 0u,
                             c_ + 3 < d.k_real ? ~0u : 0u);
    }
    const int arow = ((tid >> 3) & ~3) | (((tid >> 3) & 1) << 1) | (((tid >> 3) >> 1) & 1);
    int a_iyb[A_ROWS], a_ixb[A_ROWS], a_off0[A_ROWS], a_off1[A_ROWS];
    bool a_mv[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int row = arow + 32 * i;
        const long m = m0 + row;
        a_mv[i] = m < M;
        const long mm = a_mv[i] ? m : 0;
        int n, rem, py;
        if (mg.use32) {
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
        a_off0[i] = (pix0 * xC0 + a_col4 * 4) * 4;
        a_off1[i] = (pix0 * xC1 + a_col4 * 4) * 4;
        if (a_col4 == 0)
            rowpix[row] = ((long)n * d.OH + py * d.out_stride + ph.ooff_y) * d.OW + px * d.out_stride + ph.ooff_x;
    }

    // which taps bring row i inside the image: one bit per tap (the launcher takes at most 32 taps), decoded once per tile
    // instead of compared every K step
    unsigned a_vm[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) a_vm[i] = 0u;
    {
        const int ntaps = d.TH * TWv;
        int ty = 0, tx = 0;
        for (int t = 0; t < ntaps; ++t) {
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                const bool v = a_mv[i] & ((unsigned)(a_iyb[i] + ty) < (unsigned)xH) & ((unsigned)(a_ixb[i] + tx) < (unsigned)xW);
                a_vm[i] |= v ? (1u << t) : 0u;
            }
            tx += 1;
            if (tx == TWv) { tx = 0; ty += 1; }
        }
    }

    // filter side: fragment q of this wave = f = wave * B_IPW + q of the K-tile's [kc][plane][block] image
    const int NBP = d.ws_nbp, KC = d.ws_kc;
    const int nb0 = (d.n_off + n0) >> 5;
    unsigned bd_off[B_IPW];
#pragma unroll
    for (int q = 0; q < B_IPW; ++q) {
        const int f = wave * B_IPW + q;
        const int kc = f / (3 * NBT), pl = (f / NBT) % 3, nbl = f % NBT;
        bd_off[q] = (unsigned)((((kc * 3 + pl) * NBP) + nb0 + nbl) * 1024 + lane * 16);
    }
    const char* const wsp = reinterpret_cast<const char*>(d.wsplit);
    const long ktile_bytes = (long)6 * NBP * 1024;      // 2 chunks x 3 planes x NBP blocks

    const int nkt = d.TH * d.TW * tpt;
    const int per = (nkt + sk - 1) / sk;
    const int kt_begin = ks * per;
    const int kt_end = min(nkt, kt_begin + per);

    // Two accumulators per 32x32 block: the hh products (the sum's magnitude) in one, the five correction products (2^-8 of it and
    // below) in the other, added once behind the K loop.  The matrix pipe aligns the 16 products of an instruction and the
    // accumulator to the largest exponent among them, so every pass into a LARGE accumulator costs a rounding of the large value:
    // six passes per K step into one accumulator measured 3.8e-7 rms at K = 512 against 1.4e-7 for this form and 4.5e-7 for the
    // fp32 fmaf chain (scripts/mfma_bf16_acc_probe.hip) -- the split form is then MORE exact than the exact-fp32 MFMA's chain.
    f32x16 acc[SM][SN], accc[SM][SN];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = accc[i][j][r] = 0.f;

    // K-tile index -> (tap row, tap column, chunk), advanced by one K-tile at a time on the scalar unit instead of decoded by
    // division every step.  Order of the K-tiles (korder, wave-uniform):
    //   0: chunk fastest, then tap column, then tap row (rounds 1-5);
    //   1: TAP fastest (row-major), chunk slowest: the taps of a stride-1 window read almost the same pixels of a chunk -- one
    //      K step apart instead of a whole tap's chunks apart, i.e. out of L2 instead of over the fabric again;
    //   2: (in_stride 2, even tap counts) tap fastest in the order of the four PARITY classes (ty & 1, tx & 1): the taps of a
    //      class read the same input pixels shifted by whole output pixels, the classes share nothing.
    // Any order is a valid summation order; K slices (split-K, tail split) are ranges of the sequence.
    struct KTile { int ty, tx, chunk, tappix, tapidx; long woff; };  // tappix = ty * xW + tx, tapidx = ty * TW + tx; woff: byte offset
                                                                     // of the K-tile's filter data inside the planes
    const long KB = ktile_bytes;
    const int THv = d.TH, ntaps = d.TH * d.TW;
    auto kt_place = [&](KTile& t) {       // (ty, tx, chunk) -> the derived fields
        t.tappix = t.ty * xW + t.tx;
        t.tapidx = t.ty * TWv + t.tx;
        t.woff = ((long)((ph.ky0 + t.ty * kstep) * KWv + ph.kx0 + t.tx * kstep) * (KC >> 1) + t.chunk) * KB;
    };
    const long SX = (long)kstep * (KC >> 1) * KB, SY = (long)kstep * KWv * (KC >> 1) * KB;     // one tap column / row further
    const long DX = SX - (long)tpt * KB, DY = SY - (long)TWv * SX;
    auto kt_decode = [&](int kt) {
        KTile t;
        if (korder != 0) {      // once per workgroup: plain divisions
            t.chunk = kt / ntaps;
            const int sidx = kt - t.chunk * ntaps;
            if (korder == 1) {
                t.ty = sidx / TWv;
                t.tx = sidx - t.ty * TWv;
            } else {
                const int hw = TWv >> 1, per = ntaps >> 2;
                const int cls = sidx / per, j = sidx - cls * per;
                const int jy = j / hw, jx = j - jy * hw;
                t.ty = 2 * jy + (cls >> 1);
                t.tx = 2 * jx + (cls & 1);
            }
            kt_place(t);
            return t;
        }
        const int tap = div32(kt, mg.mC, mg.oneC);      // mC: magic of tpt
        t.chunk = kt - tap * tpt;
        const int ty = div32(tap, mg.mTW, mg.oneTW);
        t.ty = ty;
        t.tx = tap - ty * TWv;
        t.tappix = ty * xW + t.tx;
        t.tapidx = tap;
        t.woff = ((long)((ph.ky0 + ty * kstep) * KWv + ph.kx0 + t.tx * kstep) * (KC >> 1) + t.chunk) * KB;
        return t;
    };
    // one K-tile further, with (wave-uniform) BRANCHES: the usual case -- the next chunk of the same tap -- is four scalar
    // instructions; the branch-free form (selects on every field) cost forty per step
    // one K-tile further: the usual case is a handful of scalar additions; whole recomputations (kt_place: multiplications)
    // only where a tap class or a chunk ends
    const long W0 = (long)(ph.ky0 * KWv + ph.kx0) * (KC >> 1) * KB;
    const long SX2 = 2 * SX, DY2 = 2 * SY - (long)TWv * SX;
    auto kt_advance = [&](KTile& t) {
        if (korder == 1) {
            t.tx += 1;
            t.tapidx += 1;
            t.tappix += 1;
            t.woff += SX;
            if (t.tx == TWv) {
                t.tx = 0;
                t.ty += 1;
                t.tappix += xW - TWv;
                t.woff += DY;
                if (t.ty == THv) {
                    t.ty = 0;
                    t.chunk += 1;
                    t.tapidx = 0;
                    t.tappix = 0;
                    t.woff = W0 + (long)t.chunk * KB;
                }
            }
            return;
        }
        if (korder == 2) {
            t.tx += 2;
            t.tapidx += 2;
            t.tappix += 2;
            t.woff += SX2;
            if (t.tx >= TWv) {
                t.tx -= TWv;
                t.ty += 2;
                t.tapidx += TWv;
                t.tappix += 2 * xW - TWv;
                t.woff += DY2;
                if (t.ty >= THv) {      // next parity class: (0,0) (0,1) (1,0) (1,1), then the next chunk
                    t.ty &= 1;
                    if (t.tx == 0) t.tx = 1;
                    else {
                        t.tx = 0;
                        if (t.ty == 0) t.ty = 1;
                        else { t.ty = 0; t.chunk += 1; }
                    }
                    kt_place(t);
                }
            }
            return;
        }
        t.chunk += 1;
        t.woff += KB;
        if (t.chunk == tpt) {
            t.chunk = 0;
            t.tx += 1;
            t.tapidx += 1;
            t.tappix += 1;
            t.woff += DX;
            if (t.tx == TWv) {
                t.tx = 0;
                t.tappix += xW - TWv;
                t.woff += DY;
            }
        }
    };

    // a staged K-tile of the gathered side in registers: raw rows, their 1.0 / 0.0 validity, the source's norm table and slope
    struct ASet { float4 r[A_ROWS]; float v[A_ROWS]; float4 aa, ab; float slope; uint4 km; };

    auto issue_loads = [&](const KTile& t, ASet& S) {
        const bool first = ONE ? true : t.chunk < nch0;
        const int cs = first ? xC0 : xC1;
        const int cc = (first ? t.chunk : t.chunk - nch0) * BK;
        const char* sbase = reinterpret_cast<const char*>((first ? xs0 : xs1) + cc);
        const int tapshift = t.tappix * cs * 4;
        const int tapidx = t.tapidx;
        // KM: does this thread's float4 of the chunk exist?  (a wave-uniform test on the chunk, a per-thread constant for the last one)
        const bool cv = !KM || t.chunk != nch0 - 1 || a_lastv;
        if (KM) S.km = t.chunk == nch0 - 1 ? a_lastm : make_uint4(~0u, ~0u, ~0u, ~0u);
        if (!PLAIN) {
            const char* pa = reinterpret_cast<const char*>((first ? tab.a0 : tab.a1) + cc);
            const char* pb = reinterpret_cast<const char*>((first ? tab.b0 : tab.b1) + cc);
            const int tc = (KM && !cv) ? 0 : a_col4 * 16;       // no table entries beyond the source
            S.aa = *reinterpret_cast<const float4*>(pa + tc);
            S.ab = *reinterpret_cast<const float4*>(pb + tc);
            S.slope = first ? slope0 : slope1;
        }
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const unsigned vb = KM ? (cv ? (a_vm[i] >> tapidx) & 1u : 0u) : (a_vm[i] >> tapidx) & 1u;
            const int osel = first ? a_off0[i] : a_off1[i];
            const unsigned off = vb ? (unsigned)(osel + tapshift) : (unsigned)(a_col4 * 16);
            S.v[i] = (float)vb;
            if (SSC_BF_DIAG_BUILD & 8) { S.r[i] = make_float4(1.f, 2.f, 3.f, 4.f); continue; }
            S.r[i] = *reinterpret_cast<const float4*>(sbase + off);
        }
    };

    // filter K-tile `t` straight into LDS buffer `buf`: run `half` (0 / 1) of this wave's fragments
    constexpr int RUN0 = (B_IPW + 1) / 2, RUN1 = B_IPW - RUN0;
    auto dma_b = [&](const KTile& t, int buf, int half) {
        if (SSC_BF_DIAG_BUILD & 2) return;
        const char* wtap = wsp + t.woff;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(B_BASE + buf * B_BYTES + wave * B_IPW * 1024));
        if (half == 0) glds16_run<RUN0>(wtap, bd_off, dst);
        else glds16_run<RUN1>(wtap, bd_off + RUN0, dst + RUN0 * 1024);
    };

    // row i of the staged K-tile in two pieces: transform + split of the first pair | split of the second pair + three 8-byte stores
    // (plain scalars between the pieces: a struct handed from one lambda to the other went through scratch memory)
#define BF_STAGE_A(S, i, Z, W, H0, M0, L0)                                                        \
    do {                                                                                          \
        float4 v_ = (S).r[i];                                                                     \
        if (SSC_BF_DIAG_BUILD & 32) {                                                             \
        } else if (PLAIN) {                                                                       \
            v_.x *= (S).v[i]; v_.y *= (S).v[i]; v_.z *= (S).v[i]; v_.w *= (S).v[i];               \
        } else {                                                                                  \
            float t_;                                                                             \
            t_ = fmaf((S).aa.x, v_.x, (S).ab.x); v_.x = fmaxf(t_, t_ * (S).slope) * (S).v[i];     \
            t_ = fmaf((S).aa.y, v_.y, (S).ab.y); v_.y = fmaxf(t_, t_ * (S).slope) * (S).v[i];     \
            t_ = fmaf((S).aa.z, v_.z, (S).ab.z); v_.z = fmaxf(t_, t_ * (S).slope) * (S).v[i];     \
            t_ = fmaf((S).aa.w, v_.w, (S).ab.w); v_.w = fmaxf(t_, t_ * (S).slope) * (S).v[i];     \
        }                                                                                         \
        if (KM) {                                                                                 \
            v_.x = __uint_as_float(__float_as_uint(v_.x) & (S).km.x);                             \
            v_.y = __uint_as_float(__float_as_uint(v_.y) & (S).km.y);                             \
            v_.z = __uint_as_float(__float_as_uint(v_.z) & (S).km.z);                             \
            v_.w = __uint_as_float(__float_as_uint(v_.w) & (S).km.w);                             \
        }                                                                                         \
        split3_pair(v_.x, v_.y, H0, M0, L0);                                                      \
        Z = v_.z; W = v_.w;                                                                       \
    } while (0)
#define BF_STAGE_B(buf, i, Z, W, H0, M0, L0)                                                      \
    do {                                                                                          \
        unsigned h1_, m1_, l1_;                                                                   \
        split3_pair(Z, W, h1_, m1_, l1_);                                                         \
        char* p_ = sm_b + (buf) * A_BYTES + (arow + 32 * (i)) * BF_A_RS + a_col4 * 8;             \
        if (SSC_BF_DIAG_BUILD & 4) break;                                                         \
        *reinterpret_cast<uint2*>(p_) = make_uint2(H0, h1_);                                      \
        *reinterpret_cast<uint2*>(p_ + 64) = make_uint2(M0, m1_);                                 \
        *reinterpret_cast<uint2*>(p_ + 128) = make_uint2(L0, l1_);                                \
    } while (0)

    if (kt_begin < kt_end) {
        const int last = kt_end - 1;
        ASet S0, S1;        // two register sets of staged K-tiles, used in turn (no copies): one is written to LDS while the other fills
        KTile tl = kt_decode(kt_begin);     // decode of the K-tile the next issue_loads takes
        dma_b(tl, 0, 0);
        if (RUN1 > 0) dma_b(tl, 0, 1);
        issue_loads(tl, S0);
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            float z, w;
            unsigned h0, m0_, l0;
            BF_STAGE_A(S0, i, z, w, h0, m0_, l0);
            BF_STAGE_B(0, i, z, w, h0, m0_, l0);
        }
        KTile td = tl;                      // decode of the K-tile the next dma_b takes
        if (kt_begin + 1 <= last) kt_advance(tl);      // (the clamp at the last K-tile: stay)
        td = tl;
        issue_loads(tl, S0);
        BF_WAIT_ALL();
        __builtin_amdgcn_s_barrier();
        // one K step: MFMAs of K-tile kt from LDS buffer `cur`; set SA (K-tile kt+1) -> LDS buffer cur ^ 1; K-tile kt+2 -> set SB
        auto kstep = [&](ASet& SA, ASet& SB, const int cur_set, const int kt) {
            const int cur = SS ? 0 : cur_set, nxt = SS ? 0 : cur_set ^ 1;       // LDS stages read / written in this step
            const char* Ab = sm_b + cur * A_BYTES + (wm * SM * 32 + l31) * BF_A_RS + lhi * 16;
            const char* Bb = sm_b + B_BASE + cur * B_BYTES + (wn * SN) * 1024 + lane * 16;
            bf16x8 av[2][SM][3], bv[2][SN][3];
            auto fetch_a = [&](int kc) {
#pragma unroll
                for (int i = 0; i < SM; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        if (SSC_BF_DIAG_BUILD & 64) { asm volatile("" : "=v"(av[kc][i][p])); continue; }
                        av[kc][i][p] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * BF_A_RS + p * 64 + kc * 32);
                    }
            };
            auto fetch_b = [&](int kc) {
#pragma unroll
                for (int j = 0; j < SN; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        if (SSC_BF_DIAG_BUILD & 64) { asm volatile("" : "=v"(bv[kc][j][p])); continue; }
                        bv[kc][j][p] = *reinterpret_cast<const bf16x8*>(Bb + ((kc * 3 + p) * NBT + j) * 1024);
                    }
            };
            // one product over the wave's blocks (consecutive MFMAs go to different accumulators); products smallest first
            auto group = [&](int kc, int t) {
                if (SSC_BF_DIAG_BUILD & 1) return;
                constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < SM; ++i)
#pragma unroll
                    for (int j = 0; j < SN; ++j) {
                        if (t == 5)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[kc][i][0], bv[kc][j][0], acc[i][j], 0, 0, 0);
                        else
                            accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[kc][i][pa[t]], bv[kc][j][pb[t]], accc[i][j], 0, 0, 0);
                    }
            };
#define BF_SB __builtin_amdgcn_sched_barrier(0)
            // The K step, hand-placed: twelve groups of G MFMAs with one piece of the step's other work behind each, full
            // scheduling barriers in between (the compiler's own order put the six DMA instructions and their scalar code in one
            // run without a single MFMA).  Right behind the barrier that freed buffer cur ^ 1: the filter tile of K-tile kt+1 by
            // DMA into it and the gather loads of K-tile kt+2 into the second register set -- both have the whole step to land.
            fetch_a(0);
            fetch_b(0);
            if (SS) {       // one stage: every wave takes ALL its operands of this K-tile now; then the stage may be overwritten
                fetch_a(1);
                fetch_b(1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            BF_SB;
            float z0, w0, z1, w1;
            unsigned ha, ma, la, hb, mb, lb;
            group(0, 0); dma_b(td, nxt, 0); BF_SB;
            group(0, 1); if (RUN1 > 0) dma_b(td, nxt, 1); BF_SB;
            if (kt + 2 <= last) kt_advance(tl);
            group(0, 2); issue_loads(tl, SB); BF_SB;
            group(0, 3); if (!SS) fetch_a(1); BF_SB;
            group(0, 4); if (!SS) fetch_b(1); BF_SB;
            // K-tile kt+1: registers -> the LDS buffer nobody reads now
            if (A_ROWS == 2) {
                group(0, 5); BF_STAGE_A(SA, 0, z0, w0, ha, ma, la); BF_SB;
                group(1, 0); BF_STAGE_B(nxt, 0, z0, w0, ha, ma, la); BF_SB;
                group(1, 1); BF_STAGE_A(SA, 1, z1, w1, hb, mb, lb); BF_SB;
                group(1, 2); BF_STAGE_B(nxt, 1, z1, w1, hb, mb, lb); BF_SB;
                group(1, 3); BF_SB;
                group(1, 4); BF_SB;
                group(1, 5); BF_SB;
            } else {
                group(0, 5); BF_STAGE_A(SA, 0, z0, w0, ha, ma, la); BF_SB;
                group(1, 0); BF_STAGE_B(nxt, 0, z0, w0, ha, ma, la); BF_STAGE_A(SA, 1, z1, w1, hb, mb, lb); BF_SB;
                group(1, 1); BF_STAGE_B(nxt, 1, z1, w1, hb, mb, lb); BF_STAGE_A(SA, 2, z0, w0, ha, ma, la); BF_SB;
                group(1, 2); BF_STAGE_B(nxt, 2, z0, w0, ha, ma, la); BF_STAGE_A(SA, 3, z1, w1, hb, mb, lb); BF_SB;
                group(1, 3); BF_STAGE_B(nxt, 3, z1, w1, hb, mb, lb); BF_SB;
                group(1, 4); BF_SB;
                group(1, 5); BF_SB;
            }
#undef BF_SB
            // everything issued in this step has landed: the DMA tile and this wave's ds_writes are in LDS (published by the
            // barrier), the loads of K-tile kt+2 are in registers and become the staged set
            BF_WAIT_ALL();
            td = tl;
            __builtin_amdgcn_s_barrier();
        };
        for (int kt = kt_begin; kt < kt_end; kt += 2) {
            kstep(S0, S1, 0, kt);
            if (kt + 1 < kt_end) kstep(S1, S0, 1, kt + 1);
        }
    }

#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += accc[i][j][r];
    ut_epilogue<BM, BN, WM, WN, SM, SN, LDS_FLOATS>(acc, d, smem, rowpix, part, ks, sk, slot, flags, ts_s, splitk, slab_base,
                                                   slab_stride, m0, n0, phase, M, g_sk_cfg_bf);
}

// ---------------------------------------------------------------------------------------------
// 128 x 128 tile on 16-k LDS stages (round 6)
// ---------------------------------------------------------------------------------------------
// What bounds conv_bf_kernel is the number of issue slots beside its MFMAs, spread over the gathered side's path, the filter DMA
// and the LDS operand reads (profiles/NOTEBOOK_r06.md section 1).  A wave tile of 64 x 64 instead of 32 x 64 halves the filter DMA
// and cuts the LDS reads by a third per MFMA -- but with 32-k K-tiles it needs 100 KB of LDS or 300 registers, one workgroup per CU
// (-9 % in round 5).  Here an LDS stage holds 16 k: 26 KB per stage, the fragments of ONE 16-k half in registers, 24 MFMAs per wave
// and barrier as before -- two workgroups per CU (75 KB with the epilogue's C image, ~220 registers).
//   A stage [128 rows][112 bytes]: three planes of 32 bytes (16 bf16) + 16 bytes of padding (28 banks per row: the 16-lane groups
//     of a ds_read_b128 over 32 rows are conflict-free);  B stage [plane 3][4 blocks][1 KiB fragment].
//   Channel counts multiples of 32 (the uniform form); K-tiles are 16-channel chunks of one tap and one source.
#define BFH_A_RS 112
template <int BM, int BN> struct BfhLds {
    static constexpr int A_BYTES = BM * BFH_A_RS;
    static constexpr int B_BYTES = 3 * (BN / 32) * 1024;
    static constexpr int OPER_BYTES = 2 * (A_BYTES + B_BYTES);
    static constexpr int EPI_BYTES = (BM * (BN + 4) + 2 * 256 * 4) * 4;
    static constexpr int LDS_BYTES = OPER_BYTES > EPI_BYTES ? OPER_BYTES : EPI_BYTES;
    static constexpr int TOTAL = LDS_BYTES + BM * 8;
};

// <WM, WN>: 2 x 2 waves = the 128 x 128 tile.  (1 x 4 waves = 64 x 256 -- every wave the same 64 rows, half the gathered-side work
// per MFMA instead of half the filter DMA -- measured equal on the 256- and 512-column layers, 131 / 132 / 176 vs 131 / 129 / 175
// TFLOP/s: not instantiated.)
// KM (with ONE): the source's channel count is any multiple of 4 (conv_bf_kernel<KM>'s contract): the last 16-wide chunk of every tap
// is partly empty -- and a tap costs ceil(C / 16) chunks here instead of 2 * ceil(C / 32) (68 channels: 80 k instead of 96)
template <int WM, int WN, bool PLAIN, bool ONE, bool KM = false>
__global__ __launch_bounds__(256, 2) void conv_bfh_kernel(const ssc_conv_desc d, const Magics mg, float* __restrict__ slab_base,
                                                           long slab_stride, int splitk, int ts_full, int ts_s,
                                                           unsigned* __restrict__ flags, const BfTabs tab, const int korder) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int SM = 2, SN = 2, BM = WM * SM * 32, BN = WN * SN * 32, NBT = BN / 32, KS = 16;
    constexpr int A_BYTES = BfhLds<BM, BN>::A_BYTES, B_BYTES = BfhLds<BM, BN>::B_BYTES;
    constexpr int LDS_FLOATS = BfhLds<BM, BN>::LDS_BYTES / 4;
    constexpr int B_BASE = 2 * A_BYTES;
    constexpr int A_ROWS = BM / 64;         // float4s per thread and stage: 4 threads cover a row's 16 k
    constexpr int B_IPW = 3 * NBT / 4;      // 3 planes x NBT fragments per stage over 4 waves

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm_b = reinterpret_cast<char*>(smem);
    long* rowpix = reinterpret_cast<long*>(smem + LDS_FLOATS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    const float* const xs0 = d.x.s0;
    const float* const xs1 = d.x.s1;
    const int xC0 = d.x.C0, xC1 = d.x.C1, xH = d.x.H, xW = d.x.W;
    const int TWv = d.TW, kstep = d.kstep, KWv = d.KW;
    const float slope0 = act_slope(d.x.act), slope1 = act_slope(d.x.act1 >= 0 ? d.x.act1 : d.x.act);

    static_assert(!KM || ONE, "partial chunks: one source only");
    const int nch0 = KM ? (xC0 + KS - 1) / KS : xC0 / KS, nch1 = xC1 / KS;
    const int tpt = nch0 + nch1;
    const long M = (long)d.NB * d.PH * d.PW;
    const int PHW = d.PH * d.PW;
    // workgroup -> (tile, K range): conv_bf_kernel's layouts
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
            if (bid < ts_full) tile = (bid & 7) * (ts_full >> 3) + (bid >> 3);
            const int r2 = tile / nt;
            n0 = (tile - r2 * nt) * BN;
            if (ts_s & 0x20000) {
                phase = r2 & 3;
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

    // gathered side: thread t stages the float4 (4 k) at piece t & 3 of the rows arow + 64 i.  arow: the 64 rows t >> 2 with the
    // row's low bits rotated so that the four rows of a 16-lane ds_write_b64 group lie two rows (224 bytes = 56 banks = 24 mod
    // 32) apart: their 32-byte pieces fall on banks 0-7, 24-31, 16-23, 8-15
    const int a_col4 = tid & 3;
    const int aq = tid >> 2;
    const int arow = (aq & ~7) | ((aq & 3) << 1) | ((aq >> 2) & 1);
    const bool a_lastv = !KM || (nch0 - 1) * KS + a_col4 * 4 < xC0;      // KM: this thread's float4 of the LAST chunk exists
    uint4 a_lastm = make_uint4(~0u, ~0u, ~0u, ~0u);                      // ... and which of its elements are real channels (< k_real)
    if (KM) {
        const int c_ = (nch0 - 1) * KS + a_col4 * 4;
        a_lastm = make_uint4(c_ < d.k_real ? ~0u : 0u, c_ + 1 < d.k_real ? ~0u : 0u, c_ + 2 < d.k_real ? ~0u : 0u,
                             c_ + 3 < d.k_real ? ~0u : 0u);
    }
    int a_iyb[A_ROWS], a_ixb[A_ROWS], a_off0[A_ROWS], a_off1[A_ROWS];
    bool a_mv[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int row = arow + 64 * i;
        const long m = m0 + row;
        a_mv[i] = m < M;
        const long mm = a_mv[i] ? m : 0;
        int n, rem, py;
        if (mg.use32) {
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
        a_off0[i] = (pix0 * xC0 + a_col4 * 4) * 4;
        a_off1[i] = (pix0 * xC1 + a_col4 * 4) * 4;
        if (a_col4 == 0)
            rowpix[row] = ((long)n * d.OH + py * d.out_stride + ph.ooff_y) * d.OW + px * d.out_stride + ph.ooff_x;
    }
    unsigned a_vm[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) a_vm[i] = 0u;
    {
        const int ntaps = d.TH * TWv;
        int ty = 0, tx = 0;
        for (int t = 0; t < ntaps; ++t) {
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                const bool v = a_mv[i] & ((unsigned)(a_iyb[i] + ty) < (unsigned)xH) & ((unsigned)(a_ixb[i] + tx) < (unsigned)xW);
                a_vm[i] |= v ? (1u << t) : 0u;
            }
            tx += 1;
            if (tx == TWv) { tx = 0; ty += 1; }
        }
    }

    // filter side: fragment q of this wave = f = wave * 3 + q of the stage's [plane][block] image
    const int NBP = d.ws_nbp, KC = d.ws_kc;         // KC: 16-k chunks per tap in the planes
    const int nb0 = (d.n_off + n0) >> 5;
    unsigned bd_off[B_IPW];
#pragma unroll
    for (int q = 0; q < B_IPW; ++q) {
        const int f = wave * B_IPW + q;
        const int pl = f / NBT, nbl = f % NBT;
        bd_off[q] = (unsigned)(((pl * NBP) + nb0 + nbl) * 1024 + lane * 16);
    }
    const char* const wsp = reinterpret_cast<const char*>(d.wsplit);
    const long KB = (long)3 * NBP * 1024;           // one 16-k chunk: 3 planes x NBP blocks

    const int nkt = d.TH * d.TW * tpt;
    const int per = (nkt + sk - 1) / sk;
    const int kt_begin = ks * per;
    const int kt_end = min(nkt, kt_begin + per);

    f32x16 acc[SM][SN], accc[SM][SN];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = accc[i][j][r] = 0.f;

    // K-tile sequence: conv_bf_kernel's three orders over 16-k chunks
    struct KTile { int ty, tx, chunk, tappix, tapidx; long woff; };
    const int THv = d.TH, ntaps = d.TH * d.TW;
    auto kt_place = [&](KTile& t) {
        t.tappix = t.ty * xW + t.tx;
        t.tapidx = t.ty * TWv + t.tx;
        t.woff = ((long)((ph.ky0 + t.ty * kstep) * KWv + ph.kx0 + t.tx * kstep) * KC + t.chunk) * KB;
    };
    const long SX = (long)kstep * KC * KB, SY = (long)kstep * KWv * KC * KB;
    const long DX = SX - (long)tpt * KB, DY = SY - (long)TWv * SX;
    auto kt_decode = [&](int kt) {
        KTile t;
        if (korder != 0) {
            t.chunk = kt / ntaps;
            const int sidx = kt - t.chunk * ntaps;
            if (korder == 1) {
                t.ty = sidx / TWv;
                t.tx = sidx - t.ty * TWv;
            } else {
                const int hw = TWv >> 1, pc = ntaps >> 2;
                const int cls = sidx / pc, j = sidx - cls * pc;
                const int jy = j / hw, jx = j - jy * hw;
                t.ty = 2 * jy + (cls >> 1);
                t.tx = 2 * jx + (cls & 1);
            }
            kt_place(t);
            return t;
        }
        const int tap = div32(kt, mg.mC, mg.oneC);      // mC: magic of tpt
        t.chunk = kt - tap * tpt;
        t.ty = div32(tap, mg.mTW, mg.oneTW);
        t.tx = tap - t.ty * TWv;
        kt_place(t);
        return t;
    };
    const long W0 = (long)(ph.ky0 * KWv + ph.kx0) * KC * KB;
    const long SX2 = 2 * SX, DY2 = 2 * SY - (long)TWv * SX;
    auto kt_advance = [&](KTile& t) {
        if (korder == 1) {
            t.tx += 1;
            t.tapidx += 1;
            t.tappix += 1;
            t.woff += SX;
            if (t.tx == TWv) {
                t.tx = 0;
                t.ty += 1;
                t.tappix += xW - TWv;
                t.woff += DY;
                if (t.ty == THv) {
                    t.ty = 0;
                    t.chunk += 1;
                    t.tapidx = 0;
                    t.tappix = 0;
                    t.woff = W0 + (long)t.chunk * KB;
                }
            }
            return;
        }
        if (korder == 2) {
            t.tx += 2;
            t.tapidx += 2;
            t.tappix += 2;
            t.woff += SX2;
            if (t.tx >= TWv) {
                t.tx -= TWv;
                t.ty += 2;
                t.tapidx += TWv;
                t.tappix += 2 * xW - TWv;
                t.woff += DY2;
                if (t.ty >= THv) {
                    t.ty &= 1;
                    if (t.tx == 0) t.tx = 1;
                    else {
                        t.tx = 0;
                        if (t.ty == 0) t.ty = 1;
                        else { t.ty = 0; t.chunk += 1; }
                    }
                    kt_place(t);
                }
            }
            return;
        }
        t.chunk += 1;
        t.woff += KB;
        if (t.chunk == tpt) {
            t.chunk = 0;
            t.tx += 1;
            t.tapidx += 1;
            t.tappix += 1;
            t.woff += DX;
            if (t.tx == TWv) {
                t.tx = 0;
                t.tappix += xW - TWv;
                t.woff += DY;
            }
        }
    };

    struct ASet { float4 r[A_ROWS]; float v[A_ROWS]; float4 aa, ab; float slope; uint4 km; };
    auto issue_loads = [&](const KTile& t, ASet& S) {
        const bool first = ONE ? true : t.chunk < nch0;
        const int cs = first ? xC0 : xC1;
        const int cc = (first ? t.chunk : t.chunk - nch0) * KS;
        const char* sbase = reinterpret_cast<const char*>((first ? xs0 : xs1) + cc);
        const int tapshift = t.tappix * cs * 4;
        const int tapidx = t.tapidx;
        const bool cv = !KM || t.chunk != nch0 - 1 || a_lastv;
        if (KM) S.km = t.chunk == nch0 - 1 ? a_lastm : make_uint4(~0u, ~0u, ~0u, ~0u);
        if (!PLAIN) {
            const char* pa = reinterpret_cast<const char*>((first ? tab.a0 : tab.a1) + cc);
            const char* pb = reinterpret_cast<const char*>((first ? tab.b0 : tab.b1) + cc);
            const int tc = (KM && !cv) ? 0 : a_col4 * 16;       // no table entries beyond the source
            S.aa = *reinterpret_cast<const float4*>(pa + tc);
            S.ab = *reinterpret_cast<const float4*>(pb + tc);
            S.slope = first ? slope0 : slope1;
        }
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const unsigned vb = cv ? (a_vm[i] >> tapidx) & 1u : 0u;
            const int osel = first ? a_off0[i] : a_off1[i];
            const unsigned off = vb ? (unsigned)(osel + tapshift) : (unsigned)(a_col4 * 16);
            S.v[i] = (float)vb;
            S.r[i] = *reinterpret_cast<const float4*>(sbase + off);
        }
    };
    auto dma_b = [&](const KTile& t, int buf) {
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(B_BASE + buf * B_BYTES + wave * B_IPW * 1024));
        glds16_run<3>(wsp + t.woff, bd_off, dst);
        if (B_IPW == 6) glds16_run<3>(wsp + t.woff, bd_off + 3, dst + 3 * 1024);
    };
#define BFH_STAGE_A(S, i, Z, W, H0, M0, L0)                                                       \
    do {                                                                                          \
        float4 v_ = (S).r[i];                                                                     \
        if (PLAIN) {                                                                              \
            v_.x *= (S).v[i]; v_.y *= (S).v[i]; v_.z *= (S).v[i]; v_.w *= (S).v[i];               \
        } else {                                                                                  \
            float t_;                                                                             \
            t_ = fmaf((S).aa.x, v_.x, (S).ab.x); v_.x = fmaxf(t_, t_ * (S).slope) * (S).v[i];     \
            t_ = fmaf((S).aa.y, v_.y, (S).ab.y); v_.y = fmaxf(t_, t_ * (S).slope) * (S).v[i];     \
            t_ = fmaf((S).aa.z, v_.z, (S).ab.z); v_.z = fmaxf(t_, t_ * (S).slope) * (S).v[i];     \
            t_ = fmaf((S).aa.w, v_.w, (S).ab.w); v_.w = fmaxf(t_, t_ * (S).slope) * (S).v[i];     \
        }                                                                                         \
        if (KM) {                                                                                 \
            v_.x = __uint_as_float(__float_as_uint(v_.x) & (S).km.x);                             \
            v_.y = __uint_as_float(__float_as_uint(v_.y) & (S).km.y);                             \
            v_.z = __uint_as_float(__float_as_uint(v_.z) & (S).km.z);                             \
            v_.w = __uint_as_float(__float_as_uint(v_.w) & (S).km.w);                             \
        }                                                                                         \
        split3_pair(v_.x, v_.y, H0, M0, L0);                                                      \
        Z = v_.z; W = v_.w;                                                                       \
    } while (0)
#define BFH_STAGE_B(buf, i, Z, W, H0, M0, L0)                                                     \
    do {                                                                                          \
        unsigned h1_, m1_, l1_;                                                                   \
        split3_pair(Z, W, h1_, m1_, l1_);                                                         \
        char* p_ = sm_b + (buf) * A_BYTES + (arow + 64 * (i)) * BFH_A_RS + a_col4 * 8;            \
        *reinterpret_cast<uint2*>(p_) = make_uint2(H0, h1_);                                      \
        *reinterpret_cast<uint2*>(p_ + 32) = make_uint2(M0, m1_);                                 \
        *reinterpret_cast<uint2*>(p_ + 64) = make_uint2(L0, l1_);                                 \
    } while (0)

    if (kt_begin < kt_end) {
        const int last = kt_end - 1;
        ASet S0, S1;
        KTile tl = kt_decode(kt_begin);
        dma_b(tl, 0);
        issue_loads(tl, S0);
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            float z, w;
            unsigned h0, m0_, l0;
            BFH_STAGE_A(S0, i, z, w, h0, m0_, l0);
            BFH_STAGE_B(0, i, z, w, h0, m0_, l0);
        }
        KTile td = tl;
        if (kt_begin + 1 <= last) kt_advance(tl);
        td = tl;
        issue_loads(tl, S0);
        BF_WAIT_ALL();
        __builtin_amdgcn_s_barrier();
        // one step = one 16-k stage: MFMAs of K-tile kt from stage `cur`; set SA (K-tile kt+1) -> stage cur ^ 1; K-tile kt+2 -> set SB
        auto kstep = [&](ASet& SA, ASet& SB, const int cur, const int kt) {
            const int nxt = cur ^ 1;
            const char* Ab = sm_b + cur * A_BYTES + (wm * SM * 32 + l31) * BFH_A_RS + lhi * 16;
            const char* Bb = sm_b + B_BASE + cur * B_BYTES + (wn * SN) * 1024 + lane * 16;
            bf16x8 av[SM][3], bv[SN][3];
#pragma unroll
            for (int i = 0; i < SM; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) av[i][p] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * BFH_A_RS + p * 32);
#pragma unroll
            for (int j = 0; j < SN; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) bv[j][p] = *reinterpret_cast<const bf16x8*>(Bb + (p * NBT + j) * 1024);
            auto group = [&](int t) {
                constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < SM; ++i)
#pragma unroll
                    for (int j = 0; j < SN; ++j) {
                        if (t == 5) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][0], bv[j][0], acc[i][j], 0, 0, 0);
                        else accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][pa[t]], bv[j][pb[t]], accc[i][j], 0, 0, 0);
                    }
            };
#define BFH_SB __builtin_amdgcn_sched_barrier(0)
            BFH_SB;
            float z0, w0, z1, w1;
            unsigned ha, ma, la, hb, mb, lb;
            group(0); dma_b(td, nxt); BFH_SB;
            if (kt + 2 <= last) kt_advance(tl);
            group(1); issue_loads(tl, SB); BFH_SB;
            group(2); BFH_STAGE_A(SA, 0, z0, w0, ha, ma, la); BFH_SB;
            group(3); BFH_STAGE_B(nxt, 0, z0, w0, ha, ma, la); BFH_SB;
            if (A_ROWS == 2) {
                group(4); BFH_STAGE_A(SA, A_ROWS - 1, z1, w1, hb, mb, lb); BFH_SB;
                group(5); BFH_STAGE_B(nxt, A_ROWS - 1, z1, w1, hb, mb, lb); BFH_SB;
            } else {
                group(4); BFH_SB;
                group(5); BFH_SB;
            }
#undef BFH_SB
            BF_WAIT_ALL();
            td = tl;
            __builtin_amdgcn_s_barrier();
        };
        for (int kt = kt_begin; kt < kt_end; kt += 2) {
            kstep(S0, S1, 0, kt);
            if (kt + 1 < kt_end) kstep(S1, S0, 1, kt + 1);
        }
    }
#undef BFH_STAGE_A
#undef BFH_STAGE_B

#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += accc[i][j][r];
    ut_epilogue<BM, BN, WM, WN, SM, SN, LDS_FLOATS>(acc, d, smem, rowpix, part, ks, sk, slot, flags, ts_s, splitk, slab_base,
                                                   slab_stride, m0, n0, phase, M, g_sk_cfg_bf);
}

#ifdef SSC_ISA_ONLY
template __global__ void conv_bf_kernel<2, 2, 1, 2, false, true>(const ssc_conv_desc, const Magics, float*, long, int, int, int, unsigned*, const BfTabs, const int);
template __global__ void conv_bf_kernel<2, 2, 1, 2, false, false>(const ssc_conv_desc, const Magics, float*, long, int, int, int, unsigned*, const BfTabs, const int);
#else
// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void ssc_launch_slab_reduce(const float* ws, long out_count, int splitk, const ssc_conv_desc& d, hipStream_t st);     // igemm.hip

// ones[4096] then zeros[4096] in device memory (per device): the norm "table" of a source that has none
#define BF_IDENT_C 4096
static const float* bf_identity_rows() {
    static float* rows[16] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (rows[dev] == nullptr) {
        float* p = nullptr;
        if (hipMalloc(&p, 2 * BF_IDENT_C * sizeof(float)) != hipSuccess) return nullptr;
        float* h = (float*)malloc(2 * BF_IDENT_C * sizeof(float));
        for (int i = 0; i < BF_IDENT_C; ++i) { h[i] = 1.f; h[BF_IDENT_C + i] = 0.f; }
        const hipError_t e = hipMemcpy(p, h, 2 * BF_IDENT_C * sizeof(float), hipMemcpyHostToDevice);
        free(h);
        if (e != hipSuccess) return nullptr;
        rows[dev] = p;
    }
    return rows[dev];
}
extern "C" int ssc_bf16_prepare(void) { return bf_identity_rows() != nullptr ? 0 : -1; }

static bool bf_tabs(const ssc_conv_desc& d, BfTabs& t) {
    const float* id = bf_identity_rows();
    if (id == nullptr || d.x.C0 > BF_IDENT_C || d.x.C1 > BF_IDENT_C) return false;
    t.a0 = d.x.ab0 ? d.x.ab0 : id;
    t.b0 = d.x.ab0 ? d.x.ab0 + d.x.C0 : id + BF_IDENT_C;
    t.a1 = d.x.ab1 ? d.x.ab1 : id;
    t.b1 = d.x.ab1 ? d.x.ab1 + d.x.C1 : id + BF_IDENT_C;
    return true;
}

template <int WM, int WN, int SM, int SN, bool PLAIN, bool ONE, bool KM = false, bool SS = false>
static int launch_bf_t(const ssc_conv_desc& d, int splitk, float* ws, hipStream_t st, long ts_full, int ts_s, int64_t ws_bytes,
                       int xcd) {
    constexpr int BM = WM * SM * 32, BN = WN * SN * 32;
    constexpr size_t lds = BfLds<BM, BN, SS>::TOTAL;
    const long M = (long)d.NB * d.PH * d.PW;
    const int tpt = (d.x.C0 + BK - 1) / BK + d.x.C1 / BK;
    const Magics mg = make_magics((unsigned)tpt, (unsigned)d.TW, (unsigned long)d.PW, (unsigned long)d.PH * d.PW, (unsigned long)M);
    const long mt = (M + BM - 1) / BM;
    const int nt = (d.Nstore + BN - 1) / BN;
    const long out_count = (long)d.NB * d.OH * d.OW * d.ldc;
    static unsigned long long attr_done = 0;
    {
        const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&conv_bf_kernel<WM, WN, SM, SN, PLAIN, ONE, KM, SS>), (int)lds, &attr_done);
        if (arc != 0) return arc;
    }
    BfTabs tab = {nullptr, nullptr, nullptr, nullptr};
    if (!PLAIN && !bf_tabs(d, tab)) return -5;
    const int xflag = 0x10000 | ((xcd >= 2 && d.nphase == 4) ? 0x20000 : 0);
    static int diag = -1;       // SSC_BF_DIAG bit 0: no in-launch K slices (plain tiles), bit 1: no XCD-aware order (diagnostics)
    if (diag < 0) {
        const char* e = ssc_dev_getenv("SSC_BF_DIAG");
        diag = e != nullptr ? atoi(e) : 0;
    }
    if (diag & 2) xcd = 0;
    static int kord = -2;       // SSC_BF_KORDER=0 / 1 / 2 pins the K-tile order (A/B); default: by the launch's geometry
    if (kord == -2) {
        const char* e = ssc_dev_getenv("SSC_BF_KORDER");
        kord = e != nullptr ? atoi(e) : -1;
    }
    const bool par_ok = d.in_stride == 2 && (d.TH & 1) == 0 && (d.TW & 1) == 0;
    int korder = par_ok ? 2 : 1;
    if (kord >= 0) korder = (kord == 2 && !par_ok) ? 1 : kord;
    if (d.TH * d.TW == 1) korder = 0;
    if (!(diag & 1) && splitk == 1 && ws != nullptr && d.sk_flags != nullptr && ts_s > 1) {        // whole tiles + K slices combined in the launch
        const long tiles = mt * nt * d.nphase;
        const long full = ts_full, tail = tiles - full, s = ts_s;
        if (full >= 0 && tail > 0 && (int64_t)tail * s * BM * BN * 4 <= ws_bytes && tail * s < SSC_SK_FLAG_WORDS - 1 &&
            full + tail * s < 0x7fffffffL) {
            hipLaunchKernelGGL((conv_bf_kernel<WM, WN, SM, SN, PLAIN, ONE, KM, SS>), dim3((unsigned)(full + tail * s)), dim3(256), lds, st, d, mg,
                               ws, out_count, 1, (int)full, (int)s | ((xcd && (full & 7) == 0) ? xflag : 0), d.sk_flags, tab, korder);
            return (int)hipGetLastError();
        }
    }
    if (splitk == 1 && xcd) {       // whole tiles only, 1-D grid in the XCD-aware order
        const long tiles = mt * nt * d.nphase;
        const long full = tiles & ~7L;
        if (tiles < 0x7fffffffL && full > 0) {
            hipLaunchKernelGGL((conv_bf_kernel<WM, WN, SM, SN, PLAIN, ONE, KM, SS>), dim3((unsigned)tiles), dim3(256), lds, st, d, mg, ws,
                               out_count, 1, (int)full, 1 | xflag, (unsigned*)nullptr, tab, korder);
            return (int)hipGetLastError();
        }
    }
    dim3 grid((unsigned)mt, (unsigned)nt, (unsigned)(d.nphase * splitk));
    hipLaunchKernelGGL((conv_bf_kernel<WM, WN, SM, SN, PLAIN, ONE, KM, SS>), grid, dim3(256), lds, st, d, mg, ws, out_count, splitk, 0, 0,
                       (unsigned*)nullptr, tab, korder);
    if (splitk > 1) ssc_launch_slab_reduce(ws, out_count, splitk, d, st);
    return (int)hipGetLastError();
}

// the 128 x 128 tile on 16-k stages (conv_bfh_kernel): launch_bf_t's three grid layouts
template <int WM, int WN, bool PLAIN, bool ONE, bool KM = false>
static int launch_bfh_t(const ssc_conv_desc& d, int splitk, float* ws, hipStream_t st, long ts_full, int ts_s, int64_t ws_bytes,
                        int xcd) {
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr size_t lds = BfhLds<BM, BN>::TOTAL;
    const long M = (long)d.NB * d.PH * d.PW;
    const int tpt = (d.x.C0 + 15) / 16 + d.x.C1 / 16;
    const Magics mg = make_magics((unsigned)tpt, (unsigned)d.TW, (unsigned long)d.PW, (unsigned long)d.PH * d.PW, (unsigned long)M);
    const long mt = (M + BM - 1) / BM;
    const int nt = (d.Nstore + BN - 1) / BN;
    const long out_count = (long)d.NB * d.OH * d.OW * d.ldc;
    static unsigned long long attr_done = 0;
    {
        const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&conv_bfh_kernel<WM, WN, PLAIN, ONE, KM>), (int)lds, &attr_done);
        if (arc != 0) return arc;
    }
    BfTabs tab = {nullptr, nullptr, nullptr, nullptr};
    if (!PLAIN && !bf_tabs(d, tab)) return -5;
    const int xflag = 0x10000 | ((xcd >= 2 && d.nphase == 4) ? 0x20000 : 0);
    const bool par_ok = d.in_stride == 2 && (d.TH & 1) == 0 && (d.TW & 1) == 0;
    int korder = par_ok ? 2 : 1;
    if (d.TH * d.TW == 1) korder = 0;
    if (splitk == 1 && ws != nullptr && d.sk_flags != nullptr && ts_s > 1) {
        const long tiles = mt * nt * d.nphase;
        const long full = ts_full, tail = tiles - full, s = ts_s;
        if (full >= 0 && tail > 0 && (int64_t)tail * s * BM * BN * 4 <= ws_bytes && tail * s < SSC_SK_FLAG_WORDS - 1 &&
            full + tail * s < 0x7fffffffL) {
            hipLaunchKernelGGL((conv_bfh_kernel<WM, WN, PLAIN, ONE, KM>), dim3((unsigned)(full + tail * s)), dim3(256), lds, st, d, mg, ws, out_count,
                               1, (int)full, (int)s | ((xcd && (full & 7) == 0) ? xflag : 0), d.sk_flags, tab, korder);
            return (int)hipGetLastError();
        }
    }
    if (splitk == 1 && xcd) {
        const long tiles = mt * nt * d.nphase;
        const long full = tiles & ~7L;
        if (tiles < 0x7fffffffL && full > 0) {
            hipLaunchKernelGGL((conv_bfh_kernel<WM, WN, PLAIN, ONE, KM>), dim3((unsigned)tiles), dim3(256), lds, st, d, mg, ws, out_count, 1,
                               (int)full, 1 | xflag, (unsigned*)nullptr, tab, korder);
            return (int)hipGetLastError();
        }
    }
    dim3 grid((unsigned)mt, (unsigned)nt, (unsigned)(d.nphase * splitk));
    hipLaunchKernelGGL((conv_bfh_kernel<WM, WN, PLAIN, ONE, KM>), grid, dim3(256), lds, st, d, mg, ws, out_count, splitk, 0, 0, (unsigned*)nullptr,
                       tab, korder);
    if (splitk > 1) ssc_launch_slab_reduce(ws, out_count, splitk, d, st);
    return (int)hipGetLastError();
}

template <int WM, int WN, int SM, int SN, bool SS>
static int launch_bf_form(bool plain, bool one, bool km, const ssc_conv_desc& d, int splitk, float* ws, hipStream_t st, long ts_full,
                          int ts_s, int64_t ws_bytes, int xcd) {
    if (km) return plain ? launch_bf_t<WM, WN, SM, SN, true, true, true, SS>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd)
                         : launch_bf_t<WM, WN, SM, SN, false, true, true, SS>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd);
    return one ? (plain ? launch_bf_t<WM, WN, SM, SN, true, true, false, SS>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd)
                        : launch_bf_t<WM, WN, SM, SN, false, true, false, SS>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd))
               : (plain ? launch_bf_t<WM, WN, SM, SN, true, false, false, SS>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd)
                        : launch_bf_t<WM, WN, SM, SN, false, false, false, SS>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd));
}

// the 128 x 128 tile runs on 16-k stages (conv_bfh_kernel) unless SSC_BF_HK=0 under SSC_DEV_SWITCHES (A/B: round 5's 32-k form);
// the planner prices the tile accordingly (igemm.hip: plan_fwd)
bool ssc_bf_hk_enabled() {
    static int hk_env = -2;
    if (hk_env == -2) {
        const char* e = ssc_dev_getenv("SSC_BF_HK");
        hk_env = e != nullptr ? (e[0] == '1' ? 1 : 0) : -1;
    }
    return hk_env != 0;
}

// cfg: 0 = 128x128, 1 = 64x128, 2 = 128x64, 4 = 64x64 (the ids of igemm.hip's tile table)
int ssc_launch_conv_bf(int cfg, bool plain, const ssc_conv_desc& d, int splitk, float* ws, hipStream_t st, long ts_full, int ts_s,
                       int64_t ws_bytes, int xcd) {
    const bool one = d.x.C1 == 0;
    const bool km = one && (d.x.C0 % BK) != 0;
    // one LDS stage per operand (64x128 / 128x64 / 64x64 tiles) when the caller says the launch shares the chip with other streams'
    // launches (ssc_conv_desc.lds_hint); SSC_BF_SS=0 / 1 under SSC_DEV_SWITCHES pins it (A/B)
    static int ss_env = -2;
    if (ss_env == -2) {
        const char* e = ssc_dev_getenv("SSC_BF_SS");
        ss_env = e != nullptr ? (e[0] == '1' ? 1 : 0) : -1;
    }
    const bool ss = ss_env >= 0 ? ss_env == 1 : (d.lds_hint & 1) != 0;
#define BF_CASE(WM, WN, SM, SN)                                                                                              \
    return ss ? launch_bf_form<WM, WN, SM, SN, true>(plain, one, km, d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd)         \
              : launch_bf_form<WM, WN, SM, SN, false>(plain, one, km, d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd)
    // 128 x 128 tiles: on 16-k stages, two workgroups per CU (conv_bfh_kernel; SSC_BF_HK=0 under SSC_DEV_SWITCHES: the 32-k form)
#define BFH_CASE(WMV, WNV)                                                                                        \
    if (km) return plain ? launch_bfh_t<WMV, WNV, true, true, true>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd)     \
                         : launch_bfh_t<WMV, WNV, false, true, true>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd);   \
    return one ? (plain ? launch_bfh_t<WMV, WNV, true, true>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd)        \
                        : launch_bfh_t<WMV, WNV, false, true>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd))      \
               : (plain ? launch_bfh_t<WMV, WNV, true, false>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd)       \
                        : launch_bfh_t<WMV, WNV, false, false>(d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd))
    if (cfg == 0 && ssc_bf_hk_enabled()) {
        BFH_CASE(2, 2);
    }
#undef BFH_CASE
    switch (cfg) {
        case 0: return launch_bf_form<2, 2, 2, 2, false>(plain, one, km, d, splitk, ws, st, ts_full, ts_s, ws_bytes, xcd);
        case 1: BF_CASE(2, 2, 1, 2);
        case 2: BF_CASE(2, 2, 2, 1);
        case 4: BF_CASE(2, 2, 1, 1);
        default: return -4;
    }
#undef BF_CASE
}

extern "C" int ssc_filter_split(const float* w, int taps, int c0, int c1, int orient, void* dst, void* stream) {
    ssc_split_job jb;
    jb.w = w; jb.dst = dst; jb.taps = taps; jb.c0 = c0; jb.c1 = c1; jb.orient = orient; jb.first_thread = 0;
    const int64_t threads = ssc_filter_split_threads(taps, c0, c1, orient);
    if (threads / 256 > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(filter_split_one_kernel, dim3((unsigned)(threads / 256)), dim3(256), 0, (hipStream_t)stream, jb);
    return (int)hipGetLastError();
}
#endif
