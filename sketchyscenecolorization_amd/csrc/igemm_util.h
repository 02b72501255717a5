// igemm_util.h -- device / host helpers shared by the implicit-GEMM kernels (igemm.hip, wgrad128.hip): magic-number division,
// gather-view access, the folded-norm transform on packed math, LDS-DMA instructions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sketchycolor_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Exact division by a launch-constant through multiply-high (the tile loaders run every K-tile:
// no integer-division expansions, no branches).
struct Magics {
    unsigned mC, oneC;          // kcol / C   (kcol < 2^20, C < 2^12); one* = ~0 when the divisor is 1
    unsigned mTW, oneTW;        // tap / TW
    unsigned long mPW, onePW;   // rem / PW   (64-bit magic: exact for every 32-bit numerator)
    unsigned long mPHPW, onePHPW; // m / (PH*PW)
    unsigned mPW32, mPHPW32;    // 32-bit forms, valid when use32 (numerator * divisor < 2^32)
    int use32, _pad;
};
static inline unsigned magic32(unsigned d) { return d == 1 ? 0u : (unsigned)(0x100000000ULL / d) + 1u; }
static inline unsigned long magic64(unsigned long d) {
    return d == 1 ? 0UL : (unsigned long)((((unsigned __int128)1) << 64) / d) + 1UL;
}
static Magics make_magics(unsigned C, unsigned TW, unsigned long PW, unsigned long PHW, unsigned long maxnum) {
    Magics m;
    m.mC = magic32(C); m.oneC = C == 1 ? ~0u : 0u;
    m.mTW = magic32(TW); m.oneTW = TW == 1 ? ~0u : 0u;
    m.mPW = magic64(PW); m.onePW = PW == 1 ? ~0UL : 0UL;
    m.mPHPW = magic64(PHW); m.onePHPW = PHW == 1 ? ~0UL : 0UL;
    m.mPW32 = magic32((unsigned)PW); m.mPHPW32 = magic32((unsigned)PHW);
    m.use32 = (maxnum * PHW < 0xFFFFFFFFUL && PHW < 0xFFFFFFFFUL) ? 1 : 0;
    m._pad = 0;
    return m;
}
__device__ __forceinline__ int div32(int n, unsigned magic, unsigned one) {
    return (int)(__umulhi((unsigned)n, magic) + ((unsigned)n & one));
}
__device__ __forceinline__ long div64(long n, unsigned long magic, unsigned long one) {
    return (long)(__umul64hi((unsigned long)n, magic) + ((unsigned long)n & one));
}

// per-thread affine (a, b) for channels [c, c+4) of a gather view; identity when the source has none.
// Branch-free: a null table is replaced by a valid dummy address and the result by (1, 0).
__device__ __forceinline__ void gview_affine4(const ssc_gview& g, int c, float4& a, float4& b) {
    const bool first = c < g.C0;
    const float* ab = first ? g.ab0 : g.ab1;
    const bool has = ab != nullptr;
    const int cc = first ? c : c - g.C0;
    const int Cs = first ? g.C0 : g.C1;
    const float* pa = has ? ab + cc : g.s0;
    const float* pb = has ? ab + Cs + cc : g.s0;
    const float4 va = *reinterpret_cast<const float4*>(pa);
    const float4 vb = *reinterpret_cast<const float4*>(pb);
    a = has ? va : make_float4(1.f, 1.f, 1.f, 1.f);
    b = has ? vb : make_float4(0.f, 0.f, 0.f, 0.f);
}

// base pointer + row stride of the source holding channel c (c, C0 multiples of 4)
__device__ __forceinline__ void gview_src(const ssc_gview& g, int c, const float*& base, int& cs) {
    const bool first = c < g.C0;
    base = first ? g.s0 + c : g.s1 + (c - g.C0);
    cs = first ? g.C0 : g.C1;
}

// act(a*v+b)*m with act(t) = max(t, slope*t): slope 1 = identity, 0 = relu, 0.2 = lrelu; m in {0,1} zeroes the
// out-of-image / out-of-range elements.  Straight-line code: fp32 MFMA shares the vector lanes with VALU on
// gfx950 (SQ_VALU_MFMA_COEXEC_CYCLES = 0), so every VALU instruction here is paid for in matrix throughput.
__device__ __forceinline__ float act_slope(int act) {
    return act == SSC_ACT_RELU ? 0.f : (act == SSC_ACT_LRELU ? 0.2f : 1.f);
}
// NO packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in kernels that issue fp32 MFMAs: measured on MI355X
// (scripts/mfma_mix_probe.py, round 5), a wave that mixes v_mfma_f32_32x32x2_f32 with packed-fp32 vector instructions computes
// WRONG values (errors of whole product terms) while another wave on the same SIMD issues v_mfma_f32_32x32x16_bf16 -- i.e. as
// soon as a bf16-split conv launch of another stream shares a CU with the filter-gradient kernel.  The transform is therefore
// written on scalars, and the library is compiled with -fno-slp-vectorize (build.py) so that the compiler does not pack it again.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 xform4(float4 v, const float4& a, const float4& b, float slope, float m) {
    float t;
    t = fmaf(a.x, v.x, b.x); v.x = fmaxf(t, t * slope) * m;
    t = fmaf(a.y, v.y, b.y); v.y = fmaxf(t, t * slope) * m;
    t = fmaf(a.z, v.z, b.z); v.z = fmaxf(t, t * slope) * m;
    t = fmaf(a.w, v.w, b.w); v.w = fmaxf(t, t * slope) * m;
    return v;
}
__device__ __forceinline__ float4 mask4(float4 v, float m) {
    return make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
}

// One LDS-DMA instruction: 16 bytes per lane from sbase (wave-uniform) + voff straight into LDS at lds_addr + 16 * lane.
// Inline asm, not __builtin_amdgcn_global_load_lds: behind the builtin hipcc waits vmcnt(0) before the next ds_read (it
// assumes the DMA's LDS write may alias it), which would expose the DMA's whole latency in every K step.  M0 (the LDS
// destination) is saved and restored inside the statement; the completion is counted by hand (s_waitcnt vmcnt(N) before the
// barrier that publishes the tile).
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

// The same through a buffer descriptor: sbase/range from rsrc, voff per lane (an offset >= the descriptor's size makes the
// lane's 16 bytes in LDS zeros -- measured, scripts/glds_oob_test.hip), soff wave-uniform.
__device__ __forceinline__ void glds16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
                 : "memory");
}

