// head1.hip -- the PatchGAN logit conv (512 -> 1, 4x4, stride 1, pad 1; models_collection.py:833-835 discriminate_pix2pix
// layer_5) in its three forms.  With one output channel the GEMM forms degenerate (N = 1 forward, K = 16 for the data gradient,
// a matrix-vector product for the filter gradient): 0.25 GFLOP against 35 MB of activations, i.e. HBM-bound streaming work.
// All three kernels walk the PIXELS of the 512-channel tensor once, a wavefront per pixel, lane l holding channels
// [4l, 4l+4) and [256+4l, 256+4l+4) (two coalesced 1 KB accesses per pixel) and the 16 x 8 filter values of its channels in
// registers:
//   forward        t[q][tap] = sum_c act(a*x[q][c]+b) * w[tap][c]   (cross-lane butterfly), then out[p] = sum_tap t[p@tap][tap]
//   data gradient  g[q][c]   = sum_tap dy[q@tap] * w[tap][c]
//   filter grad    dw[tap][c] = sum_q act(x[q][c]) * dy[q@tap]      (register accumulators, block slabs, ordered reduce)
// ssc_conv_forward / ssc_conv_wgrad dispatch here (ssc_head1_*_supported); everything else keeps the general kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

#define CHECK_LAUNCH() ((int)hipGetLastError())

// elementwise.hip: rows of [sum dz | sum dz*xhat] -> coef = their means, scale / offset gradients
extern "C" int ssc_bn_bwd_finalize(const float* partial, int nblk, int C, int64_t M, float* coef, float* dscale, float* doffset,
                                   void* stream);
// igemm.hip: dw (+)= the sum of `splitk` slabs of `count` floats, in a fixed order
void ssc_launch_wgrad_reduce(const float* ws, long count, int splitk, float* out, int accumulate, hipStream_t st);

namespace {

constexpr int HC = 512;         // channels of the wide tensor
constexpr int CPL = 8;          // channels per lane
constexpr int MAXT = 16;        // taps

struct H1Geo {
    int NB, H, W;               // the wide tensor's pixel grid
    int OH, OW;                 // the one-channel tensor's pixel grid (and its row stride in floats: ld1)
    int ld1;
    int TH, TW;
    int off_y, off_x;           // wide pixel (iy, ix) meets narrow pixel (iy - off_y - ty, ix - off_x - tx) through tap (ty, tx)
    unsigned mHW, mW;           // q / (H*W), r / W by multiply-high (exact: q * H*W < 2^32)
    float slope;                // act(t) = max(t, slope * t): 1 none, 0 relu, 0.2 lrelu
    int has_ab;
};

__device__ __forceinline__ void h1_decode(const H1Geo& g, int q, int& n, int& iy, int& ix) {
    n = (int)__umulhi((unsigned)q, g.mHW);
    const int r = q - n * g.H * g.W;
    iy = (int)__umulhi((unsigned)r, g.mW);
    ix = r - iy * g.W;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
// lane t's value to every lane through an SGPR (v_readlane_b32; t is a compile-time constant after unrolling)
__device__ __forceinline__ float h1_bcast(float v, int t) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t));
}

// this lane's 8 raw channels of pixel q, and act(a*x+b) of them (split so that a pixel's loads fly during the FMAs of the
// previous one)
__device__ __forceinline__ void h1_load(const float* __restrict__ x, long q, int lane, float4& u0, float4& u1) {
    const float* p = x + q * HC + 4 * lane;
    u0 = ld4(p);
    u1 = ld4(p + 256);
}
__device__ __forceinline__ void h1_act(const float4& u0, const float4& u1, const float4& a0, const float4& a1, const float4& b0,
                                       const float4& b1, float slope, float (&v)[CPL]) {
    const float t[CPL] = {fmaf(a0.x, u0.x, b0.x), fmaf(a0.y, u0.y, b0.y), fmaf(a0.z, u0.z, b0.z), fmaf(a0.w, u0.w, b0.w),
                          fmaf(a1.x, u1.x, b1.x), fmaf(a1.y, u1.y, b1.y), fmaf(a1.z, u1.z, b1.z), fmaf(a1.w, u1.w, b1.w)};
#pragma unroll
    for (int j = 0; j < CPL; ++j) v[j] = fmaxf(t[j], slope * t[j]);
}

__device__ __forceinline__ void h1_load_ab(const float* ab, int has_ab, int lane, float4& a0, float4& a1, float4& b0, float4& b1) {
    a0 = a1 = make_float4(1.f, 1.f, 1.f, 1.f);
    b0 = b1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_ab) {
        a0 = ld4(ab + 4 * lane); a1 = ld4(ab + 256 + 4 * lane);
        b0 = ld4(ab + HC + 4 * lane); b1 = ld4(ab + HC + 256 + 4 * lane);
    }
}

// this lane's filter values: wr[tap][j], w laid out [tap][HC] (one output channel: KN and NK orientations coincide)
__device__ __forceinline__ void h1_load_w(const float* __restrict__ w, int ntap, int lane, float (&wr)[MAXT][CPL]) {
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
        if (t < ntap) {
            u0 = ld4(w + (long)t * HC + 4 * lane);
            u1 = ld4(w + (long)t * HC + 256 + 4 * lane);
        }
        wr[t][0] = u0.x; wr[t][1] = u0.y; wr[t][2] = u0.z; wr[t][3] = u0.w;
        wr[t][4] = u1.x; wr[t][5] = u1.y; wr[t][6] = u1.z; wr[t][7] = u1.w;
    }
}

// the narrow tensor's value met by wide pixel (n, iy, ix) through lattice tap (ty, tx) = this lane's (lanes >= ntap and
// out-of-range taps: 0);
// filter tap index of lattice tap (ty, tx) = (ky0 + kstep*ty) * TW + kx0 + kstep*tx is applied by the callers
__device__ __forceinline__ float h1_narrow_at(const float* __restrict__ y, const H1Geo& g, int n, int iy, int ix, int ty, int tx) {
    const int oy = iy - g.off_y - ty, ox = ix - g.off_x - tx;      // lanes >= TH*TW come with ty >= TH
    const bool ok = ty < g.TH && (unsigned)oy < (unsigned)g.OH && (unsigned)ox < (unsigned)g.OW;
    return ok ? y[(((long)n * g.OH + oy) * g.OW + ox) * g.ld1] : 0.f;
}

// ---------------------------------------------------------------------------------------------------- forward, stage 1
// t[q][16]: lane's 16 partial dots, then a butterfly that halves the values per lane at every exchange (8+4+2+1 exchanges,
// then two plain steps): lane l ends with the total of tap ((l>>5)&1)*8 + ((l>>4)&1)*4 + ((l>>3)&1)*2 + ((l>>2)&1).
__global__ __launch_bounds__(256) void head1_fwd_dots_kernel(const float* __restrict__ x, const float* __restrict__ ab,
                                                             const float* __restrict__ w, H1Geo g, int ntap, int npix,
                                                             float* __restrict__ t) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
    float wr[MAXT][CPL];
    h1_load_w(w, ntap, lane, wr);
    float4 a0, a1, b0, b1;
    h1_load_ab(ab, g.has_ab, lane, a0, a1, b0, b1);
    float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
    if (wave < npix) h1_load(x, wave, lane, u0, u1);
    for (int q = wave; q < npix; q += nwave) {
        float4 n0 = u0, n1 = u1;
        if (q + nwave < npix) h1_load(x, q + nwave, lane, n0, n1);
        float v[CPL];
        h1_act(u0, u1, a0, a1, b0, b1, g.slope, v);
        u0 = n0;
        u1 = n1;
        float d[MAXT];
#pragma unroll
        for (int k = 0; k < MAXT; ++k) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < CPL; ++j) s = fmaf(v[j], wr[k][j], s);
            d[k] = s;
        }
        // 16 -> 8 -> 4 -> 2 -> 1 values per lane
        float e8[8], e4[4], e2[2];
        const bool h32 = lane & 32, h16 = lane & 16, h8 = lane & 8, h4 = lane & 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) e8[i] = (h32 ? d[i + 8] : d[i]) + __shfl_xor(h32 ? d[i] : d[i + 8], 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) e4[i] = (h16 ? e8[i + 4] : e8[i]) + __shfl_xor(h16 ? e8[i] : e8[i + 4], 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) e2[i] = (h8 ? e4[i + 2] : e4[i]) + __shfl_xor(h8 ? e4[i] : e4[i + 2], 8);
        float r = (h4 ? e2[1] : e2[0]) + __shfl_xor(h4 ? e2[0] : e2[1], 4);
        r += __shfl_xor(r, 2);
        r += __shfl_xor(r, 1);
        if ((lane & 3) == 0) {
            const int tap = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
            t[(long)q * MAXT + tap] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- forward, stage 2
// out[n, oy, ox, 0] = sum over taps in filter order of t[(n, oy*1 + ioff + ty, ...)][tap]; columns 1..nstore-1 = 0
__global__ void head1_fwd_gather_kernel(const float* __restrict__ t, H1Geo g, int ioff_y, int ioff_x, int nstore,
                                        float* __restrict__ out, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ox = i % g.OW, r = i / g.OW, oy = r % g.OH, n = r / g.OH;
    float s = 0.f;
    for (int ty = 0; ty < g.TH; ++ty) {
        const int iy = oy + ioff_y + ty;
        if ((unsigned)iy >= (unsigned)g.H) continue;
        for (int tx = 0; tx < g.TW; ++tx) {
            const int ix = ox + ioff_x + tx;
            if ((unsigned)ix >= (unsigned)g.W) continue;
            s += t[(((long)n * g.H + iy) * g.W + ix) * MAXT + ty * g.TW + tx];
        }
    }
    float* o = out + (long)i * g.ld1;
    o[0] = s;
    for (int k = 1; k < nstore; ++k) o[k] = 0.f;
}

// ---------------------------------------------------------------------------------------------------- data gradient
// g[q][c] = sum_tap dy[q@tap] * w[ftap][c], ftap = the filter tap of lattice tap (ty, tx): (ky0 + kstep*ty)*TW + kx0 + kstep*tx
__global__ __launch_bounds__(256) void head1_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, H1Geo g,
                                                          int ky0, int kx0, int kstep, int npix, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
    const int ntap = g.TH * g.TW;
    // registers hold the filter in LATTICE tap order
    float wr[MAXT][CPL];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int ty = t / g.TW, tx = t - ty * g.TW;
        const int ft = (ky0 + kstep * ty) * g.TW + kx0 + kstep * tx;
        float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
        if (t < ntap) {
            u0 = ld4(w + (long)ft * HC + 4 * lane);
            u1 = ld4(w + (long)ft * HC + 256 + 4 * lane);
        }
        wr[t][0] = u0.x; wr[t][1] = u0.y; wr[t][2] = u0.z; wr[t][3] = u0.w;
        wr[t][4] = u1.x; wr[t][5] = u1.y; wr[t][6] = u1.z; wr[t][7] = u1.w;
    }
    const int lty = lane / g.TW, ltx = lane - lty * g.TW;
    for (int q = wave; q < npix; q += nwave) {
        int n, iy, ix;
        h1_decode(g, q, n, iy, ix);
        const float mine = h1_narrow_at(dy, g, n, iy, ix, lty, ltx);
        float acc[CPL] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const float s = h1_bcast(mine, t);
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[j] = fmaf(s, wr[t][j], acc[j]);
        }
        float* o = out + (long)q * HC + 4 * lane;
        *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(o + 256) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

// ---------------------------------------------------------------------------------------------------- filter gradient
// slab[block][tap < TH*TW][c] = sum over the block's pixels; waves 1..3 hand their registers to wave 0 through LDS in wave order
__global__ __launch_bounds__(256) void head1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ ab,
                                                          const float* __restrict__ dy, H1Geo g, int npix,
                                                          float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float sh[];      // [3][MAXT * HC]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wv, nwave = gridDim.x * 4;
    float4 a0, a1, b0, b1;
    h1_load_ab(ab, g.has_ab, lane, a0, a1, b0, b1);
    float acc[MAXT][CPL];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[t][j] = 0.f;
    const int lty = lane / g.TW, ltx = lane - lty * g.TW;
    float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
    float mine = 0.f;
    if (wave < npix) {
        int n, iy, ix;
        h1_decode(g, wave, n, iy, ix);
        mine = h1_narrow_at(dy, g, n, iy, ix, lty, ltx);
        h1_load(x, wave, lane, u0, u1);
    }
    for (int q = wave; q < npix; q += nwave) {
        float4 n0 = u0, n1 = u1;
        float nmine = 0.f;
        if (q + nwave < npix) {
            int n, iy, ix;
            h1_decode(g, q + nwave, n, iy, ix);
            nmine = h1_narrow_at(dy, g, n, iy, ix, lty, ltx);
            h1_load(x, q + nwave, lane, n0, n1);
        }
        float v[CPL];
        h1_act(u0, u1, a0, a1, b0, b1, g.slope, v);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const float s = h1_bcast(mine, t);
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[t][j] = fmaf(v[j], s, acc[t][j]);
        }
        u0 = n0;
        u1 = n1;
        mine = nmine;
    }
    if (wv > 0) {
        float* p = sh + (long)(wv - 1) * MAXT * HC;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            *reinterpret_cast<float4*>(p + t * HC + 4 * lane) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
            *reinterpret_cast<float4*>(p + t * HC + 256 + 4 * lane) = make_float4(acc[t][4], acc[t][5], acc[t][6], acc[t][7]);
        }
    }
    __syncthreads();
    if (wv == 0) {
        const int ntap = g.TH * g.TW;
        float* o = slabs + (long)blockIdx.x * ntap * HC;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            float4 s0 = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
            float4 s1 = make_float4(acc[t][4], acc[t][5], acc[t][6], acc[t][7]);
            for (int k = 0; k < 3; ++k) {
                const float4 u0 = *reinterpret_cast<const float4*>(sh + (long)k * MAXT * HC + t * HC + 4 * lane);
                const float4 u1 = *reinterpret_cast<const float4*>(sh + (long)k * MAXT * HC + t * HC + 256 + 4 * lane);
                s0.x += u0.x; s0.y += u0.y; s0.z += u0.z; s0.w += u0.w;
                s1.x += u1.x; s1.y += u1.y; s1.z += u1.z; s1.w += u1.w;
            }
            if (t < ntap) {
                *reinterpret_cast<float4*>(o + t * HC + 4 * lane) = s0;
                *reinterpret_cast<float4*>(o + t * HC + 256 + 4 * lane) = s1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- data gradient fused
// with the backward of the norm + activation that follows the wide tensor x (layer_4's batch norm + lrelu): the gradient
// g = sum_tap dy*w (+ rowb[n]*rowb_scale, the class head's term) is recomputed in both passes and never stored.
//   pass 1: rows of [sum dz | sum dz*xhat] per workgroup, dz = g * act'(a*x+b), xhat = (x - mean) * rstd
//   pass 2 (after ssc_bn_bwd_finalize): dx = a * (dz - c1 - xhat*c2)          (bn_bwd_apply_kernel's formula)
struct H1Bn {
    const float* x;         // [npix][HC]
    const float* ab;        // [2][HC]
    const float* stats;     // [2][HC]: mean, rstd
    const float* rowb;      // [NB][HC] or NULL
    float rowb_scale;
    int act;
};
__device__ __forceinline__ float h1_dact(float z, int act) {
    if (act == SSC_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == SSC_ACT_LRELU) return z > 0.f ? 1.f : 0.2f;
    return 1.f;
}
__device__ __forceinline__ void h1_load_wt(const float* __restrict__ w, const H1Geo& g, int ky0, int kx0, int kstep, int lane,
                                           float (&wr)[MAXT][CPL]) {
    const int ntap = g.TH * g.TW;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int ty = t / g.TW, tx = t - ty * g.TW;
        const int ft = (ky0 + kstep * ty) * g.TW + kx0 + kstep * tx;
        float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
        if (t < ntap) {
            u0 = ld4(w + (long)ft * HC + 4 * lane);
            u1 = ld4(w + (long)ft * HC + 256 + 4 * lane);
        }
        wr[t][0] = u0.x; wr[t][1] = u0.y; wr[t][2] = u0.z; wr[t][3] = u0.w;
        wr[t][4] = u1.x; wr[t][5] = u1.y; wr[t][6] = u1.z; wr[t][7] = u1.w;
    }
}
// dz and x of this lane's 8 channels of pixel q (sample n): mine = this lane's tap value of dy
__device__ __forceinline__ void h1_dz(const H1Bn& b, int q, int n, int lane, float mine, const float (&wr)[MAXT][CPL],
                                      const float (&a)[CPL], const float (&bb)[CPL], const float4& x0, const float4& x1,
                                      float (&xv)[CPL], float (&dz)[CPL]) {
    float gsum[CPL] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const float s = h1_bcast(mine, t);
#pragma unroll
        for (int j = 0; j < CPL; ++j) gsum[j] = fmaf(s, wr[t][j], gsum[j]);
    }
    if (b.rowb != nullptr) {
        const float4 r0 = ld4(b.rowb + (long)n * HC + 4 * lane), r1 = ld4(b.rowb + (long)n * HC + 256 + 4 * lane);
        const float rr[CPL] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int j = 0; j < CPL; ++j) gsum[j] = fmaf(rr[j], b.rowb_scale, gsum[j]);
    }
    xv[0] = x0.x; xv[1] = x0.y; xv[2] = x0.z; xv[3] = x0.w; xv[4] = x1.x; xv[5] = x1.y; xv[6] = x1.z; xv[7] = x1.w;
#pragma unroll
    for (int j = 0; j < CPL; ++j) dz[j] = gsum[j] * h1_dact(fmaf(a[j], xv[j], bb[j]), b.act);
}
__device__ __forceinline__ void h1_lane8(const float* p, int lane, float (&v)[CPL]) {
    const float4 u0 = ld4(p + 4 * lane), u1 = ld4(p + 256 + 4 * lane);
    v[0] = u0.x; v[1] = u0.y; v[2] = u0.z; v[3] = u0.w; v[4] = u1.x; v[5] = u1.y; v[6] = u1.z; v[7] = u1.w;
}

__global__ __launch_bounds__(256) void head1_bnbwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                  H1Geo g, int ky0, int kx0, int kstep, int npix, H1Bn b,
                                                                  float* __restrict__ partial) {
    __shared__ float sh[3][2][HC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wv, nwave = gridDim.x * 4;
    float wr[MAXT][CPL];
    h1_load_wt(w, g, ky0, kx0, kstep, lane, wr);
    float a[CPL], bb[CPL], mu[CPL], rs[CPL];
    h1_lane8(b.ab, lane, a); h1_lane8(b.ab + HC, lane, bb); h1_lane8(b.stats, lane, mu); h1_lane8(b.stats + HC, lane, rs);
    const int lty = lane / g.TW, ltx = lane - lty * g.TW;
    float s1[CPL] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2[CPL] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
    if (wave < npix) h1_load(b.x, wave, lane, u0, u1);
    for (int q = wave; q < npix; q += nwave) {
        float4 n0 = u0, n1 = u1;
        if (q + nwave < npix) h1_load(b.x, q + nwave, lane, n0, n1);
        int n, iy, ix;
        h1_decode(g, q, n, iy, ix);
        const float mine = h1_narrow_at(dy, g, n, iy, ix, lty, ltx);
        float xv[CPL], dz[CPL];
        h1_dz(b, q, n, lane, mine, wr, a, bb, u0, u1, xv, dz);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            s1[j] += dz[j];
            s2[j] += dz[j] * (xv[j] - mu[j]) * rs[j];
        }
        u0 = n0;
        u1 = n1;
    }
    if (wv > 0) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = (j < 4 ? 4 * lane + j : 256 + 4 * lane + j - 4);
            sh[wv - 1][0][c] = s1[j];
            sh[wv - 1][1][c] = s2[j];
        }
    }
    __syncthreads();
    if (wv == 0) {
        float* o = partial + (long)blockIdx.x * 2 * HC;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = (j < 4 ? 4 * lane + j : 256 + 4 * lane + j - 4);
            o[c] = ((s1[j] + sh[0][0][c]) + sh[1][0][c]) + sh[2][0][c];
            o[HC + c] = ((s2[j] + sh[0][1][c]) + sh[1][1][c]) + sh[2][1][c];
        }
    }
}

__global__ __launch_bounds__(256) void head1_bnbwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                H1Geo g, int ky0, int kx0, int kstep, int npix, H1Bn b,
                                                                const float* __restrict__ coef, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
    float wr[MAXT][CPL];
    h1_load_wt(w, g, ky0, kx0, kstep, lane, wr);
    float a[CPL], bb[CPL], mu[CPL], rs[CPL], c1[CPL], c2[CPL];
    h1_lane8(b.ab, lane, a); h1_lane8(b.ab + HC, lane, bb); h1_lane8(b.stats, lane, mu); h1_lane8(b.stats + HC, lane, rs);
    h1_lane8(coef, lane, c1); h1_lane8(coef + HC, lane, c2);
    const int lty = lane / g.TW, ltx = lane - lty * g.TW;
    float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
    if (wave < npix) h1_load(b.x, wave, lane, u0, u1);
    for (int q = wave; q < npix; q += nwave) {
        float4 n0 = u0, n1 = u1;
        if (q + nwave < npix) h1_load(b.x, q + nwave, lane, n0, n1);
        int n, iy, ix;
        h1_decode(g, q, n, iy, ix);
        const float mine = h1_narrow_at(dy, g, n, iy, ix, lty, ltx);
        float xv[CPL], dz[CPL], o[CPL];
        h1_dz(b, q, n, lane, mine, wr, a, bb, u0, u1, xv, dz);
#pragma unroll
        for (int j = 0; j < CPL; ++j) o[j] = a[j] * (dz[j] - c1[j] - (xv[j] - mu[j]) * rs[j] * c2[j]);
        float* p = dx + (long)q * HC + 4 * lane;
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(p + 256) = make_float4(o[4], o[5], o[6], o[7]);
        u0 = n0;
        u1 = n1;
    }
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

bool view_ok(const ssc_gview& v) {      // the wide tensor: one 512-channel source, relu-family activation
    return v.C0 == HC && v.C1 == 0 && v.s1 == nullptr && al16(v.s0) && (v.ab0 == nullptr || al16(v.ab0)) &&
           (v.act == SSC_ACT_NONE || v.act == SSC_ACT_RELU || v.act == SSC_ACT_LRELU);
}
bool narrow_ok(const ssc_gview& v) {    // the one-channel tensor: plain, channel 0 of a 4-float row
    return v.C1 == 0 && v.s1 == nullptr && v.ab0 == nullptr && v.act == SSC_ACT_NONE && v.C0 >= 1;
}
float slope_of(int act) { return act == SSC_ACT_RELU ? 0.f : (act == SSC_ACT_LRELU ? 0.2f : 1.f); }

H1Geo make_geo(int NB, int H, int W, int OH, int OW, int ld1, int TH, int TW, int off_y, int off_x, const ssc_gview* wide) {
    H1Geo g;
    g.NB = NB; g.H = H; g.W = W; g.OH = OH; g.OW = OW; g.ld1 = ld1; g.TH = TH; g.TW = TW; g.off_y = off_y; g.off_x = off_x;
    const unsigned hw = (unsigned)(H * W);
    g.mHW = hw <= 1 ? 0u : (unsigned)(0x100000000ULL / hw) + 1u;
    g.mW = W <= 1 ? 0u : (unsigned)(0x100000000ULL / (unsigned)W) + 1u;
    g.slope = wide ? slope_of(wide->act) : 1.f;
    g.has_ab = (wide && wide->ab0 != nullptr) ? 1 : 0;
    return g;
}
bool geo_ok(int NB, int H, int W) {     // multiply-high decode exact: q * (H*W) < 2^32, H*W > 1, W > 1
    return H > 1 && W > 1 && (unsigned long)NB * H * W * (unsigned long)(H * W) < 0xffffffffUL;
}
int env_off() {
    static int off = -1;
    if (off < 0) {
        const char* e = ssc_dev_getenv("SSC_HEAD1");
        off = (e != nullptr && e[0] == '0') ? 1 : 0;
    }
    return off;
}
unsigned wave_blocks(int npix) {        // 4 waves per block, a few pixels per wave
    int b = (npix + 15) / 16;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

// forward form: out[p][0] = sum_{tap, c} act(x[p@tap][c]) * w[tap][c][0]
extern "C" int ssc_head1_forward_supported(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    if (env_off()) return 0;
    return d.nphase == 1 && d.bmode == 0 && d.in_stride == 1 && d.kstep == 1 && d.ky0 == 0 && d.kx0 == 0 && d.out_stride == 1 &&
           d.ooff_y == 0 && d.ooff_x == 0 && d.Nn == 1 && d.n_off == 0 && d.Nstore >= 1 && d.Nstore <= d.ldc &&
           d.TH * d.TW <= MAXT && d.KH == d.TH && d.KW == d.TW && view_ok(d.x) && d.k_real == HC && d.wC0 == HC && d.wC1 == 1 &&
           d.bias == nullptr && d.epi == 0 && !d.accumulate && d.stat_partial == nullptr && d.sb_x == nullptr && al16(d.w) &&
           d.OH == d.PH && d.OW == d.PW && geo_ok(d.NB, d.x.H, d.x.W);
}

extern "C" int ssc_head1_forward(const ssc_conv_desc* dp, float* ws, int64_t ws_bytes, void* stream) {
    if (!ssc_head1_forward_supported(dp)) return -1;
    const ssc_conv_desc& d = *dp;
    const int npix = d.NB * d.x.H * d.x.W;
    if (ws == nullptr || (int64_t)npix * MAXT * 4 > ws_bytes) return -2;
    // wide pixel iy = oy + ioff_y + ty  <=>  oy = iy - ioff_y - ty
    const H1Geo g = make_geo(d.NB, d.x.H, d.x.W, d.OH, d.OW, d.ldc, d.TH, d.TW, d.ioff_y, d.ioff_x, &d.x);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(head1_fwd_dots_kernel, dim3(wave_blocks(npix)), dim3(256), 0, st, d.x.s0, d.x.ab0, d.w, g, d.TH * d.TW,
                       npix, ws);
    const int total = d.NB * d.OH * d.OW;
    hipLaunchKernelGGL(head1_fwd_gather_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ws, g, d.ioff_y, d.ioff_x, d.Nstore,
                       d.out, total);
    return CHECK_LAUNCH();
}

// data-gradient form (hip.conv_dgrad, stride 1): x = dy [N,h,w,>=1] channel 0, "NK" filter [KH][KW][512][1], flipped taps
extern "C" int ssc_head1_dgrad_supported(const ssc_conv_desc* dp) {
    const ssc_conv_desc& d = *dp;
    if (env_off()) return 0;
    return d.nphase == 1 && d.bmode == 1 && d.in_stride == 1 && d.out_stride == 1 && d.ooff_y == 0 && d.ooff_x == 0 &&
           (d.kstep == 1 || d.kstep == -1) && d.k_real == 1 && narrow_ok(d.x) && d.wC0 == HC && d.wC1 == 1 && d.n_off == 0 &&
           d.Nn == HC && d.Nstore == HC && d.ldc == HC && d.TH * d.TW <= MAXT && d.KH == d.TH && d.KW == d.TW &&
           d.bias == nullptr && d.epi == 0 && !d.accumulate && d.stat_partial == nullptr && d.sb_x == nullptr && al16(d.w) &&
           al16(d.out) && d.OH == d.PH && d.OW == d.PW && geo_ok(d.NB, d.OH, d.OW);
}

extern "C" int ssc_head1_dgrad(const ssc_conv_desc* dp, void* stream) {
    if (!ssc_head1_dgrad_supported(dp)) return -1;
    const ssc_conv_desc& d = *dp;
    const int npix = d.NB * d.OH * d.OW;
    // lattice point (py, px) of the wide grid reads dy at (py + ioff_y + ty, px + ioff_x + tx): "narrow pixel = iy - off - ty"
    // with off = -ioff and ty counted downwards; h1_narrow_at subtracts, so mirror the tap index instead: dy row =
    // iy - (-ioff_y - (TH-1)) - (TH-1-ty)
    H1Geo g = make_geo(d.NB, d.OH, d.OW, d.x.H, d.x.W, d.x.C0, d.TH, d.TW, -d.ioff_y - (d.TH - 1), -d.ioff_x - (d.TW - 1),
                       nullptr);
    // lane t reads the mirrored lattice tap (TH-1-ty, TW-1-tx), whose filter tap is (ky0 + kstep*(TH-1-ty), ...)
    const int ky0 = d.ky0 + d.kstep * (d.TH - 1), kx0 = d.kx0 + d.kstep * (d.TW - 1), kstep = -d.kstep;
    hipLaunchKernelGGL(head1_dgrad_kernel, dim3(wave_blocks(npix)), dim3(256), 0, (hipStream_t)stream, d.x.s0, d.w, g, ky0, kx0,
                       kstep, npix, d.out);
    return CHECK_LAUNCH();
}

// filter-gradient form: dw[tap][c] = sum_p act(x[p@tap][c]) * dy[p][0]
extern "C" int ssc_head1_wgrad_supported(const ssc_wgrad_desc* dp) {
    const ssc_wgrad_desc& d = *dp;
    if (env_off()) return 0;
    return view_ok(d.g) && narrow_ok(d.d) && d.in_stride == 1 && d.Cg_real == HC && d.Nn == 1 && d.ldc == 1 &&
           d.TH * d.TW <= MAXT && al16(d.out) && geo_ok(d.NB, d.g.H, d.g.W);
}

extern "C" int ssc_head1_wgrad(const ssc_wgrad_desc* dp, float* ws, int64_t ws_bytes, void* stream) {
    if (!ssc_head1_wgrad_supported(dp)) return -1;
    const ssc_wgrad_desc& d = *dp;
    const int npix = d.NB * d.g.H * d.g.W;
    int blocks = 256;
    while (blocks > 1 && (int64_t)blocks * MAXT * HC * 4 > ws_bytes) blocks /= 2;
    if (ws == nullptr || (int64_t)blocks * MAXT * HC * 4 > ws_bytes) return -2;
    const H1Geo g = make_geo(d.NB, d.g.H, d.g.W, d.PH, d.PW, d.d.C0, d.TH, d.TW, d.ioff_y, d.ioff_x, &d.g);
    hipStream_t st = (hipStream_t)stream;
    static unsigned long long attr_done = 0;
    const size_t lds = (size_t)3 * MAXT * HC * sizeof(float);
    {
        const int arc = ssc_set_max_lds(reinterpret_cast<const void*>(&head1_wgrad_kernel), (int)lds, &attr_done);
        if (arc != 0) return arc;
    }
    hipLaunchKernelGGL(head1_wgrad_kernel, dim3(blocks), dim3(256), lds, st, d.g.s0, d.g.ab0, d.d.s0, g, npix, ws);
    ssc_launch_wgrad_reduce(ws, (long)d.TH * d.TW * HC, blocks, d.out, d.accumulate, st);     // slabs are [TH*TW][HC]
    return CHECK_LAUNCH();
}

// data gradient of the patch head fused with the backward of the batch norm + activation of the wide tensor x [npix][512]
// (ab = [a; b], stats = [mean; rstd]): dx = d loss / d x; dscale / doffset (may be NULL) the norm's parameter gradients;
// rowb [NB][512] (may be NULL): a per-image gradient term added to every pixel, scaled by rowb_scale.
// ws: (blocks * 2 + 2) * 512 floats.  `dp` is the hip.conv_dgrad descriptor of the head (its `out` is not written).
extern "C" int ssc_head1_dgrad_bn_backward(const ssc_conv_desc* dp, const float* x, const float* ab, const float* stats, int act,
                                           const float* rowb, float rowb_scale, float* dx, float* dscale, float* doffset,
                                           float* ws, int64_t ws_bytes, void* stream) {
    ssc_conv_desc t = *dp;
    t.out = dx;             // only its alignment is looked at
    if (!ssc_head1_dgrad_supported(&t)) return -1;
    if (x == nullptr || ab == nullptr || stats == nullptr || dx == nullptr || !al16(x) || !al16(ab) || !al16(stats) ||
        (rowb != nullptr && !al16(rowb)))
        return -1;
    const ssc_conv_desc& d = *dp;
    const int npix = d.NB * d.OH * d.OW;
    int blocks = (int)wave_blocks(npix);
    if (blocks > 512) blocks = 512;
    if (ws == nullptr || !al16(ws) || ((int64_t)blocks * 2 + 2) * HC * 4 > ws_bytes) return -2;
    const H1Geo g = make_geo(d.NB, d.OH, d.OW, d.x.H, d.x.W, d.x.C0, d.TH, d.TW, -d.ioff_y - (d.TH - 1), -d.ioff_x - (d.TW - 1),
                             nullptr);
    const int ky0 = d.ky0 + d.kstep * (d.TH - 1), kx0 = d.kx0 + d.kstep * (d.TW - 1), kstep = -d.kstep;
    H1Bn b;
    b.x = x; b.ab = ab; b.stats = stats; b.rowb = rowb; b.rowb_scale = rowb_scale; b.act = act;
    float* partial = ws;
    float* coef = ws + (long)blocks * 2 * HC;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(head1_bnbwd_partial_kernel, dim3(blocks), dim3(256), 0, st, d.x.s0, d.w, g, ky0, kx0, kstep, npix, b,
                       partial);
    const int rc = ssc_bn_bwd_finalize(partial, blocks, HC, (int64_t)npix, coef, dscale, doffset, stream);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(head1_bnbwd_apply_kernel, dim3(wave_blocks(npix)), dim3(256), 0, st, d.x.s0, d.w, g, ky0, kx0, kstep, npix,
                       b, coef, dx);
    return CHECK_LAUNCH();
}
