// mru_ops.hip -- HBM-bound pointwise / reduction kernels of the MRU blocks (mru.py:353-461, 527-591):
// conditional batch-norm folding, miu_relu, min-max normalised gates, the gated merges, 2x2 mean-pool,
// nearest 2x upsample fused into the channel-concat writer.  NHWC fp32.  Channel counts that are multiples of 4 (every MRU
// state tensor) take the 16-byte forms (`*_v4`): a workgroup covers 256 >> txl rows x all groups of 4 channels, thread x walks
// the groups, so there is no per-element integer division and every lane moves 16 bytes per access; the sample index of a row
// comes from one multiply-high.  The scalar forms (grid-stride over (row, channel)) remain for other widths.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sketchycolor_hip.h"

#define CHECK_LAUNCH() ((int)hipGetLastError())

__device__ __forceinline__ float mru_act(float v, int act) {
    switch (act) {
        case SSC_ACT_RELU: return fmaxf(v, 0.f);
        case SSC_ACT_LRELU: return fmaxf(v, 0.2f * v);
        case SSC_ACT_TANH: return tanhf(v);
        case SSC_ACT_MIU: return (v + sqrtf(0.09f + v * v)) * 0.5f;   // miu = 0.7: (1 - miu)^2 = 0.09 (models_collection.py:63-65)
        default: return v;
    }
}

// ---- 16-byte forms: thread layout and row decode
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte access at 4-byte alignment (concat slices)
struct PwMap {
    int txl, cg;                // log2 of the threads that walk the channel groups of a row (fallback layout); groups per row
    unsigned long mP, oneP;     // row / P by multiply-high (exact for 32-bit rows)
    unsigned mW, oneW;          // pix / W (pix < P, P * W < 2^32)
    unsigned long mG, oneG;     // flat index / cg
};
static inline PwMap pw_map(int cg, long P, int W) {
    PwMap m;
    m.txl = 0;
    while ((1 << m.txl) < cg && m.txl < 8) ++m.txl;
    m.cg = cg;
    m.mP = P <= 1 ? 0UL : (unsigned long)((((unsigned __int128)1) << 64) / (unsigned long)P) + 1UL;
    m.oneP = P <= 1 ? ~0UL : 0UL;
    m.mW = W <= 1 ? 0u : (unsigned)(0x100000000ULL / (unsigned)W) + 1u;
    m.oneW = W <= 1 ? ~0u : 0u;
    m.mG = cg <= 1 ? 0UL : (unsigned long)((((unsigned __int128)1) << 64) / (unsigned long)cg) + 1UL;
    m.oneG = cg <= 1 ? ~0UL : 0UL;
    return m;
}
// the flat layout (below) covers rows * groups < 2^32
static inline bool pw_flat(long M, const PwMap& m) { return M * (long)m.cg < 0xffffffffL; }
static inline unsigned pw_blocks(long M, const PwMap& m) {
    long b;
    if (pw_flat(M, m)) {
        b = (M * m.cg + 255) / 256;
    } else {
        const long ty = 256 >> m.txl;
        b = (M + ty - 1) / ty;
    }
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}
__device__ __forceinline__ int pw_sample(long row, const PwMap& m) {
    return (int)(__umul64hi((unsigned long)row, m.mP) + ((unsigned long)row & m.oneP));
}
__device__ __forceinline__ int pw_div_w(int pix, const PwMap& m) {
    return (int)(__umulhi((unsigned)pix, m.mW) + ((unsigned)pix & m.oneW));
}
// f(row, c): every (row < M, c = 4 * group < 4 * cg).  Flat layout: consecutive threads take consecutive 16-byte groups ACROSS rows, so
// no lane idles when the groups per row are not a power of two (33 of 64 lanes worked on the 128 + 4-channel concats); rows *
// groups >= 2^32 keeps the row-per-thread-group layout
template <class F>
__device__ __forceinline__ void pw_rows(long M, int cg, const PwMap& m, F f) {
    const long total = M * (long)cg;
    if (total < 0xffffffffL) {
        for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
            const long row = (long)(__umul64hi((unsigned long)idx, m.mG) + ((unsigned long)idx & m.oneG));
            f(row, (int)(idx - row * cg) * 4);
        }
        return;
    }
    const int TX = 1 << m.txl, TY = 256 >> m.txl;
    const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> m.txl;
    for (long row = (long)blockIdx.x * TY + ty; row < M; row += (long)gridDim.x * TY)
        for (int g = tx; g < cg; g += TX) f(row, g * 4);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 ld4u(const float* p) {
    const f4u v = *reinterpret_cast<const f4u*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4u(float* p, const float4& v) {
    const f4u u = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<f4u*>(p) = u;
}
static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
// the multiply-high decodes are exact for rows < 2^32 and pix * W < 2^32
static inline bool pw_ok(long rows, long P, int W) { return rows < 0xffffffffL && P * (long)(W > 0 ? W : 1) < 0xffffffffL; }
#define PW_EACH(expr_x, expr_y, expr_z, expr_w) make_float4(expr_x, expr_y, expr_z, expr_w)

static inline unsigned grid_for(long total) {
    long b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ------------------------------------------------------------------ 2x2 mean pool (mru.py:15-19)
__global__ void mean_pool2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo, int N, int H,
                                  int W, int C) {
    const int oh = H / 2, ow = W / 2;
    const long tot = (long)N * oh * ow * C, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const int c = (int)(i % C);
        long r = i / C;
        const int xo = (int)(r % ow);
        r /= ow;
        const int yo = (int)(r % oh);
        const int n = (int)(r / oh);
        const float* p = x + (((long)n * H + 2 * yo) * W + 2 * xo) * ldx + c;
        // add_n([x[::2,::2], x[1::2,::2], x[::2,1::2], x[1::2,1::2]]) / 4
        const float v = ((p[0] + p[(long)W * ldx]) + p[ldx]) + p[(long)W * ldx + ldx];
        out[(((long)n * oh + yo) * ow + xo) * ldo + c] = v * 0.25f;
    }
}

// 16-byte form of mean_pool2 / pool2: out (+)= ((p00 + p10) + p01 + p11) * scale, the scalar kernels' order of additions
__global__ __launch_bounds__(256) void pool2_v4_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo,
                                                       int N, int H, int W, int C, float scale, int accumulate, PwMap m) {
    const int oh = H / 2, ow = W / 2;
    pw_rows((long)N * oh * ow, C / 4, m, [&](long row, int c) {
        const int n = pw_sample(row, m);                 // m decodes output rows: P = oh * ow, W = ow
        const int pix = (int)(row - (long)n * oh * ow);
        const int yo = pw_div_w(pix, m), xo = pix - yo * ow;
        const float* p = x + (((long)n * H + 2 * yo) * W + 2 * xo) * ldx + c;
        const float4 a = ld4(p), b = ld4(p + (long)W * ldx), e = ld4(p + ldx), f = ld4(p + (long)W * ldx + ldx);
        float4 v = PW_EACH((((a.x + b.x) + e.x) + f.x) * scale, (((a.y + b.y) + e.y) + f.y) * scale,
                           (((a.z + b.z) + e.z) + f.z) * scale, (((a.w + b.w) + e.w) + f.w) * scale);
        float* o = out + row * ldo + c;
        if (accumulate) {
            const float4 t = ld4(o);
            v = PW_EACH(t.x + v.x, t.y + v.y, t.z + v.z, t.w + v.w);
        }
        st4(o, v);
    });
}
static bool pool2_v4(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int C, float scale, int accumulate,
                     hipStream_t st) {
    if ((C & 3) || (ldx & 3) || (ldo & 3) || !al16(x) || !al16(out) || !pw_ok((long)N * (H / 2) * (W / 2), (long)(H / 2) * (W / 2), W / 2))
        return false;
    const PwMap m = pw_map(C / 4, (long)(H / 2) * (W / 2), W / 2);
    hipLaunchKernelGGL(pool2_v4_kernel, dim3(pw_blocks((long)N * (H / 2) * (W / 2), m)), dim3(256), 0, st, x, ldx, out, ldo, N,
                       H, W, C, scale, accumulate, m);
    return true;
}

extern "C" int ssc_mean_pool2(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int C, void* stream) {
    if ((H | W) & 1) return -1;
    if (pool2_v4(x, ldx, out, ldo, N, H, W, C, 0.25f, 0, (hipStream_t)stream)) return CHECK_LAUNCH();
    hipLaunchKernelGGL(mean_pool2_kernel, dim3(grid_for((long)N * (H / 2) * (W / 2) * C)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, out, ldo, N, H, W, C);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ conditional batch norm fold (models_collection.py:22-35)
// stats = [mean(C); rstd(C)] from ssc_bn_stats; abn[n] = [scale[label_n]*rstd ; offset[label_n] - mean*a]
__global__ void cbn_fold_kernel(const float* __restrict__ stats, const float* __restrict__ scale_m,
                                const float* __restrict__ offset_m, const int* __restrict__ labels, int N, int C,
                                float* __restrict__ abn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    const int l = labels[n];
    const float a = scale_m[(long)l * C + c] * stats[C + c];
    abn[(long)n * 2 * C + c] = a;
    abn[(long)n * 2 * C + C + c] = offset_m[(long)l * C + c] - stats[c] * a;
}

extern "C" int ssc_cbn_fold(const float* stats, const float* scale_m, const float* offset_m, const int32_t* labels,
                            int N, int C, float* abn, void* stream) {
    hipLaunchKernelGGL(cbn_fold_kernel, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, scale_m,
                       offset_m, labels, N, C, abn);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ per-(sample, channel) min / max over H*W
// stage 1: block (channel chunk of 64, split s, sample n) scans its rows; stage 2 folds the splits.
__global__ void minmax_partial_kernel(const float* __restrict__ x, int ld, int P, int C, int nsplit,
                                      float* __restrict__ part) {
    __shared__ float smn[4][64], smx[4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, s = blockIdx.y, n = blockIdx.z;
    const int rows = (P + nsplit - 1) / nsplit;
    const int r0 = s * rows, r1 = min(P, r0 + rows);
    float mn = INFINITY, mx = -INFINITY;
    if (c < C) {
        const float* p = x + (long)n * P * ld + c;
        for (int r = r0 + rl; r < r1; r += 4) {
            const float v = p[(long)r * ld];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    }
    smn[rl][lane] = mn;
    smx[rl][lane] = mx;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int k = 1; k < 4; ++k) {
            mn = fminf(mn, smn[k][lane]);
            mx = fmaxf(mx, smx[k][lane]);
        }
        float* o = part + (((long)n * nsplit + s) * 2) * C + c;
        o[0] = mn;
        o[C] = mx;
    }
}

// 16-byte forms of the per-(sample, channel) reductions: block (chunk of 4 * tg channels, split s, sample n); thread =
// (channel group of 4, row lane); the row lanes of a group combine through LDS.  Same `part` layout as the scalar kernels.
struct RedMap { int tgl, _pad; };      // log2 of the channel groups per block
static inline RedMap red_map(int C) {
    RedMap m;
    m.tgl = 0;
    while ((1 << m.tgl) < C / 4 && m.tgl < 6) ++m.tgl;
    m._pad = 0;
    return m;
}
static inline unsigned red_chunks(int C, const RedMap& m) { return (unsigned)((C / 4 + (1 << m.tgl) - 1) >> m.tgl); }

__global__ __launch_bounds__(256) void minmax_partial_v4_kernel(const float* __restrict__ x, int ld, int P, int C, int nsplit,
                                                                float* __restrict__ part, RedMap m) {
    __shared__ float4 smn[256], smx[256];
    const int TG = 1 << m.tgl, RL = 256 >> m.tgl;
    const int tg = threadIdx.x & (TG - 1), rl = threadIdx.x >> m.tgl;
    const int c = (blockIdx.x * TG + tg) * 4, s = blockIdx.y, n = blockIdx.z;
    const int rows = (P + nsplit - 1) / nsplit;
    const int r0 = s * rows, r1 = min(P, r0 + rows);
    float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY), mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (c < C) {
        const float* p = x + (long)n * P * ld + c;
        int r = r0 + rl;
        for (; r + RL < r1; r += 2 * RL) {
            const float4 a = ld4(p + (long)r * ld), b = ld4(p + (long)(r + RL) * ld);
            mn = PW_EACH(fminf(mn.x, fminf(a.x, b.x)), fminf(mn.y, fminf(a.y, b.y)), fminf(mn.z, fminf(a.z, b.z)),
                         fminf(mn.w, fminf(a.w, b.w)));
            mx = PW_EACH(fmaxf(mx.x, fmaxf(a.x, b.x)), fmaxf(mx.y, fmaxf(a.y, b.y)), fmaxf(mx.z, fmaxf(a.z, b.z)),
                         fmaxf(mx.w, fmaxf(a.w, b.w)));
        }
        if (r < r1) {
            const float4 a = ld4(p + (long)r * ld);
            mn = PW_EACH(fminf(mn.x, a.x), fminf(mn.y, a.y), fminf(mn.z, a.z), fminf(mn.w, a.w));
            mx = PW_EACH(fmaxf(mx.x, a.x), fmaxf(mx.y, a.y), fmaxf(mx.z, a.z), fmaxf(mx.w, a.w));
        }
    }
    smn[threadIdx.x] = mn;
    smx[threadIdx.x] = mx;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int k = 1; k < RL; ++k) {
            const float4 a = smn[k * TG + tg], b = smx[k * TG + tg];
            mn = PW_EACH(fminf(mn.x, a.x), fminf(mn.y, a.y), fminf(mn.z, a.z), fminf(mn.w, a.w));
            mx = PW_EACH(fmaxf(mx.x, b.x), fmaxf(mx.y, b.y), fmaxf(mx.z, b.z), fmaxf(mx.w, b.w));
        }
        float* o = part + (((long)n * nsplit + s) * 2) * C + c;
        st4(o, mn);
        st4(o + C, mx);
    }
}

// fold of the partial extrema [n][split][2][C]: block = (32 channels, one sample), its 8 thread groups take every 8th split and
// meet in LDS.  (One thread per (sample, channel) walking all splits alone took up to 257 us behind a conv epilogue's 576 row
// tiles per 192 x 192 sample with eight workgroups on the chip: 5 % of an MRU inference pass.)  min / max: any order, same bits.
__global__ __launch_bounds__(256) void minmax_final_kernel(const float* __restrict__ part, int nsplit, int N, int C,
                                                           float* __restrict__ mnmx) {
    __shared__ float smn[8][32], smx[8][32];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int n = blockIdx.y, c = blockIdx.x * 32 + cl;
    float mn = INFINITY, mx = -INFINITY;
    if (c < C) {
        for (int s = g; s < nsplit; s += 8) {
            const float* p = part + (((long)n * nsplit + s) * 2) * C + c;
            mn = fminf(mn, p[0]);
            mx = fmaxf(mx, p[C]);
        }
    }
    smn[g][cl] = mn;
    smx[g][cl] = mx;
    __syncthreads();
    if (g == 0 && c < C) {
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            mn = fminf(mn, smn[q][cl]);
            mx = fmaxf(mx, smx[q][cl]);
        }
        mnmx[(long)n * 2 * C + c] = mn;
        mnmx[(long)n * 2 * C + C + c] = mx;
    }
}

extern "C" int ssc_minmax_finalize(const float* part, int nsplit, int N, int C, float* mnmx, void* stream) {
    hipLaunchKernelGGL(minmax_final_kernel, dim3((C + 31) / 32, N), dim3(256), 0, (hipStream_t)stream, part, nsplit, N, C, mnmx);
    return CHECK_LAUNCH();
}

extern "C" int ssc_minmax_hw(const float* x, int ld, int N, int P, int C, float* mnmx, float* workspace,
                             int64_t workspace_bytes, void* stream) {
    int nsplit = (P + 255) / 256;
    if (nsplit > 64) nsplit = 64;
    if ((int64_t)N * nsplit * 2 * C * 4 > workspace_bytes) return -2;
    if ((C & 3) == 0 && (ld & 3) == 0 && al16(x) && al16(workspace)) {
        const RedMap m = red_map(C);
        hipLaunchKernelGGL(minmax_partial_v4_kernel, dim3(red_chunks(C, m), nsplit, N), dim3(256), 0, (hipStream_t)stream, x, ld,
                           P, C, nsplit, workspace, m);
    } else {
        hipLaunchKernelGGL(minmax_partial_kernel, dim3((C + 63) / 64, nsplit, N), dim3(256), 0, (hipStream_t)stream, x, ld,
                           P, C, nsplit, workspace);
    }
    hipLaunchKernelGGL(minmax_final_kernel, dim3((C + 31) / 32, N), dim3(256), 0, (hipStream_t)stream, workspace,
                       nsplit, N, C, mnmx);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ channel-concat writer
__device__ __forceinline__ float cat_elem(const ssc_cat_part& q, long srow, long row, int n, int c) {
    float v = q.x[srow * q.ld + c];
    if (q.act == SSC_ACT_PRELU) {
        v = fmaxf(q.ab[0] * v, v);       // q.ab points at the scalar leak
    } else {
        if (q.ab != nullptr) {
            const float* ab = q.ab + (long)n * q.ab_sample_stride;
            v = fmaf(ab[c], v, ab[q.C + c]);
        }
        v = mru_act(v, q.act);
    }
    if (q.gate != nullptr) {
        const float mn = q.mnmx[(long)n * 2 * q.C + c], mx = q.mnmx[(long)n * 2 * q.C + q.C + c];
        v *= (q.gate[row * q.C + c] - mn) / (mx - mn);
    }
    return v;
}

// one thread per (row, group of 4 output columns).  A group that lies inside one part whose channel count and row
// stride are multiples of 4 moves as float4 (the common case: the state / gradient tensors); the 3-channel image
// part and everything after it fall back to per-element code.
__global__ void concat_parts_kernel(ssc_cat_desc d) {
    const int c0 = d.p[0].C, c1 = c0 + (d.nparts > 1 ? d.p[1].C : 0), ct = c1 + (d.nparts > 2 ? d.p[2].C : 0);
    const int ng = (ct + 3) / 4;
    const long P = (long)d.H * d.W;
    const long tot = (long)d.N * P * ng, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const int col = (int)(i % ng) * 4;
        const long row = i / ng;
        const int n = (int)(row / P);
        const int pix = (int)(row - (long)n * P);
        const int py = pix / d.W, px = pix - py * d.W;
        const long uprow = ((long)n * (d.H / 2) + (py >> 1)) * (d.W / 2) + (px >> 1);
        const int k = col < c0 ? 0 : (col < c1 ? 1 : 2);
        const int base = k == 0 ? 0 : (k == 1 ? c0 : c1);
        const ssc_cat_part& q = d.p[k];
        const int c = col - base;
        float* o = d.out + row * d.ldo + col;
        if (c + 4 <= q.C && ((q.C | q.ld | c) & 3) == 0 && (d.ldo & 3) == 0) {
            const long srow = q.upsample ? uprow : row;
            float4 v = *reinterpret_cast<const float4*>(q.x + srow * q.ld + c);
            if (q.act == SSC_ACT_PRELU) {
                const float lk = q.ab[0];
                v.x = fmaxf(lk * v.x, v.x); v.y = fmaxf(lk * v.y, v.y); v.z = fmaxf(lk * v.z, v.z); v.w = fmaxf(lk * v.w, v.w);
            } else {
                if (q.ab != nullptr) {
                    const float* ab = q.ab + (long)n * q.ab_sample_stride;
                    const float4 a = *reinterpret_cast<const float4*>(ab + c), b = *reinterpret_cast<const float4*>(ab + q.C + c);
                    v.x = fmaf(a.x, v.x, b.x); v.y = fmaf(a.y, v.y, b.y); v.z = fmaf(a.z, v.z, b.z); v.w = fmaf(a.w, v.w, b.w);
                }
                if (q.act != SSC_ACT_NONE) {
                    v.x = mru_act(v.x, q.act); v.y = mru_act(v.y, q.act); v.z = mru_act(v.z, q.act); v.w = mru_act(v.w, q.act);
                }
            }
            if (q.gate != nullptr) {
                const float* mm = q.mnmx + (long)n * 2 * q.C;
                const float4 mn = *reinterpret_cast<const float4*>(mm + c), mx = *reinterpret_cast<const float4*>(mm + q.C + c);
                const float4 g = *reinterpret_cast<const float4*>(q.gate + row * q.C + c);
                v.x *= (g.x - mn.x) / (mx.x - mn.x); v.y *= (g.y - mn.y) / (mx.y - mn.y);
                v.z *= (g.z - mn.z) / (mx.z - mn.z); v.w *= (g.w - mn.w) / (mx.w - mn.w);
            }
            *reinterpret_cast<float4*>(o) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cc = col + e;
                if (cc >= ct) break;
                const int kk = cc < c0 ? 0 : (cc < c1 ? 1 : 2);
                const ssc_cat_part& qq = d.p[kk];
                const long srow = qq.upsample ? uprow : row;
                o[e] = cat_elem(qq, srow, row, n, cc - (kk == 0 ? 0 : (kk == 1 ? c0 : c1)));
            }
        }
    }
}

// 16-byte form: thread x walks the groups of 4 SOURCE channels of all parts (so the reads are aligned whatever the part's
// offset in the output); a part that starts at an odd output column (everything after the 3-channel image) is stored with
// 4-byte-aligned 16-byte stores.  Parts: C % 4 == 0 with ld % 4 == 0, or C < 4 with ld == 4 (the padded image).
struct CatV4 { int ng[3], base[3], al[3]; int ngt; };
__global__ __launch_bounds__(256) void concat_parts_v4_kernel(ssc_cat_desc d, CatV4 cv, PwMap m) {
    const long P = (long)d.H * d.W;
    pw_rows((long)d.N * P, cv.ngt, m, [&](long row, int c4) {
        const int g = c4 >> 2;
        const int k = g < cv.ng[0] ? 0 : (g < cv.ng[0] + cv.ng[1] ? 1 : 2);
        const ssc_cat_part& q = d.p[k];
        const int c = (g - (k == 0 ? 0 : (k == 1 ? cv.ng[0] : cv.ng[0] + cv.ng[1]))) * 4;
        const int n = pw_sample(row, m);
        long srow = row;
        if (q.upsample) {
            const int pix = (int)(row - (long)n * P);
            const int py = pw_div_w(pix, m), px = pix - py * d.W;
            srow = ((long)n * (d.H / 2) + (py >> 1)) * (d.W / 2) + (px >> 1);
        }
        float4 v = ld4(q.x + srow * q.ld + c);
        if (q.act == SSC_ACT_PRELU) {
            const float lk = q.ab[0];
            v = PW_EACH(fmaxf(lk * v.x, v.x), fmaxf(lk * v.y, v.y), fmaxf(lk * v.z, v.z), fmaxf(lk * v.w, v.w));
        } else {
            if (q.ab != nullptr && q.C >= 4) {
                const float* ab = q.ab + (long)n * q.ab_sample_stride;
                const float4 a = ld4(ab + c), b = ld4(ab + q.C + c);
                v = PW_EACH(fmaf(a.x, v.x, b.x), fmaf(a.y, v.y, b.y), fmaf(a.z, v.z, b.z), fmaf(a.w, v.w, b.w));
            }
            if (q.act != SSC_ACT_NONE)
                v = PW_EACH(mru_act(v.x, q.act), mru_act(v.y, q.act), mru_act(v.z, q.act), mru_act(v.w, q.act));
        }
        if (q.gate != nullptr) {
            const float* mm = q.mnmx + (long)n * 2 * q.C;
            const float4 mn = ld4(mm + c), mx = ld4(mm + q.C + c), gt = ld4(q.gate + row * q.C + c);
            v = PW_EACH(v.x * ((gt.x - mn.x) / (mx.x - mn.x)), v.y * ((gt.y - mn.y) / (mx.y - mn.y)),
                        v.z * ((gt.z - mn.z) / (mx.z - mn.z)), v.w * ((gt.w - mn.w) / (mx.w - mn.w)));
        }
        float* o = d.out + row * d.ldo + cv.base[k] + c;
        if (q.C < 4) {                   // the image part: its real channels only (the pad column belongs to the next part)
            o[0] = v.x;
            if (q.C > 1) o[1] = v.y;
            if (q.C > 2) o[2] = v.z;
        } else if (cv.al[k]) {
            st4(o, v);
        } else {
            st4u(o, v);
        }
    });
}

extern "C" int ssc_concat_parts(const ssc_cat_desc* desc, void* stream) {
    const ssc_cat_desc& d = *desc;
    if (d.nparts < 1 || d.nparts > 3) return -1;
    long ct = 0;
    for (int k = 0; k < d.nparts; ++k) ct += d.p[k].C;
    // 16-byte form?
    {
        CatV4 cv;
        bool ok = true;
        int base = 0, ngt = 0;
        const bool out16 = (d.ldo & 3) == 0 && al16(d.out);
        for (int k = 0; k < 3; ++k) { cv.ng[k] = 0; cv.base[k] = 0; cv.al[k] = 0; }
        for (int k = 0; k < d.nparts && ok; ++k) {
            const ssc_cat_part& q = d.p[k];
            const bool wide = q.C >= 4 && (q.C & 3) == 0 && (q.ld & 3) == 0;
            const bool image = q.C < 4 && q.ld == 4 && q.ab == nullptr && q.gate == nullptr;
            if (!(wide || image) || !al16(q.x) || (q.gate != nullptr && (!al16(q.gate) || !al16(q.mnmx))) ||
                (q.ab != nullptr && q.act != SSC_ACT_PRELU && (!al16(q.ab) || (q.ab_sample_stride & 3))))
                ok = false;
            cv.ng[k] = (q.C + 3) / 4;
            cv.base[k] = base;
            cv.al[k] = (out16 && (base & 3) == 0) ? 1 : 0;
            base += q.C;
            ngt += cv.ng[k];
        }
        if (ok && pw_ok((long)d.N * d.H * d.W, (long)d.H * d.W, d.W)) {
            cv.ngt = ngt;
            const PwMap m = pw_map(ngt, (long)d.H * d.W, d.W);
            hipLaunchKernelGGL(concat_parts_v4_kernel, dim3(pw_blocks((long)d.N * d.H * d.W, m)), dim3(256), 0,
                               (hipStream_t)stream, d, cv, m);
            return CHECK_LAUNCH();
        }
    }
    hipLaunchKernelGGL(concat_parts_kernel, dim3(grid_for((long)d.N * d.H * d.W * ((ct + 3) / 4))), dim3(256), 0,
                       (hipStream_t)stream, d);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ ht + minmax(rg) * img_new   (mru.py:422)
__global__ void mru_gate_merge_kernel(const float* __restrict__ ht, const float* __restrict__ rg,
                                      const float* __restrict__ mnmx, const float* __restrict__ img,
                                      float* __restrict__ out, int N, long P, int C) {
    const long tot = (long)N * P * C, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const int c = (int)(i % C);
        const int n = (int)(i / (P * C));
        const float mn = mnmx[(long)n * 2 * C + c], mx = mnmx[(long)n * 2 * C + C + c];
        out[i] = ht[i] + (rg[i] - mn) / (mx - mn) * img[i];
    }
}

__global__ __launch_bounds__(256) void mru_gate_merge_v4_kernel(const float* __restrict__ ht, const float* __restrict__ rg,
                                                                const float* __restrict__ mnmx,
                                                                const float* __restrict__ img, float* __restrict__ out, int N,
                                                                long P, int C, PwMap m) {
    pw_rows((long)N * P, C / 4, m, [&](long row, int c) {
        const int n = pw_sample(row, m);
        const float4 mn = ld4(mnmx + (long)n * 2 * C + c), mx = ld4(mnmx + (long)n * 2 * C + C + c);
        const long i = row * C + c;
        const float4 h = ld4(ht + i), r = ld4(rg + i), g = ld4(img + i);
        st4(out + i, PW_EACH(h.x + (r.x - mn.x) / (mx.x - mn.x) * g.x, h.y + (r.y - mn.y) / (mx.y - mn.y) * g.y,
                             h.z + (r.z - mn.z) / (mx.z - mn.z) * g.z, h.w + (r.w - mn.w) / (mx.w - mn.w) * g.w));
    });
}

extern "C" int ssc_mru_gate_merge(const float* ht, const float* rg, const float* mnmx, const float* img, float* out,
                                  int N, int64_t P, int C, void* stream) {
    if ((C & 3) == 0 && al16(ht) && al16(rg) && al16(mnmx) && al16(img) && al16(out) && pw_ok((long)N * P, 1, 1)) {
        const PwMap m = pw_map(C / 4, (long)P, 1);
        hipLaunchKernelGGL(mru_gate_merge_v4_kernel, dim3(pw_blocks((long)N * P, m)), dim3(256), 0, (hipStream_t)stream, ht, rg,
                           mnmx, img, out, N, (long)P, C, m);
        return CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(mru_gate_merge_kernel, dim3(grid_for((long)N * P * C)), dim3(256), 0, (hipStream_t)stream, ht,
                       rg, mnmx, img, out, N, (long)P, C);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ ht*(1-zg) + h_new*zg   (mru.py:589)
// hp  = ht_ab ? miu(a*ht+b) : ht, read at (y/2, x/2) when ht_lowres (nearest upsample commutes with the 1x1 projection)
// h   = miu(a2*h2+b2);  z = (zg - mn)/(mx - mn)
__global__ void mru_blend_kernel(const float* __restrict__ ht, const float* __restrict__ ht_ab, int ht_lowres,
                                 const float* __restrict__ h2, const float* __restrict__ h2_ab,
                                 const float* __restrict__ zg, const float* __restrict__ mnmx, float* __restrict__ out,
                                 int N, int H, int W, int C) {
    const long P = (long)H * W;
    const long tot = (long)N * P * C, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const int c = (int)(i % C);
        const long row = i / C;
        const int n = (int)(row / P);
        long srow = row;
        if (ht_lowres) {
            const int pix = (int)(row - (long)n * P);
            const int y = pix / W, x = pix - y * W;
            srow = ((long)n * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1);
        }
        float hp = ht[srow * C + c];
        if (ht_ab != nullptr) {
            const float* ab = ht_ab + (long)n * 2 * C;
            hp = mru_act(fmaf(ab[c], hp, ab[C + c]), SSC_ACT_MIU);
        }
        const float* ab2 = h2_ab + (long)n * 2 * C;
        const float h = mru_act(fmaf(ab2[c], h2[i], ab2[C + c]), SSC_ACT_MIU);
        const float mn = mnmx[(long)n * 2 * C + c], mx = mnmx[(long)n * 2 * C + C + c];
        const float z = (zg[i] - mn) / (mx - mn);
        out[i] = hp * (1.f - z) + h * z;
    }
}

__device__ __forceinline__ float4 miu_affine4(const float4& v, const float* ab, int C, int c) {
    const float4 a = ld4(ab + c), b = ld4(ab + C + c);
    return PW_EACH(mru_act(fmaf(a.x, v.x, b.x), SSC_ACT_MIU), mru_act(fmaf(a.y, v.y, b.y), SSC_ACT_MIU),
                   mru_act(fmaf(a.z, v.z, b.z), SSC_ACT_MIU), mru_act(fmaf(a.w, v.w, b.w), SSC_ACT_MIU));
}
__device__ __forceinline__ long up_row(long row, int n, int H, int W, const PwMap& m) {
    const int pix = (int)(row - (long)n * H * W);
    const int y = pw_div_w(pix, m), x = pix - y * W;
    return ((long)n * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1);
}
__global__ __launch_bounds__(256) void mru_blend_v4_kernel(const float* __restrict__ ht, const float* __restrict__ ht_ab,
                                                           int ht_lowres, const float* __restrict__ h2,
                                                           const float* __restrict__ h2_ab, const float* __restrict__ zg,
                                                           const float* __restrict__ mnmx, float* __restrict__ out, int N,
                                                           int H, int W, int C, PwMap m) {
    pw_rows((long)N * H * W, C / 4, m, [&](long row, int c) {
        const int n = pw_sample(row, m);
        const long srow = ht_lowres ? up_row(row, n, H, W, m) : row;
        float4 hp = ld4(ht + srow * C + c);
        if (ht_ab != nullptr) hp = miu_affine4(hp, ht_ab + (long)n * 2 * C, C, c);
        const long i = row * C + c;
        const float4 h = miu_affine4(ld4(h2 + i), h2_ab + (long)n * 2 * C, C, c);
        const float4 mn = ld4(mnmx + (long)n * 2 * C + c), mx = ld4(mnmx + (long)n * 2 * C + C + c), zz = ld4(zg + i);
        const float4 z = PW_EACH((zz.x - mn.x) / (mx.x - mn.x), (zz.y - mn.y) / (mx.y - mn.y), (zz.z - mn.z) / (mx.z - mn.z),
                                 (zz.w - mn.w) / (mx.w - mn.w));
        st4(out + i, PW_EACH(hp.x * (1.f - z.x) + h.x * z.x, hp.y * (1.f - z.y) + h.y * z.y, hp.z * (1.f - z.z) + h.z * z.z,
                             hp.w * (1.f - z.w) + h.w * z.w));
    });
}

extern "C" int ssc_mru_blend(const float* ht, const float* ht_ab, int ht_lowres, const float* h2, const float* h2_ab,
                             const float* zg, const float* mnmx, float* out, int N, int H, int W, int C, void* stream) {
    if ((C & 3) == 0 && al16(ht) && al16(h2) && al16(zg) && al16(mnmx) && al16(out) && al16(h2_ab) &&
        (ht_ab == nullptr || al16(ht_ab)) && pw_ok((long)N * H * W, (long)H * W, W)) {
        const PwMap m = pw_map(C / 4, (long)H * W, W);
        hipLaunchKernelGGL(mru_blend_v4_kernel, dim3(pw_blocks((long)N * H * W, m)), dim3(256), 0, (hipStream_t)stream, ht,
                           ht_ab, ht_lowres, h2, h2_ab, zg, mnmx, out, N, H, W, C, m);
        return CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(mru_blend_kernel, dim3(grid_for((long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, ht,
                       ht_ab, ht_lowres, h2, h2_ab, zg, mnmx, out, N, H, W, C);
    return CHECK_LAUNCH();
}

// =====================================================================================================
// backward kernels of the MRU blocks
// =====================================================================================================
__device__ __forceinline__ float mru_act_grad(float z, int act) {
    switch (act) {
        case SSC_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case SSC_ACT_LRELU: return z > 0.f ? 1.f : 0.2f;
        case SSC_ACT_MIU: return 0.5f * (1.f + z * rsqrtf(0.09f + z * z));
        default: return 1.f;
    }
}

// ------------------------------------------------------------------ strided copy / add, 2x2 pooling with scale
__global__ void strided_copy_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, long M,
                                    int C, int accumulate) {
    const long tot = M * C, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const float v = src[r * lds + c];
        float* o = dst + r * ldd + c;
        *o = accumulate ? *o + v : v;
    }
}

// C % 4 == 0; the slices of a concat start at any column: 4-byte-aligned 16-byte accesses
__global__ __launch_bounds__(256) void strided_copy_v4_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst,
                                                              int ldd, long M, int C, int accumulate, PwMap m) {
    pw_rows(M, C / 4, m, [&](long row, int c) {
        float4 v = ld4u(src + row * lds + c);
        float* o = dst + row * ldd + c;
        if (accumulate) {
            const float4 t = ld4u(o);
            v = PW_EACH(t.x + v.x, t.y + v.y, t.z + v.z, t.w + v.w);
        }
        st4u(o, v);
    });
}

extern "C" int ssc_strided_copy(const float* src, int lds, float* dst, int ldd, int64_t M, int C, int accumulate,
                                void* stream) {
    if ((C & 3) == 0 && C >= 4) {
        const PwMap m = pw_map(C / 4, 1, 1);
        hipLaunchKernelGGL(strided_copy_v4_kernel, dim3(pw_blocks((long)M, m)), dim3(256), 0, (hipStream_t)stream, src, lds, dst,
                           ldd, (long)M, C, accumulate, m);
        return CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(strided_copy_kernel, dim3(grid_for((long)M * C)), dim3(256), 0, (hipStream_t)stream, src, lds,
                       dst, ldd, (long)M, C, accumulate);
    return CHECK_LAUNCH();
}

__global__ void pool2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo, int N, int H, int W,
                             int C, float scale, int accumulate) {
    const int oh = H / 2, ow = W / 2;
    const long tot = (long)N * oh * ow * C, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const int c = (int)(i % C);
        long r = i / C;
        const int xo = (int)(r % ow);
        r /= ow;
        const int yo = (int)(r % oh);
        const int n = (int)(r / oh);
        const float* p = x + (((long)n * H + 2 * yo) * W + 2 * xo) * ldx + c;
        const float v = (((p[0] + p[(long)W * ldx]) + p[ldx]) + p[(long)W * ldx + ldx]) * scale;
        float* o = out + (((long)n * oh + yo) * ow + xo) * ldo + c;
        *o = accumulate ? *o + v : v;
    }
}

extern "C" int ssc_pool2(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int C, float scale,
                         int accumulate, void* stream) {
    if ((H | W) & 1) return -1;
    if (pool2_v4(x, ldx, out, ldo, N, H, W, C, scale, accumulate, (hipStream_t)stream)) return CHECK_LAUNCH();
    hipLaunchKernelGGL(pool2_kernel, dim3(grid_for((long)N * (H / 2) * (W / 2) * C)), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, out, ldo, N, H, W, C, scale, accumulate);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ conditional norm + activation backward
// y = act(a[n,c]*x + b[n,c]),  a = scale[l_n,c]*rstd[c],  xhat = (x - mean)*rstd
// pass 1: S1[n,c] = sum_p gz, S2[n,c] = sum_p gz*xhat  (gz = gy*act'(z));   pass 2: tables + k1,k2;   pass 3: dx
__global__ void cbn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ abn,
                                       const float* __restrict__ stats, const float* __restrict__ gy, int ldg, int act,
                                       int P, int C, int nsplit, float* __restrict__ part) {
    __shared__ float s1[4][64], s2[4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, s = blockIdx.y, n = blockIdx.z;
    const int rows = (P + nsplit - 1) / nsplit;
    const int r0 = s * rows, r1 = min(P, r0 + rows);
    float a1 = 0.f, a2 = 0.f;
    if (c < C) {
        const float a = abn[(long)n * 2 * C + c], b = abn[(long)n * 2 * C + C + c];
        const float mean = stats[c], rstd = stats[C + c];
        for (int r = r0 + rl; r < r1; r += 4) {
            const long row = (long)n * P + r;
            const float xv = x[row * C + c];
            const float gz = gy[row * ldg + c] * mru_act_grad(fmaf(a, xv, b), act);
            a1 += gz;
            a2 += gz * (xv - mean) * rstd;
        }
    }
    s1[rl][lane] = a1;
    s2[rl][lane] = a2;
    __syncthreads();
    if (rl == 0 && c < C) {
        float* o = part + (((long)n * nsplit + s) * 2) * C + c;
        o[0] = (s1[0][lane] + s1[1][lane]) + (s1[2][lane] + s1[3][lane]);
        o[C] = (s2[0][lane] + s2[1][lane]) + (s2[2][lane] + s2[3][lane]);
    }
}

__global__ __launch_bounds__(256) void cbn_bwd_partial_v4_kernel(const float* __restrict__ x, const float* __restrict__ abn,
                                                                 const float* __restrict__ stats, const float* __restrict__ gy,
                                                                 int ldg, int act, int P, int C, int nsplit,
                                                                 float* __restrict__ part, RedMap m) {
    __shared__ float4 s1[256], s2[256];
    const int TG = 1 << m.tgl, RL = 256 >> m.tgl;
    const int tg = threadIdx.x & (TG - 1), rl = threadIdx.x >> m.tgl;
    const int c = (blockIdx.x * TG + tg) * 4, s = blockIdx.y, n = blockIdx.z;
    const int rows = (P + nsplit - 1) / nsplit;
    const int r0 = s * rows, r1 = min(P, r0 + rows);
    float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
    if (c < C) {
        const float4 a = ld4(abn + (long)n * 2 * C + c), b = ld4(abn + (long)n * 2 * C + C + c);
        const float4 mean = ld4(stats + c), rstd = ld4(stats + C + c);
        for (int r = r0 + rl; r < r1; r += RL) {
            const long row = (long)n * P + r;
            const float4 xv = ld4(x + row * C + c), g = ld4u(gy + row * ldg + c);
            const float4 gz = PW_EACH(g.x * mru_act_grad(fmaf(a.x, xv.x, b.x), act), g.y * mru_act_grad(fmaf(a.y, xv.y, b.y), act),
                                      g.z * mru_act_grad(fmaf(a.z, xv.z, b.z), act), g.w * mru_act_grad(fmaf(a.w, xv.w, b.w), act));
            a1 = PW_EACH(a1.x + gz.x, a1.y + gz.y, a1.z + gz.z, a1.w + gz.w);
            a2 = PW_EACH(a2.x + gz.x * (xv.x - mean.x) * rstd.x, a2.y + gz.y * (xv.y - mean.y) * rstd.y,
                         a2.z + gz.z * (xv.z - mean.z) * rstd.z, a2.w + gz.w * (xv.w - mean.w) * rstd.w);
        }
    }
    s1[threadIdx.x] = a1;
    s2[threadIdx.x] = a2;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int k = 1; k < RL; ++k) {
            const float4 u = s1[k * TG + tg], v = s2[k * TG + tg];
            a1 = PW_EACH(a1.x + u.x, a1.y + u.y, a1.z + u.z, a1.w + u.w);
            a2 = PW_EACH(a2.x + v.x, a2.y + v.y, a2.z + v.z, a2.w + v.w);
        }
        float* o = part + (((long)n * nsplit + s) * 2) * C + c;
        st4(o, a1);
        st4(o + C, a2);
    }
}

// folds the splits -> sn[n][2][C]
__global__ void cbn_bwd_fold_kernel(const float* __restrict__ part, int nsplit, int N, int C, float* __restrict__ sn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    float a1 = 0.f, a2 = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* p = part + (((long)n * nsplit + s) * 2) * C + c;
        a1 += p[0];
        a2 += p[C];
    }
    sn[(long)n * 2 * C + c] = a1;
    sn[(long)n * 2 * C + C + c] = a2;
}

// thread (l, c): table gradients; thread (L, c): k1, k2
__global__ void cbn_bwd_final_kernel(const float* __restrict__ sn, const float* __restrict__ scale_m,
                                     const int* __restrict__ labels, int N, int C, int L, float inv_m,
                                     float* __restrict__ dscale_m, float* __restrict__ doffset_m, int accumulate,
                                     float* __restrict__ kk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (L + 1) * C) return;
    const int l = i / C, c = i - l * C;
    if (l < L) {
        float ds = 0.f, dof = 0.f;
        for (int n = 0; n < N; ++n)
            if (labels[n] == l) {
                dof += sn[(long)n * 2 * C + c];
                ds += sn[(long)n * 2 * C + C + c];
            }
        if (dscale_m != nullptr) {
            dscale_m[(long)l * C + c] = accumulate ? dscale_m[(long)l * C + c] + ds : ds;
            doffset_m[(long)l * C + c] = accumulate ? doffset_m[(long)l * C + c] + dof : dof;
        }
    } else {
        float k1 = 0.f, k2 = 0.f;
        for (int n = 0; n < N; ++n) {
            const float sc = scale_m[(long)labels[n] * C + c];
            k1 += sc * sn[(long)n * 2 * C + c];
            k2 += sc * sn[(long)n * 2 * C + C + c];
        }
        kk[c] = k1 * inv_m;
        kk[C + c] = k2 * inv_m;
    }
}

__global__ void cbn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ abn,
                                     const float* __restrict__ stats, const float* __restrict__ kk,
                                     const float* __restrict__ gy, int ldg, int act, int N, long P, int C,
                                     float* __restrict__ dx, int lddx, int accumulate) {
    const long tot = (long)N * P * C, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const int c = (int)(i % C);
        const long row = i / C;
        const int n = (int)(row / P);
        const float a = abn[(long)n * 2 * C + c], b = abn[(long)n * 2 * C + C + c];
        const float xv = x[i];
        const float gz = gy[row * ldg + c] * mru_act_grad(fmaf(a, xv, b), act);
        const float mean = stats[c], rstd = stats[C + c];
        const float v = gz * a - rstd * (kk[c] + (xv - mean) * rstd * kk[C + c]);
        float* o = dx + row * lddx + c;
        *o = accumulate ? *o + v : v;
    }
}

__global__ __launch_bounds__(256) void cbn_bwd_apply_v4_kernel(const float* __restrict__ x, const float* __restrict__ abn,
                                                               const float* __restrict__ stats, const float* __restrict__ kk,
                                                               const float* __restrict__ gy, int ldg, int act, int N, long P,
                                                               int C, float* __restrict__ dx, int lddx, int accumulate,
                                                               PwMap m) {
    pw_rows((long)N * P, C / 4, m, [&](long row, int c) {
        const int n = pw_sample(row, m);
        const float4 a = ld4(abn + (long)n * 2 * C + c), b = ld4(abn + (long)n * 2 * C + C + c);
        const float4 xv = ld4(x + row * C + c), g = ld4u(gy + row * ldg + c);
        const float4 mean = ld4(stats + c), rstd = ld4(stats + C + c), k1 = ld4(kk + c), k2 = ld4(kk + C + c);
#define CBN1(f) (g.f * mru_act_grad(fmaf(a.f, xv.f, b.f), act) * a.f - rstd.f * (k1.f + (xv.f - mean.f) * rstd.f * k2.f))
        float4 v = PW_EACH(CBN1(x), CBN1(y), CBN1(z), CBN1(w));
#undef CBN1
        float* o = dx + row * lddx + c;
        if (accumulate) {
            const float4 t = ld4u(o);
            v = PW_EACH(t.x + v.x, t.y + v.y, t.z + v.z, t.w + v.w);
        }
        st4u(o, v);
    });
}

extern "C" int ssc_cbn_act_backward(const float* x, const float* abn, const float* stats, const float* scale_m,
                                    const int32_t* labels, int n_labels, const float* gy, int ldg, int act, int N,
                                    int P, int C, float* dx, int lddx, int accumulate_dx, float* dscale_m,
                                    float* doffset_m, int accumulate_params, float* workspace,
                                    int64_t workspace_bytes, void* stream) {
    int nsplit = (P + 255) / 256;
    if (nsplit > 64) nsplit = 64;
    const int64_t need = ((int64_t)N * nsplit * 2 * C + (int64_t)N * 2 * C + 2 * C) * 4;
    if (need > workspace_bytes) return -2;
    float* part = workspace;
    float* sn = part + (long)N * nsplit * 2 * C;
    float* kk = sn + (long)N * 2 * C;
    hipStream_t st = (hipStream_t)stream;
    if ((C & 3) == 0 && al16(x) && al16(abn) && al16(stats) && al16(part)) {
        const RedMap rm = red_map(C);
        hipLaunchKernelGGL(cbn_bwd_partial_v4_kernel, dim3(red_chunks(C, rm), nsplit, N), dim3(256), 0, st, x, abn, stats, gy,
                           ldg, act, P, C, nsplit, part, rm);
    } else {
        hipLaunchKernelGGL(cbn_bwd_partial_kernel, dim3((C + 63) / 64, nsplit, N), dim3(256), 0, st, x, abn, stats, gy, ldg,
                           act, P, C, nsplit, part);
    }
    hipLaunchKernelGGL(cbn_bwd_fold_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, part, nsplit, N, C, sn);
    hipLaunchKernelGGL(cbn_bwd_final_kernel, dim3(((n_labels + 1) * C + 255) / 256), dim3(256), 0, st, sn, scale_m,
                       labels, N, C, n_labels, 1.f / ((float)N * (float)P), dscale_m, doffset_m, accumulate_params, kk);
    if ((C & 3) == 0 && al16(x) && al16(abn) && al16(stats) && pw_ok((long)N * P, 1, 1)) {
        const PwMap m = pw_map(C / 4, (long)P, 1);
        hipLaunchKernelGGL(cbn_bwd_apply_v4_kernel, dim3(pw_blocks((long)N * P, m)), dim3(256), 0, st, x, abn, stats, kk, gy,
                           ldg, act, N, (long)P, C, dx, lddx, accumulate_dx, m);
    } else {
        hipLaunchKernelGGL(cbn_bwd_apply_kernel, dim3(grid_for((long)N * P * C)), dim3(256), 0, st, x, abn, stats, kk, gy, ldg,
                           act, N, (long)P, C, dx, lddx, accumulate_dx);
    }
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ prelu backward (models_collection.py:56-60)
// y = max(leak*x, x): the first argument takes the gradient where leak*x >= x (tf.maximum's tie rule).  The leak's gradient is ONE
// scalar summed over the whole tensor with heavy cancellation: products in fp32 (as TensorFlow forms them), the sum in double
// (the pass is bound by its loads; a float sum left the scalar at the noise floor of the end-to-end gradient test)
__global__ void prelu_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ leak_p,
                                 const float* __restrict__ gy, int ldg, long M, int C, float* __restrict__ dx, int lddx,
                                 int accumulate, double* __restrict__ part) {
    __shared__ double sh[256];
    const float leak = *leak_p;
    const long tot = M * C, stride = (long)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const float xv = x[r * ldx + c], g = gy[r * ldg + c];
        const bool first = leak * xv >= xv;
        if (first) acc += g * xv;
        if (dx != nullptr) {
            const float v = g * (first ? leak : 1.f);
            float* o = dx + r * lddx + c;
            *o = accumulate ? *o + v : v;
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void prelu_bwd_v4_kernel(const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ leak_p, const float* __restrict__ gy,
                                                           int ldg, long M, int C, float* __restrict__ dx, int lddx,
                                                           int accumulate, double* __restrict__ part, PwMap m) {
    __shared__ double sh[256];
    const float leak = *leak_p;
    double acc = 0.0;
    pw_rows(M, C / 4, m, [&](long row, int c) {
        const float4 xv = ld4u(x + row * ldx + c), g = ld4u(gy + row * ldg + c);
        const bool fx = leak * xv.x >= xv.x, fy = leak * xv.y >= xv.y, fz = leak * xv.z >= xv.z, fw = leak * xv.w >= xv.w;
        if (fx) acc += g.x * xv.x;
        if (fy) acc += g.y * xv.y;
        if (fz) acc += g.z * xv.z;
        if (fw) acc += g.w * xv.w;
        if (dx != nullptr) {
            float4 v = PW_EACH(g.x * (fx ? leak : 1.f), g.y * (fy ? leak : 1.f), g.z * (fz ? leak : 1.f), g.w * (fw ? leak : 1.f));
            float* o = dx + row * lddx + c;
            if (accumulate) {
                const float4 t = ld4u(o);
                v = PW_EACH(t.x + v.x, t.y + v.y, t.z + v.z, t.w + v.w);
            }
            st4u(o, v);
        }
    });
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ void scalar_fold_kernel(const double* __restrict__ part, int n, float* __restrict__ out, int accumulate) {
    __shared__ double sh[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = accumulate ? (float)((double)*out + sh[0]) : (float)sh[0];
}

extern "C" int ssc_prelu_backward(const float* x, int ldx, const float* leak, const float* gy, int ldg, int64_t M, int C,
                                  float* dx, int lddx, int accumulate_dx, float* dleak, int accumulate_leak,
                                  float* workspace, int64_t workspace_bytes, void* stream) {
    long blocks = ((long)M * C + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (blocks * 8 > workspace_bytes) return -2;
    double* const part = reinterpret_cast<double*>(workspace);
    if ((C & 3) == 0 && C >= 4) {
        const PwMap m = pw_map(C / 4, 1, 1);
        long vb = pw_blocks((long)M, m);
        if (vb > 2048) vb = 2048;
        if (vb * 8 > workspace_bytes) return -2;
        blocks = vb;
        hipLaunchKernelGGL(prelu_bwd_v4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, leak, gy,
                           ldg, (long)M, C, dx, lddx, accumulate_dx, part, m);
    } else
    hipLaunchKernelGGL(prelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, leak, gy, ldg,
                       (long)M, C, dx, lddx, accumulate_dx, part);
    hipLaunchKernelGGL(scalar_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, (int)blocks, dleak,
                       accumulate_leak);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ min-max gate backward (mru.py:414-415, 560-568)
// r = (g - mn)/(mx - mn), g = lrelu(pre) stored;  given gr: dpre.  reduce_min / reduce_max send their gradient to
// every position that attains the extremum, divided by the number of ties (TF's _MinOrMaxGrad).
__global__ void gate_bwd_partial_kernel(const float* __restrict__ g, const float* __restrict__ mnmx,
                                        const float* __restrict__ gr, int P, int C, int nsplit,
                                        float* __restrict__ part) {
    __shared__ float sh[4][4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, s = blockIdx.y, n = blockIdx.z;
    const int rows = (P + nsplit - 1) / nsplit;
    const int r0 = s * rows, r1 = min(P, r0 + rows);
    float A = 0.f, B = 0.f, cmn = 0.f, cmx = 0.f;
    if (c < C) {
        const float mn = mnmx[(long)n * 2 * C + c], mx = mnmx[(long)n * 2 * C + C + c];
        for (int r = r0 + rl; r < r1; r += 4) {
            const long i = ((long)n * P + r) * C + c;
            const float gv = g[i], q = gr[i];
            A += q * (gv - mx);
            B += q * (gv - mn);
            cmn += (gv == mn) ? 1.f : 0.f;
            cmx += (gv == mx) ? 1.f : 0.f;
        }
    }
    sh[0][rl][lane] = A; sh[1][rl][lane] = B; sh[2][rl][lane] = cmn; sh[3][rl][lane] = cmx;
    __syncthreads();
    if (rl == 0 && c < C) {
        float* o = part + (((long)n * nsplit + s) * 4) * C + c;
        for (int k = 0; k < 4; ++k) o[(long)k * C] = (sh[k][0][lane] + sh[k][1][lane]) + (sh[k][2][lane] + sh[k][3][lane]);
    }
}

__global__ __launch_bounds__(256) void gate_bwd_partial_v4_kernel(const float* __restrict__ g, const float* __restrict__ mnmx,
                                                                  const float* __restrict__ gr, int P, int C, int nsplit,
                                                                  float* __restrict__ part, RedMap m) {
    __shared__ float4 sh[4][256];
    const int TG = 1 << m.tgl, RL = 256 >> m.tgl;
    const int tg = threadIdx.x & (TG - 1), rl = threadIdx.x >> m.tgl;
    const int c = (blockIdx.x * TG + tg) * 4, s = blockIdx.y, n = blockIdx.z;
    const int rows = (P + nsplit - 1) / nsplit;
    const int r0 = s * rows, r1 = min(P, r0 + rows);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 A = zero, B = zero, cmn = zero, cmx = zero;
    if (c < C) {
        const float4 mn = ld4(mnmx + (long)n * 2 * C + c), mx = ld4(mnmx + (long)n * 2 * C + C + c);
        for (int r = r0 + rl; r < r1; r += RL) {
            const long i = ((long)n * P + r) * C + c;
            const float4 gv = ld4(g + i), q = ld4(gr + i);
            A = PW_EACH(A.x + q.x * (gv.x - mx.x), A.y + q.y * (gv.y - mx.y), A.z + q.z * (gv.z - mx.z), A.w + q.w * (gv.w - mx.w));
            B = PW_EACH(B.x + q.x * (gv.x - mn.x), B.y + q.y * (gv.y - mn.y), B.z + q.z * (gv.z - mn.z), B.w + q.w * (gv.w - mn.w));
            cmn = PW_EACH(cmn.x + (gv.x == mn.x ? 1.f : 0.f), cmn.y + (gv.y == mn.y ? 1.f : 0.f), cmn.z + (gv.z == mn.z ? 1.f : 0.f),
                          cmn.w + (gv.w == mn.w ? 1.f : 0.f));
            cmx = PW_EACH(cmx.x + (gv.x == mx.x ? 1.f : 0.f), cmx.y + (gv.y == mx.y ? 1.f : 0.f), cmx.z + (gv.z == mx.z ? 1.f : 0.f),
                          cmx.w + (gv.w == mx.w ? 1.f : 0.f));
        }
    }
    sh[0][threadIdx.x] = A; sh[1][threadIdx.x] = B; sh[2][threadIdx.x] = cmn; sh[3][threadIdx.x] = cmx;
    __syncthreads();
    if (rl == 0 && c < C) {
        float* o = part + (((long)n * nsplit + s) * 4) * C + c;
        for (int k = 0; k < 4; ++k) {
            float4 t = sh[k][tg];
            for (int j = 1; j < RL; ++j) {
                const float4 u = sh[k][j * TG + tg];
                t = PW_EACH(t.x + u.x, t.y + u.y, t.z + u.z, t.w + u.w);
            }
            st4(o + (long)k * C, t);
        }
    }
}

__global__ void gate_bwd_fold_kernel(const float* __restrict__ part, const float* __restrict__ mnmx, int nsplit, int N,
                                     int C, float* __restrict__ coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    float A = 0.f, B = 0.f, cmn = 0.f, cmx = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* p = part + (((long)n * nsplit + s) * 4) * C + c;
        A += p[0]; B += p[C]; cmn += p[2 * (long)C]; cmx += p[3 * (long)C];
    }
    const float d = mnmx[(long)n * 2 * C + C + c] - mnmx[(long)n * 2 * C + c];
    coef[(long)n * 2 * C + c] = A / (d * d) / cmn;          // d loss / d min, per tied position
    coef[(long)n * 2 * C + C + c] = -B / (d * d) / cmx;     // d loss / d max
}

__global__ void gate_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ mnmx,
                                      const float* __restrict__ coef, const float* __restrict__ gr, int N, long P, int C,
                                      float* __restrict__ dpre) {
    const long tot = (long)N * P * C, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const int c = (int)(i % C);
        const int n = (int)(i / (P * C));
        const float mn = mnmx[(long)n * 2 * C + c], mx = mnmx[(long)n * 2 * C + C + c];
        const float gv = g[i];
        float dg = gr[i] / (mx - mn);
        if (gv == mn) dg += coef[(long)n * 2 * C + c];
        if (gv == mx) dg += coef[(long)n * 2 * C + C + c];
        dpre[i] = dg * (gv > 0.f ? 1.f : 0.2f);
    }
}

__global__ __launch_bounds__(256) void gate_bwd_apply_v4_kernel(const float* __restrict__ g, const float* __restrict__ mnmx,
                                                                const float* __restrict__ coef, const float* __restrict__ gr,
                                                                int N, long P, int C, float* __restrict__ dpre, PwMap m) {
    pw_rows((long)N * P, C / 4, m, [&](long row, int c) {
        const int n = pw_sample(row, m);
        const float* mm = mnmx + (long)n * 2 * C;
        const float* cf = coef + (long)n * 2 * C;
        const float4 mn = ld4(mm + c), mx = ld4(mm + C + c), c0 = ld4(cf + c), c1 = ld4(cf + C + c);
        const long i = row * C + c;
        const float4 gv = ld4(g + i), q = ld4(gr + i);
#define GB1(f) ((q.f / (mx.f - mn.f) + (gv.f == mn.f ? c0.f : 0.f) + (gv.f == mx.f ? c1.f : 0.f)) * (gv.f > 0.f ? 1.f : 0.2f))
        st4(dpre + i, PW_EACH(GB1(x), GB1(y), GB1(z), GB1(w)));
#undef GB1
    });
}

extern "C" int ssc_minmax_gate_backward(const float* g, const float* mnmx, const float* gr, int N, int P, int C,
                                        float* dpre, float* workspace, int64_t workspace_bytes, void* stream) {
    int nsplit = (P + 255) / 256;
    if (nsplit > 64) nsplit = 64;
    const int64_t need = ((int64_t)N * nsplit * 4 * C + (int64_t)N * 2 * C) * 4;
    if (need > workspace_bytes) return -2;
    float* part = workspace;
    float* coef = part + (long)N * nsplit * 4 * C;
    hipStream_t st = (hipStream_t)stream;
    if ((C & 3) == 0 && al16(g) && al16(mnmx) && al16(gr) && al16(part)) {
        const RedMap rm = red_map(C);
        hipLaunchKernelGGL(gate_bwd_partial_v4_kernel, dim3(red_chunks(C, rm), nsplit, N), dim3(256), 0, st, g, mnmx, gr, P, C,
                           nsplit, part, rm);
    } else {
        hipLaunchKernelGGL(gate_bwd_partial_kernel, dim3((C + 63) / 64, nsplit, N), dim3(256), 0, st, g, mnmx, gr, P, C, nsplit,
                           part);
    }
    hipLaunchKernelGGL(gate_bwd_fold_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, part, mnmx, nsplit, N, C, coef);
    if ((C & 3) == 0 && al16(g) && al16(mnmx) && al16(gr) && al16(dpre) && al16(coef) && pw_ok((long)N * P, 1, 1)) {
        const PwMap m = pw_map(C / 4, (long)P, 1);
        hipLaunchKernelGGL(gate_bwd_apply_v4_kernel, dim3(pw_blocks((long)N * P, m)), dim3(256), 0, st, g, mnmx, coef, gr, N,
                           (long)P, C, dpre, m);
    } else {
        hipLaunchKernelGGL(gate_bwd_apply_kernel, dim3(grid_for((long)N * P * C)), dim3(256), 0, st, g, mnmx, coef, gr, N,
                           (long)P, C, dpre);
    }
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ ht_plus = ht + r*img backward: gr = g*img, g_img = g*r
__global__ void gate_merge_bwd_kernel(const float* __restrict__ ghtp, const float* __restrict__ rg,
                                      const float* __restrict__ mnmx, const float* __restrict__ img,
                                      float* __restrict__ gr, float* __restrict__ gimg, int N, long P, int C) {
    const long tot = (long)N * P * C, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const int c = (int)(i % C);
        const int n = (int)(i / (P * C));
        const float mn = mnmx[(long)n * 2 * C + c], mx = mnmx[(long)n * 2 * C + C + c];
        const float q = ghtp[i];
        gr[i] = q * img[i];
        gimg[i] = q * ((rg[i] - mn) / (mx - mn));
    }
}

__global__ __launch_bounds__(256) void gate_merge_bwd_v4_kernel(const float* __restrict__ ghtp, const float* __restrict__ rg,
                                                                const float* __restrict__ mnmx,
                                                                const float* __restrict__ img, float* __restrict__ gr,
                                                                float* __restrict__ gimg, int N, long P, int C, PwMap m) {
    pw_rows((long)N * P, C / 4, m, [&](long row, int c) {
        const int n = pw_sample(row, m);
        const float4 mn = ld4(mnmx + (long)n * 2 * C + c), mx = ld4(mnmx + (long)n * 2 * C + C + c);
        const long i = row * C + c;
        const float4 q = ld4(ghtp + i), im = ld4(img + i), r = ld4(rg + i);
        st4(gr + i, PW_EACH(q.x * im.x, q.y * im.y, q.z * im.z, q.w * im.w));
        st4(gimg + i, PW_EACH(q.x * ((r.x - mn.x) / (mx.x - mn.x)), q.y * ((r.y - mn.y) / (mx.y - mn.y)),
                              q.z * ((r.z - mn.z) / (mx.z - mn.z)), q.w * ((r.w - mn.w) / (mx.w - mn.w))));
    });
}

extern "C" int ssc_mru_gate_merge_backward(const float* ghtp, const float* rg, const float* mnmx, const float* img,
                                           float* gr, float* gimg, int N, int64_t P, int C, void* stream) {
    if ((C & 3) == 0 && al16(ghtp) && al16(rg) && al16(mnmx) && al16(img) && al16(gr) && al16(gimg) && pw_ok((long)N * P, 1, 1)) {
        const PwMap m = pw_map(C / 4, (long)P, 1);
        hipLaunchKernelGGL(gate_merge_bwd_v4_kernel, dim3(pw_blocks((long)N * P, m)), dim3(256), 0, (hipStream_t)stream, ghtp,
                           rg, mnmx, img, gr, gimg, N, (long)P, C, m);
        return CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(gate_merge_bwd_kernel, dim3(grid_for((long)N * P * C)), dim3(256), 0, (hipStream_t)stream, ghtp,
                       rg, mnmx, img, gr, gimg, N, (long)P, C);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ blend backward: out = hp*(1-z) + h*z
// ghp = gout*(1-z) (full resolution), gh = gout*z (w.r.t. miu(a2*h2+b2)), gz = gout*(h - hp) (w.r.t. the normalised gate)
__global__ void blend_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ ht,
                                 const float* __restrict__ ht_ab, int ht_lowres, const float* __restrict__ h2,
                                 const float* __restrict__ h2_ab, const float* __restrict__ zg,
                                 const float* __restrict__ mnmx, float* __restrict__ ghp, float* __restrict__ gh,
                                 float* __restrict__ gz, int N, int H, int W, int C) {
    const long P = (long)H * W;
    const long tot = (long)N * P * C, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const int c = (int)(i % C);
        const long row = i / C;
        const int n = (int)(row / P);
        long srow = row;
        if (ht_lowres) {
            const int pix = (int)(row - (long)n * P);
            const int y = pix / W, x = pix - y * W;
            srow = ((long)n * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1);
        }
        float hp = ht[srow * C + c];
        if (ht_ab != nullptr) {
            const float* ab = ht_ab + (long)n * 2 * C;
            hp = mru_act(fmaf(ab[c], hp, ab[C + c]), SSC_ACT_MIU);
        }
        const float* ab2 = h2_ab + (long)n * 2 * C;
        const float h = mru_act(fmaf(ab2[c], h2[i], ab2[C + c]), SSC_ACT_MIU);
        const float mn = mnmx[(long)n * 2 * C + c], mx = mnmx[(long)n * 2 * C + C + c];
        const float z = (zg[i] - mn) / (mx - mn);
        const float q = gout[i];
        ghp[i] = q * (1.f - z);
        gh[i] = q * z;
        gz[i] = q * (h - hp);
    }
}

__global__ __launch_bounds__(256) void blend_bwd_v4_kernel(const float* __restrict__ gout, const float* __restrict__ ht,
                                                           const float* __restrict__ ht_ab, int ht_lowres,
                                                           const float* __restrict__ h2, const float* __restrict__ h2_ab,
                                                           const float* __restrict__ zg, const float* __restrict__ mnmx,
                                                           float* __restrict__ ghp, float* __restrict__ gh,
                                                           float* __restrict__ gz, int N, int H, int W, int C, PwMap m) {
    pw_rows((long)N * H * W, C / 4, m, [&](long row, int c) {
        const int n = pw_sample(row, m);
        const long srow = ht_lowres ? up_row(row, n, H, W, m) : row;
        float4 hp = ld4(ht + srow * C + c);
        if (ht_ab != nullptr) hp = miu_affine4(hp, ht_ab + (long)n * 2 * C, C, c);
        const long i = row * C + c;
        const float4 h = miu_affine4(ld4(h2 + i), h2_ab + (long)n * 2 * C, C, c);
        const float4 mn = ld4(mnmx + (long)n * 2 * C + c), mx = ld4(mnmx + (long)n * 2 * C + C + c), zz = ld4(zg + i);
        const float4 z = PW_EACH((zz.x - mn.x) / (mx.x - mn.x), (zz.y - mn.y) / (mx.y - mn.y), (zz.z - mn.z) / (mx.z - mn.z),
                                 (zz.w - mn.w) / (mx.w - mn.w));
        const float4 q = ld4(gout + i);
        st4(ghp + i, PW_EACH(q.x * (1.f - z.x), q.y * (1.f - z.y), q.z * (1.f - z.z), q.w * (1.f - z.w)));
        st4(gh + i, PW_EACH(q.x * z.x, q.y * z.y, q.z * z.z, q.w * z.w));
        st4(gz + i, PW_EACH(q.x * (h.x - hp.x), q.y * (h.y - hp.y), q.z * (h.z - hp.z), q.w * (h.w - hp.w)));
    });
}

extern "C" int ssc_mru_blend_backward(const float* gout, const float* ht, const float* ht_ab, int ht_lowres,
                                      const float* h2, const float* h2_ab, const float* zg, const float* mnmx,
                                      float* ghp, float* gh, float* gz, int N, int H, int W, int C, void* stream) {
    if ((C & 3) == 0 && al16(gout) && al16(ht) && al16(h2) && al16(h2_ab) && al16(zg) && al16(mnmx) && al16(ghp) && al16(gh) &&
        al16(gz) && (ht_ab == nullptr || al16(ht_ab)) && pw_ok((long)N * H * W, (long)H * W, W)) {
        const PwMap m = pw_map(C / 4, (long)H * W, W);
        hipLaunchKernelGGL(blend_bwd_v4_kernel, dim3(pw_blocks((long)N * H * W, m)), dim3(256), 0, (hipStream_t)stream, gout, ht,
                           ht_ab, ht_lowres, h2, h2_ab, zg, mnmx, ghp, gh, gz, N, H, W, C, m);
        return CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(grid_for((long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, gout,
                       ht, ht_ab, ht_lowres, h2, h2_ab, zg, mnmx, ghp, gh, gz, N, H, W, C);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ [rg*up(ht) | ...] backward
// G[:, 0:C] holds d loss / d (r * up(ht)):  gr = G * up(ht);  G[:, 0:C] <- G * r   (in place)
__global__ void in2_gate_bwd_kernel(float* __restrict__ G, int ldG, const float* __restrict__ rg,
                                    const float* __restrict__ mnmx, const float* __restrict__ ht_low,
                                    float* __restrict__ gr, int N, int H, int W, int C) {
    const long P = (long)H * W;
    const long tot = (long)N * P * C, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const int c = (int)(i % C);
        const long row = i / C;
        const int n = (int)(row / P);
        const int pix = (int)(row - (long)n * P);
        const int y = pix / W, x = pix - y * W;
        const long srow = ((long)n * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1);
        const float mn = mnmx[(long)n * 2 * C + c], mx = mnmx[(long)n * 2 * C + C + c];
        const float q = G[row * ldG + c];
        gr[i] = q * ht_low[srow * C + c];
        G[row * ldG + c] = q * ((rg[i] - mn) / (mx - mn));
    }
}

__global__ __launch_bounds__(256) void in2_gate_bwd_v4_kernel(float* __restrict__ G, int ldG, const float* __restrict__ rg,
                                                              const float* __restrict__ mnmx,
                                                              const float* __restrict__ ht_low, float* __restrict__ gr, int N,
                                                              int H, int W, int C, PwMap m) {
    pw_rows((long)N * H * W, C / 4, m, [&](long row, int c) {
        const int n = pw_sample(row, m);
        const long srow = up_row(row, n, H, W, m);
        const float4 mn = ld4(mnmx + (long)n * 2 * C + c), mx = ld4(mnmx + (long)n * 2 * C + C + c);
        const long i = row * C + c;
        const float4 q = ld4(G + row * ldG + c), hl = ld4(ht_low + srow * C + c), r = ld4(rg + i);
        st4(gr + i, PW_EACH(q.x * hl.x, q.y * hl.y, q.z * hl.z, q.w * hl.w));
        st4(G + row * ldG + c, PW_EACH(q.x * ((r.x - mn.x) / (mx.x - mn.x)), q.y * ((r.y - mn.y) / (mx.y - mn.y)),
                                       q.z * ((r.z - mn.z) / (mx.z - mn.z)), q.w * ((r.w - mn.w) / (mx.w - mn.w))));
    });
}

extern "C" int ssc_mru_in2_gate_backward(float* G, int ldG, const float* rg, const float* mnmx, const float* ht_low,
                                         float* gr, int N, int H, int W, int C, void* stream) {
    if ((C & 3) == 0 && (ldG & 3) == 0 && al16(G) && al16(rg) && al16(mnmx) && al16(ht_low) && al16(gr) &&
        pw_ok((long)N * H * W, (long)H * W, W)) {
        const PwMap m = pw_map(C / 4, (long)H * W, W);
        hipLaunchKernelGGL(in2_gate_bwd_v4_kernel, dim3(pw_blocks((long)N * H * W, m)), dim3(256), 0, (hipStream_t)stream, G, ldG,
                           rg, mnmx, ht_low, gr, N, H, W, C, m);
        return CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(in2_gate_bwd_kernel, dim3(grid_for((long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, G,
                       ldG, rg, mnmx, ht_low, gr, N, H, W, C);
    return CHECK_LAUNCH();
}
