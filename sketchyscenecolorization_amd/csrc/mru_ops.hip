// mru_ops.hip -- HBM-bound pointwise / reduction kernels of the MRU blocks (mru.py:353-461, 527-591):
// conditional batch-norm folding, miu_relu, min-max normalised gates, the gated merges, 2x2 mean-pool,
// nearest 2x upsample fused into the channel-concat writer.  NHWC fp32; every kernel is a grid-stride
// loop over (row, channel) with channels fastest, so a wavefront reads/writes contiguous 256-byte runs.
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

extern "C" int ssc_mean_pool2(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int C, void* stream) {
    if ((H | W) & 1) return -1;
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

__global__ void minmax_final_kernel(const float* __restrict__ part, int nsplit, int N, int C, float* __restrict__ mnmx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    float mn = INFINITY, mx = -INFINITY;
    for (int s = 0; s < nsplit; ++s) {
        const float* p = part + (((long)n * nsplit + s) * 2) * C + c;
        mn = fminf(mn, p[0]);
        mx = fmaxf(mx, p[C]);
    }
    mnmx[(long)n * 2 * C + c] = mn;
    mnmx[(long)n * 2 * C + C + c] = mx;
}

extern "C" int ssc_minmax_hw(const float* x, int ld, int N, int P, int C, float* mnmx, float* workspace,
                             int64_t workspace_bytes, void* stream) {
    int nsplit = (P + 255) / 256;
    if (nsplit > 64) nsplit = 64;
    if ((int64_t)N * nsplit * 2 * C * 4 > workspace_bytes) return -2;
    hipLaunchKernelGGL(minmax_partial_kernel, dim3((C + 63) / 64, nsplit, N), dim3(256), 0, (hipStream_t)stream, x, ld,
                       P, C, nsplit, workspace);
    hipLaunchKernelGGL(minmax_final_kernel, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace,
                       nsplit, N, C, mnmx);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ channel-concat writer
__global__ void concat_parts_kernel(ssc_cat_desc d) {
    const int c0 = d.p[0].C, c1 = c0 + (d.nparts > 1 ? d.p[1].C : 0), ct = c1 + (d.nparts > 2 ? d.p[2].C : 0);
    const long P = (long)d.H * d.W;
    const long tot = (long)d.N * P * ct, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const int col = (int)(i % ct);
        const long row = i / ct;
        const int k = col < c0 ? 0 : (col < c1 ? 1 : 2);
        const int c = col - (k == 0 ? 0 : (k == 1 ? c0 : c1));
        const ssc_cat_part& q = d.p[k];
        const int n = (int)(row / P);
        long srow = row;
        if (q.upsample) {
            const int pix = (int)(row - (long)n * P);
            const int y = pix / d.W, x = pix - y * d.W;
            srow = ((long)n * (d.H / 2) + (y >> 1)) * (d.W / 2) + (x >> 1);
        }
        float v = q.x[srow * q.ld + c];
        if (q.ab != nullptr) {
            const float* ab = q.ab + (long)n * q.ab_sample_stride;
            v = fmaf(ab[c], v, ab[q.C + c]);
        }
        v = mru_act(v, q.act);
        if (q.gate != nullptr) {
            const float mn = q.mnmx[(long)n * 2 * q.C + c], mx = q.mnmx[(long)n * 2 * q.C + q.C + c];
            v *= (q.gate[row * q.C + c] - mn) / (mx - mn);
        }
        d.out[row * d.ldo + col] = v;
    }
}

extern "C" int ssc_concat_parts(const ssc_cat_desc* desc, void* stream) {
    const ssc_cat_desc& d = *desc;
    if (d.nparts < 1 || d.nparts > 3) return -1;
    long ct = 0;
    for (int k = 0; k < d.nparts; ++k) ct += d.p[k].C;
    hipLaunchKernelGGL(concat_parts_kernel, dim3(grid_for((long)d.N * d.H * d.W * ct)), dim3(256), 0,
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

extern "C" int ssc_mru_gate_merge(const float* ht, const float* rg, const float* mnmx, const float* img, float* out,
                                  int N, int64_t P, int C, void* stream) {
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

extern "C" int ssc_mru_blend(const float* ht, const float* ht_ab, int ht_lowres, const float* h2, const float* h2_ab,
                             const float* zg, const float* mnmx, float* out, int N, int H, int W, int C, void* stream) {
    hipLaunchKernelGGL(mru_blend_kernel, dim3(grid_for((long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, ht,
                       ht_ab, ht_lowres, h2, h2_ab, zg, mnmx, out, N, H, W, C);
    return CHECK_LAUNCH();
}
