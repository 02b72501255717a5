// bn_bwd.h -- the pointwise half of the backward pass of  act(batch-statistics norm(x))  (models_collection.py:36-46 under
// tf.gradients), used by the stand-alone kernels of elementwise.hip:
//
//   dx = a * (dz - mean(dz) - xhat * mean(dz * xhat)),   dz = g1 * act1'(z) [+ g2 * act2'(z)],   z = a x + b
//
// is one streaming pass over x, g1 (g2) -> dx once the two per-channel means are known (coef).  (Round 4 also let the filter-
// gradient launch of the layer above carry it as a side job; measured slower in the step, removed in round 5 -- profiles/
// NOTEBOOK_r04.md.)
#pragma once
#include <hip/hip_runtime.h>
#include "sketchycolor_hip.h"

struct BnBwdArgs {
    const float* x; long M; int C; int ldx;
    const float* ab; const float* stats;
    const float* g1; int ldg1; int act1;
    const float* g2; int ldg2; int act2;
    int has_bn;
    const float* rowb; float rowb_scale; int rowb_P;    // g1[r][c] += rowb[r / rowb_P][c] * rowb_scale (NULL: nothing added)
};

static inline BnBwdArgs bn_args_of(const ssc_bn_apply_job& j) {
    BnBwdArgs a;
    a.x = j.x; a.M = (long)j.M; a.C = j.C; a.ldx = j.ldx; a.ab = j.ab; a.stats = j.stats;
    a.g1 = j.g1; a.ldg1 = j.ldg1; a.act1 = j.act1; a.g2 = j.g2; a.ldg2 = j.ldg2; a.act2 = j.act2; a.has_bn = j.has_bn;
    a.rowb = j.rowb; a.rowb_scale = j.rowb_scale; a.rowb_P = j.rowb_P;
    return a;
}
static inline bool bn_job_ok(const ssc_bn_apply_job& j) {
    return !((j.C & 3) || (j.ldx & 3) || (j.ldg1 & 3) || (j.lddx & 3) || (j.g2 != nullptr && (j.ldg2 & 3)) ||
             (j.rowb != nullptr && j.rowb_P <= 0) || j.x == nullptr || j.g1 == nullptr || j.dx == nullptr ||
             (j.has_bn && (j.ab == nullptr || j.stats == nullptr || j.coef == nullptr)));
}
__device__ __forceinline__ float dact(float z, int act) {
    if (act == SSC_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == SSC_ACT_LRELU) return z > 0.f ? 1.f : 0.2f;
    return 1.f;
}

__device__ __forceinline__ void bn_bwd_dz(const BnBwdArgs& a, long r, int c, const float4& aa, const float4& bb,
                                          float4& xv, float4& dz) {
    xv = *reinterpret_cast<const float4*>(a.x + r * a.ldx + c);
    float4 z;
    if (a.has_bn) {
        z.x = fmaf(aa.x, xv.x, bb.x); z.y = fmaf(aa.y, xv.y, bb.y);
        z.z = fmaf(aa.z, xv.z, bb.z); z.w = fmaf(aa.w, xv.w, bb.w);
    } else {
        z = xv;
    }
    float4 g = *reinterpret_cast<const float4*>(a.g1 + r * a.ldg1 + c);
    if (a.rowb != nullptr) {        // a per-image term broadcast over the pixels (the class head's gradient through its spatial mean)
        const float4 t = *reinterpret_cast<const float4*>(a.rowb + (r / a.rowb_P) * a.C + c);
        g.x = fmaf(t.x, a.rowb_scale, g.x); g.y = fmaf(t.y, a.rowb_scale, g.y);
        g.z = fmaf(t.z, a.rowb_scale, g.z); g.w = fmaf(t.w, a.rowb_scale, g.w);
    }
    dz.x = g.x * dact(z.x, a.act1); dz.y = g.y * dact(z.y, a.act1);
    dz.z = g.z * dact(z.z, a.act1); dz.w = g.w * dact(z.w, a.act1);
    if (a.g2 != nullptr) {
        const float4 h = *reinterpret_cast<const float4*>(a.g2 + r * a.ldg2 + c);
        dz.x += h.x * dact(z.x, a.act2); dz.y += h.y * dact(z.y, a.act2);
        dz.z += h.z * dact(z.z, a.act2); dz.w += h.w * dact(z.w, a.act2);
    }
}

// one 16-byte group (row r, channels c..c+3) of the apply pass
__device__ __forceinline__ float4 bn_bwd_apply_one(const BnBwdArgs& a, const float* __restrict__ coef, long r, int c) {
    float4 aa = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.has_bn) {
        aa = *reinterpret_cast<const float4*>(a.ab + c);
        bb = *reinterpret_cast<const float4*>(a.ab + a.C + c);
    }
    float4 xv, dz;
    bn_bwd_dz(a, r, c, aa, bb, xv, dz);
    float4 o = dz;
    if (a.has_bn) {
        const float4 mu = *reinterpret_cast<const float4*>(a.stats + c);
        const float4 rs = *reinterpret_cast<const float4*>(a.stats + a.C + c);
        const float4 c1 = *reinterpret_cast<const float4*>(coef + c);
        const float4 c2 = *reinterpret_cast<const float4*>(coef + a.C + c);
        o.x = aa.x * (dz.x - c1.x - (xv.x - mu.x) * rs.x * c2.x);
        o.y = aa.y * (dz.y - c1.y - (xv.y - mu.y) * rs.y * c2.y);
        o.z = aa.z * (dz.z - c1.z - (xv.z - mu.z) * rs.z * c2.z);
        o.w = aa.w * (dz.w - c1.w - (xv.w - mu.w) * rs.w * c2.w);
    }
    return o;
}

// The apply pass as `nblocks` workgroups of 256 threads, this one being `block`.
// Fast path (C / 4 column groups divide 256 -- every channel count of these models --, no per-image term): a thread
// keeps ONE column group for the whole walk, so its twelve per-channel constants are loaded and folded once --
//     dx = a (dz - c1 - (x - mu) rs c2) = dz a + (k1 x + k0),   k1 = -a rs c2,  k0 = -a c1 - k1 mu,
//     dz a = g1 (z > 0 ? a : a s1) [+ g2 (z > 0 ? a : a s2)],   z = a x + b
// -- five vector-ALU instructions per element instead of ~30 and no integer division: fp32 MFMA issues on the same lanes as
// the vector ALU, so what a hosted pass costs its host is exactly its vector-ALU work.  Four rows in flight per thread (a
// hosted pass has one workgroup per CU).  The general path keeps the one-group-at-a-time form.  A stand-alone launch and a
// hosted one of the same site take the same path whenever their workgroup counts both divide evenly, and the arithmetic per
// element does not depend on the workgroup count: same bits.
__device__ __forceinline__ float4 bn_bwd_fast_one(const float4& xv, const float4& g1, const float4& g2, bool two, bool has_bn,
                                                   const float4& a, const float4& b, const float4& a1, const float4& a2,
                                                   const float4& k1, const float4& k0) {
    float4 o;
#define SSC_BWD_LANE(f)                                                        \
    {                                                                          \
        const float z = has_bn ? fmaf(a.f, xv.f, b.f) : xv.f;                  \
        const float m1 = z > 0.f ? a.f : a1.f;                                 \
        float t = has_bn ? fmaf(k1.f, xv.f, k0.f) : 0.f;                       \
        if (two) t = fmaf(g2.f, z > 0.f ? a.f : a2.f, t);                      \
        o.f = fmaf(g1.f, m1, t);                                               \
    }
    SSC_BWD_LANE(x) SSC_BWD_LANE(y) SSC_BWD_LANE(z) SSC_BWD_LANE(w)
#undef SSC_BWD_LANE
    return o;
}

__device__ __forceinline__ float bn_act_neg_slope(int act) {
    return act == SSC_ACT_RELU ? 0.f : (act == SSC_ACT_LRELU ? 0.2f : 1.f);
}

__device__ __forceinline__ void bn_bwd_apply_blocks(const BnBwdArgs& a, const float* __restrict__ coef, float* __restrict__ dx,
                                                    int lddx, int block, int nblocks) {
    const int cg = a.C / 4;
    const int per_sweep = nblocks * 256;
    if (a.rowb == nullptr && (256 % cg) == 0) {      // (independent of the workgroup count: hosted and stand-alone agree)
        const int i0 = block * 256 + (int)threadIdx.x;
        const int c = (i0 % cg) * 4;
        const long rstep = per_sweep / cg;
        long r = i0 / cg;
        const bool two = a.g2 != nullptr, has_bn = a.has_bn != 0;
        float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f), k1 = bv, k0 = bv;
        if (has_bn) {
            av = *reinterpret_cast<const float4*>(a.ab + c);
            bv = *reinterpret_cast<const float4*>(a.ab + a.C + c);
            const float4 mu = *reinterpret_cast<const float4*>(a.stats + c);
            const float4 rs = *reinterpret_cast<const float4*>(a.stats + a.C + c);
            const float4 c1 = *reinterpret_cast<const float4*>(coef + c);
            const float4 c2 = *reinterpret_cast<const float4*>(coef + a.C + c);
            k1 = make_float4(-av.x * rs.x * c2.x, -av.y * rs.y * c2.y, -av.z * rs.z * c2.z, -av.w * rs.w * c2.w);
            k0 = make_float4(-av.x * c1.x - k1.x * mu.x, -av.y * c1.y - k1.y * mu.y, -av.z * c1.z - k1.z * mu.z,
                             -av.w * c1.w - k1.w * mu.w);
        }
        const float s1 = bn_act_neg_slope(a.act1), s2 = bn_act_neg_slope(a.act2);
        const float4 a1 = make_float4(av.x * s1, av.y * s1, av.z * s1, av.w * s1);
        const float4 a2 = make_float4(av.x * s2, av.y * s2, av.z * s2, av.w * s2);
        const float* xp = a.x + c;
        const float* g1p = a.g1 + c;
        const float* g2p = two ? a.g2 + c : a.g1 + c;
        float* dp = dx + c;
        for (; r + 3 * rstep < a.M; r += 4 * rstep) {
            float4 xv[4], gv[4], hv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long rr = r + u * rstep;
                xv[u] = *reinterpret_cast<const float4*>(xp + rr * a.ldx);
                gv[u] = *reinterpret_cast<const float4*>(g1p + rr * a.ldg1);
                if (two) hv[u] = *reinterpret_cast<const float4*>(g2p + rr * a.ldg2);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                *reinterpret_cast<float4*>(dp + (r + u * rstep) * lddx) =
                    bn_bwd_fast_one(xv[u], gv[u], hv[u], two, has_bn, av, bv, a1, a2, k1, k0);
        }
        for (; r < a.M; r += rstep) {
            const float4 xv = *reinterpret_cast<const float4*>(xp + r * a.ldx);
            const float4 gv = *reinterpret_cast<const float4*>(g1p + r * a.ldg1);
            float4 hv = gv;
            if (two) hv = *reinterpret_cast<const float4*>(g2p + r * a.ldg2);
            *reinterpret_cast<float4*>(dp + r * lddx) = bn_bwd_fast_one(xv, gv, hv, two, has_bn, av, bv, a1, a2, k1, k0);
        }
        return;
    }
    const long tot = a.M * cg;
    const long stride = (long)per_sweep;
    for (long i = (long)block * 256 + threadIdx.x; i < tot; i += stride) {
        const long r = i / cg;
        const int c = (int)(i - r * cg) * 4;
        *reinterpret_cast<float4*>(dx + r * lddx + c) = bn_bwd_apply_one(a, coef, r, c);
    }
}
