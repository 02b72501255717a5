// igemm_epilogue.h -- the epilogue shared by the uniform-tap implicit-GEMM kernels (conv_ut_kernel in igemm.hip: exact fp32 on
// v_mfma_f32_32x32x2_f32; conv_bf_kernel in igemm_bf16.hip: 3-way bf16 split on v_mfma_f32_32x32x16_bf16).  Both produce the
// same 32x32 accumulator layout (row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), column = lane & 31), so everything behind the
// K loop is one piece of code: the in-launch K-slice hand-off of the tail split, the vector store through LDS with bias /
// activation / accumulate, the batch-statistics sums, the norm-backward sums, the extrema rows, the in-launch fold.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sketchycolor_hip.h"
#include "igemm_util.h"

// 8-byte write-through (agent-scope relaxed atomic) stores: data another workgroup of the same launch will read
__device__ __forceinline__ void st_agent2(float* p, float a, float b) {
    union { float f[2]; unsigned long long u; } cv;
    cv.f[0] = a; cv.f[1] = b;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), cv.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_d(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

struct FwdPhase {
    int ioff_y, ioff_x, ky0, kx0, ooff_y, ooff_x;
};

__device__ __forceinline__ FwdPhase fwd_phase(const ssc_conv_desc& d, int phase) {
    FwdPhase p;
    if (d.nphase == 4) {   // stride-2 transposed conv, k=4, pad 1: output parity (ry,rx)
        const int ry = phase >> 1, rx = phase & 1;
        p.ioff_y = ry - 1; p.ioff_x = rx - 1;
        p.ky0 = 3 - ry;    p.kx0 = 3 - rx;
        p.ooff_y = ry;     p.ooff_x = rx;
    } else {
        p.ioff_y = d.ioff_y; p.ioff_x = d.ioff_x;
        p.ky0 = d.ky0;       p.kx0 = d.kx0;
        p.ooff_y = d.ooff_y; p.ooff_x = d.ooff_x;
    }
    return p;
}

// Workgroup -> (tile, K range) of the uniform-tap kernels (see conv_ut_kernel for the layouts): legacy 3-D grid, or the 1-D
// grid of the tail split / XCD-aware order.
struct UtTile {
    int phase, ks, sk, n0, slot;
    long m0;
    float* part;
};

// LDS_FLOATS: floats of the operand buffers (free after the K loop's last barrier; the C image and the reduction scratch live
// there); rowpix [BM] sits behind them.  skcfg: the translation unit's copy of the hand-off configuration (ssc_sk_configure).
template <int BM, int BN, int WM, int WN, int SM, int SN, int LDS_FLOATS>
__device__ __forceinline__ void ut_epilogue(f32x16 (&acc)[SM][SN], const ssc_conv_desc& d, float* smem, const long* rowpix,
                                            float* part, int ks, int sk, int slot, unsigned* __restrict__ flags, int ts_s,
                                            int splitk, float* __restrict__ slab_base, long slab_stride, long m0, int n0,
                                            int phase, long M, const unsigned* skcfg) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    if (part != nullptr) {      // tail split
        if (ks != sk - 1) {
            // producer: raw partial tile as [pair of accumulator entries][thread], 8-byte write-through (sc1) stores
#pragma unroll
            for (int i = 0; i < SM; ++i)
#pragma unroll
                for (int j = 0; j < SN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int e2 = ((i * SN + j) * 16 + r) >> 1;
                        union { float f[2]; unsigned long long u; } cv;
                        cv.f[0] = acc[i][j][r]; cv.f[1] = acc[i][j][r + 1];
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(part + ((long)e2 * 256 + tid) * 2), cv.u,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its stores
            __syncthreads();
            if (tid == 0 && skcfg[2] == 0u) __hip_atomic_store(flags + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // owner: the other slices of this tile are workgroups slot-(sk-1) .. slot-1.  The wait is bounded (a deadlock guard:
        // everything waited for was dispatched earlier); an owner that gives up REPORTS it -- the timeout word gets
        // 0x80000000 | sk_tag, which the host must read wherever it reads results (hip.check_sk) -- because what it then
        // stores is a partial sum.  A flag that was never seen set is not cleared (its late producer would otherwise leave
        // a 1 behind for the next launch to trust); the host zeroes the array after a reported timeout.
        if (tid == 0) {
            const unsigned long long t0 = wall_clock64();
            const unsigned long long bound = ((unsigned long long)skcfg[1] << 32) | skcfg[0];
            bool gave_up = false;
            for (int q = sk - 1; q >= 1; --q) {
                bool seen = true;
                while (__hip_atomic_load(flags + slot - q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    __builtin_amdgcn_s_sleep(2);
                    if (gave_up || wall_clock64() - t0 > bound) {
                        gave_up = true;
                        seen = false;
                        break;
                    }
                }
                if (seen) __hip_atomic_store(flags + slot - q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // leave them zero
            }
            if (gave_up)
                __hip_atomic_store(flags + (SSC_SK_FLAG_WORDS - 1), 0x80000000u | (unsigned)d.sk_tag, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // one L1 invalidate after the last flag
        }
        __syncthreads();
#pragma nounroll
        for (int q = sk - 1; q >= 1; --q) {
            const float* pp = part - (long)q * (BM * BN);
#pragma unroll
            for (int i = 0; i < SM; ++i)
#pragma unroll
                for (int j = 0; j < SN; ++j) {
                    float2 pv[8];       // one accumulator's worth of loads in flight
#pragma unroll
                    for (int r2 = 0; r2 < 8; ++r2)
                        pv[r2] = *reinterpret_cast<const float2*>(pp + ((long)((i * SN + j) * 8 + r2) * 256 + tid) * 2);
#pragma unroll
                    for (int r2 = 0; r2 < 8; ++r2) {
                        acc[i][j][2 * r2] += pv[r2].x;
                        acc[i][j][2 * r2 + 1] += pv[r2].y;
                    }
                }
        }
    }
    float* outp = (ts_s == 0 && splitk > 1) ? (slab_base + (long)ks * slab_stride) : d.out;
    const bool final_pass = (ts_s > 0) || (splitk == 1);
    // Vector epilogue: the accumulators go through LDS (the tile buffers are free after the loop's last barrier) and leave
    // as one 16-byte store per thread and 4 columns -- 8 stores per thread instead of 32 with a 64-bit address, a row test
    // and a column test each.  Every workgroup of a round reaches its epilogue at about the same time, so the epilogue's
    // length is matrix-pipe idle time.  Needs 16-byte aligned rows (the scalar form below covers the rest).
    constexpr int C_LD = BN + 4;
    static_assert(BM * C_LD <= LDS_FLOATS, "the C tile fits the operand buffers");
    if ((((d.Nstore | d.ldc) & 3) == 0) & ((reinterpret_cast<unsigned long>(outp) & 15) == 0)) {
        float* Cs = smem;
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
            for (int j = 0; j < SN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * SM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    Cs[row * C_LD + wn * SN * 32 + j * 32 + l31] = acc[i][j][r];
                }
        __syncthreads();
        const float* const bias = final_pass ? d.bias : nullptr;
        const int epi = final_pass ? d.epi : 0;
        const bool accum = final_pass && d.accumulate != 0;
        const int Nn = d.Nn, Nst = d.Nstore, ldc = d.ldc;
        // batch-statistics norm of this layer's output (models_collection.py:36-46): per-column sum and sum of squares of the
        // tile, taken here from the values on their way out instead of by a second pass over the tensor; one row of
        // partials per row tile, folded per channel by bn_stats_finalize (ssc_conv_forward_bn)
        float* const stat = final_pass ? d.stat_partial : nullptr;
        const bool mmode = d.stat_mode == 1;        // rows of per-column minimum / maximum instead of sum / sum of squares
        float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = ssum;
        if (mmode) {
            ssum = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
            ssq = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
        // ... or, when the output is the gradient w.r.t. the activated norm of a tensor x (sb_x: same layout as the output),
        // the two sums of that norm's backward: sum dz and sum dz * xhat with dz = out * act'(a x + b) (ssc_conv_forward_bnbwd);
        // the column group of a thread is the same in every pass of the loop below (256 % (BN / 4) == 0)
        // two normed tensors side by side (ssc_conv_forward_bnbwd2): this workgroup's column tile lies in one of them; cb = its
        // first column, sbC = its channel count (the width of its tables and of its rows of sums)
        const bool sb_two = (stat != nullptr) && d.sb2_x != nullptr;
        const bool sb_second = sb_two && n0 >= d.sb2_col0;
        const int cb = sb_second ? d.sb2_col0 : 0;
        const int sbC = sb_two ? (sb_second ? Nst - d.sb2_col0 : d.sb2_col0) : Nst;
        float* const statw = sb_second ? d.stat_partial2 : stat;
        const float* const sbx = (stat != nullptr) ? (sb_second ? d.sb2_x : d.sb_x) : nullptr;
        const int sb_ldx = sb_second ? d.sb2_ldx : d.sb_ldx;
        float4 sb_a = make_float4(1.f, 1.f, 1.f, 1.f), sb_b = make_float4(0.f, 0.f, 0.f, 0.f), sb_mu = sb_b, sb_rs = sb_a;
        float sb_neg = 1.f;         // act'(z) for z <= 0
        if (sbx != nullptr) {
            const int c = n0 + (tid % (BN / 4)) * 4;
            if (c < Nst) {
                const float* const tab = sb_second ? d.sb2_ab : d.sb_ab;
                const float* const tst = sb_second ? d.sb2_stats : d.sb_stats;
                sb_a = *reinterpret_cast<const float4*>(tab + (c - cb));
                sb_b = *reinterpret_cast<const float4*>(tab + sbC + (c - cb));
                sb_mu = *reinterpret_cast<const float4*>(tst + (c - cb));
                sb_rs = *reinterpret_cast<const float4*>(tst + sbC + (c - cb));
            }
            const int sact = sb_second ? d.sb2_act : d.sb_act;
            sb_neg = sact == SSC_ACT_RELU ? 0.f : (sact == SSC_ACT_LRELU ? 0.2f : 1.f);
        }
#pragma unroll
        for (int p = 0; p < BM * BN / 1024; ++p) {
            const int e = p * 256 + tid;
            const int row = e / (BN / 4), col = n0 + (e % (BN / 4)) * 4;
            if ((m0 + row < M) & (col < Nst)) {
                float4 v = *reinterpret_cast<const float4*>(Cs + row * C_LD + (col - n0));
                // the filter loads of columns >= Nn were not masked
                v.x = col + 0 < Nn ? v.x : 0.f; v.y = col + 1 < Nn ? v.y : 0.f;
                v.z = col + 2 < Nn ? v.z : 0.f; v.w = col + 3 < Nn ? v.w : 0.f;
                if (bias != nullptr) {
                    v.x += col + 0 < Nn ? bias[col + 0] : 0.f; v.y += col + 1 < Nn ? bias[col + 1] : 0.f;
                    v.z += col + 2 < Nn ? bias[col + 2] : 0.f; v.w += col + 3 < Nn ? bias[col + 3] : 0.f;
                }
                if (mmode) {        // of the ACTIVATED output (lrelu is monotone: applied to the extrema's candidates here)
                    const float tx = epi == 2 ? fmaxf(v.x, 0.2f * v.x) : v.x, ty = epi == 2 ? fmaxf(v.y, 0.2f * v.y) : v.y;
                    const float tz = epi == 2 ? fmaxf(v.z, 0.2f * v.z) : v.z, tw = epi == 2 ? fmaxf(v.w, 0.2f * v.w) : v.w;
                    ssum.x = fminf(ssum.x, tx); ssum.y = fminf(ssum.y, ty); ssum.z = fminf(ssum.z, tz); ssum.w = fminf(ssum.w, tw);
                    ssq.x = fmaxf(ssq.x, tx); ssq.y = fmaxf(ssq.y, ty); ssq.z = fmaxf(ssq.z, tz); ssq.w = fmaxf(ssq.w, tw);
                } else if (sbx == nullptr) {
                    ssum.x += v.x; ssum.y += v.y; ssum.z += v.z; ssum.w += v.w;
                    ssq.x += v.x * v.x; ssq.y += v.y * v.y; ssq.z += v.z * v.z; ssq.w += v.w * v.w;
                } else {
                    const float4 xv = *reinterpret_cast<const float4*>(sbx + rowpix[row] * sb_ldx + (col - cb));
                    float4 dz;
                    dz.x = v.x * (fmaf(sb_a.x, xv.x, sb_b.x) > 0.f ? 1.f : sb_neg);
                    dz.y = v.y * (fmaf(sb_a.y, xv.y, sb_b.y) > 0.f ? 1.f : sb_neg);
                    dz.z = v.z * (fmaf(sb_a.z, xv.z, sb_b.z) > 0.f ? 1.f : sb_neg);
                    dz.w = v.w * (fmaf(sb_a.w, xv.w, sb_b.w) > 0.f ? 1.f : sb_neg);
                    ssum.x += dz.x; ssum.y += dz.y; ssum.z += dz.z; ssum.w += dz.w;
                    ssq.x += dz.x * (xv.x - sb_mu.x) * sb_rs.x; ssq.y += dz.y * (xv.y - sb_mu.y) * sb_rs.y;
                    ssq.z += dz.z * (xv.z - sb_mu.z) * sb_rs.z; ssq.w += dz.w * (xv.w - sb_mu.w) * sb_rs.w;
                }
                if (epi == 1) {
                    v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
                } else if (epi == 2) {
                    v.x = fmaxf(v.x, 0.2f * v.x); v.y = fmaxf(v.y, 0.2f * v.y);
                    v.z = fmaxf(v.z, 0.2f * v.z); v.w = fmaxf(v.w, 0.2f * v.w);
                }
                float4* o = reinterpret_cast<float4*>(outp + rowpix[row] * ldc + col);
                if (accum) {
                    const float4 t = *o;
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                *o = v;
            }
        }
        if (stat != nullptr) {
            // thread t holds columns 4*(t % (BN/4)).. of the rows t / (BN/4) + k*1024/BN: fold the 1024/BN row groups in a
            // fixed order through LDS (behind the C tile), one writer per column group
            constexpr int CG = BN / 4, RG = 256 / CG;
            // (the 128 x 128 tile's C image leaves no room behind it: there the scratch takes the image's place, once every
            // thread has read its part of it)
            constexpr bool RED_IN_C = BM * C_LD + 2 * 256 * 4 > LDS_FLOATS;
            static_assert(!RED_IN_C || BM * BN >= 128 * 128, "reduction scratch behind the C tile");
            if (RED_IN_C) __syncthreads();
            float4* red = reinterpret_cast<float4*>(RED_IN_C ? smem : smem + BM * C_LD);
            red[tid] = ssum;
            red[256 + tid] = ssq;
            __syncthreads();
            if (tid < CG) {
                float4 s = red[tid], q = red[256 + tid];
#pragma unroll
                for (int g = 1; g < RG; ++g) {
                    const float4 a = red[g * CG + tid], c = red[256 + g * CG + tid];
                    if (mmode) {
                        s.x = fminf(s.x, a.x); s.y = fminf(s.y, a.y); s.z = fminf(s.z, a.z); s.w = fminf(s.w, a.w);
                        q.x = fmaxf(q.x, c.x); q.y = fmaxf(q.y, c.y); q.z = fmaxf(q.z, c.z); q.w = fmaxf(q.w, c.w);
                    } else {
                        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
                        q.x += c.x; q.y += c.y; q.z += c.z; q.w += c.w;
                    }
                }
                const int col = n0 + tid * 4;
                if (col < Nst) {
                    const long blk = (long)phase * ((M + BM - 1) / BM) + m0 / BM;
                    float* sp = statw + blk * 2 * sbC;
                    *reinterpret_cast<float4*>(sp + (col - cb)) = s;
                    *reinterpret_cast<float4*>(sp + sbC + (col - cb)) = q;
                }
            }
        }
        return;
    }
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
                float v = col < d.Nn ? acc[i][j][r] : 0.f;    // the filter loads of columns >= Nn were not masked
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
