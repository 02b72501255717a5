// text_lstm.hip -- pointwise / reduction kernels of the caption branch
// (encode_feat_with_text, models_collection.py:150-248) and of the small dense heads.
//
// The two BasicLSTMCell matmuls are decomposed on the host side
// (gates = visual*Kv + (emb*Kw + lang*Kl) + h*Kh) and run
// on the implicit-GEMM kernel; this file holds what surrounds them: embedding
// lookup, l2-normalisation (wavefront reductions), the gate nonlinearities with
// the per-sample "token == 0 -> skip the step" select (tf.cond, :235), the
// atanh-like squash (:238-242) and their backward forms.  All HBM-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sketchycolor_hip.h"
#include "host_util.h"

#define CHECK_LAUNCH() ((int)hipGetLastError())

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------ embedding
__global__ void embedding_gather_kernel(const float* __restrict__ table, const int* __restrict__ tok, int rows, int C,
                                        float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = C / 4;
    if (i >= (long)rows * c4) return;
    const int r = (int)(i / c4), c = (int)(i - (long)r * c4) * 4;
    *reinterpret_cast<float4*>(out + (long)r * C + c) = *reinterpret_cast<const float4*>(table + (long)tok[r] * C + c);
}

// Deterministic form of the sparse embedding gradient: thread (v, c) of the (small) table scans the token list
// in order and adds the rows that hit vocabulary entry v; pad tokens (id 0) never reach the table because
// tf.cond skips the lookup's consumer.  blockIdx.y = vocabulary row.
__global__ void embedding_scatter_add_kernel(float* __restrict__ dtable, const int* __restrict__ tok, int rows, int C,
                                             const float* __restrict__ g) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y;
    if (c >= C || v == 0) return;
    float acc = 0.f;
    for (int r = 0; r < rows; ++r)
        if (tok[r] == v) acc += g[(long)r * C + c];
    dtable[(long)v * C + c] += acc;
}

extern "C" int ssc_embedding_gather(const float* table, const int* tok, int rows, int C, float* out, void* stream) {
    if (C & 3) return -1;
    const long tot = (long)rows * (C / 4);
    hipLaunchKernelGGL(embedding_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       table, tok, rows, C, out);
    return CHECK_LAUNCH();
}

extern "C" int ssc_embedding_scatter_add(float* dtable, int vocab, const int* tok, int rows, int C, const float* g,
                                         void* stream) {
    hipLaunchKernelGGL(embedding_scatter_add_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)vocab), dim3(256), 0,
                       (hipStream_t)stream, dtable, tok, rows, C, g);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ row l2-normalise (tf.nn.l2_normalize)
// one wavefront per row; z = a*x+b (folded norm) when ab != NULL
__global__ __launch_bounds__(256) void row_l2norm_fwd_kernel(const float* __restrict__ x, int ldx,
                                                              const float* __restrict__ ab, long M, int C,
                                                              float* __restrict__ y, float* __restrict__ ss) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(x + row * ldx + c);
        if (ab != nullptr) {
            const float4 a = *reinterpret_cast<const float4*>(ab + c), b = *reinterpret_cast<const float4*>(ab + C + c);
            v.x = fmaf(a.x, v.x, b.x); v.y = fmaf(a.y, v.y, b.y); v.z = fmaf(a.z, v.z, b.z); v.w = fmaf(a.w, v.w, b.w);
        }
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = wave_sum(s);
    const float r = rsqrtf(fmaxf(s, 1e-12f));
    for (int c = lane * 4; c < C; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(x + row * ldx + c);
        if (ab != nullptr) {
            const float4 a = *reinterpret_cast<const float4*>(ab + c), b = *reinterpret_cast<const float4*>(ab + C + c);
            v.x = fmaf(a.x, v.x, b.x); v.y = fmaf(a.y, v.y, b.y); v.z = fmaf(a.z, v.z, b.z); v.w = fmaf(a.w, v.w, b.w);
        }
        v.x *= r; v.y *= r; v.z *= r; v.w *= r;
        *reinterpret_cast<float4*>(y + row * C + c) = v;
    }
    if (lane == 0) ss[row] = s;
}

// dz = r*(dy - y*(dy.y)) when the sum of squares was not clamped, else r*dy
__global__ __launch_bounds__(256) void row_l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ ss,
                                                              const float* __restrict__ dy, long M, int C,
                                                              float* __restrict__ dz, int accumulate) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float dot = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(y + row * C + c);
        const float4 g = *reinterpret_cast<const float4*>(dy + row * C + c);
        dot += a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
    }
    dot = wave_sum(dot);
    const float s = ss[row];
    const float r = rsqrtf(fmaxf(s, 1e-12f));
    if (s < 1e-12f) dot = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(y + row * C + c);
        const float4 g = *reinterpret_cast<const float4*>(dy + row * C + c);
        float4 o;
        o.x = r * (g.x - a.x * dot); o.y = r * (g.y - a.y * dot);
        o.z = r * (g.z - a.z * dot); o.w = r * (g.w - a.w * dot);
        float4* p = reinterpret_cast<float4*>(dz + row * C + c);
        if (accumulate) { const float4 q = *p; o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
        *p = o;
    }
}

extern "C" int ssc_row_l2norm_fwd(const float* x, int ldx, const float* ab, int64_t M, int C, float* y, float* ss,
                                  void* stream) {
    if ((C & 3) || (ldx & 3)) return -1;
    hipLaunchKernelGGL(row_l2norm_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       ab, (long)M, C, y, ss);
    return CHECK_LAUNCH();
}

extern "C" int ssc_row_l2norm_bwd(const float* y, const float* ss, const float* dy, int64_t M, int C, float* dz,
                                  int accumulate, void* stream) {
    if (C & 3) return -1;
    hipLaunchKernelGGL(row_l2norm_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, ss,
                       dy, (long)M, C, dz, accumulate);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ BasicLSTMCell pointwise
// gates[row] = g0[row] + g1[row] (opt) + g2[row / div2] (opt), split i,j,f,o (each C wide);
// c' = c*sigmoid(f+1) + sigmoid(i)*tanh(j); h' = tanh(c')*sigmoid(o).
// mask[row / mdiv] == 0 -> the step is skipped for that sample (state copied through).
__global__ void lstm_fwd_kernel(const float* __restrict__ g0, const float* __restrict__ g1,
                                const float* __restrict__ g2, int div2, const int* __restrict__ mask, int mdiv,
                                const float* __restrict__ c_in, const float* __restrict__ h_in, long rows, int C,
                                float* __restrict__ c_out, float* __restrict__ h_out, float* __restrict__ acts) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const long r = i / C;
    const int u = (int)(i - r * C);
    const float c0 = c_in[i], h0 = h_in[i];
    if (mask[r / mdiv] == 0) {
        c_out[i] = c0;
        h_out[i] = h0;
        return;     // acts are never read for skipped steps
    }
    const long gb = r * 4 * C + u;
    float gi = g0[gb], gj = g0[gb + C], gf = g0[gb + 2 * C], go = g0[gb + 3 * C];
    if (g1 != nullptr) { gi += g1[gb]; gj += g1[gb + C]; gf += g1[gb + 2 * C]; go += g1[gb + 3 * C]; }
    if (g2 != nullptr) {
        const long q = (r / div2) * 4 * C + u;
        gi += g2[q]; gj += g2[q + C]; gf += g2[q + 2 * C]; go += g2[q + 3 * C];
    }
    const float ai = sigmoidf_(gi), aj = tanhf(gj), af = sigmoidf_(gf + 1.0f), ao = sigmoidf_(go);
    const float c1 = c0 * af + ai * aj;
    c_out[i] = c1;
    h_out[i] = tanhf(c1) * ao;
    acts[gb] = ai; acts[gb + C] = aj; acts[gb + 2 * C] = af; acts[gb + 3 * C] = ao;
}

// dh/dc: gradients w.r.t. (h_out, c_out).  Writes pre-activation gate gradients dg[rows,4C],
// dc_in, and dh_pass (= dh for skipped steps, 0 otherwise; the h-part GEMM accumulates onto it).
// gacc (optional) += dg  (step-invariant addend, e.g. visual*Kv).
__global__ void lstm_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc,
                                const float* __restrict__ acts, const float* __restrict__ c_in,
                                const float* __restrict__ c_out, const int* __restrict__ mask, int mdiv, long rows,
                                int C, float* __restrict__ dg, float* __restrict__ dc_in,
                                float* __restrict__ dh_pass, float* __restrict__ gacc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const long r = i / C;
    const int u = (int)(i - r * C);
    const long gb = r * 4 * C + u;
    const float dhv = dh[i], dcv = dc[i];
    if (mask[r / mdiv] == 0) {
        dg[gb] = 0.f; dg[gb + C] = 0.f; dg[gb + 2 * C] = 0.f; dg[gb + 3 * C] = 0.f;
        dc_in[i] = dcv;
        dh_pass[i] = dhv;
        return;
    }
    const float ai = acts[gb], aj = acts[gb + C], af = acts[gb + 2 * C], ao = acts[gb + 3 * C];
    const float tc = tanhf(c_out[i]);
    const float dct = dcv + dhv * ao * (1.f - tc * tc);
    const float di = dct * aj * ai * (1.f - ai);
    const float dj = dct * ai * (1.f - aj * aj);
    const float df = dct * c_in[i] * af * (1.f - af);
    const float dO = dhv * tc * ao * (1.f - ao);
    dg[gb] = di; dg[gb + C] = dj; dg[gb + 2 * C] = df; dg[gb + 3 * C] = dO;
    dc_in[i] = dct * af;
    dh_pass[i] = 0.f;
    if (gacc != nullptr) { gacc[gb] += di; gacc[gb + C] += dj; gacc[gb + 2 * C] += df; gacc[gb + 3 * C] += dO; }
}

// ------------------------------------------------------------------ one recurrent step in one launch
// gates = h_in @ Kh + g1[row] + g2[row / div2], then the pointwise cell above: the [rows, C] x [C, 4C] product of a step
// (models_collection.py:230-236, the only sequential GEMM of the caption branch) and its gate math, which were a GEMM
// launch + split-K reduce + pointwise launch per step.  A workgroup owns 64 rows x 16 hidden units: the four 16-column
// gate slices i, j, f, o of those units form its 64-column B tile, so the gate math finds its four inputs in the
// workgroup's own accumulators (through LDS).  fp32 MFMA (v_mfma_f32_32x32x2_f32), 4 waves, one 32x32 accumulator each;
// K-tile 32, register prefetch of the next K-tile, one barrier per K-tile.
typedef float lstm_f32x16 __attribute__((ext_vector_type(16)));

// Workgroup -> (row block, unit block) of the step kernels, XCD-aware: consecutive workgroup ids go to consecutive XCDs (8 of them,
// one L2 each), so unit block ub is given to the ids with id % 8 == ub % 8 -- every XCD then reads ONE EIGHTH of K_h (which stays
// in its L2 from step to step) instead of all of it.  With the plain (row block, unit block) grid each of the 8 L2s pulled the
// whole 4-6 MB of K_h through the fabric in every one of a caption's 28 dependent steps.  Unit-block counts that are not a multiple of 8 keep the plain order.
__device__ __forceinline__ void lstm_wg_map(int id, int row_blocks, int unit_blocks, int& rb, int& ub) {
    if (unit_blocks & 7) {      // small C (tests): plain order
        ub = id / row_blocks;
        rb = id - ub * row_blocks;
        return;
    }
    const int xcd = id & 7, q = id >> 3;
    const int ul = q / row_blocks;
    rb = q - ul * row_blocks;
    ub = ul * 8 + xcd;
}

__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(const float* __restrict__ h_in, const float* __restrict__ Kh,
                                                            int ldk, const float* __restrict__ g1,
                                                            const float* __restrict__ g2, int div2,
                                                            const int* __restrict__ mask, int mdiv,
                                                            const float* __restrict__ c_in, int rows, int C, int with_gemm,
                                                            float* __restrict__ c_out, float* __restrict__ h_out,
                                                            float* __restrict__ acts) {
    constexpr int BM = 64, BK = 32, A_LD = 36, B_LD = 64, C_LD = 68;
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * A_LD + 2 * BK * B_LD];     // 34.8 KB; the C tile reuses it
    float* As = smem;
    float* Bs = smem + 2 * BM * A_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
    int rb_, ub_;
    lstm_wg_map(blockIdx.x, (rows + BM - 1) / BM, C / 16, rb_, ub_);
    const int m0 = rb_ * BM, u0 = ub_ * 16;

    lstm_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if (with_gemm) {
        // A: thread t stages rows t >> 3 and 32 + (t >> 3), 16-byte chunk t & 7 of the K-tile
        // B: k row t >> 4 (and + 16), 16-byte piece t & 15 = gate (t & 15) >> 2, units 4 * (t & 3) ..
        const int a_r = tid >> 3, a_c = (tid & 7) * 4;
        const int b_k = tid >> 4, b_g = (tid & 15) >> 2, b_u = (tid & 3) * 4;
        const float* ap0 = h_in + (long)min(m0 + a_r, rows - 1) * C + a_c;
        const float* ap1 = h_in + (long)min(m0 + a_r + 32, rows - 1) * C + a_c;
        const float* bp = Kh + (long)b_k * ldk + b_g * C + u0 + b_u;
        float4 ra0, ra1, rb0, rb1;
        auto load = [&](int kt) {
            ra0 = *reinterpret_cast<const float4*>(ap0 + kt * BK);
            ra1 = *reinterpret_cast<const float4*>(ap1 + kt * BK);
            rb0 = *reinterpret_cast<const float4*>(bp + (long)kt * BK * ldk);
            rb1 = *reinterpret_cast<const float4*>(bp + (long)(kt * BK + 16) * ldk);
        };
        auto store = [&](int buf) {
            float* A = As + buf * BM * A_LD;
            float* B = Bs + buf * BK * B_LD;
            *reinterpret_cast<float4*>(A + a_r * A_LD + a_c) = ra0;
            *reinterpret_cast<float4*>(A + (a_r + 32) * A_LD + a_c) = ra1;
            *reinterpret_cast<float4*>(B + b_k * B_LD + (tid & 15) * 4) = rb0;
            *reinterpret_cast<float4*>(B + (b_k + 16) * B_LD + (tid & 15) * 4) = rb1;
        };
        const int nkt = C / BK;
        load(0);
        store(0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt) load(kt + 1);
            // K index of MFMA step kk on lane half lhi: 16 * lhi + kk (A and B agree)
            const float* A = As + cur * BM * A_LD + (wm * 32 + l31) * A_LD + lhi * 16;
            const float* B = Bs + cur * BK * B_LD + lhi * 16 * B_LD + wn * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 av = *reinterpret_cast<const float4*>(A + g * 4);
                const float b0 = B[(g * 4 + 0) * B_LD], b1 = B[(g * 4 + 1) * B_LD], b2 = B[(g * 4 + 2) * B_LD],
                            b3 = B[(g * 4 + 3) * B_LD];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc, 0, 0, 0);
            }
            if (kt + 1 < nkt) store(cur ^ 1);
            __syncthreads();
        }
    }
    // accumulators -> LDS [row][gate * 16 + unit]
    float* Cs = smem;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * C_LD + wn * 32 + l31] = acc[r];
    __syncthreads();
    // gate math: 64 rows x 16 units, 4 per thread (a row's 16 units by 16 consecutive threads)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = p * 256 + tid;
        const int rl = e >> 4, u = e & 15;
        const long r = m0 + rl;
        if (r >= rows) continue;
        const long i = r * C + u0 + u;
        const float c0 = c_in[i], h0 = h_in[i];
        if (mask[r / mdiv] == 0) {
            c_out[i] = c0;
            h_out[i] = h0;
            continue;       // acts are never read for skipped steps
        }
        const long gb = r * 4 * C + u0 + u;
        float gi = Cs[rl * C_LD + u], gj = Cs[rl * C_LD + 16 + u], gf = Cs[rl * C_LD + 32 + u], go = Cs[rl * C_LD + 48 + u];
        if (g1 != nullptr) { gi += g1[gb]; gj += g1[gb + C]; gf += g1[gb + 2 * C]; go += g1[gb + 3 * C]; }
        if (g2 != nullptr) {
            const long q = (r / div2) * 4 * C + u0 + u;
            gi += g2[q]; gj += g2[q + C]; gf += g2[q + 2 * C]; go += g2[q + 3 * C];
        }
        const float ai = sigmoidf_(gi), aj = tanhf(gj), af = sigmoidf_(gf + 1.0f), ao = sigmoidf_(go);
        const float c1 = c0 * af + ai * aj;
        c_out[i] = c1;
        h_out[i] = tanhf(c1) * ao;
        acts[gb] = ai; acts[gb + C] = aj; acts[gb + 2 * C] = af; acts[gb + 3 * C] = ao;
    }
}

// The same step for FEW rows (inference batches: N*36 = 576 rows are 9 x 32 = 288 of the workgroups above, 1.1 per CU: two
// rounds, the second one almost empty).  A workgroup here owns 64 rows x 8 hidden units (32 gate columns) and its two wave
// pairs split the K steps of every K-tile between them (partial sums added through LDS before the gate math): twice the
// workgroups, half the MFMA chain each -- all of them resident at once.
__global__ __launch_bounds__(256) void lstm_step_fwd_k2_kernel(const float* __restrict__ h_in, const float* __restrict__ Kh,
                                                                int ldk, const float* __restrict__ g1,
                                                                const float* __restrict__ g2, int div2,
                                                                const int* __restrict__ mask, int mdiv,
                                                                const float* __restrict__ c_in, int rows, int C, int with_gemm,
                                                                float* __restrict__ c_out, float* __restrict__ h_out,
                                                                float* __restrict__ acts) {
    constexpr int BM = 64, BK = 32, A_LD = 36, B_LD = 32, C_LD = 36;
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * A_LD + 2 * BK * B_LD];     // 26 KB; the two C images reuse it
    static_assert(2 * BM * C_LD <= 2 * BM * A_LD + 2 * BK * B_LD, "C images fit");
    float* As = smem;
    float* Bs = smem + 2 * BM * A_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, kh = wave >> 1, l31 = lane & 31, lhi = lane >> 5;
    int rb_, ub_;
    lstm_wg_map(blockIdx.x, (rows + BM - 1) / BM, C / 8, rb_, ub_);
    const int m0 = rb_ * BM, u0 = ub_ * 8;

    lstm_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if (with_gemm) {
        // A as above; B: k row t >> 3, 16-byte piece t & 7 = gate (t & 7) >> 1, units 4 * (t & 1) ..
        const int a_r = tid >> 3, a_c = (tid & 7) * 4;
        const int b_k = tid >> 3, b_g = (tid & 7) >> 1, b_u = (tid & 1) * 4;
        const float* ap0 = h_in + (long)min(m0 + a_r, rows - 1) * C + a_c;
        const float* ap1 = h_in + (long)min(m0 + a_r + 32, rows - 1) * C + a_c;
        const float* bp = Kh + (long)b_k * ldk + b_g * C + u0 + b_u;
        // K = C is 16 tiles of 32 at C = 512 and a tile is 8 MFMAs per wave (0.2 us) against > 1 us of load latency: the loads
        // run PF tiles ahead in registers (3 float4 per tile and thread), LDS stays double-buffered
        float4 ra0_0, ra1_0, rb0_0, ra0_1, ra1_1, rb0_1, ra0_2, ra1_2, rb0_2, ra0_3, ra1_3, rb0_3;      // 4 register sets (named: no scratch)
        const int nkt = C / BK;         // host: a multiple of 4
#define K2_LOAD(S, kt)                                                          \
    {                                                                           \
        ra0_##S = *reinterpret_cast<const float4*>(ap0 + (kt) * BK);            \
        ra1_##S = *reinterpret_cast<const float4*>(ap1 + (kt) * BK);            \
        rb0_##S = *reinterpret_cast<const float4*>(bp + (long)(kt) * BK * ldk); \
    }
#define K2_STORE(buf, S)                                                                  \
    {                                                                                     \
        float* A_ = As + (buf) * BM * A_LD;                                               \
        float* B_ = Bs + (buf) * BK * B_LD;                                               \
        *reinterpret_cast<float4*>(A_ + a_r * A_LD + a_c) = ra0_##S;                      \
        *reinterpret_cast<float4*>(A_ + (a_r + 32) * A_LD + a_c) = ra1_##S;               \
        *reinterpret_cast<float4*>(B_ + b_k * B_LD + (tid & 7) * 4) = rb0_##S;            \
    }
        // tile kt (set S, LDS buffer S & 1): the load of tile kt + 4 goes to set S (tile kt went to LDS in the previous
        // sub-step), the MFMAs of tile kt, then tile kt + 1 (set SN) to the other LDS buffer
#define K2_STEP(S, SN)                                                                                                  \
    {                                                                                                                   \
        const int kt = kt0 + S;                                                                                         \
        constexpr int cur = S & 1;                                                                                      \
        K2_LOAD(S, min(kt + 4, nkt - 1)) /* branch-free (the tail re-reads the last tile): the wait counts stay exact */  \
        /* K index of MFMA step kk on lane half lhi: 16 * lhi + kk; this wave pair takes kk in [8 * kh, 8 * kh + 8) */ \
        const float* A = As + cur * BM * A_LD + (wm * 32 + l31) * A_LD + lhi * 16 + kh * 8;                             \
        const float* B = Bs + cur * BK * B_LD + (lhi * 16 + kh * 8) * B_LD + l31;                                       \
        _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                 \
            const float4 av = *reinterpret_cast<const float4*>(A + g * 4);                                              \
            const float b0 = B[(g * 4 + 0) * B_LD], b1 = B[(g * 4 + 1) * B_LD], b2 = B[(g * 4 + 2) * B_LD],             \
                        b3 = B[(g * 4 + 3) * B_LD];                                                                     \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc, 0, 0, 0);                                         \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc, 0, 0, 0);                                         \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc, 0, 0, 0);                                         \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc, 0, 0, 0);                                         \
        }                                                                                                               \
        K2_STORE(cur ^ 1, SN)                                                                                           \
        __syncthreads();                                                                                                \
    }
        K2_LOAD(0, 0) K2_LOAD(1, 1) K2_LOAD(2, 2) K2_LOAD(3, 3)
        K2_STORE(0, 0)
        __syncthreads();
        for (int kt0 = 0; kt0 < nkt; kt0 += 4) {
            K2_STEP(0, 1) K2_STEP(1, 2) K2_STEP(2, 3) K2_STEP(3, 0)
        }
#undef K2_LOAD
#undef K2_STORE
#undef K2_STEP
    }
    // accumulators -> LDS [K half][row][gate * 8 + unit]
    float* Cs = smem + kh * BM * C_LD;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * C_LD + l31] = acc[r];
    __syncthreads();
    const float* C0 = smem;
    const float* C1 = smem + BM * C_LD;
    // gate math: 64 rows x 8 units, 2 per thread
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int e = p * 256 + tid;
        const int rl = e >> 3, u = e & 7;
        const long r = m0 + rl;
        if (r >= rows) continue;
        const long i = r * C + u0 + u;
        const float c0 = c_in[i], h0 = h_in[i];
        if (mask[r / mdiv] == 0) {
            c_out[i] = c0;
            h_out[i] = h0;
            continue;
        }
        const long gb = r * 4 * C + u0 + u;
        const int o = rl * C_LD + u;
        float gi = C0[o] + C1[o], gj = C0[o + 8] + C1[o + 8], gf = C0[o + 16] + C1[o + 16], go = C0[o + 24] + C1[o + 24];
        if (g1 != nullptr) { gi += g1[gb]; gj += g1[gb + C]; gf += g1[gb + 2 * C]; go += g1[gb + 3 * C]; }
        if (g2 != nullptr) {
            const long q = (r / div2) * 4 * C + u0 + u;
            gi += g2[q]; gj += g2[q + C]; gf += g2[q + 2 * C]; go += g2[q + 3 * C];
        }
        const float ai = sigmoidf_(gi), aj = tanhf(gj), af = sigmoidf_(gf + 1.0f), ao = sigmoidf_(go);
        const float c1 = c0 * af + ai * aj;
        c_out[i] = c1;
        h_out[i] = tanhf(c1) * ao;
        acts[gb] = ai; acts[gb + C] = aj; acts[gb + 2 * C] = af; acts[gb + 3 * C] = ao;
    }
}

// ------------------------------------------------------------------ the same step on the bf16 matrix pipe (bf16x6)
// h @ Kh as six bf16 products per fp32 product (igemm_bf16.hip: x = h + m + l exactly, hh in one fp32 accumulator, the five
// correction products in another, added at the end).  The exact-fp32 step is bound by its own MFMA chain -- 64 x 32 x 512 per
// workgroup is 8192 cycles of v_mfma_f32_32x32x2_f32 per wave -- and a first bf16 form that split h in registers from
// row-major loads (a lane's 8 consecutive k of ITS row: 64 cache lines per load instruction) was slower still, bound by the
// address rate of the vector memory path.  This form has NO LDS, NO barrier and NO split in its K loop, only lane-linear 16-byte
// loads (1 KiB per wave and instruction):
//   * K_h comes pre-split in the fragment-major planes of ssc_filter_split (taps 1, c0 = C, c1 = 4C, orient 0); the workgroup's
//     32 gate columns (4 gates x 8 units) sit in four fragments, a lane takes the 16 bytes of its column's lane there;
//   * h comes pre-split too: the step's epilogue writes h_out a second time as bf16 planes in the A-operand fragment layout
//     (hp_out: fragment (rb, kc, plane) = 1 KiB at ((rb * C/16 + kc) * 3 + plane) * 1024, rb = 32-row block, lane L holds
//     h[rb * 32 + (L & 31)][kc * 16 + (L >> 5) * 8 .. + 7]) -- a workgroup's 8 units are exactly one lane's 16 bytes per row.
// Workgroup = 64 rows x 8 units; wave = one K quarter of both 32-row blocks (the B fragments are loaded once for the two);
// loads run PF chunks ahead; the four partial tiles meet in LDS before the gate math.
typedef short lstm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 lstm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float lstm_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned lstm_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lstm_cvt_pk_bf16(float a, float b) {
    const lstm_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, lstm_bf16x2));
}
__device__ __forceinline__ void lstm_split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = lstm_cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = lstm_cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = lstm_cvt_pk_bf16(s0, s1);
}

// h [rows, C] -> its planes (the form the step's epilogue writes): thread = (row, 8 consecutive k)
__global__ void lstm_hsplit_kernel(const float* __restrict__ h, int rows, int C, char* __restrict__ hp) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = C / 8;
    const long r = t / per_row;
    if (r >= rows) return;
    const int k8 = (int)(t - r * per_row);
    const float4 x0 = *reinterpret_cast<const float4*>(h + r * C + k8 * 8);
    const float4 x1 = *reinterpret_cast<const float4*>(h + r * C + k8 * 8 + 4);
    lstm_u32x4 ph, pm, pl;
    unsigned a, b, c;
    lstm_split3_pair(x0.x, x0.y, a, b, c); ph[0] = a; pm[0] = b; pl[0] = c;
    lstm_split3_pair(x0.z, x0.w, a, b, c); ph[1] = a; pm[1] = b; pl[1] = c;
    lstm_split3_pair(x1.x, x1.y, a, b, c); ph[2] = a; pm[2] = b; pl[2] = c;
    lstm_split3_pair(x1.z, x1.w, a, b, c); ph[3] = a; pm[3] = b; pl[3] = c;
    const long rb = r >> 5;
    const int kc = k8 >> 1, ln = (int)(r & 31) + 32 * (k8 & 1);
    char* f = hp + ((rb * (C / 16) + kc) * 3) * 1024 + ln * 16;
    *reinterpret_cast<lstm_u32x4*>(f) = ph;
    *reinterpret_cast<lstm_u32x4*>(f + 1024) = pm;
    *reinterpret_cast<lstm_u32x4*>(f + 2048) = pl;
}

// UB: blocks of 8 units (32 gate columns) per workgroup that share the A fragments.  1: 64 x 32 tiles, three workgroups per CU
// (few rows: as many workgroups as possible).  2: 64 x 64 -- a third less operand traffic per product from L2, which is what
// bounds the step at many rows (the Background module's C = 1024 cell over 2304 rows: 590 KB of fragments per 4.2 MFLOP tile).
template <int PF, int UB>
__global__ __launch_bounds__(256, UB == 1 ? 3 : 2) void lstm_step_fwd_bf_kernel(const float* __restrict__ h_in, const char* __restrict__ hp_in,
                                                                const char* __restrict__ Kp, int nbp,
                                                                const float* __restrict__ g1, const float* __restrict__ g2,
                                                                int div2, const int* __restrict__ mask, int mdiv,
                                                                const float* __restrict__ c_in, int rows, int C,
                                                                float* __restrict__ c_out, float* __restrict__ h_out,
                                                                char* __restrict__ hp_out, float* __restrict__ acts) {
    constexpr int BM = 64, C_LD = 36;
    __shared__ __attribute__((aligned(16))) float smem[4 * BM * C_LD];      // the four K quarters' C images (36 KB)
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    int rb_, ub_;
    lstm_wg_map(blockIdx.x, (rows + BM - 1) / BM, C / (8 * UB), rb_, ub_);
    const int m0 = rb_ * BM, u0 = ub_ * 8 * UB;
    const int KC = C / 16;

    lstm_f32x16 acc[UB][2], accc[UB][2];
#pragma unroll
    for (int j = 0; j < UB; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = accc[j][i][r] = 0.f;

    // The gate math's inputs are asked for NOW: thread = (row tid >> 2, units 2 * (tid & 3), + 1), so the quad of a row holds the
    // row's 8 units (= one lane's 16 bytes of the next step's A fragment per plane).  With ~250 registers only two waves share
    // a SIMD and nothing hides an epilogue that starts its loads behind the K loop (17 us of a 28 us step that way).
    const int rl = tid >> 2, u = (tid & 3) * 2;
    const long r = m0 + rl;
    const bool valid = r < rows;
    bool live = false;
    float2 c0[UB], h0[UB], gv1[UB][4], gv2[UB][4];
#pragma unroll
    for (int j = 0; j < UB; ++j) {
        c0[j] = h0[j] = make_float2(0.f, 0.f);
#pragma unroll
        for (int g = 0; g < 4; ++g) gv1[j][g] = gv2[j][g] = make_float2(0.f, 0.f);
    }
    if (valid) {
        live = mask[r / mdiv] != 0;
#pragma unroll
        for (int j = 0; j < UB; ++j) {
            const long i = r * C + u0 + 8 * j + u;
            c0[j] = *reinterpret_cast<const float2*>(c_in + i);
            h0[j] = *reinterpret_cast<const float2*>(h_in + i);
            // (not behind `live`: a second round trip to memory costs more than the skipped rows' addends)
            const long gb = r * 4 * C + u0 + 8 * j + u;
            if (g1 != nullptr) {
#pragma unroll
                for (int g = 0; g < 4; ++g) gv1[j][g] = *reinterpret_cast<const float2*>(g1 + gb + g * C);
            }
            if (g2 != nullptr) {
                const long q = (r / div2) * 4 * C + u0 + 8 * j + u;
#pragma unroll
                for (int g = 0; g < 4; ++g) gv2[j][g] = *reinterpret_cast<const float2*>(g2 + q + g * C);
            }
        }
    }

    if (hp_in != nullptr) {
        // A: fragments (2 * rb_ + i, kc, plane), lane-linear;  B: column l31 of the tile = gate l31 >> 3, unit u0 + (l31 & 7)
        const char* ap = hp_in + ((long)(2 * rb_) * KC * 3) * 1024 + lane * 16;
        const long a_rb = (long)KC * 3 * 1024;
        const char* bp[UB];
#pragma unroll
        for (int j = 0; j < UB; ++j) {
            const int col = (l31 >> 3) * C + u0 + 8 * j + (l31 & 7);
            bp[j] = Kp + (long)(col >> 5) * 1024 + ((col & 31) + 32 * lhi) * 16;
        }
        const long pl_stride = (long)nbp * 1024, kc_stride = 3 * pl_stride;
        const int nkc = KC / 4;             // chunks of 16 k per K quarter (host: a multiple of PF)
        const int kc0 = kq * nkc;

        lstm_u32x4 ra[PF][2][3], rb[PF][UB][3];
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const char* a = ap + (long)(kc0 + s) * 3 * 1024;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                ra[s][0][p] = *reinterpret_cast<const lstm_u32x4*>(a + p * 1024);
                ra[s][1][p] = *reinterpret_cast<const lstm_u32x4*>(a + a_rb + p * 1024);
#pragma unroll
                for (int j = 0; j < UB; ++j)
                    rb[s][j][p] = *reinterpret_cast<const lstm_u32x4*>(bp[j] + (kc0 + s) * kc_stride + p * pl_stride);
            }
        }
        // compiler fences keep the loads where they are written: without them the prologue's loads sink into the loop and every
        // iteration waits for the data it has just asked for
        asm volatile("" ::: "memory");
        for (int k0 = 0; k0 < nkc; k0 += PF) {
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                lstm_bf16x8 A[2][3], Bv[UB][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    A[0][p] = __builtin_bit_cast(lstm_bf16x8, ra[s][0][p]);
                    A[1][p] = __builtin_bit_cast(lstm_bf16x8, ra[s][1][p]);
#pragma unroll
                    for (int j = 0; j < UB; ++j) Bv[j][p] = __builtin_bit_cast(lstm_bf16x8, rb[s][j][p]);
                }
                // the chunk PF further on into the set just consumed (branch-free: the tail re-reads the last chunk)
                {
                    const int kn = kc0 + min(k0 + s + PF, nkc - 1);
                    const char* a = ap + (long)kn * 3 * 1024;
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        ra[s][0][p] = *reinterpret_cast<const lstm_u32x4*>(a + p * 1024);
                        ra[s][1][p] = *reinterpret_cast<const lstm_u32x4*>(a + a_rb + p * 1024);
#pragma unroll
                        for (int j = 0; j < UB; ++j)
                            rb[s][j][p] = *reinterpret_cast<const lstm_u32x4*>(bp[j] + kn * kc_stride + p * pl_stride);
                    }
                    asm volatile("" ::: "memory");
                }
                // products smallest first, the two row blocks in turn (no back-to-back dependence)
                constexpr int pa[5] = {2, 0, 1, 1, 0}, pb[5] = {0, 2, 1, 0, 1};
#pragma unroll
                for (int t = 0; t < 5; ++t)
#pragma unroll
                    for (int j = 0; j < UB; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            accc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][pa[t]], Bv[j][pb[t]], accc[j][i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < UB; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], Bv[j][0], acc[j][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // the next chunk's MFMAs stay behind these (else all sets are waited for at once)
            }
        }
    }
    // accumulators -> LDS [K quarter][row][gate * 8 + unit], one block of 8 units at a time
    float* Cs = smem + kq * BM * C_LD;
#pragma unroll
    for (int j = 0; j < UB; ++j) {
        if (j > 0) __syncthreads();         // the previous block's images have been read
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr)
                Cs[(i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lhi) * C_LD + l31] = acc[j][i][rr] + accc[j][i][rr];
        __syncthreads();
        if (!valid) continue;
        const int uj = u0 + 8 * j;
        const long i = r * C + uj + u;
        float2 hv = h0[j];
        if (!live) {
            *reinterpret_cast<float2*>(c_out + i) = c0[j];
            *reinterpret_cast<float2*>(h_out + i) = h0[j];      // acts are never read for skipped steps
        } else {
            const long gb = r * 4 * C + uj + u;
            const int o = rl * C_LD + u;
            float2 z[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                z[g] = make_float2(gv1[j][g].x + gv2[j][g].x, gv1[j][g].y + gv2[j][g].y);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float2 p = *reinterpret_cast<const float2*>(smem + q * BM * C_LD + o + g * 8);
                    z[g].x += p.x; z[g].y += p.y;
                }
            }
            const float2 ai = make_float2(sigmoidf_(z[0].x), sigmoidf_(z[0].y)), aj = make_float2(tanhf(z[1].x), tanhf(z[1].y)),
                         af = make_float2(sigmoidf_(z[2].x + 1.0f), sigmoidf_(z[2].y + 1.0f)),
                         ao = make_float2(sigmoidf_(z[3].x), sigmoidf_(z[3].y));
            const float2 c1 = make_float2(c0[j].x * af.x + ai.x * aj.x, c0[j].y * af.y + ai.y * aj.y);
            hv = make_float2(tanhf(c1.x) * ao.x, tanhf(c1.y) * ao.y);
            *reinterpret_cast<float2*>(c_out + i) = c1;
            *reinterpret_cast<float2*>(h_out + i) = hv;
            if (acts != nullptr) {      // the backward pass's copy of the activated gates; inference passes NULL
                *reinterpret_cast<float2*>(acts + gb) = ai;
                *reinterpret_cast<float2*>(acts + gb + C) = aj;
                *reinterpret_cast<float2*>(acts + gb + 2 * C) = af;
                *reinterpret_cast<float2*>(acts + gb + 3 * C) = ao;
            }
        }
        if (hp_out != nullptr) {
            unsigned ph, pm, pl;
            lstm_split3_pair(hv.x, hv.y, ph, pm, pl);
            const int ln = (int)(r & 31) + 32 * ((uj >> 3) & 1);
            char* f = hp_out + (((r >> 5) * KC + (uj >> 4)) * 3) * 1024 + ln * 16 + (tid & 3) * 4;
            *reinterpret_cast<unsigned*>(f) = ph;
            *reinterpret_cast<unsigned*>(f + 1024) = pm;
            *reinterpret_cast<unsigned*>(f + 2048) = pl;
        }
    }
}

// h [rows, C] (fp32) -> hp, the bf16 planes ssc_lstm_step_fwd_bf reads: ceil(rows / 64) * 2 * (C / 16) * 3 KiB (whole 64-row tiles)
extern "C" int ssc_lstm_hsplit(const float* h, int64_t rows, int C, void* hp, void* stream) {
    if ((C & 15) || rows <= 0) return -1;
    const long threads = (long)rows * (C / 8);
    hipLaunchKernelGGL(lstm_hsplit_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h, (int)rows,
                       C, (char*)hp);
    return CHECK_LAUNCH();
}

// Kp: the planes of K_h [C, 4C] from ssc_filter_split(Kh, 1, C, 4 * C, 0, ...), nbp their blocks-per-plane count
// (ssc_filter_split_geom).  hp_in: the planes of h_in (ssc_lstm_hsplit, or the previous step's hp_out); NULL = h_in is zero, no
// product (ssc_lstm_step_fwd's with_gemm = 0).  hp_out (may be NULL): receives the planes of h_out.  Otherwise the arguments and
// results of ssc_lstm_step_fwd.
extern "C" int ssc_lstm_step_fwd_bf(const float* h_in, const void* hp_in, const void* Kp, int nbp, const float* g1,
                                    const float* g2, int div2, const int* mask, int mdiv, const float* c_in, int64_t rows,
                                    int C, float* c_out, float* h_out, void* hp_out, float* acts, void* stream) {
    if ((C & 127) || Kp == nullptr || nbp < (4 * C) / 32 || rows <= 0 || rows > 0x7fffffffL / (4L * C)) return -1;
    // 64 x 64 tiles (two blocks of 8 units per workgroup) from 1152 workgroups of the 64 x 32 grid on (4.5 per CU).  Measured
    // (scripts/lstm_ub_ab.sh, us per step 64x32 -> 64x64): C = 512: 576 rows 23.9 -> 25.7, 1152 rows 39.7 -> 38.4, 2304 rows 91.5 -> 72.7;
    // C = 1024: 576 rows 57.5 -> 56.2, 2304 rows 312 -> 192, 4608 rows 598 -> 383
    static int ub_env = -2;     // SSC_LSTM_UB=1 / 2 pins the tile (A/B)
    if (ub_env == -2) {
        const char* e = ssc_dev_getenv("SSC_LSTM_UB");
        ub_env = e != nullptr ? atoi(e) : -1;
    }
    const long wgs1 = ((rows + 63) / 64) * (C / 8);
    const int ub = ((C / 8) & 1) ? 1 : (ub_env == 1 || ub_env == 2 ? ub_env : (wgs1 >= 1152 ? 2 : 1));
    const dim3 grid((unsigned)(wgs1 / ub));
#define LSTM_BF_LAUNCH(PF, UBV)                                                                                           \
    hipLaunchKernelGGL((lstm_step_fwd_bf_kernel<PF, UBV>), grid, dim3(256), 0, (hipStream_t)stream, h_in, (const char*)hp_in, \
                       (const char*)Kp, nbp, g1, g2, div2 > 0 ? div2 : 1, mask, mdiv > 0 ? mdiv : 1, c_in, (int)rows, C,  \
                       c_out, h_out, (char*)hp_out, acts)
    if (ub == 2) LSTM_BF_LAUNCH(2, 2);
    else LSTM_BF_LAUNCH(2, 1);      // C / 64 chunks per K quarter, a multiple of PF = 2 (18 KiB of loads in flight per wave)
#undef LSTM_BF_LAUNCH
    return CHECK_LAUNCH();
}

extern "C" int ssc_lstm_step_fwd(const float* h_in, const float* Kh, int ldk, const float* g1, const float* g2, int div2,
                                 const int* mask, int mdiv, const float* c_in, int64_t rows, int C, int with_gemm,
                                 float* c_out, float* h_out, float* acts, void* stream) {
    if ((C & 31) || (ldk & 3) || rows <= 0 || rows > 0x7fffffffL / (4L * C)) return -1;
    // few rows: the K-split form while the plain grid holds under ~4.5 workgroups per CU (SSC_LSTM_K2=0 / 1 pins it).  Measured:
    // batch 16 (576 rows) generator forward 9080 -> 9390 images/s; batch 32 (1152 rows) train step 17.96 -> 17.88 ms
    static int k2 = -2;
    if (k2 == -2) {
        const char* e = ssc_dev_getenv("SSC_LSTM_K2");
        k2 = (e == nullptr) ? -1 : atoi(e);
    }
    const long wgs = ((rows + 63) / 64) * (C / 16);
    const bool use_k2 = with_gemm && (C % 128) == 0 && (k2 == 1 || (k2 == -1 && wgs < 1200));     // its loads run 4 K tiles ahead
    if (use_k2)
        hipLaunchKernelGGL(lstm_step_fwd_k2_kernel, dim3((unsigned)(((rows + 63) / 64) * (C / 8))), dim3(256), 0,
                           (hipStream_t)stream, h_in, Kh, ldk, g1, g2, div2 > 0 ? div2 : 1, mask, mdiv > 0 ? mdiv : 1, c_in,
                           (int)rows, C, with_gemm, c_out, h_out, acts);
    else
        hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3((unsigned)(((rows + 63) / 64) * (C / 16))), dim3(256), 0,
                           (hipStream_t)stream, h_in, Kh, ldk, g1, g2, div2 > 0 ? div2 : 1, mask, mdiv > 0 ? mdiv : 1, c_in,
                           (int)rows, C, with_gemm, c_out, h_out, acts);
    return CHECK_LAUNCH();
}

extern "C" int ssc_lstm_pointwise_fwd(const float* g0, const float* g1, const float* g2, int div2, const int* mask,
                                      int mdiv, const float* c_in, const float* h_in, int64_t rows, int C,
                                      float* c_out, float* h_out, float* acts, void* stream) {
    const long tot = (long)rows * C;
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g0, g1,
                       g2, div2 > 0 ? div2 : 1, mask, mdiv > 0 ? mdiv : 1, c_in, h_in, (long)rows, C, c_out, h_out, acts);
    return CHECK_LAUNCH();
}

extern "C" int ssc_lstm_pointwise_bwd(const float* dh, const float* dc, const float* acts, const float* c_in,
                                      const float* c_out, const int* mask, int mdiv, int64_t rows, int C, float* dg,
                                      float* dc_in, float* dh_pass, float* gacc, void* stream) {
    const long tot = (long)rows * C;
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh, dc,
                       acts, c_in, c_out, mask, mdiv > 0 ? mdiv : 1, (long)rows, C, dg, dc_in, dh_pass, gacc);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ squash: relu(0.5*(log(1+1e-3+h) - log(1+1e-3-h)))
__global__ void squash_fwd_kernel(const float* __restrict__ h, long n, float* __restrict__ o) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = h[i];
    const float s = 0.5f * (logf(1.0f + 1e-3f + v) - logf(1.0f + 1e-3f - v));
    o[i] = fmaxf(s, 0.f);
}

// dh = go * [o>0] * 0.5*(1/(1.001+h) + 1/(1.001-h))
__global__ void squash_bwd_kernel(const float* __restrict__ h, const float* __restrict__ o,
                                  const float* __restrict__ go, long n, float* __restrict__ dh) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = h[i];
    const float d = 0.5f * (1.f / (1.0f + 1e-3f + v) + 1.f / (1.0f + 1e-3f - v));
    dh[i] = o[i] > 0.f ? go[i] * d : 0.f;
}

extern "C" int ssc_squash_fwd(const float* h, int64_t n, float* o, void* stream) {
    hipLaunchKernelGGL(squash_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h,
                       (long)n, o);
    return CHECK_LAUNCH();
}

extern "C" int ssc_squash_bwd(const float* h, const float* o, const float* go, int64_t n, float* dh, void* stream) {
    hipLaunchKernelGGL(squash_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h, o,
                       go, (long)n, dh);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ small reductions
// out[g][c] (+)= sum_{r in group g} x[g*G + r][c]     (G rows per group; G = M -> column sum)
// block = 64 columns x 4 row lanes; grid = (column chunks, groups)
__global__ __launch_bounds__(256) void group_rowsum_kernel(const float* __restrict__ x, int ldx, long groups, int G,
                                                            int C, float* __restrict__ out, int accumulate) {
    __shared__ float sh[256];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long g = blockIdx.y;
    float s = 0.f;
    if (c < C) {
        const float* p = x + (g * G) * ldx + c;
        for (int r = rl; r < G; r += 4) s += p[(long)r * ldx];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        s = sh[cl] + sh[64 + cl] + sh[128 + cl] + sh[192 + cl];
        const long i = g * C + c;
        if (accumulate) s += out[i];
        out[i] = s;
    }
}

extern "C" int ssc_group_rowsum(const float* x, int ldx, int64_t groups, int G, int C, float* out, int accumulate,
                                void* stream) {
    hipLaunchKernelGGL(group_rowsum_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)groups), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, (long)groups, G, C, out, accumulate);
    return CHECK_LAUNCH();
}

// out[n][c] = mean_p act(a*x[n,p,c]+b)      (tf.reduce_mean over H,W; models_collection.py:838)
// block = 64 channels x 4 row lanes; grid = (channel chunks, N)
__global__ __launch_bounds__(256) void act_mean_hw_kernel(const float* __restrict__ x, const float* __restrict__ ab,
                                                           int act, int N, int P, int C, float* __restrict__ out) {
    __shared__ float sh[256];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int n = blockIdx.y;
    float s = 0.f;
    if (c < C) {
        const float a = ab != nullptr ? ab[c] : 1.f, b = ab != nullptr ? ab[C + c] : 0.f;
        const float* p = x + (long)n * P * C + c;
        const float slope = act == SSC_ACT_RELU ? 0.f : (act == SSC_ACT_LRELU ? 0.2f : 1.f);   // act(t) = max(t, slope*t)
        // 8 independent partial sums so that the strided loads overlap (256 workgroups, each a serial walk over P)
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int q = rl;
        for (; q + 28 < P; q += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float v = fmaf(a, p[(long)(q + 4 * u) * C], b);
                s8[u] += fmaxf(v, slope * v);
            }
        }
        for (; q < P; q += 4) {
            const float v = fmaf(a, p[(long)q * C], b);
            s8[0] += fmaxf(v, slope * v);
        }
        s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && c < C) out[(long)n * C + c] = (sh[cl] + sh[64 + cl] + sh[128 + cl] + sh[192 + cl]) / (float)P;
}

// g[n,p,c] += v[n,c]*scale
__global__ void add_row_bcast_kernel(float* __restrict__ g, const float* __restrict__ v, float scale, int N, int P,
                                     int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * P * C) return;
    const int c = (int)(i % C);
    const int n = (int)(i / ((long)P * C));
    g[i] += v[(long)n * C + c] * scale;
}

// the same, 4 channels per thread (C % 4 == 0, 16-byte aligned): one 16-byte read-modify-write instead of four 4-byte ones
__global__ __launch_bounds__(256) void add_row_bcast4_kernel(float4* __restrict__ g, const float4* __restrict__ v, float scale,
                                                             unsigned tot4, unsigned C4, unsigned PC4) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < tot4; i += gridDim.x * 256u) {
        const unsigned n = i / PC4, c4 = i % C4;
        float4 x = g[i];
        const float4 t = v[n * C4 + c4];
        x.x = fmaf(t.x, scale, x.x); x.y = fmaf(t.y, scale, x.y); x.z = fmaf(t.z, scale, x.z); x.w = fmaf(t.w, scale, x.w);
        g[i] = x;
    }
}

extern "C" int ssc_act_mean_hw(const float* x, const float* ab, int act, int N, int P, int C, float* out,
                               void* stream) {
    hipLaunchKernelGGL(act_mean_hw_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, x, ab, act, N, P, C, out);
    return CHECK_LAUNCH();
}

extern "C" int ssc_add_row_bcast(float* g, const float* v, float scale, int N, int P, int C, void* stream) {
    const long tot = (long)N * P * C;
    if ((C & 3) == 0 && tot / 4 < 0x7fffffffL && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        const unsigned tot4 = (unsigned)(tot / 4);
        unsigned blocks = (tot4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(add_row_bcast4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(g),
                           reinterpret_cast<const float4*>(v), scale, tot4, (unsigned)(C / 4), (unsigned)((long)P * C / 4));
        return CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(add_row_bcast_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, v,
                       scale, N, P, C);
    return CHECK_LAUNCH();
}

// ------------------------------------------------------------------ noise head: miu_relu + NCHW->NHWC reshape
// pre[n, c*P + p] -> out[n, p, c] = (x + sqrt(0.09 + x^2))/2      (models_collection.py:63-65, 493-499)
__global__ void miu_permute_fwd_kernel(const float* __restrict__ pre, int N, int Cc, int P, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * Cc * P) return;
    const int c = (int)(i % Cc);
    const int p = (int)((i / Cc) % P);
    const int n = (int)(i / ((long)Cc * P));
    const float x = pre[(long)n * Cc * P + (long)c * P + p];
    out[i] = 0.5f * (x + sqrtf(0.09f + x * x));
}

// dpre[n, c*P+p] = g[n,p,c] * [out>0] * 0.5*(1 + x/sqrt(0.09+x^2))
__global__ void miu_permute_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ g, int N, int Cc, int P,
                                       float* __restrict__ dpre) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * Cc * P) return;
    const int p = (int)(i % P);
    const int c = (int)((i / P) % Cc);
    const int n = (int)(i / ((long)Cc * P));
    const float x = pre[i];
    const float sq = sqrtf(0.09f + x * x);
    const float o = 0.5f * (x + sq);
    const float gv = g[((long)n * P + p) * Cc + c];
    dpre[i] = o > 0.f ? gv * 0.5f * (1.f + x / sq) : 0.f;
}

extern "C" int ssc_miu_permute_fwd(const float* pre, int N, int Cc, int P, float* out, void* stream) {
    const long tot = (long)N * Cc * P;
    hipLaunchKernelGGL(miu_permute_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       pre, N, Cc, P, out);
    return CHECK_LAUNCH();
}

extern "C" int ssc_miu_permute_bwd(const float* pre, const float* g, int N, int Cc, int P, float* dpre, void* stream) {
    const long tot = (long)N * Cc * P;
    hipLaunchKernelGGL(miu_permute_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       pre, g, N, Cc, P, dpre);
    return CHECK_LAUNCH();
}
