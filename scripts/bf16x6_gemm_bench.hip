// Feasibility microbenchmark: fp32 GEMM emulated on the bf16 matrix pipe with a 3-way operand split
// (x = h + m + l, each a bf16) and the 6 largest cross products (hh, hm, mh, hl, lh, mm), fp32 accumulate.
//   C[M,N] = A[M,K] * B[K,N];  A fp32 row-major split on the fly, B pre-split into three [N][K] bf16 planes.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/bf16x6_gemm_bench.hip -o gpurun_out/bf16x6_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define BM 128
#define BN 128
#define BK 32
#define LDT 40            // bf16 elements per LDS row (80 B: 16-byte slots with an odd multiple)
#define NTERMS_DEFAULT 6
#ifndef NBUF
#define NBUF 1     // LDS buffers: 1 = 60 KB per workgroup (2 workgroups per CU), 2 = double-buffered (1 per CU)
#endif

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned xb = __float_as_uint(x);
    h = xb & 0xffff0000u;
    const float r1 = x - __uint_as_float(h);
    const unsigned r1b = __float_as_uint(r1);
    m = r1b & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m);
    l = __float_as_uint(r2);     // upper half taken by the packer
}

__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) {   // [a.hi16 | b.hi16 << 16]
    return (a >> 16) | (b & 0xffff0000u);
}

__global__ void split_b_kernel(const float* __restrict__ B, int K, int N, unsigned short* __restrict__ ph,
                               unsigned short* __restrict__ pm, unsigned short* __restrict__ pl) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)K * N) return;
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    unsigned h, m, l;
    split3(B[(long)k * N + n], h, m, l);
    ph[i] = h >> 16;
    pm[i] = m >> 16;
    pl[i] = l >> 16;
}

template <int NTERMS>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Bh,
                                                   const unsigned short* __restrict__ Bm,
                                                   const unsigned short* __restrict__ Bl, float* __restrict__ C, int M,
                                                   int N, int K) {
    extern __shared__ unsigned short lds[];
    // [buf][plane][row][LDT] for A then B
    unsigned short* As = lds;                                  // NBUF * 3 * BM * LDT
    unsigned short* Bs = lds + NBUF * 3 * BM * LDT;            // NBUF * 3 * BN * LDT
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int l31 = lane & 31, lhi = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // loaders
    const int a_row = tid >> 3, a_k = (tid & 7) * 4;           // + 32*i rows
    const int b_n = tid >> 2, b_k = (tid & 3) * 8;             // + 64*j cols
    float4 ra[4];
    u32x4 rb[3][2];

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ra[i] = *reinterpret_cast<const float4*>(A + (long)(m0 + a_row + 32 * i) * K + k0 + a_k);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long off = (long)(n0 + b_n + 64 * j) * K + k0 + b_k;
            rb[0][j] = *reinterpret_cast<const u32x4*>(Bh + off);
            rb[1][j] = *reinterpret_cast<const u32x4*>(Bm + off);
            rb[2][j] = *reinterpret_cast<const u32x4*>(Bl + off);
        }
    };
    auto lstore = [&](int buf) {
        unsigned short* as = As + buf * 3 * BM * LDT;
        unsigned short* bs = Bs + buf * 3 * BN * LDT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h[4], m[4], l[4];
            split3(ra[i].x, h[0], m[0], l[0]);
            split3(ra[i].y, h[1], m[1], l[1]);
            split3(ra[i].z, h[2], m[2], l[2]);
            split3(ra[i].w, h[3], m[3], l[3]);
            const int o = (a_row + 32 * i) * LDT + a_k;
            u32x2 vh = {pack_hi(h[0], h[1]), pack_hi(h[2], h[3])};
            u32x2 vm = {pack_hi(m[0], m[1]), pack_hi(m[2], m[3])};
            u32x2 vl = {pack_hi(l[0], l[1]), pack_hi(l[2], l[3])};
            *reinterpret_cast<u32x2*>(as + 0 * BM * LDT + o) = vh;
            *reinterpret_cast<u32x2*>(as + 1 * BM * LDT + o) = vm;
            *reinterpret_cast<u32x2*>(as + 2 * BM * LDT + o) = vl;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = (b_n + 64 * j) * LDT + b_k;
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(bs + p * BN * LDT + o) = rb[p][j];
        }
    };

    const int nk = K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kt + 1);
        const unsigned short* as = As + cur * 3 * BM * LDT;
        const unsigned short* bs = Bs + cur * 3 * BN * LDT;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    a[i][p] = *reinterpret_cast<const bf16x8*>(as + p * BM * LDT + (wm * 64 + i * 32 + l31) * LDT +
                                                               kc * 16 + lhi * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    b[j][p] = *reinterpret_cast<const bf16x8*>(bs + p * BN * LDT + (wn * 64 + j * 32 + l31) * LDT +
                                                               kc * 16 + lhi * 8);
            // smallest terms first; consecutive MFMAs hit different accumulators
            // terms by decreasing magnitude: hh, hm, mh, mm, hl, lh, ml, lm (planes 0 = h, 1 = m, 2 = l)
            const int pa[8] = {0, 0, 1, 1, 0, 2, 1, 2}, pb[8] = {0, 1, 0, 1, 2, 0, 2, 1};
#pragma unroll
            for (int t = NTERMS - 1; t >= 0; --t) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa[t]], b[j][pb[t]], acc[i][j], 0, 0, 0);
            }
        }
        if (NBUF == 2) {
            if (more) lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        } else {
            __syncthreads();            // every wave has read its operands
            if (more) lstore(0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int col = n0 + wn * 64 + j * 32 + l31;
                C[(long)row * N + col] = acc[i][j][r];
            }
}

// ---------------------------------------------------------------------------------------------------------
// variant 2: B fragments straight from global memory.  The pre-split planes are stored fragment-major:
//   plane[((nb * (K/16) + kc) * 64 + lane) * 8 + e] = bf16 part of B[k = kc*16 + (lane/32)*8 + e][n = nb*32 + lane%32]
// so that one wavefront's 32x16 operand of v_mfma_f32_32x32x16_bf16 is one contiguous 1 KiB global load.
// Only A goes through LDS (double-buffered, 60 KB -> 2 workgroups per CU).
__global__ void split_b_frag_kernel(const float* __restrict__ B, int K, int N, unsigned short* __restrict__ ph,
                                    unsigned short* __restrict__ pm, unsigned short* __restrict__ pl) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)K * N) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long blk = i >> 9;
    const int kcn = K / 16;
    const int kc = (int)(blk % kcn), nb = (int)(blk / kcn);
    const int k = kc * 16 + (lane >> 5) * 8 + e, n = nb * 32 + (lane & 31);
    unsigned h, m, l;
    split3(B[(long)k * N + n], h, m, l);
    ph[i] = h >> 16;
    pm[i] = m >> 16;
    pl[i] = l >> 16;
}

template <int NTERMS>
__global__ __launch_bounds__(256) void gemm2_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Bh,
                                                    const unsigned short* __restrict__ Bm,
                                                    const unsigned short* __restrict__ Bl, float* __restrict__ C, int M,
                                                    int N, int K) {
    extern __shared__ unsigned short lds[];
    unsigned short* As = lds;                                  // [2][3][BM][LDT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int kcn = K / 16;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_row = tid >> 3, a_k = (tid & 7) * 4;
    float4 ra[4];
    bf16x8 bq[2][2][2][3];          // [buf][kc][j][plane]
    const unsigned short* planes[3] = {Bh, Bm, Bl};

    auto gload_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ra[i] = *reinterpret_cast<const float4*>(A + (long)(m0 + a_row + 32 * i) * K + kt * BK + a_k);
    };
    auto gload_b = [&](int kt, int buf) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long nb = (n0 + wn * 64 + j * 32) / 32;
                const long off = ((nb * kcn + (kt * 2 + kc)) * 64 + lane) * 8;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(planes[p] + off);
                    if (buf == 0) bq[0][kc][j][p] = v; else bq[1][kc][j][p] = v;
                }
            }
    };
    auto lstore = [&](int buf) {
        unsigned short* as = As + buf * 3 * BM * LDT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h[4], m[4], l[4];
            split3(ra[i].x, h[0], m[0], l[0]);
            split3(ra[i].y, h[1], m[1], l[1]);
            split3(ra[i].z, h[2], m[2], l[2]);
            split3(ra[i].w, h[3], m[3], l[3]);
            const int o = (a_row + 32 * i) * LDT + a_k;
            u32x2 vh = {pack_hi(h[0], h[1]), pack_hi(h[2], h[3])};
            u32x2 vm = {pack_hi(m[0], m[1]), pack_hi(m[2], m[3])};
            u32x2 vl = {pack_hi(l[0], l[1]), pack_hi(l[2], l[3])};
            *reinterpret_cast<u32x2*>(as + 0 * BM * LDT + o) = vh;
            *reinterpret_cast<u32x2*>(as + 1 * BM * LDT + o) = vm;
            *reinterpret_cast<u32x2*>(as + 2 * BM * LDT + o) = vl;
        }
    };

    const int nk = K / BK;          // even (K % 64 == 0)
    auto compute = [&](int buf, auto BUF) {
        constexpr int B = decltype(BUF)::value;
        const unsigned short* as = As + buf * 3 * BM * LDT;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 a[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    a[i][p] = *reinterpret_cast<const bf16x8*>(as + p * BM * LDT + (wm * 64 + i * 32 + l31) * LDT +
                                                               kc * 16 + lhi * 8);
            const int pa[8] = {0, 0, 1, 1, 0, 2, 1, 2}, pb[8] = {0, 1, 0, 1, 2, 0, 2, 1};
#pragma unroll
            for (int t = NTERMS - 1; t >= 0; --t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa[t]], bq[B][kc][j][pb[t]], acc[i][j], 0, 0, 0);
        }
    };
    gload_a(0);
    gload_b(0, 0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        gload_a(kt + 1);
        gload_b(kt + 1, 1);
        compute(0, std::integral_constant<int, 0>());
        lstore(1);
        __syncthreads();
        const bool more = kt + 2 < nk;
        if (more) {
            gload_a(kt + 2);
            gload_b(kt + 2, 0);
        }
        compute(1, std::integral_constant<int, 1>());
        if (more) lstore(0);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int col = n0 + wn * 64 + j * 32 + l31;
                C[(long)row * N + col] = acc[i][j][r];
            }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NT>
static void run(const float* dA, const unsigned short* h, const unsigned short* m, const unsigned short* l, float* dC,
                int M, int N, int K, hipStream_t st) {
    const size_t lds = (size_t)NBUF * 3 * (BM + BN) * LDT * 2;
    CK(hipFuncSetAttribute((const void*)gemm_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(gemm_kernel<NT>, dim3(N / BN, M / BM), dim3(256), lds, st, dA, h, m, l, dC, M, N, K);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N), hC((size_t)M * N);
    srand(1);
    const int relu = argc > 4 ? atoi(argv[4]) : 0;
    for (auto& v : hA) { v = (float)rand() / RAND_MAX * 2.f - 1.f; if (relu) v = v < 0.f ? 0.f : 3.f * v; }
    for (auto& v : hB) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
    float *dA, *dB, *dC;
    unsigned short *ph, *pm, *pl;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, hC.size() * 4));
    CK(hipMalloc(&ph, hB.size() * 2)); CK(hipMalloc(&pm, hB.size() * 2)); CK(hipMalloc(&pl, hB.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split_b_kernel, dim3((unsigned)(((size_t)K * N + 255) / 256)), dim3(256), 0, 0, dB, K, N, ph, pm, pl);
    CK(hipDeviceSynchronize());
    for (int nt : {3, 6, 8}) {
        auto launch = [&]() {
            if (nt == 3) run<3>(dA, ph, pm, pl, dC, M, N, K, 0);
            else if (nt == 6) run<6>(dA, ph, pm, pl, dC, M, N, K, 0);
            else run<8>(dA, ph, pm, pl, dC, M, N, K, 0);
        };
        launch();
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        // accuracy on sampled entries vs double and vs a float fmaf chain
        double worst = 0, worst32 = 0, scale = 0;
        for (int s = 0; s < 400; ++s) {
            const int i = rand() % M, j = rand() % N;
            double ref = 0;
            float ref32 = 0.f;
            for (int k = 0; k < K; ++k) {
                ref += (double)hA[(size_t)i * K + k] * hB[(size_t)k * N + j];
                ref32 = fmaf(hA[(size_t)i * K + k], hB[(size_t)k * N + j], ref32);
            }
            worst = fmax(worst, fabs(hC[(size_t)i * N + j] - ref));
            worst32 = fmax(worst32, fabs((double)ref32 - ref));
            scale = fmax(scale, fabs(ref));
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 10;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("terms=%d  M=%d N=%d K=%d  %.3f ms  %.1f TFLOP/s (fp32-equivalent)  max|err|=%.3e (fp32 fmaf chain %.3e) scale %.3f\n",
               nt, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, worst, worst32, scale);
    }
    // ---- variant 2
    hipLaunchKernelGGL(split_b_frag_kernel, dim3((unsigned)(((size_t)K * N + 255) / 256)), dim3(256), 0, 0, dB, K, N, ph, pm, pl);
    CK(hipDeviceSynchronize());
    {
        const size_t lds2 = (size_t)2 * 3 * BM * LDT * 2;
        CK(hipFuncSetAttribute((const void*)gemm2_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        auto launch = [&]() { hipLaunchKernelGGL(gemm2_kernel<6>, dim3(N / BN, M / BM), dim3(256), lds2, 0, dA, ph, pm, pl, dC, M, N, K); };
        launch();
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int s = 0; s < 400; ++s) {
            const int i = rand() % M, j = rand() % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * K + k] * hB[(size_t)k * N + j];
            worst = fmax(worst, fabs(hC[(size_t)i * N + j] - ref));
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int r = 0; r < 10; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        printf("variant2 (B fragments from global) terms=6  %.3f ms  %.1f TFLOP/s (fp32-equivalent)  max|err|=%.3e\n", ms,
               2.0 * M * N * K / ms / 1e9, worst);
    }
    return 0;
}
