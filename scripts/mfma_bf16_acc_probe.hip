// How exact is the fp32 accumulation inside v_mfma_f32_32x32x16_bf16, and what does the 3-way split cost in accuracy?
// One wavefront per 32x32 output tile, K from argv.  fp32 inputs a, b (gaussian); compared with a float64 dot product:
//   (1) "hh only" chain of MFMAs on the bf16-rounded inputs vs the float64 dot product of THOSE values: the matrix pipe's own
//       accumulation error, beside an fp32 fmaf chain over the same values;
//   (2) six products into ONE accumulator;  (3) hh into one accumulator, the five correction products into a second one, added at
//       the end;  (4) the fp32 fmaf chain on the fp32 inputs.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_bf16_acc_probe.hip -o sketchyscenecolorization_amd/lib/mfma_acc_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#include <cstring>
__device__ __host__ inline unsigned rne(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}
__device__ __host__ inline float asf(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ inline void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const unsigned hb = rne(x); const float r1 = x - asf(hb);
    const unsigned mb = rne(r1); const float r2 = r1 - asf(mb);
    h = hb >> 16; m = mb >> 16; l = rne(r2) >> 16;
}

// A [T][32][K] row-major, B [T][32][K] (column n, k contiguous); out [4][T][32][32]
__global__ void probe(const float* A, const float* B, float* out, int K, int T) {
    const int t = blockIdx.x, lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
    const float* a = A + ((long)t * 32 + l31) * K;
    const float* b = B + ((long)t * 32 + l31) * K;
    f32x16 c_hh = {0}, c_one = {0}, c_main = {0}, c_corr = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int e = 0; e < 8; ++e) {
            unsigned short h, m, l;
            split3(a[k0 + lhi * 8 + e], h, m, l); ah[e] = h; am[e] = m; al[e] = l;
            split3(b[k0 + lhi * 8 + e], h, m, l); bh[e] = h; bm[e] = m; bl[e] = l;
        }
        c_hh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c_hh, 0, 0, 0);
        c_one = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c_one, 0, 0, 0);
        c_one = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c_one, 0, 0, 0);
        c_one = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c_one, 0, 0, 0);
        c_one = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c_one, 0, 0, 0);
        c_one = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c_one, 0, 0, 0);
        c_one = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c_one, 0, 0, 0);
        c_corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c_corr, 0, 0, 0);
        c_corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c_corr, 0, 0, 0);
        c_corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c_corr, 0, 0, 0);
        c_corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c_corr, 0, 0, 0);
        c_corr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c_corr, 0, 0, 0);
        c_main = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c_main, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const long o = ((long)t * 32 + row) * 32 + l31;
        out[o] = c_hh[r];
        out[(long)T * 1024 + o] = c_one[r];
        out[(long)2 * T * 1024 + o] = c_main[r] + c_corr[r];
    }
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 512, T = 64, relu = argc > 2 ? atoi(argv[2]) : 0;
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hA((size_t)T * 32 * K), hB((size_t)T * 32 * K), hO((size_t)3 * T * 1024);
    for (auto& v : hA) { v = nd(g); if (relu && v < 0.f) v = 0.f; }
    for (auto& v : hB) v = 0.05f * nd(g);
    float *dA, *dB, *dO;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dO, hO.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(T), dim3(64), 0, 0, dA, dB, dO, K, T);
    hipMemcpy(hO.data(), dO, hO.size() * 4, hipMemcpyDeviceToHost);
    double e[5][2] = {{0}}, sc = 0;     // hh-mfma, hh-fmaf, one-acc, two-acc, fp32 fmaf: {sum sq, max}
    long cnt = 0;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                const float* a = &hA[((size_t)t * 32 + i) * K];
                const float* b = &hB[((size_t)t * 32 + j) * K];
                double ref = 0, refh = 0;
                float ch = 0.f, c32 = 0.f;
                for (int k = 0; k < K; ++k) {
                    ref += (double)a[k] * b[k];
                    const float ah = asf(rne(a[k])), bh = asf(rne(b[k]));
                    refh += (double)ah * bh;
                    ch = fmaf(ah, bh, ch);
                    c32 = fmaf(a[k], b[k], c32);
                }
                const long o = ((long)t * 32 + i) * 32 + j;
                const double d[5] = {hO[o] - refh, ch - refh, hO[(size_t)T * 1024 + o] - ref, hO[(size_t)2 * T * 1024 + o] - ref, c32 - ref};
                for (int q = 0; q < 5; ++q) { e[q][0] += d[q] * d[q]; e[q][1] = fmax(e[q][1], fabs(d[q])); }
                sc += ref * ref; ++cnt;
            }
    const char* nm[5] = {"hh chain on the bf16 MFMA (vs f64 of the rounded values)", "hh chain as fp32 fmaf", "six products, one accumulator",
                         "hh + corrections in two accumulators", "fp32 fmaf chain (exact-fp32 MFMA)"};
    printf("K=%d relu=%d  output rms %.3f\n", K, relu, sqrt(sc / cnt));
    for (int q = 0; q < 5; ++q) printf("  %-60s rms %.3e  max %.3e\n", nm[q], sqrt(e[q][0] / cnt), e[q][1]);
    return 0;
}
