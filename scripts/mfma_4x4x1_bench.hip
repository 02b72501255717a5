// v_mfma_f32_4x4x1_16b_f32 on gfx950: cycles per instruction with NACC independent accumulators, one wave per SIMD.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/m4 scripts/mfma_4x4x1_bench.hip && /tmp/m4
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(64) void k4(float* out, const float* in, int iters, long long* cyc) {
    f4 acc[NACC];
    float a = in[threadIdx.x], b = in[64 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(64) void k16(float* out, const float* in, int iters, long long* cyc) {
    f4 acc[NACC];
    float a = in[threadIdx.x], b = in[64 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

template <class K>
static void run(const char* name, K kern, float* out, float* in, long long* cyc) {
    const int iters = 4096;
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, out, in, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, out, in, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %8.2f counter ticks / instr (100 MHz counter: x ~24 = shader cycles), %.3f us / instr wall\n", name,
           (double)c / (iters * 16.0), ms * 1e3 / (iters * 16.0));
}
int main() {
    float *out, *in;
    long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&in, 4096); hipMalloc(&cyc, 8);
    hipMemset(in, 0, 4096);
    run("4x4x1_16b, 1 accumulator", k4<1>, out, in, cyc);
    run("4x4x1_16b, 4 accumulators", k4<4>, out, in, cyc);
    run("4x4x1_16b, 16 accumulators", k4<16>, out, in, cyc);
    run("16x16x4, 1 accumulator", k16<1>, out, in, cyc);
    run("16x16x4, 4 accumulators", k16<4>, out, in, cyc);
    return 0;
}
