// How far apart must two v_mfma_f32_32x32x2_f32 on the SAME accumulator be?  Pure register loop, NACC accumulators used round
// robin, 1-3 waves per SIMD (blocks of 256 threads, 1-3 blocks per CU).  Prints the fraction of the 157.3 TFLOP/s matrix peak.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_dep_bench.hip -o sketchyscenecolorization_amd/lib/mfma_dep_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float a0 = seed * threadIdx.x, b0 = seed * 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 64 / NACC; ++kk)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(int blocks_per_cu) {
    const int ncu = 256, iters = 2000;
    float* out;
    hipMalloc(&out, ncu * 3 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(ncu * blocks_per_cu), dim3(256), 0, 0, out, 10, 0.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(ncu * blocks_per_cu), dim3(256), 0, 0, out, iters, 0.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)ncu * blocks_per_cu * 4 * iters * 64 * 2.0 * 32 * 32 * 2;
    printf("accumulators %d  waves/SIMD %d  %.1f TFLOP/s  %.3f of peak\n", NACC, blocks_per_cu, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
    hipFree(out);
}

int main() {
    for (int b = 1; b <= 3; ++b) { run<1>(b); run<2>(b); run<4>(b); }
    return 0;
}
