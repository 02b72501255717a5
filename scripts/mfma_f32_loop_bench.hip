// What limits a ds_read + v_mfma_f32_32x32x2_f32 loop on gfx950?  Each variant runs the matrix-wave loop of the
// implicit-GEMM kernels (wave tile 64x64 = 4 accumulators, 16 k-steps per K-tile) with one ingredient added:
//   0: operands in registers, no LDS, no barrier        (pure matrix-pipe rate)
//   1: operands read from LDS (the kernels' ds_read2_b32 pattern), no barrier
//   2: variant 1 + one s_barrier per K-tile
//   3: variant 2 with operands read as ds_read_b128 (k-contiguous fragments)
// Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_f32_loop_bench.hip -o sketchyscenecolorization_amd/lib/mfma_loop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int iters, float seed) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int A_LD = 33, A_SZ = 128 * A_LD, B_SZ = 32 * 128;
    for (int i = tid; i < 2 * (A_SZ + B_SZ) + 128 * 36 * 2; i += 256) smem[i] = seed * (float)((i * 7) & 15);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float a0 = seed * lane, a1 = seed * (lane + 1), b0 = seed * 3, b1 = seed * 5;
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else if (V == 1 || V == 2) {
            const float* Ab = smem + cur * A_SZ + (wm * 64 + l31) * A_LD + lhi;
            const float* Bb = smem + 2 * A_SZ + cur * B_SZ + lhi * 128 + wn * 64 + l31;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                float av[2], bv[2];
                for (int i = 0; i < 2; ++i) av[i] = Ab[i * 32 * A_LD + kk * 2];
                for (int j = 0; j < 2; ++j) bv[j] = Bb[kk * 2 * 128 + j * 32];
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            if (V == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cur ^= 1;
        } else {
            // k-permuted fragments: lane (row, lhi) holds k = lhi*16 + kk for kk = 0..15 -> four b128 reads per row block
            constexpr int LD = 36;
            const float* Ab = smem + 2 * (A_SZ + B_SZ) + (wm * 64 + l31) * LD + lhi * 16;
            const float* Bb = smem + 2 * (A_SZ + B_SZ) + 128 * LD + (wn * 64 + l31) * LD + lhi * 16;
            f32x4 av[2][4], bv[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                for (int i = 0; i < 2; ++i) av[i][q] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LD + q * 4);
                for (int j = 0; j < 2; ++j) bv[j][q] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LD + q * 4);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    for (int i = 0; i < 2; ++i)
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q][e], bv[j][q][e], acc[i][j], 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(long)blockIdx.x * 256 + tid] = s;
}

// Staging variants: the matrix-wave loop of variant 2 plus, every K-tile, a fresh 32 x 128 B tile brought from global
// memory into the other LDS buffer:
//   4: through registers (global_load_dwordx4 -> ds_write_b128), loads issued one step ahead
//   5: directly (global_load_lds_dwordx4, no registers, no ds_write)
template <int V>
__global__ __launch_bounds__(256) void stage_kernel(float* out, const float* __restrict__ gB, int iters, float seed) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int A_LD = 33, A_SZ = 128 * A_LD, B_SZ = 32 * 128;
    for (int i = tid; i < 2 * (A_SZ + B_SZ); i += 256) smem[i] = seed * (float)((i * 7) & 15);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float* Bs = smem + 2 * A_SZ;
    // each workgroup streams its own 16 KB slices of gB (L2-resident after the first pass)
    const float* gsrc = gB + ((long)(blockIdx.x & 63) * 64) * B_SZ;
    f32x4 rb[4];
    int cur = 0;
    if (V == 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) rb[s] = *reinterpret_cast<const f32x4*>(gsrc + (s * 256 + tid) * 4);
    }
    for (int it = 0; it < iters; ++it) {
        const float* Ab = smem + cur * A_SZ + (wm * 64 + l31) * A_LD + lhi;
        const float* Bb = Bs + cur * B_SZ + lhi * 128 + wn * 64 + l31;
        const float* gnext = gsrc + (long)((it + 1) & 63) * B_SZ;
        if (V == 4) {
            // registers (tile it+1, loaded one step ago) -> LDS buffer cur^1, then issue the loads of tile it+2
#pragma unroll
            for (int s = 0; s < 4; ++s) *reinterpret_cast<f32x4*>(Bs + (cur ^ 1) * B_SZ + (s * 256 + tid) * 4) = rb[s];
            const float* g2 = gsrc + (long)((it + 2) & 63) * B_SZ;
#pragma unroll
            for (int s = 0; s < 4; ++s) rb[s] = *reinterpret_cast<const f32x4*>(g2 + (s * 256 + tid) * 4);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                __builtin_amdgcn_global_load_lds(gnext + (s * 256 + tid) * 4,
                                                 (__attribute__((address_space(3))) void*)(Bs + (cur ^ 1) * B_SZ + (s * 256 + wave * 64) * 4),
                                                 16, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av[2], bv[2];
            for (int i = 0; i < 2; ++i) av[i] = Ab[i * 32 * A_LD + kk * 2];
            for (int j = 0; j < 2; ++j) bv[j] = Bb[kk * 2 * 128 + j * 32];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (V == 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cur ^= 1;
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(long)blockIdx.x * 256 + tid] = s;
}

template <int V>
static void run_stage(int wg_per_cu, int iters, float* out, const float* gB) {
    const size_t lds = (2 * (128 * 33 + 32 * 128)) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(stage_kernel<V>, dim3(blocks), dim3(256), lds, 0, out, gB, iters, 1e-3f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(stage_kernel<V>, dim3(blocks), dim3(256), lds, 0, out, gB, iters, 1e-3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 64 * (2.0 * 32 * 32 * 2);
    printf("variant %d  %d workgroup(s)/CU  %8.3f ms  %6.1f TFLOP/s   (%s)\n", V, wg_per_cu, ms, flops / ms / 1e9,
           V == 4 ? "B tile staged through registers" : "B tile by global_load_lds");
}

template <int V>
static void run(int wg_per_cu, int iters, float* out) {
    const size_t lds = (2 * (128 * 33 + 32 * 128) + 128 * 36 * 2) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(loop_kernel<V>, dim3(blocks), dim3(256), lds, 0, out, iters, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(loop_kernel<V>, dim3(blocks), dim3(256), lds, 0, out, iters, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 64 * (2.0 * 32 * 32 * 2);
    printf("variant %d  %d workgroup(s)/CU  %8.3f ms  %6.1f TFLOP/s\n", V, wg_per_cu, ms, flops / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int iters = 2000;
    float* gB;
    (void)hipMalloc(&gB, (size_t)64 * 64 * 32 * 128 * sizeof(float) + 65536);
    (void)hipMemset(gB, 0, (size_t)64 * 64 * 32 * 128 * sizeof(float) + 65536);
    for (int w = 1; w <= 2; ++w) {
        run_stage<4>(w, iters, out, gB);
        run_stage<5>(w, iters, out, gB);
    }
    for (int w = 1; w <= 2; ++w) {
        run<0>(w, iters, out);
        run<1>(w, iters, out);
        run<2>(w, iters, out);
        run<3>(w, iters, out);
    }
    return 0;
}
