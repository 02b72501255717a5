// LDS victim: workgroups that fill their LDS with a pattern, spin, and check it -- run beside another kernel to see whether that
// kernel writes outside its own LDS allocation.   extern "C" int victim_run(int blocks, int lds_bytes, int spins, int* bad_dev, void* stream)
#include <hip/hip_runtime.h>
extern "C" __global__ void victim_kernel(int lds_words, int spins, int* bad) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) lds[i] = 0xA5000000u ^ (unsigned)i ^ (blockIdx.x << 12);
    __syncthreads();
    int nbad = 0;
    for (int s = 0; s < spins; ++s) {
        for (int i = threadIdx.x; i < lds_words; i += blockDim.x)
            if (lds[i] != (0xA5000000u ^ (unsigned)i ^ (blockIdx.x << 12))) { ++nbad; if (nbad == 1) atomicMax(&bad[1], i); }
        __builtin_amdgcn_s_sleep(20);
    }
    if (nbad) atomicAdd(&bad[0], nbad);
}
extern "C" int victim_run(int blocks, int lds_bytes, int spins, int* bad_dev, void* stream) {
    static bool set = false;
    if (!set) { hipFuncSetAttribute((const void*)victim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    hipLaunchKernelGGL(victim_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, lds_bytes / 4, spins, bad_dev);
    return (int)hipGetLastError();
}
