// Does buffer_load_dwordx4 ... lds write ZEROS for lanes whose offset is out of range (or does it skip them)?
// hipcc --offload-arch=gfx950 -O2 scripts/glds_oob_test.hip -o /tmp/glds_oob && /tmp/glds_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* p, int nbytes, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) smem[i] = -7.f;      // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    // even lanes read their 16 bytes, odd lanes are sent out of range
    const unsigned voff = (lane & 1) ? 0x80000000u : (unsigned)(lane * 16);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(r), "s"(0u) : "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = smem[i];
}
int main() {
    float *d, *o;
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (float)(i + 1);
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, 1024, o);
    hipMemcpy(h.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("lane0: %g %g %g %g | lane1 (OOB): %g %g %g %g | lane2: %g ... lane3 (OOB): %g\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[12]);
    int zeros = 0, poison = 0, data = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { float v = h[l * 4 + j]; if (l & 1) { zeros += v == 0.f; poison += v == -7.f; } else data += v == (float)(l * 4 + j + 1); }
    printf("odd lanes: %d zero, %d poison (of 128); even lanes correct %d (of 128)\n", zeros, poison, data);
    return 0;
}
