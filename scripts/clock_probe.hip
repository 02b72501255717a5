// Effective shader clock while another process/stream keeps the GPU busy: one wave spins for `ms` milliseconds of wall
// time (s_memrealtime, 100 MHz) and reports the shader-clock counter (s_memtime) advance per microsecond.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/clock_probe.hip -o sketchyscenecolorization_amd/lib/clock_probe_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void probe(unsigned long long* out, unsigned long long wall_ticks) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w = w0;
    while (w - w0 < wall_ticks) w = wall_clock64();
    const unsigned long long c1 = clock64();
    out[0] = c1 - c0;
    out[1] = w - w0;
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    unsigned long long* d;
    hipMalloc(&d, 16);
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 100000ULL * 200);   // 200 ms at 100 MHz
        unsigned long long h[2];
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("shader clock %.1f MHz (counter %llu over %.1f ms)\n", (double)h[0] / ((double)h[1] / 100.0), h[0], h[1] / 1e5);
        fflush(stdout);
    }
    return 0;
}
