// Do external event record / wait nodes work inside a stream-captured hipGraph on this ROCm?  A graph increments x, records an
// external event, later waits on a second external event and doubles x; between, an eager side stream waits for the first event,
// adds 10 and records the second.  Expected x after replay k: 22, 66, 154.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void add(float* p, float v) { p[0] += v; }
__global__ void mul(float* p, float v) { p[0] *= v; }
__global__ void spin(float* p, int n) { float a = p[1]; for (int i = 0; i < n; ++i) a = a * 1.0000001f + 1e-9f; p[1] = a; }
int main() {
    float* x; CK(hipMalloc(&x, 8)); CK(hipMemset(x, 0, 8));
    hipStream_t cap, side; CK(hipStreamCreate(&cap)); CK(hipStreamCreate(&side));
    hipEvent_t ready, done; CK(hipEventCreate(&ready)); CK(hipEventCreate(&done));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(add, 1, 1, 0, cap, x, 1.f);
    CK(hipEventRecordWithFlags(ready, cap, hipEventRecordExternal));
    hipLaunchKernelGGL(spin, 1, 1, 0, cap, x, 100000);
    CK(hipStreamWaitEvent(cap, done, hipEventWaitExternal));
    hipLaunchKernelGGL(mul, 1, 1, 0, cap, x, 2.f);
    CK(hipStreamEndCapture(cap, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("graph nodes: %zu\n", nn);
    for (int it = 0; it < 3; ++it) {
        CK(hipGraphLaunch(ge, cap));
        CK(hipStreamWaitEvent(side, ready, 0));
        hipLaunchKernelGGL(add, 1, 1, 0, side, x, 10.f);
        CK(hipEventRecord(done, side));
        CK(hipDeviceSynchronize());
        float h[2]; CK(hipMemcpy(h, x, 8, hipMemcpyDeviceToHost));
        printf("replay %d: x = %g (expected %g)\n", it, h[0], it == 0 ? 22.f : (it == 1 ? 66.f : 154.f));
    }
    return 0;
}
