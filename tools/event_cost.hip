// tools/event_cost.hip -- what does handing work to a second stream and back cost?  (NOT product code.)
// A chain of short kernels on one stream, against the same chain with every second kernel on another stream joined
// by events (record -> hipStreamWaitEvent each way): the price of a fork / join, the thing to know before
// moving the boundary launches of a pass beside the march on small meshes.
//
//   hipcc --offload-arch=gfx950 -O3 tools/event_cost.hip -o tools/event_cost && tools/event_cost
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define CK(x)                                                                                                       \
    do {                                                                                                            \
        hipError_t e__ = (x);                                                                                       \
        if (e__ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e__), __LINE__); return 1; } \
    } while (0)

__global__ void spin(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}

int main() {
    float* buf;
    CK(hipMalloc((void**)&buf, 4096));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int n = 2000;
    for (int iters : {200, 4000}) {  // kernels of ~2 us and ~25 us
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, a, buf, iters);
                if (mode == 0) {
                    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, a, buf + 512, iters);
                } else {
                    CK(hipEventRecord(fork, a));
                    CK(hipStreamWaitEvent(b, fork, 0));
                    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, b, buf + 512, iters);
                    CK(hipEventRecord(join, b));
                    if (mode == 2) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, a, buf, iters);  // overlaps with stream b
                    CK(hipStreamWaitEvent(a, join, 0));
                }
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
            const char* what[3] = {"two kernels, one stream", "second kernel on another stream, fork + join",
                                   "the same with a third kernel on the first stream meanwhile"};
            printf("kernel of %4d iterations: %-62s %.2f us per round\n", iters, what[mode], us);
        }
    }
    return 0;
}
