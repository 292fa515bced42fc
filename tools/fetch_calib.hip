// tools/fetch_calib.hip -- what does rocprofv3's FETCH_SIZE count per access pattern on gfx950?  (NOT product code.)
//
// The guide calibrates it for wide coalesced reads only (16 B per lane: the counter shows half the bytes).
// boundary_kernel reads 8 B per lane, coalesced on y / z walls and one word per cache line on x walls, so the same
// question is put for those patterns: every kernel below reads a known set of bytes / lines of a 2 GiB array
// (8 x the Infinity Cache) exactly once.
//
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- tools/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e__ = (x);                                                                   \
        if (e__ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e__), __LINE__); return 1; } \
    } while (0)

// every thread reads WORDS consecutive doubles at byte offset i * STRIDE
template <int STRIDE, int WORDS>
__global__ void __launch_bounds__(256) read_kernel(const char* base, size_t n, double* sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* p = reinterpret_cast<const double*>(base + i * STRIDE);
    double s = 0;
#pragma unroll
    for (int w = 0; w < WORDS; ++w) s += p[w];
    if (s == 12345.678) *sink = s;
}

int main() {
    const size_t bytes = 2ull << 30;
    char* a;
    double* sink;
    CK(hipMalloc((void**)&a, bytes));
    CK(hipMalloc((void**)&sink, 8));
    CK(hipMemset(a, 0, bytes));
    CK(hipDeviceSynchronize());
    auto go = [&](auto kernel, size_t threads) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, a, threads, sink);
        return hipDeviceSynchronize();
    };
    // name                      bytes touched usefully           lines (128 B) touched
    CK(go(read_kernel<16, 2>, bytes / 16));    // 16 B per lane, coalesced:   2 GiB      all
    CK(go(read_kernel<8, 1>, bytes / 8));      //  8 B per lane, coalesced:   2 GiB      all
    CK(go(read_kernel<128, 1>, bytes / 128));  //  8 B of every 128-B line: 128 MiB      all
    CK(go(read_kernel<64, 1>, bytes / 64));    //  8 B of every 64-B half:  256 MiB      all
    CK(go(read_kernel<32, 1>, bytes / 32));    //  8 B of every 32-B sector: 512 MiB     all
    CK(go(read_kernel<256, 1>, bytes / 256));  //  8 B of every other line:   64 MiB     half
    CK(go(read_kernel<128, 3>, bytes / 128));  // 24 B of every line (an x-wall node's x-1, x, x+1)
    printf("done\n");
    return 0;
}
