// tools/grid_sync_bench.hip -- what does a grid-wide barrier cost on this chip?  (pricing harness, NOT product)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/grid_sync_bench.hip -o tools/grid_sync_bench
// A cooperative launch of G workgroups of 256 threads runs `iters` rounds of {touch a little memory,
// grid.sync()}; also a hand-rolled sense-reversing barrier (one atomic per workgroup, device-scope fences).
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

namespace cg = cooperative_groups;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e__ = (x);                                                              \
        if (e__ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

__global__ void __launch_bounds__(256) cg_kernel(double* buf, int iters) {
    cg::grid_group grid = cg::this_grid();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double v = buf[i];
    for (int it = 0; it < iters; ++it) {
        buf[i] = v + 1.0;
        grid.sync();
        v = buf[(i + 256) % ((size_t)gridDim.x * blockDim.x)];  // a neighbour workgroup's value
    }
    buf[i] = v;
}

__device__ __forceinline__ void my_barrier(unsigned* count, unsigned* gen, unsigned n_wg, unsigned* abort_flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned arrived = __hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (arrived == n_wg) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) {  // ~ seconds: something is wrong, do not hang the GPU
                    *abort_flag = 1;
                    break;
                }
            }
        }
        __threadfence();
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) my_kernel(double* buf, int iters, unsigned* count, unsigned* gen, unsigned* abort_flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double v = buf[i];
    for (int it = 0; it < iters; ++it) {
        buf[i] = v + 1.0;
        my_barrier(count, gen, gridDim.x, abort_flag);
        v = __builtin_nontemporal_load(buf + (i + 256) % ((size_t)gridDim.x * blockDim.x));
    }
    buf[i] = v;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    int dev = 0, cus = 0, per_cu = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cg_kernel, 256, 0));
    printf("CUs %d, co-resident workgroups of 256 per CU (cg kernel): %d\n", cus, per_cu);
    unsigned* sync_words;
    CK(hipMalloc((void**)&sync_words, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int mult : {1, 2, 4, 8}) {
        if (mult > per_cu) break;
        const unsigned grid = (unsigned)(cus * mult);
        double* buf;
        CK(hipMalloc((void**)&buf, (size_t)grid * 256 * 8));
        CK(hipMemset(buf, 0, (size_t)grid * 256 * 8));
        float ms = 0;
        {
            int it = iters;
            void* args[] = {&buf, &it};
            CK(hipLaunchCooperativeKernel((void*)cg_kernel, dim3(grid), dim3(256), args, 0, 0));  // warm-up
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((void*)cg_kernel, dim3(grid), dim3(256), args, 0, 0));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        double h = 0;
        CK(hipMemcpy(&h, buf, 8, hipMemcpyDeviceToHost));
        printf("grid %5u  cooperative_groups grid.sync: %.2f us per round (value %g, expect %d)\n", grid, ms * 1e3 / iters, h,
               2 * iters);
        CK(hipMemset(buf, 0, (size_t)grid * 256 * 8));
        CK(hipMemset(sync_words, 0, 64));
        {
            int it = iters;
            unsigned *count = sync_words, *gen = sync_words + 1, *abort_flag = sync_words + 2;
            void* args[] = {&buf, &it, &count, &gen, &abort_flag};
            CK(hipLaunchCooperativeKernel((void*)my_kernel, dim3(grid), dim3(256), args, 0, 0));
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((void*)my_kernel, dim3(grid), dim3(256), args, 0, 0));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        unsigned words[3];
        CK(hipMemcpy(words, sync_words, 12, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&h, buf, 8, hipMemcpyDeviceToHost));
        printf("grid %5u  hand-rolled barrier:          %.2f us per round (value %g, expect %d, abort %u)\n", grid,
               ms * 1e3 / iters, h, 2 * iters, words[2]);
        CK(hipFree(buf));
    }
    return 0;
}
