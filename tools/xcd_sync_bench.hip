// tools/xcd_sync_bench.hip -- the neighbour hand-over of tools/neighbour_sync_bench.hip with every participating workgroup on ONE XCD
// (pricing harness, NOT product).   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xcd_sync_bench.hip -o tools/xcd_sync_bench
//
// Across XCDs a hand-over costs an L2 invalidate per consumer (buffer_inv sc1), which serialises per XCD: the resident form of
// small-mesh stepping loses to per-step launches for it (HISTORY.md, round 5).  Inside ONE XCD the 32 CUs share the L2: cacheable
// memory, stores acknowledged by the L2, only the per-CU L1 to get around.  Workgroups are dealt to the XCDs round-robin, so a launch
// of 8 K workgroups in which only those that find themselves on XCD 0 (hardware register XCC_ID) take part runs K of them there.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

// MODE 0: plain loads, nothing else; 1: buffer_inv sc0 after the wait; 2: loads with sc0 sc1 semantics through relaxed agent-scope atomics;
// 3: buffer_inv sc1 after the wait
template <int MODE>
__global__ void __launch_bounds__(256) chain_kernel(double* buf, unsigned* counters, unsigned* rank_counter, unsigned* xcc_seen, int iters, int n,
                                                    int K, unsigned* abort_flag) {
    __shared__ unsigned s_rank;
    if (xcc_id() != 0) return;
    if (threadIdx.x == 0) {
        s_rank = atomicAdd(rank_counter, 1u);
        atomicOr(xcc_seen, 1u << xcc_id());
    }
    __syncthreads();
    const int w = (int)s_rank;
    if (w >= K) return;
    const int lo = (w + K - 1) % K, hi = (w + 1) % K;
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x < 2) {
            const unsigned* c = counters + 32 * (threadIdx.x ? hi : lo);
            unsigned spins = 0;
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { *abort_flag = 1; break; }
            }
        }
        __syncthreads();
        if (MODE == 1) asm volatile("buffer_inv sc0" ::: "memory");
        if (MODE == 3) asm volatile("buffer_inv sc1" ::: "memory");
        const double* src = buf + (size_t)(it & 1) * K * n;
        double* dst = buf + (size_t)((it + 1) & 1) * K * n;
        double v[16];
        const int per = n / 256;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < per) {
                const int i = threadIdx.x + k * 256;
                if (MODE == 2)
                    v[k] = __hip_atomic_load(src + (size_t)lo * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                           __hip_atomic_load(src + (size_t)hi * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    v[k] = src[(size_t)lo * n + i] + src[(size_t)hi * n + i];
            }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < per) {
                const int i = threadIdx.x + k * 256;
                dst[(size_t)w * n + i] = v[k] - 4096.0 * floor(v[k] / 4096.0);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(counters + 32 * w, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const char* modes[4] = {"plain loads", "buffer_inv sc0", "agent-scope atomic loads", "buffer_inv sc1"};
    for (int mode = 0; mode < 4; ++mode)
        for (int K : {16, 32, 64, 96})
            for (int n : {1024, 4096}) {
                double* buf = nullptr;
                unsigned* counters = nullptr;
                const size_t bytes = (size_t)2 * K * n * sizeof(double);
                CK(hipMalloc((void**)&buf, bytes));
                CK(hipMalloc((void**)&counters, (32 * K + 8) * sizeof(unsigned)));
                unsigned *rank = counters + 32 * K, *seen = rank + 1, *abort_flag = rank + 2;
                std::vector<double> h((size_t)2 * K * n, 0.0);
                for (int w = 0; w < K; ++w)
                    for (int i = 0; i < n; ++i) h[(size_t)w * n + i] = (double)((w * 131 + i * 7) % 4096);
                float best = 1e30f;
                unsigned ranks = 0, xs = 0, aborted = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice));
                    CK(hipMemset(counters, 0, (32 * K + 8) * sizeof(unsigned)));
                    hipEvent_t e0, e1;
                    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                    CK(hipEventRecord(e0, 0));
                    const dim3 grid(8 * K), block(256);
                    if (mode == 0) hipLaunchKernelGGL(chain_kernel<0>, grid, block, 0, 0, buf, counters, rank, seen, iters, n, K, abort_flag);
                    if (mode == 1) hipLaunchKernelGGL(chain_kernel<1>, grid, block, 0, 0, buf, counters, rank, seen, iters, n, K, abort_flag);
                    if (mode == 2) hipLaunchKernelGGL(chain_kernel<2>, grid, block, 0, 0, buf, counters, rank, seen, iters, n, K, abort_flag);
                    if (mode == 3) hipLaunchKernelGGL(chain_kernel<3>, grid, block, 0, 0, buf, counters, rank, seen, iters, n, K, abort_flag);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
                    CK(hipMemcpy(&ranks, rank, 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(&xs, seen, 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(&aborted, abort_flag, 4, hipMemcpyDeviceToHost));
                }
                std::vector<double> a(h.begin(), h.begin() + (size_t)K * n), b((size_t)K * n);
                for (int it = 0; it < iters; ++it) {
                    for (int w = 0; w < K; ++w)
                        for (int i = 0; i < n; ++i) {
                            const double v = a[(size_t)((w + K - 1) % K) * n + i] + a[(size_t)((w + 1) % K) * n + i];
                            b[(size_t)w * n + i] = v - 4096.0 * floor(v / 4096.0);
                        }
                    a.swap(b);
                }
                std::vector<double> got((size_t)2 * K * n);
                CK(hipMemcpy(got.data(), buf, bytes, hipMemcpyDeviceToHost));
                const size_t off = (size_t)(iters & 1) * K * n;
                size_t wrong = 0;
                for (size_t i = 0; i < (size_t)K * n; ++i) wrong += got[off + i] != a[i];
                printf("%-26s K %3d x %5d doubles: %6.2f us per round; workgroups found on XCD 0: %u of %d launched (XCC ids seen: 0x%x)%s; %zu of %zu values wrong\n",
                       modes[mode], K, n, best * 1e3 / iters, ranks, 8 * K, xs, aborted ? " (a wait gave up)" : "", wrong, (size_t)K * n);
                CK(hipFree(buf));
                CK(hipFree(counters));
            }
    return 0;
}
