// tools/neighbour_sync_bench.hip -- what does a step cost when workgroups only wait for their two NEIGHBOURS?  (pricing harness, NOT product)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/neighbour_sync_bench.hip -o tools/neighbour_sync_bench
//
// Round 2 priced a persistent "all steps in one launch" kernel for small meshes with a GRID-WIDE barrier per step: 27 us for 256
// workgroups, the arrivals serialised on one counter (tools/grid_sync_bench.hip) -- two whole 64^3 steps.  A z-decomposition needs no
// such thing: workgroup w, owner of a few planes, needs the planes of w - 1 and w + 1 only.  Here: G persistent workgroups in a chain,
// each owning N doubles per time level; a round = wait until both neighbours have published the previous round, read their values,
// write one's own, fence, publish one's own counter.  The recurrence v[w] <- v[w-1] + v[w+1] (mod arithmetic on small integers kept
// exact in doubles) is checked against the host, so a stale read shows.  Memory kinds: coarse-grained (hipMalloc: cross-XCD
// coherence by L2 write-back / invalidate at the fences), fine-grained, uncached.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e__ = (x);                                                              \
        if (e__ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

// the same without agent-scope fences: for UNCACHED memory, whose stores go through to the memory side and whose loads do not
// stop in a cache -- "my stores have been acknowledged" (s_waitcnt) is all the release there is, and relaxed atomics carry the counters
template <int MODE>
__global__ void __launch_bounds__(256) chain_nofence_kernel(double* buf, unsigned* counters, int iters, int n, unsigned* abort_flag) {
    const int w = blockIdx.x, G = gridDim.x;
    const int lo = (w + G - 1) % G, hi = (w + 1) % G;
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x < 2) {
            const unsigned* c = counters + 32 * (threadIdx.x ? hi : lo);
            unsigned spins = 0;
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) {
                    *abort_flag = 1;
                    break;
                }
            }
        }
        __syncthreads();
        if (MODE == 2) asm volatile("buffer_inv sc1" ::: "memory");
        if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const double* src = buf + (size_t)(it & 1) * G * n;
        double* dst = buf + (size_t)((it + 1) & 1) * G * n;
        // all loads first, then all stores (the real units do the same): 16 or 64 values per lane
        double v[64];
        const int per = n / 256;
#pragma unroll
        for (int k = 0; k < 64; ++k)
            if (k < per) {
                const int i = threadIdx.x + k * 256;
                v[k] = MODE == 0 ? __builtin_nontemporal_load(src + (size_t)lo * n + i) + __builtin_nontemporal_load(src + (size_t)hi * n + i)
                                 : src[(size_t)lo * n + i] + src[(size_t)hi * n + i];
            }
#pragma unroll
        for (int k = 0; k < 64; ++k)
            if (k < per) {
                const int i = threadIdx.x + k * 256;
                const double o = v[k] - 4096.0 * floor(v[k] / 4096.0);
                if (MODE == 0) __builtin_nontemporal_store(o, dst + (size_t)w * n + i);
                else dst[(size_t)w * n + i] = o;
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(counters + 32 * w, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void __launch_bounds__(256) chain_kernel(double* buf, unsigned* counters, int iters, int n, unsigned* abort_flag) {
    const int w = blockIdx.x, G = gridDim.x;
    const int lo = (w + G - 1) % G, hi = (w + 1) % G;
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x < 2) {
            const unsigned* c = counters + 32 * (threadIdx.x ? hi : lo);  // (one counter per 128-byte line)
            unsigned spins = 0;
            while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) {
                    *abort_flag = 1;
                    break;
                }
            }
        }
        __syncthreads();
        __threadfence();  // (acquire side for the other lanes)
        const double* src = buf + (size_t)(it & 1) * G * n;
        double* dst = buf + (size_t)((it + 1) & 1) * G * n;
        for (int i = threadIdx.x; i < n; i += 256) {
            const double v = src[(size_t)lo * n + i] + src[(size_t)hi * n + i];
            dst[(size_t)w * n + i] = v - 4096.0 * floor(v / 4096.0);
        }
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(counters + 32 * w, (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const char* kinds[8] = {"coarse-grained (hipMalloc)", "fine-grained", "uncached", "uncached, no fences, nt", "coarse, no fences (WRONG?)",
                            "uncached, no fences, plain", "uncached, plain + buffer_inv sc1", "uncached, plain + acquire fence"};
    for (int kind = 3; kind < 8; ++kind) {
        for (int G : {64, 128, 256}) {
            for (int n : {4096, 16384}) {
                double* buf = nullptr;
                unsigned *counters = nullptr, *abort_flag = nullptr;
                const size_t bytes = (size_t)2 * G * n * sizeof(double);
                if (kind == 0 || kind == 4) CK(hipMalloc((void**)&buf, bytes));
                else if (kind >= 5) CK(hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocUncached));
                else CK(hipExtMallocWithFlags((void**)&buf, bytes, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
                if (kind >= 3) CK(hipExtMallocWithFlags((void**)&counters, 32 * G * sizeof(unsigned) + sizeof(unsigned), hipDeviceMallocUncached));
                else CK(hipMalloc((void**)&counters, 32 * G * sizeof(unsigned) + sizeof(unsigned)));
                abort_flag = counters + 32 * G;
                std::vector<double> h((size_t)2 * G * n, 0.0);
                for (int w = 0; w < G; ++w)
                    for (int i = 0; i < n; ++i) h[(size_t)w * n + i] = (double)((w * 131 + i * 7) % 4096);
                float best = 1e30f;
                bool ok = true;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice));
                    CK(hipMemset(counters, 0, 32 * G * sizeof(unsigned) + sizeof(unsigned)));
                    hipEvent_t e0, e1;
                    CK(hipEventCreate(&e0));
                    CK(hipEventCreate(&e1));
                    CK(hipEventRecord(e0, 0));
                    if (kind == 5) hipLaunchKernelGGL(chain_nofence_kernel<1>, dim3(G), dim3(256), 0, 0, buf, counters, iters, n, abort_flag);
                    else if (kind == 6) hipLaunchKernelGGL(chain_nofence_kernel<2>, dim3(G), dim3(256), 0, 0, buf, counters, iters, n, abort_flag);
                    else if (kind == 7) hipLaunchKernelGGL(chain_nofence_kernel<3>, dim3(G), dim3(256), 0, 0, buf, counters, iters, n, abort_flag);
                    else if (kind >= 3) hipLaunchKernelGGL(chain_nofence_kernel<0>, dim3(G), dim3(256), 0, 0, buf, counters, iters, n, abort_flag);
                    else hipLaunchKernelGGL(chain_kernel, dim3(G), dim3(256), 0, 0, buf, counters, iters, n, abort_flag);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                    CK(hipEventDestroy(e0));
                    CK(hipEventDestroy(e1));
                }
                // check column 0..7 of every workgroup against the host recurrence
                std::vector<double> a(h.begin(), h.begin() + (size_t)G * n), b((size_t)G * n);
                for (int it = 0; it < iters; ++it) {
                    for (int w = 0; w < G; ++w)
                        for (int i = 0; i < 8; ++i) {
                            const double v = a[(size_t)((w + G - 1) % G) * n + i] + a[(size_t)((w + 1) % G) * n + i];
                            b[(size_t)w * n + i] = v - 4096.0 * floor(v / 4096.0);
                        }
                    a.swap(b);
                }
                std::vector<double> got((size_t)2 * G * n);
                CK(hipMemcpy(got.data(), buf, bytes, hipMemcpyDeviceToHost));
                unsigned aborted = 0;
                CK(hipMemcpy(&aborted, abort_flag, sizeof(unsigned), hipMemcpyDeviceToHost));
                const size_t off = (size_t)(iters & 1) * G * n;
                for (int w = 0; w < G && ok; ++w)
                    for (int i = 0; i < 8; ++i) ok = ok && got[off + (size_t)w * n + i] == a[(size_t)w * n + i];
                printf("%-28s G %3d x %5d doubles: %.2f us per round  %s%s\n", kinds[kind], G, n, best * 1e3 / iters, ok ? "values correct" : "VALUES WRONG",
                       aborted ? " (a wait gave up)" : "");
                CK(hipFree(buf));
                CK(hipFree(counters));
            }
        }
    }
    return 0;
}
