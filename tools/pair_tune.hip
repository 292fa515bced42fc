// tools/pair_tune.hip -- standalone pricing harness (NOT product code) for the engine's two-step pass
// (wayverb_amd/csrc/pair_kernels.hip.h): what does each piece of pair_march_kernel cost?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/pair_tune.hip -o tools/pair_tune
//   tools/pair_tune [n=1024] [iters=6]
//
// Fields are seeded noise, the pair map is the box's (inside everywhere but a 2-node shell), so the
// loads, stores and the arithmetic are the product's; the experiment switches (PX_*) drop or change one
// piece at a time.  Times are per launch = per TWO time steps.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../wayverb_amd/csrc/pair_kernels.hip.h"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e__ = (x);                                                              \
        if (e__ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

__global__ void init_kernel(double* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        p[i] = ((double)(h & 0xFFFF) / 65536.0 - 0.5) * 0.5;
    }
}

// box pair map: code 1 strictly inside the 2-node shell, 3 next to it, 2 on the boundary shell, 0 outside
__global__ void box_map_kernel(uint8_t* map, int n, int cls_pitch) {
    const int64_t n_bytes = (int64_t)cls_pitch * n * n;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bytes) return;
    const int xb = (int)(i % cls_pitch);
    const int64_t row = i / cls_pitch;
    const int y = (int)(row % n), z = (int)(row / n);
    uint32_t out = 0;
    for (int k = 0; k < 4; ++k) {
        const int x = xb * 4 + k;
        auto shell = [&](int c) { return c == 0 || c == n - 1 ? 0 : (c == 1 || c == n - 2 ? 1 : (c == 2 || c == n - 3 ? 2 : 3)); };
        const int sx = shell(x), sy = shell(y), sz = shell(z);
        const int m = sx < sy ? (sx < sz ? sx : sz) : (sy < sz ? sy : sz);
        const uint32_t code = m == 0 ? 0u : (m == 1 ? 2u : (m == 2 ? 3u : 1u));
        out |= code << (2 * k);
    }
    map[wv::cls_byte_index(xb * 4, y, z, n, cls_pitch)] = (uint8_t)out;
}

// strips of TWO rows, three waves per SIMD asked of the compiler (rows of 5-7 waves leave wave slots empty at two)
template <int X>
__global__ void __launch_bounds__(64 * wv::kPairMaxWaves) __attribute__((amdgpu_waves_per_eu(3, 3))) pair_march_ry2_kernel(const wv::PairArgs<double> a) {
    wv::pair_march_body<double, X, 0, false, 2>(a);
}
template <int X>
__global__ void __launch_bounds__(64 * wv::kPairMaxWaves) pair_march_ry2_free_kernel(const wv::PairArgs<double> a) {
    wv::pair_march_body<double, X, 0, false, 2>(a);
}

template <int X, int NWC = 0>
float time_variant(const wv::PairArgs<double>& a, unsigned grid, int iters, hipEvent_t e0, hipEvent_t e1) {
    for (int it = 0; it < iters + 2; ++it) {
        if (it == 2) CK(hipEventRecord(e0));
        hipLaunchKernelGGL((wv::pair_march_kernel<double, X, NWC>), dim3(grid), dim3(64u * (unsigned)a.nw), 0, 0, a);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 6;
    const int pitch = (n + 127) / 128 * 128;
    const int64_t N = (int64_t)pitch * n * n;
    double *A, *B, *O1, *O2;
    for (double** p : {&A, &B, &O1, &O2}) CK(hipMalloc((void**)p, N * 8 + 256));
    hipLaunchKernelGGL(init_kernel, dim3(4096), dim3(256), 0, 0, A, N, 1u);
    hipLaunchKernelGGL(init_kernel, dim3(4096), dim3(256), 0, 0, B, N, 2u);
    const int cls_pitch = pitch / 4;
    const int64_t map_bytes = (int64_t)cls_pitch * 4 * ((n + 3) / 4) * n;
    uint8_t* map;
    CK(hipMalloc((void**)&map, map_bytes + 16));
    CK(hipMemset(map, 0, map_bytes + 16));
    hipLaunchKernelGGL(box_map_kernel, dim3((unsigned)(((int64_t)cls_pitch * n * n + 255) / 256)), dim3(256), 0, 0, map, n, cls_pitch);
    int* flags;
    CK(hipMalloc((void**)&flags, 8));
    CK(hipMemset(flags, 0, 8));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    wv::PairArgs<double> a{};
    a.prev = A;
    a.cur = B;
    a.out1 = O1;
    a.out2 = O2;
    a.pair_map = map;
    a.flag1 = flags;
    a.flag2 = flags + 1;
    a.ny = n;
    a.nz = n;
    a.pitch = pitch;
    a.cls_pitch = cls_pitch;
    a.z_begin = 0;
    a.z_end = n;
    a.out1_z0 = 0;
    a.out1_z1 = n;
    a.nw = pitch / 128;
    a.strips = (n + 3) / 4;
    a.strips_per_xcd = (a.strips + 7) / 8;
    const double gnodes = 2.0 * (double)n * n * n / 1e6;
    for (int chunks : {1, 2, 4}) {
        a.chunks = chunks;
        a.zc = (n + chunks - 1) / chunks;
        a.chunks = (n + a.zc - 1) / a.zc;
        const unsigned grid = 8u * (unsigned)a.strips_per_xcd * (unsigned)a.chunks;
        printf("chunks %d (grid %u workgroups of %d waves)\n", a.chunks, grid, a.nw);
        float ms;
        ms = time_variant<0>(a, grid, iters, e0, e1);
        printf("  product                          %.3f ms  %.1f Gnode-updates/s\n", ms, gnodes / ms);
        ms = time_variant<wv::PX_NO_MAP>(a, grid, iters, e0, e1);
        printf("  no pair map                      %.3f ms  %.1f\n", ms, gnodes / ms);
        ms = time_variant<wv::PX_NO_FLAGS>(a, grid, iters, e0, e1);
        printf("  no inf/nan flags                 %.3f ms  %.1f\n", ms, gnodes / ms);
        ms = time_variant<wv::PX_NO_MAP | wv::PX_NO_FLAGS>(a, grid, iters, e0, e1);
        printf("  no map, no flags                 %.3f ms  %.1f\n", ms, gnodes / ms);
        if (n == 1024) {
            ms = time_variant<0, 8>(a, grid, iters, e0, e1);
            printf("  pitch / waves compile-time       %.3f ms  %.1f\n", ms, gnodes / ms);
            ms = time_variant<wv::PX_NO_MAP | wv::PX_NO_FLAGS, 8>(a, grid, iters, e0, e1);
            printf("  compile-time, no map, no flags   %.3f ms  %.1f\n", ms, gnodes / ms);
        }
        ms = time_variant<wv::PX_NO_MEMORY | wv::PX_NO_MAP>(a, grid, iters, e0, e1);
        printf("  no loads, no stores (instructions only) %.3f ms  %.1f\n", ms, gnodes / ms);
        ms = time_variant<wv::PX_NO_COMPUTE | wv::PX_NO_MAP>(a, grid, iters, e0, e1);
        printf("  loads and stores only (no arithmetic, no exchange) %.3f ms  %.1f\n", ms, gnodes / ms);
        {   // the same march in strips of two rows (no pair map: its row groups are four rows)
            wv::PairArgs<double> b = a;
            b.strips = (n + 1) / 2;
            b.strips_per_xcd = (b.strips + 7) / 8;
            const unsigned g2 = 8u * (unsigned)b.strips_per_xcd * (unsigned)b.chunks;
            for (int which = 0; which < 2; ++which) {
                for (int it = 0; it < iters + 2; ++it) {
                    if (it == 2) CK(hipEventRecord(e0));
                    if (which == 0)
                        hipLaunchKernelGGL((pair_march_ry2_kernel<wv::PX_NO_MAP>), dim3(g2), dim3(64u * (unsigned)b.nw), 0, 0, b);
                    else
                        hipLaunchKernelGGL((pair_march_ry2_free_kernel<wv::PX_NO_MAP>), dim3(g2), dim3(64u * (unsigned)b.nw), 0, 0, b);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float m2 = 0;
                CK(hipEventElapsedTime(&m2, e0, e1));
                m2 /= iters;
                printf("  strips of 2 rows, no map, %s  %.3f ms  %.1f\n", which == 0 ? "3 waves per SIMD asked:" : "registers as they come:", m2, gnodes / m2);
            }
        }
        ms = time_variant<wv::PX_PREV_NT>(a, grid, iters, e0, e1);
        printf("  previous loaded with nt hint     %.3f ms  %.1f\n", ms, gnodes / ms);
        ms = time_variant<wv::PX_STORE_CACHED>(a, grid, iters, e0, e1);
        printf("  stores without nt                %.3f ms  %.1f\n", ms, gnodes / ms);
    }
    return 0;
}
