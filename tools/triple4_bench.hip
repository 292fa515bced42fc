// tools/triple4_bench.hip -- pricing and self-check harness (NOT product code) for the three-step march of
// wayverb_amd/csrc/triple_kernels.hip.h: strips of four rows, the "lo" planes in LDS, row pointers in scalar registers.
//
// Interior only (map = null) and with a box's triple map; checks itself against three plain steps, bit for bit, then
// times n^3: the pass, its instruction stream alone (no loads, no stores), and for comparison what 32 B per node cost
// at the rate the two-step march runs at.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/triple4_bench.hip -o tools/triple4_bench
//   tools/triple4_bench [n=1024] [iters=5]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../wayverb_amd/csrc/triple_kernels.hip.h"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e__ = (x);                                                              \
        if (e__ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

template <typename Real>
__global__ void plain_step_kernel(const Real* prev, const Real* cur, Real* next, int nx, int ny, int nz, int pitch) {
    const int64_t n = (int64_t)pitch * ny * nz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % pitch);
        const int64_t q = i / pitch;
        const int y = (int)(q % ny), z = (int)(q / ny);
        if (x >= nx) {
            next[i] = 0;
            continue;
        }
        const int64_t plane = (int64_t)pitch * ny;
        Real s = Real(0) + (x > 0 ? cur[i - 1] : Real(0));
        s += (x + 1 < nx ? cur[i + 1] : Real(0));
        s += (y > 0 ? cur[i - pitch] : Real(0));
        s += (y + 1 < ny ? cur[i + pitch] : Real(0));
        s += (z > 0 ? cur[i - plane] : Real(0));
        s += (z + 1 < nz ? cur[i + plane] : Real(0));
        s = wv::div3(s);
        s -= prev[i];
        next[i] = s;
    }
}

template <typename Real>
__global__ void init_kernel(Real* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        p[i] = (Real)(((double)(h & 0xFFFF) / 65536.0 - 0.5) * 0.5);
    }
}

// a box's triple map: boundary nodes on the outermost shell (code 2), shell nodes on the two shells inside it (3), deep further in (1)
__global__ void box_map_kernel(uint8_t* map, int nx, int ny, int nz, int cls_pitch, int all_deep) {
    const int64_t n_bytes = (int64_t)cls_pitch * ny * nz;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bytes) return;
    const int xb = (int)(i % cls_pitch);
    const int64_t row = i / cls_pitch;
    const int y = (int)(row % ny), z = (int)(row / ny);
    uint32_t out = 0;
    for (int k = 0; k < 4; ++k) {
        const int x = xb * 4 + k;
        auto depth = [&](int c, int n) { return c < n - 1 - c ? c : n - 1 - c; };
        int d = depth(x, nx);
        d = min(d, depth(y, ny));
        d = min(d, depth(z, nz));
        const uint32_t code = x >= nx ? 0u : (all_deep ? 1u : (d == 0 ? 2u : (d <= 2 ? 3u : 1u)));
        out |= code << (2 * k);
    }
    map[wv::cls_byte_index(xb * 4, y, z, ny, cls_pitch)] = (uint8_t)out;
}

template <typename Real>
struct Bench {
    int lb = 8;  // bytes of a row per lane
    static constexpr int PX = 64 * (16 / (int)sizeof(Real));                     // the engine's pitch granularity
    int* flags = nullptr;
    bool full_first = true;
    bool dma = false;
    int wx() const { return 64 * (lb / (int)sizeof(Real)); }  // columns per wave

    wv::TripleArgs<Real> make_args(const Real* prev, const Real* cur, Real* o1, Real* o2, Real* o3, const uint8_t* map, int pitch, int ny, int nz, int chunks) {
        if (!flags) {
            CK(hipMalloc((void**)&flags, 4 * sizeof(int)));
            CK(hipMemset(flags, 0, 4 * sizeof(int)));
        }
        wv::TripleArgs<Real> a{};
        a.prev = prev;
        a.cur = cur;
        a.out1 = o1;
        a.out2 = o2;
        a.out3 = o3;
        a.map = map;
        a.suspect = flags;
        a.ny = ny;
        a.nz = nz;
        a.pitch = pitch;
        a.cls_pitch = pitch / 4;
        a.z_begin = 0;
        a.z_end = nz;
        uint8_t win[4][wv::kTripleMaxWindows];
        int widest = 0;
        a.windows = wv::triple_windows(pitch / wx(), win, &widest, full_first, dma ? wv::kTripleMaxWavesDma : wv::triple_max_waves(lb));
        if (a.windows < 0) {
            printf("row too long\n");
            exit(1);
        }
        a.nw = widest;
        for (int k = 0; k < a.windows; ++k) {
            a.win_first |= (uint64_t)win[0][k] << (8 * k);
            a.win_count |= (uint64_t)win[1][k] << (8 * k);
            a.win_store_lo |= (uint64_t)win[2][k] << (8 * k);
            a.win_store_hi |= (uint64_t)win[3][k] << (8 * k);
        }
        a.zc = (nz + chunks - 1) / chunks;
        a.chunks = (nz + a.zc - 1) / a.zc;
        a.strips = (ny + wv::kTripleRows - 1) / wv::kTripleRows;
        a.strips_per_xcd = (a.strips + 7) / 8;
        return a;
    }

    template <int X, bool DMA = false, int LB = 8>
    void launch_lb(const wv::TripleArgs<Real>& a) {
        static bool set = false;
        if (!set) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wv::triple_march_kernel<Real, X, DMA, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            set = true;
        }
        const unsigned grid = 8u * (unsigned)a.strips_per_xcd * (unsigned)a.chunks * (unsigned)(a.windows ? a.windows : 1);
        hipLaunchKernelGGL((wv::triple_march_kernel<Real, X, DMA, LB>), dim3(grid), dim3(64u * (unsigned)a.nw), wv::triple_lds_bytes(a.nw, DMA, LB), 0, a);
        CK(hipGetLastError());
    }
    template <int X>
    void launch_dense(const wv::TripleArgs<Real>& a) {
        static bool set = false;
        if (!set) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wv::triple_march_kernel<Real, X, false, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            set = true;
        }
        const unsigned grid = 8u * (unsigned)a.strips_per_xcd * (unsigned)a.chunks * (unsigned)(a.windows ? a.windows : 1);
        hipLaunchKernelGGL((wv::triple_march_kernel<Real, X, false, 8, true>), dim3(grid), dim3(64u * (unsigned)a.nw), wv::triple_lds_bytes(a.nw, false, 8), 0, a);
        CK(hipGetLastError());
    }
    bool dense = false;
    template <int X, bool DMA = false>
    void launch(const wv::TripleArgs<Real>& a) {
        if (dense) {
            if constexpr (!DMA) launch_dense<X>(a);
        } else if (lb == 16) {
            if constexpr (!DMA) launch_lb<X, false, 16>(a);
        } else {
            launch_lb<X, DMA, 8>(a);
        }
    }

    bool check(int nx, int ny, int nz, int chunks, bool with_map) {
        const int pitch = (nx + PX - 1) / PX * PX;
        const int64_t N = (int64_t)pitch * ny * nz;
        Real *A, *B, *T1, *T2, *T3, *O1, *O2, *O3;
        for (Real** p : {&A, &B, &T1, &T2, &T3, &O1, &O2, &O3}) CK(hipMalloc((void**)p, N * sizeof(Real) + 256));
        hipLaunchKernelGGL(init_kernel<Real>, dim3(1024), dim3(256), 0, 0, A, N, 11u);
        hipLaunchKernelGGL(init_kernel<Real>, dim3(1024), dim3(256), 0, 0, B, N, 22u);
        // (pad columns hold zeros in every field)
        if (pitch != nx) {
            std::vector<Real> h(N);
            for (Real* p : {A, B}) {
                CK(hipMemcpy(h.data(), p, N * sizeof(Real), hipMemcpyDeviceToHost));
                for (int64_t i = 0; i < N; ++i)
                    if (i % pitch >= nx) h[i] = 0;
                CK(hipMemcpy(p, h.data(), N * sizeof(Real), hipMemcpyHostToDevice));
            }
        }
        hipLaunchKernelGGL(plain_step_kernel<Real>, dim3(1024), dim3(256), 0, 0, A, B, T1, nx, ny, nz, pitch);
        hipLaunchKernelGGL(plain_step_kernel<Real>, dim3(1024), dim3(256), 0, 0, B, T1, T2, nx, ny, nz, pitch);
        hipLaunchKernelGGL(plain_step_kernel<Real>, dim3(1024), dim3(256), 0, 0, T1, T2, T3, nx, ny, nz, pitch);
        CK(hipMemset(O1, 0xFF, N * sizeof(Real)));
        CK(hipMemset(O2, 0xFF, N * sizeof(Real)));
        CK(hipMemset(O3, 0xFF, N * sizeof(Real)));
        uint8_t* map = nullptr;
        const int64_t map_bytes = (int64_t)(pitch / 4) * 4 * ((ny + 3) / 4) * nz;
        std::vector<uint8_t> hmap;
        {
            CK(hipMalloc((void**)&map, map_bytes + 16));
            CK(hipMemset(map, 0, map_bytes + 16));
            const int64_t n_bytes = (int64_t)(pitch / 4) * ny * nz;
            hipLaunchKernelGGL(box_map_kernel, dim3((unsigned)((n_bytes + 255) / 256)), dim3(256), 0, 0, map, nx, ny, nz, pitch / 4, with_map ? 0 : 1);
            hmap.resize(map_bytes);
            CK(hipMemcpy(hmap.data(), map, map_bytes, hipMemcpyDeviceToHost));
        }
        if (dma)
            launch<0, true>(make_args(A, B, O1, O2, O3, map, pitch, ny, nz, chunks));
        else
            launch<0>(make_args(A, B, O1, O2, O3, map, pitch, ny, nz, chunks));
        CK(hipDeviceSynchronize());
        std::vector<Real> t1(N), t2(N), t3(N), o1(N), o2(N), o3(N);
        CK(hipMemcpy(t1.data(), T1, N * sizeof(Real), hipMemcpyDeviceToHost));
        CK(hipMemcpy(t2.data(), T2, N * sizeof(Real), hipMemcpyDeviceToHost));
        CK(hipMemcpy(t3.data(), T3, N * sizeof(Real), hipMemcpyDeviceToHost));
        CK(hipMemcpy(o1.data(), O1, N * sizeof(Real), hipMemcpyDeviceToHost));
        CK(hipMemcpy(o2.data(), O2, N * sizeof(Real), hipMemcpyDeviceToHost));
        CK(hipMemcpy(o3.data(), O3, N * sizeof(Real), hipMemcpyDeviceToHost));
        int64_t bad1 = 0, bad2 = 0, bad3 = 0, first = -1, stored1 = 0;
        const Real zero = 0;
        // what the march promises: zeros at "none" nodes; t+1 at shell nodes; t+2 where nothing but plain nodes lies within one node,
        // t+3 within two (placeholders elsewhere: their owners overwrite them)
        auto code_at = [&](int x, int y, int z) -> uint32_t {
            if (x < 0 || x >= pitch || y < 0 || y >= ny || z < 0 || z >= nz) return 1u;
            return (hmap[wv::cls_byte_index(x, y, z, ny, pitch / 4)] >> ((x & 3) * 2)) & 3u;
        };
        std::vector<uint8_t> near1(N), near2(N);
        const int d[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
        for (int64_t i = 0; i < N; ++i) {
            const int x = (int)(i % pitch), y = (int)((i / pitch) % ny), z = (int)(i / ((int64_t)pitch * ny));
            bool n = !(code_at(x, y, z) & 1u);
            for (int p = 0; p < 6; ++p) n = n || !(code_at(x + d[p][0], y + d[p][1], z + d[p][2]) & 1u);
            near1[i] = n;
        }
        for (int64_t i = 0; i < N; ++i) {
            const int x = (int)(i % pitch), y = (int)((i / pitch) % ny), z = (int)(i / ((int64_t)pitch * ny));
            bool n = near1[i];
            for (int p = 0; p < 6; ++p) {
                const int xx = x + d[p][0], yy = y + d[p][1], zz = z + d[p][2];
                if (xx >= 0 && xx < pitch && yy >= 0 && yy < ny && zz >= 0 && zz < nz) n = n || near1[((int64_t)zz * ny + yy) * pitch + xx];
            }
            near2[i] = n;
        }
        for (int64_t i = 0; i < N; ++i) {
            const int x = (int)(i % pitch), y = (int)((i / pitch) % ny), z = (int)(i / ((int64_t)pitch * ny));
            const uint32_t code = code_at(x, y, z);
            if (code == 0) {
                if (std::memcmp(&zero, &o2[i], sizeof(Real))) ++bad2;
                if (std::memcmp(&zero, &o3[i], sizeof(Real))) ++bad3;
                continue;
            }
            if (!near1[i] && std::memcmp(&t2[i], &o2[i], sizeof(Real))) ++bad2;
            if (!near2[i] && std::memcmp(&t3[i], &o3[i], sizeof(Real))) {
                ++bad3;
                if (first < 0) first = i;
            }
            if (code == 3) {
                ++stored1;
                if (std::memcmp(&t1[i], &o1[i], sizeof(Real))) ++bad1;
            }
        }
        const bool ok = !bad1 && !bad2 && !bad3;
        printf("check %s%s %d-byte lanes %dx%dx%d (pitch %d), %d chunk(s), %s: t+1 %lld of %lld shell values, t+2 %lld and t+3 %lld of %lld values differ from three plain steps%s\n",
               sizeof(Real) == 8 ? "f64" : "f32", dma ? " dma" : "", lb, nx, ny, nz, pitch, chunks, with_map ? "box map" : "interior", (long long)bad1, (long long)stored1, (long long)bad2,
               (long long)bad3, (long long)N, ok ? " -- bit-identical" : "");
        if (first >= 0)
            printf("   first t+3 difference at x %lld y %lld z %lld\n", (long long)(first % pitch), (long long)((first / pitch) % ny), (long long)(first / ((int64_t)pitch * ny)));
        for (Real* p : {A, B, T1, T2, T3, O1, O2, O3}) CK(hipFree(p));
        if (map) CK(hipFree(map));
        return ok;
    }

    void time_it(int n, int iters) {
        const int pitch = (n + PX - 1) / PX * PX;
        const int64_t N = (int64_t)pitch * n * n;
        Real *A, *B, *O1, *O2, *O3;
        for (Real** p : {&A, &B, &O1, &O2, &O3}) CK(hipMalloc((void**)p, N * sizeof(Real) + 256));
        hipLaunchKernelGGL(init_kernel<Real>, dim3(4096), dim3(256), 0, 0, A, N, 1u);
        hipLaunchKernelGGL(init_kernel<Real>, dim3(4096), dim3(256), 0, 0, B, N, 2u);
        uint8_t *map = nullptr, *deep = nullptr;
        const int64_t map_bytes = (int64_t)(pitch / 4) * 4 * ((n + 3) / 4) * n;
        CK(hipMalloc((void**)&map, map_bytes + 16));
        CK(hipMemset(map, 0, map_bytes + 16));
        CK(hipMalloc((void**)&deep, map_bytes + 16));
        CK(hipMemset(deep, 0, map_bytes + 16));
        const int64_t n_bytes = (int64_t)(pitch / 4) * n * n;
        hipLaunchKernelGGL(box_map_kernel, dim3((unsigned)((n_bytes + 255) / 256)), dim3(256), 0, 0, map, n, n, n, pitch / 4, 0);
        hipLaunchKernelGGL(box_map_kernel, dim3((unsigned)((n_bytes + 255) / 256)), dim3(256), 0, 0, deep, n, n, n, pitch / 4, 1);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int pass = 0; pass < (sizeof(Real) == 4 ? 6 : 4); ++pass) {
            const int chunks = 1 << (pass % 2);
            full_first = false;
            lb = (pass >= 2 && pass < 4) ? 16 : 8;
            dense = pass >= 4;
            if (n / chunks < 16) continue;
            for (int variant = 0; variant < 6; ++variant) {
                if (variant == 3 || variant == 4) continue;
                if (lb == 16 && variant == 3) continue;
                dma = variant == 3;
                const wv::TripleArgs<Real> a = make_args(A, B, O1, O2, O3, (variant == 1 || variant >= 3) ? map : deep, pitch, n, n, chunks);
                for (int it = 0; it < iters + 2; ++it) {
                    if (it == 2) CK(hipEventRecord(e0));
                    if (variant == 2)
                        launch<wv::TX_NO_MEMORY>(a);
                    else if (variant == 3)
                        launch<0, true>(a);
                    else if (variant == 4)
                        launch<wv::TX_STORE_CACHED>(a);
                    else if (variant == 5)
                        launch<wv::TX_WRAP_Z>(a);
                    else
                        launch<0>(a);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= iters;
                printf("%s %d-byte lanes%s %d^3, %d chunk(s), %d window(s) of <= %d waves, %s: %.3f ms per pass of THREE steps = %.3f ms per step = %.1f Gnode-updates/s (%d B per node: %.0f GB/s)\n",
                       sizeof(Real) == 8 ? "f64" : "f32", lb, dense ? " (four waves per SIMD)" : "", n, a.chunks, a.windows ? a.windows : 1, a.nw,
                       variant == 5 ? "three-step pass, box map, every plane wrapped onto two (cache-resident traffic)" : variant == 4 ? "three-step pass, box map, stores without the nt hint" : variant == 3 ? "three-step pass, box map, `previous` by LDS-DMA" : variant == 2 ? "instructions only (no loads, no stores)" : (variant == 1 ? "three-step pass, box map" : "three-step pass, interior only"), ms, ms / 3,
                       3.0 * N / ms / 1e6, (int)(4 * sizeof(Real)), 4.0 * sizeof(Real) * N / ms / 1e6);
            }
        }
        for (Real* p : {A, B, O1, O2, O3}) CK(hipFree(p));
        CK(hipFree(map));
        CK(hipFree(deep));
    }
};

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 5;
    const bool f32_too = argc > 3 && atoi(argv[3]) != 0;
    Bench<double> d;
    bool ok = true;
    for (int with_map = 0; with_map < 2; ++with_map) {
        ok = d.check(256, 22, 19, 1, with_map) && ok;
        ok = d.check(128, 9, 40, 3, with_map) && ok;
        ok = d.check(1024, 12, 14, 2, with_map) && ok;
        ok = d.check(100, 37, 23, 2, with_map) && ok;
        ok = d.check(384, 64, 33, 1, with_map) && ok;
        ok = d.check(1000, 20, 12, 1, with_map) && ok;   // two windows, pad columns
        ok = d.check(2048, 8, 11, 1, with_map) && ok;    // four windows
    }
    d.dma = true;
    for (int with_map = 0; with_map < 2; ++with_map) {
        ok = d.check(256, 22, 19, 1, with_map) && ok;
        ok = d.check(128, 9, 40, 3, with_map) && ok;
        ok = d.check(100, 37, 23, 2, with_map) && ok;
        ok = d.check(1000, 20, 12, 1, with_map) && ok;
        ok = d.check(2048, 8, 11, 1, with_map) && ok;
        ok = d.check(640, 256, 70, 1, with_map) && ok;
    }
    d.dma = false;
    d.lb = 16;
    for (int with_map = 0; with_map < 2; ++with_map) {
        ok = d.check(256, 22, 19, 1, with_map) && ok;
        ok = d.check(128, 9, 40, 3, with_map) && ok;
        ok = d.check(1024, 12, 14, 2, with_map) && ok;
        ok = d.check(100, 37, 23, 2, with_map) && ok;
        ok = d.check(1000, 20, 12, 1, with_map) && ok;
        ok = d.check(2048, 8, 11, 1, with_map) && ok;
        ok = d.check(640, 256, 70, 1, with_map) && ok;
    }
    d.lb = 8;
    Bench<float> s;
    if (f32_too) {
        s.dma = true;
        ok = s.check(500, 21, 27, 2, true) && ok;
        ok = s.check(1024, 64, 40, 1, true) && ok;
        s.dma = false;
        s.lb = 16;
        ok = s.check(500, 21, 27, 2, true) && ok;
        ok = s.check(2048, 64, 40, 1, true) && ok;
        s.lb = 8;
        s.dense = true;
        ok = s.check(500, 21, 27, 2, true) && ok;
        ok = s.check(1024, 64, 40, 1, true) && ok;
        ok = s.check(2048, 24, 19, 1, true) && ok;
        s.dense = false;
        ok = s.check(256, 22, 19, 1, false) && ok;
        ok = s.check(500, 21, 27, 2, true) && ok;
    }
    if (!ok) printf("SELF-CHECK FAILED\n");
    d.time_it(n, iters);
    if (f32_too) s.time_it(n, iters);
    return ok ? 0 : 1;
}
