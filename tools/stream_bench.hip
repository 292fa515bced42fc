// tools/stream_bench.hip -- standalone pricing harness for the streaming kernel (NOT product code).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/stream_bench.hip -o tools/stream_bench
//   tools/stream_bench [n=1024] [iters=20]
//
// Times (HIP events) (a) plain bandwidth kernels with the same 2-read/1-write mix as the update,
// (b) the march kernel across tile shapes / z-chunk counts, (c) ablations that remove one piece
// of the kernel at a time (results of those are wrong on purpose: they price the piece).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../wayverb_amd/csrc/stream_kernels.hip.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e__ = (x);                                                      \
        if (e__ != hipSuccess) {                                                   \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void init_kernel(double* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        p[i] = ((double)(h & 0xFFFF) / 65536.0 - 0.5) * 0.5;
    }
}

// box class map: 0 on the outer layer, 2 on the shell, 1 inside
__global__ void cls_kernel(uint8_t* cls, int nx, int ny, int nz, int pitch) {
    const int64_t n = (int64_t)pitch * ny * nz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int xb = (int)(i % pitch);
        const int64_t row = i / pitch;
        const int y = (int)(row % ny), z = (int)(row / ny);
        uint32_t byte = 0;
        for (int j = 0; j < 4; ++j) {
            const int x = xb * 4 + j;
            if (x >= nx) break;
            uint32_t c = 1;
            if (x == 0 || y == 0 || z == 0 || x == nx - 1 || y == ny - 1 || z == nz - 1) c = 0;
            else if (x == 1 || y == 1 || z == 1 || x == nx - 2 || y == ny - 2 || z == nz - 2) c = 2;
            byte |= c << (2 * j);
        }
        cls[wv::cls_byte_index(xb * 4, y, z, ny, pitch)] = (uint8_t)byte;
    }
}

// ---- bandwidth references ------------------------------------------------------------------------
template <bool NT>
__global__ void __launch_bounds__(256) copy_kernel(v2d* __restrict__ dst, const v2d* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        v2d v = NT ? __builtin_nontemporal_load(src + i) : src[i];
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
// prev[i] = cur[i] - prev[i]: the update's 2R + 1W mix, in place, perfectly streaming
template <bool NT>
__global__ void __launch_bounds__(256) triad_kernel(v2d* __restrict__ prev, const v2d* __restrict__ cur, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        v2d c = NT ? __builtin_nontemporal_load(cur + i) : cur[i];
        v2d p = NT ? __builtin_nontemporal_load(prev + i) : prev[i];
        v2d o = c - p;
        if (NT) __builtin_nontemporal_store(o, prev + i);
        else prev[i] = o;
    }
}

template <int U>
__global__ void __launch_bounds__(256) triad_chunk_kernel(v2d* __restrict__ prev, const v2d* __restrict__ cur, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
    v2d c[U], p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = base + (int64_t)u * 256;
        if (i < n) {
            c[u] = cur[i];
            p[u] = __builtin_nontemporal_load(prev + i);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = base + (int64_t)u * 256;
        if (i < n) __builtin_nontemporal_store(c[u] - p[u], prev + i);
    }
}

// pattern probes: one 4 KB chunk (256 lanes x 16 B) per block-iteration, same bytes as the triad
//   MODE 0: block b -> chunk b                       (linear order; = triad_chunk<1>)
//   MODE 1: block b -> chunk permuted plane-major    (consecutive blocks are `stride` chunks apart)
//   MODE 2: block handles LOOP consecutive chunks one after the other (serial, no prefetch)
//   MODE 3: like 2 but software-pipelined (next chunk's loads issued before this chunk's store)
//   MODE 4: like 0 but prev[i] is read one chunk-plane (8 MB) ahead of where the block stores
template <int MODE, int LOOP>
__global__ void __launch_bounds__(256) probe_kernel(v2d* __restrict__ prev, const v2d* __restrict__ cur, int64_t n,
                                                    int64_t stride) {
    const int64_t chunks = n / 256;
    if (MODE == 0 || MODE == 1) {
        int64_t ch = blockIdx.x;
        if (MODE == 1) ch = (ch % stride) * (chunks / stride) + ch / stride;
        const int64_t i = ch * 256 + threadIdx.x;
        v2d c = cur[i];
        v2d p = __builtin_nontemporal_load(prev + i);
        __builtin_nontemporal_store(c - p, prev + i);
    } else if (MODE == 2) {
        for (int k = 0; k < LOOP; ++k) {
            const int64_t i = ((int64_t)blockIdx.x * LOOP + k) * 256 + threadIdx.x;
            v2d c = cur[i];
            v2d p = __builtin_nontemporal_load(prev + i);
            __builtin_nontemporal_store(c - p, prev + i);
        }
    } else if (MODE == 3) {
        int64_t i = ((int64_t)blockIdx.x * LOOP) * 256 + threadIdx.x;
        v2d c = cur[i];
        v2d p = __builtin_nontemporal_load(prev + i);
        for (int k = 0; k < LOOP; ++k) {
            v2d cn = c, pn = p;
            if (k + 1 < LOOP) {
                cn = cur[i + 256];
                pn = __builtin_nontemporal_load(prev + i + 256);
            }
            __builtin_nontemporal_store(c - p, prev + i);
            c = cn;
            p = pn;
            i += 256;
        }
    }
}

// stride probe: one wave per block; the wave touches U pieces of 1 KB that are `stride_vec` 16-byte
// vectors apart (all issued together), pieces tile memory exactly once over the grid.
template <int U>
__global__ void __launch_bounds__(64) stride_probe_kernel(v2d* __restrict__ prev, const v2d* __restrict__ cur,
                                                          int64_t stride_vec) {
    const int64_t per = stride_vec / 64;                 // 1 KB pieces per stride
    const int64_t b = blockIdx.x;
    const int64_t base = (b / per) * (U * stride_vec) + (b % per) * 64 + threadIdx.x;
    v2d c[U], p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        c[u] = cur[base + u * stride_vec];
        p[u] = __builtin_nontemporal_load(prev + base + u * stride_vec);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(c[u] - p[u], prev + base + u * stride_vec);
}

struct Ctx {
    double *a, *b;
    uint8_t* cls;
    int* flag;
    int nx, ny, nz, pitch;
    int iters;
    hipStream_t s;
};

static double time_ms(Ctx& c, const std::function<void(double*, double*)>& launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double* p = c.a;
    double* q = c.b;
    for (int i = 0; i < 3; ++i) {
        launch(p, q);
        std::swap(p, q);
    }
    CK(hipStreamSynchronize(c.s));
    CK(hipEventRecord(e0, c.s));
    for (int i = 0; i < c.iters; ++i) {
        launch(p, q);
        std::swap(p, q);
    }
    CK(hipEventRecord(e1, c.s));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / c.iters;
}

template <int RY, int NWX, int NWY, int X>
static void run_march(Ctx& c, const char* name, int zchunks, int lds_bytes = 0) {
    constexpr int WX = 128;
    wv::StreamArgs<double> a{};
    a.cls = c.cls;
    a.flag = c.flag;
    a.nx = c.nx;
    a.ny = c.ny;
    a.nz = c.nz;
    a.pitch = c.nx;
    a.cls_pitch = c.pitch;
    a.z_begin = 0;
    a.z_end = c.nz;
    a.zc = (c.nz + zchunks - 1) / zchunks;
    a.tiles_x = (c.nx + WX * NWX - 1) / (WX * NWX);
    a.tiles_y = (c.ny + RY * NWY - 1) / (RY * NWY);
    a.chunks_z = (c.nz + a.zc - 1) / a.zc;
    a.total_tiles = a.tiles_x * a.tiles_y * a.chunks_z;
    a.tiles_per_xcd = (a.total_tiles + 7) / 8;
    const unsigned grid = (unsigned)a.tiles_per_xcd * 8u;
    const double ms = time_ms(c, [&](double* prev, double* cur) {
        wv::StreamArgs<double> b = a;
        b.prev = prev;
        b.cur = cur;
        hipLaunchKernelGGL((wv::stream_march_kernel<double, RY, NWX, NWY, X>), dim3(grid), dim3(64 * NWX * NWY), lds_bytes, c.s, b);
    });
    const double bytes = 24.0 * c.nx * c.ny * c.nz;
    if (lds_bytes) printf("{\"lds_bytes\": %d}\n", lds_bytes);
    printf("{\"kernel\": \"march\", \"name\": \"%s\", \"ry\": %d, \"nwx\": %d, \"nwy\": %d, \"x\": %d, \"zchunks\": %d, \"ms\": %.4f, \"alg_gbs\": %.1f}\n",
           name, RY, NWX, NWY, X, zchunks, ms, bytes / ms / 1e6);
    fflush(stdout);
}

template <int RY, int NWX, int NWY, int X>
static void run_sweep(Ctx& c, const char* name, int stripe_rows) {
    constexpr int WX = 128;
    wv::StreamArgs<double> a{};
    a.cls = c.cls;
    a.flag = c.flag;
    a.nx = c.nx;
    a.ny = c.ny;
    a.nz = c.nz;
    a.pitch = c.nx;
    a.cls_pitch = c.pitch;
    a.z_begin = 0;
    a.z_end = c.nz;
    a.tiles_x = (c.nx + WX * NWX - 1) / (WX * NWX);
    a.stripe_rows = stripe_rows;
    a.tiles_y_stripe = (stripe_rows + RY * NWY - 1) / (RY * NWY);
    const int stripes = (c.ny + stripe_rows - 1) / stripe_rows;
    a.passes = (stripes + 7) / 8;
    const unsigned grid = 8u * (unsigned)a.passes * (unsigned)c.nz * (unsigned)(a.tiles_x * a.tiles_y_stripe);
    const double ms = time_ms(c, [&](double* prev, double* cur) {
        wv::StreamArgs<double> b = a;
        b.prev = prev;
        b.cur = cur;
        hipLaunchKernelGGL((wv::stream_sweep_nolds_kernel<double, RY, NWX, NWY, X>), dim3(grid), dim3(64 * NWX * NWY), 0, c.s, b);
    });
    const double bytes = 24.0 * c.nx * c.ny * c.nz;
    printf("{\"kernel\": \"sweep\", \"name\": \"%s\", \"ry\": %d, \"nwx\": %d, \"nwy\": %d, \"x\": %d, \"stripe_rows\": %d, \"ms\": %.4f, \"alg_gbs\": %.1f}\n",
           name, RY, NWX, NWY, X, stripe_rows, ms, bytes / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    Ctx c{};
    c.nx = c.ny = c.nz = n;
    c.iters = argc > 2 ? atoi(argv[2]) : 20;
    c.pitch = (n + 3) / 4;
    const int64_t N = (int64_t)n * n * n;
    CK(hipStreamCreate(&c.s));
    CK(hipMalloc((void**)&c.a, N * 8 + 256));
    CK(hipMalloc((void**)&c.b, N * 8 + 256 + (4 << 20)));
    CK(hipMalloc((void**)&c.cls, (int64_t)c.pitch * 4 * ((n + 3) / 4) * n + 16));
    CK(hipMemset(c.cls, 0, (int64_t)c.pitch * 4 * ((n + 3) / 4) * n + 16));
    CK(hipMalloc((void**)&c.flag, 4));
    CK(hipMemset(c.flag, 0, 4));
    hipLaunchKernelGGL(init_kernel, dim3(4096), dim3(256), 0, c.s, c.a, N, 1u);
    hipLaunchKernelGGL(init_kernel, dim3(4096), dim3(256), 0, c.s, c.b, N, 2u);
    hipLaunchKernelGGL(cls_kernel, dim3(4096), dim3(256), 0, c.s, c.cls, n, n, n, c.pitch);
    CK(hipStreamSynchronize(c.s));

    using namespace wv;
    constexpr int P = X_PRODUCT;
    if (argc > 3 && std::string(argv[3]) == "abl") {
        constexpr int S2 = X_NT_STORE | X_NT_PREV;
        for (int rep = 0; rep < 2; ++rep) {
            run_sweep<4, 1, 4, S2>(c, "base", 32);
            run_sweep<4, 1, 4, S2 | X_MUL_THIRD>(c, "mul_third", 32);
            run_sweep<4, 1, 4, S2 | X_NO_EDGE>(c, "no_edge", 32);
            run_sweep<4, 1, 4, S2 | X_NO_CLS>(c, "no_cls", 32);
            run_sweep<4, 1, 4, S2 | X_MUL_THIRD | X_NO_EDGE | X_NO_CLS>(c, "bare", 32);
            run_sweep<4, 1, 4, S2 | X_STORE_ALL>(c, "store_all", 32);
            run_sweep<4, 1, 4, S2 | X_STORE_ALL | X_MUL_THIRD>(c, "store_all_mul", 32);
            run_sweep<4, 2, 2, S2 | X_STORE_ALL>(c, "store_all", 32);
            run_sweep<4, 1, 4, S2 | X_STORE_ALL>(c, "store_all", 64);
        }
        return 0;
    }
    if (argc > 3 && std::string(argv[3]) == "prof2") {
        constexpr int S2 = X_NT_STORE | X_NT_PREV;
        run_sweep<4, 1, 4, S2>(c, "sweep", 16);
        run_sweep<4, 1, 4, S2>(c, "sweep", 32);
        run_sweep<4, 1, 4, S2>(c, "sweep", 64);
        run_sweep<4, 1, 4, S2>(c, "sweep", 128);
        run_sweep<4, 1, 4, S2 | X_NT_BELOW>(c, "nt_below", 32);
        run_sweep<4, 1, 4, S2 | X_NT_BELOW>(c, "nt_below", 64);
        run_sweep<4, 1, 4, S2 | X_NT_BELOW | X_NT_MID>(c, "nt_below_mid", 64);
        run_sweep<4, 1, 4, S2 | X_NT_CUR>(c, "nt_above", 64);
        run_sweep<4, 1, 4, 0>(c, "no_nt", 64);
        run_sweep<4, 1, 4, X_NT_STORE>(c, "nt_store_only", 64);
        run_sweep<4, 1, 4, X_NT_PREV>(c, "nt_prev_only", 64);
        run_sweep<2, 1, 4, S2>(c, "sweep", 64);
        run_sweep<4, 4, 1, S2>(c, "sweep", 64);
        run_sweep<4, 8, 1, S2>(c, "sweep", 64);
        return 0;
    }
    if (argc > 3 && std::string(argv[3]) == "prof") {
        // short list for rocprofv3 --pmc passes
        double ms = time_ms(c, [&](double* p, double* q) {
            hipLaunchKernelGGL(triad_chunk_kernel<4>, dim3((unsigned)((N / 2 + 1023) / 1024)), dim3(256), 0, c.s, (v2d*)p, (const v2d*)q, N / 2);
        });
        printf("{\"kernel\": \"triad_chunk_nt\", \"ms\": %.4f}\n", ms);
        run_march<2, 1, 4, P>(c, "product", 32);
        run_march<2, 4, 1, P>(c, "product", 32);
        run_march<2, 8, 1, P>(c, "product", 32);
        run_march<2, 4, 2, P>(c, "product", 32);
        run_sweep<4, 1, 4, X_NT_STORE | X_NT_PREV>(c, "sweep", 64);
        run_sweep<4, 1, 4, X_NT_STORE | X_NT_PREV>(c, "sweep", 128);
        run_sweep<4, 4, 1, X_NT_STORE | X_NT_PREV>(c, "sweep", 32);
        return 0;
    }

    // ---- (a) bandwidth references: grid-stride vs block-contiguous chunks
    for (int grid : {16384, 65536}) {
        double ms = time_ms(c, [&](double* p, double* q) {
            hipLaunchKernelGGL(copy_kernel<true>, dim3(grid), dim3(256), 0, c.s, (v2d*)p, (const v2d*)q, N / 2);
        });
        printf("{\"kernel\": \"copy_nt\", \"grid\": %d, \"ms\": %.4f, \"gbs\": %.1f}\n", grid, ms, 16.0 * N / ms / 1e6);
        ms = time_ms(c, [&](double* p, double* q) {
            hipLaunchKernelGGL(triad_kernel<true>, dim3(grid), dim3(256), 0, c.s, (v2d*)p, (const v2d*)q, N / 2);
        });
        printf("{\"kernel\": \"triad_inplace_nt\", \"grid\": %d, \"ms\": %.4f, \"gbs\": %.1f}\n", grid, ms, 24.0 * N / ms / 1e6);
    }
    {
        const int64_t nvec = N / 2;
        auto chunk = [&](auto kern, int u, const char* nm) {
            const int64_t per_block = 256LL * u;
            const unsigned grid = (unsigned)((nvec + per_block - 1) / per_block);
            double ms = time_ms(c, [&](double* p, double* q) {
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, c.s, (v2d*)p, (const v2d*)q, nvec);
            });
            printf("{\"kernel\": \"%s\", \"per_thread\": %d, \"ms\": %.4f, \"gbs\": %.1f}\n", nm, u, ms, 24.0 * N / ms / 1e6);
            fflush(stdout);
        };
        chunk(triad_chunk_kernel<1>, 1, "triad_chunk_nt");
        chunk(triad_chunk_kernel<2>, 2, "triad_chunk_nt");
        chunk(triad_chunk_kernel<3>, 3, "triad_chunk_nt");
        chunk(triad_chunk_kernel<4>, 4, "triad_chunk_nt");
        chunk(triad_chunk_kernel<6>, 6, "triad_chunk_nt");
    }

    constexpr int S = X_NT_STORE | X_NT_PREV;
    for (int sr : {16, 32, 64, 128}) {
        run_sweep<2, 1, 1, S>(c, "sweep", sr);
        run_sweep<4, 1, 1, S>(c, "sweep", sr);
        run_sweep<2, 1, 4, S>(c, "sweep", sr);
        run_sweep<4, 1, 4, S>(c, "sweep", sr);
        run_sweep<2, 4, 1, S>(c, "sweep", sr);
        run_sweep<4, 4, 1, S>(c, "sweep", sr);
        run_sweep<4, 8, 1, S>(c, "sweep", sr);
    }
    run_sweep<4, 1, 4, 0>(c, "sweep_no_nt", 64);
    run_sweep<4, 1, 4, S | X_NT_CUR>(c, "sweep_nt_cur", 64);
    // ---- (d) does the relative placement of the two fields matter (DRAM bank aliasing)?
    for (int64_t off : {65536LL}) {
        Ctx d = c;
        d.b = (double*)((char*)c.b + off);
        printf("{\"offset\": %lld}\n", (long long)off);
        run_march<2, 1, 4, P>(d, "offset", 32);
        const int64_t nvec = N / 2;
        double ms = time_ms(d, [&](double* p, double* q) {
            hipLaunchKernelGGL(triad_chunk_kernel<4>, dim3((unsigned)((nvec + 1023) / 1024)), dim3(256), 0, d.s, (v2d*)p, (const v2d*)q, nvec);
        });
        printf("{\"kernel\": \"triad_chunk_nt\", \"offset\": %lld, \"ms\": %.4f, \"gbs\": %.1f}\n", (long long)off, ms, 24.0 * N / ms / 1e6);
    }
    return 0;
}
