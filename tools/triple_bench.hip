// tools/triple_bench.hip -- PROTOTYPE (not product code): THREE time steps of the 7-point update per pass over the fields.
//
// Why: the engine's two-step pass (wayverb_amd/csrc/pair_kernels.hip.h) runs at the memory system's speed for its 32 B per
// node, and its instruction stream alone takes 2.73 ms of the 5.87 ms (tools/pair_tune, "no loads, no stores").  A pass
// that produces t+2 AND t+3 from (t-1, t) moves the same 32 B per node for three updates -- 10.7 B per node-update, the one
// way past SURVEY 8(d)'s 333 Gnode-updates/s line -- if its state fits on the chip: level k of a strip of RY rows keeps
// three planes of RY + 2 (3 - k) rows.  This prototype prices the form that fits: RY = 2, the rings of
// `current` (3 x 8 rows) and t+2 (3 x 4 rows) in registers, the ring of t+1 (3 x 6 rows) in LDS (144 KB of the CU's 160),
// 12 row updates per plane for 2 rows x 3 levels of output.  Interior only, like round 1's tools/pair_bench.hip was for
// the two-step pass: every node takes the update, what lies off the grid counts as 0, no class map, no flags.  It checks
// itself against three plain steps on a small mesh, bit for bit, then times 1024^3.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/triple_bench.hip -o tools/triple_bench
//   tools/triple_bench [n=1024] [iters=5]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
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

using wv::Vec16;
typedef Vec16<double>::type V;
constexpr int VX = 2;
constexpr int RY = 2;             // rows per strip
constexpr int R0 = RY + 6;        // rows of `current` per plane (level 0)
constexpr int R1 = RY + 4;        // rows of t+1 and of `previous`
constexpr int R2 = RY + 2;        // rows of t+2
constexpr int KE = R1 + R2 + RY;  // edge rows per plane: current rows feeding t+1, t+1 rows feeding t+2, t+2 rows feeding t+3
constexpr int NWMAX = 8;

struct TripleArgs {
    const double* prev;  // t-1
    const double* cur;   // t
    double* out2;        // t+2
    double* out3;        // t+3
    int ny, nz, pitch, nw;
    int zc, chunks, strips, strips_per_xcd;
};

// LDS: the t+1 ring, one 16-byte slot per lane: [plane slot][row][thread]; then the x-edge words
extern __shared__ char triple_lds[];

template <int X>  // X & 1: no loads, no stores (instruction stream only); X & 2: touch the lines of the NEXT front plane a trip ahead
__global__ void __launch_bounds__(64 * NWMAX) triple_march_kernel(const TripleArgs a) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int threads = 64 * a.nw;
    V* ring1 = reinterpret_cast<V*>(triple_lds);                                  // [3][R1][threads]
    double(*sl)[KE][NWMAX] = reinterpret_cast<double(*)[KE][NWMAX]>(triple_lds + (size_t)3 * R1 * threads * sizeof(V));
    double(*sr)[KE][NWMAX] = sl + 2;

    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int strip = xcd * a.strips_per_xcd + j % a.strips_per_xcd;
    const int chunk = j / a.strips_per_xcd;
    if (strip >= a.strips || chunk >= a.chunks) return;
    const int y0 = strip * RY;
    const int zb = chunk * a.zc, ze = min(zb + a.zc, a.nz);
    if (zb >= ze) return;

    wv::PairTile<double> t;
    t.ny = a.ny;
    t.nz = a.nz;
    t.pitch = a.pitch;
    t.plane = (int64_t)a.pitch * a.ny;
    t.col = (wave * 64 + lane) * VX;
    const wv::PairEdges<double, KE> edges{sl, sr, lane, wave, a.nw};
    auto in_grid = [&](int y, int z) { return y >= 0 && y < a.ny && z >= 0 && z < a.nz; };
    auto made_up = [&](int y, int z) -> V {
        V v;
        for (int k = 0; k < VX; ++k) v[k] = double(y) * 0.001 + double(z + k + lane);
        return v;
    };
    auto load0 = [&](V(&dst)[R0], int z) {
#pragma unroll
        for (int q = 0; q < R0; ++q) dst[q] = (X & 1) ? made_up(y0 - 3 + q, z) : t.load(a.cur, y0 - 3 + q, z);
    };
    auto loadp = [&](V(&dst)[R1], int z) {
#pragma unroll
        for (int q = 0; q < R1; ++q) dst[q] = (X & 1) ? made_up(y0 - 2 + q, -z) : t.load(a.prev, y0 - 2 + q, z);
    };
    auto r1 = [&](int slot, int row) -> V& { return ring1[((size_t)slot * R1 + row) * threads + tid]; };

    // Register rings: three planes of `current`, three of t+2.  t+1 lives in LDS (slot = plane mod 3).
    V bA[R0], bB[R0], bC[R0];
    V uA[R2], uB[R2], uC[R2];
    V pv[R1];
    bool seen = false;
    float touched = 0.f;  // (X & 2) keeps the touching loads alive

    // One trip: the front plane f = z + 3 comes in; t+1 on plane f-1, t+2 on plane f-2, t+3 on plane f-3 = z come out.
    // b_lo / b_mid = current(f-2) / current(f-1), b_new receives current(f);
    // u_lo / u_mid = t+2(f-4) / t+2(f-3), u_new receives t+2(f-2).
    auto trip = [&](int f, int set, const V(&b_lo)[R0], const V(&b_mid)[R0], V(&b_new)[R0], const V(&u_lo)[R2], const V(&u_mid)[R2],
                    V(&u_new)[R2]) {
        const int z = f - 3;
        load0(b_new, f);
        loadp(pv, f - 1);
        // X & 2: one dword per 128-byte line of the rows the NEXT trip loads, issued behind this trip's loads and consumed
        // only after this trip's arithmetic: by then the lines are in L2 and the next trip's loads do not wait for HBM
        float pf = 0.f;
        if (X & 2) {
            const int col = wave * 128 + (lane & 7) * 16;
#pragma unroll
            for (int q = 0; q < R0; ++q) {
                const int y = y0 - 3 + q;
                if (y >= 0 && y < a.ny && f + 1 >= 0 && f + 1 < a.nz)
                    pf += reinterpret_cast<const float*>(a.cur + (int64_t)(f + 1) * t.plane + (int64_t)y * a.pitch + col)[0];
            }
#pragma unroll
            for (int q = 0; q < R1; ++q) {
                const int y = y0 - 2 + q;
                if (y >= 0 && y < a.ny && f >= 0 && f < a.nz)
                    pf += reinterpret_cast<const float*>(a.prev + (int64_t)f * t.plane + (int64_t)y * a.pitch + col)[0];
            }
        }
        const int fo = f + 3;  // (f starts at zb - 1 >= -1: keep the modulus away from negative numbers)
        const int s_lo = (fo + 1) % 3, s_mid = (fo + 2) % 3, s_new = fo % 3;  // LDS slots of t+1(f-3), t+1(f-2), t+1(f-1)
        // x edges: current(f-1) rows feeding t+1(f-1); t+1(f-2) rows feeding t+2(f-2); t+2(f-3) rows feeding t+3(f-3)
#pragma unroll
        for (int q = 0; q < R1; ++q) edges.publish(set, q, b_mid[q + 1]);
        V t_mid[R1];  // t+1(f-2), all six rows: centre plane of the t+2 update
#pragma unroll
        for (int q = 0; q < R1; ++q) t_mid[q] = r1(s_mid, q);
#pragma unroll
        for (int q = 0; q < R2; ++q) edges.publish(set, R1 + q, t_mid[q + 1]);
#pragma unroll
        for (int r = 0; r < RY; ++r) edges.publish(set, R1 + R2 + r, u_mid[r + 1]);
        wv::lds_barrier();
        // level 1: t+1 on plane f-1, rows y0-2 .. y0+RY+1
        V t_new[R1];
#pragma unroll
        for (int q = 0; q < R1; ++q) {
            const V v = wv::pair_step_row<double>(b_mid[q + 1], b_mid[q], b_mid[q + 2], b_lo[q + 1], b_new[q + 1], pv[q], edges.left(set, q),
                                                  edges.right(set, q));
            t_new[q] = in_grid(y0 - 2 + q, f - 1) ? v : (V)(0.0);
            r1(s_new, q) = t_new[q];
        }
        // level 2: t+2 on plane f-2, rows y0-1 .. y0+RY; its own old value is current(f-2)
#pragma unroll
        for (int q = 0; q < R2; ++q) {
            const V zm = r1(s_lo, q + 1);
            const V v = wv::pair_step_row<double>(t_mid[q + 1], t_mid[q], t_mid[q + 2], zm, t_new[q + 1], b_lo[q + 2], edges.left(set, R1 + q),
                                                  edges.right(set, R1 + q));
            u_new[q] = in_grid(y0 - 1 + q, f - 2) ? v : (V)(0.0);
        }
        // level 3: t+3 on plane z = f-3, rows y0 .. y0+RY-1; its own old value is t+1(z)
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const V old1 = r1(s_lo, r + 2);
            const V v3 = wv::pair_step_row<double>(u_mid[r + 1], u_mid[r], u_mid[r + 2], u_lo[r + 1], u_new[r + 1], old1, edges.left(set, R1 + R2 + r),
                                                   edges.right(set, R1 + R2 + r));
            if (z >= zb && z < ze && y0 + r < a.ny) {
                if (X & 1) {
                    seen = seen || !wv::is_finite(v3[0]) || !wv::is_finite(v3[1]);
                    if (seen) {
                        t.store(a.out2, y0 + r, z, u_mid[r + 1]);
                        t.store(a.out3, y0 + r, z, v3);
                    }
                } else {
                    t.store(a.out2, y0 + r, z, u_mid[r + 1]);
                    t.store(a.out3, y0 + r, z, v3);
                }
            }
        }
        if (X & 2) touched += pf;
    };

    // prologue: current(zb-3), current(zb-2) in place; the t+1 / t+2 rings hold zeros (what lies below the first planes
    // a chunk needs is either off the grid -- zero is right -- or recomputed by the warm-up trips before anything is stored)
#pragma unroll
    for (int q = 0; q < R1; ++q)
        for (int s = 0; s < 3; ++s) r1(s, q) = (V)(0.0);
#pragma unroll
    for (int q = 0; q < R2; ++q) uA[q] = uB[q] = uC[q] = (V)(0.0);
    load0(bA, zb - 3);
    load0(bB, zb - 2);
    // trips f = zb-1 .. ze+2; six per loop turn so that the register rings rotate without copies and the LDS edge set is
    // the trip's parity.  (f - (zb-1)) mod 3 selects the ring roles.
    for (int f = zb - 1; f <= ze + 2; f += 6) {
        trip(f, 0, bA, bB, bC, uA, uB, uC);
        if (f + 1 <= ze + 2) trip(f + 1, 1, bB, bC, bA, uB, uC, uA);
        if (f + 2 <= ze + 2) trip(f + 2, 0, bC, bA, bB, uC, uA, uB);
        if (f + 3 <= ze + 2) trip(f + 3, 1, bA, bB, bC, uA, uB, uC);
        if (f + 4 <= ze + 2) trip(f + 4, 0, bB, bC, bA, uB, uC, uA);
        if (f + 5 <= ze + 2) trip(f + 5, 1, bC, bA, bB, uC, uA, uB);
    }
    if ((X & 2) && touched == 1.2345e-30f) a.out3[0] = touched;  // (never)
}

// one plain step of every node: next = (sum of six neighbours, off-grid = 0) / 3 - prev, the reference's order
__global__ void plain_step_kernel(const double* prev, const double* cur, double* next, int nx, int ny, int nz, int pitch) {
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
        double s = 0.0 + (x > 0 ? cur[i - 1] : 0.0);
        s += (x + 1 < nx ? cur[i + 1] : 0.0);
        s += (y > 0 ? cur[i - pitch] : 0.0);
        s += (y + 1 < ny ? cur[i + pitch] : 0.0);
        s += (z > 0 ? cur[i - plane] : 0.0);
        s += (z + 1 < nz ? cur[i + plane] : 0.0);
        s = wv::div3(s);
        s -= prev[i];
        next[i] = s;
    }
}

__global__ void init_kernel(double* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        p[i] = ((double)(h & 0xFFFF) / 65536.0 - 0.5) * 0.5;
    }
}

static TripleArgs make_args(const double* prev, const double* cur, double* o2, double* o3, int nx, int ny, int nz, int chunks) {
    TripleArgs a{};
    a.prev = prev;
    a.cur = cur;
    a.out2 = o2;
    a.out3 = o3;
    a.ny = ny;
    a.nz = nz;
    a.pitch = nx;
    a.nw = nx / 128;
    a.zc = (nz + chunks - 1) / chunks;
    a.chunks = (nz + a.zc - 1) / a.zc;
    a.strips = (ny + RY - 1) / RY;
    a.strips_per_xcd = (a.strips + 7) / 8;
    return a;
}

template <int X>
static void launch(const TripleArgs& a) {
    const size_t lds = (size_t)3 * R1 * 64 * a.nw * sizeof(V) + (size_t)4 * KE * NWMAX * sizeof(double);
    static bool set = false;
    if (!set) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&triple_march_kernel<X>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        set = true;
    }
    const unsigned grid = 8u * (unsigned)a.strips_per_xcd * (unsigned)a.chunks;
    hipLaunchKernelGGL((triple_march_kernel<X>), dim3(grid), dim3(64u * (unsigned)a.nw), lds, 0, a);
}

static bool check(int nx, int ny, int nz, int chunks) {
    const int64_t N = (int64_t)nx * ny * nz;
    double *A, *B, *T1, *T2, *T3, *O2, *O3;
    for (double** p : {&A, &B, &T1, &T2, &T3, &O2, &O3}) CK(hipMalloc((void**)p, N * 8 + 256));
    hipLaunchKernelGGL(init_kernel, dim3(1024), dim3(256), 0, 0, A, N, 11u);
    hipLaunchKernelGGL(init_kernel, dim3(1024), dim3(256), 0, 0, B, N, 22u);
    hipLaunchKernelGGL(plain_step_kernel, dim3(1024), dim3(256), 0, 0, A, B, T1, nx, ny, nz, nx);
    hipLaunchKernelGGL(plain_step_kernel, dim3(1024), dim3(256), 0, 0, B, T1, T2, nx, ny, nz, nx);
    hipLaunchKernelGGL(plain_step_kernel, dim3(1024), dim3(256), 0, 0, T1, T2, T3, nx, ny, nz, nx);
    CK(hipMemset(O2, 0xFF, N * 8));
    CK(hipMemset(O3, 0xFF, N * 8));
    launch<0>(make_args(A, B, O2, O3, nx, ny, nz, chunks));
    CK(hipDeviceSynchronize());
    std::vector<double> t2(N), t3(N), o2(N), o3(N);
    CK(hipMemcpy(t2.data(), T2, N * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(t3.data(), T3, N * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o2.data(), O2, N * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o3.data(), O3, N * 8, hipMemcpyDeviceToHost));
    const bool ok = std::memcmp(t2.data(), o2.data(), N * 8) == 0 && std::memcmp(t3.data(), o3.data(), N * 8) == 0;
    int64_t bad2 = 0, bad3 = 0, first = -1;
    for (int64_t i = 0; i < N; ++i) {
        if (std::memcmp(&t2[i], &o2[i], 8)) ++bad2;
        if (std::memcmp(&t3[i], &o3[i], 8)) {
            ++bad3;
            if (first < 0) first = i;
        }
    }
    printf("check %dx%dx%d, %d chunk(s): t+2 %lld and t+3 %lld of %lld values differ from three plain steps%s\n", nx, ny, nz, chunks,
           (long long)bad2, (long long)bad3, (long long)N, ok ? " -- bit-identical" : "");
    if (first >= 0)
        printf("   first t+3 difference at x %lld y %lld z %lld\n", (long long)(first % nx), (long long)((first / nx) % ny), (long long)(first / ((int64_t)nx * ny)));
    for (double* p : {A, B, T1, T2, T3, O2, O3}) CK(hipFree(p));
    return ok;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 5;
    bool ok = check(256, 22, 19, 1);
    ok = check(128, 9, 40, 3) && ok;
    ok = check(1024, 12, 14, 2) && ok;
    if (!ok) return 1;
    const int nx = (n + 127) / 128 * 128;
    if (nx / 128 > NWMAX) {
        printf("rows of more than %d waves are not in this prototype\n", NWMAX);
        return 0;
    }
    const int64_t N = (int64_t)nx * n * n;
    double *A, *B, *O2, *O3;
    for (double** p : {&A, &B, &O2, &O3}) CK(hipMalloc((void**)p, N * 8 + 256));
    hipLaunchKernelGGL(init_kernel, dim3(4096), dim3(256), 0, 0, A, N, 1u);
    hipLaunchKernelGGL(init_kernel, dim3(4096), dim3(256), 0, 0, B, N, 2u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int chunks : {1, 2, 4}) {
        const TripleArgs a = make_args(A, B, O2, O3, nx, n, n, chunks);
        for (int variant = 0; variant < 3; ++variant) {
            for (int it = 0; it < iters + 2; ++it) {
                if (it == 2) CK(hipEventRecord(e0));
                if (variant == 0)
                    launch<0>(a);
                else if (variant == 1)
                    launch<1>(a);
                else
                    launch<2>(a);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters;
            printf("%d^3, %d chunk(s), %s: %.3f ms per pass of THREE steps = %.3f ms per step = %.1f Gnode-updates/s (32 B per node: %.0f GB/s)\n", n,
                   a.chunks, variant == 1 ? "instructions only (no loads, no stores)" : (variant == 2 ? "three-step pass, next plane's lines touched a trip ahead" : "three-step pass"), ms, ms / 3, 3.0 * N / ms / 1e6,
                   32.0 * N / ms / 1e6);
        }
    }
    return 0;
}
