// tools/pair_bench.hip -- standalone pricing harness (NOT product code): what would two time steps
// per pass cost?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/pair_bench.hip -o tools/pair_bench
//   tools/pair_bench [ny=1024] [nz=1024] [iters=10]        (nx is fixed at 1024: one workgroup row)
//
// The product sweep moves 24 B per node and step (read previous, read current, write next).  A pass
// that produces steps t+1 AND t+2 from (t-1, t) reads two fields and writes two: 32 B per node for
// two steps.  It can stay a short-lived, plane-sweeping workgroup (no dependency between
// workgroups) if every workgroup recomputes the t+1 values it needs itself:
//   t+2 on the tile of plane z  needs  t+1 on the tile of planes z-1, z+1 and on tile + 1 ring of z,
//   which need  `current` on planes z-2 .. z+2 (tile, +1 ring, +2 rings, +1 ring, tile) and
//   `previous` on planes z-1 .. z+1 -- all read-only inputs; outputs go to two other buffers.
// That is 3x the arithmetic (free) and more loads per workgroup than two single steps (42 vs 36
// rows per 4-row strip), most of them L2 hits, against 1/3 less HBM traffic.  This harness prices
// exactly that trade on the interior update (no classes, no boundary nodes, outside = 0), against
// two passes of the single-step kernel written the same way, and checks the two agree bit for bit.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e__ = (x);                                                                 \
        if (e__ != hipSuccess) {                                                              \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__);    \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

typedef double V __attribute__((ext_vector_type(2)));
constexpr int NX = 1024, NW = 8, RY = 4;

__device__ __forceinline__ double from_below(double edge, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_above(double edge, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

struct Args {
    const double* prev;  // t-1
    const double* cur;   // t
    double* out1;        // t+1
    double* out2;        // t+2 (pair kernel only)
    int ny, nz, stripe_rows, strips_per_stripe;
    int nz_total;  // single2_kernel: planes in the field (nz then counts plane pairs)
};

struct Tile {
    int lane, wave, y0, z, ny, nz;
    int64_t col;
    __device__ __forceinline__ V load(const double* p, int y, int zz, bool nt = false) const {
        if (y < 0 || y >= ny || zz < 0 || zz >= nz) return (V)(0.0);
        const V* q = reinterpret_cast<const V*>(p + ((int64_t)zz * ny + y) * NX + col);
        return nt ? __builtin_nontemporal_load(q) : *q;
    }
    __device__ __forceinline__ void store(double* p, int y, int zz, V v) const {
        __builtin_nontemporal_store(v, reinterpret_cast<V*>(p + ((int64_t)zz * ny + y) * NX + col));
    }
};

// ((left + right + ym + yp + zm + zp) / 3) - pv, in the product kernel's operation order
__device__ __forceinline__ V step_row(V c0, V ym, V yp, V zm, V zp, V pv, double edge_l, double edge_r) {
    V out;
    {
        double s = 0.0 + from_below(edge_l, c0.y);
        s += c0.y;
        s += ym.x;
        s += yp.x;
        s += zm.x;
        s += zp.x;
        s = s / 3.0;
        out.x = s - pv.x;
    }
    {
        double s = 0.0 + c0.x;
        s += from_above(edge_r, c0.x);
        s += ym.y;
        s += yp.y;
        s += zm.y;
        s += zp.y;
        s = s / 3.0;
        out.y = s - pv.y;
    }
    return out;
}

__device__ __forceinline__ bool map_block(const Args& a, Tile& t) {
    t.lane = threadIdx.x & 63;
    t.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int strip = j % a.strips_per_stripe;
    j /= a.strips_per_stripe;
    t.z = j % a.nz;
    const int stripe = (j / a.nz) * 8 + xcd;
    t.y0 = stripe * a.stripe_rows + strip * RY;
    t.ny = a.ny;
    t.nz = a.nz;
    t.col = (int64_t)t.wave * 128 + t.lane * 2;
    return t.y0 < a.ny;
}

// x-edge exchange between the 8 waves of the workgroup: K rows per call
template <int K>
__device__ __forceinline__ void exchange_edges(const V (&rows)[K], double (&el)[K], double (&er)[K], int lane, int wave,
                                               double (*sl)[NW], double (*sr)[NW]) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (lane == 0) sl[k][wave] = rows[k].x;
        if (lane == 63) sr[k][wave] = rows[k].y;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        el[k] = wave > 0 ? sr[k][wave - 1] : 0.0;
        er[k] = wave + 1 < NW ? sl[k][wave + 1] : 0.0;
    }
    __syncthreads();
}

// ---- baseline: one step per pass, same structure -------------------------------------------------
__global__ void __launch_bounds__(64 * NW) single_kernel(const Args a) {
    __shared__ double sl[RY][NW], sr[RY][NW];
    Tile t;
    const bool alive = map_block(a, t);
    V mid[RY + 2], below[RY], above[RY], pv[RY];
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) mid[r] = alive ? t.load(a.cur, t.y0 - 1 + r, t.z) : (V)(0.0);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        above[r] = alive ? t.load(a.cur, t.y0 + r, t.z + 1) : (V)(0.0);
        pv[r] = alive ? t.load(a.prev, t.y0 + r, t.z, true) : (V)(0.0);
        below[r] = alive ? t.load(a.cur, t.y0 + r, t.z - 1) : (V)(0.0);
    }
    V own[RY];
    double el[RY], er[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) own[r] = mid[r + 1];
    exchange_edges<RY>(own, el, er, t.lane, t.wave, sl, sr);
    if (!alive) return;
#pragma unroll
    for (int r = 0; r < RY; ++r)
        if (t.y0 + r < a.ny)
            t.store(a.out1, t.y0 + r, t.z, step_row(mid[r + 1], mid[r], mid[r + 2], below[r], above[r], pv[r], el[r], er[r]));
}

// ---- one step, two planes per workgroup: plane z+1's `below` is plane z's `mid` ----------------------
// (does issuing fewer loads per output row pay?  28 row loads per 8 output rows instead of 2 x 18)
__global__ void __launch_bounds__(64 * NW) single2_kernel(const Args a) {
    __shared__ double sl[2 * RY][NW], sr[2 * RY][NW];
    Tile t;
    bool alive = map_block(a, t);
    const int z = t.z * 2;  // the grid is launched over plane pairs
    t.nz = a.nz_total;
    alive = alive && z < a.nz_total;
    V m0[RY + 2], m1[RY + 2], below[RY], above[RY], p0[RY], p1[RY];
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) {
        m0[r] = alive ? t.load(a.cur, t.y0 - 1 + r, z) : (V)(0.0);
        m1[r] = alive ? t.load(a.cur, t.y0 - 1 + r, z + 1) : (V)(0.0);
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        above[r] = alive ? t.load(a.cur, t.y0 + r, z + 2) : (V)(0.0);
        p0[r] = alive ? t.load(a.prev, t.y0 + r, z, true) : (V)(0.0);
        p1[r] = alive ? t.load(a.prev, t.y0 + r, z + 1, true) : (V)(0.0);
        below[r] = alive ? t.load(a.cur, t.y0 + r, z - 1) : (V)(0.0);
    }
    V own[2 * RY];
    double el[2 * RY], er[2 * RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        own[r] = m0[r + 1];
        own[RY + r] = m1[r + 1];
    }
    exchange_edges<2 * RY>(own, el, er, t.lane, t.wave, sl, sr);
    if (!alive) return;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        if (t.y0 + r < a.ny) {
            t.store(a.out1, t.y0 + r, z, step_row(m0[r + 1], m0[r], m0[r + 2], below[r], m1[r + 1], p0[r], el[r], er[r]));
            if (z + 1 < a.nz_total)
                t.store(a.out1, t.y0 + r, z + 1,
                        step_row(m1[r + 1], m1[r], m1[r + 2], m0[r + 1], above[r], p1[r], el[RY + r], er[RY + r]));
        }
    }
}

// ---- two steps per pass --------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * NW) pair_kernel(const Args a) {
    __shared__ double sl[RY + 2][NW], sr[RY + 2][NW];
    Tile t;
    const bool alive = map_block(a, t);
    const int y0 = t.y0, z = t.z;
    // `current`: planes z-2 .. z+2, rows tile / +1 ring / +2 rings / +1 ring / tile
    V bm2[RY], bm1[RY + 2], b0[RY + 4], bp1[RY + 2], bp2[RY];
    V am1[RY], a0[RY + 2], ap1[RY];
#pragma unroll
    for (int r = 0; r < RY + 4; ++r) b0[r] = alive ? t.load(a.cur, y0 - 2 + r, z) : (V)(0.0);
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) {
        bm1[r] = alive ? t.load(a.cur, y0 - 1 + r, z - 1) : (V)(0.0);
        bp1[r] = alive ? t.load(a.cur, y0 - 1 + r, z + 1) : (V)(0.0);
        a0[r] = alive ? t.load(a.prev, y0 - 1 + r, z) : (V)(0.0);
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        bp2[r] = alive ? t.load(a.cur, y0 + r, z + 2) : (V)(0.0);  // first touch: HBM
        bm2[r] = alive ? t.load(a.cur, y0 + r, z - 2) : (V)(0.0);
        am1[r] = alive ? t.load(a.prev, y0 + r, z - 1) : (V)(0.0);
        ap1[r] = alive ? t.load(a.prev, y0 + r, z + 1) : (V)(0.0);
    }

    // t+1 on plane z, tile + 1 ring in y (rows y0-1 .. y0+RY)
    V t0[RY + 2];
    {
        V rows[RY + 2];
        double el[RY + 2], er[RY + 2];
#pragma unroll
        for (int q = 0; q < RY + 2; ++q) rows[q] = b0[q + 1];
        exchange_edges<RY + 2>(rows, el, er, t.lane, t.wave, sl, sr);
#pragma unroll
        for (int q = 0; q < RY + 2; ++q) {
            const int y = y0 - 1 + q;
            t0[q] = (y >= 0 && y < a.ny) ? step_row(b0[q + 1], b0[q], b0[q + 2], bm1[q], bp1[q], a0[q], el[q], er[q])
                                         : (V)(0.0);
        }
    }
    // t+1 on planes z-1 and z+1, tile rows
    V tm1[RY], tp1[RY];
    {
        V rows[RY];
        double el[RY], er[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) rows[r] = bm1[r + 1];
        exchange_edges<RY>(rows, el, er, t.lane, t.wave, sl, sr);
#pragma unroll
        for (int r = 0; r < RY; ++r)
            tm1[r] = z - 1 >= 0 ? step_row(bm1[r + 1], bm1[r], bm1[r + 2], bm2[r], b0[r + 2], am1[r], el[r], er[r]) : (V)(0.0);
#pragma unroll
        for (int r = 0; r < RY; ++r) rows[r] = bp1[r + 1];
        exchange_edges<RY>(rows, el, er, t.lane, t.wave, sl, sr);
#pragma unroll
        for (int r = 0; r < RY; ++r)
            tp1[r] = z + 1 < a.nz ? step_row(bp1[r + 1], bp1[r], bp1[r + 2], b0[r + 2], bp2[r], ap1[r], el[r], er[r]) : (V)(0.0);
    }
    // t+2 on the tile
    V rows[RY];
    double el[RY], er[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) rows[r] = t0[r + 1];
    exchange_edges<RY>(rows, el, er, t.lane, t.wave, sl, sr);
    if (!alive) return;
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        if (y0 + r < a.ny) {
            t.store(a.out1, y0 + r, z, t0[r + 1]);
            t.store(a.out2, y0 + r, z, step_row(t0[r + 1], t0[r], t0[r + 2], tm1[r], tp1[r], b0[r + 2], el[r], er[r]));
        }
    }
}

// ---- two steps per pass, register-resident z-march -------------------------------------------------
// One workgroup owns a strip of RY rows x 1024 columns and marches it through `zc` planes.  It keeps
// three planes of `current` (RY+4 rows) and three planes of t+1 (RY+2 rows) in registers, so per
// plane it loads one plane of `current` (RY+4 rows) and one of `previous` (RY+2 rows) and stores one
// plane of t+1 and one of t+2: 14 row loads + 8 row stores per RY = 4 output rows and TWO steps.
struct MarchArgs {
    const double* prev;
    const double* cur;
    double* out1;
    double* out2;
    int ny, nz, zc, chunks;
    int xcd_map;
};

__global__ void __launch_bounds__(64 * NW) pair_march_kernel(const MarchArgs a) {
    __shared__ double sl[2 * RY + 2][NW], sr[2 * RY + 2][NW];
    Tile t;
    t.lane = threadIdx.x & 63;
    t.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    t.ny = a.ny;
    t.nz = a.nz;
    t.col = (int64_t)t.wave * 128 + t.lane * 2;
    // XCD k (= blockIdx % 8) takes a contiguous eighth of the strips, so that the y rings two neighbouring
    // strips both need are fetched once into that XCD's L2 (the strips march in step)
    const int strips = a.ny / RY, per_xcd = (strips + 7) / 8;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int strip = a.xcd_map ? xcd * per_xcd + j % per_xcd : (int)(blockIdx.x / a.chunks);
    const int chunk = a.xcd_map ? j / per_xcd : (int)(blockIdx.x % a.chunks);
    if (strip >= strips || chunk >= a.chunks) return;
    const int y0 = strip * RY;
    const int zb = chunk * a.zc, ze = min(zb + a.zc, a.nz);
    t.y0 = y0;

    auto load_b = [&](V (&dst)[RY + 4], int z) {
#pragma unroll
        for (int q = 0; q < RY + 4; ++q) dst[q] = t.load(a.cur, y0 - 2 + q, z);
    };
    // t+1 on plane z (rows y0-1 .. y0+RY) from b(z-1), b(z), b(z+1) and `previous`(z); el/er: x edges of b(z) rows
    auto level1 = [&](V (&dst)[RY + 2], const V (&bm)[RY + 4], const V (&b0)[RY + 4], const V (&bp)[RY + 4], int z,
                      const double* el, const double* er) {
#pragma unroll
        for (int q = 0; q < RY + 2; ++q) {
            const int y = y0 - 1 + q;
            const V pv = t.load(a.prev, y, z, true);
            dst[q] = (y >= 0 && y < a.ny && z >= 0 && z < a.nz)
                             ? step_row(b0[q + 1], b0[q], b0[q + 2], bm[q + 1], bp[q + 1], pv, el[q], er[q])
                             : (V)(0.0);
        }
    };

    V b_lo[RY + 4], b_mid[RY + 4], b_hi[RY + 4];  // `current` on planes z-1, z, z+1 while producing t+1(z)
    V t_prev[RY + 2], t_cur[RY + 2], t_next[RY + 2];
    V rows[2 * RY + 2];
    double el[2 * RY + 2], er[2 * RY + 2];

    // prologue: t+1 on planes zb-1 and zb
    load_b(b_lo, zb - 2);
    load_b(b_mid, zb - 1);
    load_b(b_hi, zb);
#pragma unroll
    for (int q = 0; q < RY + 2; ++q) rows[q] = b_mid[q + 1];
#pragma unroll
    for (int q = RY + 2; q < 2 * RY + 2; ++q) rows[q] = (V)(0.0);
    exchange_edges<2 * RY + 2>(rows, el, er, t.lane, t.wave, sl, sr);
    level1(t_prev, b_lo, b_mid, b_hi, zb - 1, el, er);
#pragma unroll
    for (int q = 0; q < RY + 4; ++q) {
        b_lo[q] = b_mid[q];
        b_mid[q] = b_hi[q];
    }
    load_b(b_hi, zb + 1);
#pragma unroll
    for (int q = 0; q < RY + 2; ++q) rows[q] = b_mid[q + 1];
    exchange_edges<2 * RY + 2>(rows, el, er, t.lane, t.wave, sl, sr);
    level1(t_cur, b_lo, b_mid, b_hi, zb, el, er);

    for (int z = zb; z < ze; ++z) {
        // now: b_lo = b(z-1), b_mid = b(z), b_hi = b(z+1); t_prev = t+1(z-1), t_cur = t+1(z)
        V b_nn[RY + 4];
        load_b(b_nn, z + 2);
        // one exchange for both levels: x edges of b(z+1) rows (for t+1(z+1)) and of t+1(z) tile rows (for t+2(z))
#pragma unroll
        for (int q = 0; q < RY + 2; ++q) rows[q] = b_hi[q + 1];
#pragma unroll
        for (int r = 0; r < RY; ++r) rows[RY + 2 + r] = t_cur[r + 1];
        exchange_edges<2 * RY + 2>(rows, el, er, t.lane, t.wave, sl, sr);
        level1(t_next, b_mid, b_hi, b_nn, z + 1, el, er);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            if (y0 + r < a.ny) {
                t.store(a.out1, y0 + r, z, t_cur[r + 1]);
                t.store(a.out2, y0 + r, z,
                        step_row(t_cur[r + 1], t_cur[r], t_cur[r + 2], t_prev[r + 1], t_next[r + 1], b_mid[r + 2], el[RY + 2 + r],
                                 er[RY + 2 + r]));
            }
        }
#pragma unroll
        for (int q = 0; q < RY + 2; ++q) {
            t_prev[q] = t_cur[q];
            t_cur[q] = t_next[q];
        }
#pragma unroll
        for (int q = 0; q < RY + 4; ++q) {
            b_lo[q] = b_mid[q];
            b_mid[q] = b_hi[q];
            b_hi[q] = b_nn[q];
        }
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

int main(int argc, char** argv) {
    const int ny = argc > 1 ? atoi(argv[1]) : 1024, nz = argc > 2 ? atoi(argv[2]) : 1024, iters = argc > 3 ? atoi(argv[3]) : 10;
    const int64_t N = (int64_t)NX * ny * nz;
    double *A, *B, *A1, *C, *S1, *S2;
    for (double** p : {&A, &B, &A1, &C, &S1, &S2}) CK(hipMalloc((void**)p, N * 8));
    hipLaunchKernelGGL(init_kernel, dim3(4096), dim3(256), 0, 0, A, N, 1u);
    hipLaunchKernelGGL(init_kernel, dim3(4096), dim3(256), 0, 0, B, N, 2u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int stripe_rows : {64, 32, 16}) {
        if (ny % stripe_rows) continue;
        Args a{A, B, S1, nullptr, ny, nz, stripe_rows, stripe_rows / RY};
        const int stripes = ny / stripe_rows, passes = (stripes + 7) / 8;
        const unsigned grid = 8u * passes * nz * (stripe_rows / RY);
        // two single steps: (A, B) -> S1 = t+1 ; (B, S1) -> S2 = t+2
        Args s1 = a, s2 = a;
        s2.prev = B;
        s2.cur = S1;
        s2.out1 = S2;
        float ms_single = 0, ms_pair = 0;
        for (int it = 0; it < iters + 2; ++it) {
            if (it == 2) CK(hipEventRecord(e0));
            hipLaunchKernelGGL(single_kernel, dim3(grid), dim3(64 * NW), 0, 0, s1);
            hipLaunchKernelGGL(single_kernel, dim3(grid), dim3(64 * NW), 0, 0, s2);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_single, e0, e1));
        {
            // one step with two planes per workgroup (grid over nz / 2 plane pairs; map_block's z = pair index)
            Args h = a;
            h.out1 = C;   // scratch here: the pair pass below overwrites it
            Args g = h;
            g.nz = (nz + 1) / 2;   // map_block cycles over this many "planes"
            g.nz_total = nz;
            float ms2 = 0;
            const unsigned grid2 = 8u * passes * ((nz + 1) / 2) * (stripe_rows / RY);
            for (int it = 0; it < iters + 2; ++it) {
                if (it == 2) CK(hipEventRecord(e0));
                hipLaunchKernelGGL(single2_kernel, dim3(grid2), dim3(64 * NW), 0, 0, g);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms2, e0, e1));
            printf("stripe %3d  one step, 1 plane per workgroup %.3f ms   2 planes per workgroup %.3f ms\n", stripe_rows,
                   ms_single / iters / 2, ms2 / iters);
        }
        Args p = a;
        p.out1 = A1;
        p.out2 = C;
        for (int it = 0; it < iters + 2; ++it) {
            if (it == 2) CK(hipEventRecord(e0));
            hipLaunchKernelGGL(pair_kernel, dim3(grid), dim3(64 * NW), 0, 0, p);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_pair, e0, e1));
        CK(hipGetLastError());
        printf("stripe %3d  two single steps %.3f ms  one pair pass %.3f ms  ratio %.3f  (%.1f / %.1f Gnode-updates/s)\n",
               stripe_rows, ms_single / iters, ms_pair / iters, ms_single / ms_pair, 2.0 * N / (ms_single / iters) / 1e6,
               2.0 * N / (ms_pair / iters) / 1e6);
    }
    // agreement (bit for bit) on a sample of planes
    std::vector<double> h1((size_t)NX * ny), h2((size_t)NX * ny);
    int bad = 0;
    // the z-march form of the pair pass (after the plane-sweep pair pass has been checked below, A1 / C are reused)
    auto check = [&](const char* what) {
        int wrong = 0;
        for (int z : {0, 1, nz / 2, nz / 2 + 1, nz - 2, nz - 1}) {
            for (auto pr : {std::make_pair(S1, A1), std::make_pair(S2, C)}) {
                CK(hipMemcpy(h1.data(), pr.first + (int64_t)z * ny * NX, h1.size() * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(h2.data(), pr.second + (int64_t)z * ny * NX, h2.size() * 8, hipMemcpyDeviceToHost));
                if (memcmp(h1.data(), h2.data(), h1.size() * 8) != 0) ++wrong;
            }
        }
        printf(wrong ? "MISMATCH: %s differs from two single steps on %d sampled planes\n"
                     : "%s == two single steps (sampled planes, bitwise)%.0d\n",
               what, wrong);
        return wrong;
    };
    bad += check("plane-sweep pair pass");
    for (int cfg = 0; cfg < 8; ++cfg) {
        const int zc = (const int[]){1024, 128, 64, 32}[cfg % 4], xcd_map = cfg / 4;
        if (zc > nz && zc != 1024) continue;
        CK(hipMemset(A1, 0xFF, N * 8));
        CK(hipMemset(C, 0xFF, N * 8));
        MarchArgs m{A, B, A1, C, ny, nz, std::min(zc, nz), (nz + std::min(zc, nz) - 1) / std::min(zc, nz), xcd_map};
        const unsigned grid = xcd_map ? 8u * (unsigned)(((ny / RY + 7) / 8) * m.chunks) : (unsigned)((ny / RY) * m.chunks);
        float ms = 0;
        for (int it = 0; it < iters + 2; ++it) {
            if (it == 2) CK(hipEventRecord(e0));
            hipLaunchKernelGGL(pair_march_kernel, dim3(grid), dim3(64 * NW), 0, 0, m);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGetLastError());
        printf("z-march pair pass, %4d planes per workgroup, strips %s: %.3f ms per pair  (%.1f Gnode-updates/s)\n", m.zc,
               xcd_map ? "grouped per XCD" : "round robin", ms / iters, 2.0 * N / (ms / iters) / 1e6);
        bad += check("z-march pair pass");
    }
    printf("done, %d mismatching checks\n", bad);
    return bad ? 1 : 0;
}
